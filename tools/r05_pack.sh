set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_polish_chain.py tests/test_gpu_packed.py tests/test_gpu_pipeline.py -x -q > gpurun_out/r05/pack_tests.log 2>&1
tail -5 gpurun_out/r05/pack_tests.log
python tools/bench_polish_chain.py make_fast /tmp/pc 16000000 > gpurun_out/r05/chain_make.log 2>&1
for pr in 128 512 1024; do
PEPPER_AMD_POLISH_PACK_REGIONS=$pr timeout 900 python tools/bench_polish_chain.py run /tmp/pc 1,8,16 > gpurun_out/r05/pack_chain_bench_$pr.json 2> gpurun_out/r05/pack_chain_bench_$pr.err
python - <<PY
import json
d=json.load(open('gpurun_out/r05/pack_chain_bench_$pr.json'))
for r in d['runs']:
    s=r['stage_seconds_summed_over_workers']
    print('pack $pr', r['threads'], r['mb_draft_per_s'], 'score', s['chain_score_kernel'], 'band', s['chain_band_kernel'], 'chain', s['chain'], 'inflate', s.get('bam_inflate_device'), 'pack', s.get('bam_pack'), 'hdf5', s.get('hdf5'))
PY
done
