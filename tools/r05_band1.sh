set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_realign.py tests/test_gpu_polish_chain.py -x -q > gpurun_out/r05/band_tests.log 2>&1
tail -5 gpurun_out/r05/band_tests.log
PA_BAND_WAVES=1 timeout 900 python -m pytest tests/test_gpu_realign.py tests/test_gpu_polish_chain.py -x -q > gpurun_out/r05/band_tests1.log 2>&1
tail -5 gpurun_out/r05/band_tests1.log
for w in 1 3; do for n in 1500 8000; do
PA_BAND_WAVES=$w timeout 120 python tools/realign_stages.py $n > gpurun_out/r05/band_stages_w${w}_$n.log 2>&1
head -4 gpurun_out/r05/band_stages_w${w}_$n.log | tail -3
done; done
python tools/bench_polish_chain.py make_fast /tmp/pc 4000000 > gpurun_out/r05/chain_make.log 2>&1
for w in 1 3; do
PA_BAND_WAVES=$w timeout 900 python tools/bench_polish_chain.py run /tmp/pc 1,4,8 > gpurun_out/r05/band_chain_bench_w$w.json 2> gpurun_out/r05/band_chain_bench_w$w.err
python - <<PY
import json
d=json.load(open('gpurun_out/r05/band_chain_bench_w$w.json'))
for r in d['runs']: print('band waves $w', r['threads'], r['mb_draft_per_s'], 'score', r['stage_seconds_summed_over_workers']['chain_score_kernel'], 'band', r['stage_seconds_summed_over_workers']['chain_band_kernel'], 'chain', r['stage_seconds_summed_over_workers']['chain'])
PY
done
