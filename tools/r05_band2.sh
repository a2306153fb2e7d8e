set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_realign.py tests/test_gpu_polish_chain.py -x -q > gpurun_out/r05/band_tests.log 2>&1
tail -5 gpurun_out/r05/band_tests.log
PA_BAND_WAVES=1 PA_REALIGN_SINGLE=0 timeout 900 python -m pytest tests/test_gpu_realign.py tests/test_gpu_polish_chain.py -x -q > gpurun_out/r05/band_tests1.log 2>&1
tail -5 gpurun_out/r05/band_tests1.log
for n in 1500 8000; do
timeout 120 python tools/realign_stages.py $n > gpurun_out/r05/band2_stages_$n.log 2>&1
head -4 gpurun_out/r05/band2_stages_$n.log | tail -3
done
python tools/bench_polish_chain.py make_fast /tmp/pc 4000000 > gpurun_out/r05/chain_make.log 2>&1
timeout 900 python tools/bench_polish_chain.py run /tmp/pc 1,4,8 > gpurun_out/r05/band2_chain_bench.json 2> gpurun_out/r05/band2_chain_bench.err
python - <<PY
import json
d=json.load(open('gpurun_out/r05/band2_chain_bench.json'))
for r in d['runs']: print(r['threads'], r['mb_draft_per_s'], 'score', r['stage_seconds_summed_over_workers']['chain_score_kernel'], 'band', r['stage_seconds_summed_over_workers']['chain_band_kernel'], 'chain', r['stage_seconds_summed_over_workers']['chain'])
PY
timeout 300 python bench.py --model realign --steps 8 --warmup 2 --cpu-seconds 2 > gpurun_out/r05/band2_realign.json 2> gpurun_out/r05/band2_realign.err; cut -c1-1200 gpurun_out/r05/band2_realign.json
