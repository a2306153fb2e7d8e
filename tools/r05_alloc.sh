cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05 /dev/shm/pc16
timeout 900 python -m pytest tests/test_gpu_realign.py tests/test_gpu_polish_chain.py -x -q -m gpu 2>&1 | tail -4
timeout 300 python bench.py --model realign --steps 10 --warmup 3 --cpu-seconds 1 > gpurun_out/r05/realign_diet.json 2> gpurun_out/r05/realign_diet.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r05/realign_diet.json').read().strip().splitlines()[-1])
print('realign', d['value'], d.get('roofline'))
print({k:v for k,v in d.items() if 'reads_per' in k or 'ms_per' in k})
PY
python tools/bench_polish_chain.py make_fast /dev/shm/pc16 16000000 > gpurun_out/r05/chain_make.log 2>&1
timeout 600 python tools/bench_polish_chain.py run /dev/shm/pc16 1,16,16 > gpurun_out/r05/chain_diet.json 2> gpurun_out/r05/chain_diet.err
python - <<PY
import json
d=json.load(open('gpurun_out/r05/chain_diet.json'))
for r in d['runs']:
    s=r['stage_seconds_summed_over_workers']
    print(r['threads'], r['seconds'], r['mb_draft_per_s'], 'score', s.get('chain_score_kernel'), 'band', s.get('chain_band_kernel'))
PY
