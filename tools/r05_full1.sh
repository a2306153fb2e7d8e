set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
(time timeout 1500 python -m pytest tests -m gpu -x -q) > gpurun_out/r05/full_gpu_tests.log 2>&1
tail -8 gpurun_out/r05/full_gpu_tests.log
(time timeout 1500 python bench.py) > gpurun_out/r05/bench_full.json 2> gpurun_out/r05/bench_full.err
tail -3 gpurun_out/r05/bench_full.err
python - <<'PY'
import json
for line in open('gpurun_out/r05/bench_full.json'):
    if line.startswith('{'):
        d=json.loads(line)
        print(d['value'], d['ms_per_step'])
        for k,v in d.get('secondary',{}).items():
            print(k, json.dumps(v)[:400])
PY
