set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
PEPPER_AMD_BENCH_SHARE_GPU=2 timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 --no-secondary > gpurun_out/r05/dist2.json 2> gpurun_out/r05/dist2.err
echo rc=$?
tail -5 gpurun_out/r05/dist2.err | cut -c1-300
python - <<'PY'
import json
for line in open('gpurun_out/r05/dist2.json'):
    if line.startswith('{'):
        d=json.loads(line); print(d['value'], d['n_gpus'], d['config'].get('ranks_seen'), d['config'].get('collective_backend'))
PY
PEPPER_AMD_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 --no-secondary > gpurun_out/r05/dist1.json 2> gpurun_out/r05/dist1.err
echo rc=$?
python - <<'PY'
import json
for line in open('gpurun_out/r05/dist1.json'):
    if line.startswith('{'):
        d=json.loads(line); print(d['value'], d['n_gpus'], d['config'].get('ranks_seen'), d['config'].get('collective_backend'))
PY
