cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05 /dev/shm/e2e
PEPPER_AMD_LANE_TRACE=1 timeout 600 python tools/bench_e2e.py polish /dev/shm/e2e/po 64000000 60 1 > gpurun_out/r05/lanetrace.json 2> gpurun_out/r05/lanetrace.err
grep -n "lanes\]\|STEP\|lane " gpurun_out/r05/lanetrace.err | tail -60
python - <<PY
import json
d=json.load(open('gpurun_out/r05/lanetrace.json')); print(d['runs_seconds'], d['runs_stage_walls'])
PY
