set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05 /dev/shm/pc64
python tools/bench_polish_chain.py make_fast /dev/shm/pc64 64000000 > gpurun_out/r05/chain_make64.log 2>&1
timeout 900 python tools/bench_polish_chain.py run /dev/shm/pc64 8,16,16 > gpurun_out/r05/chain64.json 2> gpurun_out/r05/chain64.err
python - <<PY
import json
d=json.load(open('gpurun_out/r05/chain64.json'))
for r in d['runs']:
    s=r['stage_seconds_summed_over_workers']
    print(r['threads'], r['seconds'], r['mb_draft_per_s'], {k: s[k] for k in s})
PY
free -g | head -3; df -h /dev/shm | tail -1
