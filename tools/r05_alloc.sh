cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05 /dev/shm/e2e
for cfg in "1 0" "2 0" "4 0" "1 -1" "2 -1" "4 -1"; do
set -- $cfg
PEPPER_AMD_FUSED_HANDLES=$1 PEPPER_AMD_MODEL_STREAM_PRIORITY=$2 timeout 600 python tools/bench_e2e.py polish_fused /dev/shm/e2e/po 32000000 60 2 > gpurun_out/r05/pf.json 2> gpurun_out/r05/pf_$1_$2.err
python - <<PY
import json
d=json.load(open('gpurun_out/r05/pf.json'))
print('polish fused handles $1 priority $2', d['runs_seconds'], d['runs_stage_walls'][-1], 'fused_consensus', d['image_stage_seconds_summed_over_workers'].get('fused_consensus'))
PY
done
for cfg in "1 2" "2 2" "2 3"; do
set -- $cfg
PEPPER_AMD_FUSED_HANDLES=$1 PEPPER_AMD_FUSED_SELECTORS=$2 timeout 900 python tools/bench_e2e.py call_variant_fused /dev/shm/e2e/cv 256000000 30 3 > gpurun_out/r05/cvf.json 2> gpurun_out/r05/cvf_$1_$2.err
python - <<PY
import json
d=json.load(open('gpurun_out/r05/cvf.json'))
print('variant fused handles $1 selectors $2', d['runs_seconds'])
for w in d['runs_stage_walls'][-2:]: print('   ', w)
PY
done
