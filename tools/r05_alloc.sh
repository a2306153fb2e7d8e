cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05 /dev/shm/e2e
timeout 600 python -m pytest tests/test_gpu_pipeline.py -x -q -m gpu -k "fused" 2>&1 | tail -3
for h in 2 1; do
PEPPER_AMD_FUSED_HANDLES=$h timeout 900 python tools/bench_e2e.py call_variant_fused /dev/shm/e2e/cv 256000000 30 3 > gpurun_out/r05/cvf_h$h.json 2> gpurun_out/r05/cvf_h$h.err
python - <<PY
import json
d=json.load(open('gpurun_out/r05/cvf_h$h.json'))
print('handles $h', d['runs_seconds'])
for w in d['runs_stage_walls']: print('   ', w)
PY
grep "FUSED:" gpurun_out/r05/cvf_h$h.err | tail -2
done
