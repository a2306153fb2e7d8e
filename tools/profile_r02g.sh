set -x
cd $GRAFT_REPO_ROOT
timeout 250 python -m pytest tests/test_gpu_variant.py tests/test_gpu_polish.py tests/test_gpu_arith_modes.py tests/test_gpu_shapes.py -q -m gpu > gpurun_out/r02g_tests.log 2>&1; tail -3 gpurun_out/r02g_tests.log
for BCOL in 0 1; do
PA_BIAS_COLUMN=$BCOL timeout 100 python bench.py --resident-only --no-cpu-baseline --steps 12 --warmup 3 > gpurun_out/r02g_variant_bc$BCOL.json 2>/dev/null
PA_BIAS_COLUMN=$BCOL timeout 100 python bench.py --model polish --resident-only --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/r02g_polish_bc$BCOL.json 2>/dev/null
done
python - <<'PY'
import json
for f in ("variant_bc0","variant_bc1","polish_bc0","polish_bc1"):
    d=json.loads(open("gpurun_out/r02g_%s.json"%f).read().strip().splitlines()[-1])
    print(f, round(d["value"]), {k:v["avg_ms"] for k,v in d["kernels"].items()})
PY
