export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_polish.py tests/test_gpu_arith_modes.py tests/test_gpu_variant.py tests/test_gpu_shapes.py -m gpu -x -q 2>&1 | tail -3
for model in polish variant; do
timeout 200 python bench.py --model $model --resident-only --no-cpu-baseline --steps 20 --warmup 3 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$model', d['value'], d['ms_per_step'], d['roofline']['frac'])
for k,v in d.get('kernels',{}).items(): print('   ',k,v['avg_ms'],v['launches_per_step'])
"
done
