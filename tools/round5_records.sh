# Round 5 records: the GPU suite, smoke(), the default bench line, a kernel trace of the headline.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
( time timeout 1500 python -m pytest tests -q -m gpu -x ) > gpurun_out/r05/final_gpu_tests.log 2>&1
tail -3 gpurun_out/r05/final_gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r05/final_smoke.log 2>&1; tail -1 gpurun_out/r05/final_smoke.log
( time python bench.py > gpurun_out/r05/final_bench.json 2> gpurun_out/r05/final_bench.err ) 2>&1 | tail -3
python - <<PY
import json
d=json.loads(open('gpurun_out/r05/final_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['roofline']['frac'], d['cpu_baseline']['value'])
for k,v in d['secondary'].items():
    if isinstance(v,dict): print(k, v.get('value'), v.get('unit'), v.get('seconds'), v.get('stage_walls'), (v.get('roofline') or {}).get('frac'))
PY
