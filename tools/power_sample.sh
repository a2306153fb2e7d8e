# Debug aid: sample the board power / shader clock (rocm-smi) while a device-resident bench pass runs.
#   bash tools/power_sample.sh variant|polish [steps]
MODEL=${1:-variant}
STEPS=${2:-1500}
export TMPDIR=/tmp
( timeout 150 python bench.py --model $MODEL --resident-only --no-cpu-baseline --steps $STEPS --warmup 5 > /tmp/ps_bench.log 2>&1 ) &
BP=$!
for i in $(seq 1 ${SAMPLES:-44}); do
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | python3 -c "
import sys,re
t=sys.stdin.read()
p=re.search(r'Package Power \(W\): ([\d.]+)',t); c=re.search(r'sclk clock level: \S+ \((\d+)Mhz\)',t); j=re.search(r'junction\) \(C\): ([\d.]+)',t)
print('W=%s sclk=%s Tj=%s'%(p and p.group(1), c and c.group(1), j and j.group(1)))"
  sleep 0.4
  kill -0 $BP 2>/dev/null || break
done
wait $BP
python3 -c "
import json
d=json.loads(open('/tmp/ps_bench.log').read().strip().splitlines()[-1])
print('$MODEL', d['value'], {k:v['avg_ms'] for k,v in d['kernels'].items()})"
