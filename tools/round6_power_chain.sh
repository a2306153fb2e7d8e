# Round 6: board power and shader clock (rocm-smi every ~0.3 s) while the polish image chain runs with 16 workers on a 32 Mb / 60x job
# -- what clock do the alignment kernels get?  (The issue roof of the chain's roofline is priced at 2.4 GHz.)  GPU; output under gpurun_out/r06/.
O=gpurun_out/r06; mkdir -p $O
python tools/bench_polish_chain.py make_fast /tmp/pw 32000000 60 > /dev/null 2>&1
( timeout 200 python tools/bench_polish_chain.py run /tmp/pw 16 > /tmp/pw_run.log 2>/dev/null ) &
BP=$!
: > $O/power_chain.txt
for i in $(seq 1 120); do
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | python3 -c "
import sys,re,time
t=sys.stdin.read()
p=re.search(r'Package Power \(W\): ([\d.]+)',t); c=re.search(r'sclk clock level: \S+ \((\d+)Mhz\)',t); j=re.search(r'junction\) \(C\): ([\d.]+)',t)
print('t=%.1f W=%s sclk=%s Tj=%s'%(time.time()%1000, p and p.group(1), c and c.group(1), j and j.group(1)))" >> $O/power_chain.txt
  sleep 0.2
  kill -0 $BP 2>/dev/null || break
done
wait $BP
tail -1 /tmp/pw_run.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['runs'][0]; print('# chain', r['seconds'], 's', r['mb_draft_per_s'], 'Mb/s')" >> $O/power_chain.txt
cat $O/power_chain.txt
