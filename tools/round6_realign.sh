# Round 6 (second session): the re-aligner with the 8-bit pass proven away and the strip size chosen on the device -- the chain /
# re-aligner tests, then the polish chain bench (16 Mb at 60x, 16 workers) with each switched off in turn.  GPU; outputs under gpurun_out/r06/.
O=gpurun_out/r06; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_polish_chain.py tests/test_gpu_realign.py -m gpu -x -q > $O/realign_tests.log 2>&1; tail -5 $O/realign_tests.log
python tools/bench_polish_chain.py make_fast /tmp/pb 16000000 60 > $O/mk16.log 2>&1 || tail -3 $O/mk16.log
for rep in 1 2; do for cfg in "1 1" "1 0" "0 0"; do set -- $cfg
  PA_REALIGN_ADAPT=$1 PA_REALIGN_PROOF=$2 timeout 600 python tools/bench_polish_chain.py run /tmp/pb 16 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['runs'][0]; s=r['stage_seconds_summed_over_workers']
print('adapt=$1 proof=$2', r['seconds'], r['mb_draft_per_s'], r['counts'], {k:s[k] for k in s if 'chain' in k})" | tee -a $O/realign_adapt_ab.txt
done; done
