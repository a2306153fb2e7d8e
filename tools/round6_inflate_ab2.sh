O=gpurun_out/r06; mkdir -p $O
for b in 48 64 100; do for cfg in "--genome 8000000" "--genome 8000000 --level 6 --tags 1" "--genome 8000000 --level 6 --tags 1 --quals 1" "--genome 4000000 --level 9"; do
  PA_INFLATE_WIDE=1 PA_INFLATE_WIDE_BELOW=$b timeout 300 python tools/bench_inflate.py $cfg 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('below=$b', '$cfg', d['device_GBps_inflated'], d['kernel_ms'], d['sample_identical'])" | tee -a $O/inflate_wide_below.txt
done; done
