# Round 6: the inflate kernel at 6 and at 7 wavefronts per SIMD (72 registers: 7 spilled), three workloads each, interleaved.
O=gpurun_out/r06; mkdir -p $O
for rep in 1 2; do for w in 6 7; do for cfg in "--genome 8000000" "--genome 4000000 --level 6 --tags 1" "--genome 8000000 --level 6 --tags 1" "--genome 4000000 --level 6 --tags 1 --quals 1"; do
  PA_INFLATE_WAVES=$w timeout 300 python tools/bench_inflate.py $cfg 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('waves=$w', '$cfg', d['device_GBps_inflated'], d['kernel_ms'], d['members'], d['sample_identical'])" | tee -a $O/inflate_ab.txt
done; done; done
PA_INFLATE_WAVES=7 timeout 600 python -m pytest tests/test_gpu_inflate.py -m gpu -x -q 2>&1 | tail -2
