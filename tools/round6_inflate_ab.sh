# Round 6: the inflate kernel's one-window and two-window steps side by side: tests under both, then four workloads each, interleaved.
O=gpurun_out/r06; mkdir -p $O
for w in 1 0; do PA_INFLATE_WIDE=$w timeout 600 python -m pytest tests/test_gpu_inflate.py tests/test_bam_reader.py -m gpu -x -q 2>&1 | tail -2; done
for rep in 1 2; do for w in 0 1; do for cfg in "--genome 8000000" "--genome 8000000 --level 6 --tags 1" "--genome 8000000 --level 6 --tags 1 --quals 1" "--genome 4000000 --level 9"; do
  PA_INFLATE_WIDE=$w timeout 300 python tools/bench_inflate.py $cfg 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('wide=$w', '$cfg', d['device_GBps_inflated'], d['kernel_ms'], d['members'], d['sample_identical'])" | tee -a $O/inflate_wide_ab${TAG}.txt
done; done; done
for w in 0 1; do PA_INFLATE_WIDE=$w PA_INFLATE_DEBUG=1 timeout 300 python tools/bench_inflate.py --genome 8000000 2>&1 > /dev/null | grep "^inflate:" | tail -1; done
