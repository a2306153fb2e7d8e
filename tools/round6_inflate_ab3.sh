O=gpurun_out/r06; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_inflate.py tests/test_bam_reader.py -m gpu -x -q 2>&1 | tail -2
for rep in 1 2; do for w in 1 0; do for cfg in "--genome 8000000" "--genome 8000000 --level 6 --tags 1" "--genome 8000000 --level 6 --tags 1 --quals 1" "--genome 4000000 --level 9"; do
  PA_INFLATE_WIDE=$w timeout 300 python tools/bench_inflate.py $cfg 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('${TAG} wide=$w', '$cfg', d['device_GBps_inflated'], d['kernel_ms'], d['sample_identical'])" | tee -a $O/inflate_walk_ab.txt
done; done; done
