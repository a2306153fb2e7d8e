timeout 900 python -m pytest tests/test_gpu_polish.py -x -q 2>&1 | tail -15 > gpurun_out/r4_polish_small_tests.log
python bench.py --model polish --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r4_bench_polish_small.json 2> gpurun_out/r4_bench_polish_small.err
PA_POLISH_SMALL_MAX=0 python bench.py --model polish --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r4_bench_polish_big.json 2>> gpurun_out/r4_bench_polish_small.err
