# Round 6, first GPU visit: the parity suite, the bench line as the driver runs it (is it < 6 KB and parseable?), the 2-rank line with
# the per-rank image legs on one shared GPU (plumbing).
R=$(pwd); O=gpurun_out/r06; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/gpu_tests.log 2>&1; tail -5 $O/gpu_tests.log
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err; tail -n 1 $O/bench.json | wc -c
cp gpurun_out/bench_full.json $O/bench_full.json 2>/dev/null
PEPPER_AMD_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --steps 4 --warmup 1 --no-cpu-baseline --no-extras > $O/bench_2rank_shared.json 2> $O/bench_2rank_shared.err; tail -n 1 $O/bench_2rank_shared.json | head -c 3000; tail -5 $O/bench_2rank_shared.err
