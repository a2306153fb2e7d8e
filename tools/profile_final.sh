# End-of-round evidence: full GPU suite, smoke, headline + secondary benches, rocprofv3 kernel stats.
set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout 600 python -m pytest tests -q -m gpu > gpurun_out/r01_final_gpu_tests.log 2>&1; tail -3 gpurun_out/r01_final_gpu_tests.log
timeout 120 python __graft_entry__.py smoke > gpurun_out/r01_final_smoke.log 2>&1; tail -1 gpurun_out/r01_final_smoke.log
timeout 400 python bench.py > gpurun_out/r01_bench_variant.json 2> gpurun_out/r01_bench_variant.err
timeout 400 python bench.py --model polish --steps 3 --warmup 1 > gpurun_out/r01_bench_polish.json 2> gpurun_out/r01_bench_polish.err
timeout 300 python bench.py --model realign --steps 10 --warmup 3 --cpu-seconds 10 > gpurun_out/r01_bench_realign.json 2> gpurun_out/r01_bench_realign.err
timeout 300 python tools/bench_pipeline.py --files 8 --windows 262144 > gpurun_out/r01_pipeline.json 2> gpurun_out/r01_pipeline.err
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r01f_stats -o variant -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/r01f_stats.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r01f_stats_polish -o polish -- python $R/bench.py --model polish --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r01f_stats_polish.log 2>&1
cd $R; tail -c 300 gpurun_out/r01_bench_variant.json; tail -1 gpurun_out/r01_pipeline.json
