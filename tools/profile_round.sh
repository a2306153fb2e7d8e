set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout 400 python bench.py > gpurun_out/r01_bench_variant.json 2> gpurun_out/r01_bench_variant.err
timeout 400 python bench.py --model polish --steps 3 --warmup 1 > gpurun_out/r01_bench_polish.json 2> gpurun_out/r01_bench_polish.err
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r01b_stats -o variant -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/r01b_stats.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r01b_stats_polish -o polish -- python $R/bench.py --model polish --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r01b_stats_polish.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/r01b_fetch -o variant -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r01b_fetch.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/r01b_write -o variant -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r01b_write.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $R/gpurun_out/r01b_mfma -o variant -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r01b_mfma.log 2>&1
cd $R; ls -la gpurun_out/r01b_*/* | head -30; tail -c 600 gpurun_out/r01_bench_variant.json
