# Round-2 final evidence (HEAD as committed): full GPU suite, smoke, headline + polish bench lines, rocprofv3 kernel
# statistics + PMC passes of the device-resident pass.  Tight timeouts.
set -x
R=$GRAFT_REPO_ROOT
TAG=${TAG:-r02}
mkdir -p $R/gpurun_out
cd $R
timeout 300 python -m pytest tests -q -m gpu > gpurun_out/${TAG}_gpu_tests.log 2>&1; tail -4 gpurun_out/${TAG}_gpu_tests.log
timeout 100 python __graft_entry__.py smoke > gpurun_out/${TAG}_smoke.log 2>&1; tail -1 gpurun_out/${TAG}_smoke.log
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_variant.json 2> gpurun_out/${TAG}_bench_variant.err; tail -c 700 gpurun_out/${TAG}_bench_variant.json; tail -2 gpurun_out/${TAG}_bench_variant.err
timeout 300 python bench.py --model polish --steps 6 --warmup 1 --cpu-seconds 8 > gpurun_out/${TAG}_bench_polish.json 2> gpurun_out/${TAG}_bench_polish.err; tail -c 500 gpurun_out/${TAG}_bench_polish.json
PEPPER_AMD_BENCH_SHARE_GPU=1 timeout 200 python bench.py --gpus 2 --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_bench_2rank_shared.json 2> gpurun_out/${TAG}_bench_2rank_shared.err; tail -c 300 gpurun_out/${TAG}_bench_2rank_shared.json
timeout 200 python bench.py --model ns-literal --steps 3 --warmup 1 > gpurun_out/${TAG}_bench_ns_literal.json 2> gpurun_out/${TAG}_bench_ns_literal.err; tail -c 300 gpurun_out/${TAG}_bench_ns_literal.json
cd /tmp && export TMPDIR=/tmp
PROF="python $R/bench.py --resident-only --no-cpu-baseline --steps 6 --warmup 2"
timeout 100 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_stats -o variant -- $PROF > $R/gpurun_out/${TAG}_stats.log 2>&1
timeout 100 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/${TAG}_fetch -o variant -- $PROF > $R/gpurun_out/${TAG}_fetch.log 2>&1
timeout 100 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/${TAG}_write -o variant -- $PROF > $R/gpurun_out/${TAG}_write.log 2>&1
timeout 100 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $R/gpurun_out/${TAG}_mfma -o variant -- $PROF > $R/gpurun_out/${TAG}_mfma.log 2>&1
PROFP="python $R/bench.py --model polish --resident-only --no-cpu-baseline --steps 2 --warmup 1"
timeout 100 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_stats_polish -o polish -- $PROFP > $R/gpurun_out/${TAG}_stats_polish.log 2>&1
timeout 100 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/${TAG}_fetch_polish -o polish -- $PROFP > $R/gpurun_out/${TAG}_fetch_polish.log 2>&1
timeout 100 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/${TAG}_write_polish -o polish -- $PROFP > $R/gpurun_out/${TAG}_write_polish.log 2>&1
timeout 100 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $R/gpurun_out/${TAG}_mfma_polish -o polish -- $PROFP > $R/gpurun_out/${TAG}_mfma_polish.log 2>&1
cd $R
python tools/pmc_summary.py --model variant --units 16384 --out gpurun_out/${TAG}_variant --command "rocprofv3 --kernel-trace [--stats | --pmc FETCH_SIZE | --pmc WRITE_SIZE | --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE] -- python bench.py --resident-only --no-cpu-baseline --steps 6 --warmup 2" gpurun_out/${TAG}_stats gpurun_out/${TAG}_fetch gpurun_out/${TAG}_write gpurun_out/${TAG}_mfma > /dev/null
python tools/pmc_summary.py --model polish --units 16384 --out gpurun_out/${TAG}_polish --command "rocprofv3 --kernel-trace [--stats | --pmc ...] -- python bench.py --model polish --resident-only --no-cpu-baseline --steps 2 --warmup 1" gpurun_out/${TAG}_stats_polish gpurun_out/${TAG}_fetch_polish gpurun_out/${TAG}_write_polish gpurun_out/${TAG}_mfma_polish > /dev/null
find gpurun_out -name "*.db" -delete
head -9 gpurun_out/${TAG}_variant_kernel_stats.txt
