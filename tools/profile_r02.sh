# Round-2 evidence run (one gpurun call): GPU tests, smoke, headline bench, plumbing check of the N-rank launcher,
# rocprofv3 kernel stats + PMC passes of the device-resident pass.  Outputs under gpurun_out/r02_*.
set -x
R=$GRAFT_REPO_ROOT
TAG=${TAG:-r02}
mkdir -p $R/gpurun_out
cd $R
if [ "${SKIP_TESTS:-0}" != "1" ]; then
timeout 900 python -m pytest tests -q -m gpu > gpurun_out/${TAG}_gpu_tests.log 2>&1; tail -15 gpurun_out/${TAG}_gpu_tests.log
timeout 120 python __graft_entry__.py smoke > gpurun_out/${TAG}_smoke.log 2>&1; tail -1 gpurun_out/${TAG}_smoke.log
fi
timeout 500 python bench.py > gpurun_out/${TAG}_bench_variant.json 2> gpurun_out/${TAG}_bench_variant.err; tail -c 1500 gpurun_out/${TAG}_bench_variant.json; tail -3 gpurun_out/${TAG}_bench_variant.err
PEPPER_AMD_BENCH_SHARE_GPU=1 timeout 300 python bench.py --gpus 2 --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/${TAG}_bench_2rank_shared.json 2> gpurun_out/${TAG}_bench_2rank_shared.err; tail -c 400 gpurun_out/${TAG}_bench_2rank_shared.json; tail -3 gpurun_out/${TAG}_bench_2rank_shared.err
if [ "${SKIP_POLISH:-0}" != "1" ]; then
timeout 400 python bench.py --model polish --steps 6 --warmup 1 --cpu-seconds 8 > gpurun_out/${TAG}_bench_polish.json 2> gpurun_out/${TAG}_bench_polish.err; tail -c 1200 gpurun_out/${TAG}_bench_polish.json; tail -3 gpurun_out/${TAG}_bench_polish.err
fi
# A/B of the nt cache policy on the once-through streams of the step loops (PA_NT), device-resident pass
for NT in 0 1; do
PA_NT=$NT timeout 200 python bench.py --resident-only --no-cpu-baseline --steps 12 --warmup 3 > gpurun_out/${TAG}_bench_resident_nt$NT.json 2> gpurun_out/${TAG}_bench_resident_nt$NT.err; tail -c 900 gpurun_out/${TAG}_bench_resident_nt$NT.json
done
PA_NT=1 timeout 300 python -m pytest tests/test_gpu_variant.py tests/test_gpu_polish.py -q -m gpu > gpurun_out/${TAG}_gpu_tests_nt1.log 2>&1; tail -2 gpurun_out/${TAG}_gpu_tests_nt1.log
cd /tmp && export TMPDIR=/tmp
PROF="python $R/bench.py --resident-only --no-cpu-baseline --steps 6 --warmup 2"
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_stats -o variant -- $PROF > $R/gpurun_out/${TAG}_stats.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/${TAG}_fetch -o variant -- $PROF > $R/gpurun_out/${TAG}_fetch.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/${TAG}_write -o variant -- $PROF > $R/gpurun_out/${TAG}_write.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $R/gpurun_out/${TAG}_mfma -o variant -- $PROF > $R/gpurun_out/${TAG}_mfma.log 2>&1
if [ "${SKIP_POLISH:-0}" != "1" ]; then
PROFP="python $R/bench.py --model polish --resident-only --no-cpu-baseline --steps 2 --warmup 1"
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_stats_polish -o polish -- $PROFP > $R/gpurun_out/${TAG}_stats_polish.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/${TAG}_fetch_polish -o polish -- $PROFP > $R/gpurun_out/${TAG}_fetch_polish.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/${TAG}_write_polish -o polish -- $PROFP > $R/gpurun_out/${TAG}_write_polish.log 2>&1
fi
cd $R
# summaries are made here, on the box: the databases carry every loaded code object's symbols and can be tens of MB each
python tools/pmc_summary.py --model variant --units 16384 --out gpurun_out/${TAG}_variant --command "rocprofv3 --kernel-trace [--stats | --pmc ...] -- $PROF" \
    gpurun_out/${TAG}_stats gpurun_out/${TAG}_fetch gpurun_out/${TAG}_write gpurun_out/${TAG}_mfma > /dev/null
if [ "${SKIP_POLISH:-0}" != "1" ]; then
python tools/pmc_summary.py --model polish --units 16384 --out gpurun_out/${TAG}_polish --command "rocprofv3 --kernel-trace [--stats | --pmc ...] -- $PROFP" \
    gpurun_out/${TAG}_stats_polish gpurun_out/${TAG}_fetch_polish gpurun_out/${TAG}_write_polish > /dev/null
fi
ls -la gpurun_out/${TAG}_*/ | head -40
find gpurun_out -name "*.db" -size +2M -delete
head -12 gpurun_out/${TAG}_variant_kernel_stats.txt
