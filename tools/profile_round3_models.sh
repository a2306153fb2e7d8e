# Round-3 rocprofv3 evidence for the two model paths (kernels unchanged since round 2; the per-round record the bench's
# roofline block points at): kernel statistics + PMC passes of the device-resident pass, each counter set in its own run.
set -x
R=$GRAFT_REPO_ROOT
TAG=${TAG:-r03}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
PROF="python $R/bench.py --resident-only --no-cpu-baseline --no-secondary --steps 6 --warmup 2"
timeout 120 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_stats -o variant -- $PROF > $R/gpurun_out/${TAG}_stats.log 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/${TAG}_fetch -o variant -- $PROF > $R/gpurun_out/${TAG}_fetch.log 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/${TAG}_write -o variant -- $PROF > $R/gpurun_out/${TAG}_write.log 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $R/gpurun_out/${TAG}_mfma -o variant -- $PROF > $R/gpurun_out/${TAG}_mfma.log 2>&1
PROFP="python $R/bench.py --model polish --resident-only --no-cpu-baseline --no-secondary --steps 2 --warmup 1"
timeout 120 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_stats_polish -o polish -- $PROFP > $R/gpurun_out/${TAG}_stats_polish.log 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/${TAG}_fetch_polish -o polish -- $PROFP > $R/gpurun_out/${TAG}_fetch_polish.log 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/${TAG}_write_polish -o polish -- $PROFP > $R/gpurun_out/${TAG}_write_polish.log 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $R/gpurun_out/${TAG}_mfma_polish -o polish -- $PROFP > $R/gpurun_out/${TAG}_mfma_polish.log 2>&1
# the small-call schedule (512 windows per call)
timeout 120 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_stats_small -o small -- python $R/tools/small_batch_profile.py 512 > $R/gpurun_out/${TAG}_stats_small.log 2>&1
cd $R
python tools/pmc_summary.py --model variant --units 16384 --out gpurun_out/${TAG}_variant --command "rocprofv3 --kernel-trace [--stats | --pmc FETCH_SIZE | --pmc WRITE_SIZE | --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE] -- python bench.py --resident-only --no-cpu-baseline --no-secondary --steps 6 --warmup 2" gpurun_out/${TAG}_stats gpurun_out/${TAG}_fetch gpurun_out/${TAG}_write gpurun_out/${TAG}_mfma > /dev/null
python tools/pmc_summary.py --model polish --units 16384 --out gpurun_out/${TAG}_polish --command "rocprofv3 --kernel-trace [--stats | --pmc ...] -- python bench.py --model polish --resident-only --no-cpu-baseline --no-secondary --steps 2 --warmup 1" gpurun_out/${TAG}_stats_polish gpurun_out/${TAG}_fetch_polish gpurun_out/${TAG}_write_polish gpurun_out/${TAG}_mfma_polish > /dev/null
find gpurun_out/${TAG}_stats_small -name "*kernel_stats.csv" -exec cp {} gpurun_out/${TAG}_small_call_kernel_stats.csv \;
find gpurun_out -name "*.db" -delete
head -9 gpurun_out/${TAG}_variant_kernel_stats.txt; head -9 gpurun_out/${TAG}_polish_kernel_stats.txt; head -8 gpurun_out/${TAG}_small_call_kernel_stats.csv
ls gpurun_out | grep ${TAG}_ | head -30
