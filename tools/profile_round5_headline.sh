# Round-5 record of the headline's kernels (unchanged since round 2): kernel statistics + PMC passes of the device-resident
# pass, each counter set in its own run (tools/profile_round3_models.sh, variant part only).
set -x
R=$GRAFT_REPO_ROOT
TAG=r05
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
PROF="python $R/bench.py --resident-only --no-cpu-baseline --no-secondary --steps 6 --warmup 2"
timeout 120 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_stats -o variant -- $PROF > $R/gpurun_out/${TAG}_stats.log 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/${TAG}_fetch -o variant -- $PROF > $R/gpurun_out/${TAG}_fetch.log 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/${TAG}_write -o variant -- $PROF > $R/gpurun_out/${TAG}_write.log 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $R/gpurun_out/${TAG}_mfma -o variant -- $PROF > $R/gpurun_out/${TAG}_mfma.log 2>&1
cd $R
python tools/pmc_summary.py --model variant --units 16384 --out gpurun_out/${TAG}_variant --command "rocprofv3 --kernel-trace [--stats | --pmc FETCH_SIZE | --pmc WRITE_SIZE | --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE] -- python bench.py --resident-only --no-cpu-baseline --no-secondary --steps 6 --warmup 2" gpurun_out/${TAG}_stats gpurun_out/${TAG}_fetch gpurun_out/${TAG}_write gpurun_out/${TAG}_mfma > /dev/null
find gpurun_out -name "*.db" -delete
head -9 gpurun_out/${TAG}_variant_kernel_stats.txt; ls gpurun_out | grep ${TAG}_
