set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE -d $R/gpurun_out/r01_pmc_realign -o realign -- python $R/tools/realign_stages.py 1500 > $R/gpurun_out/r01_pmc_realign.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_LDS -d $R/gpurun_out/r01_pmc_realign2 -o realign -- python $R/tools/realign_stages.py 1500 > $R/gpurun_out/r01_pmc_realign2.log 2>&1
ls $R/gpurun_out/r01_pmc_realign* ; tail -3 $R/gpurun_out/r01_pmc_realign.log
