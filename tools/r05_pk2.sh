set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
for n in 1500 8000; do
timeout 120 python tools/realign_stages.py $n > gpurun_out/r05/pk2_stages_$n.log 2>&1
PA_REALIGN_SINGLE=1 timeout 120 python tools/realign_stages.py $n > gpurun_out/r05/pk2_stages_single_$n.log 2>&1
cat gpurun_out/r05/pk2_stages_$n.log gpurun_out/r05/pk2_stages_single_$n.log
done
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $R/gpurun_out/r05/pmc_pk -o pk -- python $R/tools/realign_stages.py 8000 > $R/gpurun_out/r05/pmc_pk.log 2>&1
PA_REALIGN_SINGLE=1 timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $R/gpurun_out/r05/pmc_single -o single -- python $R/tools/realign_stages.py 8000 > $R/gpurun_out/r05/pmc_single.log 2>&1
cd $R
python tools/rocprof_db_summary.py gpurun_out/r05/pmc_pk > gpurun_out/r05/pmc_pk.txt 2>&1
python tools/rocprof_db_summary.py gpurun_out/r05/pmc_single > gpurun_out/r05/pmc_single.txt 2>&1
grep -A8 "sw_ends" gpurun_out/r05/pmc_pk.txt | head -40; grep -A8 "sw_ends" gpurun_out/r05/pmc_single.txt | head -40
