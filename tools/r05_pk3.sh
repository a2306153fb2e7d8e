set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_realign.py tests/test_gpu_polish_chain.py -x -q > gpurun_out/r05/pk_tests.log 2>&1
tail -5 gpurun_out/r05/pk_tests.log
for n in 1500 8000; do
timeout 120 python tools/realign_stages.py $n > gpurun_out/r05/pk3_stages_$n.log 2>&1
cat gpurun_out/r05/pk3_stages_$n.log
done
python tools/bench_polish_chain.py make_fast /tmp/pc 4000000 > gpurun_out/r05/chain_make.log 2>&1
timeout 900 python tools/bench_polish_chain.py run /tmp/pc 1,4,8 > gpurun_out/r05/pk3_chain_bench.json 2> gpurun_out/r05/pk3_chain_bench.err
cat gpurun_out/r05/pk3_chain_bench.json
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY -d $R/gpurun_out/r05/pmc_pk3 -o pk -- python $R/tools/realign_stages.py 8000 > $R/gpurun_out/r05/pmc_pk3.log 2>&1
cd $R
python tools/rocprof_db_summary.py gpurun_out/r05/pmc_pk3 > gpurun_out/r05/pmc_pk3.txt 2>&1
grep -B2 -A8 "sw_ends" gpurun_out/r05/pmc_pk3.txt | head -60
