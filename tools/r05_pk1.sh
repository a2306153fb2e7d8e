set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_realign.py tests/test_gpu_polish_chain.py -x -q > gpurun_out/r05/pk_tests.log 2>&1
tail -30 gpurun_out/r05/pk_tests.log
timeout 120 python tools/realign_stages.py 1500 > gpurun_out/r05/pk_stages.log 2>&1
PA_REALIGN_SINGLE=1 timeout 120 python tools/realign_stages.py 1500 > gpurun_out/r05/pk_stages_single.log 2>&1
cat gpurun_out/r05/pk_stages.log gpurun_out/r05/pk_stages_single.log
python tools/bench_polish_chain.py make_fast /tmp/pc 4000000 > gpurun_out/r05/chain_make.log 2>&1
timeout 900 python tools/bench_polish_chain.py run /tmp/pc 1,8 > gpurun_out/r05/pk_chain_bench.json 2> gpurun_out/r05/pk_chain_bench.err
PA_REALIGN_SINGLE=1 timeout 900 python tools/bench_polish_chain.py run /tmp/pc 1,8 > gpurun_out/r05/pk_chain_bench_single.json 2> gpurun_out/r05/pk_chain_bench_single.err
cat gpurun_out/r05/pk_chain_bench.json gpurun_out/r05/pk_chain_bench_single.json
