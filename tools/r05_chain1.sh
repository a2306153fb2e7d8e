set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_polish_chain.py -x -q > gpurun_out/r05/chain_tests.log 2>&1
tail -30 gpurun_out/r05/chain_tests.log
python tools/bench_polish_chain.py make_fast /tmp/pc 4000000 > gpurun_out/r05/chain_make.log 2>&1
timeout 900 python tools/bench_polish_chain.py run /tmp/pc 1,4,8 > gpurun_out/r05/chain_bench.json 2> gpurun_out/r05/chain_bench.err
tail -5 gpurun_out/r05/chain_bench.err; cat gpurun_out/r05/chain_bench.json
