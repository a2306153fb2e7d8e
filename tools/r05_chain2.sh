set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_polish_chain.py -x -q > gpurun_out/r05/chain_tests.log 2>&1
tail -15 gpurun_out/r05/chain_tests.log
python tools/bench_polish_chain.py make_fast /tmp/pc 4000000 > gpurun_out/r05/chain_make.log 2>&1
PA_REALIGN_TRACE=1 timeout 300 python -c "
from pepper_amd.polish.make_images import make_images
make_images('/tmp/pc/reads.bam','/tmp/pc/draft.fa','ctg1:0-399999','/tmp/pc/img_trace',1)
" 2> gpurun_out/r05/chain_trace.err
grep realign-device gpurun_out/r05/chain_trace.err | head -5
timeout 600 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_realign.py tests/test_gpu_polish.py -x -q > gpurun_out/r05/other_tests.log 2>&1
tail -5 gpurun_out/r05/other_tests.log
