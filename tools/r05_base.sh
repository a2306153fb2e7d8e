set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
python tools/bench_polish_images.py make_fast /tmp/pdata 1200000 > gpurun_out/r05/base_make.log 2>&1
timeout 600 python tools/bench_polish_images.py run /tmp/pdata > gpurun_out/r05/base_polish_images.json 2> gpurun_out/r05/base_polish_images.err
PA_REALIGN_TRACE=1 timeout 300 python -c "
from pepper_amd.polish.make_images import make_images
make_images('/tmp/pdata/reads.bam','/tmp/pdata/draft.fa','ctg1:0-99999','/tmp/pdata/img_trace',1)
" 2> gpurun_out/r05/base_trace.err
timeout 120 python tools/realign_stages.py 1500 > gpurun_out/r05/base_stages.log 2>&1
timeout 200 python bench.py --model realign --steps 10 --warmup 2 --cpu-seconds 2 > gpurun_out/r05/base_realign.json 2>gpurun_out/r05/base_realign.err
tail -3 gpurun_out/r05/base_polish_images.json; cat gpurun_out/r05/base_stages.log; grep -c "band kernel" gpurun_out/r05/base_trace.err; tail -c 600 gpurun_out/r05/base_realign.json
