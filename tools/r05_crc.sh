set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_inflate.py tests/test_gpu_packed.py -x -q > gpurun_out/r05/crc_tests.log 2>&1
tail -12 gpurun_out/r05/crc_tests.log
timeout 300 python tools/bench_inflate.py --genome 8000000 > gpurun_out/r05/inflate_l1.json 2> gpurun_out/r05/inflate_l1.err; cat gpurun_out/r05/inflate_l1.json
timeout 300 python tools/bench_inflate.py --genome 4000000 --level 6 --tags 1 > gpurun_out/r05/inflate_l6.json 2> gpurun_out/r05/inflate_l6.err; cat gpurun_out/r05/inflate_l6.json
python tools/bench_variant_images.py make_fast /tmp/v6 8000000 60 2027 6 1 > gpurun_out/r05/v6_make.log 2>&1; cat gpurun_out/r05/v6_make.log
timeout 600 python tools/bench_variant_images.py run /tmp/v6 16,16,16 > gpurun_out/r05/v6_run.json 2> gpurun_out/r05/v6_run.err; cut -c1-1500 gpurun_out/r05/v6_run.json
