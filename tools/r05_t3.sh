set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_images_vs_ref.py -x -q > gpurun_out/r05/vsref_tests.log 2>&1
tail -25 gpurun_out/r05/vsref_tests.log
mkdir -p /dev/shm/e2e
timeout 900 python tools/bench_e2e.py call_variant /dev/shm/e2e/cv 128000000 30 2 > gpurun_out/r05/e2e_cv.json 2> gpurun_out/r05/e2e_cv.err
tail -3 gpurun_out/r05/e2e_cv.err; cat gpurun_out/r05/e2e_cv.json
rm -rf /dev/shm/e2e/cv
timeout 900 python tools/bench_e2e.py polish /dev/shm/e2e/po 16000000 60 2 > gpurun_out/r05/e2e_po.json 2> gpurun_out/r05/e2e_po.err
tail -3 gpurun_out/r05/e2e_po.err; cat gpurun_out/r05/e2e_po.json
