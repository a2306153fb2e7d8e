set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_pipeline.py -x -q -k "fused or call_variant" > gpurun_out/r05/fused_tests.log 2>&1
tail -25 gpurun_out/r05/fused_tests.log
mkdir -p /dev/shm/e2e
for k in call_variant call_variant_fused; do
timeout 900 python tools/bench_e2e.py $k /dev/shm/e2e/cv 128000000 30 2 > gpurun_out/r05/e2e_$k.json 2> gpurun_out/r05/e2e_$k.err
tail -2 gpurun_out/r05/e2e_$k.err; cat gpurun_out/r05/e2e_$k.json
rm -rf /dev/shm/e2e/cv
done
