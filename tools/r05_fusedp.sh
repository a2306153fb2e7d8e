set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05 /dev/shm/e2e
timeout 900 python -m pytest tests/test_gpu_pipeline.py -x -q -k "fused_polish or polish_end_to_end" > gpurun_out/r05/fusedp_tests.log 2>&1
tail -25 gpurun_out/r05/fusedp_tests.log
for k in polish polish_fused; do
timeout 900 python tools/bench_e2e.py $k /dev/shm/e2e/po 32000000 60 2 > gpurun_out/r05/e2e_$k.json 2> gpurun_out/r05/e2e_$k.err
tail -1 gpurun_out/r05/e2e_$k.err; cat gpurun_out/r05/e2e_$k.json
rm -rf /dev/shm/e2e/po
done
