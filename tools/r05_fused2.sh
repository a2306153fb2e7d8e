set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05 /dev/shm/e2e
timeout 900 python tools/bench_e2e.py call_variant_fused /dev/shm/e2e/cv 128000000 30 2 > gpurun_out/r05/e2e_fused2.json 2> gpurun_out/r05/e2e_fused2.err
tail -2 gpurun_out/r05/e2e_fused2.err; cat gpurun_out/r05/e2e_fused2.json
