set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05 /dev/shm/e2e
timeout 900 python tools/bench_e2e.py polish /dev/shm/e2e/po 64000000 60 2 > gpurun_out/r05/e2e_polish_diag.json 2> gpurun_out/r05/e2e_polish_diag.err
cat gpurun_out/r05/e2e_polish_diag.json
grep -c "INFO" gpurun_out/r05/e2e_polish_diag.err
