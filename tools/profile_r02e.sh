set -x
cd $GRAFT_REPO_ROOT
export PEPPER_AMD_LANE_TRACE=1
timeout 100 python tools/bench_pipeline.py --files 8 --windows 262144 --workers 0,4,-1 > gpurun_out/r02e_pipeline.json 2> gpurun_out/r02e_pipeline.err; cat gpurun_out/r02e_pipeline.json; grep "lanes\]" gpurun_out/r02e_pipeline.err
timeout 150 python tools/bench_polish_pipeline.py --chunks 65536 --files 16 --workers 8,16,-1 > gpurun_out/r02e_polish_pipeline.json 2> gpurun_out/r02e_polish_pipeline.err; cat gpurun_out/r02e_polish_pipeline.json; grep "lanes\]" gpurun_out/r02e_polish_pipeline.err
