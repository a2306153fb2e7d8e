set -x
cd $GRAFT_REPO_ROOT
echo skip tests
export PEPPER_AMD_LANE_TRACE=1
timeout 100 python tools/bench_pipeline.py --files 8 --windows 262144 --workers=-1,0,4 > gpurun_out/r02e_pipeline.json 2> gpurun_out/r02e_pipeline.err; cat gpurun_out/r02e_pipeline.json; grep "lanes\]" gpurun_out/r02e_pipeline.err
timeout 100 python tools/bench_pipeline.py --files 8 --windows 262144 --groups 512 --workers=-1,0,4 > gpurun_out/r02e_pipeline_g512.json 2> gpurun_out/r02e_pipeline_g512.err; cat gpurun_out/r02e_pipeline_g512.json; grep "lanes\]" gpurun_out/r02e_pipeline_g512.err
timeout 150 python tools/bench_pipeline.py --files 8 --windows 1048576 --groups 2048 --workers=-1,0 > gpurun_out/r02e_pipeline_8M.json 2> gpurun_out/r02e_pipeline_8M.err; cat gpurun_out/r02e_pipeline_8M.json; grep "lanes\]" gpurun_out/r02e_pipeline_8M.err
timeout 150 python tools/bench_polish_pipeline.py --chunks 65536 --files 16 --workers=-1,0,8 > gpurun_out/r02e_polish_pipeline.json 2> gpurun_out/r02e_polish_pipeline.err; cat gpurun_out/r02e_polish_pipeline.json; grep "lanes\]" gpurun_out/r02e_polish_pipeline.err
