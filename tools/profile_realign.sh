# Secondary workloads of round 1: re-aligner bench + rocprofv3 kernel stats, NS-literal polish-stack bench.
set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
timeout 300 python bench.py --model realign --steps 10 --warmup 3 --cpu-seconds 10 > gpurun_out/r01_bench_realign.json 2> gpurun_out/r01_bench_realign.err
timeout 300 python bench.py --model ns-literal --steps 3 --warmup 1 > gpurun_out/r01_bench_ns_literal.json 2> gpurun_out/r01_bench_ns_literal.err
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r01_stats_realign -o realign -- python $R/tools/realign_stages.py 1500 > $R/gpurun_out/r01_stats_realign.log 2>&1
cd $R; ls gpurun_out/r01_stats_realign/*; tail -c 400 gpurun_out/r01_bench_realign.json
