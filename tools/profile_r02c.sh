# Round-2 evidence, part 2: full GPU suite, the inference step through the reference's entry points (HDF5 -> HDF5) with the
# reader / writer lanes, encoder kernel statistics + HBM counters.
set -x
R=$GRAFT_REPO_ROOT
TAG=${TAG:-r02c}
mkdir -p $R/gpurun_out
cd $R
timeout 900 python -m pytest tests -q -m gpu > gpurun_out/${TAG}_gpu_tests.log 2>&1; tail -8 gpurun_out/${TAG}_gpu_tests.log
timeout 400 python tools/bench_pipeline.py --files 8 --windows 262144 > gpurun_out/${TAG}_pipeline_lanes.json 2> gpurun_out/${TAG}_pipeline_lanes.err; tail -1 gpurun_out/${TAG}_pipeline_lanes.json; tail -2 gpurun_out/${TAG}_pipeline_lanes.err
timeout 400 python tools/bench_pipeline.py --files 8 --windows 262144 --groups 512 > gpurun_out/${TAG}_pipeline_lanes_g512.json 2> gpurun_out/${TAG}_pipeline_lanes_g512.err; tail -1 gpurun_out/${TAG}_pipeline_lanes_g512.json; tail -2 gpurun_out/${TAG}_pipeline_lanes_g512.err
timeout 400 python tools/bench_pipeline.py --files 8 --windows 262144 --workers 4 > gpurun_out/${TAG}_pipeline_lanes_w4.json 2> gpurun_out/${TAG}_pipeline_lanes_w4.err; tail -1 gpurun_out/${TAG}_pipeline_lanes_w4.json
timeout 600 python tools/bench_polish_pipeline.py --chunks 32768 --files 16 --workers 8 > gpurun_out/${TAG}_polish_pipeline_w8.json 2> gpurun_out/${TAG}_polish_pipeline_w8.err; tail -1 gpurun_out/${TAG}_polish_pipeline_w8.json; tail -2 gpurun_out/${TAG}_polish_pipeline_w8.err
timeout 600 python tools/bench_polish_pipeline.py --chunks 32768 --files 16 --workers 16 > gpurun_out/${TAG}_polish_pipeline_w16.json 2> gpurun_out/${TAG}_polish_pipeline_w16.err; tail -1 gpurun_out/${TAG}_polish_pipeline_w16.json
cd /tmp && export TMPDIR=/tmp
ENC="python $R/tools/bench_encoder.py --reps 5"
PENC="python $R/tools/bench_polish_encoder.py --reps 20"
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_enc_stats -o enc -- $ENC > $R/gpurun_out/${TAG}_enc_stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/${TAG}_enc_fetch -o enc -- $ENC > $R/gpurun_out/${TAG}_enc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/${TAG}_enc_write -o enc -- $ENC > $R/gpurun_out/${TAG}_enc_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_penc_stats -o penc -- $PENC > $R/gpurun_out/${TAG}_penc_stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/${TAG}_penc_fetch -o penc -- $PENC > $R/gpurun_out/${TAG}_penc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/${TAG}_penc_write -o penc -- $PENC > $R/gpurun_out/${TAG}_penc_write.log 2>&1
cd $R
grep "^{" gpurun_out/${TAG}_enc_stats.log | tail -1; grep "^{" gpurun_out/${TAG}_penc_stats.log | tail -1
python tools/pmc_summary.py --model encoder --units 1 --out gpurun_out/${TAG}_encoder_variant --command "rocprofv3 --kernel-trace [--stats | --pmc FETCH_SIZE | --pmc WRITE_SIZE] -- $ENC" gpurun_out/${TAG}_enc_stats gpurun_out/${TAG}_enc_fetch gpurun_out/${TAG}_enc_write > /dev/null
python tools/pmc_summary.py --model encoder --units 1 --out gpurun_out/${TAG}_encoder_polish --command "rocprofv3 --kernel-trace [--stats | --pmc FETCH_SIZE | --pmc WRITE_SIZE] -- $PENC" gpurun_out/${TAG}_penc_stats gpurun_out/${TAG}_penc_fetch gpurun_out/${TAG}_penc_write > /dev/null
find gpurun_out -name "*.db" -size +2M -delete
head -14 gpurun_out/${TAG}_encoder_variant_kernel_stats.txt
