# Round-2 evidence, part 2 (tight timeouts: nothing here may hold the box): full GPU suite, the inference step through the
# reference's entry points (HDF5 -> HDF5) with and without the reader / writer lanes, encoder timings (device walk vs the
# host CIGAR pass), encoder kernel statistics + HBM counters.
set -x
R=$GRAFT_REPO_ROOT
TAG=${TAG:-r02d}
mkdir -p $R/gpurun_out
cd $R
timeout 200 python -m pytest tests -q -m gpu > gpurun_out/${TAG}_gpu_tests.log 2>&1; tail -8 gpurun_out/${TAG}_gpu_tests.log
timeout 90 python tools/bench_pipeline.py --files 8 --windows 262144 --workers 0,-1,4 > gpurun_out/${TAG}_pipeline.json 2> gpurun_out/${TAG}_pipeline.err; cat gpurun_out/${TAG}_pipeline.json; tail -2 gpurun_out/${TAG}_pipeline.err
timeout 90 python tools/bench_pipeline.py --files 8 --windows 262144 --groups 512 --workers 0,-1 > gpurun_out/${TAG}_pipeline_g512.json 2> gpurun_out/${TAG}_pipeline_g512.err; cat gpurun_out/${TAG}_pipeline_g512.json; tail -2 gpurun_out/${TAG}_pipeline_g512.err
timeout 150 python tools/bench_polish_pipeline.py --chunks 65536 --files 16 --workers 8,16,-1 > gpurun_out/${TAG}_polish_pipeline.json 2> gpurun_out/${TAG}_polish_pipeline.err; cat gpurun_out/${TAG}_polish_pipeline.json; tail -2 gpurun_out/${TAG}_polish_pipeline.err
timeout 60 python tools/bench_encoder.py --reps 5 > gpurun_out/${TAG}_encoder_device_walk.json 2>&1; tail -1 gpurun_out/${TAG}_encoder_device_walk.json
PA_ENCODER_HOST_CIGAR=1 timeout 60 python tools/bench_encoder.py --reps 5 > gpurun_out/${TAG}_encoder_host_cigar.json 2>&1; tail -1 gpurun_out/${TAG}_encoder_host_cigar.json
PA_ENCODER_TRACE=1 timeout 60 python tools/bench_encoder.py --reps 2 2>&1 | grep "encoder\]" | tail -12 > gpurun_out/${TAG}_encoder_trace.txt; cat gpurun_out/${TAG}_encoder_trace.txt
cd /tmp && export TMPDIR=/tmp
ENC="python $R/tools/bench_encoder.py --reps 5"
PENC="python $R/tools/bench_polish_encoder.py --reps 20"
timeout 60 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_enc_stats -o enc -- $ENC > $R/gpurun_out/${TAG}_enc_stats.log 2>&1
timeout 60 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/${TAG}_enc_fetch -o enc -- $ENC > $R/gpurun_out/${TAG}_enc_fetch.log 2>&1
timeout 60 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/${TAG}_enc_write -o enc -- $ENC > $R/gpurun_out/${TAG}_enc_write.log 2>&1
timeout 60 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_penc_stats -o penc -- $PENC > $R/gpurun_out/${TAG}_penc_stats.log 2>&1
timeout 60 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/${TAG}_penc_fetch -o penc -- $PENC > $R/gpurun_out/${TAG}_penc_fetch.log 2>&1
timeout 60 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/${TAG}_penc_write -o penc -- $PENC > $R/gpurun_out/${TAG}_penc_write.log 2>&1
cd $R
python tools/pmc_summary.py --model encoder --units 1 --out gpurun_out/${TAG}_encoder_variant --command "rocprofv3 --kernel-trace [--stats | --pmc FETCH_SIZE | --pmc WRITE_SIZE] -- $ENC" gpurun_out/${TAG}_enc_stats gpurun_out/${TAG}_enc_fetch gpurun_out/${TAG}_enc_write > /dev/null
python tools/pmc_summary.py --model encoder --units 1 --out gpurun_out/${TAG}_encoder_polish --command "rocprofv3 --kernel-trace [--stats | --pmc FETCH_SIZE | --pmc WRITE_SIZE] -- $PENC" gpurun_out/${TAG}_penc_stats gpurun_out/${TAG}_penc_fetch gpurun_out/${TAG}_penc_write > /dev/null
find gpurun_out -name "*.db" -size +2M -delete
head -14 gpurun_out/${TAG}_encoder_variant_kernel_stats.txt
