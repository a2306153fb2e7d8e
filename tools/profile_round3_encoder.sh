# Round-3 evidence for the summary encoder: GPU parity tests, batch tool, rocprofv3 kernel statistics and the two HBM
# counter passes (separate runs, --kernel-trace only, as MI355X_MICROARCH.md prescribes).  Tight timeouts throughout.
set -x
R=$GRAFT_REPO_ROOT
TAG=${TAG:-r03}
mkdir -p $R/gpurun_out/$TAG
cd $R
timeout 300 python -m pytest tests/test_gpu_encoder.py -x -q > gpurun_out/$TAG/enc_tests.log 2>&1; tail -6 gpurun_out/$TAG/enc_tests.log
timeout 200 python tools/bench_encoder.py --regions 64 --reps 8 > gpurun_out/$TAG/bench_encoder.json 2> gpurun_out/$TAG/bench_encoder.err; cat gpurun_out/$TAG/bench_encoder.json
cd /tmp && export TMPDIR=/tmp
python $R/tools/bench_encoder.py --regions 64 --reps 1 --check 0 --cache /tmp/esyn64.pkl > /dev/null 2>&1
ENC="python $R/tools/bench_encoder.py --regions 64 --reps 5 --check 0 --cache /tmp/esyn64.pkl"
timeout 120 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$TAG/enc_stats -o enc -- $ENC > $R/gpurun_out/$TAG/enc_stats.log 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/$TAG/enc_fetch -o enc -- $ENC > $R/gpurun_out/$TAG/enc_fetch.log 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/$TAG/enc_write -o enc -- $ENC > $R/gpurun_out/$TAG/enc_write.log 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA -d $R/gpurun_out/$TAG/enc_sq -o enc -- $ENC > $R/gpurun_out/$TAG/enc_sq.log 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS -d $R/gpurun_out/$TAG/enc_sq2 -o enc -- $ENC > $R/gpurun_out/$TAG/enc_sq2.log 2>&1
cd $R
python tools/pmc_summary.py --model encoder --units 64 --out gpurun_out/$TAG/encoder_variant --command "rocprofv3 --kernel-trace [--stats | --pmc FETCH_SIZE | --pmc WRITE_SIZE] -- $ENC" gpurun_out/$TAG/enc_stats gpurun_out/$TAG/enc_fetch gpurun_out/$TAG/enc_write gpurun_out/$TAG/enc_sq gpurun_out/$TAG/enc_sq2 > /dev/null
find gpurun_out/$TAG -name "*.db" -size +2M -delete
head -14 gpurun_out/$TAG/encoder_variant_kernel_stats.txt; grep -A40 'PMC counters' gpurun_out/$TAG/encoder_variant_kernel_stats.txt | grep tile_count
