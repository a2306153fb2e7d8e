set -x
R=$GRAFT_REPO_ROOT
cd $R
df -h /tmp | tail -1
python tools/bench_variant_images.py make_fast /tmp/ib 8000000 60 > gpurun_out/r04_mk8.log 2>&1
B="python $R/tools/bench_inflate.py --bam /tmp/ib/reads.bam"
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r04_inflate_stats -o inf -- $B > $R/gpurun_out/r04_inflate_stats.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES -d $R/gpurun_out/r04_inflate_pmc1 -o inf -- $B > $R/gpurun_out/r04_inflate_pmc1.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES -d $R/gpurun_out/r04_inflate_pmc2 -o inf -- $B > $R/gpurun_out/r04_inflate_pmc2.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE WRITE_SIZE -d $R/gpurun_out/r04_inflate_pmc3 -o inf -- $B > $R/gpurun_out/r04_inflate_pmc3.log 2>&1
cd $R
python tools/rocprof_db_summary.py gpurun_out/r04_inflate_stats gpurun_out/r04_inflate_pmc1 gpurun_out/r04_inflate_pmc2 gpurun_out/r04_inflate_pmc3 --only bgzf > gpurun_out/r04_inflate_kernel_stats.txt
PA_INFLATE_DEBUG=1 $B 2> gpurun_out/r04_inflate_symbols.txt > /dev/null
find gpurun_out -name "*.db" -delete
