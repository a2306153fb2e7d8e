# Round-4 evidence for the device inflate: the kernel alone (bench, kernel trace, instruction counters in their own pass) and
# image generation with it (three runs, host inflate beside it, kernel trace of one job).  GPU; outputs under gpurun_out/.
set -x
R=$GRAFT_REPO_ROOT
cd $R
python tools/bench_inflate.py --genome 8000000 > gpurun_out/r04_inflate_bench.json 2> gpurun_out/r04_inflate_bench.err
python tools/bench_variant_images.py make_fast /tmp/vb 64000000 60 > gpurun_out/r04_mk.log 2>&1
timeout 300 python tools/bench_variant_images.py run /tmp/vb 16,16,16,16 2> /dev/null | grep "^{" > gpurun_out/r04_make_images_64mb.json
PEPPER_AMD_DEVICE_INFLATE=0 timeout 300 python tools/bench_variant_images.py run /tmp/vb 16,16,16 2> /dev/null | grep "^{" > gpurun_out/r04_make_images_64mb_host_inflate.json
PEPPER_AMD_DEVICE_WALK=0 timeout 300 python tools/bench_variant_images.py run /tmp/vb 16,16,16 2> /dev/null | grep "^{" > gpurun_out/r04_make_images_64mb_host_walk.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r04_images_stats -o img -- python $R/tools/bench_variant_images.py run /tmp/vb 16 > $R/gpurun_out/r04_images_stats.log 2>&1
cd $R
rm -rf /tmp/vb                                  # (room for the second data set)
python tools/bench_variant_images.py make_fast /tmp/ib 8000000 60 > gpurun_out/r04_mk8.log 2>&1
B="python $R/tools/bench_inflate.py --bam /tmp/ib/reads.bam"
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r04_inflate_stats -o inf -- $B > $R/gpurun_out/r04_inflate_stats.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES -d $R/gpurun_out/r04_inflate_pmc1 -o inf -- $B > $R/gpurun_out/r04_inflate_pmc1.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES -d $R/gpurun_out/r04_inflate_pmc2 -o inf -- $B > $R/gpurun_out/r04_inflate_pmc2.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/r04_inflate_fetch -o inf -- $B > $R/gpurun_out/r04_inflate_fetch.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/r04_inflate_write -o inf -- $B > $R/gpurun_out/r04_inflate_write.log 2>&1
cd $R
python tools/rocprof_db_summary.py gpurun_out/r04_images_stats > gpurun_out/r04_make_images_kernel_stats.txt
python tools/rocprof_db_summary.py gpurun_out/r04_inflate_stats gpurun_out/r04_inflate_pmc1 gpurun_out/r04_inflate_pmc2 gpurun_out/r04_inflate_fetch gpurun_out/r04_inflate_write --only bgzf > gpurun_out/r04_inflate_kernel_stats.txt
PA_INFLATE_DEBUG=1 $B 2>&1 > /dev/null | tail -1 >> gpurun_out/r04_inflate_kernel_stats.txt
find gpurun_out -name "*.db" -delete
