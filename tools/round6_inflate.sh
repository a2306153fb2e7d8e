# Round 6, the device inflate after the CRC moved into the kernel's epilogue and the long codes into the step: the inflate tests,
# the three bench workloads (level 1 / level 6 + tags / level 6 + tags + run-length qualities), and -- with PROFILE=1 -- the kernel
# trace and the counter passes (each in its own run) that bench.py's inflate roofline cites.  GPU; outputs under gpurun_out/r06/.
R=$(pwd); O=gpurun_out/r06; mkdir -p $O
[ "$SKIP_TESTS" = "1" ] || { timeout 900 python -m pytest tests/test_gpu_inflate.py tests/test_bam_reader.py -m gpu -x -q > $O/inflate_tests.log 2>&1; tail -4 $O/inflate_tests.log; }
for cfg in "--genome 8000000" "--genome 8000000 --level 6 --tags 1" "--genome 8000000 --level 6 --tags 1 --quals 1"; do
  timeout 300 python tools/bench_inflate.py $cfg 2>> $O/inflate_bench.err | tail -1 | tee -a $O/inflate_bench${TAG}.jsonl | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('device_GBps_inflated','kernel_ms','members','compressed_bytes','inflated_bytes','sample_identical','identical_to_host_library') if k in d})"
  PA_INFLATE_DEBUG=1 timeout 300 python tools/bench_inflate.py $cfg 2>&1 > /dev/null | grep "^inflate:" | tail -1
done
if [ "$PROFILE" = "1" ]; then
  python tools/bench_variant_images.py make_fast /tmp/ib 8000000 60 > $O/mk8.log 2>&1
  B="python $R/tools/bench_inflate.py --bam /tmp/ib/reads.bam"
  cd /tmp && export TMPDIR=/tmp
  timeout 200 rocprofv3 --kernel-trace --stats -d $R/$O/inflate_stats -o inf -- $B > $R/$O/inflate_stats.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES -d $R/$O/inflate_pmc1 -o inf -- $B > $R/$O/inflate_pmc1.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES -d $R/$O/inflate_pmc2 -o inf -- $B > $R/$O/inflate_pmc2.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/$O/inflate_fetch -o inf -- $B > $R/$O/inflate_fetch.log 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/$O/inflate_write -o inf -- $B > $R/$O/inflate_write.log 2>&1
  cd $R
  python tools/rocprof_db_summary.py $O/inflate_stats $O/inflate_pmc1 $O/inflate_pmc2 $O/inflate_fetch $O/inflate_write --only bgzf > $O/r06_inflate_kernel_stats.txt
  PA_INFLATE_DEBUG=1 $B 2>&1 > /dev/null | grep "^inflate:" | tail -1 >> $O/r06_inflate_kernel_stats.txt
  find $O -name "*.db" -delete
  tail -30 $O/r06_inflate_kernel_stats.txt
fi
