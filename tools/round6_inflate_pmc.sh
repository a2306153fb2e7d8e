R=$(pwd); O=gpurun_out/r06; mkdir -p $O
python tools/bench_variant_images.py make_fast /tmp/ib 8000000 60 > $O/mk8.log 2>&1
cd /tmp && export TMPDIR=/tmp
for w in 0 1; do
B="python $R/tools/bench_inflate.py --bam /tmp/ib/reads.bam"
PA_INFLATE_WIDE=$w PA_INFLATE_WIDE_BELOW=48 timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES -d $R/$O/inflate_w${w}_pmc1 -o inf -- $B > $R/$O/inflate_w${w}_pmc1.log 2>&1
PA_INFLATE_WIDE=$w PA_INFLATE_WIDE_BELOW=48 timeout 200 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES -d $R/$O/inflate_w${w}_pmc2 -o inf -- $B > $R/$O/inflate_w${w}_pmc2.log 2>&1
done
cd $R
for w in 0 1; do python tools/rocprof_db_summary.py $O/inflate_w${w}_pmc1 $O/inflate_w${w}_pmc2 --only bgzf | grep -v "^#\|calls\|^$" ; done
find $O -name "*.db" -delete
