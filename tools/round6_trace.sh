# Round 6: where run_inference's wall goes (lane trace), two job sizes.
O=gpurun_out/r06; mkdir -p $O
for w in 262144 524288; do
PEPPER_AMD_LANE_TRACE=1 timeout 300 python tools/bench_pipeline.py --files 16 --windows $w --groups 512 --workers 0 --dir /dev/shm > $O/trace_$w.out 2> $O/trace_$w.err
tail -1 $O/trace_$w.out | cut -c1-600; grep -v "INFO" $O/trace_$w.err | tail -40
done
