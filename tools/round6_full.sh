# Round 6: the whole GPU suite, smoke(), and the bench line as the driver runs it.  Outputs under gpurun_out/r06/.
R=$(pwd); O=gpurun_out/r06; mkdir -p $O; T=${TAG:-full}
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/gpu_tests_$T.log 2>&1; tail -4 $O/gpu_tests_$T.log
( time timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) > $O/smoke_$T.log 2>&1; tail -4 $O/smoke_$T.log
( time timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_$T.json 2> $O/bench_$T.err; tail -c 300 $O/bench_$T.err; tail -n 1 $O/bench_$T.json | wc -c
cp gpurun_out/bench_full.json $O/bench_full_$T.json 2>/dev/null
