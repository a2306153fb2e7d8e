mkdir -p gpurun_out/r06
python tools/bench_polish_chain.py make_fast /tmp/pc 4000000 > /dev/null 2>&1
PA_REALIGN_TRACE=1 python tools/bench_polish_chain.py run /tmp/pc 1 2>&1 | grep "realign-device" | tail -4 > gpurun_out/r06/realign_trace.txt
cat gpurun_out/r06/realign_trace.txt
