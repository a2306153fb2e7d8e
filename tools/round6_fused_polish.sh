# Round 6: the fused polish after its model passes moved to background threads: parity tests, then polish / polish_fused end to end
# on the 64 Mb / 60x job (tools/bench_e2e.py), with a sweep over the number of model handles and the gather size.
R=$(pwd); O=gpurun_out/r06; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_pipeline.py -m gpu -x -q -k "fused or two_ranks_on_one_device or polish_end_to_end" > $O/fused_tests.log 2>&1; tail -5 $O/fused_tests.log
S=/dev/shm/pe2e; mkdir -p $S
run() { # kind, env...
  kind=$1; shift
  env "$@" timeout 900 python tools/bench_e2e.py $kind $S/$kind 64000000 60 2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$kind', '$*', d['value'], d['runs_seconds'], d['stage_walls'], {k:d['image_stage_seconds_summed_over_workers'].get(k) for k in ('chain','fused_consensus','chain_score_kernel','chain_band_kernel')})"
  rm -rf $S/$kind
}
run polish A=1 | tee -a $O/fused_polish_sweep.txt
run polish_fused A=1 | tee -a $O/fused_polish_sweep.txt
run polish_fused PEPPER_AMD_FUSED_HANDLES=3 | tee -a $O/fused_polish_sweep.txt
run polish_fused PEPPER_AMD_FUSED_HANDLES=4 | tee -a $O/fused_polish_sweep.txt
run polish_fused PEPPER_AMD_FUSED_HANDLES=2 PEPPER_AMD_FUSED_PASS_CHUNKS=8192 | tee -a $O/fused_polish_sweep.txt
run polish_fused PEPPER_AMD_FUSED_HANDLES=4 PEPPER_AMD_FUSED_PASS_CHUNKS=2048 | tee -a $O/fused_polish_sweep.txt
rm -rf $S
