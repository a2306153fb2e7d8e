# Round 6: the fused forms after (i) the model passes moved to background threads, (ii) the gather copies left the process-wide default
# stream, (iii) the model streams became high-priority ones: parity tests, then polish / polish_fused (64 Mb / 60x) and call_variant /
# call_variant_fused (256 Mb / 30x) end to end (tools/bench_e2e.py) with a sweep over priority, handles and gather size.
R=$(pwd); O=gpurun_out/r06; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_pipeline.py -m gpu -x -q -k "fused_polish" > $O/fused_tests.log 2>&1; tail -3 $O/fused_tests.log
S=/dev/shm/pe2e; mkdir -p $S
run() { # kind, bases, coverage, env...
  kind=$1; bases=$2; cov=$3; shift 3
  env "$@" timeout 150 python tools/bench_e2e.py $kind $S/$kind $bases $cov 2 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$kind', '$*', d['value'], d['runs_seconds'], d['stage_walls'], {k:d['image_stage_seconds_summed_over_workers'].get(k) for k in ('chain','fused_consensus','chain_score_kernel','chain_band_kernel','fused_forward','hdf5','encode')})"
  rm -rf $S/$kind
}
run polish_fused 64000000 60 A=1 | tee -a $O/fused_sweep5.txt
run polish 64000000 60 A=1 | tee -a $O/fused_sweep5.txt
run polish_fused 64000000 60 PEPPER_AMD_FUSED_HANDLES=1 | tee -a $O/fused_sweep5.txt
rm -rf $S
