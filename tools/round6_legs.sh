# Round 6: selected end-to-end legs in the context the driver's run gives them (children of bench.py's process).
R=$(pwd); O=gpurun_out/r06; mkdir -p $O
show() { python -c "
import json,sys
d=json.load(open('gpurun_out/bench_full.json'))
for k,v in d['secondary'].items():
    print('$1', k, v.get('value'), v.get('runs_seconds'), v.get('stage_walls'))"; }
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --legs polish_e2e,polish_e2e_fused,call_variant_fused > /dev/null 2>$O/legs1.err; show default | tee -a $O/legs.txt
PEPPER_AMD_FUSED_HANDLES=4 timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --legs polish_e2e_fused,call_variant_fused > /dev/null 2>$O/legs2.err; show handles4 | tee -a $O/legs.txt
PEPPER_AMD_FUSED_HANDLES=3 timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --legs call_variant_fused > /dev/null 2>$O/legs3.err; show handles3 | tee -a $O/legs.txt
