# Where tile_count_kernel's time goes: the kernel with parts of it switched off (PA_TILE_DEBUG bits: 1 no records at all, 2 no
# matrix store, 4 no per-position pass, 8 no insert / deletion events, 32 no exception queue (no drain), 64 no run-level
# difference arrays, 128 no votes leaving the CU) on the 64-region E-syn batch.  Results differ, times do not lie.
#   bash tools/tile_ablation.sh > profiles/rNN_tile_ablation.txt        (GPU)
python tools/bench_encoder.py --regions 64 --reps 1 --check 0 --cache /tmp/esyn64.pkl > /dev/null 2>&1
for d in 0 1 3 2 4 128 8 32 40 64 104 232; do
PA_TILE_DEBUG=$d python tools/bench_encoder.py --regions 64 --reps 6 --check 0 --cache /tmp/esyn64.pkl 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$d', round(d['timing_ms']['tile_count_ms'],3), round(d['timing_ms']['records_ms'],3))"
done
