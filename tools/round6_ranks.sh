# Round 6: the N-rank code path of bench.py on the one-GPU box (ranks share the device, gloo: plumbing only, never a scaling number).
O=gpurun_out/r06; mkdir -p $O
for n in 2 8; do
  PEPPER_AMD_BENCH_SHARE_GPU=1 timeout 900 python bench.py --gpus $n --steps 3 --warmup 1 --per-gpu 65536 --pool 131072 --no-cpu-baseline --no-extras > $O/bench_${n}rank_shared.json 2> $O/bench_${n}rank_shared.err
  echo "rc=$?"; tail -n 1 $O/bench_${n}rank_shared.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['n_gpus'], d['value'], d['config']['ranks_seen'], d['config']['collective_backend'], d['config'].get('per_rank_image_legs'))"
  grep "bench\] rank" $O/bench_${n}rank_shared.err | head -8
done
# the agreement path: ranks share the device AND try RCCL (refused: two ranks per device) -> every rank falls back together, bounded
PEPPER_AMD_BENCH_SHARE_GPU=2 PEPPER_AMD_RCCL_PROBE_SECONDS=20 timeout 600 python bench.py --gpus 2 --steps 3 --warmup 1 --per-gpu 65536 --pool 131072 --no-cpu-baseline --no-extras --no-image-legs > $O/bench_2rank_agree.json 2> $O/bench_2rank_agree.err
echo "rc=$?"; tail -n 1 $O/bench_2rank_agree.json | cut -c1-300; grep "RCCL\|bench\] rank" $O/bench_2rank_agree.err | head
