# Round 6, second session: the 2-rank line as the DRIVER launches it (torch.distributed.run, one process per rank), ranks sharing the one GPU
# (plumbing only: gloo, never a scaling number) -- after the re-aligner changes the per-rank polish image leg goes through.
O=gpurun_out/r06; mkdir -p $O
PEPPER_AMD_BENCH_SHARE_GPU=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 --per-gpu 65536 --pool 131072 --no-cpu-baseline --no-extras > $O/bench_2rank_driver_form.json 2> $O/bench_2rank_driver_form.err
echo "rc=$?"; tail -n 1 $O/bench_2rank_driver_form.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['n_gpus'], d['value'], d['config']['ranks_seen'], d['config']['collective_backend'], d['config'].get('per_rank_image_legs'))"
grep "bench\] rank" $O/bench_2rank_driver_form.err | head -8
