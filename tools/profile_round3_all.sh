# Round-3 evidence, whole: GPU suite, the driver's bench command (with its secondary block), the N-rank plumbing records
# (ranks sharing the one GPU of the box over gloo: never a scaling number), WG-syn.
set -x
R=$GRAFT_REPO_ROOT
TAG=${TAG:-r03}
mkdir -p $R/gpurun_out/$TAG
cd $R
timeout 900 python -m pytest tests -q -m gpu > gpurun_out/$TAG/gpu_tests.log 2>&1; tail -5 gpurun_out/$TAG/gpu_tests.log
timeout 900 python bench.py > gpurun_out/$TAG/bench_variant.json 2> gpurun_out/$TAG/bench_variant.err; tail -3 gpurun_out/$TAG/bench_variant.err; python -c "
import json; d=json.load(open('gpurun_out/$TAG/bench_variant.json')); print(d['value'], d['roofline']['frac'], d['roofline'].get('frac_algorithmic_of_dtype_peak'), d.get('vs_baseline')); print(json.dumps(d.get('secondary'), indent=1)[:3000]); print(d.get('batch512'))"
PEPPER_AMD_BENCH_SHARE_GPU=1 timeout 400 python bench.py --gpus 8 --steps 3 --warmup 1 --per-gpu 32768 --pool 65536 --no-cpu-baseline --no-extras > gpurun_out/$TAG/bench_8rank_shared_gpu_plumbing.json 2> gpurun_out/$TAG/bench_8rank.err; grep "bench\] rank" gpurun_out/$TAG/bench_8rank.err; head -c 600 gpurun_out/$TAG/bench_8rank_shared_gpu_plumbing.json; echo
PEPPER_AMD_BENCH_SHARE_GPU=1 timeout 400 python bench.py --gpus 8 --workload wg-syn --per-gpu 1048576 --pool 262144 --no-cpu-baseline --no-extras > gpurun_out/$TAG/bench_wgsyn_8rank_shared_gpu_plumbing.json 2> gpurun_out/$TAG/bench_wgsyn_8rank.err; grep "wg-syn" gpurun_out/$TAG/bench_wgsyn_8rank.err | head -20; head -c 400 gpurun_out/$TAG/bench_wgsyn_8rank_shared_gpu_plumbing.json; echo
timeout 300 python bench.py --workload wg-syn --no-cpu-baseline > gpurun_out/$TAG/bench_wgsyn_1gpu.json 2> gpurun_out/$TAG/bench_wgsyn_1gpu.err; head -c 500 gpurun_out/$TAG/bench_wgsyn_1gpu.json; echo
timeout 100 python tools/bench_polish_encoder.py > gpurun_out/$TAG/bench_polish_encoder.json 2>/dev/null; cat gpurun_out/$TAG/bench_polish_encoder.json
timeout 100 python tools/small_batch_profile.py > gpurun_out/$TAG/small_batch_kernels.json 2>/dev/null
timeout 300 python bench.py --model encoder > gpurun_out/$TAG/bench_encoder_line.json 2> gpurun_out/$TAG/bench_encoder_line.err; head -c 700 gpurun_out/$TAG/bench_encoder_line.json; echo
timeout 300 python bench.py --model polish --no-secondary > gpurun_out/$TAG/bench_polish.json 2> gpurun_out/$TAG/bench_polish.err; head -c 400 gpurun_out/$TAG/bench_polish.json; echo
TAG=$TAG bash tools/profile_round3_encoder.sh > gpurun_out/$TAG/profile_encoder.log 2>&1; tail -30 gpurun_out/$TAG/profile_encoder.log | grep -v "^+"
