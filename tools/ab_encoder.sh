# A/B of tile_count_kernel builds on the GPU box: each flag set rebuilds the library (objects in parallel) and runs the
# batch tool.  usage: bash tools/ab_encoder.sh TAG "flags A" "flags B" ...
R=$GRAFT_REPO_ROOT; cd $R; TAG=$1; shift
mkdir -p gpurun_out/$TAG
i=0
for F in "$@"; do
  PEPPER_AMD_EXTRA_HIPCC_FLAGS="$F" python -c "from pepper_amd import build; build.build()" > gpurun_out/$TAG/build_$i.log 2>&1 || tail -5 gpurun_out/$TAG/build_$i.log
  echo "== flags: $F" | tee -a gpurun_out/$TAG/ab.txt
  timeout 200 python tools/bench_encoder.py --regions 64 --reps 8 --check 1 2> gpurun_out/$TAG/err_$i.log | tee -a gpurun_out/$TAG/ab.txt
  i=$((i+1))
done
