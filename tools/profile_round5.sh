# Round 5 profiles: re-aligner (bench.py --model realign: 12 000 reads per call) and the polish image chain (make_images on a
# synthetic 4 Mb draft at 60x): kernel stats and SQ_INSTS_VALU per kernel, written under gpurun_out/r05/ for profiles/.
set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r05
cd $R
python tools/bench_polish_chain.py make_fast /tmp/pc 4000000 > gpurun_out/r05/chain_make.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r05/realign_stats -o realign -- python $R/bench.py --model realign --steps 5 --warmup 2 --cpu-seconds 1 > $R/gpurun_out/r05/realign_stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE -d $R/gpurun_out/r05/realign_pmc -o realign -- python $R/bench.py --model realign --steps 5 --warmup 2 --cpu-seconds 1 > $R/gpurun_out/r05/realign_pmc.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r05/chain_stats -o chain -- python $R/tools/bench_polish_chain.py run /tmp/pc 1 > $R/gpurun_out/r05/chain_stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE -d $R/gpurun_out/r05/chain_pmc -o chain -- python $R/tools/bench_polish_chain.py run /tmp/pc 1 > $R/gpurun_out/r05/chain_pmc.log 2>&1
cd $R
for d in realign_stats realign_pmc chain_stats chain_pmc; do python tools/rocprof_db_summary.py gpurun_out/r05/$d > gpurun_out/r05/$d.txt 2>&1; done
tail -3 gpurun_out/r05/realign_stats.log | cut -c1-1500
head -30 gpurun_out/r05/realign_pmc.txt; head -40 gpurun_out/r05/chain_stats.txt; grep -A4 "sw_ends\|band_kernel" gpurun_out/r05/chain_pmc.txt | head -40
