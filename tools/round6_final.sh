# Round 6, second session: the record after the inflate diet, the proven-away 8-bit pass and the device-chosen strip size -- the whole
# GPU suite, smoke(), the bench line as the driver runs it, then the kernel traces / counter passes of the kernels this session
# changed (re-aligner, polish chain, inflate; each counter set in its own run).  GPU; outputs under gpurun_out/r06/ and gpurun_out/r06p/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
export GRAFT_REPO_ROOT=$R
cd $R && TAG=${TAG:-final} bash tools/round6_full.sh
mkdir -p $R/gpurun_out/r06p
python tools/bench_polish_chain.py make_fast /tmp/pc 4000000 > gpurun_out/r06p/chain_make.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r06p/realign_stats -o realign -- python $R/bench.py --model realign --steps 5 --warmup 2 --cpu-seconds 1 > $R/gpurun_out/r06p/realign_stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE -d $R/gpurun_out/r06p/realign_pmc -o realign -- python $R/bench.py --model realign --steps 5 --warmup 2 --cpu-seconds 1 > $R/gpurun_out/r06p/realign_pmc.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r06p/chain_stats -o chain -- python $R/tools/bench_polish_chain.py run /tmp/pc 1 > $R/gpurun_out/r06p/chain_stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE -d $R/gpurun_out/r06p/chain_pmc -o chain -- python $R/tools/bench_polish_chain.py run /tmp/pc 1 > $R/gpurun_out/r06p/chain_pmc.log 2>&1
cd $R
for d in realign_stats realign_pmc chain_stats chain_pmc; do python tools/rocprof_db_summary.py gpurun_out/r06p/$d > gpurun_out/r06p/$d.txt 2>&1; done
find gpurun_out -name "*.db" -delete
cd $R && PROFILE=1 SKIP_TESTS=1 TAG=_final bash tools/round6_inflate.sh
find gpurun_out -name "*.db" -delete
