# Round 6 rocprofv3 evidence, every leg of the bench line that cites counters: kernel statistics + PMC passes (each counter set in its
# own run, --kernel-trace only, as MI355X_MICROARCH.md prescribes) of the variant and polish models (device-resident pass), the variant
# and polish summary encoders, the re-aligner and the polish chain.  (The inflate kernel: tools/round6_inflate.sh with PROFILE=1.)
# bash tools/profile_round6.sh   (GPU; outputs under gpurun_out/, the summaries are copied to profiles/r06_* by hand)
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
export GRAFT_REPO_ROOT=$R
cd $R && TAG=r06 bash tools/profile_round3_models.sh > gpurun_out/r06_models.log 2>&1
cd $R && TAG=r06 bash tools/profile_round3_encoder.sh > gpurun_out/r06_encoder.log 2>&1
cd /tmp && export TMPDIR=/tmp
PENC="python $R/bench.py --model polish-encoder --steps 6 --warmup 2 --no-cpu-baseline"
timeout 120 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r06_stats_penc -o penc -- $PENC > $R/gpurun_out/r06_stats_penc.log 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES -d $R/gpurun_out/r06_pmc_penc -o penc -- $PENC > $R/gpurun_out/r06_pmc_penc.log 2>&1
timeout 120 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU -d $R/gpurun_out/r06_pmc2_penc -o penc -- $PENC > $R/gpurun_out/r06_pmc2_penc.log 2>&1
cd $R
python tools/pmc_summary.py --model polish_encoder --units 256 --out gpurun_out/r06_encoder_polish --command "rocprofv3 --kernel-trace [--stats | --pmc SQ_INSTS_* | --pmc SQ_WAVE_CYCLES ...] -- python bench.py --model polish-encoder --steps 6 --warmup 2 --no-cpu-baseline" gpurun_out/r06_stats_penc gpurun_out/r06_pmc_penc gpurun_out/r06_pmc2_penc > /dev/null
# re-aligner and the polish chain
mkdir -p $R/gpurun_out/r06p
python tools/bench_polish_chain.py make_fast /tmp/pc 4000000 > gpurun_out/r06p/chain_make.log 2>&1
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r06p/realign_stats -o realign -- python $R/bench.py --model realign --steps 5 --warmup 2 --cpu-seconds 1 > $R/gpurun_out/r06p/realign_stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE -d $R/gpurun_out/r06p/realign_pmc -o realign -- python $R/bench.py --model realign --steps 5 --warmup 2 --cpu-seconds 1 > $R/gpurun_out/r06p/realign_pmc.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r06p/chain_stats -o chain -- python $R/tools/bench_polish_chain.py run /tmp/pc 1 > $R/gpurun_out/r06p/chain_stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE -d $R/gpurun_out/r06p/chain_pmc -o chain -- python $R/tools/bench_polish_chain.py run /tmp/pc 1 > $R/gpurun_out/r06p/chain_pmc.log 2>&1
cd $R
for d in realign_stats realign_pmc chain_stats chain_pmc; do python tools/rocprof_db_summary.py gpurun_out/r06p/$d > gpurun_out/r06p/$d.txt 2>&1; done
find gpurun_out -name "*.db" -delete
ls gpurun_out | grep r06
