# Round-4 rocprofv3 evidence: kernel statistics + PMC passes (each counter set in its own run, --kernel-trace only, as
# MI355X_MICROARCH.md prescribes) of the variant and polish models (device-resident pass), the polish small-call schedule,
# both summary encoders.  TAG=r04 bash tools/profile_round4.sh   (GPU; outputs under gpurun_out/, summaries copied to profiles/)
set -x
R=$GRAFT_REPO_ROOT
TAG=${TAG:-r04}
cd $R && TAG=$TAG bash tools/profile_round3_models.sh > gpurun_out/${TAG}_models.log 2>&1
cd $R && TAG=$TAG bash tools/profile_round3_encoder.sh > gpurun_out/${TAG}_encoder.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 120 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_stats_psmall -o psmall -- python $R/tools/polish_small_sweep.py 128 > $R/gpurun_out/${TAG}_stats_psmall.log 2>&1
timeout 120 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_stats_penc -o penc -- python $R/bench.py --model polish-encoder --steps 6 --warmup 2 --no-cpu-baseline > $R/gpurun_out/${TAG}_stats_penc.log 2>&1
cd $R
python tools/pmc_summary.py --model polish_small --units 128 --out gpurun_out/${TAG}_polish_small_call --command "rocprofv3 --kernel-trace --stats -- python tools/polish_small_sweep.py 128" gpurun_out/${TAG}_stats_psmall > /dev/null
python tools/pmc_summary.py --model polish_encoder --units 256 --out gpurun_out/${TAG}_encoder_polish --command "rocprofv3 --kernel-trace --stats -- python bench.py --model polish-encoder --steps 6 --warmup 2 --no-cpu-baseline" gpurun_out/${TAG}_stats_penc > /dev/null
find gpurun_out -name "*.db" -delete
ls gpurun_out | grep ${TAG}
