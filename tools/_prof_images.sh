set -x
R=$GRAFT_REPO_ROOT
python $R/tools/bench_variant_images.py make_fast /tmp/vb 64000000 60 > $R/gpurun_out/pi_mk.log 2>&1
timeout 300 python $R/tools/bench_variant_images.py run /tmp/vb 16,16,16,20 > $R/gpurun_out/pi_run_plain.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/pi_stats -o img -- python $R/tools/bench_variant_images.py run /tmp/vb 16 > $R/gpurun_out/pi_run.log 2>&1
cd $R
python - <<'PY'
import glob, sqlite3
from collections import defaultdict
db = glob.glob("gpurun_out/pi_stats/**/*.db", recursive=True)[0]
con = sqlite3.connect(db)
kern = defaultdict(lambda: [0, 0.0])
lo, hi = None, None
for name, dur in con.execute("select name, duration from kernels"):
    kern[name][0] += 1
    kern[name][1] += dur / 1e6
with open("gpurun_out/pi_kernel_stats.txt", "w") as out:
    tot = sum(v[1] for v in kern.values())
    out.write("total kernel ms %.1f\n" % tot)
    for name, (calls, ms) in sorted(kern.items(), key=lambda kv: -kv[1][1])[:16]:
        out.write("%-70s calls %6d total_ms %9.2f avg_us %9.1f pct %5.1f\n" % (name[:70], calls, ms, 1e3 * ms / calls, 100 * ms / tot))
PY
find gpurun_out -name "*.db" -delete
