"""Turn rocprofv3 rocpd sqlite outputs (gpurun_out/<dir>/*_results.db) into the committed text
summaries under profiles/.   python profiles/summarize.py <db> [<db> ...] > profiles/<name>.txt"""
import collections
import sqlite3
import sys


def kernel_stats(cur):
    try:
        rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    except sqlite3.Error:
        return
    print("## kernel trace (rocprofv3 --kernel-trace --stats), durations in microseconds")
    print(f"{'calls':>7} {'total_us':>12} {'avg_us':>11} {'pct':>6}  kernel")
    for name, calls, total, avg, pct in rows:
        print(f"{calls:7d} {total:12.1f} {avg:11.2f} {pct:6.2f}  {name[:150]}")


def counters(cur):
    try:
        rows = list(cur.execute("select kernel_name, counter_name, count(*), avg(value), avg(duration) "
                                "from counters_collection group by kernel_name, counter_name"))
    except sqlite3.Error:
        return
    if not rows:
        return
    d = collections.defaultdict(dict)
    for k, c, n, v, dur in rows:
        d[k][c] = (n, v, dur)
    print("## PMC counters (rocprofv3 --pmc), per-dispatch averages")
    for k, v in d.items():
        print(k[:150])
        for c, (n, val, dur) in sorted(v.items()):
            print(f"    {c:34s} dispatches={n:4d} avg={val:.6g}  avg_dispatch_ns={dur:.6g}")
        if "GRBM_GUI_ACTIVE" in v and "SQ_VALU_MFMA_BUSY_CYCLES" in v:
            g, m, dur = v["GRBM_GUI_ACTIVE"][1], v["SQ_VALU_MFMA_BUSY_CYCLES"][1], v["GRBM_GUI_ACTIVE"][2]
            print(f"    -> shader clock ~ {g / dur / 8:.3f} GHz (GRBM_GUI_ACTIVE summed over 8 XCDs / duration)")
            print(f"    -> MfmaUtil = {m / (g / 8 * 1024) * 100:.1f} %  (MFMA busy cycles / (cycles x 256 CU x 4 SIMD))")
        if "FETCH_SIZE" in v:
            print(f"    -> FETCH_SIZE {v['FETCH_SIZE'][1] / 1e6:.3f} GB/dispatch as reported (KiB units); x2 for wide "
                  "coalesced streams per MI355X_MICROARCH.md HBM section")
        if "WRITE_SIZE" in v:
            print(f"    -> WRITE_SIZE {v['WRITE_SIZE'][1] / 1e6:.3f} GB/dispatch as reported (KiB units)")


for db in sys.argv[1:]:
    print(f"# {db}")
    con = sqlite3.connect(db)
    cur = con.cursor()
    kernel_stats(cur)
    counters(cur)
    print()
