#!/usr/bin/env python
"""Per-kernel totals of a rocprofv3 result database (-d DIR holding one *_results.db): calls, total / average duration from the
kernel trace, and the per-dispatch averages of whatever counters a --pmc pass collected.

    python tools/rocprof_db_summary.py DIR [DIR ...] [--top 16] [--only substring]
"""
import argparse
import glob
import os
import sqlite3
from collections import defaultdict


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dirs", nargs="+")
    ap.add_argument("--top", type=int, default=16)
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    for d in args.dirs:
        dbs = sorted(glob.glob(os.path.join(d, "**", "*.db"), recursive=True))
        if not dbs:
            print("# %s: no result database" % d)
            continue
        con = sqlite3.connect(dbs[0])
        kern = defaultdict(lambda: [0, 0.0])
        for name, dur in con.execute("select name, duration from kernels"):
            kern[name][0] += 1
            kern[name][1] += dur / 1e6
        total = sum(v[1] for v in kern.values()) or 1.0
        print("# %s" % os.path.basename(os.path.normpath(d)))
        print("## kernel trace: total %.2f ms over %d dispatches" % (total, sum(v[0] for v in kern.values())))
        print("%8s %12s %12s %7s  kernel" % ("calls", "total_ms", "avg_us", "pct"))
        for name, (calls, ms) in sorted(kern.items(), key=lambda kv: -kv[1][1])[:args.top]:
            if args.only and args.only not in name:
                continue
            print("%8d %12.3f %12.1f %7.2f  %s" % (calls, ms, 1e3 * ms / calls, 100 * ms / total, name[:110]))
        try:
            counters = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
            for name, cname, value in con.execute("select kernel_name, counter_name, value from counters_collection"):
                c = counters[name][cname]
                c[0] += 1
                c[1] += value
            if counters:
                print("## counters (--pmc), average per dispatch")
                for name, cs in counters.items():
                    if args.only and args.only not in name:
                        continue
                    print("  %s" % name[:110])
                    for cname, (n, v) in sorted(cs.items()):
                        print("    %-28s %16.1f   (%d dispatches)" % (cname, v / n, n))
        except sqlite3.Error:
            pass
        con.close()


if __name__ == "__main__":
    main()
