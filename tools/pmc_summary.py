#!/usr/bin/env python
"""Summarise rocprofv3 result databases (kernel trace + PMC passes) into the files kept under profiles/.

    python tools/pmc_summary.py --model variant --units 16384 --out profiles/r02_variant \\
        gpurun_out/r02_stats gpurun_out/r02_fetch gpurun_out/r02_write gpurun_out/r02_mfma

Each positional argument is a rocprofv3 output directory (-d) holding one *_results.db.  Writes
  <out>_kernel_stats.txt : per-kernel calls / total / average duration of every pass (the --stats view)
  <out>_pmc.json         : per bench.py kernel label: average duration, FETCH_SIZE as reported and doubled (the
                           correction MI355X_MICROARCH.md prescribes for 16-byte-per-lane streaming reads), WRITE_SIZE,
                           MFMA-busy and clock figures where those counters were collected.  bench.py reads this file for
                           roofline.traffic.
Counters are averaged per dispatch of a kernel; FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB.
"""
import argparse
import glob
import json
import os
import sqlite3
from collections import OrderedDict, defaultdict

# kernel symbol prefix -> label bench.py's profiler uses
LABELS = [
    ("lstm_rec_h2_kernel<256, 512, true, true", "lstm_dec_h2_fused"),
    ("lstm_rec_h2_kernel<256, 32, true, false", "lstm_rec_h2_fused_in"),
    ("gemm_h2_kernel", "gemm_h2_linear_1"),
    ("mlp_tail_h2_kernel", "mlp_tail_h2"),
    ("splitk_finish_kernel", "splitk_finish"),
    ("gru_rec_h2_kernel<128, 16", "gru_rec_h2_fused_in"),
    ("gru_rec_h2_kernel<128, 256, true, 2, true", "gru_dec_h2_fused_dense"),
    ("gru_rec_h2_kernel<128, 256, true, 0, true", "gru_dec_h2_fused_dense"),
    ("gru_rec_h2_kernel<128, 256", "gru_dec_h2_fused"),
    ("gru_dec_h2_kernel", "gru_dec_h2_fused"),
    ("polish_dense_acc_h2_kernel", "dense_softmax_acc"),
    ("polish_combine_kernel", "head_combine_acc"),
    ("polish_finalize_kernel", "polish_finalize"),
    ("pileup_count_kernel", "pileup_count"),
    ("apply_events_kernel", "apply_events"),
    ("init_matrix_kernel", "init_matrix"),
    ("site_threshold_kernel", "site_threshold"),
    ("gather_windows_kernel", "gather_windows"),
    ("polish_count_kernel", "polish_count"),
    ("polish_pixels_kernel", "polish_pixels"),
    ("cigar_walk_kernel", "cigar_walk"),
    ("tile_count_kernel", "tile_count"),
    ("segment_reads_kernel", "segment_reads"),
    ("tile_offsets_kernel", "tile_offsets"),
    ("bin_records_kernel", "bin_records"),
    ("compact_votes_kernel", "compact_votes"),
    ("gru_small_h2_kernel", "gru_small_h2"),
    ("unpack_clip_kernel", "unpack_clip"),
    ("pack_results_kernel", "pack_results"),
    ("polish_tile_kernel", "polish_tile"),
    ("polish_segment_kernel", "polish_segment"),
    ("polish_insert_rows_kernel", "polish_insert_rows"),
]


# FETCH_SIZE correction.  MI355X_MICROARCH.md (HBM section): on gfx950 the counter reports half the bytes of a wide
# (16 B per lane) coalesced streaming read with the default cache policy, "calibrate on a known byte count in your own
# access pattern before trusting an absolute".  Calibration done here (r02): gru_rec_h2_kernel<128,256,..,DENSE> must read
# its whole input once -- 16384 chunks x 100 rows x 1 KB = 1.678 GB per launch, its 0.6 MB of weights stay in L2 -- and
# FETCH_SIZE reports 1.689 GB for it: the nt-policy 16-byte loads of the step loops' x stream are counted in full.  So
# factor 1 for the kernels whose bulk reads carry nt, the guide's factor 2 for the rest.
# r03: tile_count_kernel (byte-per-lane nt loads) reports 874 MB for 860 MB of bytes it must read once: factor 1 as well
NT_STREAM_KERNELS = {"gru_dec_h2_fused_dense", "gru_dec_h2_fused", "lstm_dec_h2_fused", "tile_count", "segment_reads", "gather_windows",
                     "pack_results", "tile_offsets", "compact_votes"}


def fetch_factor(label):
    return 1.0 if label in NT_STREAM_KERNELS else 2.0


def short(name):
    n = name.replace("(anonymous namespace)::", "").replace("pa::", "")
    if n.startswith("void "):
        n = n[5:]
    return n.split("(")[0]


def label_of(name):
    s = short(name)
    for prefix, label in LABELS:
        if s.startswith(prefix):
            return label
    return s


def read_db(path):
    con = sqlite3.connect(path)
    kern = defaultdict(lambda: [0, 0.0])
    for name, dur in con.execute("select name, duration from kernels"):
        k = kern[name]
        k[0] += 1
        k[1] += dur / 1e3
    counters = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    try:
        for name, cname, value in con.execute("select kernel_name, counter_name, value from counters_collection"):
            c = counters[name][cname]
            c[0] += 1
            c[1] += value
    except sqlite3.Error:
        pass
    con.close()
    return kern, counters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dirs", nargs="+")
    ap.add_argument("--model", default="variant")
    ap.add_argument("--units", type=int, default=16384, help="windows / chunks one launch processes")
    ap.add_argument("--out", required=True)
    ap.add_argument("--command", default="", help="the profiled command, recorded in the outputs")
    args = ap.parse_args()
    text = []
    table = OrderedDict()
    for d in args.dirs:
        dbs = sorted(glob.glob(os.path.join(d, "**", "*.db"), recursive=True))
        if not dbs:
            text.append(f"# {d}: no result database\n")
            continue
        kern, counters = read_db(dbs[0])
        text.append(f"# {dbs[0]}")
        text.append("## kernel trace (rocprofv3 --kernel-trace), durations in microseconds")
        text.append("  calls     total_us      avg_us    pct  kernel")
        tot = sum(v[1] for v in kern.values()) or 1.0
        for name, (calls, us) in sorted(kern.items(), key=lambda kv: -kv[1][1]):
            text.append(f"{calls:7d} {us:12.1f} {us / calls:11.2f} {100 * us / tot:6.2f}  {short(name)[:150]}")
            e = table.setdefault(label_of(name), {"kernel": short(name), "units_per_launch": args.units})
            if not counters:       # duration from the un-instrumented pass only (PMC passes run slower)
                e["calls"] = calls
                e["avg_us"] = us / calls
        if counters:
            text.append("## PMC counters (rocprofv3 --pmc), per-dispatch averages")
            for name in sorted(counters):
                e = table.setdefault(label_of(name), {"kernel": short(name), "units_per_launch": args.units})
                for cname, (n, total) in sorted(counters[name].items()):
                    avg = total / n
                    text.append(f"  {short(name)[:100]:100s} {cname:28s} dispatches={n:4d} avg={avg:.6g}")
                    if cname == "FETCH_SIZE":
                        e["fetch_bytes_reported"] = avg * 1024.0
                        e["fetch_factor"] = fetch_factor(label_of(name))
                        e["fetch_bytes_corrected"] = e["fetch_factor"] * avg * 1024.0
                    elif cname == "WRITE_SIZE":
                        e["write_bytes"] = avg * 1024.0
                    else:
                        e[cname] = avg
        text.append("")
    for e in table.values():
        if "SQ_VALU_MFMA_BUSY_CYCLES" in e and "GRBM_GUI_ACTIVE" in e and e["GRBM_GUI_ACTIVE"]:
            # BUSY_CYCLES is summed over the 1024 SIMDs, GUI_ACTIVE over the 8 XCDs (r01 check: 49.8 M for a 4.13 ms
            # launch = 8 x 1.5 GHz)
            e["mfma_busy_frac"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (128.0 * e["GRBM_GUI_ACTIVE"])
            if "avg_us" in e:
                e["effective_clock_GHz_profiled"] = e["GRBM_GUI_ACTIVE"] / 8.0 / (e["avg_us"] * 1e3)
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    with open(args.out + "_kernel_stats.txt", "w") as fh:
        fh.write(f"# command: {args.command}\n" + "\n".join(text) + "\n")
    with open(args.out + "_pmc.json", "w") as fh:
        json.dump({"model": args.model, "command": args.command,
                   "note": "fetch_bytes_corrected = fetch_factor x FETCH_SIZE: 2 per MI355X_MICROARCH.md (HBM section) for default-"
                           "policy wide reads, 1 for the step loops whose x stream is loaded with the nt policy (calibrated on "
                           "gru_dec_h2_fused_dense, see tools/pmc_summary.py); WRITE_SIZE as reported; avg_us from the pass "
                           "without counters", "kernels": table}, fh, indent=1)
    print("\n".join(text))


if __name__ == "__main__":
    main()
