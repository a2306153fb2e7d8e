#!/usr/bin/env python
"""Device inflate of a synthetic BAM's BGZF members (csrc/inflate.hip): inflated GB/s of the kernel alone (HIP events, inputs
resident), beside zlib on one host core over a sample of the same members.

    python tools/bench_inflate.py [--genome 2000000] [--coverage 60] [--repeats 5] [--level 1] [--tags 0]

--level 6 --tags 1: the members as samtools / htslib write them (zlib level 6, NM / MD / RG aux data in every record).
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from pepper_amd.bgzf import DeviceInflater, block_table, inflate_host      # noqa: E402
from pepper_amd.hostinfo import usable_cpus      # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome", type=int, default=2000000)
    ap.add_argument("--coverage", type=int, default=60)
    ap.add_argument("--repeats", type=int, default=5)
    ap.add_argument("--bam", default=None)
    ap.add_argument("--level", type=int, default=1)
    ap.add_argument("--tags", type=int, default=0)
    ap.add_argument("--quals", type=int, default=0, help="1: quality strings with run-length structure (members compress > 3 x)")
    args = ap.parse_args()
    with tempfile.TemporaryDirectory() as tmp:
        bam = args.bam
        if bam is None:
            subprocess.run([os.path.join(ROOT, "tools", "synth_bam"), tmp, str(args.genome), str(args.coverage), "11", "0", "1", str(args.level), str(args.tags), str(args.quals)],
                           check=True, capture_output=True)
            bam = os.path.join(tmp, "reads.bam")
        raw = np.fromfile(bam, np.uint8)
    t0 = time.perf_counter()
    table = block_table(raw)
    t_table = time.perf_counter() - t0
    n = len(table[0])
    out_bytes = int(table[3].sum())
    with DeviceInflater() as inf:
        inf.inflate(raw, table)                       # first touch: allocations
        t0 = time.perf_counter()
        got = inf.inflate(raw, table, repeats=args.repeats)
        t_call = time.perf_counter() - t0
        ms = inf.last_kernel_ms
    sample = min(n, 400)
    t0 = time.perf_counter()
    ok = True
    for b in range(sample):
        o, l = int(table[0][b]), int(table[1][b])
        want = zlib.decompress(raw[o:o + l].tobytes(), -15)
        oo = int(table[2][b])
        ok = ok and got[oo:oo + len(want)].tobytes() == want
    t_cpu = time.perf_counter() - t0
    sample_bytes = int(table[3][:sample].sum())
    # the host's own inflate of ALL the members on every CPU the process may use (libdeflate where installed, as htslib):
    # the baseline the device form replaced in image generation, and a comparison of every byte
    cores = max(1, usable_cpus())
    inflate_host(raw, table, cores)
    t0 = time.perf_counter()
    host = inflate_host(raw, table, cores)
    t_host = time.perf_counter() - t0
    t0 = time.perf_counter()
    inflate_host(raw, [a[:sample] for a in table], 1)
    t_host1 = time.perf_counter() - t0
    all_identical = bool(host.size == got.size and np.array_equal(host, got))
    print(json.dumps({"deflate_level": args.level, "aux_tags": bool(args.tags), "members": n, "compressed_bytes": int(raw.size), "inflated_bytes": out_bytes,
                      "kernel_ms": round(ms, 3), "device_GBps_inflated": round(out_bytes / ms / 1e6, 2),
                      "device_GBps_compressed": round(raw.size / ms / 1e6, 2),
                      "host_call_s_with_transfers": round(t_call, 3), "repeats": args.repeats,
                      "zlib_one_core_GBps": round(sample_bytes / t_cpu / 1e9, 3), "zlib_sample_members": sample,
                      "sample_identical": bool(ok), "host_library_all_cores_GBps": round(out_bytes / t_host / 1e9, 2), "host_cores": cores,
                      "host_library_one_core_GBps": round(sample_bytes / t_host1 / 1e9, 3), "identical_to_host_library": all_identical,
                      "python_table_s": round(t_table, 3)}))


if __name__ == "__main__":
    main()
