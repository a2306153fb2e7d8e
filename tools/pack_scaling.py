"""How pa_bam_pack_regions scales with threads, into pageable and into page-locked arenas (no encoder call).
python tools/pack_scaling.py <dir of tools/bench_variant_images.py make_fast>"""
import ctypes
import json
import os
import sys
import threading
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pepper_amd.variant.bam import BAM_handler, PACKED_READ  # noqa: E402


def main(data):
    info = json.load(open(os.path.join(data, "synth.json")))
    n_int = info["genome_bases"] // 100000
    out = []
    for kind in ("pageable", "pinned"):
        for threads in (1, 4, 16):
            arenas, keep = [], []
            for t in range(threads):
                if kind == "pinned":
                    from pepper_amd.variant.PEPPER_VARIANT import PackedEncoder
                    e = PackedEncoder(0, arena_bytes=256 << 20)
                    keep.append(e)
                    arenas.append(e.arena)
                else:
                    a = np.zeros(256 << 20, np.uint8)
                    a[::4096] = 1
                    arenas.append(a)
            per = n_int // threads
            done = [0.0] * threads

            def work(t):
                bam = BAM_handler(os.path.join(data, "reads.bam"))
                table = np.zeros(1 << 18, PACKED_READ)
                pairs = np.zeros(1 << 19, np.int32)
                t0 = time.perf_counter()
                i = t * per
                while i < (t + 1) * per:
                    k = min(8, (t + 1) * per - i)
                    starts = [max(0, (i + j) * 100000 - 100) for j in range(k)]
                    stops = [(i + j + 1) * 100000 + 100 for j in range(k)]
                    n_done, _, _ = bam.pack_regions("ctg1", starts, stops, False, 1, arenas[t], table, pairs)
                    i += n_done
                done[t] = time.perf_counter() - t0
            ths = [threading.Thread(target=work, args=(t,)) for t in range(threads)]
            t0 = time.perf_counter()
            for th in ths:
                th.start()
            for th in ths:
                th.join()
            dt = time.perf_counter() - t0
            out.append({"arena": kind, "threads": threads, "seconds": round(dt, 3), "mb_per_s": round(per * threads * 0.1 / dt, 2),
                        "slowest_thread_s": round(max(done), 3)})
            for e in keep:
                e.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main(sys.argv[1])
