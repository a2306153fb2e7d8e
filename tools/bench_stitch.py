"""Stitch rate (prediction HDF5 files -> polished FASTA; host only): perform_stitch with the merge inside the I/O library
(default) and with the numpy form (PEPPER_AMD_STITCH_NUMPY=1), one and eight workers.
    python tools/bench_stitch.py [--chunks 65536] [--files 4] [--dir /dev/shm]"""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pepper_amd.polish.DataStorePredict import DataStore  # noqa: E402
from pepper_amd.polish.perform_stitch import perform_stitch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chunks", type=int, default=65536)
    ap.add_argument("--files", type=int, default=4)
    ap.add_argument("--dir", default=None)
    args = ap.parse_args()
    tmp = tempfile.mkdtemp(dir=args.dir)
    try:
        pred = os.path.join(tmp, "pred")
        os.makedirs(pred)
        rng = np.random.default_rng(3)
        stores = [DataStore(os.path.join(pred, "pepper_prediction_%d.hdf" % k), "w") for k in range(args.files)]
        regions = args.chunks // 2
        block = 256                                        # regions per write call
        idx = np.zeros((2 * block, 1000), np.int64)
        for r0 in range(0, regions, block):
            m = min(block, regions - r0)
            start = (r0 + np.arange(m)) * 1000
            start2 = np.repeat(start, 2)
            chunk = np.tile(np.array([0, 1]), m)
            position = start2[:, None] + (chunk * 950)[:, None] + np.arange(1000)[None, :]
            position[position > (start2 + 1200)[:, None]] = -1
            bases = rng.integers(0, 5, (2 * m, 1000)).astype(np.uint8)
            contigs = np.array([b"ctg0"] * (2 * m), dtype="S256")
            stores[(r0 // block) % args.files].write_predictions_block(contigs, start2, start2 + 1200, chunk, position, idx[:2 * m],
                                                                        bases, bases)
        for s in stores:
            s.close()
        runs = []
        for numpy_form in (False, True):
            for threads in (1, 8):
                os.environ["PEPPER_AMD_STITCH_NUMPY"] = "1" if numpy_form else "0"
                t0 = time.perf_counter()
                out = perform_stitch(pred, os.path.join(tmp, "out%d%d" % (numpy_form, threads)), threads)
                dt = time.perf_counter() - t0
                runs.append({"merge": "numpy" if numpy_form else "library", "threads": threads, "seconds": round(dt, 2),
                             "chunks_per_s": round(2 * regions / dt), "bases": os.path.getsize(out)})
        from pepper_amd.hostinfo import usable_cpus
        print(json.dumps({"metric": "perform_stitch: prediction HDF5 -> FASTA (host)", "chunks": 2 * regions, "files": args.files,
                          "usable_cpus": usable_cpus(), "runs": runs}))
    finally:
        shutil.rmtree(tmp)


if __name__ == "__main__":
    main()
