"""End-to-end rate of polish inference as the reference runs it: image HDF5 files (one group per 1000-row chunk, one file
per image-generation thread) -> call_consensus -> predictions HDF5.
    python tools/bench_polish_pipeline.py [--chunks 65536] [--files 16] [--workers 8]
--workers = call_consensus's num_workers: reader / writer process lanes (pepper_amd/hostpipe.py); 0 = the in-process loop
for small jobs, lanes by default for big ones."""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pepper_amd import synthetic  # noqa: E402
from pepper_amd.polish.DataStore import DataStore  # noqa: E402
from pepper_amd.polish.call_consensus import call_consensus  # noqa: E402

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chunks", type=int, default=65536)
    ap.add_argument("--files", type=int, default=16)
    ap.add_argument("--workers", default="0", help="comma list of num_workers values (lanes); -1 = the in-process loop")
    ap.add_argument("--dir", default=None, help="parent of the scratch directory (default: the system's temporary directory; /dev/shm = tmpfs)")
    ap.add_argument("--median", action="store_true", help="after the per-run lines, one more line: the median run, with runs_chunks_per_s")
    args = ap.parse_args()
    n = args.chunks // (2 * args.files) * 2 * args.files
    tmp = tempfile.mkdtemp(dir=args.dir)
    try:
        img_dir = os.path.join(tmp, "images")
        os.makedirs(img_dir)
        chunks = synthetic.polish_chunks_device(4096, seed=1, device="cuda").cpu().numpy()
        labels = np.zeros((2, 1000), np.uint8)
        t0 = time.perf_counter()
        per_file = n // args.files
        first = os.path.join(img_dir, "pepper_hp_images_thread_0.hdf")
        with DataStore(first, "w") as ds:             # the other image files are byte copies of this one
            for r in range(per_file // 2):
                region = ("ctg0", r * 1000, r * 1000 + 1200)
                pos = np.stack([np.stack([np.arange(1000) + region[1] + 950 * c, np.zeros(1000, np.int64)], axis=1) for c in range(2)])
                k = (2 * r) % 4096
                ds.write_summaries(region, chunks[k:k + 2], labels, pos, [0, 1])
        for fi in range(1, args.files):
            shutil.copyfile(first, os.path.join(img_dir, "pepper_hp_images_thread_%d.hdf" % fi))
        t_write = time.perf_counter() - t0
        sd = synthetic.polish_state_dict(seed=0)
        model_path = os.path.join(tmp, "polish.pkl")
        torch.save(synthetic.checkpoint_dict({k: torch.from_numpy(v) for k, v in sd.items()}, hidden_size=128), model_path)
        records = []
        for k, w in enumerate(int(v) for v in str(args.workers).split(",")):
            os.environ["PEPPER_AMD_NO_LANES"] = "1" if w < 0 else "0"
            pred = os.path.join(tmp, "pred%d" % k)
            t0 = time.perf_counter()
            call_consensus(img_dir, model_path, 512, max(w, 0), pred, "0", True, 4)
            dt = time.perf_counter() - t0
            records.append({"metric": "call_consensus HDF5 -> HDF5, 1 GPU", "chunks": n, "windows": 19 * n, "image_files": args.files,
                            "num_workers": w, "mode": "in-process loop" if w < 0 else "lanes",
                            "prediction_files": len(os.listdir(pred)), "seconds": round(dt, 3), "chunks_per_s": round(n / dt),
                            "windows_per_s": round(19 * n / dt), "image_write_seconds": round(t_write, 2),
                            "host_cpus": os.cpu_count()})
            print(json.dumps(records[-1]), flush=True)
            shutil.rmtree(pred, ignore_errors=True)          # (the next run's files take its place in the scratch directory)
        if args.median and records:
            middle = sorted(records, key=lambda r: r["seconds"])[len(records) // 2]
            print(json.dumps(dict(middle, runs_chunks_per_s=[r["chunks_per_s"] for r in records])), flush=True)
    finally:
        shutil.rmtree(tmp)


if __name__ == "__main__":       # the lanes spawn worker processes, which re-import this file
    main()
