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
    ap.add_argument("--workers", type=int, default=0)
    args = ap.parse_args()
    n = args.chunks // (2 * args.files) * 2 * args.files
    tmp = tempfile.mkdtemp()
    try:
        img_dir = os.path.join(tmp, "images")
        os.makedirs(img_dir)
        chunks = synthetic.polish_chunks_device(4096, seed=1, device="cuda").cpu().numpy()
        labels = np.zeros((2, 1000), np.uint8)
        t0 = time.perf_counter()
        per_file = n // args.files
        for fi in range(args.files):
            with DataStore(os.path.join(img_dir, "pepper_hp_images_thread_%d.hdf" % fi), "w") as ds:
                for r in range(per_file // 2):
                    g = (fi * per_file // 2 + r)
                    region = ("ctg%d" % fi, r * 1000, r * 1000 + 1200)
                    pos = np.stack([np.stack([np.arange(1000) + region[1] + 950 * c, np.zeros(1000, np.int64)], axis=1) for c in range(2)])
                    k = (2 * g) % 4096
                    ds.write_summaries(region, chunks[k:k + 2], labels, pos, [0, 1])
        t_write = time.perf_counter() - t0
        sd = synthetic.polish_state_dict(seed=0)
        model_path = os.path.join(tmp, "polish.pkl")
        torch.save(synthetic.checkpoint_dict({k: torch.from_numpy(v) for k, v in sd.items()}, hidden_size=128), model_path)
        t0 = time.perf_counter()
        call_consensus(img_dir, model_path, 512, args.workers, os.path.join(tmp, "pred"), "0", True, 4)
        dt = time.perf_counter() - t0
        outs = sorted(os.listdir(os.path.join(tmp, "pred")))
        print(json.dumps({"metric": "call_consensus HDF5 -> HDF5, 1 GPU", "chunks": n, "windows": 19 * n, "image_files": args.files,
                          "num_workers": args.workers, "prediction_files": len(outs), "seconds": round(dt, 3),
                          "chunks_per_s": round(n / dt), "windows_per_s": round(19 * n / dt), "image_write_seconds": round(t_write, 2),
                          "host_cpus": os.cpu_count()}))
    finally:
        shutil.rmtree(tmp)


if __name__ == "__main__":       # the lanes spawn worker processes, which re-import this file
    main()
