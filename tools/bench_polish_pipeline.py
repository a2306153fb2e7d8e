"""End-to-end rate of polish inference as the reference runs it: image HDF5 (one group per 1000-row chunk) ->
call_consensus -> predictions HDF5.   python tools/bench_polish_pipeline.py [n_chunks]"""
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

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
tmp = tempfile.mkdtemp()
try:
    img_dir = os.path.join(tmp, "images")
    os.makedirs(img_dir)
    chunks = synthetic.polish_chunks(n, seed=1)
    labels = np.zeros((2, 1000), np.uint8)
    t0 = time.perf_counter()
    with DataStore(os.path.join(img_dir, "pepper_hp_images_thread_0.hdf"), "w") as ds:
        for r in range(n // 2):
            region = ("ctg1", r * 1000, r * 1000 + 1200)
            pos = np.stack([np.stack([np.arange(1000) + region[1] + 950 * c, np.zeros(1000, np.int64)], axis=1) for c in range(2)])
            ds.write_summaries(region, chunks[2 * r:2 * r + 2], labels, pos, [0, 1])
    t_write = time.perf_counter() - t0
    sd = synthetic.polish_state_dict(seed=0)
    model_path = os.path.join(tmp, "polish.pkl")
    torch.save(synthetic.checkpoint_dict({k: torch.from_numpy(v) for k, v in sd.items()}, hidden_size=128), model_path)
    t0 = time.perf_counter()
    call_consensus(img_dir, model_path, 512, 0, os.path.join(tmp, "pred"), "0", True, 4)
    dt = time.perf_counter() - t0
    print(json.dumps({"metric": "call_consensus HDF5 -> HDF5, 1 GPU", "chunks": n, "windows": 19 * n, "seconds": round(dt, 3),
                      "chunks_per_s": round(n / dt), "windows_per_s": round(19 * n / dt), "image_write_seconds": round(t_write, 2)}))
finally:
    shutil.rmtree(tmp)
