"""End-to-end rate of the inference step as the reference runs it: image HDF5 files on disk -> run_inference ->
predictions HDF5 (not the headline bench: includes libhdf5 reads, H2D, D2H and the per-batch prediction writes).
    python tools/bench_pipeline.py [--files 8] [--windows 65536]"""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time
from types import SimpleNamespace

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pepper_amd import synthetic  # noqa: E402
from pepper_amd.variant.DataStore import DataStore  # noqa: E402
from pepper_amd.variant.RunInference import run_inference  # noqa: E402

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--files", type=int, default=8)
    ap.add_argument("--windows", type=int, default=65536)
    ap.add_argument("--workers", default="0", help="options.num_workers: reader / writer process lanes (0 = automatic); a comma "
                                                     "list runs one measurement per value on the same image files; -1 = the "
                                                     "in-process loop (PEPPER_AMD_NO_LANES=1), -2 = lanes with one block in flight")
    ap.add_argument("--groups", type=int, default=4, help="summaries groups (regions) per image file")
    ap.add_argument("--dir", default=None, help="parent of the scratch directory (default: the system's temporary directory; /dev/shm = tmpfs)")
    args = ap.parse_args()
    tmp = tempfile.mkdtemp(dir=args.dir)
    try:
        img_dir = os.path.join(tmp, "images")
        os.makedirs(img_dir)
        t0 = time.perf_counter()
        # one image file written through the DataStore, the others are byte copies of it (the vlen candidate strings make
        # writing slow -- 5 s per 262144 windows -- and the content of the windows does not matter for this rate)
        first = os.path.join(img_dir, "pepper_variants_images_thread_0.hdf5")
        with DataStore(first, "w") as ds:
            per = args.windows // args.groups
            pool = synthetic.variant_windows_device(min(per, 65536), seed=1000, device="cuda").cpu().numpy()
            for gi in range(args.groups):
                x = np.resize(pool, (per, 33, 26)) if per > len(pool) else np.roll(pool, 7 * gi, axis=0)[:per]
                ds.write_summary("chr20_%d_%d" % (gi * 100000, (gi + 1) * 100000), ["chr20"] * per,
                                 np.arange(per) + gi * 100000, np.full(per, 30), np.array([["1A"]] * per, dtype=object),
                                 np.full((per, 1), 7), x, [0] * per, [0] * per, False)
        for fi in range(1, args.files):
            shutil.copyfile(first, os.path.join(img_dir, "pepper_variants_images_thread_%d.hdf5" % fi))
        t_write = time.perf_counter() - t0
        sd = synthetic.variant_state_dict(seed=0)
        model_path = os.path.join(tmp, "model.pkl")
        torch.save(synthetic.checkpoint_dict({k: torch.from_numpy(v) for k, v in sd.items()}, hidden_size=128), model_path)
        n = args.files * (args.windows // args.groups) * args.groups
        size = sum(os.path.getsize(os.path.join(img_dir, f)) for f in os.listdir(img_dir))
        for k, w in enumerate(int(v) for v in str(args.workers).split(",")):
            os.environ["PEPPER_AMD_NO_LANES"] = "1" if w == -1 else "0"
            os.environ["PEPPER_AMD_ONE_BLOCK_IN_FLIGHT"] = "1" if w == -2 else "0"      # -2: lanes, one block in flight
            opts = SimpleNamespace(model_path=model_path, batch_size=512, num_workers=max(w, 0), use_hp_info=False, gpu=True,
                                   device_ids="0", callers_per_gpu=4, threads=8, quantized=False, dry=False)
            pred = os.path.join(tmp, "pred%d" % k)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run_inference(opts, img_dir, pred)
            dt = time.perf_counter() - t0
            print(json.dumps({"metric": "run_inference HDF5 -> HDF5, 1 GPU", "windows": n, "image_bytes": size,
                              "num_workers": w, "mode": "in-process loop" if w == -1 else ("lanes, one block in flight" if w == -2 else "lanes"), "groups_per_file": args.groups,
                              "prediction_files": len(os.listdir(pred)), "host_cpus": os.cpu_count(), "seconds": round(dt, 3),
                              "windows_per_s": round(n / dt), "image_write_seconds": round(t_write, 2)}), flush=True)
    finally:
        shutil.rmtree(tmp)


if __name__ == "__main__":       # the lanes spawn worker processes, which re-import this file
    main()
