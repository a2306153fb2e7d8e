"""How well polish device passes of one lane block (4 096 chunks = 64 workgroups) run side by side: K threads, each with
its own model handle (TransducerGRU.clone: own stream, own staging buffers), each running `passes` host-to-host passes.
    python tools/polish_concurrency.py [--block 4096] [--passes 6] [--threads 1,2,4,6,8]"""
import argparse
import json
import os
import sys
import threading
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pepper_amd import synthetic  # noqa: E402
from pepper_amd.polish.models.ModelHander import ModelHandler  # noqa: E402
from pepper_amd.polish.Options import ImageSizeOptions  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--block", type=int, default=4096)
    ap.add_argument("--passes", type=int, default=6)
    ap.add_argument("--threads", default="1,2,4,6,8")
    args = ap.parse_args()
    sd = synthetic.polish_state_dict(seed=0)
    first = ModelHandler.get_new_gru_model(ImageSizeOptions.IMAGE_CHANNELS, ImageSizeOptions.IMAGE_HEIGHT, 1, 128,
                                           ImageSizeOptions.TOTAL_LABELS)
    first.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    most = max(int(k) for k in args.threads.split(","))
    models = [first] + [first.clone() for _ in range(most - 1)]
    chunks = synthetic.polish_chunks_device(args.block, seed=1, device="cuda").cpu()
    bufs = [(chunks.clone().pin_memory().numpy(), torch.empty((args.block, 1000), dtype=torch.uint8).pin_memory().numpy(),
             torch.empty((args.block, 1000), dtype=torch.uint8).pin_memory().numpy()) for _ in range(most)]
    for m, b in zip(models, bufs):
        m.predict_chunks_into(*b)                      # warm-up: staging buffers, streams
    out = []
    for k in (int(v) for v in args.threads.split(",")):
        def work(i):
            torch.cuda.set_device(0)
            for _ in range(args.passes):
                models[i].predict_chunks_into(*bufs[i])
        threads = [threading.Thread(target=work, args=(i,)) for i in range(k)]
        t0 = time.perf_counter()
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        dt = time.perf_counter() - t0
        out.append({"threads": k, "ms_per_pass": round(1e3 * dt / args.passes, 2),
                    "chunks_per_s": round(k * args.passes * args.block / dt)})
    print(json.dumps({"metric": "polish device passes side by side, block of %d chunks each, host to host" % args.block, "runs": out}))


if __name__ == "__main__":
    main()
