"""Per-kernel times of the variant forward at the reference's DataLoader batch (512 windows) and around it, device-resident.
python tools/small_batch_profile.py [sizes, comma-separated]"""
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pepper_amd import _lib, synthetic  # noqa: E402


def main():
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    sd = synthetic.variant_state_dict(seed=0)
    cfg = _lib.VariantConfig(26, 33, 1, 3, 0, 16384)
    names, data, numel, n, keep = _lib.marshal_state_dict(sd)
    h = ctypes.c_void_p()
    _lib.check(lib.pa_variant_create(ctypes.byref(cfg), names, data, numel, n, None, ctypes.byref(h)))
    sizes = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [128, 512, 1024, 2048, 4096]
    pool = synthetic.variant_windows_device(max(sizes), device=dev)
    out = {}
    for b in sizes:
        probs = torch.empty((b, 3), dtype=torch.float32, device=dev)
        for _ in range(5):
            _lib.check(lib.pa_variant_forward_device(h, pool.data_ptr(), b, probs.data_ptr(), None))
        _lib.check(lib.pa_synchronize(h))
        _lib.check(lib.pa_profile_enable(h, 1))
        reps = 50
        t0 = time.perf_counter()
        for _ in range(reps):
            _lib.check(lib.pa_variant_forward_device(h, pool.data_ptr(), b, probs.data_ptr(), None))
        _lib.check(lib.pa_synchronize(h))
        dt = (time.perf_counter() - t0) / reps
        prof = _lib.profile_dict(h)
        _lib.check(lib.pa_profile_enable(h, 0))
        out[b] = {"ms_per_call": dt * 1e3, "windows_per_s": b / dt,
                  "kernels_ms": {k: round(v["ms"] / max(1, v["launches"]), 4) for k, v in prof.items()}}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
