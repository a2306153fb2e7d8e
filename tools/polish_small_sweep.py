"""Where the polish small-call schedule (gru_small_h2_kernel + projection GEMMs) stops paying: device-resident time of one
pa_polish_predict_device call of n chunks under both schedules.   python tools/polish_small_sweep.py [n,n,...]"""
import ctypes
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from pepper_amd import _lib, synthetic  # noqa: E402


def run(n_list, small_max):
    os.environ["PA_POLISH_SMALL_MAX"] = str(small_max)
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    sd = synthetic.polish_state_dict(seed=0)
    cfg = _lib.PolishConfig(10, 128, 1, 5, 1000, 100, 50, 50, 0, 16384)
    names, data, numel, k, keep = _lib.marshal_state_dict(sd)
    h = ctypes.c_void_p()
    _lib.check(lib.pa_polish_create(ctypes.byref(cfg), names, data, numel, k, ctypes.c_void_p(stream.cuda_stream), ctypes.byref(h)))
    out = {}
    for n in n_list:
        x = synthetic.polish_chunks_device(n, seed=3, device=dev)
        lab = torch.empty((n, 1000), dtype=torch.uint8, device=dev)
        ph = torch.empty((n, 1000), dtype=torch.uint8, device=dev)
        for _ in range(2):
            _lib.check(lib.pa_polish_predict_device(h, x.data_ptr(), n, lab.data_ptr(), ph.data_ptr(), None))
        _lib.check(lib.pa_synchronize(h))
        reps = 5
        t0 = time.perf_counter()
        for _ in range(reps):
            _lib.check(lib.pa_polish_predict_device(h, x.data_ptr(), n, lab.data_ptr(), ph.data_ptr(), None))
        _lib.check(lib.pa_synchronize(h))
        dt = (time.perf_counter() - t0) / reps
        out[n] = {"ms_per_call": round(dt * 1e3, 3), "windows_per_s": round(n * 19 / dt)}
    lib.pa_polish_destroy(h)
    return out


if __name__ == "__main__":
    ns = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [16, 128, 512, 1024, 2048, 4096, 8192]
    print(json.dumps({"small_call": run(ns, 1 << 30), "big_call": run(ns, 0)}))
