"""Debug aid (not product): where one workgroup's waves of the polish GRU step loops spend their cycles.
    PEPPER_AMD_EXTRA_HIPCC_FLAGS=-DPA_GRU_PHASE_TIMING python -m pepper_amd.build   (then rebuild without it)
    PA_DEBUG_TIMING=1 python tools/phase_timing_gru.py
Per kernel (decoder + fused head, fused encoder) and wave: cycles per step in the head block, the MFMA phase, the wait at
the barrier behind it, the gate phase, and the wait at the barrier behind that (s_memtime sums over the 100 steps of the
last window launch, workgroup 8)."""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from pepper_amd import _lib, synthetic  # noqa: E402


def main():
    lib = _lib.load()
    n = 16384
    cfg = _lib.PolishConfig(10, 128, 1, 5, 1000, 100, 50, 50, 0, n)
    names, data, numel, k, keep = _lib.marshal_state_dict(synthetic.polish_state_dict(seed=0))
    h = ctypes.c_void_p()
    _lib.check(lib.pa_polish_create(ctypes.byref(cfg), names, data, numel, k, None, ctypes.byref(h)))
    x = torch.from_numpy(np.resize(synthetic.polish_chunks(512, seed=1), (n, 1000, 10))).cuda()
    lab = torch.empty((n, 1000), dtype=torch.uint8, device="cuda")
    ph = torch.empty_like(lab)
    for _ in range(2):
        _lib.check(lib.pa_polish_predict_device(h, x.data_ptr(), n, lab.data_ptr(), ph.data_ptr(), None))
    _lib.check(lib.pa_synchronize(h))
    buf = np.zeros(128, np.uint64)
    lib.pa_debug_dump_gru_timing.restype = ctypes.c_int
    rc = lib.pa_debug_dump_gru_timing(buf.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0, rc
    b = buf.reshape(2, 8, 8).astype(np.float64) / 100.0
    for which, name in enumerate(("decoder + head (gru_rec_h2_kernel<128,256,XG,nt,DENSE>)", "fused encoder (gru_rec_h2_kernel<128,16>)")):
        print("==", name)
        print("   wave   head  mfma-phase  barrier  gate-phase  barrier   step")
        for w in range(8):
            r = b[which, w, :5]
            print("   %4d %6.0f %11.0f %8.0f %11.0f %8.0f %6.0f" % (w, r[0], r[1], r[2], r[3], r[4], r.sum()))
    lib.pa_polish_destroy(h)


if __name__ == "__main__":
    main()
