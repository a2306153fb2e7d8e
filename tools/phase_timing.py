"""Debug aid (not product): per-interval phase durations of the ping-pong LSTM kernel.
PA_DEBUG_TIMING=1 python tools/phase_timing.py"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pepper_amd import _lib, synthetic
lib = _lib.load()
n = 16384
cfg = _lib.VariantConfig(26, 33, 1, 3, 0, n)
names, data, numel, k, keep = _lib.marshal_state_dict(synthetic.variant_state_dict(seed=0))
h = ctypes.c_void_p()
_lib.check(lib.pa_variant_create(ctypes.byref(cfg), names, data, numel, k, None, ctypes.byref(h)))
x = torch.from_numpy(synthetic.variant_windows(n)).cuda()
out = torch.empty((n, 3), device="cuda")
for _ in range(3):
    _lib.check(lib.pa_variant_forward_device(h, x.data_ptr(), n, out.data_ptr(), None))
_lib.check(lib.pa_synchronize(h))
buf = np.zeros(2 * 8 * 80 * 2, np.uint64)
rc = lib.pa_debug_dump_timing(buf.ctypes.data_as(ctypes.c_void_p))
assert rc == 0, rc
buf = buf.reshape(2, 8, 80, 2).astype(np.int64)
for which, name in enumerate(("unfused (decoder)", "fused (encoder)")):
    b = buf[which]
    t0 = b[:, 0, 0].min()
    print("==", name)
    for wave in (0, 7):
        mf = (b[wave, 0:66:2, 1] - b[wave, 0:66:2, 0])[1:32]
        gt = (b[wave, 1:66:2, 1] - b[wave, 1:66:2, 0])[1:32]
        period = np.diff(b[wave, 0:66:2, 0])[1:31]
        bar1 = (b[wave, 1:66:2, 0] - b[wave, 0:66:2, 1])[1:32]
        print(f" wave {wave}: step period mean {period.mean():.0f} cycles; mfma phase mean {mf.mean():.0f} "
              f"(min {mf.min()} max {mf.max()}); gate phase mean {gt.mean():.0f} (min {gt.min()} max {gt.max()}); "
              f"barrier after mfma mean {bar1.mean():.0f}")
    print("  total cycles", (b[:, :66, 1].max() - t0))
