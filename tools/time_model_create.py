import ctypes, time, sys
sys.path.insert(0, "/root/repo")
import torch
from pepper_amd import _lib, synthetic
lib = _lib.load()
sd = synthetic.variant_state_dict(seed=0)
names, data, numel, k, keep = _lib.marshal_state_dict(sd)
for it in range(3):
    cfg = _lib.VariantConfig(26, 33, 1, 3, 0, 16384)
    h = ctypes.c_void_p()
    t0 = time.perf_counter()
    _lib.check(lib.pa_variant_create(ctypes.byref(cfg), names, data, numel, k, None, ctypes.byref(h)))
    print("pa_variant_create", round((time.perf_counter() - t0) * 1e3, 1), "ms")
    lib.pa_variant_destroy(h)
