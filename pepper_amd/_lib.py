"""ctypes binding of include/pepper_amd.h (the C-ABI drop-in boundary).

The product path has NO CPU fallback: if the HIP extension is missing or no gfx950 device is
visible, calls raise -- they never route through oracle/ or torch CPU ops.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libpepper_amd.so")

PA_OK = 0

c_void_p, c_int32, c_int64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64
c_char_p, c_double = ctypes.c_char_p, ctypes.c_double


class VariantConfig(ctypes.Structure):
    _fields_ = [("image_features", c_int32), ("window", c_int32), ("gru_layers", c_int32),
                ("num_classes_type", c_int32), ("device", c_int32), ("max_chunk", c_int32)]


class PolishConfig(ctypes.Structure):
    _fields_ = [("image_features", c_int32), ("hidden_size", c_int32), ("gru_layers", c_int32),
                ("num_classes", c_int32), ("seq_length", c_int32), ("window", c_int32),
                ("jump", c_int32), ("overlap", c_int32), ("device", c_int32), ("max_chunk", c_int32)]


# (name, restype, argtypes) for every symbol include/pepper_amd.h declares
SYMBOLS = [
    ("pa_last_error", c_char_p, []),
    ("pa_version", c_char_p, []),
    ("pa_device_count", ctypes.c_int, []),
    ("pa_variant_create", ctypes.c_int, [ctypes.POINTER(VariantConfig), ctypes.POINTER(c_char_p),
                                         ctypes.POINTER(c_void_p), ctypes.POINTER(c_int64), c_int32,
                                         c_void_p, ctypes.POINTER(c_void_p)]),
    ("pa_variant_destroy", None, [c_void_p]),
    ("pa_variant_overflow_rows", ctypes.c_int, [c_void_p, ctypes.POINTER(c_int64)]),
    ("pa_variant_split_fallbacks", ctypes.c_int, [c_void_p, ctypes.POINTER(c_int64)]),
    ("pa_variant_forward_device", ctypes.c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    ("pa_variant_forward_device_f32", ctypes.c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    ("pa_variant_forward_host", ctypes.c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    ("pa_polish_create", ctypes.c_int, [ctypes.POINTER(PolishConfig), ctypes.POINTER(c_char_p),
                                        ctypes.POINTER(c_void_p), ctypes.POINTER(c_int64), c_int32,
                                        c_void_p, ctypes.POINTER(c_void_p)]),
    ("pa_polish_destroy", None, [c_void_p]),
    ("pa_polish_forward_device", ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int32,
                                                c_void_p, c_void_p]),
    ("pa_polish_predict_device", ctypes.c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    ("pa_polish_predict_host", ctypes.c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    ("pa_polish_predict_host_parts", ctypes.c_int, [c_void_p, ctypes.c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    ("pa_profile_enable", ctypes.c_int, [c_void_p, c_int32]),
    ("pa_profile_count", ctypes.c_int, [c_void_p]),
    ("pa_profile_get", ctypes.c_int, [c_void_p, c_int32, c_char_p, c_int32, ctypes.POINTER(c_double),
                                      ctypes.POINTER(c_int64), ctypes.POINTER(c_double)]),
    ("pa_synchronize", ctypes.c_int, [c_void_p]),
    ("pa_host_register", ctypes.c_int, [c_void_p, c_int64]),
    ("pa_host_unregister", ctypes.c_int, [c_void_p]),
    # include/pepper_amd_encoder.h (struct pointers passed as void*; typed structs live in
    # pepper_amd/variant/PEPPER_VARIANT.py)
    ("pa_encoder_create", ctypes.c_int, [c_int32, c_void_p, ctypes.POINTER(c_void_p)]),
    ("pa_encoder_destroy", None, [c_void_p]),
    ("pa_encoder_generate_summary", ctypes.c_int, [c_void_p, c_void_p, c_void_p, ctypes.POINTER(c_int64)]),
    ("pa_encoder_generate_summary_batch", ctypes.c_int, [c_void_p, c_int32, c_void_p, c_void_p, c_void_p]),
    ("pa_encoder_stage_batch", ctypes.c_int, [c_void_p, c_int32, c_void_p, c_void_p]),
    ("pa_encoder_run_staged", ctypes.c_int, [c_void_p, c_void_p]),
    ("pa_encoder_host_arena", c_void_p, [c_void_p, c_int64]),
    ("pa_encoder_stage_packed", ctypes.c_int, [c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int32,
                                               c_void_p, c_void_p]),
    ("pa_encoder_host_span", c_void_p, [c_void_p, c_int64]),
    ("pa_encoder_inflate_bgzf", ctypes.c_int, [c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                               c_void_p]),
    ("pa_encoder_walk_records", ctypes.c_int, [c_void_p, c_int64, c_void_p, c_int32, c_int32, c_void_p, c_int64,
                                               ctypes.POINTER(c_int64), c_void_p]),
    ("pa_encoder_region_reads", ctypes.c_int, [c_void_p, c_void_p, c_int32]),
    ("pa_encoder_set_host_threads", ctypes.c_int, [c_void_p, c_int32]),
    ("pa_encoder_last_timing", ctypes.c_int, [c_void_p, c_void_p, c_int32]),
    ("pa_encoder_batch_stats", ctypes.c_int, [c_void_p, c_void_p, c_int32]),
    ("pa_encoder_get_results", ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                              c_void_p, c_int64, ctypes.POINTER(c_int64)]),
    ("pa_encoder_device_images", c_void_p, [c_void_p]),
    ("pa_polish_encoder_generate_summary", ctypes.c_int, [c_void_p, c_void_p, c_int64, c_int64,
                                                          ctypes.POINTER(c_int64)]),
    ("pa_polish_encoder_generate_summary_batch", ctypes.c_int, [c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    ("pa_polish_encoder_stage_batch", ctypes.c_int, [c_void_p, c_int32, c_void_p, c_void_p, c_void_p]),
    ("pa_polish_encoder_run_staged", ctypes.c_int, [c_void_p, c_void_p]),
    ("pa_polish_encoder_batch_stats", ctypes.c_int, [c_void_p, c_void_p, c_int32]),
    ("pa_polish_encoder_get_results", ctypes.c_int, [c_void_p, c_void_p, c_void_p]),
    ("pa_polish_encoder_last_timing", ctypes.c_int, [c_void_p, c_void_p, c_int32]),
    ("pa_polish_chain_run", ctypes.c_int, [c_void_p, c_int32, c_void_p, c_void_p, c_int64, c_void_p, c_int32, c_void_p, c_void_p,
                                           c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    ("pa_polish_chain_chunks", ctypes.c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    ("pa_polish_chain_device_chunks", ctypes.c_int, [c_void_p, c_void_p]),
    ("pa_polish_chain_last_timing", ctypes.c_int, [c_void_p, c_void_p, c_int32, c_void_p, c_int32]),
    # include/pepper_amd_realign.h
    ("pa_realigner_create", ctypes.c_int, [c_int32, c_void_p, ctypes.POINTER(c_void_p)]),
    ("pa_realigner_destroy", None, [c_void_p]),
    ("pa_realigner_align", ctypes.c_int, [c_void_p, c_char_p, c_int64, c_int64, c_int32, c_void_p, c_void_p, c_void_p,
                                          c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                          ctypes.POINTER(c_int64)]),
    ("pa_realigner_align_windows", ctypes.c_int, [c_void_p, c_int32, c_char_p, c_void_p, c_void_p, c_int32] + [c_void_p] * 10 +
                                                  [ctypes.POINTER(c_int64)]),
    ("pa_realigner_copy_cigars", ctypes.c_int, [c_void_p, c_int32, c_void_p, c_void_p, c_void_p]),
    ("pa_realigner_stage_ticks", ctypes.c_int, [c_void_p, c_void_p]),
    # include/pepper_amd_io_device.h
    ("pa_inflater_create", ctypes.c_int, [c_int32, ctypes.POINTER(c_void_p)]),
    ("pa_inflater_destroy", None, [c_void_p]),
    ("pa_inflater_inflate", ctypes.c_int, [c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                           c_int64, c_int32]),
    ("pa_inflater_last_kernel_ms", ctypes.c_int, [c_void_p, ctypes.POINTER(c_double)]),
    ("pa_realigner_last_timing", ctypes.c_int, [c_void_p, ctypes.POINTER(c_double), ctypes.POINTER(c_double),
                                                ctypes.POINTER(c_int64)]),
]

_lib = None


class PepperAmdError(RuntimeError):
    pass


def load():
    """Load libpepper_amd.so (built in-tree by pepper_amd.build).  Fails loudly if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PepperAmdError(
            f"{LIB_PATH} is missing: run `python -m pepper_amd.build` (hipcc --offload-arch=gfx950). "
            "pepper_amd has no CPU fallback.")
    # torch bundles its own libamdhip64 (SONAME libamdhip64.so.7, same as /opt/rocm's).  Import
    # torch FIRST so our NEEDED libamdhip64.so.7 resolves to the runtime torch already loaded;
    # the other order puts two HIP runtimes in one process and the second one sees no device.
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, restype, argtypes in SYMBOLS:
        fn = getattr(lib, name)          # AttributeError here = header/library mismatch
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


# include/pepper_amd.h:20-24
PA_ERR_INVALID = 1
PA_ERR_HIP = 2
PA_ERR_NO_DEVICE = 3
PA_ERR_UNSUPPORTED = 4


def check(rc):
    if rc != PA_OK:
        msg = load().pa_last_error()
        err = PepperAmdError(f"pepper_amd error {rc}: {msg.decode() if msg else '?'}")
        err.code = rc
        raise err


def _as_numpy_f32(v):
    if hasattr(v, "detach"):
        v = v.detach().cpu().numpy()
    return np.ascontiguousarray(np.asarray(v), dtype=np.float32)


def marshal_state_dict(state_dict):
    """-> (names[], data[], numel[], n, keepalive) for pa_*_create."""
    items = [(k, _as_numpy_f32(v)) for k, v in state_dict.items()]
    n = len(items)
    names = (c_char_p * n)(*[k.encode() for k, _ in items])
    data = (c_void_p * n)(*[a.ctypes.data for _, a in items])
    numel = (c_int64 * n)(*[a.size for _, a in items])
    return names, data, numel, n, items


def profile_dict(handle):
    """{label: {"ms": total_ms, "launches": n, "flops": total_flops}} for a model handle."""
    lib = load()
    out = {}
    count = lib.pa_profile_count(handle)
    if count < 0:
        check(1)
    buf = ctypes.create_string_buffer(64)
    for i in range(count):
        ms, launches, flops = c_double(), c_int64(), c_double()
        check(lib.pa_profile_get(handle, i, buf, 64, ctypes.byref(ms), ctypes.byref(launches),
                                 ctypes.byref(flops)))
        out[buf.value.decode()] = {"ms": ms.value, "launches": launches.value, "flops": flops.value}
    return out
