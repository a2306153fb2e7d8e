"""The unit-split step loops (calls of at most 1024 windows: the hidden units of a tile on eight workgroups that exchange h_t
through memory every step, pepper_amd/csrc/rnn_h2.hip lstm_rec_h2_split_kernel) under contention: 2 000 calls of 512 windows
while a second process keeps every CU busy with 16 384-window passes.  Every result must be bit for bit one of the two the
handle can legitimately return -- the split schedule's, or (for a call whose workgroups did not meet and that was run again)
the ordinary small-call schedule's -- and both lie within 1e-5 of each other; the number of calls run again is reported."""
import ctypes
import multiprocessing as mp
import os
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _handle(lib, _lib, synthetic):
    sd = synthetic.variant_state_dict(seed=0)
    cfg = _lib.VariantConfig(26, 33, 1, 3, 0, 16384)
    names, data, numel, n, keep = _lib.marshal_state_dict(sd)
    h = ctypes.c_void_p()
    _lib.check(lib.pa_variant_create(ctypes.byref(cfg), names, data, numel, n, None, ctypes.byref(h)))
    return h


def _saturate(ready, stop):
    import torch
    from pepper_amd import _lib, synthetic
    lib = _lib.load()
    h = _handle(lib, _lib, synthetic)
    x = synthetic.variant_windows_device(16384, seed=9, device=torch.device("cuda", 0))
    probs = torch.empty((16384, 3), dtype=torch.float32, device="cuda")
    _lib.check(lib.pa_variant_forward_device(h, x.data_ptr(), 16384, probs.data_ptr(), None))
    _lib.check(lib.pa_synchronize(h))
    ready.set()
    while not stop.is_set():
        for _ in range(8):
            _lib.check(lib.pa_variant_forward_device(h, x.data_ptr(), 16384, probs.data_ptr(), None))
        _lib.check(lib.pa_synchronize(h))
    lib.pa_variant_destroy(h)


def test_two_thousand_split_calls_beside_a_saturating_process():
    from pepper_amd import _lib, synthetic
    lib = _lib.load()
    b, calls, batches = 512, 2000, 8
    xs = [np.ascontiguousarray(synthetic.variant_windows(b, seed=40 + k)) for k in range(batches)]
    split = _handle(lib, _lib, synthetic)
    os.environ["PA_UNIT_SPLIT"] = "0"
    try:
        plain = _handle(lib, _lib, synthetic)
    finally:
        os.environ.pop("PA_UNIT_SPLIT", None)

    def run(h, x):
        out = np.empty((b, 3), np.float32)
        _lib.check(lib.pa_variant_forward_host(h, x.ctypes.data, b, out.ctypes.data, None))
        return out
    want_split = [run(split, x) for x in xs]          # undisturbed
    want_plain = [run(plain, x) for x in xs]
    before = ctypes.c_int64(-1)
    _lib.check(lib.pa_variant_split_fallbacks(split, ctypes.byref(before)))
    assert before.value == 0                           # the undisturbed calls did take the split schedule
    for a, p in zip(want_split, want_plain):
        assert np.abs(a - p).max() < 1e-5
    ctx = mp.get_context("spawn")
    ready, stop = ctx.Event(), ctx.Event()
    other = ctx.Process(target=_saturate, args=(ready, stop))
    other.start()
    try:
        assert ready.wait(timeout=180)
        t0 = time.perf_counter()
        as_split = as_plain = 0
        for k in range(calls):
            got = run(split, xs[k % batches])
            if np.array_equal(got, want_split[k % batches]):
                as_split += 1
            else:
                assert np.array_equal(got, want_plain[k % batches]), "call %d: neither schedule's result" % k
                as_plain += 1
        dt = time.perf_counter() - t0
    finally:
        stop.set()
        other.join(timeout=120)
        if other.is_alive():
            other.kill()
    again = ctypes.c_int64(-1)
    _lib.check(lib.pa_variant_split_fallbacks(split, ctypes.byref(again)))
    print("\nunit-split under contention: %d calls of %d windows in %.2f s (%.2f ms per call); %d run again; results: %d split, "
          "%d ordinary schedule" % (calls, b, dt, 1e3 * dt / calls, again.value, as_split, as_plain))
    # a call that was run again is followed by 256 calls on the ordinary schedule (the handle's hold-off)
    assert as_plain <= 257 * again.value
    assert as_split + as_plain == calls
    lib.pa_variant_destroy(split)
    lib.pa_variant_destroy(plain)
