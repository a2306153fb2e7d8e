"""What the unit-split step loops (calls of at most 1024 windows) do when several processes share the GPU: N processes, each
with its own handle, make `calls` calls of `b` windows at the same time; every process reports its time per call, how many
calls were run again because a tile's workgroups did not meet (pa_variant_split_fallbacks) and the largest difference from
the results of an undisturbed first call.
    python tools/split_contention.py [processes=4] [b=512] [calls=300] [threads]
"threads": the N callers are threads of ONE process (each with its own handle and streams), as the lanes' two blocks in flight are."""
import ctypes
import json
import multiprocessing as mp
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def worker(rank, b, calls, start, out):
    import numpy as np
    from pepper_amd import _lib, synthetic
    lib = _lib.load()
    sd = synthetic.variant_state_dict(seed=0)
    cfg = _lib.VariantConfig(26, 33, 1, 3, 0, 16384)
    names, data, numel, n, keep = _lib.marshal_state_dict(sd)
    h = ctypes.c_void_p()
    _lib.check(lib.pa_variant_create(ctypes.byref(cfg), names, data, numel, n, None, ctypes.byref(h)))
    x = np.ascontiguousarray(synthetic.variant_windows(b, seed=3))
    want = np.empty((b, 3), np.float32)
    _lib.check(lib.pa_variant_forward_host(h, x.ctypes.data, b, want.ctypes.data, None))     # alone (the others are still starting)
    start.wait()
    got = np.empty((b, 3), np.float32)
    worst = 0.0
    t0 = time.perf_counter()
    for _ in range(calls):
        _lib.check(lib.pa_variant_forward_host(h, x.ctypes.data, b, got.ctypes.data, None))
        worst = max(worst, float(np.abs(got - want).max()))
    dt = time.perf_counter() - t0
    again = ctypes.c_int64(-1)
    _lib.check(lib.pa_variant_split_fallbacks(h, ctypes.byref(again)))
    out.put({"rank": rank, "ms_per_call": round(1e3 * dt / calls, 3), "calls_run_again": again.value, "max_abs_difference": worst})


def main():
    procs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    b = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    calls = int(sys.argv[3]) if len(sys.argv) > 3 else 300
    if len(sys.argv) > 4 and sys.argv[4] == "threads":
        import queue
        import threading
        start, out = threading.Barrier(procs), queue.Queue()
        ts = [threading.Thread(target=worker, args=(r, b, calls, start, out)) for r in range(procs)]
        for t in ts:
            t.start()
        rows = [out.get(timeout=120) for _ in ts]
        for t in ts:
            t.join(timeout=30)
        print(json.dumps({"threads_of_one_process": procs, "windows_per_call": b, "calls": calls, "ranks": sorted(rows, key=lambda r: r["rank"])}))
        return
    ctx = mp.get_context("spawn")
    start, out = ctx.Barrier(procs), ctx.Queue()
    ps = [ctx.Process(target=worker, args=(r, b, calls, start, out)) for r in range(procs)]
    for p in ps:
        p.start()
    rows = [out.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(timeout=30)
    print(json.dumps({"processes": procs, "windows_per_call": b, "calls": calls, "ranks": sorted(rows, key=lambda r: r["rank"])}))


if __name__ == "__main__":
    main()
