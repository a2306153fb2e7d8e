#!/usr/bin/env python
"""Throughput benchmark of the RNN inference hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--model variant|polish]

A "step" is one pass of the hot path over one resident batch of synthetic summaries per GPU
(variant: 16384 candidate windows int8 [.,33,26] = 32 reference batches of 512; polish: 2048
chunks uint8 [.,1000,10] = 19 windows each).  Inputs are in HBM before the timed region; the
timed region is bracketed by barrier + synchronize on both sides and the maximum over ranks is
reported.  Rank 0 prints one JSON line (see DESIGN.md "Measurement").
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

from pepper_amd import _lib, synthetic  # noqa: E402

F32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
# kernels whose label carries "_h2" evaluate every f32-accurate product as three v_mfma_f32_32x32x16_f16
# (hi*hi + hi*lo + lo*hi, f32 accumulate): their ceiling in algorithmic (f32-equivalent) FLOP/s is the
# dense f16 MFMA peak (2.5 PFLOP/s, same guide) divided by three
H2_MFMA_PEAK_TFLOPS = 2500.0 / 3.0


def kernel_peak(label):
    return H2_MFMA_PEAK_TFLOPS if "_h2" in label else F32_MFMA_PEAK_TFLOPS
VARIANT_FLOP_PER_WINDOW = 2 * 80_664_064      # SURVEY.md 8(a) A8 / BASELINE.md section 2
POLISH_FLOP_PER_WINDOW = 2 * 40_217_600       # per 100-step window (A12)
POLISH_WINDOWS_PER_CHUNK = 19


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--model", choices=["variant", "polish", "ns-literal", "realign"], default="variant",
                    help="variant = BASELINE configs[1] shapes (the headline); polish = configs[4]; ns-literal = the polish "
                         "stack at the north_star's literal synthetic shape (100-step windows x 100 features; not a "
                         "reference shape, reported separately); realign = the polish read re-aligner (SSW) on "
                         "regions of 1500 simulated reads, reads/s and DP cell updates/s")
    ap.add_argument("--per-gpu", type=int, default=0, help="windows (variant) / chunks (polish) per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--cpu-worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-threads", type=int, default=1, help=argparse.SUPPRESS)
    return ap.parse_args()


def dist_setup(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local))
    else:
        torch.cuda.set_device(0)
        local = 0
    return world, rank, local


def broadcast_state_dict(make_sd, shapes, world, rank, dev):
    """Rank 0 owns the checkpoint; everyone else receives one packed fp32 blob over RCCL
    (the only collective on the path: SURVEY.md 8(e))."""
    if world == 1:
        return make_sd()
    from pepper_amd.parallel import broadcast_numpy_state_dict
    return broadcast_numpy_state_dict(make_sd if rank == 0 else None, shapes, device=dev)


def _cpu_runner(model_kind):
    from oracle import torch_port
    if model_kind == "variant":
        sd = synthetic.variant_state_dict(seed=0)
        model = torch_port.load_numpy_state_dict(torch_port.VariantPort(), sd)
        x = torch.from_numpy(synthetic.variant_windows(512, seed=1)).float()
        return (lambda: model(x)), 512, "batch 512 x [33,26] V-syn windows, torch.nn CPU forward"
    sd = synthetic.polish_state_dict(seed=0)
    model = torch_port.load_numpy_state_dict(torch_port.PolishPort(), sd)
    img = synthetic.polish_chunks(32, seed=1)
    return ((lambda: torch_port.polish_predict_chunks(model, img, 128)), 32 * POLISH_WINDOWS_PER_CHUNK,
            "batch 32 chunks x [1000,10] P-syn (19 windows each), torch.nn CPU loop")


def cpu_worker(model_kind, threads, seconds):
    """One worker of the reference's CPU scheme (RunInference.py:94-116: `threads` callers, one
    intra-op thread each, own file shard).  Prints {"windows", "seconds"}."""
    torch.set_num_threads(threads)
    run, units, _ = _cpu_runner(model_kind)
    with torch.no_grad():
        run()
        t0 = time.perf_counter()
        n = 0
        while n < 1 or time.perf_counter() - t0 < seconds:
            run()
            n += 1
        dt = time.perf_counter() - t0
    print(json.dumps({"windows": units * n, "seconds": dt}))


def cpu_baseline_workers(model_kind, seconds):
    """Aggregate of P single-thread workers running concurrently in fresh interpreters."""
    import subprocess
    ncpu = os.cpu_count() or 1
    procs = max(1, min(64, ncpu // 2))   # ~0.5 GB RSS each (torch + 47 MB of weights): bounded on purpose
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="", OMP_NUM_THREADS="1", MKL_NUM_THREADS="1")
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-worker", "--model", model_kind, "--cpu-threads", "1",
           "--cpu-seconds", str(seconds)]
    t0 = time.perf_counter()
    ps = [subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for _ in range(procs)]
    outs = []
    for p in ps:
        try:
            out, _ = p.communicate(timeout=6 * seconds + 120)
            outs.append(json.loads(out.strip().splitlines()[-1]))
        except Exception:
            p.kill()
    wall = time.perf_counter() - t0
    if not outs:
        return None
    windows = sum(o["windows"] for o in outs)
    span = max(o["seconds"] for o in outs)
    return {"value": windows / span, "unit": "windows/s", "cores": len(outs), "kind": "port",
            "sample": f"{len(outs)} concurrent single-thread workers (the reference's distributed_cpu scheme), "
                      f"each looping the torch.nn forward for {seconds:.0f} s; aggregate over {span:.1f} s "
                      f"(wall incl. start-up {wall:.0f} s)"}


def cpu_baseline(model_kind, seconds):
    """The torch.nn port of the reference forward (oracle/torch_port.py) on the host cores.

    ATen's small-GEMM RNN path collapses when oversubscribed (all 256 hyper-threads of the GPU
    box: >20 s per batch), so a short sweep picks the best thread count first; the reported
    `cores` is the thread count actually used for the timed sample.  Total CPU time is bounded.
    """
    ncpu = os.cpu_count() or 1
    run, units, sample = _cpu_runner(model_kind)
    unit = "windows/s"
    deadline = time.perf_counter() + 3.0 * seconds      # hard bound on the whole leg
    best_t, best_rate = None, 0.0
    with torch.no_grad():
        for nt in sorted({min(ncpu, c) for c in (8, 16, 32, 64)}):
            if time.perf_counter() > deadline - seconds:
                break
            torch.set_num_threads(nt)
            run()
            t0 = time.perf_counter()
            run()
            rate = units / (time.perf_counter() - t0)
            if rate > best_rate:
                best_t, best_rate = nt, rate
        torch.set_num_threads(best_t)
        t0 = time.perf_counter()
        n = 0
        while n < 1 or (time.perf_counter() - t0 < seconds and time.perf_counter() < deadline):
            run()
            n += 1
        dt = time.perf_counter() - t0
    return {"value": units * n / dt, "unit": unit, "cores": best_t, "host_logical_cpus": ncpu,
            "kind": "port", "sample": f"{n} x ({sample}), {best_t} threads (best of sweep), {dt:.1f} s"}


def realign_bench(args):
    """Secondary workload: one step = one polish region (1 kb of draft + 20 safe bases, 1500 region-clipped reads -- the
    reference's cap per region -- 85 % of them spanning the window, nanopore-like error mix) through pa_realigner_align,
    host buffers in, CIGARs out.  Not the headline metric."""
    import ctypes
    from oracle import ssw
    from pepper_amd.polish.PEPPER import ReadAligner
    rng = np.random.default_rng(5)
    reference = "".join("ACGT"[k] for k in rng.integers(0, 4, 1020))
    pos, seqs = synthetic.simulate_clipped_reads(rng, reference, 0, args.per_gpu or 1500, sub=0.04, ins=0.03, dele=0.04, min_len=200,
                                   full_span=0.85)
    blob = [q.encode() for q in seqs]
    off = np.zeros(len(seqs) + 1, np.int64)
    np.cumsum([len(b) for b in blob], out=off[1:])
    seq = np.frombuffer(b"".join(blob), np.uint8)
    aligner = ReadAligner(0, len(reference), reference)
    for _ in range(args.warmup):
        out = aligner.align_arrays(pos, off, seq)
    lib, h = __import__("pepper_amd.polish.PEPPER", fromlist=["_realigner"])._realigner(0)
    ends = band = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = aligner.align_arrays(pos, off, seq)
        a, b, c = ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
        _lib.check(lib.pa_realigner_last_timing(h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
        ends += a.value
        band += b.value
    dt = time.perf_counter() - t0
    cells = c.value
    # a region at ordinary coverage: 60 reads (latency of one call)
    n60 = min(60, len(seqs))
    off60 = off[:n60 + 1].copy()
    seq60 = seq[:int(off60[-1])]
    aligner.align_arrays(pos[:n60], off60, seq60)
    t60 = time.perf_counter()
    for _ in range(args.steps):
        aligner.align_arrays(pos[:n60], off60, seq60)
    ms60 = (time.perf_counter() - t60) / args.steps * 1e3
    # the image generator runs its regions on worker threads (one handle and stream each): aggregate over 4 of them
    import threading

    def worker():
        mine = ReadAligner(0, len(reference), reference)
        mine.align_arrays(pos, off, seq)
        barrier.wait()
        for _ in range(args.steps):
            mine.align_arrays(pos, off, seq)
        barrier.wait()
    barrier = threading.Barrier(5)
    threads = [threading.Thread(target=worker) for _ in range(4)]
    for th in threads:
        th.start()
    barrier.wait()
    t2 = time.perf_counter()
    barrier.wait()
    dt4 = time.perf_counter() - t2
    for th in threads:
        th.join()
    # CPU: the reference's own SSW build where it travelled with the snapshot, else the scalar restatement
    kind = "reference" if ssw.have_reference() else "port"
    fn = ssw.align_reference if kind == "reference" else ssw.align
    t1 = time.perf_counter()
    done = 0
    for p, q in zip(pos, seqs):
        fn(reference[p:], q)
        done += 1
        if time.perf_counter() - t1 > args.cpu_seconds:
            break
    cpu_dt = time.perf_counter() - t1
    n = len(seqs)
    # one score pass visits `cells`; the pipeline runs about three of them per read (8-bit prefix, 16-bit, reverse)
    print(json.dumps({
        "metric": "polish read re-alignment, reads/s (secondary workload)", "value": n * args.steps / dt, "unit": "reads/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
        "reads_per_s_4_worker_threads": 4 * n * args.steps / dt4, "ms_per_60_read_region": ms60,
        "config": {"workload": f"{n} region-clipped reads (mean {int(off[-1]) // n} bases) against a 1020-base draft window, "
                               "SSW scoring 4/6/8/2, host buffers in, CIGARs out"},
        "kernels": {"sw_ends_kernel": {"avg_ms": ends / args.steps, "gcups_one_pass_equiv": cells / (ends / args.steps * 1e-3) / 1e9},
                    "band_kernel": {"avg_ms": band / args.steps}},
        "roofline": {"bound": "valu", "kernel": "sw_ends_kernel", "achieved": 2.0 * cells / (ends / args.steps * 1e-3) / 1e9,
                     "peak": 39321.6 / 12.5, "unit": "G cell updates/s",
                     "frac": 2.0 * cells / (ends / args.steps * 1e-3) / 1e9 / (39321.6 / 12.5), "traffic": None,
                     "note": "integer DP on the vector ALUs, neither HBM nor MFMA bound: peak = 256 CU x 64 lanes x 2.4 GHz "
                             "int32 instructions / 12.5 instructions per cell (the kernel's ISA); achieved counts the forward "
                             "and the reverse pass (2 x n x m cells per read; the 8-bit prefix pass is not counted)"},
        "cpu_baseline": {"value": done / cpu_dt, "unit": "reads/s", "cores": 1, "kind": kind,
                         "sample": f"{done} of the same reads through {'the reference SSW build (oracle/_ref)' if kind == 'reference' else 'oracle/ssw_oracle.cpp'}, one thread, {cpu_dt:.1f} s"},
        "speedup_vs_cpu_baseline": (n * args.steps / dt) / (done / cpu_dt)}))


def main():
    args = parse()
    if args.cpu_worker:
        cpu_worker(args.model, args.cpu_threads, args.cpu_seconds)
        return
    if args.model == "realign":
        torch.cuda.set_device(0)
        realign_bench(args)
        return
    world, rank, local = dist_setup(args)
    if world != args.gpus and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
    dev = torch.device("cuda", local)
    lib = _lib.load()
    import ctypes
    stream = torch.cuda.Stream(device=dev)

    if args.model == "variant":
        per = args.per_gpu or 16384
        sd = broadcast_state_dict(lambda: synthetic.variant_state_dict(seed=0),
                                  synthetic.variant_param_shapes(), world, rank, dev)
        cfg = _lib.VariantConfig(26, 33, 1, 3, local, per)
        names, data, numel, n, keep = _lib.marshal_state_dict(sd)
        handle = ctypes.c_void_p()
        _lib.check(lib.pa_variant_create(ctypes.byref(cfg), names, data, numel, n,
                                         ctypes.c_void_p(stream.cuda_stream), ctypes.byref(handle)))
        x = torch.from_numpy(synthetic.variant_windows(per, seed=synthetic.VSYN_SEED + rank)).to(dev)
        out = torch.empty((per, 3), dtype=torch.float32, device=dev)

        def step():
            _lib.check(lib.pa_variant_forward_device(handle, x.data_ptr(), per, out.data_ptr(), None))
        windows_per_unit, flop_per_window = 1, VARIANT_FLOP_PER_WINDOW
        workload = ("V-syn: int8 [N,33,26] candidate windows, variant bi-LSTM(26->256)x2 + MLP head, "
                    "F=26 H=256 L=1 (BASELINE configs[1] shapes)")
    else:
        per = args.per_gpu or 16384
        feat = 100 if args.model == "ns-literal" else 10
        sd = broadcast_state_dict(lambda: synthetic.polish_state_dict(seed=0, image_features=feat),
                                  synthetic.polish_param_shapes(image_features=feat), world, rank, dev)
        cfg = _lib.PolishConfig(feat, 128, 1, 5, 1000, 100, 50, 50, local, per)
        names, data, numel, n, keep = _lib.marshal_state_dict(sd)
        handle = ctypes.c_void_p()
        _lib.check(lib.pa_polish_create(ctypes.byref(cfg), names, data, numel, n,
                                        ctypes.c_void_p(stream.cuda_stream), ctypes.byref(handle)))
        if feat == 10:
            x = torch.from_numpy(synthetic.polish_chunks(per, seed=synthetic.PSYN_SEED + rank)).to(dev)
        else:   # rows of small counts spread over 100 columns, generated on the device (1.6 GB per 16384 chunks)
            gen = torch.Generator(device=dev).manual_seed(synthetic.PSYN_SEED + rank)
            x = torch.poisson(torch.full((per, 1000, feat), 2.5, device=dev), generator=gen).clamp_(0, 254).to(torch.uint8)
        lab = torch.empty((per, 1000), dtype=torch.uint8, device=dev)
        ph = torch.empty((per, 1000), dtype=torch.uint8, device=dev)

        def step():
            _lib.check(lib.pa_polish_predict_device(handle, x.data_ptr(), per, lab.data_ptr(),
                                                    ph.data_ptr(), None))
        windows_per_unit = POLISH_WINDOWS_PER_CHUNK
        flop_per_window = POLISH_FLOP_PER_WINDOW + 2.0 * 100 * 2 * 384 * (feat - 10)
        workload = ("P-syn: uint8 [N,1000,10] chunks, polish bi-GRU(10->128)x2 + dense, 19 windows of "
                    "100 steps with hidden carry (BASELINE configs[4] shapes)") if feat == 10 else (
                    "NS-literal: uint8 [N,1000,100] chunks = 19 windows of 100 steps x 100 features through the polish "
                    "bi-GRU(100->128)x2 + dense with hidden carry; the north_star's literal synthetic shape, not a "
                    "reference shape (SURVEY.md section 0), no CPU baseline")

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()

    torch.cuda.synchronize(dev)
    for _ in range(args.warmup):
        step()
    _lib.check(lib.pa_synchronize(handle))
    _lib.check(lib.pa_profile_enable(handle, 1))
    barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    _lib.check(lib.pa_synchronize(handle))
    torch.cuda.synchronize(dev)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    prof = _lib.profile_dict(handle)
    _lib.check(lib.pa_profile_enable(handle, 0))

    if rank == 0:
        windows = world * args.steps * per * windows_per_unit
        value = windows / dt
        kern = {}
        for label, p in prof.items():
            avg_ms = p["ms"] / max(1, p["launches"])
            tf = (p["flops"] / max(1, p["launches"])) / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
            kern[label] = {"launches_per_step": p["launches"] / args.steps, "avg_ms": round(avg_ms, 4),
                           "share": 0.0, "tflops": round(tf, 2), "frac_of_peak": round(tf / kernel_peak(label), 4)}
        tot = sum(p["ms"] for p in prof.values()) or 1.0
        for label, p in prof.items():
            kern[label]["share"] = round(p["ms"] / tot, 4)
        dom = max(prof, key=lambda k: prof[k]["ms"])
        d = prof[dom]
        ach = (d["flops"] / d["launches"]) / (d["ms"] / d["launches"] * 1e-3) / 1e12
        line = {
            "metric": "inference windows/sec (whole node)",
            "value": value, "unit": "windows/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if not any("_h2" in k for k in prof) else "f32 via f16 hi/lo split operands (3 MFMAs per product), f32 accumulate",
            "data": "synthetic",
            "config": {"workload": workload, "per_gpu_per_step": per,
                       "units": "windows" if args.model == "variant" else "chunks (x19 windows)",
                       "reference_hdf5_batch": 512 if args.model == "variant" else 128,
                       "weights": "seeded random init (pepper_amd.synthetic), fp32",
                       "parallelism": f"region-shard x{world}, RCCL weight broadcast only"},
            "roofline": {"bound": "mfma", "kernel": dom, "achieved": ach, "peak": kernel_peak(dom),
                         "unit": "TFLOP/s", "frac": ach / kernel_peak(dom), "traffic": None,
                         "arithmetic": ("3 x v_mfma_f32_32x32x16_f16 per product (f16 hi/lo split operands, f32 "
                                        "accumulate); peak = 2.5 PFLOP/s dense f16 / 3") if "_h2" in dom
                         else "v_mfma_f32_32x32x2_f32"},
            "end_to_end_tflops": value * flop_per_window / 1e12,
            "kernels": kern,
        }
        if world == 1 and not args.no_cpu_baseline and args.model != "ns-literal":
            # whole-box CPU number = the reference's own scheme (many single-thread workers); the
            # single-process multi-thread figure is kept beside it
            single = cpu_baseline(args.model, args.cpu_seconds)
            multi = cpu_baseline_workers(args.model, args.cpu_seconds)
            line["cpu_baseline"] = multi if multi and multi["value"] > single["value"] else single
            line["cpu_baseline_single_process"] = single
            line["speedup_vs_cpu_baseline"] = value / line["cpu_baseline"]["value"]
        print(json.dumps(line))

    if args.model == "variant":
        lib.pa_variant_destroy(handle)
    else:
        lib.pa_polish_destroy(handle)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
