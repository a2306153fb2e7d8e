#!/usr/bin/env python
"""Throughput benchmark of the RNN inference hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--model variant|polish|ns-literal|realign|encoder]

What one "step" is (DESIGN.md "Measurement"): one pass of the hot path over one batch of synthetic
summaries per GPU, as the reference's predict loop runs it -- page-locked host buffer -> H2D of the packed int8 / uint8
summaries -> forward -> D2H of the results (predict_distributed_gpu.py:58-67 does `.cuda()` ... `.cpu()` per batch) --
through the C ABI's host entry points, whose device passes (16384 windows / 16384 chunks) overlap the copies of the
neighbouring passes with the kernels on separate HIP streams.  Variant: 2^18 windows per step, taken round robin from a
pool of 2^20 distinct V-syn windows per GPU; polish: 32768 chunks (x 19 windows) per step from a pool of 65536 P-syn
chunks.  `value` = windows of all ranks / max-over-ranks wall time of exactly K steps bracketed by barrier +
synchronize; the device-resident rate (inputs already in HBM, what round 1 reported) is kept as
`device_resident` beside it, and `batch512` gives the reference's default batch through one call.

`--gpus N` with N > 1 and no WORLD_SIZE in the environment re-launches itself as N ranks under torch.distributed.run
(one process per GPU, RCCL); under an external torchrun launch it uses the ranks it is given.  Rank 0 prints ONE JSON
line.
"""
import argparse
import ctypes
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")     # before the HIP runtime starts (pepper_amd/__init__.py says why)

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

from pepper_amd import _lib, synthetic  # noqa: E402

F32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
F16_MFMA_PEAK_TFLOPS = 2500.0     # same guide: dense f16/bf16 MFMA peak (never the 2:1 sparsity figure)
# kernels whose label carries "_h2" evaluate every f32-accurate product as three v_mfma_f32_32x32x16_f16
# (hi*hi + hi*lo + lo*hi, f32 accumulate): their ceiling in algorithmic (f32-equivalent) FLOP/s is the
# dense f16 MFMA peak divided by three
H2_MFMA_PEAK_TFLOPS = F16_MFMA_PEAK_TFLOPS / 3.0


def kernel_peak(label):
    return H2_MFMA_PEAK_TFLOPS if "_h2" in label else F32_MFMA_PEAK_TFLOPS


def newest_profile(suffix):
    """(path relative to the repository, round) of the committed profile profiles/rNN_<suffix> with the highest round, or (None, 0)."""
    import glob
    import re
    best = (None, 0)
    for path in glob.glob(os.path.join(REPO, "profiles", "r[0-9][0-9]_" + suffix)):
        m = re.match(r"r(\d+)_", os.path.basename(path))
        if m and int(m.group(1)) > best[1]:
            best = (os.path.relpath(path, REPO), int(m.group(1)))
    return best


def counters_stale(profile_round, *sources):
    """True when a kernel source the counters describe changed in a later round than the profile was taken in
    (profiles/kernel_rounds.json, written from the history by tools/kernel_rounds.py -- the GPU box has no .git)."""
    try:
        with open(os.path.join(REPO, "profiles", "kernel_rounds.json")) as fh:
            table = json.load(fh)["last_changed_in_round"]
    except (OSError, KeyError, ValueError):
        return None
    return any(table.get(src, 0) > profile_round for src in sources)


VARIANT_FLOP_PER_WINDOW = 2 * 80_664_064      # SURVEY.md 8(a) A8 / BASELINE.md section 2
POLISH_FLOP_PER_WINDOW = 2 * 40_217_600       # per 100-step window (A12)
POLISH_WINDOWS_PER_CHUNK = 19
VARIANT_BYTES_PER_WINDOW = 33 * 26 + 12       # int8 summary in, float32 probabilities out
POLISH_BYTES_PER_CHUNK = 1000 * 10 + 2000     # uint8 summary in, label + phred bytes out


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--model", choices=["variant", "polish", "ns-literal", "realign", "encoder", "polish-encoder"], default="variant",
                    help="variant = BASELINE configs[1] shapes (the headline); polish = configs[4]; ns-literal = the polish "
                         "stack at the north_star's literal synthetic shape (100-step windows x 100 features; not a "
                         "reference shape, reported separately); realign = the polish read re-aligner (SSW) on "
                         "regions of 1500 simulated reads, reads/s and DP cell updates/s; encoder = the variant pileup -> "
                         "summary encoder (the other half of north_star's hot path) on a batch of 64 E-syn regions of 100 kb "
                         "at 60x, aligned bases/s, HBM roofline, the reference's own C++ as the CPU baseline; polish-encoder = the "
                         "polish SummaryGenerator on 256 regions of 1000 + 2 x 100 positions at 60x, likewise")
    ap.add_argument("--workload", choices=["v-syn", "wg-syn"], default="v-syn",
                    help="variant model only.  v-syn (default): every rank streams its own pool, weak scaling.  wg-syn: a FIXED job of "
                         "24 chromosome-sized shards of V-syn windows dealt over the ranks (strong scaling; SURVEY.md 8(d)), once with "
                         "the reference's round robin and once largest-first onto the least loaded rank; per-rank times in the line")
    ap.add_argument("--per-gpu", type=int, default=0, help="windows (variant) / chunks (polish) per GPU per step")
    ap.add_argument("--pool", type=int, default=0, help="distinct windows / chunks in the page-locked host pool per GPU")
    ap.add_argument("--resident-only", action="store_true",
                    help="time the device-resident pass instead of the host-buffer path (kernel profiling under rocprofv3)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the device-resident and batch-512 legs")
    ap.add_argument("--no-secondary", action="store_true",
                    help="variant only: skip the `secondary` block (polish, encoder and the two HDF5 -> HDF5 pipelines)")
    ap.add_argument("--no-image-legs", action="store_true",
                    help="N > 1: skip the per-rank image generation legs (variant generate_images and the polish chain on each rank's own "
                         "synthetic BAM) that follow the timed model steps")
    ap.add_argument("--legs", default="", help="secondary block: only these legs (comma separated names of the `secondary` keys)")
    ap.add_argument("--full-line", action="store_true",
                    help="print the whole record (tens of KB) as the stdout line instead of the compact one; the whole record is "
                         "always written to gpurun_out/bench_full.json")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--cpu-worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-threads", type=int, default=1, help=argparse.SUPPRESS)
    return ap.parse_args()


def self_spawn(args):
    """`python bench.py --gpus N` on its own: become N ranks (the reference spawns its ranks itself too,
    pepper/modules/python/models/predict_distributed_gpu.py:150-166)."""
    import socket
    import subprocess
    ndev = torch.cuda.device_count()
    if ndev < args.gpus and os.environ.get("PEPPER_AMD_BENCH_SHARE_GPU") not in ("1", "2"):
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but only {ndev} HIP device(s) visible\n")
        sys.exit(2)
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd))


COLLECTIVE_NOTE = "none (one rank)"
BENCH_GROUP = None            # the RCCL process group when every rank could initialise it (dist_setup); None: the gloo default group
RCCL_ATTEMPT_LEFT_BEHIND = False   # the RCCL attempt failed on some rank: a half-made communicator may exist here; leave without tearing it down


def leave_group():
    """The end of an N-rank run.  After an RCCL attempt that some rank failed, a communicator may be half made here (its creation
    still blocked on a helper thread): tearing that down can block again, so the process leaves without it -- everything this rank
    had to print is flushed first."""
    import torch.distributed as dist
    if RCCL_ATTEMPT_LEFT_BEHIND:
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(0)
    dist.destroy_process_group()


def dist_setup(args):
    """-> (world, rank, device ordinal, ranks_seen).  ranks_seen comes out of a real all-reduce."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1:
        torch.cuda.set_device(0)
        return 1, 0, 0, 1
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    ndev = torch.cuda.device_count()
    # PEPPER_AMD_BENCH_SHARE_GPU=1: a plumbing check of the N-rank code path on a box with fewer GPUs (ranks share
    # devices, gloo instead of RCCL, the line says so); never a scaling number
    # (=2: the ranks share devices AND go through the backend agreement below -- RCCL refuses two ranks per device, so this
    # exercises the all-ranks fallback on a 1-GPU box)
    share_mode = os.environ.get("PEPPER_AMD_BENCH_SHARE_GPU")
    share = share_mode == "1" and ndev < world
    device = local % ndev if (share or (share_mode == "2" and ndev < world)) else local
    torch.cuda.set_device(device)
    global COLLECTIVE_NOTE
    if share:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        one = torch.ones(1)
        dist.all_reduce(one)
        COLLECTIVE_NOTE = "gloo (ranks share devices: plumbing check)"
        return world, rank, device, int(one.item())
    # The ranks AGREE on the backend before anyone commits to it: a gloo group comes up first (it works wherever the
    # rendezvous does), RCCL is tried as a second group on top of it, and the ranks' verdicts are summed over gloo -- RCCL is
    # used only if every rank's initialisation and first collective succeeded.  (The earlier form let each rank fall back on its
    # own: had RCCL failed on some ranks only, those would have re-initialised with gloo while the others sat in an RCCL
    # collective -- a hang instead of a soft failure.)  The data path has no collective; this carries one weight broadcast
    # and the timing barriers.
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pepper_amd.parallel import agree_on_rccl

    from pepper_amd.parallel import wait_bounded
    from datetime import timedelta
    probe_seconds = float(os.environ.get("PEPPER_AMD_RCCL_PROBE_SECONDS", 60))

    def try_rccl():
        # (on agree_on_rccl's helper thread: the current device is per thread)
        torch.cuda.set_device(device)
        group = dist.new_group(backend="nccl", timeout=timedelta(seconds=probe_seconds), device_id=torch.device("cuda", device))
        probe = torch.ones(1, device=torch.device("cuda", device))
        # no blocking wait and no device synchronize: a rank whose peers never enter polls until the deadline and then votes "failed"
        wait_bounded(dist.all_reduce(probe, group=group, async_op=True), probe_seconds, "RCCL probe all-reduce")
        if int(probe.item()) != world:
            raise RuntimeError("RCCL all-reduce over %d ranks returned %d" % (world, int(probe.item())))
        return group
    global BENCH_GROUP, RCCL_ATTEMPT_LEFT_BEHIND
    # the whole attempt is bounded too (communicator creation can block as well): 1.5 x the probe's own deadline
    BENCH_GROUP, failed, why = agree_on_rccl(world, try_rccl, timeout_s=1.5 * probe_seconds)
    RCCL_ATTEMPT_LEFT_BEHIND = BENCH_GROUP is None and failed > 0
    if why:
        sys.stderr.write("[bench] rank %d: RCCL unusable (%s)\n" % (rank, why))
    one = torch.ones(1)
    dist.all_reduce(one)
    COLLECTIVE_NOTE = ("nccl (RCCL), agreed over a gloo group" if BENCH_GROUP is not None else
                       "gloo (RCCL unusable on %d of %d ranks%s)" % (failed, world, (": " + why) if why else ""))
    return world, rank, device, int(one.item())


def broadcast_state_dict(make_sd, shapes, world, rank, dev):
    """Rank 0 owns the checkpoint; everyone else receives one packed fp32 blob over RCCL
    (the only collective on the path: SURVEY.md 8(e))."""
    if world == 1:
        return make_sd()
    import torch.distributed as dist
    from pepper_amd.parallel import broadcast_numpy_state_dict
    # over the RCCL group when every rank has it (dist_setup), over the gloo group otherwise
    return broadcast_numpy_state_dict(make_sd if rank == 0 else None, shapes, device=dev if BENCH_GROUP is not None else None,
                                      group=BENCH_GROUP)


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


def host_cores():
    """(physical cores, logical CPUs) of this host."""
    logical = os.cpu_count() or 1
    try:
        import psutil
        physical = psutil.cpu_count(logical=False) or logical
    except Exception:
        physical = max(1, logical // 2)
    return physical, logical


def worker_count():
    """Single-thread CPU workers to run at once: one per physical core, but never more than the CPUs this container may
    use (affinity and cgroup quota -- pepper_amd.hostinfo; the project's GPU boxes show 256 logical CPUs and grant 16)."""
    from pepper_amd.hostinfo import usable_cpus
    physical, _ = host_cores()
    return max(1, min(physical, usable_cpus()))


def cpu_note():
    from pepper_amd.hostinfo import cgroup_cpu_quota, usable_cpus
    q = cgroup_cpu_quota()
    return {"usable_cpus": usable_cpus(), "cgroup_cpu_quota": q}


def cpu_baseline_workers(model_kind, seconds):
    """Aggregate of single-thread workers running concurrently in fresh interpreters, ONE PER PHYSICAL CORE (the
    reference's distributed_cpu scheme on the whole box; each worker holds torch + 47 MB of weights, about 0.5 GB)."""
    import subprocess
    physical, logical = host_cores()
    procs = worker_count()
    try:
        import psutil
        procs = max(1, min(procs, int(psutil.virtual_memory().available / (0.8 * 2 ** 30))))
    except Exception:
        pass
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="", OMP_NUM_THREADS="1", MKL_NUM_THREADS="1")
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-worker", "--model", model_kind, "--cpu-threads", "1",
           "--cpu-seconds", str(seconds)]
    t0 = time.perf_counter()
    ps = [subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for _ in range(procs)]
    outs = []
    for p in ps:
        try:
            out, _ = p.communicate(timeout=6 * seconds + 180)
            outs.append(json.loads(out.strip().splitlines()[-1]))
        except Exception:
            p.kill()
    wall = time.perf_counter() - t0
    if not outs:
        return None
    windows = sum(o["windows"] for o in outs)
    span = max(o["seconds"] for o in outs)
    return dict({"value": windows / span, "unit": "windows/s", "cores": len(outs), "host_physical_cores": physical,
                 "host_logical_cpus": logical, "kind": "port",
                 "sample": f"{len(outs)} concurrent single-thread workers, one per CPU this container may use (the reference's "
                           f"distributed_cpu scheme), each looping the torch.nn forward for {seconds:.0f} s; aggregate over "
                           f"{span:.1f} s (wall incl. interpreter start-up {wall:.0f} s)"}, **cpu_note())


def cpu_baseline(model_kind, seconds):
    """The torch.nn port of the reference forward (oracle/torch_port.py) in ONE process on intra-op threads.

    ATen's small-GEMM RNN path collapses when oversubscribed (all hyper-threads of the GPU box: >20 s per batch),
    so a short sweep up to the physical core count picks the best thread count first; the reported `cores` is the
    thread count actually used for the timed sample.  Total CPU time is bounded.
    """
    physical, logical = host_cores()
    run, units, sample = _cpu_runner(model_kind)
    deadline = time.perf_counter() + 3.0 * seconds      # hard bound on the whole leg
    best_t, best_rate = None, 0.0
    with torch.no_grad():
        from pepper_amd.hostinfo import usable_cpus
        cap = max(1, min(physical, usable_cpus()))
        for nt in sorted({min(cap, c) for c in (8, 16, 32, 64, cap)}):
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
    return {"value": units * n / dt, "unit": "windows/s", "cores": best_t, "host_physical_cores": physical,
            "host_logical_cpus": logical, "kind": "port",
            "sample": f"{n} x ({sample}), {best_t} threads (best of a sweep up to {physical}), {dt:.1f} s"}


# HBM bytes one launch has to move per unit (window / chunk-window) if every operand is touched once: what `traffic` is
# compared with.  h2 layer outputs are 4 B per element like f32.  Weights (3-35 MB per launch, L2 / MALL resident) are
# not counted.
ALGORITHMIC_BYTES_PER_UNIT = {
    "lstm_rec_h2_fused_in": 33 * 26 + 33 * 512 * 4,            # int8 summary in, encoder output (h2) out
    "lstm_dec_h2_fused": 2 * 33 * 512 * 4,                     # encoder output in, decoder output out
    "gemm_h2_linear_1": 33 * 512 * 4 + 512 * 4,                # flattened decoder output in, [512] out
    "mlp_tail_h2": 512 * 4 + 12,
    "gru_rec_h2_fused_in": 100 * 10 + 100 * 256 * 4,           # per chunk and window launch
    "gru_dec_h2_fused": 2 * 100 * 256 * 4,
    "gru_dec_h2_fused_dense": 100 * 256 * 4 + 100 * 5 * 4,     # encoder output in, one direction's partial logits out (x 2 directions / 2)
    "dense_softmax_acc": 100 * 256 * 4 + 100 * 5 * 4 * 2,
}


def measured_traffic(model_kind, label):
    """HBM bytes per launch of `label` from the committed PMC passes (profiles/r05_<model>_pmc.json, else the latest earlier round's; written by
    tools/pmc_summary.py from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this file with --resident-only):
    FETCH_SIZE doubled as MI355X_MICROARCH.md (HBM section) prescribes for 16-byte-per-lane streaming reads, WRITE_SIZE as
    reported.  None where no pass is on file."""
    rel, rnd = newest_profile(f"{model_kind}_pmc.json")
    path = os.path.join(REPO, rel) if rel else ""
    try:
        with open(path) as fh:
            table = json.load(fh)
        k = table["kernels"][label]
        total = k["fetch_bytes_corrected"] + k["write_bytes"]
        return {"bytes_per_launch": total, "fetch_bytes_corrected": k["fetch_bytes_corrected"],
                "write_bytes": k["write_bytes"], "units_per_launch": k.get("units_per_launch"),
                # the counters north_star names, of the same profiled launches: MFMA-busy cycles / (GRBM_GUI_ACTIVE x 1024 SIMDs),
                # and the HBM bytes above over the launch's profiled duration
                "mfma_busy_frac": k.get("mfma_busy_frac"),
                "hbm_GBps_profiled": (total / (k["avg_us"] * 1e-6) / 1e9) if k.get("avg_us") else None,
                "avg_us_profiled": k.get("avg_us"),
                "source": os.path.relpath(path, REPO), "round": rnd,
                "stale": counters_stale(rnd, "rnn_h2.hip", "gemm_h2.hip", "mlp_h2.hip", "head.hip")}
    except Exception:
        return None


def realign_bench(args):
    """Secondary workload: one step = one call of the re-aligner over EIGHT polish regions (1 kb of draft + 20 safe bases each, 1500
    region-clipped reads per region -- the reference's cap -- 85 % of them spanning the window, nanopore-like error mix) through
    pa_realigner_align_windows, host buffers in, CIGARs out: 12 000 reads per call, what a caller that batches its regions hands
    over (the image chain hands over ~8 600 per call).  The one-region call (1 500 reads) and the 60-read region are reported
    beside it.  Not the headline metric."""
    import ctypes
    from oracle import ssw
    from pepper_amd.polish import PEPPER
    from pepper_amd.polish.PEPPER import ReadAligner
    rng = np.random.default_rng(5)
    n_windows = 8
    per_window = args.per_gpu or 1500
    windows, pos_all, seqs_all, which = [], [], [], []
    for w in range(n_windows):
        text = "".join("ACGT"[k] for k in rng.integers(0, 4, 1020))
        p, q = synthetic.simulate_clipped_reads(rng, text, 0, per_window, sub=0.04, ins=0.03, dele=0.04, min_len=200, full_span=0.85)
        windows.append((0, text))
        pos_all += list(p)
        seqs_all += list(q)
        which += [w] * len(q)
    reference, pos, seqs = windows[0][1], pos_all[:per_window], seqs_all[:per_window]

    def flat(seq_list):
        blob = [q.encode() for q in seq_list]
        off = np.zeros(len(seq_list) + 1, np.int64)
        np.cumsum([len(b) for b in blob], out=off[1:])
        return off, np.frombuffer(b"".join(blob), np.uint8)
    off_all, seq_all = flat(seqs_all)
    off, seq = flat(seqs)
    which = np.asarray(which, np.int32)
    pos_all = np.asarray(pos_all, np.int64)
    lib, h = PEPPER._realigner(0)
    for _ in range(args.warmup):
        PEPPER.align_windows(windows, which, pos_all, off_all, seq_all)
    ends = band = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        PEPPER.align_windows(windows, which, pos_all, off_all, seq_all)
        a, b, c = ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
        _lib.check(lib.pa_realigner_last_timing(h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
        ends += a.value
        band += b.value
    dt = time.perf_counter() - t0
    cells = c.value
    n = len(seqs_all)
    # one region per call: the reference's cap of 1 500 reads
    aligner = ReadAligner(0, len(reference), reference)
    aligner.align_arrays(pos, off, seq)
    t1r = time.perf_counter()
    for _ in range(args.steps):
        aligner.align_arrays(pos, off, seq)
    dt1 = time.perf_counter() - t1r
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
        PEPPER.align_windows(windows, which, pos_all, off_all, seq_all)
        barrier.wait()
        for _ in range(args.steps):
            PEPPER.align_windows(windows, which, pos_all, off_all, seq_all)
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
    # the roof is vector instruction issue: 1 024 SIMDs x 2.4 GHz / 4 cycles per wave instruction; the two kernels' instructions
    # per read come from the committed counter pass of this very workload (profiles/r05_realign_pmc.txt, SQ_INSTS_VALU)
    roof = {"bound": "valu issue", "kernel": "sw_ends_pair_kernel + band_kernel", "unit": "G wave-instructions/s", "peak": 1024 * 2.4 / 4,
            "achieved": None, "frac": None, "traffic": None}
    try:
        ins = {}
        src, src_round = newest_profile("realign_pmc.txt")
        for line in open(os.path.join(REPO, src)):
            parts = line.split()
            if len(parts) >= 2 and parts[0].startswith("valu_wave_instructions_per_read"):
                ins[parts[0]] = float(parts[1])
        per_read = ins["valu_wave_instructions_per_read_score"] + ins["valu_wave_instructions_per_read_band"]
        kernel_s = (ends + band) / args.steps * 1e-3
        wall_s = dt / args.steps
        roof.update(achieved=per_read * n / kernel_s / 1e9, frac=per_read * n * 4.0 / (1024 * 2.4e9 * kernel_s),
                    frac_over="kernel time (the two kernels' HIP events)",
                    frac_over_caller_wall=per_read * n * 4.0 / (1024 * 2.4e9 * wall_s),
                    valu_wave_instructions_per_read=per_read, source=src, stale=counters_stale(src_round, "realign.hip"),
                    note="integer DP on the vector ALUs, neither HBM nor MFMA bound: achieved = the two kernels' vector instructions (counter "
                         "pass of this workload) over their HIP-event time; frac = the share of the chip's issue slots they fill")
    except (OSError, KeyError, ValueError):
        pass
    print(json.dumps({
        "metric": "polish read re-alignment, reads/s (secondary workload)", "value": n * args.steps / dt, "unit": "reads/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int16 pairs (two reads per 32-bit lane register)", "data": "synthetic",
        "reads_per_s_4_worker_threads": 4 * n * args.steps / dt4, "reads_per_s_one_region_call": len(seqs) * args.steps / dt1,
        "ms_per_60_read_region": ms60,
        "config": {"workload": f"{n} region-clipped reads (mean {int(off_all[-1]) // n} bases) of {n_windows} draft windows of 1020 bases in one call, "
                               "SSW scoring 4/6/8/2, host buffers in, CIGARs out"},
        "kernels": {"score_kernels": {"avg_ms": ends / args.steps, "gcups_one_pass_equiv": cells / (ends / args.steps * 1e-3) / 1e9},
                    "band_kernel": {"avg_ms": band / args.steps}},
        "roofline": roof,
        "cpu_baseline": {"value": done / cpu_dt, "unit": "reads/s", "cores": 1, "kind": kind,
                         "sample": f"{done} of the same reads through {'the reference SSW build (oracle/_ref)' if kind == 'reference' else 'oracle/ssw_oracle.cpp'}, one thread, {cpu_dt:.1f} s"},
        "speedup_vs_cpu_baseline": (n * args.steps / dt) / (done / cpu_dt)}))


ENCODER_BYTES_PER_BASE = 2          # one base + one quality byte per aligned base
ENCODER_BYTES_PER_ROW = 104         # 26 int32 of the summary matrix per region position
HBM_PEAK_GBPS = 8000.0              # MI355X_MICROARCH.md


def encoder_traffic():
    """HBM bytes and instruction counts per launch of tile_count_kernel from the committed PMC passes (profiles/
    r04_encoder_variant_pmc.json, else r03)."""
    rel, rnd = newest_profile("encoder_variant_pmc.json")
    path = os.path.join(REPO, rel) if rel else ""
    try:
        with open(path) as fh:
            k = json.load(fh)["kernels"]["tile_count"]
        return {"bytes_per_launch": k["fetch_bytes_reported"] + k["write_bytes"], "fetch_bytes_reported": k["fetch_bytes_reported"],
                "write_bytes": k["write_bytes"], "source": os.path.relpath(path, REPO), "round": rnd,
                "stale": counters_stale(rnd, "encoder.hip", "encoder_common.h"),
                "insts": {c: k.get(c) for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_WAIT_ANY",
                                                "SQ_WAVE_CYCLES") if k.get(c) is not None},
                "avg_us_profiled": k.get("avg_us"),
                "note": "FETCH_SIZE as reported (byte-per-lane nt loads: on the same batch it reports 0.99 GB against 0.74 GB of "
                        "read bytes + 0.10 GB of CIGAR operations + 0.04 GB of records and tables, so no doubling applies to "
                        "this access pattern), WRITE_SIZE as reported"}
    except Exception:
        return None


def encoder_issue_roof(traffic, launch_ms):
    """The roof that binds tile_count_kernel is instruction issue, not HBM (the launch moves 1.1x its algorithmic bytes at a tenth
    of the HBM rate): vector wave-instructions of one launch (rocprofv3 SQ_INSTS_VALU, committed PMC pass) x 4 cycles each over
    the SIMD-cycles the launch had (1024 SIMDs x launch duration x 2.4 GHz); the scalar and LDS instructions of the same launch
    are listed beside it (they issue on their own ports)."""
    try:
        ins = traffic["insts"]
        valu = float(ins["SQ_INSTS_VALU"])
        simd_cycles = 1024 * launch_ms * 1e-3 * 2.4e9
        return {"bound": "valu issue", "valu_wave_instructions": valu, "salu_wave_instructions": ins.get("SQ_INSTS_SALU"),
                "lds_wave_instructions": ins.get("SQ_INSTS_LDS"), "cycles_per_valu_instruction": 4, "simd_cycles_available": simd_cycles,
                "frac": 4.0 * valu / simd_cycles,
                "wait_share_of_wave_cycles": (ins["SQ_WAIT_ANY"] / ins["SQ_WAVE_CYCLES"]) if ins.get("SQ_WAVE_CYCLES") else None,
                "frac_over": "kernel time (tile_count_kernel's HIP events)", "source": traffic["source"], "stale": traffic.get("stale")}
    except Exception:
        return None


def encoder_cpu_all_cores(seconds, region_size):
    """The reference's image generation is one single-thread worker per core, a region at a time
    (ImageGenerationUI.py:262-274): that many workers of oracle/encoder_cpu.py, concurrently, each on its own region."""
    import subprocess
    physical, logical = host_cores()
    procs = worker_count()
    try:
        import psutil
        procs = max(1, min(procs, int(psutil.virtual_memory().available / (0.4 * 2 ** 30))))
    except Exception:
        pass
    cmd = [sys.executable, os.path.join(REPO, "oracle", "encoder_cpu.py"), "--seconds", str(seconds), "--region-size", str(region_size)]
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="", OMP_NUM_THREADS="1")
    t0 = time.perf_counter()
    ps = [subprocess.Popen(cmd + ["--seed", str(k)], env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for k in range(procs)]
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
    span = max(o["seconds"] for o in outs)
    return dict({"value": sum(o["bases"] for o in outs) / span, "unit": "aligned bases/s", "cores": len(outs),
                 "host_physical_cores": physical, "host_logical_cpus": logical, "kind": outs[0]["kind"],
                 "sample": f"{len(outs)} concurrent single-thread workers, one per CPU this container may use, each looping the reference's "
                           f"RegionalSummaryGenerator (oracle/_ref) on its own E-syn region for {seconds:.0f} s; "
                           f"{sum(o['regions'] for o in outs)} regions in {span:.1f} s (wall incl. start-up {wall:.0f} s)"}, **cpu_note())


def packed_arena_of(regions):
    """The E-syn regions in the packed form pa_bam_pack_regions produces (include/pepper_amd_io.h): per read its CIGAR words, its
    bases as 4-bit codes and its qualities, once, 4-byte aligned, in one arena + the read table + the per-region read lists.
    (Synthetic stand-in for the BAM reader: every read belongs to one region.)"""
    from pepper_amd.variant.bam import PACKED_READ
    code = np.zeros(256, np.uint8)
    for k, c in enumerate(b"=ACMGRSVTWYHKDBN"):
        code[c] = k
    n_reads = sum(int(flat["n_reads"]) for _, flat, _, _ in regions)
    table = np.zeros(n_reads, PACKED_READ)
    pair_read = np.arange(n_reads, dtype=np.int32)
    region_pairs = np.zeros(len(regions) + 1, np.int32)
    chunks, at, k = [], 0, 0
    for r, (_, flat, rs, _) in enumerate(regions):
        so, co = flat["seq_offset"], flat["cigar_offset"]
        words = (flat["cigar_len"][:co[-1]].astype(np.uint32) << 4) | flat["cigar_op"][:co[-1]].astype(np.uint32)
        codes = code[flat["seq"][:so[-1]]]
        for i in range(int(flat["n_reads"])):
            nc, ls = int(co[i + 1] - co[i]), int(so[i + 1] - so[i])
            c4 = codes[so[i]:so[i + 1]]
            if ls & 1:
                c4 = np.concatenate([c4, np.zeros(1, np.uint8)])
            blob = words[co[i]:co[i + 1]].tobytes() + ((c4[0::2] << 4) | c4[1::2]).tobytes() + flat["qual"][so[i]:so[i + 1]].tobytes()
            pad = (-len(blob)) & 3
            table[k] = (at, int(flat["read_pos"][i]), nc, ls, (16 if flat["read_reverse"][i] else 0) | (int(flat["read_mapq"][i]) << 16))
            chunks.append(blob + b"\0" * pad)
            at += len(blob) + pad
            k += 1
        region_pairs[r + 1] = k
    return np.frombuffer(b"".join(chunks) + b"\0" * 64, np.uint8), table, pair_read, region_pairs, at


def encoder_bench(args):
    """`--model encoder`: one step = one pass of pa_encoder_run_staged over a batch of E-syn regions resident in HBM (record
    kernels, tile_count_kernel, vote compaction, candidate enumeration on the host, window gather)."""
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_spawn(args)
    world, rank, device, ranks_seen = dist_setup(args)
    from pepper_amd.variant.PEPPER_VARIANT import RegionalSummaryGenerator, StagedBatch
    n_regions = args.per_gpu or 64
    region_size = 100_000
    t0 = time.perf_counter()
    regions = synthetic.encoder_regions(n_regions, seed=synthetic.ESYN_SEED + 1000 * rank, region=region_size)
    t_gen = time.perf_counter() - t0
    gens = [RegionalSummaryGenerator("chr20", rs, re_, ref, device=device) for ref, _, rs, re_ in regions]
    flats = [flat for _, flat, _, _ in regions]
    cand = [(rs + 100, re_ - 100) for _, _, rs, re_ in regions]
    ont = (1, 1, 0.10, 0.15, 0.15, 3, 0.10, 0.12, 2, False)          # SetParameters.py:20-36 ont_r9_guppy5_sup
    t0 = time.perf_counter()
    batch = StagedBatch(gens, flats, ont, cand)                       # upload: outside the timed region
    t_stage = time.perf_counter() - t0
    for _ in range(max(1, args.warmup)):
        counts = batch.run()
    stats = batch.stats()

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
    torch.cuda.synchronize(device)
    barrier()
    times = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        batch.run()                                                   # returns after its own stream synchronise
        times.append(batch.timing())
    torch.cuda.synchronize(device)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        on_gpu = dist.get_backend() == "nccl"
        tt = torch.tensor([dt], dtype=torch.float64, device=torch.device("cuda", device) if on_gpu else None)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    if rank != 0:
        if world > 1:
            leave_group()
        return
    avg = {k: float(np.mean([t[k] for t in times])) for k in times[0]}
    value = world * args.steps * stats["bases"] / dt
    alg_bytes = ENCODER_BYTES_PER_BASE * stats["bases"] + ENCODER_BYTES_PER_ROW * stats["rows"]
    achieved = alg_bytes / (avg["tile_count_ms"] * 1e-3) / 1e9
    traffic = encoder_traffic()
    # PCIe-inclusive form (never `value`): stage + run of the same batch through one call
    t0 = time.perf_counter()
    b2 = StagedBatch(gens, flats, ont, cand)
    b2.run()
    t_one = time.perf_counter() - t0
    out = b2.results()
    # several handles on as many host threads, as image generation runs its workers (one encoder per thread): the candidate
    # enumeration of one handle's batch overlaps the kernels of the others'.  The batch is dealt over the handles.
    two = []
    for n_handles in ((2, 4) if (n_regions >= 16 and world == 1) else ()):
        import threading
        share = n_regions // n_handles
        done_bases = [0] * n_handles
        start_gate = threading.Barrier(n_handles + 1)

        def worker(k):
            sl = slice(k * share, (k + 1) * share)
            mine = StagedBatch(gens[sl], flats[sl], ont, cand[sl])        # its own handle: _encoder is per thread
            mine.run()
            start_gate.wait()
            for _ in range(2 * args.steps):
                mine.run()
            done_bases[k] = 2 * args.steps * mine.stats()["bases"]
            start_gate.wait()
        threads = [threading.Thread(target=worker, args=(k,)) for k in range(n_handles)]
        for th in threads:
            th.start()
        start_gate.wait()
        t0 = time.perf_counter()
        start_gate.wait()
        t_two = time.perf_counter() - t0
        for th in threads:
            th.join()
        two.append({"value": sum(done_bases) / t_two, "unit": "aligned bases/s", "handles": n_handles, "regions_per_handle": share})
    # the packed host-fed form (what image generation drives): reads as BAM stores them in the handle's page-locked arena ->
    # one H2D of the arena + one of the tables -> unpack_clip_kernel -> the step above.  PCIe-inclusive; never `value`.
    packed = None
    try:
        from pepper_amd.variant.PEPPER_VARIANT import PackedEncoder
        arena, table, pair_read, region_pairs, used = packed_arena_of(regions)
        enc = PackedEncoder(device, arena_bytes=int(used) + (1 << 20), max_reads=len(table) + 16, max_pairs=len(pair_read) + 16,
                            host_threads=0)
        enc.arena[:len(arena)] = arena
        enc.reads[:len(table)] = table
        enc.pair_read[:len(pair_read)] = pair_read
        spans = [(rs, re_) for _, _, rs, re_ in regions]
        refs = [ref for ref, _, _, _ in regions]
        counts3 = (len(table), len(pair_read), int(used))
        outs, live = enc.encode(spans, refs, region_pairs, counts3, ont, cand)
        same = all(np.array_equal(a["images"], b["images"]) and a["candidates"] == b["candidates"] for a, b in zip(outs, out))
        reps = max(3, args.steps // 2)
        t0 = time.perf_counter()
        for _ in range(reps):
            enc.encode(spans, refs, region_pairs, counts3, ont, cand)
        t_packed = (time.perf_counter() - t0) / reps
        tm = enc.last.timing()
        shipped = int(used) + sum(len(r) for r in refs)
        packed = {"value": stats["bases"] / t_packed, "unit": "aligned bases/s", "ms": t_packed * 1e3,
                  "bytes_over_pcie": shipped, "bytes_per_aligned_base": shipped / stats["bases"],
                  "upload_ms": tm["upload_ms"], "unpack_clip_ms": tm["unpack_clip_ms"],
                  "pcie_roof": {"peak_GBps": 63.0, "bases_per_s_at_peak": 63.0e9 / (shipped / stats["bases"]),
                                "achieved_GBps_during_upload": shipped / (tm["upload_ms"] * 1e-3) / 1e9 if tm["upload_ms"] else None,
                                "note": "PCIe Gen5 x16 (MI355X_MICROARCH.md: 63 GB/s): the packed form ships ~1.65 B per aligned base "
                                        "(4-bit bases, qualities, CIGAR words), the host-clipped form 2 B + 8 B per operation"},
                  "identical_to_resident_form": bool(same),
                  "note": "pa_encoder_stage_packed + pa_encoder_run_staged + result copy per call, arena and tables page-locked"}
        enc.close()
    except Exception as e:      # noqa: BLE001 -- the leg is extra
        packed = {"error": repr(e)[:300]}
    line = {
        "metric": "variant summary encoder, aligned bases/s (pileup -> candidate summary images)",
        "value": value, "unit": "aligned bases/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "int32 counts (uint8 bases / qualities in, int32 summary matrix, int8 images out)", "data": "synthetic",
        "config": {"workload": f"E-syn: {n_regions} regions of {region_size} + 2 x 100 positions at ~60x, ~8 kb reads, an insert or "
                               "a deletion every ~50 bases, 4 % substitutions, planted SNP and indel sites, ONT R9 guppy5 thresholds "
                               "(pepper_amd.synthetic.encoder_region); one step = pa_encoder_run_staged on the batch, inputs "
                               "resident in HBM, results (candidates, positions, int8 windows) left on the device / host",
                   "regions_per_gpu_per_step": n_regions, "aligned_bases_per_step": stats["bases"], "matrix_rows_per_step": stats["rows"],
                   "reads": stats["reads"], "cigar_operations": stats["cigar_ops"], "tiles": stats["tiles"],
                   "candidates_per_step": int(np.sum(counts)), "windows_per_s": world * args.steps * int(np.sum(counts)) / dt,
                   "timed_seconds": dt, "generate_seconds": t_gen, "stage_seconds": t_stage,
                   "parallelism": f"region-shard x{world}, no collective", "ranks_seen": ranks_seen},
        "roofline": {"bound": "hbm", "kernel": "tile_count_kernel", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic["bytes_per_launch"] if traffic else None,
                     "traffic_detail": traffic, "algorithmic_bytes_per_launch": alg_bytes,
                     "avg_launch_ms": avg["tile_count_ms"],
                     "issue": encoder_issue_roof(traffic, avg["tile_count_ms"]),
                     "note": "algorithmic bytes = 2 B per aligned base (base + quality) + 104 B per region position (26 int32), "
                             "SURVEY.md 8(d); launch duration from HIP events on the encoder's stream around the kernel, averaged "
                             "over the timed steps"},
        "kernels_ms": avg,
        "concurrent_handles": two,
        "concurrent_handles_note": "N encoder handles on N host threads (image generation's worker scheme), each running its share of "
                                   "the regions back to back: one handle's host enumeration beside the others' kernels",
        "host_buffers_one_call": {"value": stats["bases"] / t_one, "unit": "aligned bases/s", "ms": t_one * 1e3,
                                  "note": "pa_encoder_generate_summary_batch from pageable numpy arrays: validate + H2D of 0.75 GB + the "
                                          "step above (PCIe-inclusive; never `value`)"},
        "packed_host_fed": packed,
    }
    if world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(REPO, "oracle"))
        from oracle import encoder_cpu
        one = encoder_cpu.time_regions(regions[:8], args.cpu_seconds)
        single = {"value": one["bases"] / one["seconds"], "unit": "aligned bases/s", "cores": 1, "kind": one["kind"],
                  "sample": f"{one['regions']} of the same regions through the reference's RegionalSummaryGenerator "
                            f"({'oracle/_ref build of region_summary.cpp' if one['kind'] == 'reference' else 'oracle restatement'}), "
                            f"one thread, {one['seconds']:.1f} s"}
        multi = encoder_cpu_all_cores(args.cpu_seconds, region_size)
        line["cpu_baseline"] = multi or single
        line["cpu_baseline_one_core"] = single
        line["speedup_vs_cpu_baseline"] = value / line["cpu_baseline"]["value"]
        # parity spot check of the timed configuration (the tests hold the rest): region 0 against the CPU encoder
        import ctypes
        kind, run = encoder_cpu.load()
        line["candidates_region0"] = {"device": len(out[0]["candidates"]), "cpu": run(*encoder_cpu.region_structs(regions[0]))}
    print(json.dumps(line))
    if world > 1:
        leave_group()


def polish_encoder_cpu_all_cores(seconds):
    """One single-thread worker per usable CPU, each looping the reference's SummaryGenerator on its own region (the reference's
    image generation is one such worker per core: pepper/modules/python/ImageGenerationUI.py)."""
    import subprocess
    physical, logical = host_cores()
    procs = worker_count()
    cmd = [sys.executable, os.path.join(REPO, "oracle", "encoder_cpu.py"), "--polish", "--seconds", str(seconds), "--region-size", "1000"]
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="", OMP_NUM_THREADS="1")
    ps = [subprocess.Popen(cmd + ["--seed", str(k)], env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for k in range(procs)]
    outs = []
    for p in ps:
        try:
            out, _ = p.communicate(timeout=6 * seconds + 120)
            outs.append(json.loads(out.strip().splitlines()[-1]))
        except Exception:
            p.kill()
    if not outs:
        return None
    span = max(o["seconds"] for o in outs)
    return dict({"value": sum(o["bases"] for o in outs) / span, "unit": "aligned bases/s", "cores": len(outs),
                 "host_physical_cores": physical, "host_logical_cpus": logical, "kind": outs[0]["kind"],
                 "sample": f"{len(outs)} concurrent single-thread workers, one per CPU this container may use, each looping the reference's "
                           f"polish SummaryGenerator (oracle/_ref) on its own region for {seconds:.0f} s; "
                           f"{sum(o['regions'] for o in outs)} regions in {span:.1f} s"}, **cpu_note())


POLISH_ENCODER_BYTES_PER_BASE = 1     # the polish walk reads the bases only (no quality test: summary_generator.cpp:47-121)
POLISH_ENCODER_BYTES_PER_ROW = 26     # 10 pixel bytes + the (position, insert index) pair of int64 per output row


def polish_encoder_bench(args):
    """`--model polish-encoder`: one step = pa_polish_encoder_run_staged over a batch of regions resident in HBM (record kernels,
    longest-insert scan, polish_tile_kernel, insert rows); the host-buffer form (stage + run + result copy) beside it."""
    from pepper_amd.polish.PEPPER import StagedSummaries, SummaryGenerator
    n_regions = args.per_gpu or 256
    # (read_len / depth chosen so that the reads, clipped to the 1.2 kb window, cover it ~60x)
    regions = synthetic.encoder_regions(n_regions, seed=synthetic.ESYN_SEED + 5000, region=1000, read_len=2000, depth=140)
    gens = [SummaryGenerator(ref, "contig_1", rs, re_) for ref, _, rs, re_ in regions]
    flats = [flat for _, flat, _, _ in regions]
    spans = [(rs, re_) for _, _, rs, re_ in regions]
    batch = StagedSummaries(gens, flats, spans)
    for _ in range(max(1, args.warmup)):
        rows = batch.run()
    stats = batch.stats()
    torch.cuda.synchronize(0)
    times = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        batch.run()
        times.append(batch.timing())
    torch.cuda.synchronize(0)
    dt = time.perf_counter() - t0
    avg = {k: float(np.mean([t[k] for t in times])) for k in times[0]}
    value = args.steps * stats["bases"] / dt
    alg_bytes = POLISH_ENCODER_BYTES_PER_BASE * stats["bases"] + POLISH_ENCODER_BYTES_PER_ROW * stats["rows"]
    achieved = alg_bytes / (avg["tile_ms"] * 1e-3) / 1e9
    # host buffers in, host arrays out, one call (what AlignmentSummarizer.create_summaries makes per batch of regions)
    reps = max(3, args.steps // 2)
    t0 = time.perf_counter()
    for _ in range(reps):
        b2 = StagedSummaries(gens, flats, spans)
        b2.run()
        image, pos = b2.results()
    t_one = (time.perf_counter() - t0) / reps
    line = {
        "metric": "polish summary encoder, aligned bases/s (pileup -> summary rows)",
        "value": value, "unit": "aligned bases/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "int32 counts (uint8 bases in, uint8 pixels + int64 (position, index) rows out)", "data": "synthetic",
        "config": {"workload": f"P-enc-syn: {n_regions} regions of 1000 + 2 x 100 positions at ~60x (reads clipped to the region as the BAM "
                               "reader clips them), an insert or a deletion every ~50 bases, 4 % substitutions "
                               "(pepper_amd.synthetic.encoder_region(region=1000)); one step = pa_polish_encoder_run_staged, pileups "
                               "resident in HBM, rows left on the device",
                   "regions_per_step": n_regions, "aligned_bases_per_step": stats["bases"], "rows_per_step": stats["rows"],
                   "reads": stats["reads"], "cigar_operations": stats["cigar_ops"], "tiles": stats["tiles"],
                   "regions_per_s": args.steps * n_regions / dt, "timed_seconds": dt},
        "roofline": {"bound": "hbm", "kernel": "polish_tile_kernel", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBPS, "traffic": None, "algorithmic_bytes_per_launch": alg_bytes,
                     "avg_launch_ms": avg["tile_ms"],
                     "note": "algorithmic bytes = 1 B per aligned base (the polish walk tests no quality) + 26 B per output row (10 "
                             "pixels + the (position, index) int64 pair); 512-position tiles of 1.2 kb regions: 768 workgroups for 256 "
                             "regions, the launch is latency- not bandwidth-bound at this size"},
        "kernels_ms": avg,
        "host_buffers_one_call": {"value": stats["bases"] / t_one, "unit": "aligned bases/s", "ms": t_one * 1e3,
                                  "regions_per_s": n_regions / t_one,
                                  "note": "pa_polish_encoder_stage_batch + run + result copy per batch: gather into page-locked blocks, one "
                                          "H2D per array, kernels, one wait, D2H of the rows"},
    }
    # The roof that binds polish_tile_kernel: 768 workgroups are ONE round of wavefronts (6 per SIMD), so the launch is as long as its
    # slowest wavefront's dependent chain -- the issue share and the share of their cycles the wavefronts spend waiting say so
    # (newest committed counter pass of this command, profiles/rNN_encoder_polish_pmc.json)
    try:
        src, src_round = newest_profile("encoder_polish_pmc.json")
        with open(os.path.join(REPO, src)) as fh:
            k = json.load(fh)["kernels"]["polish_tile"]
        simd_cycles = 1024 * avg["tile_ms"] * 1e-3 * 2.4e9
        line["roofline"]["issue"] = {
            "bound": "latency of one round of wavefronts (valu issue beside it)", "valu_wave_instructions": k["SQ_INSTS_VALU"],
            "salu_wave_instructions": k.get("SQ_INSTS_SALU"), "lds_wave_instructions": k.get("SQ_INSTS_LDS"), "waves": k.get("SQ_WAVES"),
            "frac": 4.0 * k["SQ_INSTS_VALU"] / simd_cycles, "frac_over": "kernel time (polish_tile_kernel's HIP events)",
            "wait_share_of_wave_cycles": (k["SQ_WAIT_ANY"] / k["SQ_WAVE_CYCLES"]) if k.get("SQ_WAVE_CYCLES") else None,
            "source": src, "stale": counters_stale(src_round, "encoder_polish.hip", "encoder_common.h")}
    except Exception:       # noqa: BLE001
        pass
    if not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(REPO, "oracle"))
        from oracle import encoder_cpu
        one = encoder_cpu.time_polish_regions(regions[:16], args.cpu_seconds)
        single = {"value": one["bases"] / one["seconds"], "unit": "aligned bases/s", "cores": 1, "kind": one["kind"],
                  "sample": f"{one['regions']} of the same regions through the reference's SummaryGenerator "
                            f"({'oracle/_ref build of summary_generator.cpp' if one['kind'] == 'reference' else 'oracle restatement'}), "
                            f"one thread, {one['seconds']:.1f} s"}
        multi = polish_encoder_cpu_all_cores(args.cpu_seconds)
        line["cpu_baseline"] = multi or single
        line["cpu_baseline_one_core"] = single
        line["speedup_vs_cpu_baseline"] = value / line["cpu_baseline"]["value"]
        # parity spot check of the timed configuration: region 0's rows against the CPU encoder
        kind, run = encoder_cpu.load_polish()
        p0 = encoder_cpu.region_structs(regions[0])[0]
        img = np.zeros((int(rows[0]) + 16, 10), np.uint8)
        posn = np.zeros((int(rows[0]) + 16, 2), np.int64)
        n0 = run(p0, regions[0][2], regions[0][3], img.ctypes.data, posn.ctypes.data, len(img))
        line["rows_region0"] = {"device": int(rows[0]), "cpu": n0,
                                "identical": bool(n0 == int(rows[0]) and np.array_equal(img[:n0], image[:n0]) and np.array_equal(posn[:n0], pos[:n0]))}
    print(json.dumps(line))


def make_images_leg(scratch, level=1, tags=0, bases_default=64_000_000, quals=0):
    """generate_images (pepper_variant make_images / call_variant's first step) on a synthetic 64 Mb BAM at 60x written by
    tools/synth_bam: BAM + FASTA -> candidate image HDF5 files, Mb of reference per second with the stage times of the workers
    (tools/bench_variant_images.py).  Three runs over the same files, the median reported."""
    import shutil
    import subprocess
    import tempfile
    base = scratch or tempfile.gettempdir()
    try:
        st = os.statvfs(base)
        bases = bases_default if st.f_bavail * st.f_frsize > (12 << 30) else min(bases_default, 16_000_000)
    except OSError:
        bases = min(bases_default, 16_000_000)
    work = tempfile.mkdtemp(prefix="pepper_amd_images_", dir=base)
    tool = os.path.join(REPO, "tools", "bench_variant_images.py")
    try:
        from pepper_amd.hostinfo import usable_cpus
        threads = max(1, usable_cpus())
        p = subprocess.run([sys.executable, tool, "make_fast", work, str(bases), "60", "2027", str(level), str(tags), str(quals)], capture_output=True,
                           text=True, timeout=600)
        if p.returncode != 0:
            return {"error": (p.stderr or "synth_bam failed").strip().splitlines()[-1][:300]}
        made = json.loads(p.stdout.strip().splitlines()[-1])
        p = subprocess.run([sys.executable, tool, "run", work, ",".join([str(threads)] * 3)], capture_output=True, text=True, timeout=900)
        lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        if not lines:
            return {"error": (p.stderr or "no output").strip().splitlines()[-1][:300]}
        d = json.loads(lines[-1])
        runs = sorted(d["runs"], key=lambda r: r["mb_reference_per_s"])
        mid = runs[len(runs) // 2]
        return {"value": mid["mb_reference_per_s"], "unit": "Mb of reference/s", "aligned_gbases_per_s": mid["aligned_gbases_per_s"],
                "seconds": mid["seconds"], "threads": mid["threads"], "runs_mb_per_s": [r["mb_reference_per_s"] for r in runs],
                "stage_seconds_summed_over_workers": mid["stage_seconds_summed_over_workers"], "data": d["data"],
                "synth_seconds": made["seconds"], "image_file_mb": mid["image_file_mb"],
                "note": "pepper_amd.variant.ImageGenerationUI.generate_images, intervals of 100 kb, one worker thread per usable CPU, each "
                        "with its own BAM handle, page-locked arena and encoder: bam_span_read = the file span of a group of intervals "
                        "(pread), bam_inflate_device = upload + bgzf_inflate_kernel (one wavefront per BGZF member), bam_walk_device = the "
                        "record headers read out on the device (40 bytes per record come back; the records stay in place there), "
                        "bam_walk = filters, region test and pair lists on the host over those headers, encode = unpack_clip_kernel + "
                        "the summary kernels + candidate enumeration + result copy, hdf5 = the append-only writer; inflate_kernel = "
                        "the kernel's event time summed over the workers' streams"}
    except Exception as e:      # noqa: BLE001
        return {"error": repr(e)[:300]}
    finally:
        shutil.rmtree(work, ignore_errors=True)


def polish_make_images_leg(scratch):
    """make_images of pepper polish (BAM + draft -> image HDF5 files) on a synthetic 16 Mb draft at 60x written by tools/synth_bam,
    through the device-resident chain (tools/bench_polish_chain.py): Mb of draft per second with the workers' stage times and the
    roofline of the stage that bounds it -- the re-aligner's vector instruction issue.  Three runs over the same files, the median."""
    import shutil
    import subprocess
    import tempfile
    base = scratch or tempfile.gettempdir()
    work = tempfile.mkdtemp(prefix="pepper_amd_pimages_", dir=base)
    tool = os.path.join(REPO, "tools", "bench_polish_chain.py")
    try:
        from pepper_amd.hostinfo import usable_cpus
        threads = max(1, usable_cpus())
        p = subprocess.run([sys.executable, tool, "make_fast", work, "16000000"], capture_output=True, text=True, timeout=600)
        if p.returncode != 0:
            return {"error": (p.stderr or "synth_bam failed").strip().splitlines()[-1][:300]}
        made = json.loads(p.stdout.strip().splitlines()[-1])
        p = subprocess.run([sys.executable, tool, "run", work, ",".join([str(threads)] * 3)], capture_output=True, text=True, timeout=900)
        lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        if not lines:
            return {"error": (p.stderr or "no output").strip().splitlines()[-1][:300]}
        d = json.loads(lines[-1])
        runs = sorted(d["runs"], key=lambda r: r["mb_draft_per_s"])
        mid = runs[len(runs) // 2]
        stages = mid["stage_seconds_summed_over_workers"]
        out = {"value": mid["mb_draft_per_s"], "unit": "Mb of draft/s", "intervals_per_s": mid["intervals_per_s"],
               "reads_realigned_per_s": mid["reads_realigned_per_s"], "seconds": mid["seconds"], "threads": mid["threads"],
               "regions_per_call": d["regions_per_call"], "runs_mb_per_s": [r["mb_draft_per_s"] for r in runs], "counts": mid["counts"],
               "stage_seconds_summed_over_workers": stages, "data": d["data"], "synth_seconds": made["seconds"],
               "image_file_mb": mid["image_file_mb"]}
        # the bound: the two re-aligner kernels' vector instructions (profiles/r05_polish_chain_pmc.txt: wave instructions per read)
        # x 4 cycles against the SIMD-cycles of the wall time
        try:
            ins = {}
            src, src_round = newest_profile("polish_chain_pmc.txt")
            for line in open(os.path.join(REPO, src)):
                parts = line.split()
                if len(parts) >= 2 and parts[0] in ("valu_wave_instructions_per_read_score", "valu_wave_instructions_per_read_band"):
                    ins[parts[0]] = float(parts[1])
            per_read = ins["valu_wave_instructions_per_read_score"] + ins["valu_wave_instructions_per_read_band"]
            simd_cycles = 1024 * mid["seconds"] * 2.4e9
            out["roofline"] = {"bound": "valu issue", "kernel": "sw_ends_pair_kernel + band_kernel", "unit": "G wave-instructions/s",
                               "achieved": per_read * mid["counts"]["realigned"] / mid["seconds"] / 1e9, "peak": 1024 * 2.4 / 4,
                               "frac": 4.0 * per_read * mid["counts"]["realigned"] / simd_cycles,
                               "frac_over": "the job's wall time (all workers; what the chain could do at most is 1.0)",
                               "valu_wave_instructions_per_read": per_read, "source": src,
                               "stale": counters_stale(src_round, "realign.hip", "encoder_polish.hip")}
        except (OSError, KeyError, ValueError):
            pass
        return out
    except Exception as e:      # noqa: BLE001
        return {"error": repr(e)[:300]}
    finally:
        shutil.rmtree(work, ignore_errors=True)


def e2e_leg(kind, scratch, bases, coverage, runs, need_gb):
    """tools/bench_e2e.py: the whole job through the reference's top entry point (call_variant / polish) on synthetic inputs, the
    median of `runs` with the three steps' walls."""
    import shutil
    import subprocess
    import tempfile
    base = scratch or tempfile.gettempdir()
    try:
        st = os.statvfs(base)
        if st.f_bavail * st.f_frsize < (need_gb << 30):          # a quarter of the job where the scratch space is short
            bases //= 4
    except OSError:
        bases //= 4
    work = tempfile.mkdtemp(prefix="pepper_amd_e2e_", dir=base)
    try:
        p = subprocess.run([sys.executable, os.path.join(REPO, "tools", "bench_e2e.py"), kind, work, str(bases), str(coverage), str(runs)],
                           capture_output=True, text=True, timeout=1200)
        lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        if not lines:
            return {"error": (p.stderr or "no output").strip().splitlines()[-1][:300]}
        return json.loads(lines[-1])
    except Exception as e:      # noqa: BLE001
        return {"error": repr(e)[:300]}
    finally:
        shutil.rmtree(work, ignore_errors=True)


def secondary_block(args):
    """The other workloads of the hot path, each as its own short run of this file / the pipeline tools after the headline
    measurement (same command, same box, one after the other on the one GPU): polish (BASELINE configs[4]) windows/s with its
    dominant kernel's roofline, the summary encoder's aligned bases/s, and the two HDF5 -> HDF5 rates through the reference's
    entry points (run_inference on >= 4 M windows, call_consensus on >= 128 k chunks) with the files in tmpfs."""
    import subprocess
    me = os.path.abspath(__file__)
    out = {}
    # --legs a,b: only those legs (development runs: one leg in the context the driver's run gives it -- a child of this process)
    only = set(filter(None, (getattr(args, "legs", "") or "").split(",")))
    SKIP = {"error": "skipped (--legs)"}

    def want(name):
        return not only or name in only

    def last_json(cmd, timeout, env=None):
        t0 = time.perf_counter()
        try:
            p = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
            lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
            if not lines:
                return {"error": (p.stderr or "no output").strip().splitlines()[-1][:300] if (p.stderr or "").strip() else "no output"}
            d = json.loads(lines[-1])
            d["_seconds"] = round(time.perf_counter() - t0, 1)
            return d
        except Exception as e:          # a failing secondary leg must not take the headline line with it
            return {"error": repr(e)[:300]}
    def median_of(cmd, timeout, runs=3, key="value"):
        """The run with the median `key` of `runs` fresh processes (each a few seconds: start-up and box noise decide the second
        digit of any single one), with every run's figure beside it."""
        got = [last_json(cmd, timeout) for _ in range(runs)]
        good = sorted((g for g in got if "error" not in g), key=lambda g: g[key])
        if not good:
            return got[0]
        mid = good[len(good) // 2]
        mid["_runs"] = [round(g[key], 1) for g in good]
        return mid
    d = median_of([sys.executable, me, "--model", "polish", "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--full-line"], 300) if want("polish") else SKIP
    out["polish"] = d if "error" in d else {
        "value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "workload": d["config"]["workload"],
        "h2d_d2h": d["config"]["h2d_d2h"], "roofline": {k: d["roofline"][k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac",
                                                                                      "frac_algorithmic_of_dtype_peak", "mfma_busy_frac",
                                                                                      "hbm_GBps") if k in d["roofline"]},
        "batch128": d.get("batch128"), "device_resident": (d.get("device_resident") or {}).get("value"),
        "runs": d["_runs"], "seconds": d["_seconds"]}
    d = median_of([sys.executable, me, "--model", "encoder", "--steps", "10", "--warmup", "2", "--no-cpu-baseline"], 300) if want("encoder") else SKIP
    out["encoder"] = d if "error" in d else {
        "value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "regions_per_step": d["config"]["regions_per_gpu_per_step"],
        "aligned_bases_per_step": d["config"]["aligned_bases_per_step"], "candidates_per_step": d["config"]["candidates_per_step"],
        "roofline": {k: d["roofline"][k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch",
                                                   "issue") if k in d["roofline"]},
        "host_buffers_one_call": d["host_buffers_one_call"]["value"], "packed_host_fed": d.get("packed_host_fed"),
        "runs": d["_runs"], "seconds": d["_seconds"]}
    d = median_of([sys.executable, me, "--model", "polish-encoder", "--steps", "10", "--warmup", "2", "--no-cpu-baseline"], 300) if want("polish_encoder") else SKIP
    out["polish_encoder"] = d if "error" in d else {
        "value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "regions_per_step": d["config"]["regions_per_step"],
        "aligned_bases_per_step": d["config"]["aligned_bases_per_step"], "rows_per_step": d["config"]["rows_per_step"],
        "roofline": {k: d["roofline"][k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "algorithmic_bytes_per_launch", "issue")
                     if k in d["roofline"]},
        "host_buffers_one_call": d["host_buffers_one_call"]["value"], "runs": d["_runs"], "seconds": d["_seconds"]}
    d = median_of([sys.executable, me, "--model", "realign", "--steps", "10", "--warmup", "2", "--cpu-seconds", "3"], 300) if want("realign") else SKIP
    out["realign"] = d if "error" in d else {
        "value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "workload": d["config"]["workload"],
        "reads_per_s_4_worker_threads": d.get("reads_per_s_4_worker_threads"), "reads_per_s_one_region_call": d.get("reads_per_s_one_region_call"),
        "ms_per_60_read_region": d.get("ms_per_60_read_region"),
        "roofline": {k: d["roofline"].get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "valu_wave_instructions_per_read", "source")},
        "kernels": d.get("kernels"), "cpu_baseline": d.get("cpu_baseline"), "runs": d["_runs"], "seconds": d["_seconds"]}
    scratch = None
    try:
        st = os.statvfs("/dev/shm")
        if st.f_bavail * st.f_frsize > 24 << 30:
            scratch = "/dev/shm"
    except OSError:
        pass
    extra = ["--dir", scratch] if scratch else []
    out["make_images"] = make_images_leg(scratch) if want("make_images") else SKIP
    # the same job on a BAM as samtools writes it: zlib level 6 members, NM / MD / RG aux data in every record (32 Mb: zlib level 6
    # writes the synthetic file at a tenth of libdeflate level 1's rate)
    lv6 = make_images_leg(scratch, level=6, tags=1, bases_default=32_000_000) if want("make_images_level6") else SKIP
    out["make_images_level6"] = lv6 if "error" in lv6 else {k: lv6[k] for k in ("value", "unit", "seconds", "threads", "runs_mb_per_s", "data",
                                                                                  "stage_seconds_summed_over_workers", "synth_seconds")}
    lvr = make_images_leg(scratch, level=6, tags=1, bases_default=32_000_000, quals=1) if want("make_images_realistic") else SKIP
    out["make_images_realistic"] = lvr if "error" in lvr else {k: lvr[k] for k in ("value", "unit", "seconds", "threads", "runs_mb_per_s", "data",
                                                                                     "stage_seconds_summed_over_workers", "synth_seconds")}
    out["polish_make_images"] = polish_make_images_leg(scratch) if want("polish_make_images") else SKIP
    # the two top entry points as one job each (stage walls inside): 256 Mb at 30x for call_variant, 64 Mb at 60x for polish
    out["call_variant"] = e2e_leg("call_variant", scratch, 256_000_000, 30, 3, 24) if want("call_variant") else SKIP
    # ... and with image generation and inference fused (options.fused_inference: the encoder's windows go to the model on the
    # device, candidate selection runs while the predictions are written; both HDF5 stores are still written)
    out["call_variant_fused"] = e2e_leg("call_variant_fused", scratch, 256_000_000, 30, 3, 24) if want("call_variant_fused") else SKIP
    out["polish_e2e"] = e2e_leg("polish", scratch, 64_000_000, 60, 2, 16) if want("polish_e2e") else SKIP
    out["polish_e2e_fused"] = e2e_leg("polish_fused", scratch, 64_000_000, 60, 2, 16) if want("polish_e2e_fused") else SKIP
    d = last_json([sys.executable, os.path.join(REPO, "tools", "bench_inflate.py"), "--genome", "8000000"], 300) if want("bgzf_inflate") else SKIP

    def inflate_roofline(d, profile="inflate_kernel_stats.txt"):
        """Two roofs of bgzf_inflate_kernel for the launch just timed: HBM (algorithmic bytes = compressed in + inflated out) and
        vector instruction issue (SQ_INSTS_VALU of the newest committed PMC pass of the same workload x 4 cycles over the SIMD-cycles
        the launch had).  The second one binds.  Counters from a round before inflate.hip last changed are marked stale."""
        roof = {"bound": "hbm", "kernel": "bgzf_inflate_kernel (inflate + the member's CRC-32 in its epilogue)",
                "algorithmic_bytes_per_launch": d["compressed_bytes"] + d["inflated_bytes"],
                "achieved": (d["compressed_bytes"] + d["inflated_bytes"]) / d["kernel_ms"] / 1e6, "peak": 8000.0, "unit": "GB/s", "traffic": None,
                "frac_over": "kernel time (HIP events around the launch)"}
        roof["frac"] = roof["achieved"] / roof["peak"]
        try:
            ins = {}
            src, src_round = newest_profile(profile)
            path = os.path.join(REPO, src)
            for line in open(path):
                parts = line.split()
                if len(parts) >= 2 and (parts[0].startswith("SQ_") or parts[0] in ("FETCH_SIZE", "WRITE_SIZE")):
                    ins[parts[0]] = float(parts[1])
            members = next(int(line.split()[1]) for line in open(path) if line.startswith("inflate:"))
            scale = d["members"] / members                      # (the committed pass had this many members per launch)
            stale = counters_stale(src_round, "inflate.hip")
            if "FETCH_SIZE" in ins and "WRITE_SIZE" in ins:      # KB per launch, each counter in its own pass (MI355X_MICROARCH.md)
                roof["traffic"] = (ins["FETCH_SIZE"] + ins["WRITE_SIZE"]) * 1024.0 * scale
                roof["traffic_over_algorithmic"] = roof["traffic"] / roof["algorithmic_bytes_per_launch"]
                roof["fetch_over_compressed"] = ins["FETCH_SIZE"] * 1024.0 * scale / d["compressed_bytes"]
                roof["traffic_note"] = ("FETCH_SIZE + WRITE_SIZE of %s: the writes are the inflated bytes once; the fetches are the "
                                        "compressed input plus the match sources (byte-wide gathers that pull whole lines, mostly of "
                                        "output this wavefront wrote moments ago) plus the epilogue's pass over the member for its CRC" % src)
            simd_cycles = 1024 * d["kernel_ms"] * 1e-3 * 2.4e9
            roof["issue"] = {"bound": "valu issue", "valu_wave_instructions": ins["SQ_INSTS_VALU"] * scale,
                             "salu_wave_instructions": ins["SQ_INSTS_SALU"] * scale, "lds_wave_instructions": ins["SQ_INSTS_LDS"] * scale,
                             "cycles_per_valu_instruction": 4, "simd_cycles_available": simd_cycles,
                             "frac": 4.0 * ins["SQ_INSTS_VALU"] * scale / simd_cycles,
                             # the CU's ONE scalar unit (one instruction per cycle for its four SIMDs): the unit the symbol walk lives on
                             "scalar_unit_frac": ins["SQ_INSTS_SALU"] * scale / (256 * d["kernel_ms"] * 1e-3 * 2.4e9),
                             "wait_share_of_wave_cycles": ins["SQ_WAIT_ANY"] / ins["SQ_WAVE_CYCLES"] if ins.get("SQ_WAVE_CYCLES") else None,
                             "source": src, "stale": stale}
            roof["stale"] = stale
        except Exception:       # noqa: BLE001
            roof["issue"] = None
        return roof
    out["bgzf_inflate"] = d if "error" in d else {
        "roofline": inflate_roofline(d),
        "value": d["device_GBps_inflated"], "unit": "GB/s of inflated bytes", "kernel": "bgzf_inflate_kernel", "kernel_ms": d["kernel_ms"],
        "stale_counters": bool((inflate_roofline(d).get("stale"))),
        "members": d["members"], "wave_slot_rounds": round(d["members"] / 6144.0, 2), "compressed_bytes": d["compressed_bytes"], "inflated_bytes": d["inflated_bytes"],
        "cpu_zlib_one_core_GBps": d["zlib_one_core_GBps"], "identical_to_zlib": d["sample_identical"],
        "cpu_baseline": {"value": d.get("host_library_all_cores_GBps"), "unit": "GB/s of inflated bytes", "cores": d.get("host_cores"),
                         "kind": "port", "one_core": d.get("host_library_one_core_GBps"), "identical": d.get("identical_to_host_library"),
                         "sample": "all the members through pa_bgzf_inflate_host (libdeflate, htslib's inflate), one thread per usable CPU"},
        "note": "csrc/inflate.hip on the BGZF members of a synthetic 8 Mb / 60x BAM (tools/synth_bam, libdeflate level 1), inputs "
                "resident, HIP events: one wavefront per member, 64 bit offsets decoded speculatively per step, then the member's "
                "CRC-32 against its trailer in the same wavefront's epilogue (one launch); bound by instruction "
                "issue and the symbol-to-symbol dependency of DEFLATE, not by HBM (the bytes moved are compressed in + inflated out)"}
    # the members as samtools writes them: zlib level 6, NM / MD / RG aux data (longer matches, longer codes)
    # (the same 8 Mb genome as the level-1 leg: a launch is whole rounds of 6 144 wavefront slots, and the 4 Mb file rounds 1-5 used
    # here -- 6 723 members, 1.09 rounds: a full round and a nearly empty one -- measured the tail, 32 GB/s, not the kernel, 40 GB/s:
    # profiles/r06_inflate_waves_ab.txt has both sizes side by side)
    d6 = last_json([sys.executable, os.path.join(REPO, "tools", "bench_inflate.py"), "--genome", "8000000", "--level", "6", "--tags", "1"], 400) if want("bgzf_inflate_level6") else SKIP
    out["bgzf_inflate_level6"] = d6 if "error" in d6 else {
        "value": d6["device_GBps_inflated"], "unit": "GB/s of inflated bytes", "kernel_ms": d6["kernel_ms"], "members": d6["members"],
        "compressed_bytes": d6["compressed_bytes"], "inflated_bytes": d6["inflated_bytes"], "identical_to_zlib": d6["sample_identical"],
        "identical_to_host_library": d6.get("identical_to_host_library"),
        "cpu_baseline": {"value": d6.get("host_library_all_cores_GBps"), "unit": "GB/s of inflated bytes", "cores": d6.get("host_cores"), "kind": "port"},
        "wave_slot_rounds": round(d6["members"] / 6144.0, 2),
        "note": "the same kernel on a synthetic 8 Mb / 60x BAM written with zlib level 6 and NM / MD / RG aux data in every record "
                "(rounds 1-5: a 4 Mb file, 1.09 rounds of wavefront slots -- 32.3 GB/s on it today)"}
    # ... and with quality strings that have run-length structure (binned plateaus, as a binning basecaller writes them): the members
    # compress > 3 x instead of 1.5 x -- more output per consumed bit, the case a real BAM is closer to
    dr = last_json([sys.executable, os.path.join(REPO, "tools", "bench_inflate.py"), "--genome", "8000000", "--level", "6", "--tags", "1",
                    "--quals", "1"], 400) if want("bgzf_inflate_realistic") else SKIP
    out["bgzf_inflate_realistic"] = dr if "error" in dr else {
        "value": dr["device_GBps_inflated"], "unit": "GB/s of inflated bytes", "kernel_ms": dr["kernel_ms"], "members": dr["members"],
        "compressed_bytes": dr["compressed_bytes"], "inflated_bytes": dr["inflated_bytes"],
        "compression_ratio": round(dr["inflated_bytes"] / max(1, dr["compressed_bytes"]), 2), "identical_to_zlib": dr["sample_identical"],
        "identical_to_host_library": dr.get("identical_to_host_library"),
        "cpu_baseline": {"value": dr.get("host_library_all_cores_GBps"), "unit": "GB/s of inflated bytes", "cores": dr.get("host_cores"), "kind": "port"},
        "wave_slot_rounds": round(dr["members"] / 6144.0, 2),
        "note": "zlib level 6, NM / MD / RG aux data, run-length quality strings (tools/synth_bam quals = 1), 8 Mb / 60x"}
    d = last_json([sys.executable, os.path.join(REPO, "tools", "bench_pipeline.py"), "--files", "16", "--windows", "524288", "--groups", "512",
                   "--workers", "0"] + extra, 600) if want("run_inference_hdf5") else SKIP
    out["run_inference_hdf5"] = d if "error" in d else {
        "value": d["windows_per_s"], "unit": "windows/s", "windows": d["windows"], "image_bytes": d["image_bytes"], "seconds": d["seconds"],
        "mode": d["mode"], "scratch": scratch or "system temporary directory",
        "note": "pepper_amd.variant.RunInference.run_inference: image HDF5 files -> predictions HDF5 (libhdf5 reads, H2D, forward, D2H, "
                "per-batch prediction groups), SURVEY.md 8(d) 'a second number including I/O'"}
    # three runs over the same files, the median reported: the job is 2-3 s of sixteen host CPUs' work beside the device passes and
    # its time varies by +-15 % from run to run on one box (profiles/r03_polish_pipeline_runs_ab.json)
    d = last_json([sys.executable, os.path.join(REPO, "tools", "bench_polish_pipeline.py"), "--chunks", "524288", "--files", "64",
                   "--workers", "0,0,0", "--median"] + extra, 900) if want("call_consensus_hdf5") else SKIP
    out["call_consensus_hdf5"] = d if "error" in d else {
        "value": d["chunks_per_s"], "unit": "chunks/s", "windows_per_s": d["windows_per_s"], "chunks": d["chunks"], "seconds": d["seconds"],
        "runs_chunks_per_s": d.get("runs_chunks_per_s"), "mode": d["mode"], "scratch": scratch or "system temporary directory",
        "note": "pepper_amd.polish.call_consensus.call_consensus: image HDF5 files -> predictions HDF5, start-up included (chunk reads "
                "bypass libhdf5, prediction files laid out by h5build.cpp, blocks of several reader lanes per device pass)"}
    return {k: v for k, v in out.items() if not (isinstance(v, dict) and v.get("error") == SKIP["error"])}


def wg_syn_bench(args, world, rank, device, ranks_seen, lib, handle, pool, unit_bytes, pool_n, host_call, sync):
    """--workload wg-syn: 24 shards with the chromosomes' proportions, a fixed total, dealt over the ranks by
    RunInference.shard_files both ways; every rank runs its shards through the host entry point (H2D, forward, D2H) and reports
    its own time.  value = all windows / the slowest rank's time under the size-ordered deal."""
    from pepper_amd.variant.RunInference import shard_files
    total = args.per_gpu or (1 << 22)
    shards = synthetic.wg_syn_shards(total)
    names = ["chr%d" % (k + 1) for k in range(22)] + ["chrX", "chrY"]
    size_of = dict(zip(names, shards))
    step = 1 << 18                                     # windows per host call (16 device passes)
    probs = torch.empty((step, 3), dtype=torch.float32, pin_memory=True)

    def run(my):
        done = 0
        sync()
        t0 = time.perf_counter()
        for name in my:
            left = size_of[name]
            while left > 0:
                c = min(step, left)
                off = (done % max(1, pool_n - c + 1))
                host_call(pool.data_ptr() + off * unit_bytes, c, [probs])
                left -= c
                done += c
        sync()
        return done, time.perf_counter() - t0

    def gather(x):
        if world == 1:
            return [x]
        import torch.distributed as dist
        out = [None] * world
        dist.all_gather_object(out, x)
        return out

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
    results = {}
    host_call(pool.data_ptr(), min(step, pool_n), [probs])      # warm-up
    for label, sizes in (("round_robin", None), ("size_ordered", shards)):
        chunks = shard_files(names, world, sizes)
        chunks += [[] for _ in range(world - len(chunks))]
        barrier()
        n_done, dt = run(chunks[rank])
        barrier()
        sys.stderr.write("[bench wg-syn] %s rank %d: %d shards, %d windows, %.3f s\n" % (label, rank, len(chunks[rank]), n_done, dt))
        per = gather((n_done, dt))
        results[label] = {"windows_per_rank": [p[0] for p in per], "seconds_per_rank": [round(p[1], 4) for p in per],
                          "imbalance_max_over_mean_windows": max(p[0] for p in per) / (sum(p[0] for p in per) / world),
                          "windows_per_s": sum(p[0] for p in per) / max(p[1] for p in per)}
    if rank == 0:
        best = results["size_ordered"]
        print(json.dumps({
            "metric": "inference windows/sec (whole node)", "value": best["windows_per_s"], "unit": "windows/s", "n_gpus": world,
            "steps": 1, "warmup": 1, "ms_per_step": max(best["seconds_per_rank"]) * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32 via f16 hi/lo split operands (3 MFMAs per product), f32 accumulate", "data": "synthetic",
            "config": {"workload": "WG-syn: %d V-syn windows in 24 shards proportional to the GRCh38 chromosome lengths (one image file per "
                                   "chromosome), a fixed job dealt over the ranks; H2D and D2H included" % sum(shards),
                       "shard_windows": dict(zip(names, shards)), "parallelism": f"file-shard x{world}, one weight broadcast, no data-path collective",
                       "ranks_seen": ranks_seen, "deal": "size_ordered (largest shard first onto the least loaded rank; RunInference.shard_files)"},
            "deals": results,
            "round_robin_over_size_ordered": results["round_robin"]["windows_per_s"] / best["windows_per_s"]}))


LINE_LIMIT = 6000            # bytes of the final stdout line: the driver parses that line, and a 20 KB one came back unparsed
FULL_RECORD = os.path.join("gpurun_out", "bench_full.json")


def _sig(x, digits=4):
    """Numbers to `digits` significant figures (the line is read by people and a parser, not re-used as data)."""
    if isinstance(x, bool) or x is None:
        return x
    if isinstance(x, int):
        return x
    if isinstance(x, float):
        if x != x or x in (float("inf"), float("-inf")):
            return None
        r = float("%.*g" % (digits, x))
        return int(r) if abs(r) >= 1e4 and r == int(r) else r
    return x


def _binding_roof(leg):
    """(frac, bound, over) of the roof that binds a secondary leg: the instruction-issue roof where the leg carries one (the
    integer kernels), its `roofline.frac` otherwise.  `over` says what time the fraction is over."""
    roof = leg.get("roofline") or {}
    issue = roof.get("issue") or {}
    if issue.get("frac") is not None:
        return issue["frac"], issue.get("bound", "valu issue"), roof.get("frac_over", "kernel time")
    if roof.get("frac_algorithmic_of_dtype_peak") is not None:
        return roof["frac_algorithmic_of_dtype_peak"], "mfma (algorithmic, of the dense f16 peak)", roof.get("frac_over", "kernel time")
    if roof.get("frac") is not None:
        return roof["frac"], roof.get("bound"), roof.get("frac_over", "kernel time")
    return None, None, None


def secondary_summary(secondary):
    """One {value, unit, frac, bound, cpu} tuple per secondary leg (frac / cpu where the leg has them); an error stays an error."""
    out = {}
    for name, leg in (secondary or {}).items():
        if not isinstance(leg, dict):
            continue
        if "error" in leg:
            out[name] = {"error": str(leg["error"])[:80]}
            continue
        t = {"value": _sig(leg.get("value")), "unit": leg.get("unit")}
        frac, bound, over = _binding_roof(leg)
        if frac is not None:
            t["frac"], t["bound"], t["over"] = _sig(frac, 3), bound, over
        cpu = leg.get("cpu_baseline") or {}
        if cpu.get("value") is not None:
            t["cpu"], t["cpu_cores"] = _sig(cpu["value"]), cpu.get("cores")
        if leg.get("stale_counters"):
            t["stale"] = True
        if "seconds" in leg and "stage_walls" in leg:
            t["seconds"] = _sig(leg["seconds"])
        out[name] = t
    return out


def final_line(full, limit=LINE_LIMIT, full_record=FULL_RECORD):
    """The ONE stdout line the driver parses, assembled from the full record: the contract's headline keys, `config`, `dtype`,
    `roofline` (with `traffic` and the algorithmic fraction of the dtype's dense peak), `cpu_baseline`, the default-batch call and one
    tuple per secondary leg -- numbers to four figures, prose cut to what names the thing.  Everything else (stage times, notes,
    per-run figures, the whole `secondary` block) is in `full_record`, written next to the profiles by the same run.  Optional
    blocks are dropped, least important first, until the line is under `limit` bytes; the required ones never are."""
    def short(s, n):
        return s if not isinstance(s, str) or len(s) <= n else s[:n - 1].rstrip() + "…"
    cfg = full.get("config", {})
    line = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                       "scaling", "vs_baseline", "dtype", "data")}
    line["value"], line["ms_per_step"] = _sig(full.get("value"), 6), _sig(full.get("ms_per_step"), 6)
    line["config"] = {k: (short(cfg[k], 140) if isinstance(cfg[k], str) else cfg[k]) for k in
                      ("workload", "per_gpu_per_step", "distinct_units_per_gpu", "units", "device_pass", "h2d_d2h", "reference_hdf5_batch",
                       "weights", "parallelism", "ranks_seen", "per_rank_seconds", "collective_backend") if k in cfg}
    if cfg.get("per_rank_image_legs"):
        line["config"]["per_rank_image_legs"] = {k: v for k, v in cfg["per_rank_image_legs"].items() if k != "note"}
    roof = full.get("roofline") or {}
    line["roofline"] = {k: _sig(roof.get(k), 5) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic",
                                                           "algorithmic_bytes_per_launch", "frac_algorithmic_of_dtype_peak",
                                                           "issued_tflops", "frac_of_dense_peak_issued", "mfma_busy_frac", "hbm_GBps",
                                                           "hbm_frac_of_peak", "kernel_avg_ms", "units_per_launch") if k in roof}
    line["roofline"]["frac_over"] = "kernel time (HIP events on the launch stream)"
    line["roofline"]["arithmetic"] = short(roof.get("arithmetic"), 90)
    detail = roof.get("traffic_detail") or {}
    line["roofline"]["counters_source"] = detail.get("source")
    if "cpu_baseline" in full:
        cb = full["cpu_baseline"]
        line["cpu_baseline"] = {k: (short(cb[k], 150) if isinstance(cb[k], str) else _sig(cb[k])) for k in
                                ("value", "unit", "cores", "kind", "sample", "host_physical_cores", "usable_cpus") if k in cb}
        line["speedup_vs_cpu_baseline"] = _sig(full.get("speedup_vs_cpu_baseline"))
    for key in ("batch512", "batch128"):
        if key in full:
            b = full[key]
            line[key] = {"units_per_call": b.get("units_per_call"),
                         "host_buffers": _sig((b.get("host_buffers") or {}).get("value")),
                         "device_resident": _sig((b.get("device_resident") or {}).get("value")), "unit": "windows/s"}
    if "device_resident" in full:
        line["device_resident"] = _sig(full["device_resident"].get("value"))
    if "secondary" in full:
        line["secondary_summary"] = secondary_summary(full["secondary"])
    if "shared_gpu_plumbing_check" in full:
        line["shared_gpu_plumbing_check"] = full["shared_gpu_plumbing_check"]
    optional = [("end_to_end_tflops", _sig(full.get("end_to_end_tflops"))),
                ("kernels", {k: {"avg_ms": v.get("avg_ms"), "share": v.get("share"), "frac_of_peak": v.get("frac_of_peak")}
                             for k, v in (full.get("kernels") or {}).items()}),
                ("hbm_view", {k: _sig(v) for k, v in (full.get("hbm_view") or {}).items() if k != "note"}),
                ("cpu_baseline_single_process", {k: _sig(v) for k, v in (full.get("cpu_baseline_single_process") or {}).items()
                                                 if k in ("value", "cores")})]
    for k, v in optional:
        if v:
            line[k] = v
    line["full_record"] = full_record
    drop = ["cpu_baseline_single_process", "hbm_view", "kernels", "end_to_end_tflops", "device_resident"]
    while len(json.dumps(line)) >= limit and drop:
        line.pop(drop.pop(0), None)
    if len(json.dumps(line)) >= limit and "secondary_summary" in line:       # the tuples, cut to {value, unit, frac}
        line["secondary_summary"] = {k: {kk: vv for kk, vv in t.items() if kk in ("value", "unit", "frac", "cpu", "error")}
                                     for k, t in line["secondary_summary"].items()}
    if len(json.dumps(line)) >= limit:
        line["config"] = {k: v for k, v in line["config"].items() if k in ("workload", "per_gpu_per_step", "h2d_d2h", "parallelism")}
    return line


def write_full_record(full, path=None):
    """The whole record beside the profiles (gpurun merges gpurun_out/ back); never fatal."""
    path = path or os.path.join(REPO, FULL_RECORD)
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as fh:
            json.dump(full, fh, indent=1)
        return True
    except OSError as e:
        sys.stderr.write("[bench] full record not written (%s)\n" % e)
        return False


def rank_image_legs(rank, world, device):
    """N > 1 only: every rank ALSO runs image generation -- variant generate_images and the polish chain's make_images -- on its own
    synthetic BAM (its own shard: weak scaling, as the reference deals regions over processes, ImageGenerationUI.py:326-339) on its
    own device, all ranks at once, with the box's usable CPUs dealt evenly over the ranks.  -> this rank's dict; rank 0 gathers
    them.  The first 8-GPU run then measures region sharding of image generation, not only of the model."""
    import shutil
    import tempfile
    sys.path.insert(0, os.path.join(REPO, "tools"))
    from pepper_amd.hostinfo import usable_cpus
    out = {"rank": rank, "device": device, "threads": max(1, usable_cpus() // world)}
    base = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
    try:
        st = os.statvfs(base)
        if st.f_bavail * st.f_frsize < world * (3 << 30):
            base = tempfile.gettempdir()
    except OSError:
        base = tempfile.gettempdir()
    work = tempfile.mkdtemp(prefix="pepper_amd_rank%d_" % rank, dir=base)
    barriers = 0                 # (of the two the ranks meet at: a rank whose leg failed still has to turn up at the rest)
    try:
        import bench_variant_images
        from pepper_amd.polish.make_images import make_images
        from pepper_amd.variant.ImageGenerationUI import ImageGenerationUtils
        vdir, pdir = os.path.join(work, "v"), os.path.join(work, "p")
        vbases = int(os.environ.get("PEPPER_AMD_BENCH_RANK_VARIANT_BASES", 16_000_000))
        pbases = int(os.environ.get("PEPPER_AMD_BENCH_RANK_POLISH_BASES", 4_000_000))
        bench_variant_images.make_fast(vdir, vbases, 60, 2027 + rank)
        bench_variant_images.make_fast(pdir, pbases, 60, 3027 + rank)

        def variant(tag):
            o = bench_variant_images.options(vdir, os.path.join(vdir, tag), out["threads"], 100000, device=device)
            ImageGenerationUtils.generate_images(o)
            shutil.rmtree(os.path.join(vdir, tag), ignore_errors=True)

        def polish(tag):
            make_images(os.path.join(pdir, "reads.bam"), os.path.join(pdir, "draft.fa"), None, os.path.join(pdir, tag), out["threads"],
                        device_ids=str(device))
            shutil.rmtree(os.path.join(pdir, tag), ignore_errors=True)
        variant("warm")
        polish("warm")
        import torch.distributed as dist
        for key, fn, mb in (("make_images_mb_per_s", variant, vbases / 1e6), ("polish_make_images_mb_per_s", polish, pbases / 1e6)):
            dist.barrier()
            barriers += 1
            t0 = time.perf_counter()
            fn("timed")
            dt = time.perf_counter() - t0
            out[key], out[key.replace("mb_per_s", "seconds")] = round(mb / dt, 2), round(dt, 3)
    except Exception as e:      # noqa: BLE001 -- the model line must not be lost to a failing image leg
        out["error"] = repr(e)[:200]
        try:                    # (the others are waiting at the barriers above)
            import torch.distributed as dist
            for _ in range(2 - barriers):
                dist.barrier()
        except Exception:       # noqa: BLE001
            pass
    finally:
        shutil.rmtree(work, ignore_errors=True)
    return out


def timed_loop(fn, count, sync):
    """count calls of fn between two synchronisations -> seconds."""
    sync()
    t0 = time.perf_counter()
    for k in range(count):
        fn(k)
    sync()
    return time.perf_counter() - t0


def main():
    args = parse()
    if args.cpu_worker:
        cpu_worker(args.model, args.cpu_threads, args.cpu_seconds)
        return
    if args.model == "realign":
        torch.cuda.set_device(0)
        realign_bench(args)
        return
    if args.model == "encoder":
        encoder_bench(args)
        return
    if args.model == "polish-encoder":
        torch.cuda.set_device(0)
        polish_encoder_bench(args)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_spawn(args)
    world, rank, device, ranks_seen = dist_setup(args)
    if world != args.gpus and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
    dev = torch.device("cuda", device)
    lib = _lib.load()
    import ctypes
    stream = torch.cuda.Stream(device=dev)
    variant = args.model == "variant"

    if variant:
        chunk = 16384                                   # windows per device pass
        per = args.per_gpu or (chunk if args.resident_only else 1 << 18)
        pool_n = max(per, args.pool or (per if args.resident_only else 1 << 20))
        sd = broadcast_state_dict(lambda: synthetic.variant_state_dict(seed=0),
                                  synthetic.variant_param_shapes(), world, rank, dev)
        cfg = _lib.VariantConfig(26, 33, 1, 3, device, chunk)
        names, data, numel, n, keep = _lib.marshal_state_dict(sd)
        handle = ctypes.c_void_p()
        _lib.check(lib.pa_variant_create(ctypes.byref(cfg), names, data, numel, n,
                                         ctypes.c_void_p(stream.cuda_stream), ctypes.byref(handle)))
        if args.resident_only:
            # profiling runs: the numpy recipe (no torch kernels, so rocprofv3's database does not carry torch's code objects)
            base = synthetic.variant_windows(min(pool_n, 16384), seed=synthetic.VSYN_SEED + rank)
            pool_dev = torch.from_numpy(np.resize(base, (pool_n, 33, 26))).to(dev)
        else:
            pool_dev = synthetic.variant_windows_device(pool_n, seed=synthetic.VSYN_SEED + rank, device=dev)
        unit_shape, out_shapes, out_dtype = (33, 26), [(3,)], torch.float32
        windows_per_unit, flop_per_window = 1, VARIANT_FLOP_PER_WINDOW
        bytes_per_unit = VARIANT_BYTES_PER_WINDOW
        workload = ("V-syn: int8 [N,33,26] candidate windows, variant bi-LSTM(26->256)x2 + MLP head, "
                    "F=26 H=256 L=1 (BASELINE configs[1] shapes)")

        def host_call(x_ptr, count, outs):
            _lib.check(lib.pa_variant_forward_host(handle, x_ptr, count, outs[0].data_ptr(), None))

        def device_call(x_ptr, count, outs):
            _lib.check(lib.pa_variant_forward_device(handle, x_ptr, count, outs[0].data_ptr(), None))
    else:
        feat = 100 if args.model == "ns-literal" else 10
        chunk = 16384                                   # chunks per device pass (128 rows per workgroup x 2 directions = 256 workgroups)
        per = args.per_gpu or (chunk if (args.resident_only or feat != 10) else 32768)
        pool_n = max(per, args.pool or (per if (args.resident_only or feat != 10) else 65536))
        sd = broadcast_state_dict(lambda: synthetic.polish_state_dict(seed=0, image_features=feat),
                                  synthetic.polish_param_shapes(image_features=feat), world, rank, dev)
        cfg = _lib.PolishConfig(feat, 128, 1, 5, 1000, 100, 50, 50, device, chunk)
        names, data, numel, n, keep = _lib.marshal_state_dict(sd)
        handle = ctypes.c_void_p()
        _lib.check(lib.pa_polish_create(ctypes.byref(cfg), names, data, numel, n,
                                        ctypes.c_void_p(stream.cuda_stream), ctypes.byref(handle)))
        if feat == 10 and args.resident_only:
            base = synthetic.polish_chunks(min(pool_n, 1024), seed=synthetic.PSYN_SEED + rank)
            pool_dev = torch.from_numpy(np.resize(base, (pool_n, 1000, 10))).to(dev)
        elif feat == 10:
            pool_dev = synthetic.polish_chunks_device(pool_n, seed=synthetic.PSYN_SEED + rank, device=dev)
        else:   # rows of small counts spread over 100 columns, generated on the device (1.6 GB per 16384 chunks)
            gen = torch.Generator(device=dev).manual_seed(synthetic.PSYN_SEED + rank)
            pool_dev = torch.poisson(torch.full((pool_n, 1000, feat), 2.5, device=dev), generator=gen).clamp_(0, 254).to(torch.uint8)
        unit_shape, out_shapes, out_dtype = (1000, feat), [(1000,), (1000,)], torch.uint8
        windows_per_unit = POLISH_WINDOWS_PER_CHUNK
        flop_per_window = POLISH_FLOP_PER_WINDOW + 2.0 * 100 * 2 * 384 * (feat - 10)
        bytes_per_unit = 1000 * feat + 2000
        workload = ("P-syn: uint8 [N,1000,10] chunks, polish bi-GRU(10->128)x2 + dense, 19 windows of "
                    "100 steps with hidden carry (BASELINE configs[4] shapes)") if feat == 10 else (
                    "NS-literal: uint8 [N,1000,100] chunks = 19 windows of 100 steps x 100 features through the polish "
                    "bi-GRU(100->128)x2 + dense with hidden carry; the north_star's literal synthetic shape, not a "
                    "reference shape (SURVEY.md section 0), no CPU baseline")

        def host_call(x_ptr, count, outs):
            _lib.check(lib.pa_polish_predict_host(handle, x_ptr, count, outs[0].data_ptr(), outs[1].data_ptr(), None))

        def device_call(x_ptr, count, outs):
            _lib.check(lib.pa_polish_predict_device(handle, x_ptr, count, outs[0].data_ptr(), outs[1].data_ptr(), None))

    unit_bytes = int(np.prod(unit_shape))
    torch.cuda.synchronize(dev)
    if args.resident_only:
        pool, outs = pool_dev, [torch.empty((per,) + sh, dtype=out_dtype, device=dev) for sh in out_shapes]
        call = device_call
    else:
        # the boundary's host side: a page-locked pool of summaries (what the HDF5 reader fills) and page-locked results
        pool = torch.empty((pool_n,) + unit_shape, dtype=pool_dev.dtype, pin_memory=True)
        pool.copy_(pool_dev)
        outs = [torch.empty((per,) + sh, dtype=out_dtype, pin_memory=True) for sh in out_shapes]
        call = host_call
    slots = max(1, pool_n // per)

    def step(k):
        call(pool.data_ptr() + (k % slots) * per * unit_bytes, per, outs)

    def sync():
        _lib.check(lib.pa_synchronize(handle))
        torch.cuda.synchronize(dev)

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()

    if variant and args.workload == "wg-syn":
        if args.resident_only:
            raise SystemExit("--workload wg-syn times the host entry point: not with --resident-only")
        wg_syn_bench(args, world, rank, device, ranks_seen, lib, handle, pool, unit_bytes, pool_n, host_call, sync)
        lib.pa_variant_destroy(handle)
        if world > 1:
            leave_group()
        return
    for k in range(args.warmup):
        step(k)
    sync()
    _lib.check(lib.pa_profile_enable(handle, 1))
    barrier()
    sync()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(args.warmup + k)
    sync()
    barrier()
    dt = time.perf_counter() - t0
    per_rank_seconds = [dt]
    sys.stderr.write("[bench] rank %d of %d: %d steps in %.4f s\n" % (rank, world, args.steps, dt))     # every rank, for the SCALE log
    if world > 1:
        import torch.distributed as dist
        on_gpu = dist.get_backend() == "nccl"
        gathered = [None] * world
        dist.all_gather_object(gathered, dt)
        per_rank_seconds = [float(x) for x in gathered]
        tt = torch.tensor([dt], dtype=torch.float64, device=dev if on_gpu else None)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    prof = _lib.profile_dict(handle)
    _lib.check(lib.pa_profile_enable(handle, 0))

    image_legs = None
    if world > 1 and variant and not args.no_image_legs and not args.resident_only:
        import torch.distributed as dist
        mine = rank_image_legs(rank, world, device)
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        image_legs = {"threads_per_rank": mine["threads"],
                      "note": "every rank runs generate_images (variant) and make_images (polish chain) on its own synthetic BAM and its "
                              "own device, all ranks at once; aggregate = all ranks' Mb / the slowest rank's seconds"}
        for key, mb in (("make_images", "PEPPER_AMD_BENCH_RANK_VARIANT_BASES"), ("polish_make_images", "PEPPER_AMD_BENCH_RANK_POLISH_BASES")):
            rates = [g.get(key + "_mb_per_s") for g in gathered]
            secs = [g.get(key + "_seconds") for g in gathered]
            image_legs[key + "_mb_per_s"] = rates
            if all(r is not None for r in rates):
                image_legs["aggregate_" + key + "_mb_per_s"] = round(sum(r * t for r, t in zip(rates, secs)) / max(secs), 2)
        errs = [g["error"] for g in gathered if "error" in g]
        if errs:
            image_legs["errors"] = errs[:2]

    extras = {}
    if world == 1 and not args.no_extras and not args.resident_only:
        # (a) the same kernels on inputs already in HBM (round 1's figure): one device pass of `chunk` units, repeated
        xr = pool_dev[:chunk].contiguous()
        outs_d = [torch.empty((chunk,) + sh, dtype=out_dtype, device=dev) for sh in out_shapes]
        reps = max(10, min(60, int(0.5 * per * args.steps / chunk)))
        device_call(xr.data_ptr(), chunk, outs_d)
        t = timed_loop(lambda k: device_call(xr.data_ptr(), chunk, outs_d), reps, sync)
        extras["device_resident"] = {"value": reps * chunk * windows_per_unit / t, "unit": "windows/s",
                                     "units_per_pass": chunk, "passes": reps, "ms_per_pass": t / reps * 1e3,
                                     "note": "inputs and outputs in HBM, no copies (round 1's headline definition)"}
        # (b) the reference's default batch through one call (BASELINE configs[1]: batch = 512 windows; polish: 128 chunks)
        b = 512 if variant else 128
        outs_b = [torch.empty((b,) + sh, dtype=out_dtype, pin_memory=True) for sh in out_shapes]
        host_call(pool.data_ptr(), b, outs_b)
        nb = 200 if variant else 50
        th = timed_loop(lambda k: (host_call(pool.data_ptr() + (k % 64) * b * unit_bytes, b, outs_b)), nb, sync)
        outs_bd = [torch.empty((b,) + sh, dtype=out_dtype, device=dev) for sh in out_shapes]
        td = timed_loop(lambda k: device_call(pool_dev.data_ptr() + (k % 64) * b * unit_bytes, b, outs_bd), nb, sync)
        extras["batch512" if variant else "batch128"] = {
            "units_per_call": b,
            "host_buffers": {"value": nb * b * windows_per_unit / th, "unit": "windows/s", "ms_per_call": th / nb * 1e3,
                             "note": "one synchronous call per batch: H2D, forward, D2H, as predict_distributed_gpu.py:58-67 loops"},
            "device_resident": {"value": nb * b * windows_per_unit / td, "unit": "windows/s", "ms_per_call": td / nb * 1e3,
                                "note": "calls queued back to back on one stream, inputs in HBM"}}
        if variant and hasattr(lib, "pa_variant_split_fallbacks"):
            # calls of this size run their step loops with a tile's hidden units split over eight workgroups; a
            # call whose workgroups did not meet on the GPU is run again the ordinary way -- how many of the 400 above were
            again = ctypes.c_int64(-1)
            _lib.check(lib.pa_variant_split_fallbacks(handle, ctypes.byref(again)))
            extras["batch512"]["schedule"] = {"unit_split": os.environ.get("PA_UNIT_SPLIT", "1") != "0", "calls_run_again": again.value}

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
        h2 = "_h2" in dom
        traffic = measured_traffic(args.model, dom)
        line = {
            "metric": "inference windows/sec (whole node)",
            "value": value, "unit": "windows/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if not any("_h2" in k for k in prof) else "f32 via f16 hi/lo split operands (3 MFMAs per product), f32 accumulate",
            "data": "synthetic",
            "config": {"workload": workload, "per_gpu_per_step": per, "distinct_units_per_gpu": pool_n,
                       "units": "windows" if variant else "chunks (x19 windows)",
                       "device_pass": chunk, "h2d_d2h": "excluded (--resident-only)" if args.resident_only else "included",
                       "host_path": None if args.resident_only else
                       "page-locked host pool -> H2D -> forward -> D2H (pa_*_host), copies of neighbouring device passes "
                       "overlapped with the kernels on separate HIP streams",
                       "timed_seconds": dt,
                       "reference_hdf5_batch": 512 if variant else 128,
                       "weights": "seeded random init (pepper_amd.synthetic), fp32",
                       "parallelism": f"region-shard x{world}, one weight broadcast, no data-path collective",
                       "ranks_seen": ranks_seen, "per_rank_seconds": [round(x, 4) for x in per_rank_seconds],
                       "collective_backend": COLLECTIVE_NOTE, "per_rank_image_legs": image_legs},
            # achieved = ALGORITHMIC flops of the dominant kernel's launches / their HIP-event time; peak = the dense MFMA peak of the
            # datatype the MFMAs run in (f16: 2.5 PFLOP/s); frac = achieved / peak.  Each f32-accurate product costs three f16 MFMAs, so
            # the machine issues 3 x achieved: frac_of_dense_peak_issued says how busy the matrix cores are, frac how much useful work.
            "roofline": {"bound": "mfma", "kernel": dom, "achieved": ach, "peak": F16_MFMA_PEAK_TFLOPS if h2 else F32_MFMA_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": ach / (F16_MFMA_PEAK_TFLOPS if h2 else F32_MFMA_PEAK_TFLOPS),
                         "kernel_avg_ms": d["ms"] / d["launches"], "units_per_launch": min(per, chunk),
                         "traffic": traffic["bytes_per_launch"] if traffic else None,
                         "traffic_detail": traffic,
                         "mfma_busy_frac": traffic.get("mfma_busy_frac") if traffic else None,
                         "hbm_GBps": traffic.get("hbm_GBps_profiled") if traffic else None,
                         "hbm_frac_of_peak": (traffic["hbm_GBps_profiled"] / HBM_PEAK_GBPS) if traffic and traffic.get("hbm_GBps_profiled") else None,
                         "counters_note": "mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 1024 SIMDs), hbm_GBps = (FETCH_SIZE + "
                                          "WRITE_SIZE bytes) / profiled launch duration: rocprofv3 --pmc passes of this command with "
                                          "--resident-only, committed under profiles/ (traffic_detail.source)",
                         "algorithmic_bytes_per_launch": (ALGORITHMIC_BYTES_PER_UNIT[dom] * min(per, chunk)
                                                          if dom in ALGORITHMIC_BYTES_PER_UNIT else None),
                         "frac_algorithmic_of_dtype_peak": ach / (F16_MFMA_PEAK_TFLOPS if h2 else F32_MFMA_PEAK_TFLOPS),
                         "issued_tflops": ach * (3 if h2 else 1),
                         "frac_of_dense_peak_issued": ach * (3 if h2 else 1) / (F16_MFMA_PEAK_TFLOPS if h2 else F32_MFMA_PEAK_TFLOPS),
                         "arithmetic": ("3 x v_mfma_f32_32x32x16_f16 per product (f16 hi/lo split operands, f32 "
                                        "accumulate); peak = 2.5 PFLOP/s dense f16 / 3; `achieved` counts algorithmic "
                                        "(f32-equivalent) FLOP, `issued_tflops` the machine MFMAs") if h2
                         else "v_mfma_f32_32x32x2_f32"},
            "end_to_end_tflops": value * flop_per_window / 1e12,
            "hbm_view": {"algorithmic_bytes_per_unit": bytes_per_unit,
                         "achieved_GBps": value / windows_per_unit * bytes_per_unit / 1e9, "peak_GBps": 8000.0,
                         "note": "the path is matrix-bound (arithmetic intensity ~1e5 FLOP/B): shown only because north_star asks for it"},
            "kernels": kern,
        }
        if os.environ.get("PEPPER_AMD_BENCH_SHARE_GPU") == "1" and world > torch.cuda.device_count():
            line["shared_gpu_plumbing_check"] = True
        line.update(extras)
        if "device_resident" in extras:
            line["host_path_over_device_resident"] = value / extras["device_resident"]["value"]
        if world == 1 and not args.no_cpu_baseline and args.model != "ns-literal":
            # whole-box CPU number = the reference's own scheme (single-thread workers, one per physical core); the
            # single-process multi-thread figure is kept beside it
            single = cpu_baseline(args.model, args.cpu_seconds)
            multi = cpu_baseline_workers(args.model, args.cpu_seconds)
            line["cpu_baseline"] = multi if multi and multi["value"] > single["value"] else single
            line["cpu_baseline_single_process"] = single
            line["speedup_vs_cpu_baseline"] = value / line["cpu_baseline"]["value"]
            # BASELINE.md holds no published number for this metric: vs_baseline stays null; the ratio north_star targets (>= 50 x
            # the reference CPU run_inference on the same box's host cores) is speedup_vs_cpu_baseline
        if variant and world == 1 and not args.no_secondary and not args.resident_only and not args.no_extras:
            sync()
            line["secondary"] = secondary_block(args)
        # the whole record goes to a file; the stdout line is the part the driver parses, under LINE_LIMIT bytes (final_line)
        if args.full_line:
            print(json.dumps(line))
        else:
            write_full_record(line)
            print(json.dumps(final_line(line)))

    if variant:
        lib.pa_variant_destroy(handle)
    else:
        lib.pa_polish_destroy(handle)
    if world > 1:
        leave_group()


if __name__ == "__main__":
    main()
