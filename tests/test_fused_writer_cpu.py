"""The fused call_variant's prediction writer (pepper_amd/variant/fused.py) on the CPU: no model is touched -- intervals with
their probabilities are submitted from several threads, as the image workers do, and the predictions file must hold every
candidate once, in batches of options.batch_size cut from ONE worker's consecutive intervals (a batch that mixed two workers'
far-apart intervals would make the candidate finder fetch megabases of reference for it, or give up on the batch).
Layout: /root/reference/pepper_variant/modules/python/DataStorePredict.py:26-67."""
import threading
from types import SimpleNamespace

import numpy as np

from pepper_amd import h5


def _interval(rng, start, n):
    positions = np.sort(rng.choice(np.arange(start, start + 100000), n, replace=False)).astype(np.int32)
    names = [("%d:A%d" % (p, p % 7)).encode() for p in positions]
    blob = b"".join(x + b"\0" for x in names)
    offsets = np.concatenate([[0], np.cumsum([len(x) + 1 for x in names])]).astype(np.int64)
    out = {"positions": positions, "depths": (positions % 60).astype(np.uint8), "candidates_blob": np.frombuffer(blob, np.uint8),
           "candidates_offsets": offsets, "candidate_frequency": (positions % 11).astype(np.uint8)}
    probs = rng.random((n, 3)).astype(np.float32)
    return out, probs


def test_batches_come_from_one_workers_consecutive_intervals(tmp_path):
    from pepper_amd.variant.fused import FusedPredictor
    options = SimpleNamespace(batch_size=64, fused_candidates_off=True, model_path=None)
    sink = FusedPredictor(options, str(tmp_path) + "/")
    rng = np.random.default_rng(3)
    # worker w: two runs of three adjacent 100 kb intervals, 40 Mb apart; worker 2 moves to another contig for its second run
    plans, expected = [], {}
    for w in range(3):
        plan = []
        for run in range(2):
            contig = "ctgB" if (w == 2 and run == 1) else "ctgA"
            for k in range(3):
                start = w * 5000000 + run * 40000000 + k * 100000
                out, probs = _interval(rng, start, int(rng.integers(50, 200)))
                plan.append((contig, out, probs))
                for p, q in zip(out["positions"].tolist(), probs):
                    expected[(contig, p)] = q
        plans.append(plan)
    gate = threading.Barrier(3)

    def worker(plan):
        gate.wait()
        for contig, out, probs in plan:
            sink.submit(contig, out, probs)
    threads = [threading.Thread(target=worker, args=(plan,)) for plan in plans]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    batches, windows = sink.close()
    assert windows == len(expected)
    seen, short = {}, 0
    with h5.File(str(tmp_path) + "/pepper_prediction.hdf") as f:
        names = f.keys("predictions")
        assert sorted(names, key=lambda s: int(s.split("_")[1])) == ["batch_%d" % k for k in range(batches)]
        for name in names:
            base = "predictions/" + name + "/"
            contigs, pos = np.asarray(f[base + "contigs"]), np.asarray(f[base + "positions"])
            prob, cand = np.asarray(f[base + "base_prediction"]), f[base + "candidates"]
            assert len(pos) <= 64 and len(set(contigs.tolist())) == 1
            assert int(pos.max()) - int(pos.min()) < 400000 and (np.diff(pos) > 0).all()        # one worker's one run, in order
            short += len(pos) < 64
            for c, p, q, text in zip(contigs.tolist(), pos.tolist(), prob, np.asarray(cand).reshape(-1).tolist()):
                key = (c.decode(), p)
                assert key not in seen
                seen[key] = q
                text = text.decode() if isinstance(text, bytes) else text
                assert text == "%d:A%d" % (p, p % 7)
                assert np.allclose(q, expected[key].astype(np.float64))
    assert len(seen) == len(expected)
    assert short <= 6                      # one short batch per run of a worker at most


def test_a_model_that_cannot_be_loaded_fails_every_worker_instead_of_hanging_them(tmp_path, monkeypatch):
    """HANDLES workers take the 'make' branch; the rest wait on the free queue.  A load that raises must reach the waiters too
    (they used to block forever on a handle nobody would put back), and close(failed=True) must not publish a predictions file."""
    import os
    import time
    import torch
    from pepper_amd.variant import fused
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: None)
    gate = threading.Event()

    def load(*a, **k):
        gate.wait(5)                   # hold the makers until the waiters are parked on the queue
        raise MemoryError("no room for the checkpoint")
    monkeypatch.setattr(fused.ModelHandler, "load_simple_model_for_training", staticmethod(load))
    sink = fused.FusedPredictor(SimpleNamespace(batch_size=64, fused_candidates_off=True, model_path="nowhere.pkl"), str(tmp_path) + "/")
    errors = []

    def worker():
        try:
            sink._model(0)
            errors.append(None)
        except BaseException as err:      # noqa: BLE001
            errors.append(err)
    threads = [threading.Thread(target=worker, daemon=True) for _ in range(5)]
    for t in threads:
        t.start()
    time.sleep(0.3)
    gate.set()
    for t in threads:
        t.join(10)
    assert not any(t.is_alive() for t in threads), "a worker is still waiting for a model handle"
    assert len(errors) == 5 and all(isinstance(e, (MemoryError, RuntimeError)) for e in errors)
    assert sum(isinstance(e, MemoryError) for e in errors) == fused.FusedPredictor.HANDLES
    sink.close(failed=True)
    assert not os.path.exists(str(tmp_path) + "/pepper_prediction.hdf")


def test_polish_fused_model_failure_reaches_the_waiters(tmp_path, monkeypatch):
    import time
    import torch
    from pepper_amd.polish import fused
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: None)
    gate = threading.Event()

    def load(*a, **k):
        gate.wait(5)
        raise OSError("bad model_path")
    monkeypatch.setattr(fused.ModelHandler, "load_simple_model_for_training", staticmethod(load))
    owner = fused.FusedConsensus("nowhere.pkl", str(tmp_path) + "/")
    errors = []

    def worker():
        try:
            owner._model(0)
            errors.append(None)
        except BaseException as err:      # noqa: BLE001
            errors.append(err)
    threads = [threading.Thread(target=worker, daemon=True) for _ in range(4)]
    for t in threads:
        t.start()
    time.sleep(0.3)
    gate.set()
    for t in threads:
        t.join(10)
    assert not any(t.is_alive() for t in threads)
    assert len(errors) == 4 and all(isinstance(e, (OSError, RuntimeError)) for e in errors)


def test_polish_gather_hands_every_chunk_to_exactly_one_pass(monkeypatch):
    """The fused polish's shared gather buffer (polish/fused.py _Gather) without a device: sixteen workers reserve / commit
    pieces of 1 - 40 chunks against passes of 256, a pass takes a while (so that every set is busy and the workers wait side by
    side -- the case that once left half-filled sets behind and hung the run); every chunk reaches exactly one pass, a piece is
    never split over two passes, no pass is larger than a set, and finish() returns."""
    import random
    import time
    from concurrent.futures import ThreadPoolExecutor
    from pepper_amd.polish import fused

    class FakeSet(object):
        def __init__(self, device, chunks, seq, features):
            self.meta = [None] * chunks
            self.reserved = self.done = 0
            self.sealed = self.busy = False
    monkeypatch.setattr(fused, "_GatherSet", FakeSet)
    monkeypatch.setattr(fused.torch.cuda, "set_device", lambda d: None)
    seen, sizes, lock = [], [], threading.Lock()

    class Gather(fused._Gather):
        def _predict_and_write(self, st, n):
            time.sleep(0.004)
            with lock:
                seen.extend(st.meta[:n])
                sizes.append(n)
    owner = SimpleNamespace(PASS_CHUNKS=256, passes=ThreadPoolExecutor(max_workers=3), models_lock=threading.Lock(), chunks=0,
                            passes_run=0, error=None, check=lambda: None, fail=lambda err: None)
    g = Gather(owner, 0)
    pieces = {}

    def worker(w):
        rng = random.Random(w)
        for k in range(120):
            n = rng.randint(1, 40)
            st, off = g.reserve(n)
            for j in range(n):
                st.meta[off + j] = (w, k, j)
            if rng.random() < 0.3:
                time.sleep(0.001)
            g.commit(st, n)
            pieces[(w, k)] = n
    threads = [threading.Thread(target=worker, args=(w,), daemon=True) for w in range(16)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(60)
    assert not any(t.is_alive() for t in threads), "a worker is stuck in reserve()"
    finisher = threading.Thread(target=g.finish, daemon=True)
    finisher.start()
    finisher.join(30)
    assert not finisher.is_alive(), "finish() did not return"
    owner.passes.shutdown(wait=True)
    total = sum(pieces.values())
    assert len(seen) == total == owner.chunks and len(set(seen)) == total
    assert max(sizes) <= 256 and owner.passes_run == len(sizes)
    # a piece's chunks are adjacent inside one pass
    at = {m: i for i, m in enumerate(seen)}
    bounds = np.cumsum([0] + sizes)
    for (w, k), n in pieces.items():
        idx = [at[(w, k, j)] for j in range(n)]
        assert idx == list(range(idx[0], idx[0] + n))
        assert np.searchsorted(bounds, idx[0], side="right") == np.searchsorted(bounds, idx[-1], side="right")
