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
