"""HDF5 layouts: files written by pepper_amd's stores must be structurally and
numerically identical to files written by the REFERENCE's DataStore classes (golden fixtures made
by tests/golden/make_golden_hdf5.py), and the reader must read the reference's files."""
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

from pepper_amd import h5


def dump(path):
    """{dataset path: (shape, class, elem size, signed, values)} via pepper_amd.h5."""
    out = {}
    with h5.File(path, "r") as f:
        def walk(g):
            for name in f.keys(g):
                p = (g.rstrip("/") + "/" + name) if g != "/" else name
                try:
                    shape, cls, size, sgn = f.info(p)
                except h5.H5Error:
                    walk(p)
                    continue
                out[p] = (shape, cls, size, sgn, f[p])
        walk("/")
    return out


def assert_same_tree(a, b):
    assert sorted(a) == sorted(b)
    for k in a:
        sa, ca, za, ga, va = a[k]
        sb, cb, zb, gb, vb = b[k]
        assert (sa, ca, za, ga) == (sb, cb, zb, gb), k
        assert np.array_equal(np.asarray(va), np.asarray(vb)), k


@pytest.mark.parametrize("builder", [True, False])
def test_variant_images_reader_and_writer(golden_dir, tmp_path, monkeypatch, builder):
    """Both writers of the image store -- the append-only one image generation uses (h5build.cpp: fixed-width contig strings,
    the candidate strings as references into global heap collections) and the libhdf5 one -- against the file the reference's
    own DataStore wrote under h5py: same tree for libhdf5, no difference for h5diff, same arrays and str for h5py."""
    from pepper_amd.variant.DataStore import DataStore
    from pepper_amd.variant.models.dataloader_predict import SequenceDataset
    monkeypatch.setenv("PEPPER_AMD_H5_BUILDER", "1" if builder else "0")
    ref = os.path.join(golden_dir, "variant_images_ref.hdf5")
    inp = np.load(os.path.join(golden_dir, "variant_images_inputs.npz"), allow_pickle=True)
    names = sorted({k.split("__")[0] for k in inp.files})
    mine = str(tmp_path / "mine.hdf5")
    with DataStore(mine, "w") as ds:
        assert isinstance(ds.file_handler, h5.PredictionBuilder) == builder
        for n in names:
            g = {k.split("__")[1]: inp[k] for k in inp.files if k.startswith(n + "__")}
            ds.write_summary(n, g["contigs"].tolist(), g["positions"].tolist(), g["depths"].tolist(),
                             g["candidates"].tolist(), g["candidate_frequency"].tolist(), g["images"].tolist(),
                             [0] * len(g["contigs"]), [0] * len(g["contigs"]), False)
    assert_same_tree(dump(ref), dump(mine))
    if os.path.exists("/opt/conda/bin/h5diff"):
        r = subprocess.run(["/opt/conda/bin/h5diff", ref, mine], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    # many groups (several heap collections, a B-tree of more than one level), one without candidates, long alleles
    big = str(tmp_path / "big.hdf5")
    rng = np.random.default_rng(5)
    want = {}
    with DataStore(big, "w") as ds:
        for k in range(300):
            n = 0 if k == 17 else int(rng.integers(1, 40))
            cands = [["%d%s" % (rng.integers(1, 4), "".join(rng.choice(list("ACGT"), int(rng.integers(1, 60)))))] for _ in range(n)]
            img = rng.integers(-128, 128, (n, 33, 26)).astype(np.int8)
            pos = rng.integers(0, 1 << 30, n)
            want["ctg%d_%d_%d" % (k % 3, k, k + 1)] = (pos, cands, img)
            ds.write_summary("ctg%d_%d_%d" % (k % 3, k, k + 1), ["ctg%d" % (k % 3)] * n, pos, rng.integers(0, 126, n),
                             np.array(cands, dtype=object).reshape(n, 1), rng.integers(0, 126, (n, 1)), img, [0] * n, [0] * n, False)
    with h5.File(big) as f:
        assert sorted(f.keys("summaries")) == sorted(want)
        for name, (pos, cands, img) in want.items():
            assert np.array_equal(f["summaries/%s/positions" % name], pos.astype(np.int32))
            assert f["summaries/%s/candidates" % name].tolist() == cands and np.array_equal(f["summaries/%s/images" % name].reshape(img.shape), img)
            assert f.info("summaries/%s/contigs" % name)[0] == (len(pos),)
    if os.path.exists("/opt/conda/bin/python3.9"):
        script = ("import h5py, sys, numpy as np\nf = h5py.File(sys.argv[1], 'r')\ng = f['summaries/ctg2_17_18']\n"
                  "assert g['images'].shape == (0, 33, 26) and g['candidates'].shape == (0, 1) and g['contigs'].shape == (0,)\n"
                  "g = f['summaries/ctg0_3_4']\nassert g['contigs'].dtype == np.dtype('S4') and g['contigs'][0] == b'ctg0'\n"
                  "c = g['candidates'][0, 0]\nc = c.decode() if isinstance(c, bytes) else c\nassert c[0] in '123' and g['images'].dtype == np.int8\n"
                  "assert h5py.check_string_dtype(g['candidates'].dtype).length is None and len(f['summaries']) == 300\nprint('fine')\n")
        r = subprocess.run(["/opt/conda/bin/python3.9", "-c", script, big], capture_output=True, text=True)
        assert r.returncode == 0 and "fine" in r.stdout, r.stderr[-3000:]
    # int8 wrap of unclamped columns, as numpy 1.22 did it
    d = dump(ref)
    img = d["summaries/chr20_1000_2000/images"][4]
    assert img.dtype == np.int8 and img[0, 16, 4] == 126 and img[0, 16, 8] == -116
    # dataset view of the reference-written file
    data = SequenceDataset(golden_dir, ref)
    assert len(data) == 8
    contig, pos, depth, cand, freq, image = data[0]
    assert contig == "chr20" and image.shape == (33, 26) and cand[0] == "1A" and freq.shape == (1,)
    batch = SequenceDataset.my_collate([data[i] for i in range(3)])
    assert batch[5].shape == (3, 33, 26) and batch[5].dtype.is_floating_point
    blocks = list(data.batches(5))
    assert [len(b[1]) for b in blocks] == [5, 3]


def test_variant_predictions_writer(golden_dir, tmp_path):
    from pepper_amd.variant.DataStorePredict import DataStore
    from pepper_amd.variant.models.dataloader_predict import SequenceDataset
    ref = os.path.join(golden_dir, "variant_predictions_ref.hdf")
    probs = np.load(os.path.join(golden_dir, "variant_predictions_inputs.npz"))
    data = SequenceDataset(golden_dir, os.path.join(golden_dir, "variant_images_ref.hdf5"), )
    mine = str(tmp_path / "pred.hdf")
    with DataStore(mine, "w") as ds:
        # group order in the images file = name order; batch 0 = first group (5), batch 1 = second (3)
        off = 0
        for b, n in enumerate((5, 3)):
            sl = slice(off, off + n)
            ds.write_prediction(b, [c.decode() for c in data.all_contigs[sl]], data.all_positions[sl],
                                data.all_depths[sl], data.all_candidates[sl], data.all_candidate_frequency[sl],
                                probs[f"probs_{b}"])
            off += n
    assert_same_tree(dump(ref), dump(mine))
    assert dump(mine)["predictions/batch_0/base_prediction"][4].dtype == np.float64


def test_polish_stores(golden_dir, tmp_path):
    from pepper_amd.polish.DataStore import DataStore
    from pepper_amd.polish.DataStorePredict import DataStore as DataStorePredict
    from pepper_amd.polish.models.dataloader_predict import SequenceDataset
    ref = os.path.join(golden_dir, "polish_images_ref.hdf")
    inp = np.load(os.path.join(golden_dir, "polish_images_inputs.npz"))
    mine = str(tmp_path / "img.hdf")
    with DataStore(mine, "w") as ds:
        for cid in range(2):
            name = f"contig_1_1000_2000_{cid}"
            ds.write_summary(("contig_1", 1000, 2000), inp[name + "__image"].tolist(), [0] * 1000,
                             [tuple(p) for p in inp[name + "__position"].tolist()], list(range(1000)), cid, name)
    assert_same_tree(dump(ref), dump(mine))
    data = SequenceDataset(golden_dir, [ref])
    assert len(data) == 2
    contig, start, end, chunk_id, image, position, index = data[1]
    assert contig == "contig_1" and (start, end, chunk_id) == (1000, 2000, 1)
    assert image.shape == (1000, 10) and image.dtype == np.uint8 and tuple(position[950]) == (-1, -1)
    data.close()

    pref = os.path.join(golden_dir, "polish_predictions_ref.hdf")
    pin = np.load(os.path.join(golden_dir, "polish_predictions_inputs.npz"))
    pmine = str(tmp_path / "pred.hdf")
    with DataStorePredict(pmine, "w") as ds:
        for cid in range(2):
            ds.write_prediction("contig_1", np.int64(1000), np.int64(2000), np.int64(cid),
                                np.array([(1000 + i, 0) for i in range(1000)]), np.arange(1000),
                                pin[f"bases_{cid}"], pin[f"phred_{cid}"])
    assert_same_tree(dump(pref), dump(pmine))


@pytest.mark.skipif(not os.path.exists("/opt/conda/bin/python3.9"), reason="needs the image's h5py interpreter")
def test_h5py_sees_same_types(golden_dir, tmp_path):
    """Independent check with real h5py: dtype strings of our files equal the reference's."""
    from pepper_amd.variant.DataStorePredict import DataStore
    mine = str(tmp_path / "pred.hdf")
    with DataStore(mine, "w") as ds:
        ds.write_prediction(0, ["chr20"] * 2, [1, 2], [3, 4], np.array([["1A"], ["2ACG"]], dtype=object),
                            np.array([[5], [6]], np.uint8), np.zeros((2, 3), np.float32))
    script = ("import h5py,sys\n"
              "def d(p):\n"
              "    f=h5py.File(p,'r'); o={}\n"
              "    f.visititems(lambda n,x: o.__setitem__(n.split('/')[-1], (str(x.dtype), x.chunks, x.compression)) "
              "if isinstance(x,h5py.Dataset) else None)\n"
              "    return o\n"
              "a=d(sys.argv[1]); b=d(sys.argv[2])\n"
              "assert all(a[k]==b[k] for k in a), (a,b)\nprint('same')\n")
    r = subprocess.run(["/opt/conda/bin/python3.9", "-c", script, mine,
                        os.path.join(golden_dir, "variant_predictions_ref.hdf")], capture_output=True, text=True)
    assert r.returncode == 0 and "same" in r.stdout, r.stderr


def test_bulk_prediction_writer_and_loader_equal_the_per_dataset_path(tmp_path):
    """predict() writes each batch_<n> group with one library call from bulk arrays and reads image blocks straight into
    one (staging) buffer: same files, same loader contents as the per-dataset / per-object path pinned above."""
    from pepper_amd import synthetic
    from pepper_amd.variant.DataStore import DataStore
    from pepper_amd.variant.DataStorePredict import DataStore as PredStore
    from pepper_amd.variant.models.dataloader_predict import SequenceDataset
    rng = np.random.default_rng(3)
    path = str(tmp_path / "pepper_variants_images_thread_0.hdf5")
    per = [700, 0, 333]
    with DataStore(path, "w") as ds:
        for gi, n in enumerate(per):
            x = synthetic.variant_windows(max(n, 1), seed=50 + gi)[:n]
            cands = np.array([[["1A", "2ACCT", "3AG", "", "1T"][k % 5]] for k in range(n)], dtype=object).reshape(n, 1)
            ds.write_summary("chr20_%d_%d" % (gi * 1000, gi * 1000 + 999), [["chr20", "chr20_KI270_random"][gi % 2]] * n,
                             np.arange(n) + gi * 1000, rng.integers(1, 99, n), cands, rng.integers(0, 120, (n, 1)), x,
                             [0] * n, [0] * n, False)
    asked = []

    def alloc(n, window, features):
        asked.append((n, window, features))
        return np.full((n + 5, window, features), 99, np.int8)[:n]
    d = SequenceDataset(str(tmp_path), path, None, alloc)
    plain = SequenceDataset(str(tmp_path), path)
    assert asked == [(1033, 33, 26)] and len(d) == 1033
    assert np.array_equal(d.all_images, plain.all_images) and np.array_equal(d.all_candidates, plain.all_candidates)
    assert d.all_candidates.shape == (1033, 1) and d.all_candidates[3, 0] == "" and d.all_candidates[701, 0] == "2ACCT"
    assert d[701][3][0] == "2ACCT" and d[0][0] == "chr20" and d[1000][0] == "chr20"
    probs = rng.random((1033, 3)).astype(np.float32)
    old, new = PredStore(str(tmp_path / "old.hdf"), "w"), PredStore(str(tmp_path / "new.hdf"), "w")
    off = 0
    for b, (contigs, positions, depths, candidates, freqs, _) in enumerate(d.batches(256)):
        e = off + len(positions)
        old.write_prediction(b, [c.decode() for c in contigs], positions, depths, candidates, freqs, probs[off:e])
        new.write_prediction_arrays(b, d.all_contigs[off:e], d.all_positions[off:e], d.all_depths[off:e], d.candidate_blob,
                                    d.candidate_offsets[off:e], d.all_candidate_frequency[off:e], probs[off:e])
        off = e
    old.close()
    new.close()
    with h5.File(str(tmp_path / "old.hdf")) as a, h5.File(str(tmp_path / "new.hdf")) as c:
        names = a.keys("predictions")
        assert names == c.keys("predictions") and len(names) == 5
        for nm in names:
            for ds_name in ("contigs", "positions", "depths", "candidates", "candidate_frequency", "base_prediction"):
                p = "predictions/%s/%s" % (nm, ds_name)
                assert a.info(p) == c.info(p), (p, a.info(p), c.info(p))
                assert np.array_equal(a[p], c[p]), p


def test_polish_bulk_block_io_equals_the_per_dataset_path(tmp_path):
    """The polish predict loop reads image chunks and writes predictions a block at a time inside the library."""
    from pepper_amd import synthetic
    from pepper_amd.polish.DataStore import DataStore
    from pepper_amd.polish.DataStorePredict import DataStore as PredStore
    from pepper_amd.polish.models.dataloader_predict import SequenceDataset
    n = 37
    chunks = synthetic.polish_chunks(n, seed=1)
    path = str(tmp_path / "img.hdf")
    with DataStore(path, "w") as ds:
        for c in range(n):
            region = ("ctg1" if c < 20 else "contig_number_two", (c // 3) * 1000, (c // 3) * 1000 + 1200)
            ds.write_summary(region, chunks[c], np.zeros(1000, np.uint8), np.arange(1000) + region[1], np.arange(1000) % 3, c % 3,
                             "%s_%d_%d_%d" % (region[0], region[1], region[2], c % 3))
    d = SequenceDataset(str(tmp_path), [path])
    old_blocks, new_blocks = list(d.batches(16)), list(d.blocks(16, 1000, 10))
    assert [len(b[0]) for b in new_blocks] == [16, 16, 5]
    for (c0, s0, e0, k0, im0, p0, i0), (c1, s1, e1, k1, im1, p1, i1) in zip(old_blocks, new_blocks):
        assert [x if isinstance(x, str) else x.decode() for x in c0] == [x.decode() for x in c1]
        assert list(map(int, s0)) == s1.tolist() and list(map(int, e0)) == e1.tolist() and list(map(int, k0)) == k1.tolist()
        assert np.array_equal(im0, im1) and np.array_equal(np.stack(p0), p1) and np.array_equal(np.stack(i0), i1)
    rng = np.random.default_rng(0)
    labels, phred = rng.integers(0, 5, (n, 1000)).astype(np.uint8), rng.integers(0, 60, (n, 1000)).astype(np.uint8)
    old, new = PredStore(str(tmp_path / "old.hdf"), "w"), PredStore(str(tmp_path / "new.hdf"), "w")
    k = 0
    for (c0, s0, e0, k0, _, p0, i0), (c1, s1, e1, k1, _, p1, i1) in zip(old_blocks, new_blocks):
        m = len(c1)
        for i in range(m):
            old.write_prediction(c0[i], s0[i], e0[i], k0[i], p0[i], i0[i], labels[k + i], phred[k + i])
        new.write_predictions_block(c1, s1, e1, k1, p1, i1, labels[k:k + m], phred[k:k + m])
        new.write_predictions_block(c1[:2], s1[:2], e1[:2], k1[:2], p1[:2], i1[:2], labels[k:k + 2], phred[k:k + 2])   # duplicates are skipped
        k += m
    old.close()
    new.close()

    def walk(f, g):
        out = {}
        for name in f.keys(g):
            p = g + "/" + name
            try:
                kids = f.keys(p)
            except h5.H5Error:
                kids = None
            if kids:
                out.update(walk(f, p))
            else:
                out[p] = (f.info(p), f[p])
        return out
    with h5.File(str(tmp_path / "old.hdf")) as a, h5.File(str(tmp_path / "new.hdf")) as b:
        wa, wb = walk(a, "predictions"), walk(b, "predictions")
        assert wa.keys() == wb.keys() and len(wa) == 4 * n + 2 * 14
        for key in wa:
            assert wa[key][0] == wb[key][0], (key, wa[key][0], wb[key][0])
            assert np.array_equal(wa[key][1], wb[key][1]), key


def test_polish_image_chunks_bulk_writer_equals_write_summary(tmp_path):
    from pepper_amd import synthetic
    from pepper_amd.polish.DataStore import DataStore
    chunks = synthetic.polish_chunks(3, seed=4)
    pos = np.stack([np.stack([np.arange(1000) + 7000 + 950 * c, np.arange(1000) % 4], axis=1) for c in range(3)]).astype(np.int64)
    pos[2, 600:] = -1
    labels = np.zeros((3, 1000), np.uint8)
    region = ("contig_9", 7000, 8200)
    with DataStore(str(tmp_path / "a.hdf"), "w") as a, DataStore(str(tmp_path / "b.hdf"), "w") as b:
        for c in range(3):
            a.write_summary(region, chunks[c], labels[c], pos[c][:, 0], pos[c][:, 1], c, "contig_9_7000_8200_%d" % c)
        b.write_summaries(region, [chunks[c] for c in range(3)], list(labels), list(pos), [0, 1, 2])
        b.write_summaries(region, [chunks[0]], [labels[0]], [pos[0]], [0])                 # already written: skipped
    with h5.File(str(tmp_path / "a.hdf")) as fa, h5.File(str(tmp_path / "b.hdf")) as fb:
        assert fa.keys("summaries") == fb.keys("summaries") and len(fb.keys("summaries")) == 3
        for name in fa.keys("summaries"):
            assert sorted(fa.keys("summaries/" + name)) == sorted(fb.keys("summaries/" + name))
            for ds_name in fa.keys("summaries/" + name):
                p = "summaries/%s/%s" % (name, ds_name)
                assert fa.info(p) == fb.info(p), (p, fa.info(p), fb.info(p))
                assert np.array_equal(fa[p], fb[p]), p


def _chunks_by_dataset(f, names):
    return (np.stack([f["summaries/%s/image" % n] for n in names]), np.stack([f["summaries/%s/position" % n] for n in names]),
            np.stack([f["summaries/%s/index" % n] for n in names]))


def test_direct_chunk_reads_equal_the_library(tmp_path):
    """read_polish_chunks copies image / position / index straight out of the mapped file where hdf5io.cpp's locator knows
    the format (classic files of h5py and of mode "w"); the bytes are those libhdf5 returns dataset by dataset -- over a
    group tree three B-tree levels deep, and with the library taking over where the
    locator says it does not know (1.10-format file)."""
    n, seq, feat = 2600, 16, 10
    rng = np.random.default_rng(5)
    images = rng.integers(0, 255, (n, seq, feat), dtype=np.uint8)
    position = rng.integers(-1, 1 << 40, (n, seq, 2)).astype(np.int64)
    labels = np.zeros((n, seq), np.uint8)
    # names that sort differently as strings and as numbers, long and short contigs, several regions
    names, regions = [], []
    for i in range(n):
        contig = ("c%d" % (i % 7)) if i % 3 else "a_rather_long_contig_name_with_underscores_%d" % (i % 5)
        regions.append((contig, (i // 40) * 1000, (i // 40) * 1000 + 1200, i % 40))
        names.append("%s_%d_%d_%d" % regions[-1])
    assert len(set(names)) == n
    path = str(tmp_path / "tree.hdf")
    with h5.File(path, "w") as f:
        for i in range(n):
            c, s, e, k = regions[i]
            f.write_polish_image_chunks([names[i]], c, s, e, np.array([k], np.int64), images[i:i + 1], labels[i:i + 1],
                                        np.ascontiguousarray(position[i:i + 1, :, 0]), np.ascontiguousarray(position[i:i + 1, :, 1]))
    order = rng.permutation(n).tolist()
    asked = [names[i] for i in order]
    with h5.File(path) as f:
        contigs, start, end, chunk, im, pos, idx = f.read_polish_chunks(asked, seq, feat)
        assert f.read_stats() == (n, 0)
        a, b, c = _chunks_by_dataset(f, asked[:200])
    assert np.array_equal(im, images[order]) and np.array_equal(pos, position[order, :, 0]) and np.array_equal(idx, position[order, :, 1])
    assert np.array_equal(a, im[:200]) and np.array_equal(b, pos[:200]) and np.array_equal(c, idx[:200])
    assert [x.decode() for x in contigs] == [regions[i][0] for i in order]
    assert start.tolist() == [regions[i][1] for i in order] and chunk.tolist() == [regions[i][3] for i in order]
    with h5.File(path) as f, pytest.raises(h5.H5Error):
        f.read_polish_chunks(["c1_0_1200_39x"], seq, feat)                   # an absent group is still an error
    with h5.File(path) as f, pytest.raises(h5.H5Error):
        f.read_polish_chunks(asked[:2], seq + 1, feat)                       # a wrong shape too

    new = str(tmp_path / "v110.hdf")                                         # links in the object header, no symbol table
    with h5.File(new, "w-new") as f:
        f.write_polish_image_chunks(["x_0_1_%d" % k for k in range(5)], "x", 0, 1, np.arange(5, dtype=np.int64), images[:5], labels[:5],
                                    np.ascontiguousarray(position[:5, :, 0]), np.ascontiguousarray(position[:5, :, 1]))
    with h5.File(new) as f:
        got = f.read_polish_chunks(["x_0_1_%d" % k for k in range(5)], seq, feat)
        assert f.read_stats() == (0, 5)
        assert np.array_equal(got[4], images[:5]) and np.array_equal(got[5], position[:5, :, 0])


@pytest.mark.skipif(not os.path.exists("/opt/conda/bin/python3.9"), reason="needs the image's h5py interpreter")
def test_direct_chunk_reads_leave_unknown_layouts_to_the_library(tmp_path):
    """Files the locator must not guess about: compressed (chunked) datasets, big-endian or narrower integers, a user block
    in front of the superblock, the latest object formats.  Each is read through libhdf5 with the same values."""
    script = (
        "import h5py, numpy as np, sys\n"
        "rng = np.random.default_rng(3)\n"
        "im = rng.integers(0, 255, (4, 16, 10), dtype=np.uint8); pos = rng.integers(0, 1 << 33, (4, 16)); idx = pos % 5\n"
        "np.savez(sys.argv[1] + '/want.npz', im=im, pos=pos, idx=idx)\n"
        "def fill(f, kind):\n"
        "    for i in range(4):\n"
        "        g = f.create_group('summaries/c_0_10_%d' % i)\n"
        "        if kind == 'gzip': g.create_dataset('image', data=im[i], compression='gzip')\n"
        "        else: g['image'] = im[i]\n"
        "        g['position'] = pos[i].astype('>i8') if kind == 'be' else pos[i]\n"
        "        g['index'] = idx[i].astype('i4') if kind == 'narrow' else idx[i]\n"
        "        g['contig'] = 'c'; g['region_start'] = 0; g['region_end'] = 10; g['chunk_id'] = i\n"
        "for kind in ('plain', 'gzip', 'be', 'narrow'):\n"
        "    with h5py.File(sys.argv[1] + '/' + kind + '.hdf', 'w') as f: fill(f, kind)\n"
        "with h5py.File(sys.argv[1] + '/userblock.hdf', 'w', userblock_size=512) as f: fill(f, 'plain')\n"
        "with h5py.File(sys.argv[1] + '/latest.hdf', 'w', libver='latest') as f: fill(f, 'plain')\n")
    r = subprocess.run(["/opt/conda/bin/python3.9", "-c", script, str(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    want = np.load(str(tmp_path / "want.npz"))
    names = ["c_0_10_%d" % i for i in range(4)]
    for kind, stats in (("plain", (4, 0)), ("gzip", (0, 4)), ("be", (0, 4)), ("narrow", (0, 4)), ("userblock", (0, 4)),
                        ("latest", (0, 4))):
        with h5.File(str(tmp_path / (kind + ".hdf"))) as f:
            got = f.read_polish_chunks(names, 16, 10)
            assert f.read_stats() == stats, (kind, f.read_stats())
        assert np.array_equal(got[4], want["im"]) and np.array_equal(got[5], want["pos"]) and np.array_equal(got[6], want["idx"]), kind
        assert got[3].tolist() == [0, 1, 2, 3]


def _prediction_file(path, n, rng, builder):
    """n chunks over three contigs (one with many regions: several B-tree levels), written a block at a time."""
    from pepper_amd.polish.DataStorePredict import DataStore as PredStore
    os.environ["PEPPER_AMD_H5_BUILDER"] = "1" if builder else "0"
    try:
        store = PredStore(path, "w")
    finally:
        del os.environ["PEPPER_AMD_H5_BUILDER"]
    assert isinstance(store.file_handler, h5.PredictionBuilder) == builder
    seq = 1000
    contigs = np.array([b"chr1" if i % 5 else (b"contig_two" if i % 10 else b"z") for i in range(n)], dtype="S256")
    start = (np.arange(n) // 3) * 1000
    end = start + 1200
    chunk = np.arange(n) % 3
    position = rng.integers(0, 1 << 40, (n, seq)).astype(np.int64)
    index = rng.integers(0, 4, (n, seq)).astype(np.int64)
    bases = rng.integers(0, 5, (n, seq)).astype(np.uint8)
    phred = rng.integers(0, 60, (n, seq)).astype(np.uint8)
    for a in range(0, n, 700):
        b = min(n, a + 700)
        store.write_predictions_block(contigs[a:b], start[a:b], end[a:b], chunk[a:b], position[a:b], index[a:b], bases[a:b], phred[a:b])
    store.write_predictions_block(contigs[:3], start[:3], end[:3], chunk[:3], position[:3], index[:3], bases[:3], phred[:3])   # duplicates
    store.close()
    return contigs, start, end, chunk, position, index, bases, phred


def test_prediction_builder_writes_the_file_libhdf5_writes(tmp_path):
    """pepper_amd/csrc/h5build.cpp lays a prediction file out itself (rows appended, metadata at close).  libhdf5 must find
    in it exactly what it finds in the file it wrote itself from the same calls: every group, dataset, shape, type and value
    -- here with 1 100 regions under one contig (three B-tree levels), names that sort differently as text and as numbers,
    duplicates skipped; h5diff (the HDF5 tools' own comparison) agrees where the tools are installed."""
    n = 4000
    new, old = str(tmp_path / "builder.hdf"), str(tmp_path / "library.hdf")
    want = _prediction_file(new, n, np.random.default_rng(11), True)
    _prediction_file(old, n, np.random.default_rng(11), False)

    def walk(f, g):
        out = {}
        for name in f.keys(g):
            p = g + "/" + name
            try:
                kids = f.keys(p)
            except h5.H5Error:
                kids = None
            if kids:
                out.update(walk(f, p))
            else:
                out[p] = (f.info(p), f[p])
        return out
    with h5.File(new) as a, h5.File(old) as b:
        assert a.keys("/") == ["predictions"] and sorted(a.keys("predictions")) == ["chr1", "contig_two", "z"]
        wa, wb = walk(a, "predictions"), walk(b, "predictions")
        assert wa.keys() == wb.keys() and len(wa) == 4 * n + 2 * len(set(zip(want[0].tolist(), want[1].tolist())))
        for key in wa:
            assert wa[key][0] == wb[key][0], (key, wa[key][0], wb[key][0])
            assert np.array_equal(wa[key][1], wb[key][1]), key
        contigs, start, end, chunk, position, index, bases, phred = want
        for i in (0, 1, 2, 1234, n - 1):
            base = "predictions/%s/%s-%d-%d/%d/" % (contigs[i].decode(), contigs[i].decode(), start[i], end[i], chunk[i])
            assert np.array_equal(a[base + "position"], position[i]) and np.array_equal(a[base + "phred_score"], phred[i])
        got = a.read_polish_prediction_region("predictions/chr1/chr1-1000-2200", 1000)
        assert len(got[0]) == 2                                  # chunks 1 and 2 of that region are chr1's (3 is 'contig_two')
    if os.path.exists("/opt/conda/bin/h5diff"):
        r = subprocess.run(["/opt/conda/bin/h5diff", old, new], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    with pytest.raises(h5.H5Error):
        with h5.PredictionBuilder(str(tmp_path / "dup.hdf")) as f:
            f["a/b"] = np.arange(3)
            f["a/b"] = np.arange(3)
    # a path with an empty component is an error (libhdf5 would collapse it; a group without a name must not get into a file)
    with h5.PredictionBuilder(str(tmp_path / "paths.hdf")) as b:
        for bad in ("a//b", "a/b/", "/"):
            with pytest.raises(h5.H5Error, match="bad dataset path"):
                b[bad] = np.arange(3)
            with pytest.raises(h5.H5Error, match="bad dataset path"):
                b[bad] = "text"
        b["/a/b"] = np.arange(3)
    with h5.File(str(tmp_path / "paths.hdf")) as f:
        assert f.keys("/") == ["a"] and f.keys("a") == ["b"] and f["a/b"].tolist() == [0, 1, 2]
    with h5.PredictionBuilder(str(tmp_path / "empty.hdf")):
        pass
    with h5.File(str(tmp_path / "empty.hdf")) as f:
        assert f.keys("/") == []


@pytest.mark.skipif(not os.path.exists("/opt/conda/bin/python3.9"), reason="needs the image's h5py interpreter")
def test_h5py_reads_the_builder_file(tmp_path):
    path = str(tmp_path / "builder.hdf")
    want = _prediction_file(path, 300, np.random.default_rng(12), True)
    np.savez(str(tmp_path / "want.npz"), position=want[4], bases=want[6])
    script = ("import h5py, numpy as np, sys\n"
              "w = np.load(sys.argv[2])\n"
              "f = h5py.File(sys.argv[1], 'r')\n"
              "seen = []\n"
              "f.visititems(lambda n, x: seen.append(n) if isinstance(x, h5py.Dataset) else None)\n"
              "assert len(seen) == 4 * 300 + 2 * 160, len(seen)\n"
              "g = f['predictions/z/z-0-1200']\n"
              "assert g['contig_start'][()] == 0 and g['contig_end'][()] == 1200 and g['contig_end'].dtype == np.int64\n"
              "assert g['contig_end'].shape == () and sorted(g.keys()) == ['0', 'contig_end', 'contig_start']\n"
              "assert np.array_equal(g['0/position'][()], w['position'][0]) and g['0/bases'].dtype == np.uint8\n"
              "assert np.array_equal(f['predictions/chr1/chr1-99000-100200/2/bases'][()], w['bases'][299])\n"
              "print('fine')\n")
    r = subprocess.run(["/opt/conda/bin/python3.9", "-c", script, path, str(tmp_path / "want.npz")], capture_output=True, text=True)
    assert r.returncode == 0 and "fine" in r.stdout, r.stderr[-3000:]


def test_builder_image_files_equal_the_library_files(tmp_path, monkeypatch):
    """Polish image files laid out by h5build.cpp (variable-length contig strings in a global heap collection, three scalars
    and four arrays per chunk) against the files libhdf5 writes from the same calls: the same tree for libhdf5, no difference
    for h5diff, the same blocks for the chunk reader (which takes the large datasets straight from the mapped file and the
    first and last chunk's small ones -- the strings included -- through libhdf5), the same str for h5py."""
    from pepper_amd import synthetic
    from pepper_amd.polish.DataStore import DataStore
    from pepper_amd.polish.models.dataloader_predict import SequenceDataset
    n = 700                                                   # > 126 distinct contig strings: several heap collections
    chunks = synthetic.polish_chunks(8, seed=2)
    paths = {}
    for builder in (True, False):
        monkeypatch.setenv("PEPPER_AMD_H5_BUILDER", "1" if builder else "0")
        path = paths[builder] = str(tmp_path / ("img_%d.hdf" % builder))
        with DataStore(path, "w") as ds:
            assert isinstance(ds.file_handler, h5.PredictionBuilder) == builder
            for r in range(n // 2):
                contig = "contig_%d_with_a_longer_name" % (r % 300)
                region = (contig, r * 1000, r * 1000 + 1200)
                pos = np.stack([np.stack([np.arange(1000) + region[1] + 950 * c, np.arange(1000) % 3], axis=1) for c in range(2)])
                ds.write_summaries(region, chunks[(2 * r) % 8:(2 * r) % 8 + 2], np.zeros((2, 1000), np.uint8), pos, [0, 1])
            ds.write_summary(("häßlich", 5, 6), chunks[0].tolist(), [0] * 1000, [(i, 0) for i in range(1000)], list(range(1000)), 7, "x_5_6_7")
    assert_same_tree(dump(paths[True]), dump(paths[False]))
    if os.path.exists("/opt/conda/bin/h5diff"):
        r = subprocess.run(["/opt/conda/bin/h5diff", paths[False], paths[True]], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    blocks = {}
    for builder, path in paths.items():
        with h5.File(path) as f:
            names = [k for k in f.keys("summaries") if k != "x_5_6_7"]
            blocks[builder] = f.read_polish_chunks(names, 1000, 10)
            assert f.read_stats() == (len(names), 0)
    for a, b in zip(blocks[True], blocks[False]):
        assert np.array_equal(a, b)
    assert blocks[True][0][0].decode().startswith("contig_") and len(blocks[True][0]) == n
    # a name used twice in a group of very many chunks is found when the group's B-tree is laid out: close() fails
    img, lab, pos = np.zeros((1, 4, 10), np.uint8), np.zeros((1, 4), np.uint8), np.zeros((1, 4), np.int64)
    dup = h5.PredictionBuilder(str(tmp_path / "dup.hdf"))
    for k in list(range(100)) + [7]:
        dup.write_polish_image_chunks(["c_%d" % k], "c", 0, 1, np.array([0], np.int64), img, lab, pos, pos)
    with pytest.raises(h5.H5Error):
        dup.close()
    if os.path.exists("/opt/conda/bin/python3.9"):
        script = ("import h5py, sys\nf = h5py.File(sys.argv[1], 'r')\ng = f['summaries/x_5_6_7']\n"
                  "c = g['contig'][()]\nc = c.decode('utf-8') if isinstance(c, bytes) else c\n"
                  "assert c == 'h\\u00e4\\u00dflich', repr(c)\nassert g['position'].shape == (1000, 2) and g['chunk_id'][()] == 7\n"
                  "assert h5py.check_string_dtype(g['contig'].dtype).encoding == 'utf-8'\n"
                  "k = sorted(f['summaries'])[3]\nassert f['summaries'][k]['image'].shape == (1000, 10)\nprint('fine')\n")
        r = subprocess.run(["/opt/conda/bin/python3.9", "-c", script, paths[True]], capture_output=True, text=True)
        assert r.returncode == 0 and "fine" in r.stdout, r.stderr[-3000:]


def test_direct_chunk_reads_survive_damaged_metadata(tmp_path):
    """The locator walks file offsets it read from the file: on a damaged file it must answer "don't know" (and leave the
    chunk to libhdf5, which then reports the damage) or read what is there -- never run outside the mapping.  Random 8-byte
    overwrites all over a small image file, in a child process so that a crash would be seen as one."""
    n, seq, feat = 300, 64, 10
    rng = np.random.default_rng(0)
    im = rng.integers(0, 255, (n, seq, feat), dtype=np.uint8)
    pos = np.tile(np.arange(seq, dtype=np.int64), (n, 1))
    src = str(tmp_path / "src.hdf")
    with h5.File(src, "w") as f:
        for b in range(0, n, 50):
            f.write_polish_image_chunks(["ctg_%d_%d_%d" % (b, b + 1, k) for k in range(50)], "ctg", b, b + 1, np.arange(50, dtype=np.int64),
                                        im[b:b + 50], np.zeros((50, seq), np.uint8), pos[b:b + 50], pos[b:b + 50] % 3)
    script = (
        "import sys, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "from pepper_amd import h5\n"
        "raw = open(sys.argv[1], 'rb').read()\n"
        "with h5.File(sys.argv[1]) as f: names = f.keys('summaries')\n"
        "outcomes = []\n"
        "for seed in range(12):\n"
        "    rng = np.random.default_rng(seed)\n"
        "    bad = bytearray(raw)\n"
        "    for _ in range(400):\n"
        "        at = int(rng.integers(96, len(bad) - 8))\n"
        "        bad[at:at + 8] = rng.integers(0, 256, 8, dtype=np.uint8).tobytes()\n"
        "    path = sys.argv[1] + '.bad'\n"
        "    open(path, 'wb').write(bad)\n"
        "    try:\n"
        "        with h5.File(path) as f:\n"
        "            f.read_polish_chunks(names, %d, %d)\n"
        "        outcomes.append('read')\n"
        "    except h5.H5Error:\n"
        "        outcomes.append('error')\n"
        "print('survived', outcomes.count('read'), outcomes.count('error'))\n" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), seq, feat))
    r = subprocess.run([sys.executable, "-c", script, src], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "survived" in r.stdout, (r.returncode, r.stderr[-2000:])


def test_packed_summary_write_equals_the_per_candidate_path(tmp_path):
    """DataStore.write_summary_packed (the encoder's candidate strings handed to the writer as one block) writes the file
    write_summary writes from the same interval's lists, byte for byte -- wrapped depths and frequencies included."""
    import filecmp
    from pepper_amd.variant.DataStore import DataStore
    rng = np.random.default_rng(5)
    outs = []
    for n in (0, 1, 257):
        cands = ["1A" if k % 3 else "2ACGTT"[:2 + k % 5] for k in range(n)]
        raw = ("\0".join(cands) + "\0").encode() if n else b""
        ends = np.flatnonzero(np.frombuffer(raw, np.uint8) == 0).astype(np.int64) + 1
        outs.append(dict(positions=np.arange(n, dtype=np.int64) + 7, depths=rng.integers(0, 400, n).astype(np.int32),
                         candidate_frequency=rng.integers(0, 400, n).astype(np.int32),
                         images=rng.integers(-128, 128, (n, 33, 26)).astype(np.int8), candidates=cands, candidates_blob=raw,
                         candidates_offsets=np.concatenate([[0], ends]).astype(np.int64)))
    a, b = str(tmp_path / "a.hdf5"), str(tmp_path / "b.hdf5")
    with DataStore(a, "w") as f:
        for i, out in enumerate(outs):
            assert f.write_summary_packed("chr_%d_%d" % (i, i + 1), "chr", out)
    with DataStore(b, "w") as f:
        for i, out in enumerate(outs):
            n = len(out["positions"])
            f.write_summary("chr_%d_%d" % (i, i + 1), ["chr"] * n, out["positions"], out["depths"],
                            np.array(out["candidates"], dtype=object).reshape(n, 1), out["candidate_frequency"].reshape(n, 1),
                            out["images"], [0] * n, [0] * n, False)
    assert filecmp.cmp(a, b, shallow=False)
