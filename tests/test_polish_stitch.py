"""Polish stitch: the vectorised merge against a literal dictionary
restatement of /root/reference/pepper/modules/python/Stitch.py:36-128 written here as the checker."""
import os

import numpy as np

from pepper_amd import h5
from pepper_amd.polish.DataStorePredict import DataStore
from pepper_amd.polish.perform_stitch import perform_stitch

DECODE = {1: 'A', 2: 'C', 3: 'G', 4: 'T', 0: ''}


def dict_stitch(pred_files, contig, threads=1):
    """Literal restatement: per piece a dict keyed (pos, idx), later writes overwrite."""
    keys = []
    for fn in pred_files:
        with h5.File(fn) as f:
            if contig not in f.keys('predictions'):
                continue
            for ck in sorted(f.keys('predictions/' + contig)):
                keys.append((fn, ck, int(f[f'predictions/{contig}/{ck}/contig_start']), int(f[f'predictions/{contig}/{ck}/contig_end'])))
    keys = sorted(sorted(keys, key=lambda e: e[1]), key=lambda e: (e[2], e[3]))
    size = max(2, int(len(keys) / threads) + 1)
    pieces = []
    for i in range(0, len(keys), size):
        table = {}
        for fn, ck, st, en in keys[i:i + size]:
            with h5.File(fn) as f:
                for cid in sorted(set(f.keys(f'predictions/{contig}/{ck}')) - {'contig_start', 'contig_end'}):
                    base = f'predictions/{contig}/{ck}/{cid}/'
                    for pos, idx, b in zip(f[base + 'position'].tolist(), f[base + 'index'].tolist(), f[base + 'bases'].tolist()):
                        if st > 0 and pos <= st + 200:
                            continue
                        if idx < 0 or pos < 0:
                            continue
                        table[(pos, idx)] = b
        if table:
            order = sorted(table)
            pieces.append((order[0][0], order[-1][0], ''.join(DECODE[table[k]] for k in order)))
    return ''.join(p[2] for p in sorted(pieces, key=lambda e: (e[0], e[1])))


def make_region(rng, store, contig, start, end, n_chunks):
    """Rows (position, insert index) walking the region with occasional insert columns; chunks of 1000
    rows overlapping by 50, the last one padded with (-1, -1)."""
    pos, idx = [], []
    p = start
    while p < end:
        pos.append(p)
        idx.append(0)
        for k in range(int(rng.integers(0, 3)) if rng.random() < 0.15 else 0):
            pos.append(p)
            idx.append(k + 1)
        p += 1
    pos, idx = np.array(pos), np.array(idx)
    cid, at = 0, 0
    while at < len(pos) and cid < n_chunks:
        cp, ci = pos[at:at + 1000], idx[at:at + 1000]
        pad = 1000 - len(cp)
        cp = np.concatenate([cp, -np.ones(pad, dtype=cp.dtype)])
        ci = np.concatenate([ci, -np.ones(pad, dtype=ci.dtype)])
        bases = rng.integers(0, 5, size=1000)
        store.write_prediction(contig, start, end, cid, cp, ci, bases, rng.integers(0, 60, size=1000))
        if at + 1000 >= len(pos):
            break
        at += 950
        cid += 1


def test_perform_stitch_matches_dictionary_merge(tmp_path):
    rng = np.random.default_rng(9)
    pred = tmp_path / "pred"
    pred.mkdir()
    files = [str(pred / "pepper_prediction_0.hdf"), str(pred / "pepper_prediction_1.hdf")]
    with DataStore(files[0], "w") as a, DataStore(files[1], "w") as b:
        make_region(rng, a, "contig_2", 0, 9000, 12)            # > 10 chunks: ids sort as strings
        make_region(rng, b, "contig_2", 8900, 12000, 12)        # overlapping neighbour region, other file
        make_region(rng, a, "contig_2", 11900, 12700, 12)
        make_region(rng, b, "contig_10", 0, 700, 12)
        make_region(rng, a, "contig_1", 500, 1800, 12)           # region not starting at 0: head dropped
    (pred / "readme.txt").write_text("ignored")
    for threads in (1, 2):
        out = perform_stitch(str(pred), str(tmp_path / f"out{threads}" / "asm"), threads)
        assert out.endswith("asm_pepper_polished.fa")
        lines = open(out).read().splitlines()
        assert lines[0::2] == [">contig_1", ">contig_2", ">contig_10"]       # natural order
        for name, seq in zip(lines[0::2], lines[1::2]):
            want = dict_stitch(files, name[1:], threads)
            assert len(seq) > 300 and seq == want, name


def test_stitch_empty_and_all_gap(tmp_path):
    pred = tmp_path / "pred"
    pred.mkdir()
    with DataStore(str(pred / "p.hdf"), "w") as s:
        s.write_prediction("c1", 0, 1000, 0, np.arange(1000), np.zeros(1000, dtype=np.int64), np.zeros(1000), np.zeros(1000))
        s.write_prediction("c2", 0, 10, 0, -np.ones(1000, dtype=np.int64), -np.ones(1000, dtype=np.int64),
                           np.ones(1000), np.zeros(1000))
    out = perform_stitch(str(pred), str(tmp_path / "o"), 1)
    assert open(out).read() == ""             # all-gap consensus and padding-only chunks write nothing


def test_stitch_matches_reference_golden(golden_dir, tmp_path):
    """tests/golden/polish_stitch_ref.fa was written by the REFERENCE's perform_stitch (make_golden_stitch.py) from
    prediction files holding the arrays of polish_stitch_inputs.npz; the same arrays through pepper_amd's
    DataStore + perform_stitch must give the same FASTA, byte for byte."""
    g = np.load(os.path.join(golden_dir, "polish_stitch_inputs.npz"), allow_pickle=False)
    pred = tmp_path / "pred"
    pred.mkdir()
    stores = [DataStore(str(pred / ("pepper_prediction_%d.hdf" % i)), "w") for i in range(2)]
    for ri in range(int(g["n_regions"])):
        fi, start, end, n_chunks = (int(v) for v in g["r%d_meta" % ri])
        contig = str(g["r%d_contig" % ri])
        for cid in range(n_chunks):
            stores[fi].write_prediction(contig, start, end, cid, g["r%d_c%d_position" % (ri, cid)],
                                        g["r%d_c%d_index" % (ri, cid)], g["r%d_c%d_bases" % (ri, cid)],
                                        g["r%d_c%d_phred" % (ri, cid)])
    for s in stores:
        s.close()
    want = open(os.path.join(golden_dir, "polish_stitch_ref.fa")).read()
    for threads in (1, 2):
        out = perform_stitch(str(pred), str(tmp_path / ("o%d" % threads) / "asm"), threads)
        assert open(out).read() == want


def test_native_merge_on_rows_no_pipeline_would_write(tmp_path, monkeypatch):
    """pa_h5_stitch_polish_regions merges chunk by chunk into the tail of the piece and relies on nothing: rows out of order
    inside a chunk, the same key twice in a chunk, in two chunks, in two regions and in two files (the last write in the
    reference's loop order wins), chunks of other lengths, a region that lands before everything merged so far -- against
    the dictionary restatement and the numpy form; files written through libhdf5 (1.10 object formats: the locator does not
    know them) read the same; a label that is not a base raises KeyError as label_decoder does."""
    import pytest
    from pepper_amd.polish import Stitch
    rng = np.random.default_rng(21)

    def write(builder):
        monkeypatch.setenv("PEPPER_AMD_H5_BUILDER", "1" if builder else "0")
        tag = "b" if builder else "l"
        pred = tmp_path / ("pred_" + tag)
        pred.mkdir()
        files = [str(pred / "p0.hdf"), str(pred / "p1.hdf")]
        r = np.random.default_rng(22)
        with DataStore(files[0], "w") as a, DataStore(files[1], "w") as b:
            for store, start, end in ((a, 3000, 4000), (b, 0, 2500), (a, 2400, 3300), (b, 3000, 4000)):   # (3000, 4000) in both files
                for cid in range(3):
                    n = int(r.integers(5, 400)) if cid == 1 else 1000                                    # an odd-length chunk
                    pos = r.integers(start, end + 300, n)
                    pos[r.random(n) < 0.05] = -1
                    idx = r.integers(-1, 3, n)
                    if cid != 2:
                        order = np.lexsort((idx, pos))
                        pos, idx = pos[order], idx[order]                                               # chunk 2 stays unsorted
                    store.write_prediction("ctg", start, end, cid, pos, idx, r.integers(0, 5, n), np.zeros(n))
        return files
    want = None
    for builder in (True, False):
        files = write(builder)
        pred = os.path.dirname(files[0])
        for threads in (1, 3):
            expect = dict_stitch(files, "ctg", threads)
            monkeypatch.delenv("PEPPER_AMD_STITCH_NUMPY", raising=False)
            out = perform_stitch(pred, str(tmp_path / ("n%d%d" % (builder, threads))), threads)
            got = open(out).read().splitlines()[1]
            monkeypatch.setenv("PEPPER_AMD_STITCH_NUMPY", "1")
            out = perform_stitch(pred, str(tmp_path / ("p%d%d" % (builder, threads))), threads)
            assert got == expect and open(out).read().splitlines()[1] == expect and len(got) > 1000
        with h5.File(files[0]) as f:
            regions = f.list_polish_regions("ctg")
            assert [r[0] for r in regions] == sorted(f.keys("predictions/ctg")) and (regions[0][1], regions[0][2]) == (2400, 3300)
        want = want or expect
        assert expect == want                              # the builder's file and libhdf5's hold the same predictions
    monkeypatch.delenv("PEPPER_AMD_STITCH_NUMPY", raising=False)
    bad = str(tmp_path / "bad.hdf")
    with DataStore(bad, "w") as s:
        s.write_prediction("c", 0, 10, 0, np.arange(10), np.zeros(10, np.int64), np.array([1, 2, 3, 4, 0, 7, 1, 1, 1, 1]), np.zeros(10))
    with pytest.raises(KeyError):
        Stitch.small_chunk_stitch("c", [(bad, "c", 0, 10)])
    assert Stitch.small_chunk_stitch("c", []) == (-1, -1, "")


def test_native_merge_takes_chunk_ids_in_string_order(tmp_path, monkeypatch):
    """A region of twelve chunks: the reference iterates sorted(chunk ids) on STRINGS ("0", "1", "10", "11", "2", ...), and with
    keys written by several chunks that order decides which label stays; the native merge goes chunk by chunk in that order."""
    monkeypatch.delenv("PEPPER_AMD_STITCH_NUMPY", raising=False)
    rng = np.random.default_rng(31)
    pred = tmp_path / "pred"
    pred.mkdir()
    path = str(pred / "p.hdf")
    with DataStore(path, "w") as store:
        for start, end in ((0, 900), (700, 1600)):
            for cid in range(12):
                n = 150
                pos = np.sort(rng.integers(start, end, n))
                idx = rng.integers(0, 2, n)
                order = np.lexsort((idx, pos))
                store.write_prediction("ctg", start, end, cid, pos[order], idx[order], rng.integers(1, 5, n), np.zeros(n))
    expect = dict_stitch([path], "ctg", 1)
    out = perform_stitch(str(pred), str(tmp_path / "o"), 1)
    assert open(out).read().splitlines()[1] == expect and len(expect) > 500
