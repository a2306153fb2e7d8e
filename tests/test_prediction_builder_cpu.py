"""predictions/batch_<n> groups written by the append-only writer (pa_h5_builder_write_prediction_batch, no libhdf5) against the
same groups written through libhdf5 (pa_h5_write_prediction_batch): both files read back with libhdf5, dataset by dataset --
names, shapes, classes, element sizes, values (float64 probabilities included: the first floating-point datatype the writer
lays out) -- and through the calls the candidate finder makes.  Layout: /root/reference/pepper_variant/modules/python/
DataStorePredict.py:26-67."""
import numpy as np

from pepper_amd import h5
from pepper_amd.variant.DataStorePredict import DataStore


def _batch(rng, n, contig):
    positions = np.sort(rng.integers(0, 5_000_000, n)).astype(np.int32)
    depths = rng.integers(1, 90, n).astype(np.uint8)
    codes = [("1" + "ACGT"[int(rng.integers(4))]) if k % 3 else ("2" + "".join("ACGT"[int(x)] for x in rng.integers(0, 4, int(rng.integers(1, 40)))))
             for k in range(n)]
    blob = "".join(c + "\0" for c in codes).encode()
    offsets = np.concatenate([[0], np.cumsum([len(c) + 1 for c in codes])]).astype(np.int64)[:n]
    freqs = rng.integers(0, 60, (n, 1)).astype(np.uint8)
    probs = rng.random((n, 3)).astype(np.float32)
    contigs = np.array([contig] * n, dtype='S')
    return contigs, positions, depths, np.frombuffer(blob + b"\0", np.uint8), offsets, freqs, probs, codes


def test_builder_batches_equal_libhdf5_batches(tmp_path):
    rng = np.random.default_rng(4)
    batches = [_batch(rng, 512, "chr1"), _batch(rng, 512, "chr10_KI270_random"), _batch(rng, 37, "c"), _batch(rng, 1, "chrX")]
    mixed = _batch(rng, 6, "chr2")
    mixed = (np.array(["chr2", "chr2", "chr21_alt", "chr2", "a", "chr2"], dtype='S'),) + mixed[1:]       # several names in one batch
    batches.append(mixed)
    paths = {False: str(tmp_path / "lib.hdf"), True: str(tmp_path / "builder.hdf")}
    for bulk, path in paths.items():
        store = DataStore(path, mode='w', bulk=bulk)
        assert isinstance(store.file_handler, h5.PredictionBuilder) == bulk
        for k, b in enumerate(batches):
            store.write_prediction_arrays(k, *b[:7])
        store.write_prediction_arrays(0, *batches[1][:7])            # a batch number met twice is written once, in both
        store.close()
    with h5.File(paths[False]) as a, h5.File(paths[True]) as b:
        assert sorted(a.keys("predictions")) == sorted(b.keys("predictions")) == sorted("batch_%d" % k for k in range(len(batches)))
        for k, batch in enumerate(batches):
            base = "predictions/batch_%d/" % k
            assert sorted(a.keys(base[:-1])) == sorted(b.keys(base[:-1]))
            for name in ("contigs", "positions", "depths", "candidates", "candidate_frequency", "base_prediction"):
                assert a.info(base + name) == b.info(base + name), (k, name, a.info(base + name), b.info(base + name))
            for name in ("positions", "depths", "candidate_frequency", "base_prediction"):
                x, y = np.asarray(a[base + name]), np.asarray(b[base + name])
                assert x.dtype == y.dtype and x.shape == y.shape and np.array_equal(x, y), (k, name)
            assert np.asarray(b[base + "base_prediction"]).dtype == np.float64
            assert np.array_equal(np.asarray(b[base + "base_prediction"]), batch[6].astype(np.float64))
            assert a.read_strings_shaped(base + "contigs") == b.read_strings_shaped(base + "contigs")
            assert a.read_strings_shaped(base + "candidates") == b.read_strings_shaped(base + "candidates")
            shape, blob = b.read_strings_shaped(base + "candidates")
            assert tuple(shape) == (len(batch[7]), 1) and blob == "".join(c + "\0" for c in batch[7]).encode()
            assert [c.decode() for c in np.asarray(b[base + "contigs"]).tolist()] == [c.decode() for c in batch[0].tolist()]


def test_a_batch_read_in_one_call_equals_the_six_reads(tmp_path):
    """File.read_prediction_batch (pa_h5_prediction_batch_load: the locator, variable-length strings out of the global heap
    collections, float64 rows) against the dataset-by-dataset reads of libhdf5 -- on the append-only writer's file (where it must
    succeed: that is the file the pipeline reads) and on libhdf5's (where it may answer None)."""
    rng = np.random.default_rng(9)
    batches = [_batch(rng, 512, "chr1"), _batch(rng, 300, "chr10_KI270_random"), _batch(rng, 1, "c")]
    mixed = _batch(rng, 5, "chr2")
    batches.append((np.array(["chr2", "chr2", "chr21_alt", "chr2", "a"], dtype='S'),) + mixed[1:])
    for bulk in (True, False):
        path = str(tmp_path / ("b%d.hdf" % bulk))
        store = DataStore(path, mode='w', bulk=bulk)
        for k, b in enumerate(batches):
            store.write_prediction_arrays(k, *b[:7])
        store.close()
        with h5.File(path) as f:
            for k in (2, 0, 1, 3, 1):                       # not in file order: the heap scan's cursor must not matter
                base = "predictions/batch_%d" % k
                got = f.read_prediction_batch(base)
                if got is None:
                    assert not bulk
                    continue
                contigs, blob, positions, depths, freq, probs = got
                n = len(batches[k][1])
                assert contigs.shape[0] == n and positions.dtype == np.int32 and probs.dtype == np.float64
                shape, want_blob = f.read_strings_shaped(base + "/candidates")
                assert blob == want_blob and tuple(shape) == (n, 1)
                names = [bytes(row).rstrip(b"\0") for row in contigs]
                assert names == [c for c in np.asarray(f[base + "/contigs"]).tolist()]
                assert np.array_equal(positions, np.asarray(f[base + "/positions"])) and np.array_equal(depths, np.asarray(f[base + "/depths"]))
                assert np.array_equal(freq, np.asarray(f[base + "/candidate_frequency"]))
                assert np.array_equal(probs, np.asarray(f[base + "/base_prediction"]))
            assert f.read_prediction_batch("predictions/batch_99") is None
    # the pipeline's own file is read the fast way
    with h5.File(str(tmp_path / "b1.hdf")) as f:
        assert f.read_prediction_batch("predictions/batch_0") is not None


def test_builder_store_is_published_by_close_only(tmp_path):
    import os
    path = str(tmp_path / "p.hdf")
    store = DataStore(path, mode='w', bulk=True)
    rng = np.random.default_rng(1)
    store.write_prediction_arrays(0, *_batch(rng, 8, "chr1")[:7])
    assert not os.path.exists(path)
    store.abort()
    assert not os.path.exists(path) and not os.path.exists(path + ".tmp")
