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


def test_builder_store_is_published_by_close_only(tmp_path):
    import os
    path = str(tmp_path / "p.hdf")
    store = DataStore(path, mode='w', bulk=True)
    rng = np.random.default_rng(1)
    store.write_prediction_arrays(0, *_batch(rng, 8, "chr1")[:7])
    assert not os.path.exists(path)
    store.abort()
    assert not os.path.exists(path) and not os.path.exists(path + ".tmp")
