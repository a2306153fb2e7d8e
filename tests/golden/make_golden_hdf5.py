"""Generate golden HDF5 files with the REFERENCE's own DataStore writers (build container only).

    PYTHONDONTWRITEBYTECODE=1 /opt/conda/bin/python3.9 tests/golden/make_golden_hdf5.py

Needs h5py, which only the conda interpreter of this image has (h5py 3.3 / numpy 1.26; the
reference pins h5py 2.10 / numpy 1.22 -- same on-disk layout).  `np.float` was removed from numpy
1.24+, so the alias the reference relies on (DataStorePredict.py:64) is restored here, in this
script, before importing it.  Inputs are written next to the outputs as .npz so the tests can feed
pepper_amd's writers the same data.  Only data files are committed.
"""
import os
import sys
import warnings

import numpy as np

np.float = float  # alias removed in numpy >= 1.24; the reference targets numpy 1.22
warnings.filterwarnings("ignore")
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))


def variant_inputs():
    rng = np.random.default_rng(77)
    groups = {}
    for name, n in (("chr20_1000_2000", 5), ("chr20_2000_3000", 3)):
        images = rng.integers(-125, 126, size=(n, 33, 26))
        images[0, 16, 4] = -130     # unclamped column: wraps through int8 in the reference writer
        images[0, 16, 8] = 140
        groups[name] = dict(
            contigs=["chr20"] * n,
            positions=[int(p) for p in rng.integers(1000, 3000, size=n)],
            depths=[int(d) for d in rng.integers(1, 126, size=n)],
            candidates=[["1A"], ["2ACGT"], ["3AC"], ["1T"], ["2AG"]][:n],
            candidate_frequency=[[int(v)] for v in rng.integers(1, 126, size=n)],
            images=images.tolist())
    return groups


def make_variant():
    from pepper_variant.modules.python.DataStore import DataStore
    from pepper_variant.modules.python.DataStorePredict import DataStore as DataStorePredict
    groups = variant_inputs()
    path = os.path.join(OUT, "variant_images_ref.hdf5")
    if os.path.exists(path):
        os.remove(path)
    with DataStore(path, 'w') as ds:
        for name, g in groups.items():
            ds.write_summary(name, g["contigs"], g["positions"], g["depths"], g["candidates"],
                             g["candidate_frequency"], g["images"], [0] * len(g["contigs"]),
                             [0] * len(g["contigs"]), False)
    np.savez_compressed(os.path.join(OUT, "variant_images_inputs.npz"),
                        **{f"{name}__{k}": np.array(v, dtype=object if k == "candidates" else None)
                           for name, g in groups.items() for k, v in g.items()})
    # predictions: two batches as predict() writes them
    path = os.path.join(OUT, "variant_predictions_ref.hdf")
    if os.path.exists(path):
        os.remove(path)
    ds = DataStorePredict(path, mode='w')
    rng = np.random.default_rng(78)
    pred_inputs = {}
    for b, name in enumerate(groups):
        g = groups[name]
        probs = rng.random((len(g["contigs"]), 3)).astype(np.float32)
        # what the collate hands to write_prediction: python lists of per-item numpy values
        ds.write_prediction(b, g["contigs"], [np.int32(p) for p in g["positions"]],
                            [np.uint8(d) for d in g["depths"]],
                            [np.array(c, dtype=object) for c in g["candidates"]],
                            [np.array(f, dtype=np.uint8) for f in g["candidate_frequency"]], probs)
        pred_inputs[f"probs_{b}"] = probs
    ds.file_handler.close()
    np.savez_compressed(os.path.join(OUT, "variant_predictions_inputs.npz"), **pred_inputs)


def make_polish():
    from pepper.modules.python.DataStore import DataStore
    from pepper.modules.python.DataStorePredict import DataStore as DataStorePredict
    rng = np.random.default_rng(79)
    path = os.path.join(OUT, "polish_images_ref.hdf")
    if os.path.exists(path):
        os.remove(path)
    inputs = {}
    with DataStore(path, 'w') as ds:
        for cid in range(2):
            image = rng.integers(0, 255, size=(1000, 10)).astype(np.float64).tolist()
            position = [(int(1000 + i), 0) for i in range(1000)]
            if cid == 1:
                position[900:] = [(-1, -1)] * 100
            index = list(range(1000))
            label = [0] * 1000
            name = f"contig_1_1000_2000_{cid}"
            ds.write_summary(("contig_1", 1000, 2000), image, label, position, index, cid, name)
            inputs[f"{name}__image"] = np.array(image)
            inputs[f"{name}__position"] = np.array(position)
    np.savez_compressed(os.path.join(OUT, "polish_images_inputs.npz"), **inputs)

    path = os.path.join(OUT, "polish_predictions_ref.hdf")
    if os.path.exists(path):
        os.remove(path)
    ds = DataStorePredict(path, mode='w')
    pin = {}
    for cid in range(2):
        bases = rng.integers(0, 5, size=1000)
        phred = rng.random(1000).astype(np.float32) * 60
        position = np.array([(1000 + i, 0) for i in range(1000)])
        index = np.arange(1000)
        ds.write_prediction("contig_1", np.int64(1000), np.int64(2000), np.int64(cid), position, index, bases, phred)
        pin[f"bases_{cid}"], pin[f"phred_{cid}"] = bases, phred
    ds.file_handler.close()
    np.savez_compressed(os.path.join(OUT, "polish_predictions_inputs.npz"), **pin)


if __name__ == "__main__":
    make_variant()
    make_polish()
    import h5py

    def show(n, o):
        if isinstance(o, h5py.Dataset):
            print(" ", n, o.shape, o.dtype)
    for fn in ("variant_images_ref.hdf5", "variant_predictions_ref.hdf", "polish_images_ref.hdf",
               "polish_predictions_ref.hdf"):
        print(fn)
        h5py.File(os.path.join(OUT, fn), "r").visititems(show)
