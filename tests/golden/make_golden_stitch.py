"""Golden FASTA of the polish stitch, produced by the REFERENCE's own perform_stitch (build container only):

    PYTHONDONTWRITEBYTECODE=1 /opt/conda/bin/python3.9 tests/golden/make_golden_stitch.py

Needs h5py (conda interpreter: h5py 3.3 / numpy 1.26; the reference pins h5py 2.10 / numpy 1.22, so the
`np.int` / `np.float` aliases it relies on are restored here before importing it).  The prediction files are
written with the reference's DataStorePredict; their contents are stored as an .npz so the test can write the
same files with pepper_amd's store and compare its stitch with the committed FASTA.  Only data is committed.
"""
import os
import shutil
import sys
import tempfile
import warnings

import numpy as np

np.int = int
np.float = float
warnings.filterwarnings("ignore")
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))


def regions(rng):
    """(file index, contig, start, end, [chunk arrays]) with overlapping neighbours, >10 chunks, padding."""
    out = []

    def region(fi, contig, start, end):
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
        chunks, at, cid = [], 0, 0
        while True:
            cp, ci = pos[at:at + 1000], idx[at:at + 1000]
            pad = 1000 - len(cp)
            cp = np.concatenate([cp, -np.ones(pad, dtype=cp.dtype)])
            ci = np.concatenate([ci, -np.ones(pad, dtype=ci.dtype)])
            chunks.append((cid, cp, ci, rng.integers(0, 5, size=1000), rng.integers(0, 60, size=1000)))
            if at + 1000 >= len(pos):
                break
            at += 950
            cid += 1
        out.append((fi, contig, start, end, chunks))

    region(0, "contig_2", 0, 9000)
    region(1, "contig_2", 8900, 12000)
    region(0, "contig_2", 11900, 12700)
    region(1, "contig_10", 0, 700)
    region(0, "contig_1", 500, 1800)
    return out


def main():
    from pepper.modules.python.DataStorePredict import DataStore
    from pepper.modules.python.perform_stitch import perform_stitch
    rng = np.random.default_rng(2024)
    regs = regions(rng)
    tmp = tempfile.mkdtemp()
    try:
        stores = [DataStore(os.path.join(tmp, "pepper_prediction_%d.hdf" % i), mode="w") for i in range(2)]
        flat = {}
        for ri, (fi, contig, start, end, chunks) in enumerate(regs):
            for cid, cp, ci, bases, phred in chunks:
                stores[fi].write_prediction(contig, np.int64(start), np.int64(end), np.int64(cid), cp, ci, bases, phred)
                for nm, arr in (("position", cp), ("index", ci), ("bases", bases), ("phred", phred)):
                    flat["r%d_c%d_%s" % (ri, cid, nm)] = np.asarray(arr)
            flat["r%d_meta" % ri] = np.array([fi, start, end, len(chunks)])
            flat["r%d_contig" % ri] = np.array(contig)
        for s in stores:
            s.file_handler.close() if hasattr(s, "file_handler") else None
        perform_stitch(tmp, os.path.join(tmp, "out"), 2)
        fasta = open(os.path.join(tmp, "out_pepper_polished.fa")).read()
    finally:
        pass
    with open(os.path.join(OUT, "polish_stitch_ref.fa"), "w") as fh:
        fh.write(fasta)
    np.savez_compressed(os.path.join(OUT, "polish_stitch_inputs.npz"), n_regions=len(regs), **flat)
    print("contigs:", [l for l in fasta.splitlines() if l.startswith(">")], "bytes", len(fasta))
    shutil.rmtree(tmp)


if __name__ == "__main__":
    main()
