"""Writes tests/golden/realign_cases.npz: one region of reads with the results of the REFERENCE's own SSW build
(oracle/_ref/libref_ssw.so, compiled by oracle/Makefile from /root/reference/pepper/modules/src/local_reassembly/
{ssw.c,ssw_cpp.cpp}), driven the way ReadAligner::align_reads_to_reference drives it (simple_aligner.cpp:66-106).
Run in the build container (needs /root/reference):  python tests/golden/make_golden_realign.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import ssw  # noqa: E402

BASES = "ACGT"


def main():
    assert ssw.have_reference(), "oracle/_ref/libref_ssw.so missing: make -C oracle ref"
    rng = np.random.default_rng(20260926)
    reference = "".join(BASES[k] for k in rng.integers(0, 4, 1100))
    reference = reference[:400] + "A" * 23 + reference[423:700] + "N" * 3 + reference[703:]
    start = 70000
    pos, seqs = ssw.simulate_reads(rng, reference, start, 90)
    p2, s2 = ssw.simulate_reads(rng, reference, start, 40, sub=0.12, ins=0.09, dele=0.1)
    pos += p2
    seqs += s2
    # long gaps (wide bands), short reads (8-bit cells), unrelated / N reads, a read before the region
    for k in range(12):
        a = int(rng.integers(0, 300))
        seg = reference[a:a + 500]
        cut = int(rng.integers(50, 400))
        gap = int(rng.integers(30, 200))
        seqs.append(seg[:cut] + ("".join(BASES[x] for x in rng.integers(0, 4, gap)) if k % 2 else "") + seg[cut + (0 if k % 2 else gap):])
        pos.append(start + a)
    for k in range(25):
        a = int(rng.integers(0, 1050))
        seqs.append(reference[a:a + int(rng.integers(1, 60))] or "G")
        pos.append(start + a)
    pos += [start + 10, start + 20, start - 5, start + 1099]
    seqs += ["".join(BASES[x] for x in rng.integers(0, 4, 200)), "N" * 30, "ACGTACGTAC", "T"]
    res = ssw.realign_reads(reference, start, pos, seqs, aligner=lambda r, q: ssw.align_reference(r, q) + (0,))
    texts = []
    for (st, score, p, pe, ops) in res:
        texts.append("".join("%d%s" % (n, {7: "=", 8: "X", 1: "I", 2: "D", 4: "S"}[o]) for o, n in ops))
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "realign_cases.npz")
    np.savez_compressed(out, reference=np.array(reference), region_start=np.int64(start), read_pos=np.array(pos, np.int64),
                        sequences=np.array("|".join(seqs)), status=np.array([r[0] for r in res], np.int32),
                        score=np.array([r[1] for r in res], np.int32), new_pos=np.array([r[2] for r in res], np.int64),
                        new_pos_end=np.array([r[3] for r in res], np.int64), cigars=np.array("|".join(texts)))
    print(out, len(pos), "reads,", sum(r[0] == 1 for r in res), "aligned")


if __name__ == "__main__":
    main()
