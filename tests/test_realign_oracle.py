"""Pins the re-alignment restatement (oracle/ssw_oracle.cpp): against the vectors the reference's own SSW build
produced (tests/golden/realign_cases.npz, tests/golden/make_golden_realign.py) and, where that build is present
(oracle/_ref/libref_ssw.so; it travels to the GPU box with the snapshot), directly on seeded read sets."""
import os

import numpy as np
import pytest

from conftest import need_reference_build
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


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_restatement_equals_reference_build_on_seeded_reads(seed):
    if not ssw.have_reference():
        need_reference_build("oracle/_ref/libref_ssw.so")
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


def test_restatement_equals_reference_build_on_low_complexity_sequences():
    """Repeats and homopolymers: equal-scoring cells everywhere, every tie rule matters (9000 further cases were
    compared offline, 0 differences)."""
    if not ssw.have_reference():
        need_reference_build("oracle/_ref/libref_ssw.so")
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


def test_apply_alignment_assembles_the_read_set():
    """Host-side assembly of the re-aligner's flat results into a ReadSet (pepper_amd.polish.PEPPER.apply_alignment),
    fed with the restatement's results instead of the GPU's: dropped reads disappear, aligned reads get the new position /
    end / CIGAR ('=' and 'X' as MATCH), the others keep what they had; `first` selects a slice of a multi-region result."""
    from pepper_amd.polish.PEPPER import apply_alignment
    from pepper_amd.variant.bam import ReadSet
    rng = np.random.default_rng(4)
    window = _rand(rng, 500)
    pos, seqs = ssw.simulate_reads(rng, window, 2000, 14)
    pos[2] = 1990                                   # dropped
    seqs[5] = "N" * 25                              # score 0: kept as it was
    res = ssw.realign_reads(window, 2000, pos, seqs)
    assert [r[0] for r in res].count(-1) == 1 and [r[0] for r in res].count(0) == 1
    n = len(seqs)
    so = np.zeros(n + 1, np.int64)
    np.cumsum([len(s) for s in seqs], out=so[1:])
    reads = ReadSet(np.array(pos, np.int64), np.array([p + len(s) for p, s in zip(pos, seqs)], np.int64), np.zeros(n, np.uint8),
                    np.full(n, 60, np.int32), np.zeros(n, np.int32), np.zeros(n, np.int32), so,
                    np.frombuffer("".join(seqs).encode(), np.uint8), np.full(int(so[-1]), 20, np.uint8),
                    np.arange(n + 1, dtype=np.int64), np.zeros(n, np.int32), np.array([len(s) for s in seqs], np.int32),
                    ["r%d" % k for k in range(n)])
    # flat result arrays as align_windows returns them, preceded by three reads of "another region"
    lead = 3
    ops_all = [(0 if o in (7, 8) else o, ln) for r in res for (o, ln) in r[4]]
    counts = [0] * lead + [len(r[4]) for r in res]
    off = np.zeros(lead + n + 1, np.int64)
    np.cumsum(counts, out=off[1:])
    out = dict(status=np.array([1] * lead + [r[0] for r in res], np.int32), score=np.zeros(lead + n, np.int32),
               pos=np.array([0] * lead + [r[2] for r in res], np.int64), pos_end=np.array([0] * lead + [r[3] for r in res], np.int64),
               cigar_offset=off, cigar_op=np.array([o for o, _ in ops_all], np.int32), cigar_len=np.array([ln for _, ln in ops_all], np.int32))
    got = apply_alignment(reads, out, first=lead)
    kept = [(k, r) for k, r in enumerate(res) if r[0] >= 0]
    assert len(got) == n - 1 and got.names == ["r%d" % k for k, _ in kept]
    for i, (k, (st, score, p, pe, ops)) in enumerate(kept):
        a, b = int(got.cigar_offset[i]), int(got.cigar_offset[i + 1])
        cigar = list(zip(got.cigar_op[a:b].tolist(), got.cigar_len[a:b].tolist()))
        s0, s1 = int(got.seq_offset[i]), int(got.seq_offset[i + 1])
        assert got.seq[s0:s1].tobytes().decode() == seqs[k]
        if st == 1:
            assert (int(got.pos[i]), int(got.pos_end[i])) == (p, pe)
            assert cigar == [(0 if o in (7, 8) else o, ln) for o, ln in ops]
        else:
            assert (int(got.pos[i]), int(got.pos_end[i])) == (pos[k], pos[k] + len(seqs[k])) and cigar == [(0, len(seqs[k]))]
