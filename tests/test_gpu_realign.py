"""GPU parity of the read re-aligner (include/pepper_amd_realign.h) with the restatement of the reference's
ReadAligner / SSW (oracle/ssw_oracle.cpp, pinned against the reference's own SSW build by
tests/test_realign_oracle.py): integer work, so every field and every CIGAR operation must be identical."""
import os

import numpy as np
import pytest

from oracle import ssw

pytestmark = pytest.mark.gpu
BASES = "ACGT"


def _rand_seq(rng, n):
    return "".join(BASES[k] for k in rng.integers(0, 4, n))


def _gpu_align(reference, region_start, pos, seqs):
    from pepper_amd.polish.PEPPER import ReadAligner
    blob = [s.encode() for s in seqs]
    off = np.zeros(len(seqs) + 1, np.int64)
    np.cumsum([len(b) for b in blob], out=off[1:])
    aligner = ReadAligner(region_start, region_start + len(reference), reference)
    return aligner.align_arrays(pos, off, np.frombuffer(b"".join(blob), np.uint8), collapse_eqx=False)


def _assert_same(out, expect):
    bad = []
    for k, (status, score, pos, pos_end, ops) in enumerate(expect):
        a, b = int(out["cigar_offset"][k]), int(out["cigar_offset"][k + 1])
        got_ops = list(zip(out["cigar_op"][a:b].tolist(), out["cigar_len"][a:b].tolist()))
        got = (int(out["status"][k]), int(out["score"][k]) if status >= 0 else 0, int(out["pos"][k]),
               int(out["pos_end"][k]) if status == 1 else -1, got_ops)
        want = (status, score, pos, pos_end, [tuple(o) for o in ops])
        if got != want:
            bad.append((k, got[:4], want[:4], got_ops[:6], ops[:6]))
    assert not bad, (len(bad), bad[:3])


def test_region_of_simulated_reads():
    rng = np.random.default_rng(11)
    reference = _rand_seq(rng, 1120)
    pos, seqs = ssw.simulate_reads(rng, reference, 5000, 160)
    hi_pos, hi_seqs = ssw.simulate_reads(rng, reference, 5000, 60, sub=0.1, ins=0.08, dele=0.1)
    pos += hi_pos
    seqs += hi_seqs
    # special reads: before the region (dropped), starting at the last base, unrelated, all N, very short, homopolymers
    pos += [4990, 5000 + 1119, 5100, 5200, 5300, 5400]
    seqs += ["ACGTACGT", "A", _rand_seq(rng, 300), "N" * 40, reference[300:307], "A" * 50]
    out = _gpu_align(reference, 5000, pos, seqs)
    _assert_same(out, ssw.realign_reads(reference, 5000, pos, seqs))
    assert (out["status"] == 1).sum() > 200 and (out["status"] == -1).sum() == 1


def test_wide_bands_and_long_gaps():
    """Long insertions / deletions force band doubling past the first workspace (129 slots per row)."""
    rng = np.random.default_rng(12)
    reference = _rand_seq(rng, 900)
    pos, seqs = [], []
    for k in range(40):
        a = int(rng.integers(0, 200))
        seg = reference[a:a + int(rng.integers(300, 700))]
        cut = int(rng.integers(20, len(seg) - 20))
        gap = int(rng.integers(40, 260))
        if k % 2:
            seq = seg[:cut] + _rand_seq(rng, gap) + seg[cut:]            # long insertion
        else:
            seq = seg[:cut] + seg[min(len(seg) - 10, cut + gap):]         # long deletion
        pos.append(100 + a)
        seqs.append(seq)
    out = _gpu_align(reference, 100, pos, seqs)
    _assert_same(out, ssw.realign_reads(reference, 100, pos, seqs))


def test_short_reads_stay_in_narrow_cells():
    """Scores below 249 never leave the 8-bit segmentation (16 segments); mixed with reads that do."""
    rng = np.random.default_rng(13)
    reference = _rand_seq(rng, 400)
    pos, seqs = [], []
    for k in range(150):
        a = int(rng.integers(0, 380))
        n = int(rng.integers(1, 75))
        seq = reference[a:a + n]
        if k % 3 == 0 and len(seq) > 4:
            seq = seq[:2] + BASES[int(rng.integers(4))] + seq[3:]
        if k % 7 == 0:
            seq = seq + _rand_seq(rng, 5)
        pos.append(a)
        seqs.append(seq or "C")
    out = _gpu_align(reference, 0, pos, seqs)
    _assert_same(out, ssw.realign_reads(reference, 0, pos, seqs))


def test_golden_vectors_from_the_reference_build(golden_dir):
    g = np.load(os.path.join(golden_dir, "realign_cases.npz"), allow_pickle=False)
    reference = str(g["reference"])
    pos = g["read_pos"].tolist()
    seqs = [s for s in str(g["sequences"]).split("|")]
    out = _gpu_align(reference, int(g["region_start"]), pos, seqs)
    cig = str(g["cigars"]).split("|")
    expect = []
    for k in range(len(pos)):
        st = int(g["status"][k])
        expect.append((st, int(g["score"][k]), int(g["new_pos"][k]), int(g["new_pos_end"][k]) if st == 1 else -1,
                       ssw.parse_cigar(cig[k]) if st == 1 else []))
    _assert_same(out, expect)


def test_readset_and_object_interfaces():
    from pepper_amd.polish.PEPPER import ReadAligner, type_read, CigarOp
    from pepper_amd.variant.bam import ReadSet
    rng = np.random.default_rng(14)
    reference = _rand_seq(rng, 600)
    pos, seqs = ssw.simulate_reads(rng, reference, 1000, 30)
    pos[3] = 990                                     # dropped
    expect = ssw.realign_reads(reference, 1000, pos, seqs)
    reads = []
    for k, (p, s) in enumerate(zip(pos, seqs)):
        r = type_read()
        r.pos, r.pos_end, r.sequence, r.query_name = p, p + len(s), s, "r%d" % k
        r.cigar_tuples = [CigarOp(0, len(s))]
        r.base_qualities = [20] * len(s)
        r.mapping_quality = 60
        reads.append(r)
    aligner = ReadAligner(1000, 1600, reference)
    got = aligner.align_reads_to_reference(reads)
    kept = [e for e in expect if e[0] >= 0]
    assert len(got) == len(kept) == 29
    for r, e in zip(got, kept):
        if e[0] == 1:
            assert (r.pos, r.pos_end) == (e[2], e[3])
            assert [(c.cigar_op, c.cigar_len) for c in r.cigar_tuples] == [(0 if o in (7, 8) else o, n) for o, n in e[4]]
    # structure-of-arrays form
    blob = [s.encode() for s in seqs]
    so = np.zeros(len(seqs) + 1, np.int64)
    np.cumsum([len(b) for b in blob], out=so[1:])
    n = len(seqs)
    rs = ReadSet(np.array(pos, np.int64), np.array([p + len(s) for p, s in zip(pos, seqs)], np.int64), np.zeros(n, np.uint8),
                 np.full(n, 60, np.int32), np.zeros(n, np.int32), np.zeros(n, np.int32), so,
                 np.frombuffer(b"".join(blob), np.uint8), np.full(int(so[-1]), 20, np.uint8), np.arange(n + 1, dtype=np.int64),
                 np.zeros(n, np.int32), np.array([len(s) for s in seqs], np.int32), ["r%d" % k for k in range(n)])
    out = aligner.align_reads_to_reference(rs)
    assert len(out) == 29 and out.names[3] == "r4"
    for i, e in enumerate(kept):
        a, b = int(out.cigar_offset[i]), int(out.cigar_offset[i + 1])
        if e[0] == 1:
            assert (int(out.pos[i]), int(out.pos_end[i])) == (e[2], e[3])
            assert list(zip(out.cigar_op[a:b].tolist(), out.cigar_len[a:b].tolist())) == [(0 if o in (7, 8) else o, n) for o, n in e[4]]
        else:
            assert b - a == 1


def test_long_reads_take_the_lds_strip_form():
    """Reads beyond 1536 padded rows do not fit the register-resident strip (24 rows per lane)."""
    rng = np.random.default_rng(15)
    reference = _rand_seq(rng, 3200)
    pos, seqs = ssw.simulate_reads(rng, reference, 0, 10, min_len=1700, full_span=0.6)
    pos += [100, 0]
    seqs += [reference[100:1700], reference[:1530]]          # just around the switch
    assert max(len(s) for s in seqs) > 2500
    out = _gpu_align(reference, 0, pos, seqs)
    _assert_same(out, ssw.realign_reads(reference, 0, pos, seqs))


def test_degenerate_read_sets():
    from pepper_amd import _lib
    reference = "ACGTTGCA" * 20
    empty = _gpu_align(reference, 10, [], [])
    assert len(empty["status"]) == 0 and empty["cigar_offset"].tolist() == [0]
    out = _gpu_align(reference, 10, [3, 9], ["ACGT", "TTGCA"])            # every read starts before the region
    assert out["status"].tolist() == [-1, -1] and out["cigar_offset"].tolist() == [0, 0, 0]
    out = _gpu_align(reference, 10, [10 + len(reference)], ["ACGT"])      # starts exactly at the end: nothing to align to
    assert out["status"].tolist() == [0]
    with pytest.raises(_lib.PepperAmdError):
        _gpu_align(reference, 10, [10 + len(reference) + 1], ["ACGT"])    # beyond it: substr() would throw in the reference
    with pytest.raises(_lib.PepperAmdError):
        _gpu_align("ACGT" * 2000, 0, [0], ["ACGT" * 1001])                # 4004 bases: above the supported read length


def test_several_windows_in_one_call():
    """pa_realigner_align_windows: reads of three regions (own reference window each) through one job table."""
    from pepper_amd.polish.PEPPER import align_windows
    rng = np.random.default_rng(16)
    windows, pos, seqs, which, expect = [], [], [], [], []
    for w, (start, width, n_reads) in enumerate([(1000, 700, 25), (90000, 1220, 40), (5, 60, 8)]):
        text = _rand_seq(rng, width)
        p, s = ssw.simulate_reads(rng, text, start, n_reads, min_len=min(30, width // 3))
        if w == 1:
            p[0] = start - 3                       # dropped: starts before its own window
        windows.append((start, text))
        pos += p
        seqs += s
        which += [w] * n_reads
        expect += ssw.realign_reads(text, start, p, s)
    blob = [s.encode() for s in seqs]
    off = np.zeros(len(seqs) + 1, np.int64)
    np.cumsum([len(b) for b in blob], out=off[1:])
    out = align_windows(windows, which, pos, off, np.frombuffer(b"".join(blob), np.uint8), collapse_eqx=False)
    _assert_same(out, expect)
    assert (out["status"] == -1).sum() == 1 and (out["status"] == 1).sum() > 60


def test_low_complexity_sequences():
    """Tandem repeats, two-letter alphabets and homopolymer runs: many equal-scoring cells, so the end / begin cell
    choice and every tie rule of the band stage decide the result."""
    rng = np.random.default_rng(17)

    def lowc(n, kind):
        if kind == 0:
            unit = _rand_seq(rng, int(rng.integers(1, 4)))
            return (unit * (n // len(unit) + 1))[:n]
        if kind == 1:
            return "".join(rng.choice(list("AC"), n))
        out = []
        while len(out) < n:
            out += [BASES[int(rng.integers(4))]] * int(rng.integers(1, 12))
        return "".join(out[:n])
    for kind in (0, 1, 2):
        reference = lowc(700, kind)
        pos, seqs = ssw.simulate_reads(rng, reference, 300, 70, sub=0.03, ins=0.03, dele=0.03, min_len=20)
        pos += [300, 310, 320]
        seqs += [lowc(150, kind), reference[10:400], reference[20:30] * 6]
        out = _gpu_align(reference, 300, pos, seqs)
        _assert_same(out, ssw.realign_reads(reference, 300, pos, seqs))


def test_a_call_that_fills_the_chip_takes_the_packed_kernels():
    """>= 3 072 reads in one call: the score pass with two reads per wavefront (sw_ends_pair_kernel: reads paired by window
    length, several strip sizes in one call, the 24-row catch-all, a read beyond 24 rows per lane in the LDS form) and the band
    stage with one wavefront per read (band_kernel<1>: widths tried in turn, first rows of 257 slots, the wider relaunch for long
    gaps) -- the forms no smaller call reaches by itself.  Every field and operation against the SSW restatement."""
    from pepper_amd.polish.PEPPER import align_windows
    rng = np.random.default_rng(19)
    windows, pos, seqs, which = [], [], [], []
    plans = [(1000, 1220, 520, {}), (50000, 1220, 520, dict(sub=0.08, ins=0.06, dele=0.08)), (90000, 700, 500, {}),
             (200000, 1220, 520, {}), (300000, 400, 480, dict(min_len=20)), (400000, 1530, 300, dict(min_len=900)),
             (500000, 1220, 300, {})]
    for w, (start, width, n_reads, kw) in enumerate(plans):
        text = _rand_seq(rng, width)
        p, s = ssw.simulate_reads(rng, text, start, n_reads, **kw)
        if w == 6:
            # long insertions and deletions: bands beyond the first rows' 257 slots
            for k in range(0, 60):
                a = int(rng.integers(0, 200))
                seg = text[a:a + int(rng.integers(500, 900))]
                cut = int(rng.integers(20, len(seg) - 20))
                gap = int(rng.integers(150, 400))
                s[k] = seg[:cut] + _rand_seq(rng, gap) + seg[cut:] if k % 2 else seg[:cut] + seg[min(len(seg) - 10, cut + gap):]
                p[k] = start + a
        windows.append((start, text))
        pos += p
        seqs += s
        which += [w] * len(p)
    # one read beyond the register strips (1 600 bases against its own long window), N runs, a one-base read
    long_text = _rand_seq(rng, 2000)
    windows.append((700000, long_text))
    pos += [700000, 700010, 700020, 700030]
    seqs += [long_text[:1600], "N" * 50 + long_text[60:300], "A", long_text[30:1300]]
    which += [7] * 4
    assert len(seqs) >= 3072
    expect = []
    for w, (start, text) in enumerate(windows):
        idx = [k for k in range(len(seqs)) if which[k] == w]
        expect += ssw.realign_reads(text, start, [pos[k] for k in idx], [seqs[k] for k in idx])
    blob = [s.encode() for s in seqs]
    off = np.zeros(len(seqs) + 1, np.int64)
    np.cumsum([len(b) for b in blob], out=off[1:])
    out = align_windows(windows, which, pos, off, np.frombuffer(b"".join(blob), np.uint8), collapse_eqx=False)
    _assert_same(out, expect)
    assert (out["status"] == 1).sum() > 3000


@pytest.mark.parametrize("single,band_waves", [("0", "1"), ("0", "3"), ("1", "1")])
def test_every_case_in_the_forced_kernel_forms(single, band_waves):
    """PA_REALIGN_SINGLE / PA_BAND_WAVES pin the kernel forms for a whole process (they are read once): the cases above, small
    calls included, through the two-reads-per-wavefront score pass and the one-wavefront band stage (and the other mixes)."""
    import subprocess
    import sys
    if os.environ.get("PEPPER_AMD_REALIGN_FORMS_CHILD") == "1":
        pytest.skip("the child run itself")
    env = dict(os.environ, PA_REALIGN_SINGLE=single, PA_BAND_WAVES=band_waves, PEPPER_AMD_REALIGN_FORMS_CHILD="1")
    run = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"],
                         env=env, capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert run.returncode == 0, run.stdout[-3000:] + run.stderr[-2000:]
