"""Candidate finder + VCF writer.  pysam/htslib and the reference's compiled
module are absent here, so these are hand-derived cases of the rules in
/root/reference/pepper_variant/modules/python/CandidateFinder.py:356-581 and VcfWriter.py:48-218
(parity unpinned against the reference run; the expectations below were worked out on paper)."""
import gzip
import os
import types

import numpy as np
import pytest

from pepper_amd.variant import bgzf
from pepper_amd.variant.CandidateFinder import find_candidates, repeat_annotation, small_chunk_stitch
from pepper_amd.variant.DataStorePredict import DataStore
from pepper_amd.variant.fasta import FASTA_handler
from pepper_amd.variant.FindCandidates import process_candidates

#       0         1         2         3         4         5         6         7         8
#       012345678901234567890123456789012345678901234567890123456789012345678901234567890123
CHR1 = "ACGTACGTAGCTAGCTAGCATCGATCGNTCAGCTAGCTAGTCGATAGCAAAAAATCGATCGATCGTAGCTAGCATCGATCAGCT"
CHR2 = "GATTACAGATTACAGATTACAGATTACACCCCCCCCGATTACAGATTACA"


def options(**kw):
    o = types.SimpleNamespace(
        fasta=None, sample_name="SAMPLE", threads=1, allowed_multiallelics=4,
        snp_p_value=0.1, insert_p_value=0.25, delete_p_value=0.25,
        snp_p_value_in_lc=0.1, insert_p_value_in_lc=0.3, delete_p_value_in_lc=0.3,
        snp_q_cutoff=20, indel_q_cutoff=15, snp_q_cutoff_in_lc=20, indel_q_cutoff_in_lc=10,
        report_snp_above_freq=0, report_indel_above_freq=0)
    o.__dict__.update(kw)
    return o


@pytest.fixture()
def fasta(tmp_path):
    path = tmp_path / "ref.fa"
    with open(path, "w") as fh:
        for name, seq in (("chr1", CHR1), ("chr2", CHR2)):
            fh.write(">" + name + " some description\n")
            for i in range(0, len(seq), 30):          # wrapped lines, lower case on purpose
                fh.write(seq[i:i + 30].lower() + "\n")
    return str(path)


def naive_repeat_annotation(sequence):
    """Definition restated from CandidateFinder.py:279-297 for kmer_size 1 (quadratic)."""
    out = [1] * len(sequence)
    for i in range(len(sequence)):
        count, end = 0, i
        for j in range(i, len(sequence)):
            if sequence[j] != sequence[i]:
                break
            count += 1
            end = j + 1
        for k in range(i, min(len(sequence), end)):
            out[k] = max(out[k], count)
    return out


def test_repeat_annotation_matches_definition():
    rng = np.random.default_rng(3)
    for _ in range(200):
        n = int(rng.integers(0, 24))
        seq = "".join(rng.choice(list("ACGT"), size=n, p=[0.55, 0.15, 0.15, 0.15]))
        assert repeat_annotation(seq, 1) == naive_repeat_annotation(seq), seq


def test_fasta_handler(fasta):
    fh = FASTA_handler(fasta)
    assert fh.get_chromosome_names() == ["chr1", "chr2"]
    assert fh.get_chromosome_sequence_length("chr1") == len(CHR1)
    assert fh.get_chromosome_sequence_length("nope") == -1
    for start, stop in [(0, 1), (0, 30), (29, 31), (25, 70), (59, 61), (0, len(CHR1)), (80, 500), (-5, 3), (10, 10), (12, 3)]:
        assert fh.get_reference_sequence("chr1", start, stop) == CHR1[max(0, start):max(0, stop)], (start, stop)
    assert fh.get_reference_sequence("chr2", 28, 36) == "CCCCCCCC"
    with pytest.raises(KeyError):
        fh.get_reference_sequence("chrX", 0, 5)
    # same answers from a .fai on disk
    with open(fasta + ".fai", "w") as out:
        off1 = len(">chr1 some description\n")
        lines1 = (len(CHR1) + 29) // 30
        off2 = off1 + len(CHR1) + lines1 + len(">chr2 some description\n")
        out.write("chr1\t%d\t%d\t30\t31\n" % (len(CHR1), off1))
        out.write("chr2\t%d\t%d\t30\t31\n" % (len(CHR2), off2))
    fh2 = FASTA_handler(fasta)
    assert fh2.get_reference_sequence("chr2", 0, 50) == CHR2
    assert fh2.get_reference_sequence("chr1", 25, 70) == CHR1[25:70]


def test_bgzf_and_tabix_roundtrip(tmp_path):
    path = str(tmp_path / "x.gz")
    w = bgzf.BgzfWriter(path)
    rng = np.random.default_rng(1)
    blob = bytes(rng.integers(0, 256, size=200_000, dtype=np.uint8))   # incompressible, spans 4 blocks
    offsets = []
    for i in range(0, len(blob), 777):
        offsets.append(w.tell())
        w.write(blob[i:i + 777])
    w.close()
    assert gzip.open(path).read() == blob                 # valid multi-member gzip
    assert bgzf.read_bgzf(path) == blob
    raw = open(path, "rb").read()
    assert raw.endswith(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))
    # every virtual offset points at the right byte: block start in the file, offset inside the block
    pos, starts = 0, {}
    total = 0
    while pos < len(raw):
        assert raw[pos:pos + 4] == b"\x1f\x8b\x08\x04" and raw[pos + 12:pos + 14] == b"BC"
        bsize = int.from_bytes(raw[pos + 16:pos + 18], "little") + 1
        isize = int.from_bytes(raw[pos + bsize - 4:pos + bsize], "little")
        assert isize <= 0xff00
        starts[pos] = total
        total += isize
        pos += bsize
    for k, v in enumerate(offsets):
        assert starts[v >> 16] + (v & 0xffff) == k * 777

    tb = bgzf.TabixBuilder()
    tb.add("chr1", 10, 11, 100, 200)
    tb.add("chr1", 20, 25, 200, 300)          # adjacent chunk in the same bin merges
    tb.add("chr1", 40000, 40001, 300, 400)    # second 16 kb window
    tb.add("chr2", 5, 6, 400, 500)
    tb.write(path + ".tbi")
    idx = bgzf.parse_tbi(path + ".tbi")
    assert idx["names"] == ["chr1", "chr2"] and idx["format"] == 2 and idx["cols"] == (1, 2, 0) and idx["meta"] == "#"
    assert idx["refs"][0]["bins"] == {4681: [(100, 300)], 4683: [(300, 400)]}
    assert idx["refs"][0]["ioff"] == [100, 100, 300]
    assert idx["refs"][1]["bins"] == {4681: [(400, 500)]}
    assert bgzf.reg2bin(0, 1 << 29) == 0 and bgzf.reg2bin(1 << 14, (1 << 14) + 1) == 4682


def write_predictions(path, records, batch=4):
    """records: (contig, position, depth, allele code, support, [p0, p1, p2])."""
    with DataStore(path, "w") as store:
        for b, i in enumerate(range(0, len(records), batch)):
            part = records[i:i + batch]
            store.write_prediction(b, [r[0] for r in part], [r[1] for r in part], [r[2] for r in part],
                                   np.array([[r[3]] for r in part], dtype=object),
                                   np.array([[r[4]] for r in part], dtype=np.uint8),
                                   np.array([r[5] for r in part], dtype=np.float32))


RECORDS = [
    # het SNP, moderate confidence: QUAL int(-10 log10(1 - float32(0.9))) = 9 -> fails the Q20 cut-off
    ("chr1", 20, 30, "1A", 12, [0.05, 0.9, 0.05]),
    # confident hom-alt SNP: 1 - float32(0.999) -> QUAL 30 -> stays in the PEPPER set
    ("chr1", 31, 40, "1A", 38, [0.0005, 0.0005, 0.999]),
    # reference base N: skipped
    ("chr1", 27, 40, "1A", 20, [0.0, 1.0, 0.0]),
    # allele with a non-ACGT base: skipped
    ("chr1", 33, 40, "1N", 20, [0.0, 1.0, 0.0]),
    # hom-ref call above the SNP p-value: reported as refCall for re-genotyping
    ("chr1", 35, 50, "1C", 9, [0.8, 0.15, 0.05]),
    # hom-ref call below the p-value: dropped
    ("chr1", 36, 50, "1G", 3, [0.95, 0.03, 0.02]),
    # two SNP alleles at one site (separate records) + a duplicate of the first
    ("chr1", 40, 60, "1A", 25, [0.01, 0.98, 0.01]),
    ("chr1", 40, 60, "1G", 22, [0.02, 0.97, 0.01]),
    ("chr1", 40, 60, "1A", 25, [0.3, 0.4, 0.3]),
    # deletion of GA after anchor C(41): REF CGA ALT C, het, not in repeat (runs near 41 are short)
    ("chr1", 41, 44, "3CGA", 20, [0.001, 0.9985, 0.0005]),
    # insertion inside the A6 homopolymer (chr1:48-53): in_lc thresholds; 0.28 < 0.3 -> dropped
    ("chr1", 50, 30, "2AAT", 8, [0.72, 0.28, 0.0]),
    # same place, 0.35 >= 0.3 -> kept, hom-ref -> refCall, REP=1
    ("chr1", 51, 30, "2AAG", 9, [0.65, 0.35, 0.0]),
    # SNP + deletion at one site on chr2: REF padded to the deletion's REF
    ("chr2", 3, 20, "1C", 8, [0.1, 0.85, 0.05]),
    ("chr2", 3, 20, "3TAC", 10, [0.05, 0.9, 0.05]),
]


def read_vcf(path):
    text = bgzf.read_bgzf(path).decode()
    lines = text.splitlines()
    return [l for l in lines if l.startswith("#")], [l.split("\t") for l in lines if not l.startswith("#")]


def test_selection_rules(fasta, tmp_path):
    pred = str(tmp_path / "pred.hdf")
    write_predictions(pred, RECORDS)
    opt = options(fasta=fasta)
    pairs = [(pred, "batch_%d" % b) for b in range((len(RECORDS) + 3) // 4)]
    margin, calling = small_chunk_stitch(opt, pairs)
    # margin list: SNP alleles at sites whose own record is not hom-ref
    assert [(m[0], m[1], m[4], m[5]) for m in margin] == [
        ("chr1", 20, ["A"], [0, 1]), ("chr1", 31, ["A"], [1, 1]), ("chr1", 40, ["A"], [0, 1]),
        ("chr1", 40, ["G"], [0, 1]), ("chr1", 40, ["A"], [0, 1]), ("chr2", 3, ["C"], [0, 1])]
    got = [(c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7], c[11]) for c in calling]
    assert got == [
        ("chr1", 20, 21, "T", ["A"], [0, 1], 30, [12], False),
        ("chr1", 31, 32, "G", ["A"], [1, 1], 40, [38], False),
        ("chr1", 35, 36, "G", ["C"], [0, 0], 50, [9], False),
        ("chr1", 40, 41, "T", ["A"], [0, 1], 60, [25], False),
        ("chr1", 40, 41, "T", ["G"], [0, 1], 60, [22], False),
        ("chr1", 40, 41, "T", ["A"], [0, 1], 60, [25], False),
        ("chr1", 41, 44, "CGA", ["C"], [0, 1], 44, [20], False),
        ("chr1", 51, 52, "A", ["AAG"], [0, 0], 30, [9], True),
        ("chr2", 3, 4, "T", ["C"], [0, 1], 20, [8], False),
        ("chr2", 3, 6, "TAC", ["T"], [0, 1], 20, [10], False),
    ]
    assert calling[0][8] == pytest.approx(np.float32(0.9)) and isinstance(calling[0][8], float)

    contigs, phasing_sites, calling_sites = find_candidates(opt, str(tmp_path), pairs)
    assert contigs == ["chr1", "chr2"]
    assert [len(calling_sites[k]) for k in sorted(calling_sites)] == [1, 1, 1, 2, 1, 1, 2]   # duplicate (T,A) dropped
    assert len(phasing_sites[("chr1", 40)]) == 2

    # frequency rescue: a dropped hom-ref SNP comes back when report_snp_above_freq is set
    opt2 = options(fasta=fasta, report_snp_above_freq=0.05, report_indel_above_freq=0.2)
    _, calling2 = small_chunk_stitch(opt2, pairs)
    keys2 = [(c[0], c[1], c[3], c[4]) for c in calling2]
    assert ("chr1", 36, "C", ["G"]) in keys2                       # 3/50 = 0.06 >= 0.05
    assert ("chr1", 50, "A", ["AAT"]) in keys2                     # 8/30 >= 0.2


def test_vcf_files(fasta, tmp_path):
    pred_dir = tmp_path / "pred"
    pred_dir.mkdir()
    write_predictions(str(pred_dir / "pepper_prediction_0.hdf"), RECORDS[:7])
    write_predictions(str(pred_dir / "pepper_prediction_1.hdf"), RECORDS[7:])
    (pred_dir / "notes.txt").write_text("ignored")
    out_dir = str(tmp_path / "vcf")
    totals = process_candidates(options(fasta=fasta), str(pred_dir), out_dir)
    assert totals == (7, 2, 5, 3, 2)

    header, full = read_vcf(os.path.join(out_dir, "PEPPER_VARIANT_FULL.vcf.gz"))
    assert header[0] == "##fileformat=VCFv4.2"
    assert header[1] == '##FILTER=<ID=PASS,Description="All filters passed">'
    assert sum(h.startswith("##FORMAT=<ID=GT,") for h in header) == 1
    assert header[-3:] == ["##contig=<ID=chr1,length=%d>" % len(CHR1), "##contig=<ID=chr2,length=%d>" % len(CHR2),
                           "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tSAMPLE"]
    F = "GT:AP:GQ:DP:AD:VAF:REP"
    assert full == [
        ["chr1", "21", ".", "T", "A", "9", "PASS", ".", F, "0/1:0.9:9:30:12:0.4:0"],
        ["chr1", "32", ".", "G", "A", "30", "PASS", ".", F, "1/1:0.999:30:40:38:0.95:0"],
        ["chr1", "36", ".", "G", "C", "1", "refCall", ".", F, "0/0:0.15:1:50:9:0.18:0"],
        # two het alleles: sorted by probability (0.98 first) -> GT 1/2; QUAL from the weaker one (0.97 -> 15)
        ["chr1", "41", ".", "T", "A,G", "15", "PASS", ".", F, "1/2:0.98,0.97:15:60:25,22:0.417,0.367:0"],
        ["chr1", "42", ".", "CGA", "C", "28", "PASS", ".", F, "0/1:0.9985:28:44:20:0.455:0"],
        ["chr1", "52", ".", "A", "AAG", "1", "refCall", ".", F, "0/0:0.35:1:30:9:0.3:1"],
        # SNP padded with the deletion's tail: T>C becomes TAC>CAC; the deletion record sorts first (0.9 > 0.85)
        ["chr2", "4", ".", "TAC", "T,CAC", "8", "PASS", ".", F, "1/2:0.9,0.85:8:20:10,8:0.5,0.4:0"],
    ]
    _, pepper = read_vcf(os.path.join(out_dir, "PEPPER_VARIANT_OUTPUT_PEPPER.vcf.gz"))
    _, calling = read_vcf(os.path.join(out_dir, "PEPPER_VARIANT_OUTPUT_VARIANT_CALLING.vcf.gz"))
    _, snps = read_vcf(os.path.join(out_dir, "PEPPER_VARIANT_OUTPUT_VARIANT_CALLING_SNPs.vcf.gz"))
    _, indels = read_vcf(os.path.join(out_dir, "PEPPER_VARIANT_OUTPUT_VARIANT_CALLING_INDEL.vcf.gz"))
    assert [(r[0], r[1]) for r in pepper] == [("chr1", "32"), ("chr1", "42")]
    assert [(r[0], r[1]) for r in snps] == [("chr1", "21"), ("chr1", "36"), ("chr1", "41")]
    assert [(r[0], r[1]) for r in indels] == [("chr1", "52"), ("chr2", "4")]
    assert [(r[0], r[1]) for r in calling] == [("chr1", "21"), ("chr1", "36"), ("chr1", "41"), ("chr1", "52"), ("chr2", "4")]
    assert all(r in full for r in pepper + calling)

    for name in ("PEPPER_VARIANT_FULL", "PEPPER_VARIANT_OUTPUT_PEPPER", "PEPPER_VARIANT_OUTPUT_VARIANT_CALLING",
                 "PEPPER_VARIANT_OUTPUT_VARIANT_CALLING_SNPs", "PEPPER_VARIANT_OUTPUT_VARIANT_CALLING_INDEL"):
        assert os.path.isfile(os.path.join(out_dir, name + ".vcf.gz.tbi"))
    idx = bgzf.parse_tbi(os.path.join(out_dir, "PEPPER_VARIANT_FULL.vcf.gz.tbi"))
    assert idx["names"] == ["chr1", "chr2"]
    # the first chunk of chr1 starts right after the header, the last one of chr2 ends at the end of the data
    raw = bgzf.read_bgzf(os.path.join(out_dir, "PEPPER_VARIANT_FULL.vcf.gz"))
    header_len = sum(len(h) + 1 for h in header)
    (beg, _), = idx["refs"][0]["bins"][4681]
    assert beg == header_len                         # single block: virtual offset == uncompressed offset
    (_, end), = idx["refs"][1]["bins"][4681]
    assert end == len(raw)


def test_vcfs_do_not_depend_on_the_number_of_worker_processes(tmp_path):
    """Selection and record formatting run in `threads` processes; the duplicate-start rule and the writers are
    sequential: all ten output files are byte-identical for 1 and 3 workers."""
    import hashlib
    from pepper_amd.variant.DataStorePredict import DataStore
    from pepper_amd.variant.FindCandidates import process_candidates
    rng = np.random.default_rng(8)
    n, length = 9000, 60000
    ref = "".join("ACGT"[k] for k in rng.integers(0, 4, length))
    ref = ref[:3000] + "A" * 12 + ref[3012:]
    fa = str(tmp_path / "r.fa")
    with open(fa, "w") as fh:
        fh.write(">chr20\n" + "\n".join(ref[i:i + 60] for i in range(0, length, 60)) + "\n")
    (tmp_path / "pred").mkdir()
    store = DataStore(str(tmp_path / "pred" / "pepper_prediction_0.hdf"), "w")
    positions = np.sort(rng.choice(np.arange(100, length - 100), n, replace=True)).astype(np.int32)   # repeats: multi-allelic sites
    for b, s in enumerate(range(0, n, 512)):
        e = min(n, s + 512)
        m = e - s
        cands = []
        for p in positions[s:e]:
            r, k = ref[int(p)], int(rng.integers(3))
            cands.append(["1" + "ACGT"[("ACGT".index(r) + 1 + int(rng.integers(3))) % 4]] if k == 0 else
                         (["2" + r + "AC"] if k == 1 else ["3" + ref[int(p):int(p) + 3]]))
        store.write_prediction(b, ["chr20"] * m, positions[s:e], rng.integers(20, 80, m).astype(np.uint8),
                               np.array(cands, dtype=object), rng.integers(5, 40, (m, 1)).astype(np.uint8),
                               rng.dirichlet([1.0, 1.0, 1.0], m))
    store.close()
    digests = []
    for threads in (1, 3):
        out = str(tmp_path / ("out%d" % threads))
        process_candidates(options(fasta=fa, threads=threads), str(tmp_path / "pred"), out)
        digests.append({f: hashlib.md5(open(out + "/" + f, "rb").read()).hexdigest() for f in sorted(os.listdir(out))})
    assert len(digests[0]) == 10 and digests[0] == digests[1]


def test_low_complexity_flags_of_a_batch_equal_the_per_site_scan():
    """_in_repeat_many (run lengths of one fetched window, all positions of a batch at once) against _in_repeat (the
    reference's per-site scan of the 20-base context, CandidateFinder.py:397-418) -- including positions within ten bases
    of either end of the contig, where the context is cut."""
    from pepper_amd.variant import CandidateFinder as cf

    class Fasta(object):
        def __init__(self, text):
            self.text = text

        def get_reference_sequence(self, contig, start, stop):
            return self.text[max(0, start):max(0, stop)]
    rng = np.random.default_rng(3)
    pieces = []
    while sum(len(p) for p in pieces) < 3000:
        pieces.append("ACGTacgtN"[int(rng.integers(0, 9))] * int(rng.choice([1, 1, 1, 2, 3, 4, 5, 6, 9])))
    text = "".join(pieces)
    fasta = Fasta(text)
    positions = np.unique(np.concatenate([np.arange(0, 25), np.arange(len(text) - 25, len(text) + 3), rng.integers(0, len(text), 400)]))
    window = cf._ReferenceWindow(fasta, "c", int(positions.min()) - 16, int(positions.max()) + 16)
    bases, flags = cf._in_repeat_many(window, "c", positions)
    assert any(flags) and not all(flags)
    for p, b, f in zip(positions.tolist(), bases, flags):
        assert b == fasta.get_reference_sequence("c", p, p + 1).upper()
        if b:
            assert f == cf._in_repeat(fasta, "c", p), p
    # the same out of the I/O library (pa_candidates_reference_flags), for windows that start at 0 and further in
    from pepper_amd import h5
    for lo_pos in (0, 37, 1500):
        some = positions[positions >= lo_pos]
        window = cf._ReferenceWindow(fasta, "c", int(some.min()) - 16, int(some.max()) + 16)
        want_letters, want_flags = cf._in_repeat_arrays(window, "c", some)
        raw = window.text.encode("latin-1")
        letters, rep = np.empty(len(some), np.uint8), np.empty(len(some), np.uint8)
        pos64 = np.ascontiguousarray(some, dtype=np.int64)
        assert h5.load().pa_candidates_reference_flags(raw, len(raw), window.lo, len(some), pos64.ctypes.data, letters.ctypes.data,
                                                       rep.ctypes.data) == 0
        assert np.array_equal(letters, want_letters) and np.array_equal(rep.astype(bool), want_flags)


def _synthetic_predictions(tmp_path, n, seed, with_multi):
    """Prediction batches like the pipeline's (one allele per row, sorted positions) on two contigs, with sites that carry
    two or three allele records, duplicate (REF, ALT) records, low-complexity stretches, N reference bases and alleles
    outside ACGT."""
    rng = np.random.default_rng(seed)
    length = 12 * n + 1000
    pieces = []
    while sum(len(p) for p in pieces) < length:
        pieces.append("ACGTN"[int(rng.choice(5, p=[0.24, 0.24, 0.24, 0.24, 0.04]))] * int(rng.choice([1, 1, 1, 1, 2, 3, 6, 8])))
    refs = {"chrB": "".join(pieces)[:length], "chrA": "".join(reversed(pieces))[:length]}
    fa = str(tmp_path / "syn.fa")
    with open(fa, "w") as fh:
        for name, seq in refs.items():                     # file order chrB, chrA: the VCF order is by name
            fh.write(">" + name + "\n" + "\n".join(seq[i:i + 70] for i in range(0, length, 70)) + "\n")
    pred_dir = tmp_path / "pred"
    pred_dir.mkdir()
    for fi, contig in enumerate(("chrB", "chrA")):
        ref = refs[contig]
        store = DataStore(str(pred_dir / ("pepper_prediction_%d.hdf" % fi)), "w")
        positions = np.sort(rng.choice(np.arange(50, length - 50), n // 2, replace=False))
        if with_multi:
            extra = rng.choice(positions, n // 10)
            positions = np.sort(np.concatenate([positions, extra, extra[: n // 40]]))
        m_all = len(positions)
        for b, s in enumerate(range(0, m_all, 256)):
            pos = positions[s:s + 256].astype(np.int32)
            m = len(pos)
            cands = []
            for p in pos:
                r = ref[int(p)] if ref[int(p)] in "ACGT" else "A"
                k = int(rng.integers(0, 7))
                if k <= 2:
                    cands.append(["1" + "ACGT"[("ACGT".index(r) + 1 + k) % 4]])
                elif k == 3:
                    cands.append(["2" + r + "ACGTT"[: int(rng.integers(1, 5))]])
                elif k == 4:
                    cands.append(["3" + ref[int(p):int(p) + int(rng.integers(2, 6))].replace("N", "A")])
                elif k == 5:
                    cands.append(["1N"])                                   # not an A C G T allele: skipped
                else:
                    cands.append(["2" + r + "A"])
            probs = rng.dirichlet([0.6, 0.6, 0.6], m)
            probs[rng.random(m) < 0.05] = [0.0, 1.0, 0.0]                  # exact 1: QUAL capped through max(1e-9, .)
            store.write_prediction(b, [contig] * m, pos, rng.integers(1, 90, m).astype(np.uint8), np.array(cands, dtype=object),
                                   rng.integers(0, 60, (m, 1)).astype(np.uint8), probs)
        store.close()
    return fa, str(pred_dir)


@pytest.mark.parametrize("with_multi,freq", [(False, 0), (True, 0), (True, 0.2)])
def test_column_path_writes_the_files_of_the_tuple_path(tmp_path, monkeypatch, with_multi, freq):
    """FastCandidates (numpy columns, bulk writes, arithmetic tabix offsets) against the reference-shaped path (one tuple per
    allele, one record at a time): the five .vcf.gz hold the same text and the five .tbi the same index."""
    fa, pred_dir = _synthetic_predictions(tmp_path, 6000, 5, with_multi)
    opts = options(fasta=fa, report_snp_above_freq=freq, report_indel_above_freq=freq, allowed_multiallelics=2)
    monkeypatch.setenv("PEPPER_AMD_CANDIDATES_TUPLES", "1")
    want_totals = process_candidates(opts, pred_dir, str(tmp_path / "tuples"))
    monkeypatch.delenv("PEPPER_AMD_CANDIDATES_TUPLES")
    got_totals = process_candidates(opts, pred_dir, str(tmp_path / "columns"))
    assert tuple(got_totals) == tuple(want_totals) and want_totals[0] > 1000
    assert want_totals[1] > 0 and want_totals[3] > 0 and want_totals[4] > 0
    for name in sorted(os.listdir(tmp_path / "tuples")):
        a, b = str(tmp_path / "tuples" / name), str(tmp_path / "columns" / name)
        if name.endswith(".tbi"):
            assert bgzf.parse_tbi(a) != {} and _resolved(bgzf.parse_tbi(a), a[:-4]) == _resolved(bgzf.parse_tbi(b), b[:-4]), name
        else:
            assert bgzf.read_bgzf(a) == bgzf.read_bgzf(b), name


def _resolved(index, vcf_path):
    """Virtual offsets -> offsets in the decompressed text (the two paths cut BGZF blocks at the same places, but that is not
    part of the contract)."""
    raw = open(vcf_path, "rb").read()
    starts, pos, total = {}, 0, 0
    while pos < len(raw):
        bsize = int.from_bytes(raw[pos + 16:pos + 18], "little") + 1
        starts[pos] = total
        total += int.from_bytes(raw[pos + bsize - 4:pos + bsize], "little")
        pos += bsize
    starts[len(raw)] = total
    conv = lambda v: starts[v >> 16] + (v & 0xffff)        # noqa: E731
    refs = [{"bins": {b: [(conv(x), conv(y)) for x, y in chunks] for b, chunks in r["bins"].items()},
             "ioff": [conv(v) if v else 0 for v in r["ioff"]]} for r in index["refs"]]
    return dict(index, refs=refs)


def test_library_batches_equal_the_python_loops(tmp_path, monkeypatch):
    """pa_candidates_select_format (candidates.cpp) against _select_batch + _format_single on the same batch: the same rows
    kept, the same record text to the byte -- with support / depth ratios that are exact ties in binary (1/16, 3/16, 5/32:
    round(x, 3) rounds half to even), probabilities that tie, exact zeros and ones, every threshold of every allele kind in
    play; batches of another form (two contigs, a zero depth the reference would divide by) are left to the Python loops."""
    from pepper_amd.variant import FastCandidates as fc
    from pepper_amd.variant.CandidateFinder import _fasta
    rng = np.random.default_rng(8)
    ref = "".join("ACGTN"[int(k)] * int(r) for k, r in zip(rng.choice(5, 9000, p=[0.24, 0.24, 0.24, 0.24, 0.04]), rng.choice([1, 1, 2, 6], 9000)))
    length = len(ref)
    fa = str(tmp_path / "r.fa")
    with open(fa, "w") as fh:
        fh.write(">c1\n" + ref + "\n>c2\n" + ref[::-1] + "\n")
    n = 4000
    pos = np.sort(rng.choice(np.arange(20, length - 20), n, replace=False)).astype(np.int32)
    depth = rng.choice([1, 2, 8, 16, 32, 64, 90], n).astype(np.uint8)
    support = rng.integers(0, 60, (n, 1)).astype(np.uint8)
    cands = []
    for p in pos:
        r = ref[int(p)] if ref[int(p)] in "ACGT" else "A"
        cands.append([("1" + "ACGT"[int(rng.integers(4))], "2" + r + "ACG"[: int(rng.integers(1, 4))], "3" + "ACGTA"[: int(rng.integers(2, 6))],
                       "1n", "4A", "2")[int(rng.choice(6, p=[0.4, 0.25, 0.25, 0.04, 0.03, 0.03]))]])
    probs = rng.dirichlet([0.5, 0.5, 0.5], n).astype(np.float32)
    probs[rng.random(n) < 0.1] = [0.25, 0.5, 0.25]
    probs[rng.random(n) < 0.05] = [0.5, 0.25, 0.25]
    probs[rng.random(n) < 0.05] = [0.0, 0.0, 1.0]
    probs[rng.random(n) < 0.05] = [1.0 / 3, 1.0 / 3, 1.0 / 3]
    pred = str(tmp_path / "p.hdf")
    with DataStore(pred, "w") as store:
        store.write_prediction(0, ["c1"] * n, pos, depth, np.array(cands, dtype=object), support, probs)
        store.write_prediction(1, ["c1"] * 3 + ["c2"], pos[:4], depth[:4], np.array(cands[:4], dtype=object), support[:4], probs[:4])
        zero = depth[:4].copy()
        zero[2] = 0
        low = probs[:4].copy()
        low[2] = [0.98, 0.01, 0.01]                         # not called by probability: the frequency rule divides by the depth
        store.write_prediction(2, ["c2"] * 4, pos[:4], zero, np.array([["1A"]] * 4, dtype=object), support[:4], low)
    for freq in (0, 0.15):
        opts = options(fasta=fa, report_snp_above_freq=freq, report_indel_above_freq=freq)
        handler = _fasta(opts)
        seg = fc._native_batch(opts, fc._rules(opts), handler, pred, "batch_0")
        left = []
        want = fc._python_batch(opts, handler, pred, "batch_0", left)
        assert seg is not None and seg.raw is not None and not left and len(seg) == len(want) > 1500
        assert seg.lines == want.lines
        assert np.array_equal(seg.pos, want.pos) and np.array_equal(seg.ref_len, want.ref_len)
        assert np.array_equal(seg.snp, want.snp) and np.array_equal(seg.sel, want.sel)
        for k in (0, 1, len(seg) // 2, len(seg) - 1):
            assert seg.record(k) == want.record(k) and seg.pair(k) == want.pair(k)
        assert fc._native_batch(opts, fc._rules(opts), handler, pred, "batch_1") is None          # two contigs
        # a row of depth 0 with a valid allele: the reference divides by it whatever the rules (CandidateFinder.py:478)
        assert fc._native_batch(opts, fc._rules(opts), handler, pred, "batch_2") is None
        with pytest.raises(ZeroDivisionError):
            fc._python_batch(opts, handler, pred, "batch_2", [])


def test_native_reference_flags_equal_the_array_form_on_random_windows():
    """pa_candidates_reference_flags (per-position scan of the 20-base context) against _in_repeat_arrays (run tables over the
    whole window): low-complexity text, lower case, windows that start at the contig's first base or later, positions at and
    beyond both edges."""
    import numpy as np
    from types import SimpleNamespace
    from pepper_amd import h5
    from pepper_amd.variant.CandidateFinder import _in_repeat_arrays
    rng = np.random.default_rng(23)
    lib = h5.load()
    for trial in range(60):
        m = int(rng.integers(1, 400))
        alphabet = "ACGT" if trial % 3 else "AC"
        out = []
        while len(out) < m:
            out += [alphabet[int(rng.integers(len(alphabet)))]] * int(rng.integers(1, 9 if trial % 2 else 3))
        text = "".join(out[:m])
        if trial % 5 == 0:
            text = text.lower()
        if trial % 7 == 0:
            text = text[:m // 2] + "N" * min(6, m - m // 2) + text[m // 2 + 6:]
        lo = 0 if trial % 4 == 0 else int(rng.integers(0, 50))
        positions = np.unique(np.concatenate([rng.integers(lo - 12, lo + m + 12, 200), [lo, lo + m - 1, lo + m, lo - 1, 0]])).astype(np.int64)
        want_letters, want_flags = _in_repeat_arrays(SimpleNamespace(text=text, lo=lo), "c", positions)
        letters, flags = np.empty(len(positions), np.uint8), np.empty(len(positions), np.uint8)
        raw = text.encode("latin-1")
        assert lib.pa_candidates_reference_flags(raw, len(raw), lo, len(positions), positions.ctypes.data, letters.ctypes.data,
                                                 flags.ctypes.data) == 0
        assert np.array_equal(letters, want_letters), trial
        assert np.array_equal(flags.astype(bool), want_flags), (trial, positions[flags.astype(bool) != want_flags][:5])
