"""The polish image chain (pa_polish_chain_run: BAM records -> clipped reads -> re-aligned reads -> summary rows -> chunks, all on
the device) against (a) the reference's own builds -- oracle/_ref/libref_ssw.so for every re-alignment and oracle/_ref/
libref_polish_encoder.so for every summary, fed by the tests' restatement of get_reads' clipping -- and (b) the host form of the
same pipeline (PEPPER_AMD_POLISH_CHAIN=0: host arrays between the stages), whose image files must be identical dataset by dataset.
Reference: pepper/modules/python/AlignmentSummarizer.py:18-56, 296-358; simple_aligner.cpp:66-106; summary_generator.cpp."""
import glob
import os
from types import SimpleNamespace

import numpy as np
import pytest

import bam_utils as bu
import pileup_utils as pu
from conftest import need_reference_build
from oracle import ssw
from pepper_amd import h5

pytestmark = pytest.mark.gpu


def _dataset(tmp_path, seed, length, n_reads, read_len, deep_at=None, extra=()):
    rng = np.random.default_rng(seed)
    draft = pu.random_reference(rng, length)
    reads = pu.simulate_reads(rng, draft, 0, n_reads=n_reads, read_len=read_len, ins_rate=0.03, del_rate=0.03)
    reads = [r for r in reads if not any(op in (3, 6) for op, _ in r["cigar"])]
    if deep_at is not None:          # a pile beyond MAX_READS_IN_REGION: short reads stacked on one stretch
        a, b, n = deep_at
        more = pu.simulate_reads(rng, draft[a:b], a, n_reads=n, read_len=(120, 260), ins_rate=0.02, del_rate=0.02)
        reads += [r for r in more if not any(op in (3, 6) for op, _ in r["cigar"])]
    reads += list(extra(draft) if callable(extra) else extra)
    reads.sort(key=lambda r: r["pos"])
    for i, r in enumerate(reads):
        r["name"] = "q%d" % i
    bam_path, fa_path = str(tmp_path / "reads.bam"), str(tmp_path / "draft.fa")
    bu.write_bam(bam_path, [("ctg1", len(draft))], {0: reads})
    with open(fa_path, "w") as fh:
        fh.write(">ctg1\n" + draft + "\n")
    return draft, reads, bam_path, fa_path


def _borderline_reads(draft):
    """Reads around the 8-bit pass's overflow threshold (a running maximum of 249 = 63 matches): exact copies of 58 .. 70 draft
    bases, the same with one substitution, and alignments whose path takes a row move next to a column move, a reference skip, N
    bases -- the shapes the overflow proof's rules treat one by one."""
    out = []

    def read(pos, seq, cigar):
        out.append(dict(pos=int(pos), reverse=False, mapq=60, seq=seq, qual=np.full(len(seq), 25, np.uint8), cigar=cigar))
    flip = {"A": "C", "C": "G", "G": "T", "T": "A"}
    for k, n in enumerate(range(58, 71)):
        p = 150 + 37 * k
        read(p, draft[p:p + n], [(0, n)])
        q = 1400 + 41 * k
        seq = draft[q:q + n + 4]
        read(q, seq[:30] + flip[seq[30]] + seq[31:], [(0, n + 4)])
    p = 2300
    read(p, draft[p:p + 40] + "ACG" + draft[p + 43:p + 120], [(0, 40), (1, 3), (2, 3), (0, 77)])          # rows, then columns
    p = 2500
    read(p, draft[p:p + 35] + draft[p + 40:p + 40 + 5] + "TTGCA" + draft[p + 45:p + 130], [(0, 35), (2, 5), (0, 5), (1, 5), (0, 85)])
    p = 2900
    read(p, draft[p:p + 20] + "N" * 3 + draft[p + 23:p + 100], [(0, 100)])                                   # N bases never match
    p = 3100
    read(p, "GGGTT" + draft[p:p + 90] + "ACACA", [(4, 5), (0, 90), (4, 5)])                                  # soft clips
    return out


def _path_bound_reaches_249(seq, cigar, window):
    """The test's own statement of the overflow proof: lower bounds of the 8-bit pass's h and hs along the read's
    clipped alignment -- +4 / -6 on the diagonal, 8 + 2 (k - 1) off for k rows (h only; hs falls to 0) or k columns (from hs)."""
    h = hs = i = j = 0
    for op, n in cigar:
        if op in (0, 7, 8):
            for k in range(min(n, len(seq) - i, len(window) - j)):
                a, b = seq[i + k], window[j + k]
                h = max(h + (4 if a == b and a in "ACGT" else -6), 0)
                if h >= 249:
                    return True
            hs = h
            i += n
            j += n
        elif op in (1, 4):
            h, hs = max(h - 8 - 2 * (n - 1), 0), 0
            i += n
        elif op in (2, 3):
            hs = max(hs - 8 - 2 * (n - 1), 0)
            h = hs
            j += n
        if i >= len(seq) or j >= len(window):
            break
    return False


def _groups(path):
    out = {}
    with h5.File(path) as f:
        for name in f.keys("summaries"):
            base = "summaries/" + name + "/"
            out[name] = {k: np.asarray(f[base + k]) for k in ("image", "label", "position", "index", "region_start", "region_end", "chunk_id")}
            out[name]["contig"] = f[base + "contig"]
    return out


def _make(bam_path, fa_path, out_dir, threads, chain, monkeypatch, regions=None, device_ids=None):
    from pepper_amd.polish import ImageGenerationUI as ui
    from pepper_amd.polish.make_images import make_images
    monkeypatch.setenv("PEPPER_AMD_POLISH_CHAIN", "1" if chain else "0")
    if regions is not None:
        monkeypatch.setattr(ui.UserInterfaceSupport, "CHAIN_REGIONS", regions)
    make_images(bam_path, fa_path, None, out_dir, threads, device_ids=device_ids)
    merged = {}
    for path in glob.glob(os.path.join(out_dir, "*.hdf")):
        merged.update(_groups(path))
    return merged


def _assert_same(got, want):
    assert sorted(got) == sorted(want)
    for name in want:
        for key, w in want[name].items():
            g = got[name][key]
            assert (g == w) if isinstance(w, (str, bytes)) else np.array_equal(g, w), (name, key)


def test_chain_images_equal_the_reference_builds(tmp_path, monkeypatch):
    """Every interval of a 6.3 kb draft through the chain (one call: 7 intervals), compared with the reference's own SSW and
    SummaryGenerator builds fed by the restated clipping: one hop to the reference."""
    from pepper_amd.polish.AlignmentSummarizer import AlignmentSummarizer
    from pepper_amd.polish.ImageGenerationUI import UserInterfaceSupport
    ref_enc = pu.load_reference_polish_encoder()
    if ref_enc is None or not ssw.have_reference():
        need_reference_build("oracle/_ref (polish encoder, SSW)")
    low = dict(pos=2950, reverse=False, mapq=0, seq="ACGT" * 60, qual=np.full(240, 20, np.uint8), cigar=[(0, 240)])
    draft, reads, bam_path, fa_path = _dataset(tmp_path, 311, 6300, 330, (500, 3000), extra=lambda d: [low] + _borderline_reads(d))
    got = _make(bam_path, fa_path, str(tmp_path / "chain"), 1, True, monkeypatch)
    _, intervals = UserInterfaceSupport.make_intervals([("ctg1", None)], fa_path)
    assert len(intervals) == 7
    checked = 0
    for (_, start, end) in intervals:
        checked += _check_interval(got, ref_enc, "ctg1", draft, start, end, bu.restated_get_reads(reads, start, end, False, 0))
    assert checked == len(got) and checked >= 10


def _check_interval(got, ref_enc, contig, draft, start, end, clipped):
    """One interval's chunks in `got` against the reference's SSW build (every clipped read re-aligned to draft[start, end + 20))
    and its SummaryGenerator build on the re-aligned reads; -> chunks compared."""
    from pepper_amd.polish.AlignmentSummarizer import AlignmentSummarizer
    res = ssw.realign_reads(draft[start:end + 20], start, [r["pos"] for r in clipped], [r["seq"] for r in clipped],
                            aligner=ssw.align_reference)
    realigned = [dict(r, pos=p, cigar=[(0 if o in (7, 8) else o, n) for o, n in ops]) if st == 1 else r
                 for r, (st, score, p, pe, ops) in zip(clipped, res)]
    img, pos = pu.run_polish_reference(ref_enc, pu.FlatPileup(start, end, draft[start:end + 1], realigned), start, end)
    want = AlignmentSummarizer.chunk_images(SimpleNamespace(image=img, genomic_pos=[tuple(x) for x in pos.tolist()]), 1000, 50)
    for cid, (wi, wp) in enumerate(zip(want[0], want[2])):
        g = got["%s_%d_%d_%d" % (contig, start, end, cid)]
        assert np.array_equal(g["image"], wi), (start, cid)
        assert np.array_equal(g["position"], wp[:, 0]) and np.array_equal(g["index"], wp[:, 1])
        assert int(g["region_start"]) == start and int(g["region_end"]) == end and int(g["chunk_id"]) == cid
        assert not g["label"].any() and g["contig"] == contig
    assert "%s_%d_%d_%d" % (contig, start, end, len(want[0])) not in got
    return len(want[0])


def test_sampled_intervals_of_a_bench_shaped_job_equal_the_reference_builds(tmp_path, monkeypatch):
    """tools/synth_bam's data (the shape bench.py's polish_make_images leg runs on: 60x of 4-12 kb reads with mapq-0 / duplicate /
    secondary / supplementary records among them), 2 Mb here: every interval through default make_images with 8 workers -- calls
    of 128 intervals, ~8 500 reads each, i.e. the packed kernel forms the product picks by itself -- and 12 of them (the first,
    the last, 10 drawn at random) against the reference's SSW and SummaryGenerator builds, the reads parsed from the BAM by the
    tests' own BGZF / BAM reader."""
    import json
    import subprocess
    import test_gpu_images_vs_ref as vr
    from pepper_amd import build
    from pepper_amd.polish.ImageGenerationUI import UserInterfaceSupport
    from pepper_amd.variant.fasta import FASTA_handler
    ref_enc = pu.load_reference_polish_encoder()
    tool = build.build_tools()
    if ref_enc is None or not ssw.have_reference():
        need_reference_build("oracle/_ref (polish encoder, SSW)")
    assert tool is not None, "tools/synth_bam did not build"
    work = str(tmp_path)
    info = json.loads(subprocess.run([tool, work, "2000000", "60", "78"], check=True, capture_output=True, text=True).stdout)
    bam, fa = os.path.join(work, "reads.bam"), os.path.join(work, "draft.fa")
    got = _make(bam, fa, os.path.join(work, "chain"), 8, True, monkeypatch)
    fasta = FASTA_handler(fa)
    contig = fasta.get_chromosome_names()[0]
    length = fasta.get_chromosome_sequence_length(contig)
    assert length == info["genome_bases"]
    draft = fasta.get_reference_sequence(contig, 0, length)
    _, intervals = UserInterfaceSupport.make_intervals([(contig, None)], fa)
    assert len(intervals) > 1500
    linear = vr._bai_linear(bam + ".bai", 0)
    rng = np.random.default_rng(6)
    picks = sorted({0, len(intervals) - 1} | {int(k) for k in rng.choice(np.arange(1, len(intervals) - 1), 10, replace=False)})
    checked = 0
    for k in picks:
        _, start, end = intervals[k]
        records = vr._records_reaching(bam, linear, start, end + 1)
        clipped = vr._clip(records, start, end, False, 0)
        assert 0 < len(clipped) < 1500             # (below the reservoir cap: the interval goes through the chain)
        checked += _check_interval(got, ref_enc, contig, draft, start, end, clipped)
    assert checked >= 12


@pytest.mark.parametrize("threads,regions", [(1, 128), (3, 2), (2, 5)])
def test_chain_files_equal_the_host_form(tmp_path, monkeypatch, threads, regions):
    """Image files of the chain and of the host form, dataset by dataset: a stretch without reads (no groups), a pile of more
    than 1 500 reads (reservoir sample on the host, the chain's other intervals unaffected), several calls per worker."""
    draft, reads, bam_path, fa_path = _dataset(tmp_path, 97 + threads, 9400, 260, (400, 2600), deep_at=(5100, 5500, 1900))
    # nothing reaches 6 850 .. 8 150: the interval 6 900 - 8 100 has no reads
    reads2 = [r for r in reads if r["pos"] + bu.ref_length(r["cigar"]) < 6850 or r["pos"] > 8150]
    bu.write_bam(bam_path, [("ctg1", len(draft))], {0: reads2})
    want = _make(bam_path, fa_path, str(tmp_path / "host"), 1, False, monkeypatch)
    got = _make(bam_path, fa_path, str(tmp_path / "chain"), threads, True, monkeypatch, regions=regions)
    assert not any(name.startswith("ctg1_6900_") for name in want) and any(name.startswith("ctg1_7900_") for name in want)
    assert any(name.startswith("ctg1_4900_") for name in want)
    _assert_same(got, want)


def test_chain_host_packed_form_and_no_realignment(tmp_path, monkeypatch):
    """PEPPER_AMD_DEVICE_INFLATE=0: the arena filled by the host packer and uploaded (the form a BAM without an index takes);
    and PolishChain.run(realign=False) against create_summary(realignment_flag=False)."""
    from pepper_amd.polish import PEPPER
    from pepper_amd.polish.AlignmentSummarizer import AlignmentSummarizer
    from pepper_amd.variant.PEPPER_VARIANT import PackedEncoder
    from pepper_amd.variant.bam import BAM_handler
    from pepper_amd.variant.fasta import FASTA_handler
    draft, reads, bam_path, fa_path = _dataset(tmp_path, 5, 3300, 150, (300, 1500))
    want = _make(bam_path, fa_path, str(tmp_path / "host"), 1, False, monkeypatch)
    monkeypatch.setenv("PEPPER_AMD_DEVICE_INFLATE", "0")
    got = _make(bam_path, fa_path, str(tmp_path / "chain"), 1, True, monkeypatch)
    _assert_same(got, want)

    bam, fasta = BAM_handler(bam_path), FASTA_handler(fa_path)
    enc = PackedEncoder(0, 64 << 20, host_threads=1)
    chain = PEPPER.PolishChain(enc)
    bounds = [(0, 1100), (900, 2100), (1900, 3100)]
    starts, stops = [a for a, _ in bounds], [b for _, b in bounds]
    n_done, region_pairs, counts = enc.pack(bam, "ctg1", starts, stops, False, 0)
    assert n_done == 3
    rows, live, chunks = chain.run(bounds, [b"" for _ in bounds], region_pairs, counts, realign=False)
    img, pos, idx = [a.copy() for a in chain.chunk_arrays()]
    at = 0
    for (a, b), k, n_live in zip(bounds, chunks, live):
        s = AlignmentSummarizer(bam, fasta, "ctg1", a, b)
        wi, _, wp, ids = s.create_summary(realignment_flag=False)
        assert len(wi) == k and n_live == len(bam.get_reads("ctg1", a, b, False, 0, 0))
        for c in range(k):
            assert np.array_equal(img[at + c], wi[c]) and np.array_equal(pos[at + c], wp[c][:, 0]) and np.array_equal(idx[at + c], wp[c][:, 1])
        at += k
    assert at == len(img)
    enc.close()


def test_proven_overflows_skip_the_8_bit_pass_and_change_nothing(tmp_path, monkeypatch):
    """The job builder walks every read's BAM alignment and, where the path's score reaches 249, marks the read: its 8-bit score
    pass (whose results ssw.c:819-824 throws away on overflow) is not run.  PA_REALIGN_PROOF=0 runs every pass: the same chunks,
    byte for byte, in the one-read and in the two-reads-per-wavefront kernels; most long reads are proven, reads shorter than 63
    bases never are.  (The comparisons with the reference's SSW build above run with the proof on: it is the default.)"""
    from pepper_amd.polish import PEPPER
    from pepper_amd.variant.PEPPER_VARIANT import PackedEncoder
    from pepper_amd.variant.bam import BAM_handler
    rng = np.random.default_rng(12)
    short = [dict(pos=int(p), reverse=False, mapq=60, seq=None, qual=np.full(50, 20, np.uint8), cigar=[(0, 50)]) for p in (300, 1500, 2600)]
    draft, reads, bam_path, fa_path = _dataset(tmp_path, 41, 3300, 260, (200, 2500))
    for r in short:
        r["seq"] = draft[r["pos"]:r["pos"] + 50]
    edge = [r for r in _borderline_reads(draft) if r["pos"] + 200 < len(draft)]
    reads = sorted(reads + short + edge, key=lambda r: r["pos"])
    for i, r in enumerate(reads):
        r["name"] = "q%d" % i
    bu.write_bam(bam_path, [("ctg1", len(draft))], {0: reads})
    bam = BAM_handler(bam_path)
    bounds = [(0, 1100), (900, 2100), (1900, 3100)]
    windows = [draft[a:b + 20].encode() for a, b in bounds]
    results = {}
    for single in ("1", "0"):
        monkeypatch.setenv("PA_REALIGN_SINGLE", single)
        for proof in ("1", "0"):
            monkeypatch.setenv("PA_REALIGN_PROOF", proof)
            enc = PackedEncoder(0, 64 << 20, host_threads=1)
            chain = PEPPER.PolishChain(enc)
            n_done, region_pairs, counts = enc.pack(bam, "ctg1", [a for a, _ in bounds], [b for _, b in bounds], False, 0)
            assert n_done == 3
            rows, live, chunks = chain.run(bounds, windows, region_pairs, counts, realign=True)
            t = chain.timing()
            results[single, proof] = ([a.copy() for a in chain.chunk_arrays()], list(chunks), t["proven_overflows"], t["realigned"], t["pairs"])
            enc.close()
    for single in ("1", "0"):
        on, off = results[single, "1"], results[single, "0"]
        assert off[2] == 0 and on[2] > 0.6 * on[3] and on[2] <= on[4] - 3, (on[2:], off[2:])
        assert on[1] == off[1] and on[3] == off[3]
        for a, b in zip(on[0], off[0]):
            assert np.array_equal(a, b)
    for a, b in zip(results["1", "1"][0], results["0", "1"][0]):
        assert np.array_equal(a, b)
    # which reads: the test's own walk over the clipped alignments marks exactly as many, and every read it marks scores >= 249 in the
    # reference's SSW build (the bound holds for the 16-bit pass by the same rules) -- a claim the library's result contradicts
    # would show here before it showed in an image
    if ssw.have_reference():
        marked = 0
        for (a, b), window in zip(bounds, windows):
            clipped = [r for r in bu.restated_get_reads(reads, a, b, False, 0) if r.get("mapq", 60) > 0 and r["pos"] >= a]
            res = ssw.realign_reads(window.decode(), a, [r["pos"] for r in clipped], [r["seq"] for r in clipped], aligner=ssw.align_reference)
            for r, (st, score, _p, _pe, _ops) in zip(clipped, res):
                if _path_bound_reaches_249(r["seq"], r["cigar"], window.decode()[r["pos"] - a:]):
                    marked += 1
                    assert st == 1 and score >= 249, (r["pos"], score)
        assert marked == results["1", "1"][2], (marked, results["1", "1"][2])
    else:
        need_reference_build("oracle/_ref (SSW)")
    # ... and the packed form again with the strip size of the call taken from the host's bound (PA_REALIGN_ADAPT=0) instead of
    # chosen on the device from the reads' lengths
    monkeypatch.setenv("PA_REALIGN_SINGLE", "0")
    monkeypatch.setenv("PA_REALIGN_PROOF", "1")
    monkeypatch.setenv("PA_REALIGN_ADAPT", "0")
    enc = PackedEncoder(0, 64 << 20, host_threads=1)
    chain = PEPPER.PolishChain(enc)
    n_done, region_pairs, counts = enc.pack(bam, "ctg1", [a for a, _ in bounds], [b for _, b in bounds], False, 0)
    chain.run(bounds, windows, region_pairs, counts, realign=True)
    for a, b in zip(results["0", "1"][0], chain.chunk_arrays()):
        assert np.array_equal(a, b)
    enc.close()


def test_chain_refuses_what_it_cannot_hold(tmp_path):
    """A read that keeps more bases of a region than a pair's slot (2 L + 64): PA_ERR_UNSUPPORTED, and image generation takes the
    host form for that run of intervals."""
    from pepper_amd import _lib
    from pepper_amd.polish import PEPPER
    from pepper_amd.variant.PEPPER_VARIANT import PackedEncoder
    from pepper_amd.variant.bam import BAM_handler
    rng = np.random.default_rng(8)
    draft = pu.random_reference(rng, 900)
    ins = "".join(rng.choice(list("ACGT"), 700))
    fat = dict(pos=100, reverse=False, mapq=60, seq=draft[100:200] + ins + draft[200:300], qual=np.full(900, 20, np.uint8),
               cigar=[(0, 100), (1, 700), (0, 100)], name="fat")
    bam_path = str(tmp_path / "fat.bam")
    bu.write_bam(bam_path, [("ctg1", len(draft))], {0: [fat]})
    enc = PackedEncoder(0, 16 << 20, host_threads=1)
    chain = PEPPER.PolishChain(enc)
    n_done, region_pairs, counts = enc.pack(BAM_handler(bam_path), "ctg1", [90], [310], False, 0)
    with pytest.raises(_lib.PepperAmdError) as err:
        chain.run([(90, 310)], [draft[90:331].encode()], region_pairs, counts)
    assert err.value.code == _lib.PA_ERR_UNSUPPORTED
    enc.close()


def test_worker_device_map_and_device_ids(tmp_path, monkeypatch):
    """device_ids "0,0": two workers, each on `its` device -- the files are those of the single-device run."""
    from pepper_amd.polish.ImageGenerationUI import UserInterfaceSupport, parse_device_ids
    assert parse_device_ids("0,2,5") == [0, 2, 5] and parse_device_ids(None) == [0] and parse_device_ids([1]) == [1]
    assert [UserInterfaceSupport.worker_device("0,1,2", t) for t in range(5)] == [0, 1, 2, 0, 1]
    draft, reads, bam_path, fa_path = _dataset(tmp_path, 77, 4200, 160, (400, 2000))
    want = _make(bam_path, fa_path, str(tmp_path / "one"), 1, True, monkeypatch)
    got = _make(bam_path, fa_path, str(tmp_path / "two"), 2, True, monkeypatch, regions=2, device_ids="0,0")
    _assert_same(got, want)


def test_chain_in_the_packed_kernel_forms():
    """The chain's re-aligner takes the two-reads-per-wavefront score pass and the one-wavefront band stage by itself only for
    calls of thousands of reads (what a 128-interval call at 60x is); PA_REALIGN_SINGLE=0 / PA_BAND_WAVES=1 pin those forms for a
    process: the comparisons with the reference builds and with the host form again, in a child process."""
    import subprocess
    import sys
    if os.environ.get("PEPPER_AMD_REALIGN_FORMS_CHILD") == "1":
        pytest.skip("the child run itself")
    env = dict(os.environ, PA_REALIGN_SINGLE="0", PA_BAND_WAVES="1", PEPPER_AMD_REALIGN_FORMS_CHILD="1")
    run = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider",
                          "-k", "reference_builds or host_form"],
                         env=env, capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert run.returncode == 0, run.stdout[-3000:] + run.stderr[-2000:]
