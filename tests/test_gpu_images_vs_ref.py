"""The DEFAULT variant image path (generate_images: packed reads, BGZF members inflated on the device, records walked there,
unpack_clip_kernel, the summary kernels, candidate windows) compared DIRECTLY with the reference's own encoder build
(oracle/_ref/libref_variant_encoder.so, region_summary.cpp as it lies) fed by the tests' restatement of the reference's read
clipping (tests/bam_utils.py: bam_handler.cpp:176-303) -- one hop to the reference, at the sizes the pipeline runs at: intervals
of 100 kb (196 tiles each), a pile of more than 5 000 reads (the reservoir sample of AlignmentSummarizer.py:192-199), a record
with its CIGAR in the CG tag, two contigs in one job; and 10 sampled intervals of a tools/synth_bam job of bench shape."""
import os
import struct
import zlib
from types import SimpleNamespace

import numpy as np
import pytest

import bam_utils as bu
import pileup_utils as pu
from conftest import need_reference_build
from pepper_amd import h5

pytestmark = pytest.mark.gpu

LETTERS = np.frombuffer(b"ACGT", np.uint8)


def _options(bam, fasta, out, region, region_size, threads, **over):
    o = SimpleNamespace(
        bam=bam, fasta=fasta, region=region, region_size=region_size, threads=threads, train_mode=False, use_hp_info=False,
        image_output_directory=out, include_supplementary=False, min_mapq=1, min_snp_baseq=1, min_indel_baseq=1,
        snp_frequency=0.10, insert_frequency=0.15, delete_frequency=0.15, min_coverage_threshold=3,
        snp_candidate_frequency_threshold=0.10, indel_candidate_frequency_threshold=0.12, candidate_support_threshold=2,
        skip_indels=False, downsample_rate=1.0)
    for k, v in over.items():
        setattr(o, k, v)
    return o


def _groups(directory):
    out = {}
    for fn in sorted(os.listdir(directory)):
        with h5.File(os.path.join(directory, fn)) as f:
            for name in (f.keys("summaries") if "summaries" in f else []):
                assert name not in out
                g = "summaries/" + name + "/"
                out[name] = dict(images=f[g + "images"], positions=f[g + "positions"], depths=f[g + "depths"],
                                 candidates=[c[0] for c in f[g + "candidates"].tolist()], freq=f[g + "candidate_frequency"],
                                 contigs=f[g + "contigs"].tolist())
    return out


def _fast_reads(rng, draft, n_reads, read_len, het_snps, het_del, het_ins, first=0, last=None, name="r"):
    """Nanopore-like reads of `draft` (uint8 codes 0..3) built per read with numpy: 3 % substitutions, ~2 % deletions and ~1.5 %
    inserts of 1-3 bases, and -- on one of two haplotypes -- the planted sites: het_snps {pos: alt code}, het_del {pos: length}
    (the bases after pos), het_ins {pos: codes} (after the base at pos)."""
    length = len(draft)
    last = length - 300 if last is None else last
    starts = np.sort(rng.integers(first, last, n_reads))
    snp_pos = np.array(sorted(het_snps), np.int64)
    reads = []
    for i, pos in enumerate(starts.tolist()):
        n = min(length - pos, int(rng.integers(*read_len)))
        hap = int(rng.integers(2))
        base = draft[pos:pos + n].copy()
        u = rng.random(n)
        deleted = np.zeros(n + 8, np.int32)
        dstart = np.flatnonzero(u < 0.02)
        if len(dstart):
            dl = rng.integers(1, 4, len(dstart))
            np.add.at(deleted, dstart, 1)
            np.add.at(deleted, dstart + dl, -1)
        ins = np.where((u >= 0.02) & (u < 0.035), rng.integers(1, 4, n), 0)
        planted_ins = {}
        if hap == 1:
            inside = snp_pos[(snp_pos >= pos) & (snp_pos < pos + n)]
            base[inside - pos] = [het_snps[int(p)] for p in inside]
            for p, k in het_del.items():
                if pos + 2 <= p and p + k + 2 < pos + n:
                    deleted[p + 1 - pos] += 1
                    deleted[p + 1 + k - pos] -= 1
            for p, codes in het_ins.items():
                if pos + 2 <= p < pos + n - 2:
                    ins[p - pos] = len(codes)
                    planted_ins[p - pos] = codes
        deleted = np.cumsum(deleted)[:n] > 0
        deleted[:2] = False
        deleted[-2:] = False
        m = ~deleted
        ins = np.where(m & np.roll(m, -1), ins, 0)
        ins[-1] = 0
        sub = (rng.random(n) < 0.03) & m
        base[sub] = rng.integers(0, 4, int(sub.sum()))
        has_ins = ins > 0
        total = n + int(has_ins.sum())
        at = np.arange(n) + np.concatenate([[0], np.cumsum(has_ins)[:-1]])
        ops = np.full(total, 1, np.int64)
        lens = np.ones(total, np.int64)
        ops[at] = np.where(m, 0, 2)
        lens[at[has_ins] + 1] = ins[has_ins]
        cut = np.flatnonzero(np.concatenate([[True], ops[1:] != ops[:-1]]))
        cigar = list(zip(ops[cut].tolist(), np.add.reduceat(lens, cut).tolist()))
        counts = np.where(m, 1 + ins, 0)
        seq = np.repeat(base, counts)
        firsts = np.concatenate([[0], np.cumsum(counts)[:-1]])
        extra = np.ones(len(seq), bool)
        extra[firsts[m]] = False
        seq[extra] = rng.integers(0, 4, int(extra.sum()))
        for off, codes in planted_ins.items():
            if m[off] and ins[off] == len(codes):
                seq[firsts[off] + 1:firsts[off] + 1 + len(codes)] = codes
        reads.append(dict(pos=pos, reverse=bool(rng.random() < 0.5), mapq=int(rng.choice([60, 60, 60, 20, 0], p=[.5, .2, .2, .07, .03])),
                          seq=LETTERS[seq].tobytes().decode(), qual=rng.integers(2, 40, len(seq)).astype(np.uint8), cigar=cigar,
                          name="%s%d" % (name, i)))
    return reads


def _clip(records, start, stop, include_supplementary, min_mapq):
    """restated_get_reads with the closed form of the clipping walk (bam_utils.closed_form_clip, which tests/test_bam_reader.py
    holds equal to the sequential restatement): the kept bases of a read are one stretch of it."""
    out = []
    for rec in records:
        flag = rec.get("flag", 16 if rec.get("reverse") else 0)
        end = rec["pos"] + max(1, bu.ref_length(rec["cigar"]))
        if not (rec["pos"] < stop and end > start) or flag & (0x200 | 0x400 | 0x100 | 0x4):
            continue
        if (not include_supplementary and flag & 0x800) or rec.get("mapq", 60) < min_mapq:
            continue
        c = bu.closed_form_clip(rec, start, stop)
        if c is None:
            continue
        a, b = c["first_idx"], c["first_idx"] + c["written"]
        out.append(dict(pos=c["pos"], seq=rec["seq"][a:b].upper(), qual=np.asarray(rec["qual"])[a:b], cigar=c["cigar"],
                        mapq=rec.get("mapq", 60), reverse=bool(flag & 0x10)))
    return out


def _reference_groups(ref_lib, contig, sequence, records, intervals, opts, max_reads=5000):
    """What generate_images must write for `intervals` of one contig: per interval the clipped reads (sampled down as
    AlignmentSummarizer.py:192-199 does) through the reference's own generate_summary."""
    want = {}
    for (start, end) in intervals:
        rs, re_ = max(0, start - 100), end + 100
        clipped = _clip(records, rs, re_, opts.include_supplementary, opts.min_mapq)
        if len(clipped) > max_reads:
            random = np.random.RandomState(2719747673)
            sample = []
            for i in range(len(clipped)):
                if len(sample) < max_reads:
                    sample.append(i)
                else:
                    j = random.randint(0, i + 1)
                    if j < max_reads:
                        sample[j] = i
            clipped = [clipped[i] for i in sample]
        if not clipped:
            continue
        params = pu.make_params(start, end, min_snp_baseq=opts.min_snp_baseq, min_indel_baseq=opts.min_indel_baseq)
        res = pu.run_variant(ref_lib, pu.FlatPileup(rs, re_, sequence[rs:re_ + 1], clipped), params, reference_impl=True)
        want["%s_%d_%d" % (contig, start, end)] = res
    return want


def _compare(got, want):
    n = 0
    assert sorted(got) == sorted(want), (sorted(set(got) ^ set(want))[:6])
    for name, res in want.items():
        g = got[name]
        assert g["candidates"] == list(res["candidates"]), name
        assert np.array_equal(g["positions"], res["positions"]) and np.array_equal(g["depths"], res["depths"].astype(np.uint8))
        assert np.array_equal(g["freq"].reshape(-1).astype(np.int64), res["candidate_frequency"].astype(np.int64))
        assert np.array_equal(g["images"], res["images"].astype(np.int64).astype(np.int8)), name
        n += len(res["candidates"])
    return n


def test_default_image_path_equals_the_reference_build_at_size(tmp_path, monkeypatch):
    from pepper_amd.variant.ImageGenerationUI import ImageGenerationUtils
    ref_lib = pu.load_reference_encoder()
    if ref_lib is None:
        need_reference_build("oracle/_ref/libref_variant_encoder.so")
    rng = np.random.default_rng(20260927)
    length_a, length_b = 231000, 31000
    draft_a, draft_b = rng.integers(0, 4, length_a).astype(np.uint8), rng.integers(0, 4, length_b).astype(np.uint8)
    snps = {int(p): int((draft_a[p] + 1 + rng.integers(3)) % 4) for p in rng.choice(np.arange(500, length_a - 500), 260, replace=False)}
    dels = {int(p): int(rng.integers(1, 12)) for p in rng.choice(np.arange(600, length_a - 600), 50, replace=False)}
    inss = {int(p): rng.integers(0, 4, int(rng.integers(1, 10))).astype(np.uint8) for p in rng.choice(np.arange(700, length_a - 700), 50, replace=False)}
    inss[99950] = rng.integers(0, 4, 14).astype(np.uint8)            # an allele of more than 8 bytes right at an interval edge
    reads_a = _fast_reads(rng, draft_a, 1150, (3000, 9000), snps, dels, inss, name="a")
    # a pile of 5 600 short reads on 120-128 kb: the second interval is sampled down to 5 000 reads on the host
    reads_a += _fast_reads(rng, draft_a, 5600, (250, 500), snps, dels, inss, first=120000, last=128000, name="d")
    reads_a.sort(key=lambda r: r["pos"])
    reads_a[40]["long_cigar"] = True                                  # its operations travel in the CG tag
    reads_a[41]["flag"] = 0x400 | (16 if reads_a[41]["reverse"] else 0)
    reads_b = _fast_reads(rng, draft_b, 160, (2000, 6000), {int(p): 0 for p in range(1000, 30000, 977)}, {}, {}, name="b")
    seq_a, seq_b = LETTERS[draft_a].tobytes().decode(), LETTERS[draft_b].tobytes().decode()
    bam, fa = str(tmp_path / "in.bam"), str(tmp_path / "ref.fa")
    bu.write_bam(bam, [("ctgA", length_a), ("ctgB", length_b)], {0: reads_a, 1: reads_b}, flush_every=23)
    with open(fa, "w") as fh:
        for n, s in (("ctgA", seq_a), ("ctgB", seq_b)):
            fh.write(">" + n + "\n" + "\n".join(s[i:i + 80] for i in range(0, len(s), 80)) + "\n")

    monkeypatch.delenv("PEPPER_AMD_PACKED_READS", raising=False)
    monkeypatch.delenv("PEPPER_AMD_DEVICE_INFLATE", raising=False)
    monkeypatch.delenv("PEPPER_AMD_DEVICE_WALK", raising=False)
    stats = {}
    opts = _options(bam, fa, str(tmp_path / "images"), None, 100000, 2, stage_seconds=stats)
    ImageGenerationUtils.generate_images(opts)
    assert stats.get("inflated_bytes", 0) > 0 and "bam_walk_device" in stats and "encode" in stats      # the default path really ran
    got = _groups(str(tmp_path / "images"))
    want = _reference_groups(ref_lib, "ctgA", seq_a, reads_a, [(0, 100000), (100000, 200000), (200000, length_a - 1)], opts)
    want.update(_reference_groups(ref_lib, "ctgB", seq_b, reads_b, [(0, length_b - 1)], opts))
    assert len(want) == 4
    assert _compare(got, want) > 600
    long_alleles = [c for g in got.values() for c in g["candidates"] if len(c) > 10]
    assert len(long_alleles) >= 1


# ---- a BAM written by tools/synth_bam, read back by this file's own record parser (zlib + struct) --------------------------
def _bai_linear(path, tid):
    data = open(path, "rb").read()
    assert data[:4] == b"BAI\1"
    n_ref, at = struct.unpack_from("<i", data, 4)[0], 8
    for t in range(n_ref):
        n_bin = struct.unpack_from("<i", data, at)[0]
        at += 4
        for _ in range(n_bin):
            _, n_chunk = struct.unpack_from("<Ii", data, at)
            at += 8 + 16 * n_chunk
        n_intv = struct.unpack_from("<i", data, at)[0]
        at += 4
        if t == tid:
            return np.frombuffer(data, "<u8", n_intv, at)
        at += 8 * n_intv
    raise KeyError(tid)


def _records_reaching(bam_path, linear, start, stop, max_read=14000):
    """Record dicts of tid 0 that overlap [start, stop), read from the file with zlib: from the linear index's offset of the
    window max_read bases in front of start (no record of the synthetic data is longer) up to the first record past stop."""
    win = max(0, (start - max_read) >> 14)
    while win < len(linear) and linear[win] == 0:
        win += 1
    voff = int(linear[min(win, len(linear) - 1)])
    coff, uoff = voff >> 16, voff & 0xffff
    out, buf = [], b""
    with open(bam_path, "rb") as fh:
        fh.seek(coff)
        skip = uoff
        while True:
            while len(buf) < 4 or len(buf) < 4 + struct.unpack_from("<i", buf, 0)[0]:
                head = fh.read(18)
                if len(head) < 18:
                    return out
                bsize = struct.unpack_from("<H", head, 16)[0] + 1
                body = fh.read(bsize - 18)
                data = zlib.decompress(body[:-8], -15)
                buf += data[skip:]
                skip = 0
            size = struct.unpack_from("<i", buf, 0)[0]
            rec, buf = buf[4:4 + size], buf[4 + size:]
            tid, pos, l_name, mapq, _bin, n_cig, flag, l_seq = struct.unpack_from("<iiBBHHHi", rec, 0)
            if tid != 0 or pos >= stop:
                return out
            o = 32 + l_name
            words = np.frombuffer(rec, "<u4", n_cig, o)
            cigar = [(int(w) & 15, int(w) >> 4) for w in words]
            o += 4 * n_cig
            packed = np.frombuffer(rec, np.uint8, (l_seq + 1) // 2, o)
            codes = np.empty(2 * len(packed), np.uint8)
            codes[0::2], codes[1::2] = packed >> 4, packed & 15
            seq = np.frombuffer(b"=ACMGRSVTWYHKDBN", np.uint8)[codes[:l_seq]].tobytes().decode()
            o += (l_seq + 1) // 2
            qual = np.frombuffer(rec, np.uint8, l_seq, o)
            if pos + max(1, bu.ref_length(cigar)) > start:
                out.append(dict(pos=pos, cigar=cigar, seq=seq, qual=qual, mapq=mapq, flag=flag, reverse=bool(flag & 16)))


def test_sampled_intervals_of_a_bench_shaped_job_equal_the_reference_build(tmp_path):
    """tools/synth_bam's data (the shape bench.py's make_images leg runs on: 60x of 4-12 kb reads, mapq-0 / duplicate / secondary /
    supplementary records among them), 8 Mb here: all 80 intervals through default generate_images with one worker per CPU, 10 of
    them -- the first, the last and 8 drawn at random -- against the reference build."""
    import json
    import subprocess
    from pepper_amd import build
    from pepper_amd.variant.ImageGenerationUI import ImageGenerationUtils
    from pepper_amd.variant.fasta import FASTA_handler
    ref_lib = pu.load_reference_encoder()
    tool = build.build_tools()
    if ref_lib is None:
        need_reference_build("oracle/_ref/libref_variant_encoder.so")
    assert tool is not None, "tools/synth_bam did not build"
    work = str(tmp_path)
    info = json.loads(subprocess.run([tool, work, "8000000", "60", "77"], check=True, capture_output=True, text=True).stdout)
    bam, fa = os.path.join(work, "reads.bam"), os.path.join(work, "draft.fa")
    opts = _options(bam, fa, os.path.join(work, "images"), None, 100000, 8)
    ImageGenerationUtils.generate_images(opts)
    got = _groups(os.path.join(work, "images"))
    fasta = FASTA_handler(fa)
    contig = fasta.get_chromosome_names()[0]
    length = fasta.get_chromosome_sequence_length(contig)
    sequence = fasta.get_reference_sequence(contig, 0, length)
    assert length == info["genome_bases"]
    linear = _bai_linear(bam + ".bai", 0)
    rng = np.random.default_rng(5)
    starts = sorted({0, (length - 1) // 100000 * 100000} | {int(s) * 100000 for s in rng.choice(np.arange(1, length // 100000), 8, replace=False)})
    checked = 0
    for start in starts:
        end = min(length - 1, start + 100000)
        records = _records_reaching(bam, linear, max(0, start - 100), end + 101)
        want = _reference_groups(ref_lib, contig, sequence, records, [(start, end)], opts)
        name = "%s_%d_%d" % (contig, start, end)
        assert name in want and name in got
        checked += _compare({name: got[name]}, want)
    assert checked > 1500
