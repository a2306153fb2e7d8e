"""The packed form of image generation (pa_bam_pack_regions -> pa_encoder_stage_packed: reads cross PCIe as BAM stores them and
are clipped + decoded by unpack_clip_kernel) against the host-clipped form (BAM_handler.get_reads -> pa_encoder_stage_batch),
which tests/test_gpu_encoder.py holds bit-exact to the reference's own C++ build: every dataset of every summary group equal."""
import os
from types import SimpleNamespace

import numpy as np
import pytest

import bam_utils as bu
import pileup_utils as pu
from pepper_amd import h5

pytestmark = pytest.mark.gpu


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
                                 candidates=f[g + "candidates"].tolist(), freq=f[g + "candidate_frequency"], contigs=f[g + "contigs"].tolist())
    return out


def _same(a, b):
    assert sorted(a) == sorted(b)
    n = 0
    for name in a:
        for key in a[name]:
            x, y = a[name][key], b[name][key]
            assert (x == y) if isinstance(x, list) else (x.dtype == y.dtype and np.array_equal(x, y)), (name, key)
        n += len(a[name]["candidates"])
    return n


def _both(monkeypatch, tmp_path, tag, make_options, **env):
    from pepper_amd.variant.ImageGenerationUI import ImageGenerationUtils
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    monkeypatch.setenv("PEPPER_AMD_PACKED_READS", "1")
    stats = {}
    ImageGenerationUtils.generate_images(make_options(str(tmp_path / (tag + "_packed")), stage_seconds=stats))
    monkeypatch.setenv("PEPPER_AMD_PACKED_READS", "0")
    ImageGenerationUtils.generate_images(make_options(str(tmp_path / (tag + "_host"))))
    assert "encode" in stats                       # the packed path really ran
    return _groups(str(tmp_path / (tag + "_packed"))), _groups(str(tmp_path / (tag + "_host")))


def _write(tmp_path, refs, reads_by_tid, **kw):
    bam, fa = str(tmp_path / "in.bam"), str(tmp_path / "ref.fa")
    bu.write_bam(bam, [(n, len(s)) for n, s in refs], reads_by_tid, **kw)
    with open(fa, "w") as fh:
        for n, s in refs:
            fh.write(">" + n + "\n" + "\n".join(s[i:i + 60] for i in range(0, len(s), 60)) + "\n")
    return bam, fa


def test_packed_equals_host_clipped(tmp_path, monkeypatch):
    """Two contigs; soft / hard clips, reads over every interval edge, deletions of 40 and inserts of 14 and 20 bases (alleles of
    more than 8 bytes: the pool; two that agree on their first 8), mapq-0 reads, flagged records, long CIGARs in the CG tag, an
    interval without reads, intervals of 1.7 kb so that a read reaches three of them."""
    rng = np.random.default_rng(911)
    ref = pu.random_reference(rng, 24000, n_frac=0.002)
    sites = {int(p): ("ACGT"[(("ACGT".index(ref[p]) if ref[p] in "ACGT" else 0) + 1) % 4], 0.5) for p in rng.choice(np.arange(300, 19000), 60, replace=False)}
    indels = {1111: ("I", "ACG", 0.6), 2222: ("D", 4, 0.7), 3333: ("D", 40, 0.5), 5100: ("I", "ACGTACGTTTGACA", 0.35),
              6800: ("I", "GGGTTTAACCGGTTAACCGG", 0.5), 1699: ("I", "TT", 0.6), 1700: ("D", 3, 0.6), 3400: ("I", "CAG", 0.6)}
    reads = pu.simulate_reads(rng, ref[:20000], 0, n_reads=1500, read_len=(300, 4000), snp_sites=sites, indel_sites=indels,
                              clip_rate=0.4, mapq_zero_rate=0.05, long_indel_rate=0.02)
    reads = [r for r in reads if not any(op in (3, 6) for op, _ in r["cigar"])]
    # a second long insert at 5100 that shares its first 8 bytes with the planted one (anchor base + 7): ordered through the pool
    twin = "ACGTACGAAAAAAA"
    for r in reads:
        if r["pos"] < 5000 and rng.random() < 0.5:
            at, k = r["pos"], 0
            for i, (op, n) in enumerate(r["cigar"]):
                if op == 1 and n == 14 and at == 5101:
                    r["seq"] = r["seq"][:k] + twin + r["seq"][k + 14:]
                if op in (0, 7, 8, 2):
                    at += n
                if op in (0, 7, 8, 1, 4):
                    k += n
    other = pu.simulate_reads(rng, ref[:5000], 0, n_reads=150, read_len=(200, 900))
    other = [r for r in other if not any(op in (3, 6) for op, _ in r["cigar"])]
    for i, r in enumerate(reads):
        r["name"] = "r%d" % i
        r["flag"] = (16 if r["reverse"] else 0) | int(rng.choice([0, 0x800, 0x100, 0x400, 0x200], p=[.9, .04, .03, .02, .01]))
        r["long_cigar"] = i % 7 == 3
    for i, r in enumerate(other):
        r["name"] = "o%d" % i
    bam, fa = _write(tmp_path, [("ctgA", ref[:5000]), ("ctgB", ref)], {0: other, 1: reads}, flush_every=41)
    for tag, region, size, threads, over in (("b", "ctgB", 1700, 2, {}), ("all", None, 5000, 3, {"include_supplementary": True, "min_mapq": 0}),
                                             ("one", "ctgB:1000-9000", 100000, 1, {"min_snp_baseq": 10, "min_indel_baseq": 12})):
        got, want = _both(monkeypatch, tmp_path, tag, lambda out, **kw: _options(bam, fa, out, region, size, threads, **over, **kw))
        assert _same(got, want) > 40
        if tag == "b":
            assert not any(name.startswith("ctgB_22100") for name in want)      # no read reaches the contig's tail
            long_alleles = [c[0] for g in got.values() for c in g["candidates"] if len(c[0]) > 10]
            assert any(a.startswith("2") and len(a) == 16 for a in long_alleles) and len(set(long_alleles)) >= 3


def test_small_arena_and_sampled_intervals(tmp_path, monkeypatch):
    """An arena of 1 MB cuts the runs of intervals into several packer calls (and one interval's reads outgrow it: host-clipped
    form for that interval); a downsample rate below 1 sends every interval through the host-clipped form -- same files."""
    rng = np.random.default_rng(912)
    ref = pu.random_reference(rng, 30000)
    sites = {int(p): ("ACGT"[("ACGT".index(ref[p]) + 2) % 4], 0.5) for p in rng.choice(np.arange(300, 29000), 80, replace=False)}
    reads = pu.simulate_reads(rng, ref, 0, n_reads=1200, read_len=(500, 3000), snp_sites=sites)
    reads = [r for r in reads if not any(op in (3, 6) for op, _ in r["cigar"])]
    # a pile of 700 more reads on 12-14 kb: that interval alone is more than the arena
    deep = pu.simulate_reads(rng, ref[12000:14500], 12000, n_reads=900, read_len=(1500, 2400), snp_sites=sites)
    reads = sorted(reads + [r for r in deep if not any(op in (3, 6) for op, _ in r["cigar"])], key=lambda r: r["pos"])
    for i, r in enumerate(reads):
        r["name"] = "r%d" % i
    bam, fa = _write(tmp_path, [("ctg", ref)], {0: reads}, flush_every=29)
    got, want = _both(monkeypatch, tmp_path, "small", lambda out, **kw: _options(bam, fa, out, "ctg", 2000, 2, **kw), PEPPER_AMD_ARENA_MB="1")
    assert _same(got, want) > 60
    from pepper_amd.variant.ImageGenerationUI import ImageGenerationUtils
    monkeypatch.setenv("PEPPER_AMD_PACKED_READS", "1")
    ImageGenerationUtils.generate_images(_options(bam, fa, str(tmp_path / "ds_a"), "ctg", 6000, 1, downsample_rate=0.5))
    monkeypatch.setenv("PEPPER_AMD_PACKED_READS", "0")
    ImageGenerationUtils.generate_images(_options(bam, fa, str(tmp_path / "ds_b"), "ctg", 6000, 1, downsample_rate=0.5))
    assert _same(_groups(str(tmp_path / "ds_a")), _groups(str(tmp_path / "ds_b"))) > 10


def test_packed_form_refuses_what_it_cannot_walk(tmp_path):
    """A record whose CIGAR walks over more bases than it holds fails the run (as get_reads fails the query); an operation of
    2^24 bases is PA_ERR_UNSUPPORTED, which image generation answers with the host-clipped form."""
    from pepper_amd import _lib
    from pepper_amd.variant.bam import BAM_handler
    from pepper_amd.variant.PEPPER_VARIANT import PackedEncoder
    ok = dict(name="ok", pos=100, cigar=[(0, 50)], seq="ACGTA" * 10, qual=[30] * 50)
    short = dict(name="short", pos=120, cigar=[(0, 80)], seq="ACGT" * 5, qual=[30] * 20)
    huge = dict(name="huge", pos=130, cigar=[(0, 10), (3, 1 << 24), (0, 10)], seq="ACGT" * 5, qual=[30] * 20)
    params = (1, 1, 0.1, 0.15, 0.15, 3, 0.1, 0.12, 2, False)
    for bad, code in ((short, 1), (huge, _lib.PA_ERR_UNSUPPORTED)):
        path = str(tmp_path / (bad["name"] + ".bam"))
        bu.write_bam(path, [("ctg", 1 << 26)], {0: [ok, bad]})
        enc = PackedEncoder(0, arena_bytes=1 << 20, max_reads=64, max_pairs=64)
        n_done, region_pairs, counts = enc.pack(BAM_handler(path), "ctg", [0], [1000], False, 0)
        assert n_done == 1 and counts[0] == 2
        with pytest.raises(_lib.PepperAmdError) as err:
            enc.encode([(0, 1000)], ["A" * 1001], region_pairs, counts, params, [(100, 900)])
        assert err.value.code == code
        enc.close()
