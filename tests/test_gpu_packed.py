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
    """The packed form twice -- BGZF members inflated on the device (the default with an index) and on the host -- and the
    host-clipped form: the first two must agree with each other here, the caller compares with the third."""
    from pepper_amd.variant.ImageGenerationUI import ImageGenerationUtils
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    monkeypatch.setenv("PEPPER_AMD_PACKED_READS", "1")
    monkeypatch.setenv("PEPPER_AMD_DEVICE_INFLATE", "1")
    stats = {}
    ImageGenerationUtils.generate_images(make_options(str(tmp_path / (tag + "_packed")), stage_seconds=stats))
    monkeypatch.setenv("PEPPER_AMD_DEVICE_INFLATE", "0")
    stats_host = {}
    ImageGenerationUtils.generate_images(make_options(str(tmp_path / (tag + "_packed_host")), stage_seconds=stats_host))
    monkeypatch.setenv("PEPPER_AMD_PACKED_READS", "0")
    ImageGenerationUtils.generate_images(make_options(str(tmp_path / (tag + "_host"))))
    assert "encode" in stats and "encode" in stats_host                       # the packed path really ran
    assert "bam_inflate_device" in stats and stats.get("inflated_bytes", 0) > 0 and "bam_inflate_device" not in stats_host
    assert "bam_pack" in stats_host
    packed = _groups(str(tmp_path / (tag + "_packed")))
    _same(packed, _groups(str(tmp_path / (tag + "_packed_host"))))
    return packed, _groups(str(tmp_path / (tag + "_host")))


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
        # the same records left in place in the span inflated on the device
        n_done, region_pairs, counts = enc.pack_device(BAM_handler(path), "ctg", [0], [1000], False, 0)
        assert n_done == 1 and counts[0] == 2
        with pytest.raises(_lib.PepperAmdError) as err:
            enc.encode([(0, 1000)], ["A" * 1001], region_pairs, counts, params, [(100, 900)], resident=True)
        assert err.value.code == code
        enc.close()


def test_device_inflated_span_gives_the_host_packers_summaries(tmp_path):
    """PackedEncoder.pack_device (span read, inflate on the device, records walked in place) against PackedEncoder.pack for the
    same run of regions: the same tables (slices at any byte offset instead of copied to word boundaries) and the same encoder
    output; a BAM without an index and a record with its CIGAR in the CG tag send the batch to the host packer (None)."""
    from pepper_amd.variant.bam import BAM_handler
    from pepper_amd.variant.PEPPER_VARIANT import PackedEncoder
    rng = np.random.default_rng(913)
    ref = pu.random_reference(rng, 60000)
    sites = {int(p): ("ACGT"[("ACGT".index(ref[p]) + 1) % 4], 0.5) for p in rng.choice(np.arange(300, 59000), 150, replace=False)}
    reads = pu.simulate_reads(rng, ref, 0, n_reads=2500, read_len=(500, 6000), snp_sites=sites,
                              indel_sites={20000: ("I", "ACGTACGTTTGACA", 0.5), 30000: ("D", 12, 0.5)}, clip_rate=0.3)
    reads = [r for r in reads if not any(op in (3, 6) for op, _ in r["cigar"])]
    for i, r in enumerate(reads):
        r["name"] = "read_%05d" % i
    bam, _ = _write(tmp_path, [("ctg", ref)], {0: reads}, flush_every=19)
    params = (1, 1, 0.1, 0.15, 0.15, 3, 0.1, 0.12, 2, False)
    edges = list(range(5000, 56000, 5000))
    starts, stops = [a - 100 for a in edges[:-1]], [b + 100 for b in edges[1:]]
    regions = list(zip(starts, stops))
    refs = [ref[a:b + 1] for a, b in regions]
    cands = list(zip(edges[:-1], edges[1:]))
    enc = PackedEncoder(0, arena_bytes=64 << 20)
    handler = BAM_handler(bam)
    n_done, rp_h, counts_h = enc.pack(handler, "ctg", starts, stops, False, 1)
    assert n_done == len(starts)
    want, live_h = enc.encode(regions, refs, rp_h, counts_h, params, cands)
    laps = {}
    n_done, rp_d, counts_d = enc.pack_device(handler, "ctg", starts, stops, False, 1, laps=laps)
    assert n_done == len(starts) and counts_d[:2] == counts_h[:2] and rp_d.tolist() == rp_h.tolist()
    assert {"bam_span_read", "bam_inflate_device", "bam_walk"} <= set(laps) and enc.inflated_bytes == counts_d[2] > 0
    got, live_d = enc.encode(regions, refs, rp_d, counts_d, params, cands, resident=True)
    assert live_d.tolist() == live_h.tolist() and sum(len(g["candidates"]) for g in got) > 100
    for g, w in zip(got, want):
        assert sorted(g) == sorted(w)
        for key in g:
            assert (g[key] == w[key]) if isinstance(g[key], list) else np.array_equal(g[key], w[key]), key
    # a resident span is used once: the next staging without one must bring its arena
    from pepper_amd import _lib
    n_done, rp_h, counts_h = enc.pack(handler, "ctg", starts[:2], stops[:2], False, 1)
    again, _ = enc.encode(regions[:2], refs[:2], rp_h, counts_h, params, cands[:2])
    with pytest.raises(_lib.PepperAmdError, match="resident"):
        enc.encode(regions[:2], refs[:2], rp_h, counts_h, params, cands[:2], resident=True)
    for key in again[0]:
        assert (again[0][key] == want[0][key]) if isinstance(want[0][key], list) else np.array_equal(again[0][key], want[0][key])
    # no index: the host packer's batch
    os.remove(bam + ".bai")
    assert enc.pack_device(BAM_handler(bam), "ctg", starts, stops, False, 1) is None
    enc.close()
    cg = [dict(reads[k], long_cigar=True) if k == 40 else reads[k] for k in range(len(reads))]
    bam2 = str(tmp_path / "cg.bam")
    bu.write_bam(bam2, [("ctg", len(ref))], {0: cg})
    enc = PackedEncoder(0, arena_bytes=64 << 20)
    assert enc.pack_device(BAM_handler(bam2), "ctg", starts, stops, False, 1) is None
    enc.close()



def test_device_record_walk_equals_the_host_walk(tmp_path, monkeypatch):
    """pack_device with the records read out on the device (pa_encoder_walk_records + pa_bam_pack_headers, the default) and with
    the span walked on the host (PEPPER_AMD_DEVICE_WALK=0): the same tables; a window with more records than a lane has
    slots (3 000 short reads inside 16 kb) takes the host walk by itself."""
    from pepper_amd.variant.bam import BAM_handler
    from pepper_amd.variant.PEPPER_VARIANT import PackedEncoder
    rng = np.random.default_rng(914)
    ref = pu.random_reference(rng, 80000)
    reads = pu.simulate_reads(rng, ref, 0, n_reads=2500, read_len=(500, 7000), clip_rate=0.3, mapq_zero_rate=0.05)
    short = pu.simulate_reads(rng, ref[40000:52000], 40000, n_reads=3000, read_len=(80, 200))
    for tag, recs in (("long", reads), ("short", sorted(reads + short, key=lambda r: r["pos"]))):
        recs = [r for r in recs if not any(op in (3, 6) for op, _ in r["cigar"])]
        for i, r in enumerate(recs):
            r["name"] = "q%d" % i
        bam = str(tmp_path / (tag + ".bam"))
        bu.write_bam(bam, [("ctg", len(ref))], {0: recs}, flush_every=53)
        edges = list(range(5000, 76000, 7000))
        starts, stops = [a - 100 for a in edges[:-1]], [b + 100 for b in edges[1:]]
        got = {}
        for mode in ("1", "0"):
            monkeypatch.setenv("PEPPER_AMD_DEVICE_WALK", mode)
            enc = PackedEncoder(0, arena_bytes=64 << 20)
            laps = {}
            n_done, rp, counts = enc.pack_device(BAM_handler(bam), "ctg", starts, stops, False, 1, laps=laps)
            assert ("bam_walk_device" in laps) == (mode == "1")
            got[mode] = (n_done, rp.tolist(), counts, enc.reads[:counts[0]].tobytes(), enc.pair_read[:counts[1]].tolist())
            enc.close()
        assert got["1"] == got["0"] and got["1"][0] == len(starts) and got["1"][2][0] > 500


def test_workers_over_device_ids_write_the_same_files(tmp_path, monkeypatch):
    """device_ids "0,0": two workers, each on the device the map gives it (worker t -> device_ids[t % n]) -- the files are those
    of the run without device_ids."""
    from pepper_amd.variant.ImageGenerationUI import ImageGenerationUtils
    rng = np.random.default_rng(913)
    ref = pu.random_reference(rng, 16000)
    sites = {int(p): ("ACGT"[("ACGT".index(ref[p]) + 1) % 4], 0.5) for p in rng.choice(np.arange(300, 15000), 40, replace=False)}
    reads = pu.simulate_reads(rng, ref, 0, n_reads=500, read_len=(400, 2500), snp_sites=sites)
    reads = [r for r in reads if not any(op in (3, 6) for op, _ in r["cigar"])]
    for i, r in enumerate(reads):
        r["name"] = "r%d" % i
    bam, fa = _write(tmp_path, [("ctg", ref)], {0: reads}, flush_every=31)
    ImageGenerationUtils.generate_images(_options(bam, fa, str(tmp_path / "one"), None, 2000, 1))
    ImageGenerationUtils.generate_images(_options(bam, fa, str(tmp_path / "two"), None, 2000, 2, device_ids="0,0"))
    assert _same(_groups(str(tmp_path / "two")), _groups(str(tmp_path / "one"))) > 20
