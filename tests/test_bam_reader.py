"""BAM reader (pepper_amd/csrc/bamio.cpp behind include/pepper_amd_io.h).  htslib and
therefore the reference's own reader are absent: PARITY UNPINNED.  The checker is a Python restatement of
bam_handler.cpp:115-451 (tests/bam_utils.py) on BAM files this test writes."""
import struct

import numpy as np
import pytest

import bam_utils as bu
import pileup_utils as pu
from pepper_amd.variant.bam import BAM_handler, BamError


def compare(readset, want):
    assert len(readset) == len(want)
    for got, w in zip(readset, want):
        assert got.query_name == w["name"]
        assert (got.pos, got.pos_end) == (w["pos"], w["pos_end"])
        assert got.sequence == w["seq"] and got.base_qualities == w["qual"]
        assert [(c.cigar_op, c.cigar_len) for c in got.cigar_tuples] == w["cigar"]
        assert got.mapping_quality == w["mapq"] and got.flags.is_reverse == w["reverse"] and got.hp_tag == w["hp"]


def hp_aux(value, kind="C"):
    fmt = {"c": "<b", "C": "<B", "s": "<h", "S": "<H", "i": "<i", "I": "<I"}[kind]
    return b"HP" + kind.encode() + struct.pack(fmt, value)


@pytest.mark.parametrize("with_index", [True, False])
def test_random_pileup_regions(tmp_path, with_index):
    rng = np.random.default_rng(17)
    ref = pu.random_reference(rng, 60000)
    reads = pu.simulate_reads(rng, ref, 0, n_reads=900, read_len=(300, 6000), clip_rate=0.4, mapq_zero_rate=0.1)
    other = pu.simulate_reads(rng, ref[:9000], 0, n_reads=60, read_len=(200, 900))
    # (the simulator emits read bases for N / P operations to exercise an encoder quirk; BAM records never do)
    reads = [r for r in reads if not any(op in (3, 6) for op, _ in r["cigar"])]
    other = [r for r in other if not any(op in (3, 6) for op, _ in r["cigar"])]
    for i, r in enumerate(reads):
        r["name"] = "read%d" % i
        r["flag"] = (16 if r["reverse"] else 0) | int(rng.choice([0, 0, 0, 0, 0x800, 0x100, 0x400, 0x200, 0x4], p=[.8, .04, .04, .02, .04, .02, .02, .01, .01]))
        if rng.random() < 0.3:
            r["hp"] = int(rng.integers(1, 3))
            r["aux"] = b"NMC\x05" + hp_aux(r["hp"], str(rng.choice(list("cCsSiI")))) + b"RGZgroup1\0" + b"XBBs\x02\0\0\0\x01\0\x02\0"
    for i, r in enumerate(other):
        r["name"] = "o%d" % i
    path = str(tmp_path / "x.bam")
    bu.write_bam(path, [("chrA", 9000), ("chrB", 60000), ("chrC", 100)], {0: other, 1: reads},
                 header_text="@HD\tVN:1.6\tSO:coordinate\n@RG\tID:g1\tSM:HG003\tPL:ONT\n@RG\tID:g2\tSM:other\n",
                 with_index=with_index, flush_every=37)
    bam = BAM_handler(path)
    assert bam.has_index() == with_index
    assert bam.get_chromosome_sequence_names() == ["chrA", "chrB", "chrC"]
    assert [s.sequence_length for s in bam.get_chromosome_sequence_names_with_length()] == [9000, 60000, 100]
    assert bam.get_sample_names() == {"HG003", "other"}
    regions = [(0, 60000), (0, 1), (59990, 60000), (16380, 16390), (20000, 20100), (32768, 49152), (100, 2100)]
    regions += [tuple(sorted(rng.integers(0, 60000, size=2).tolist())) for _ in range(12)]
    for start, stop in regions:
        if stop == start:
            stop += 1
        for include_supp, min_mapq in ((False, 1), (True, 0), (False, 30)):
            got = bam.get_reads("chrB", start, stop, include_supp, min_mapq, 1)
            compare(got, bu.restated_get_reads(reads, start, stop, include_supp, min_mapq))
    # many short regions: the cursor fast-forward over the operations in front of a region ends at every kind of operation
    # (a match run cut by `start`, an insert or a soft clip sitting at it, a deletion spanning it)
    for start in rng.integers(0, 59990, size=70).tolist():
        stop = start + int(rng.integers(1, 300))
        compare(bam.get_reads("chrB", start, stop, True, 0, 1), bu.restated_get_reads(reads, start, stop, True, 0))
    compare(bam.get_reads("chrA", 100, 5000, True, 0, 0), bu.restated_get_reads(other, 100, 5000, True, 0))
    assert len(bam.get_reads("chrC", 0, 100, True, 0, 0)) == 0
    with pytest.raises(BamError):
        bam.get_reads("chrZ", 0, 10, True, 0, 0)
    bam.close()


def test_hand_made_clipping_cases(tmp_path):
    """Each cigar pattern the clipping loop treats specially, against expectations worked out by hand."""
    M, I, D, N, S, H, P, EQ, X = range(9)
    q = lambda n: list(range(10, 10 + n))
    recs = [
        # 0: starts left of the window: first M is cut at start; insertion right at the cut is kept (anchored)
        dict(name="a", pos=90, cigar=[(M, 15), (I, 2), (M, 10)], seq="A" * 15 + "GG" + "C" * 10, qual=q(27)),
        # 1: begins with soft clip + insertion before any aligned base inside the window: both dropped
        dict(name="b", pos=100, cigar=[(S, 3), (I, 2), (M, 5)], seq="TTTGGACGTA", qual=q(10)),
        # 2: deletion running past stop is cut; trailing M past stop dropped
        dict(name="c", pos=115, cigar=[(EQ, 3), (D, 10), (X, 4)], seq="ACGTTTT", qual=q(7)),
        # 3: N skips, hard clip ignored, lower-case never happens in BAM (4-bit codes) but N base kept
        dict(name="d", pos=105, cigar=[(H, 5), (M, 2), (N, 4), (M, 2), (S, 2)], seq="ANGTCC", qual=q(6)),
        # 4: entirely right of stop by position: not returned; 5: left of start entirely: not returned
        dict(name="e", pos=121, cigar=[(M, 5)], seq="AAAAA", qual=q(5)),
    ]
    left = dict(name="l", pos=50, cigar=[(M, 40)], seq="A" * 40, qual=q(40))
    path = str(tmp_path / "h.bam")
    bu.write_bam(path, [("c", 1000)], {0: [left] + recs})
    bam = BAM_handler(path)
    got = bam.get_reads("c", 100, 120, False, 0, 0)
    want = [
        dict(name="a", pos=100, pos_end=115, seq="A" * 5 + "GG" + "C" * 10, qual=q(27)[10:], cigar=[(M, 5), (I, 2), (M, 10)]),
        dict(name="b", pos=100, pos_end=105, seq="ACGTA", qual=q(10)[5:], cigar=[(M, 5)]),
        dict(name="c", pos=115, pos_end=121, seq="ACG", qual=q(7)[:3], cigar=[(EQ, 3), (D, 3)]),
        dict(name="d", pos=105, pos_end=113, seq="ANGTCC", qual=q(6), cigar=[(M, 2), (N, 4), (M, 2), (S, 2)]),
    ]
    for w in want:
        w.update(mapq=60, reverse=False, hp=0)
    compare(got, want)
    # stop is inclusive for bases (pos <= stop) but the iterator is half-open (pos < stop): a read starting AT stop
    # is not fetched, a read covering stop contributes the base at stop
    assert [r.query_name for r in bam.get_reads("c", 89, 90, False, 0, 0)] == ["l", "a"][:1]
    one = bam.get_reads("c", 89, 90, False, 0, 0)[0]
    assert (one.pos, one.pos_end, one.sequence) == (89, 90, "A")
    # the structure-of-arrays form feeds the encoder directly; take() reorders consistently
    flat = got.as_pileup()
    assert flat["n_reads"] == 4 and flat["seq_offset"].tolist() == [0, 17, 22, 25, 31]
    sub = got.take([3, 0])
    assert [r.query_name for r in sub] == ["d", "a"] and sub[1].sequence == want[0]["seq"]
    assert sub.as_pileup()["cigar_offset"].tolist() == [0, 4, 7]


def test_block_cache_eviction_and_buffer_reuse(tmp_path):
    """More BGZF blocks than the reader caches (256): region queries in random order keep evicting blocks and reusing
    their buffers; every answer must still equal the restatement."""
    rng = np.random.default_rng(23)
    ref = pu.random_reference(rng, 40000)
    reads = pu.simulate_reads(rng, ref, 0, n_reads=1400, read_len=(300, 2500), clip_rate=0.2)
    reads = [r for r in reads if not any(op in (3, 6) for op, _ in r["cigar"])]
    for i, r in enumerate(reads):
        r["name"] = "r%d" % i
    path = str(tmp_path / "many_blocks.bam")
    bu.write_bam(path, [("ctg", 40000)], {0: reads}, flush_every=2)          # a block every two records: > 500 blocks
    bam = BAM_handler(path)
    starts = rng.integers(0, 39000, size=60).tolist() + list(range(0, 39000, 3000)) + list(range(38000, 0, -3500))
    for start in starts:
        stop = int(start) + int(rng.integers(50, 3000))
        compare(bam.get_reads("ctg", int(start), stop, False, 0, 0), bu.restated_get_reads(reads, int(start), stop, False, 0))
    bam.close()


def test_long_cigar_in_cg_tag(tmp_path):
    """Records whose CIGAR lives in the CG:B,I tag behind the <l_seq>S<ref_len>N placeholder (reads with more than 65535
    operations; htslib swaps the real CIGAR in transparently): same reads as with the CIGAR in the core field; a
    placeholder without a usable tag keeps its core CIGAR, as htslib's bam_tag2cigar does (sam.c: no CG tag, a tag that is
    not B,I / B,i, or one with fewer operations than the core field -> the record is left alone)."""
    rng = np.random.default_rng(31)
    ref = pu.random_reference(rng, 20000)
    reads = pu.simulate_reads(rng, ref, 0, n_reads=120, read_len=(300, 4000), clip_rate=0.3)
    reads = [r for r in reads if not any(op in (3, 6) for op, _ in r["cigar"])]
    for i, r in enumerate(reads):
        r["name"] = "r%d" % i
        r["long_cigar"] = i % 3 != 1
        if i % 4 == 0:
            r["hp"] = 2
            r["aux"] = hp_aux(2) + b"XBBs\x02\0\0\0\x01\0\x02\0"     # tags in front of CG are walked over
        if i % 8 == 2:
            r["aux"] = r.get("aux", b"") + b"XDd" + np.float64(1.5).tobytes()   # including an 8-byte 'd' value
    path = str(tmp_path / "cg.bam")
    bu.write_bam(path, [("ctg", 20000)], {0: reads}, flush_every=11)
    bam = BAM_handler(path)
    for start, stop in ((0, 20000), (5000, 5100), (9000, 13000), (19990, 20000)):
        compare(bam.get_reads("ctg", start, stop, False, 0, 0), bu.restated_get_reads(reads, start, stop, False, 0))
    bam.close()
    # one real read with more than 65535 operations: 35000 x (1M 1I) + 1M
    n = 35000
    big = dict(name="ultra", pos=10, cigar=[(0, 1), (1, 1)] * n + [(0, 1)], seq="AC" * n + "G", qual=[20] * (2 * n + 1),
               long_cigar=True)
    path2 = str(tmp_path / "ultra.bam")
    bu.write_bam(path2, [("ctg", 40000)], {0: [big]})
    bam = BAM_handler(path2)
    compare(bam.get_reads("ctg", 100, 30000, False, 0, 0), bu.restated_get_reads([big], 100, 30000, False, 0))
    bam.close()
    # the placeholder with no CG tag behind it: the read keeps <l_seq>S<ref_len>N
    bad = dict(reads[0], long_cigar=True, drop_cg=True)
    path3 = str(tmp_path / "nocg.bam")
    bu.write_bam(path3, [("ctg", 20000)], {0: [bad]})
    ref_len = sum(n for op, n in bad["cigar"] if op in (0, 2, 3, 7, 8))
    as_stored = dict(bad, cigar=[(4, len(bad["seq"])), (3, ref_len)])
    bam = BAM_handler(path3)
    compare(bam.get_reads("ctg", 0, 20000, False, 0, 0), bu.restated_get_reads([as_stored], 0, 20000, False, 0))
    bam.close()


def test_corrupt_block_is_an_error_not_end_of_file(tmp_path):
    """A damaged BGZF block on a record boundary must fail the query (it used to read as end of file and
    return a partial read set)."""
    rng = np.random.default_rng(37)
    ref = pu.random_reference(rng, 30000)
    reads = pu.simulate_reads(rng, ref, 0, n_reads=300, read_len=(300, 2000))
    reads = [r for r in reads if not any(op in (3, 6) for op, _ in r["cigar"])]
    path = str(tmp_path / "c.bam")
    bu.write_bam(path, [("ctg", 30000)], {0: reads}, flush_every=5)       # every block ends on a record boundary
    raw = bytearray(open(path, "rb").read())
    # walk the blocks and break the magic of one in the middle of the file
    offs, p = [], 0
    while p < len(raw):
        offs.append(p)
        p += (raw[p + 16] | (raw[p + 17] << 8)) + 1
    victim = offs[len(offs) // 2]
    raw[victim] ^= 0xff
    open(path, "wb").write(bytes(raw))
    bam = BAM_handler(path)
    with pytest.raises(BamError):
        bam.get_reads("ctg", 0, 30000, False, 0, 0)
    bam.close()


def test_flipped_payload_bit_fails_the_crc_on_the_host_paths(tmp_path):
    """The host reader (Bgzf::read_block) and pa_bgzf_inflate_host check a member's CRC-32 as htslib's inflate_block does: a
    BAM whose blocks are stored (level 0) with one payload bit flipped inflates structurally and must fail the query."""
    import struct
    import zlib
    from pepper_amd.bgzf import BgzfError, block_table, inflate_host
    rng = np.random.default_rng(38)
    data = bytes(rng.integers(0, 256, 5000, dtype=np.uint8))

    def member(payload, level):
        c = zlib.compressobj(level, zlib.DEFLATED, -15)
        body = c.compress(payload) + c.flush()
        return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(body) + 25) + body +
                struct.pack("<II", zlib.crc32(payload), len(payload)))
    good = member(data, 0) + member(data[:777], 6)
    assert inflate_host(good, block_table(good), 2).tobytes() == data + data[:777]
    bad = bytearray(good)
    bad[18 + 5 + 2000] ^= 0x04
    with pytest.raises(BgzfError):
        inflate_host(bytes(bad), block_table(bytes(bad)), 2)
    # the BAM reader: flip a base inside a record's block; the query fails instead of returning a different base
    ref = pu.random_reference(rng, 20000)
    reads = pu.simulate_reads(rng, ref, 0, n_reads=200, read_len=(300, 1500))
    reads = [r for r in reads if not any(op in (3, 6) for op, _ in r["cigar"])]
    path = str(tmp_path / "crc.bam")
    bu.write_bam(path, [("ctg", 20000)], {0: reads}, flush_every=7)
    bam = BAM_handler(path)
    n = len(bam.get_reads("ctg", 0, 20000, False, 0, 0))
    bam.close()
    assert n > 100
    raw = bytearray(open(path, "rb").read())
    offs, p = [], 0
    while p < len(raw):
        offs.append(p)
        p += (raw[p + 16] | (raw[p + 17] << 8)) + 1
    victim = offs[len(offs) // 2]
    size = (raw[victim + 16] | (raw[victim + 17] << 8)) + 1
    # try bytes of the member's DEFLATE stream until one inflates structurally: only the CRC can tell then
    caught = False
    for at in range(victim + 30, victim + size - 9):
        trial = bytearray(raw)
        trial[at] ^= 0x01
        body = bytes(trial[victim + 18:victim + size - 8])
        try:
            out = zlib.decompress(body, -15)
        except zlib.error:
            continue
        isize = struct.unpack_from("<I", trial, victim + size - 4)[0]
        if len(out) != isize:
            continue
        open(path, "wb").write(bytes(trial))
        bam = BAM_handler(path)
        with pytest.raises(BamError):
            bam.get_reads("ctg", 0, 20000, False, 0, 0)
        bam.close()
        caught = True
        break
    assert caught


def test_records_without_their_bases(tmp_path):
    """SEQ '*' (l_seq 0 beside a CIGAR: legal, samtools writes it for some supplementary / secondary records) holds nothing to
    pile up: the record is left out.  A CIGAR that walks over more bases than the record holds is a corrupt record and fails the
    query -- neither may read past the record."""
    ok = dict(name="ok", pos=100, cigar=[(0, 50)], seq="ACGTA" * 10, qual=[30] * 50)
    bare = dict(name="bare", pos=120, cigar=[(0, 80)], seq="", qual=[])
    path = str(tmp_path / "bare.bam")
    bu.write_bam(path, [("ctg", 1000)], {0: [ok, bare, dict(ok, name="ok2", pos=130)]})
    bam = BAM_handler(path)
    got = bam.get_reads("ctg", 0, 1000, False, 0, 0)
    assert [r.query_name for r in got] == ["ok", "ok2"]
    compare(got, bu.restated_get_reads([ok, dict(ok, name="ok2", pos=130)], 0, 1000, False, 0))
    bam.close()
    short = dict(name="short", pos=120, cigar=[(0, 80)], seq="ACGT" * 5, qual=[30] * 20)
    path = str(tmp_path / "short.bam")
    bu.write_bam(path, [("ctg", 1000)], {0: [ok, short]})
    bam = BAM_handler(path)
    with pytest.raises(BamError, match="CIGAR longer than the read"):
        bam.get_reads("ctg", 0, 1000, False, 0, 0)
    compare(bam.get_reads("ctg", 0, 115, False, 0, 0), bu.restated_get_reads([ok], 0, 115, False, 0))   # the handle is still usable
    bam.close()


def _mixed_reads(rng, ref_len, n_reads, read_len):
    ref = pu.random_reference(rng, ref_len)
    reads = pu.simulate_reads(rng, ref, 0, n_reads=n_reads, read_len=read_len, clip_rate=0.4, mapq_zero_rate=0.1)
    reads = [r for r in reads if not any(op in (3, 6) for op, _ in r["cigar"])]
    for i, r in enumerate(reads):
        r["name"] = "read%d" % i
        r["flag"] = (16 if r["reverse"] else 0) | int(rng.choice([0, 0x800, 0x100, 0x400, 0x200, 0x4], p=[.86, .04, .04, .02, .02, .02]))
    return reads


def test_closed_form_of_the_clipping_walk():
    """What the device computes per (read, region) from prefix sums (bam_utils.closed_form_clip) equals the reference's
    sequential walk (bam_utils.restated_get_reads) -- on random reads with every operation kind, regions of a few bases to a few
    kb, and chunk sizes that put the first kept base on a chunk edge."""
    rng = np.random.default_rng(41)
    reads = _mixed_reads(rng, 30000, 500, (100, 5000))
    extra = [dict(pos=95, cigar=[(4, 3), (1, 2), (0, 5), (2, 4), (1, 3), (0, 6), (3, 9), (0, 2), (4, 2)], seq="A" * 23, qual=[9] * 23),
             dict(pos=100, cigar=[(5, 4), (2, 3), (0, 4)], seq="ACGT", qual=[9] * 4),
             dict(pos=90, cigar=[(0, 10), (2, 30)], seq="A" * 10, qual=[9] * 10),        # only a deletion reaches the region
             dict(pos=80, cigar=[(0, 20), (1, 4), (0, 3)], seq="A" * 27, qual=[9] * 27)]   # insert sitting right at start
    cases = 0
    for rec in reads + extra:
        end = rec["pos"] + max(1, bu.ref_length(rec["cigar"]))
        spots = [rec["pos"] - 5, rec["pos"], rec["pos"] + 7, (rec["pos"] + end) // 2, end - 2, end]
        for start in spots + (list(range(88, 125)) if rec in extra else []):
            for width in (0, 1, 13, 700):
                start, stop = max(0, int(start)), max(0, int(start)) + width
                want = bu.restated_get_reads([dict(rec, flag=0, mapq=60)], start, stop, True, 0)
                for chunk in (64, 3):
                    got = bu.closed_form_clip(rec, start, stop, chunk)
                    if not want:
                        # (the iterator's own test may have dropped the read before the walk: the closed form then keeps nothing
                        # or is never asked)
                        assert got is None or not (rec["pos"] < stop and end > start)
                        continue
                    w = want[0]
                    assert got is not None and (got["pos"], got["cigar"]) == (w["pos"], w["cigar"])
                    assert rec["seq"][got["first_idx"]:got["first_idx"] + got["written"]].upper() == w["seq"]
                    cases += 1
    assert cases > 5000


def test_pack_regions_lists_what_get_reads_returns(tmp_path):
    """pa_bam_pack_regions + the closed-form clip == get_reads for every region of a batch: filters, the iterator's region test,
    reads shared by neighbouring regions stored once, regions cut off when the arena is small (n_done < n)."""
    from pepper_amd.variant.bam import PACKED_READ
    rng = np.random.default_rng(43)
    reads = _mixed_reads(rng, 60000, 900, (300, 6000))
    other = _mixed_reads(rng, 9000, 40, (200, 900))
    path = str(tmp_path / "p.bam")
    bu.write_bam(path, [("chrA", 9000), ("chrB", 60000)], {0: other, 1: reads}, flush_every=37)
    bam = BAM_handler(path)
    edges = list(range(0, 60000, 7000)) + [60000]
    starts = np.array([max(0, a - 100) for a in edges[:-1]], np.int64)
    stops = np.array([b + 100 for b in edges[1:]], np.int64)

    def check(arena_bytes, include_supp, min_mapq):
        arena = np.zeros(arena_bytes, np.uint8)
        table = np.zeros(4000, PACKED_READ)
        pair_read = np.zeros(8000, np.int32)
        r0, calls = 0, 0
        while r0 < len(starts):
            n_done, region_pairs, (n_reads, n_pairs, used) = bam.pack_regions("chrB", starts[r0:], stops[r0:], include_supp, min_mapq,
                                                                               arena, table, pair_read)
            calls += 1
            assert 1 <= n_done <= len(starts) - r0 and used <= arena_bytes and region_pairs[n_done] == n_pairs
            assert len(set(int(table["data_off"][k]) for k in range(n_reads))) == n_reads
            for r in range(n_done):
                got = []
                for k in pair_read[region_pairs[r]:region_pairs[r + 1]].tolist():
                    rec = bu.unpack_packed_read(arena, table[k])
                    c = bu.closed_form_clip(rec, int(starts[r0 + r]), int(stops[r0 + r]))
                    if c is not None:
                        got.append((c["pos"], c["cigar"], rec["seq"][c["first_idx"]:c["first_idx"] + c["written"]],
                                    rec["qual"][c["first_idx"]:c["first_idx"] + c["written"]], bool(rec["flag"] & 16), rec["mapq"]))
                want = bam.get_reads("chrB", int(starts[r0 + r]), int(stops[r0 + r]), include_supp, min_mapq, 0)
                assert got == [(w.pos, [(c.cigar_op, c.cigar_len) for c in w.cigar_tuples], w.sequence, w.base_qualities,
                                w.flags.is_reverse, w.mapping_quality) for w in want]
            r0 += n_done
        return calls
    assert check(1 << 24, False, 1) == 1
    assert check(1 << 24, True, 0) == 1
    total = sum(4 * len(r["cigar"]) + 3 * len(r["seq"]) // 2 for r in reads)
    assert check(total // 3, False, 0) > 2           # a third of the contig's read data: the batch is cut into several calls
    with pytest.raises(BamError, match="do not fit"):
        bam.pack_regions("chrB", starts, stops, False, 0, np.zeros(total // 40, np.uint8), np.zeros(4000, PACKED_READ),
                         np.zeros(8000, np.int32))
    with pytest.raises(BamError):
        bam.pack_regions("chrB", starts[::-1], stops[::-1], False, 0, np.zeros(1 << 20, np.uint8), np.zeros(4000, PACKED_READ),
                         np.zeros(8000, np.int32))
    bam.close()


def _inflate_span(bam, chromosome, start, stop, lookahead, buf_bytes=1 << 26, blocks=4096, extra=1):
    """region_span + read_span + zlib on every member (the host stand-in of the device inflate) -> (data, first, final, complete)."""
    import zlib
    begin, first, end, final = bam.region_span(chromosome, start, stop, lookahead)
    buf = np.zeros(buf_bytes, np.uint8)
    tables = (np.zeros(blocks, np.int64), np.zeros(blocks, np.int32), np.zeros(blocks, np.int64), np.zeros(blocks, np.int32))
    n, comp_bytes, out_bytes, complete, at_eof = bam.read_span(begin, end, buf, tables, extra)
    final = final or at_eof
    data = np.zeros(max(out_bytes, 1), np.uint8)
    for k in range(n):
        o, l, at, m = int(tables[0][k]), int(tables[1][k]), int(tables[2][k]), int(tables[3][k])
        got = zlib.decompress(buf[o:o + l].tobytes(), -15)
        assert len(got) == m
        data[at:at + m] = np.frombuffer(got, np.uint8)
    return data, out_bytes, first, final, complete, n


def test_pack_inflated_equals_pack_regions(tmp_path):
    """The span form (pa_bam_region_span / read_span / pack_inflated: records left in place in an inflated stretch of the file)
    lists exactly pack_regions' reads and pairs; a span that stops short closes fewer regions; a contig's tail runs to the
    next contig's first member; unaligned slices."""
    from pepper_amd.variant.bam import PACKED_READ
    rng = np.random.default_rng(47)
    reads = _mixed_reads(rng, 200000, 1500, (300, 9000))
    other = _mixed_reads(rng, 9000, 40, (200, 900))
    last = _mixed_reads(rng, 30000, 100, (200, 3000))
    path = str(tmp_path / "s.bam")
    bu.write_bam(path, [("chrA", 9000), ("chrB", 200000), ("chrC", 30000), ("chrD", 1000)], {0: other, 1: reads, 2: last}, flush_every=23)
    bam = BAM_handler(path)

    def tables():
        return np.zeros(8000, PACKED_READ), np.zeros(16000, np.int32)

    def compare(chromosome, starts, stops, lookahead, include_supp=False, min_mapq=0):
        starts, stops = np.asarray(starts, np.int64), np.asarray(stops, np.int64)
        data, data_bytes, first, final, complete, _ = _inflate_span(bam, chromosome, int(starts[0]), int(stops[-1]), lookahead)
        assert complete
        t1, p1 = tables()
        n1, rp1, (nr1, np1, used1) = bam.pack_inflated(data, data_bytes, first, final, chromosome, starts, stops, include_supp, min_mapq,
                                                        t1, p1)
        arena = np.zeros(1 << 25, np.uint8)
        t2, p2 = tables()
        n2, rp2, (nr2, np2, used2) = bam.pack_regions(chromosome, starts[:n1], stops[:n1], include_supp, min_mapq, arena, t2, p2)
        assert n2 == n1 and nr1 == nr2 and np1 == np2
        assert rp1[:n1 + 1].tolist() == rp2[:n1 + 1].tolist() and p1[:np1].tolist() == p2[:np2].tolist()
        odd = 0
        for k in range(nr1):
            a, b = bu.unpack_packed_read(data, t1[k]), bu.unpack_packed_read(arena, t2[k])
            assert a == b
            odd += int(t1[k]["data_off"]) & 3 != 0
        return n1, nr1, odd

    edges = list(range(20000, 90000, 5000))
    n, n_reads, odd = compare("chrB", [a - 50 for a in edges[:-1]], [b + 50 for b in edges[1:]], 4)
    assert n == len(edges) - 1 and n_reads > 50 and odd > 10            # (records do not sit on word boundaries)
    assert compare("chrB", [100000], [101000], 4, True, 5)[0] == 1
    assert compare("chrB", [190000, 195000], [195000, 200000], 4)[0] == 2  # the contig's tail: the span runs to chrC's first record
    assert compare("chrC", [0, 10000, 20000], [10000, 20000, 30000], 4)[0] == 3   # the last contig with records: to the end of the file
    assert compare("chrA", [0], [9000], 0)[0] == 1
    # a contig without records: an empty span, every region done with nothing in it
    begin, first, end, final = bam.region_span("chrD", 0, 1000)
    assert (begin, end, final) == (0, 0, True)
    t, p = tables()
    n_done, rp, counts = bam.pack_inflated(np.zeros(1, np.uint8), 0, 0, True, "chrD", [0], [1000], False, 0, t, p)
    assert n_done == 1 and counts == (0, 0, 0)
    # a span that stops short of the last region's reads: the regions closed by then are done, the rest is the caller's
    starts, stops = np.arange(20000, 120000, 10000), np.arange(30000, 130000, 10000)
    data, data_bytes, first, final, complete, n_members = _inflate_span(bam, "chrB", 20000, 60000, 0)
    assert not final
    t, p = tables()
    n_done, rp, (nr, npairs, _) = bam.pack_inflated(data, data_bytes, first, False, "chrB", starts, stops, False, 0, t, p)
    assert 4 <= n_done < len(starts)
    arena = np.zeros(1 << 25, np.uint8)
    t2, p2 = tables()
    n2, rp2, (nr2, np2, _) = bam.pack_regions("chrB", starts[:n_done], stops[:n_done], False, 0, arena, t2, p2)
    assert (n2, nr2, np2) == (n_done, nr, npairs) and p[:npairs].tolist() == p2[:np2].tolist()
    # ... and one that ends before the first region closes is refused
    with pytest.raises(BamError, match="longer span") as e:
        bam.pack_inflated(data, 70000, first, False, "chrB", [20000], [150000], False, 0, t, p)
    assert e.value.code == -9
    # buffers too small for the span: whole members only, complete = False
    assert not _inflate_span(bam, "chrB", 20000, 90000, 4, buf_bytes=200000)[4]
    assert not _inflate_span(bam, "chrB", 20000, 90000, 4, blocks=3)[4]
    # the full tables: stops at a region boundary like pack_regions
    data, data_bytes, first, final, complete, _ = _inflate_span(bam, "chrB", 20000, 90000, 4)
    small = np.zeros(60, PACKED_READ)
    n_done, rp, (nr, npairs, _) = bam.pack_inflated(data, data_bytes, first, final, "chrB", [a - 50 for a in edges[:-1]],
                                                    [b + 50 for b in edges[1:]], False, 0, small, np.zeros(16000, np.int32))
    assert 1 <= n_done < len(edges) - 1 and nr <= 60
    bam.close()


def test_pack_inflated_refuses_a_cigar_in_the_cg_tag(tmp_path):
    rng = np.random.default_rng(48)
    seq = "".join("ACGT"[k] for k in rng.integers(0, 4, 2000))
    rec = dict(name="long", flag=0, pos=10, mapq=60, cigar=[(0, 500), (2, 3), (0, 1500)], seq=seq, qual=[30] * len(seq), long_cigar=True)
    path = str(tmp_path / "cg.bam")
    bu.write_bam(path, [("chrA", 400000)], {0: [rec]})
    bam = BAM_handler(path)
    from pepper_amd.variant.bam import PACKED_READ
    data, data_bytes, first, final, complete, _ = _inflate_span(bam, "chrA", 0, 1000, 1)
    with pytest.raises(BamError) as e:
        bam.pack_inflated(data, data_bytes, first, final, "chrA", [0], [1000], False, 0, np.zeros(10, PACKED_READ), np.zeros(10, np.int32))
    assert e.value.code == -8
    bam.close()


def _host_headers(data, data_bytes, first):
    """What the device's record walk returns (pa_record_header per record of the span), derived here record by record."""
    from pepper_amd.variant.bam import RECORD_HEADER
    out, at = [], int(first)
    while at + 4 <= data_bytes:
        bs = int.from_bytes(data[at:at + 4].tobytes(), "little")
        if at + 4 + bs > data_bytes:
            break
        R = data[at + 4:at + 4 + bs].tobytes()
        ref_id, pos = int.from_bytes(R[0:4], "little", signed=True), int.from_bytes(R[4:8], "little", signed=True)
        l_name, mapq = R[8], R[9]
        n_cig, flag, l_seq = int.from_bytes(R[12:14], "little"), int.from_bytes(R[14:16], "little"), int.from_bytes(R[16:20], "little")
        words = np.frombuffer(R[32 + l_name:32 + l_name + 4 * n_cig], "<u4")
        ref_len = int(sum(int(w) >> 4 for w in words if (int(w) & 15) in (0, 2, 3, 7, 8)))
        state = 1 if n_cig >= 1 and (int(words[0]) & 15) == 4 and (int(words[0]) >> 4) == l_seq else 0
        out.append((at + 4 + 32 + l_name, ref_id, pos, l_seq, n_cig, flag | mapq << 16, ref_len, state, bs))
        at += 4 + bs
    return np.array(out, RECORD_HEADER), at


def test_pack_headers_equals_pack_inflated_and_entries_are_record_starts(tmp_path):
    """pa_bam_pack_headers over the headers of a span's records (what the device's walk reads out) == pa_bam_pack_inflated over
    the span's bytes; pa_bam_span_entries lists record starts (the first record, then the linear index's windows)."""
    from pepper_amd.variant.bam import PACKED_READ
    rng = np.random.default_rng(49)
    reads = _mixed_reads(rng, 150000, 1400, (300, 9000))
    tail = _mixed_reads(rng, 20000, 60, (200, 2000))
    path = str(tmp_path / "h.bam")
    bu.write_bam(path, [("chrA", 150000), ("chrB", 20000)], {0: reads, 1: tail}, flush_every=31)
    bam = BAM_handler(path)
    for starts, stops, lookahead in (([20000, 30000, 40000], [30100, 40100, 50100], 4), ([100000, 120000], [120000, 150000], 2),
                                     ([20000, 60000, 100000], [60000, 100000, 140000], 0)):
        import zlib
        begin, first, end, final = bam.region_span("chrA", starts[0], stops[-1], lookahead)
        buf = np.zeros(1 << 26, np.uint8)
        tables = (np.zeros(4096, np.int64), np.zeros(4096, np.int32), np.zeros(4096, np.int64), np.zeros(4096, np.int32))
        n, comp_bytes, out_bytes, complete, at_eof = bam.read_span(begin, end, buf, tables, 1)
        assert complete
        final = final or at_eof
        data = np.zeros(out_bytes + 8, np.uint8)
        for k in range(n):
            o, l, at, m = int(tables[0][k]), int(tables[1][k]), int(tables[2][k]), int(tables[3][k])
            data[at:at + m] = np.frombuffer(zlib.decompress(buf[o:o + l].tobytes(), -15), np.uint8)
        headers, walked_to = _host_headers(data, out_bytes, first)
        assert len(headers) > 100
        entries = np.zeros(1024, np.int64)
        n_entries = bam.span_entries("chrA", first, tables[2], n, entries)
        starts_of_records = set()
        at = int(first)
        for h in headers:
            starts_of_records.add(at)
            at += 4 + int(h["block_size"])
        assert n_entries >= 2 and int(entries[0]) == first and all(int(e) in starts_of_records for e in entries[:n_entries])
        assert np.all(np.diff(entries[:n_entries]) > 0)
        t1, p1 = np.zeros(8000, PACKED_READ), np.zeros(16000, np.int32)
        t2, p2 = np.zeros(8000, PACKED_READ), np.zeros(16000, np.int32)
        try:
            a = bam.pack_inflated(data, out_bytes, first, final, "chrA", starts, stops, False, 1, t1, p1)
        except BamError as err:
            with pytest.raises(BamError) as again:
                bam.pack_headers(headers, len(headers), final, "chrA", starts, stops, False, 1, t2, p2)
            assert again.value.code == err.code
            continue
        b = bam.pack_headers(headers, len(headers), final, "chrA", starts, stops, False, 1, t2, p2)
        assert a[0] == b[0] and a[1].tolist() == b[1].tolist() and a[2] == b[2]
        assert t1[:a[2][0]].tobytes() == t2[:b[2][0]].tobytes() and p1[:a[2][1]].tolist() == p2[:b[2][1]].tolist()
    # a record marked as keeping its CIGAR in the CG tag sends the batch to the host packer; a corrupt one fails
    bad = headers.copy()
    kept = int(np.flatnonzero((bad["pos"] < 140000) & (bad["pos"] > 100000) & ((bad["flags"] & 0xf04) == 0) & ((bad["flags"] >> 16) > 0))[0])
    bad["state"][kept] = 1
    with pytest.raises(BamError) as e:
        bam.pack_headers(bad, len(bad), final, "chrA", starts, stops, False, 1, t2, p2)
    assert e.value.code == -8
    bad["state"][kept] = 2
    with pytest.raises(BamError, match="corrupt"):
        bam.pack_headers(bad, len(bad), final, "chrA", starts, stops, False, 1, t2, p2)
    bam.close()


def test_host_inflate_of_member_tables():
    """pa_bgzf_inflate_host (the CPU baseline of the inflate bench): the members of a table through libdeflate / zlib on several
    threads equal zlib's output; a member that does not inflate to its ISIZE fails the call."""
    import struct
    import zlib
    from pepper_amd.bgzf import BgzfError, block_table, inflate_host
    rng = np.random.default_rng(50)

    def member(data, level):
        c = zlib.compressobj(level, zlib.DEFLATED, -15)
        body = c.compress(data) + c.flush()
        return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(body) + 25) + body +
                struct.pack("<II", zlib.crc32(data), len(data)))
    datas = [bytes(rng.integers(33, 74, int(rng.integers(1, 60000)), dtype=np.uint8)) for _ in range(40)] + [b"", b"abc" * 5000]
    buf = b"".join(member(d, k % 10) for k, d in enumerate(datas))
    table = block_table(buf)
    for threads in (1, 3, 64):
        assert inflate_host(buf, table, threads).tobytes() == b"".join(datas)
    wrong = [a.copy() for a in table]
    wrong[3][5] += 1
    wrong[2][6:] += 1
    with pytest.raises(BgzfError):
        inflate_host(buf, wrong, 2)
