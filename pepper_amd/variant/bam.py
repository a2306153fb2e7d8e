"""BAM access with the reference's `PEPPER_VARIANT.BAM_handler` surface, over libpepper_amd_io.so
(include/pepper_amd_io.h, pepper_amd/csrc/bamio.cpp -- zlib only, no htslib).

replaces: /root/reference/pepper_variant/modules/cpp/bam_handler.cpp (pybind: pybind_api.h BAM_handler)
    BAM_handler(path), get_chromosome_sequence_names(), get_sample_names(),
    get_reads(contig, start, stop, include_supplementary, min_mapq, min_baseq)
`get_reads` returns a ReadSet: the clipped reads as the flat arrays the GPU encoder takes
(`as_pileup()`), indexable / iterable as `type_read`-like views for code written against the
reference objects (`.pos`, `.pos_end`, `.query_name`, `.sequence`, `.base_qualities`,
`.cigar_tuples[i].cigar_op/.cigar_len`, `.mapping_quality`, `.flags.is_reverse`, `.hp_tag`).
"""
import ctypes
from types import SimpleNamespace

import numpy as np

from pepper_amd import h5

c_void_p, c_int32, c_int64, c_char_p = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_char_p
P64 = ctypes.POINTER(c_int64)

SYMBOLS = [
    ("pa_bam_last_error", c_char_p, []),
    ("pa_bam_open", ctypes.c_int, [c_char_p, ctypes.POINTER(c_void_p)]),
    ("pa_bam_close", None, [c_void_p]),
    ("pa_bam_has_index", ctypes.c_int, [c_void_p]),
    ("pa_bam_n_targets", ctypes.c_int, [c_void_p]),
    ("pa_bam_target", ctypes.c_int, [c_void_p, c_int32, c_void_p, c_int32, P64]),
    ("pa_bam_header_text", ctypes.c_int, [c_void_p, c_void_p, c_int64, P64]),
    ("pa_bam_get_reads", ctypes.c_int, [c_void_p, c_char_p, c_int64, c_int64, c_int32, c_int32, c_int32, P64, P64, P64, P64]),
    ("pa_bam_copy_reads", ctypes.c_int, [c_void_p] + [c_void_p] * 13),
    ("pa_bam_pack_regions", ctypes.c_int, [c_void_p, c_char_p, c_int32, c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_int64,
                                           c_void_p, c_int32, c_void_p, c_int32, c_void_p, ctypes.POINTER(c_int32), c_void_p]),
    ("pa_bgzf_inflate_host", ctypes.c_int, [c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32]),
    ("pa_bam_region_span", ctypes.c_int, [c_void_p, c_char_p, c_int64, c_int64, c_int32, P64, ctypes.POINTER(c_int32), P64,
                                          ctypes.POINTER(c_int32)]),
    ("pa_bam_read_span", ctypes.c_int, [c_void_p, c_int64, c_int64, c_int32, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_int32, ctypes.POINTER(c_int32), P64, P64, ctypes.POINTER(c_int32)]),
    ("pa_bam_span_entries", ctypes.c_int, [c_void_p, c_char_p, c_int64, c_void_p, c_int32, c_void_p, c_int32, ctypes.POINTER(c_int32)]),
    ("pa_bam_pack_headers", ctypes.c_int, [c_void_p, c_void_p, c_int64, c_int32, c_char_p, c_int32, c_void_p, c_void_p, c_int32, c_int32,
                                           c_void_p, c_int32, c_void_p, c_int32, c_void_p, ctypes.POINTER(c_int32), c_void_p]),
    ("pa_bam_pack_inflated", ctypes.c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int32, c_char_p, c_int32, c_void_p, c_void_p,
                                            c_int32, c_int32, c_void_p, c_int32, c_void_p, c_int32, c_void_p,
                                            ctypes.POINTER(c_int32), c_void_p]),
]
_bound = False


class BamError(RuntimeError):
    pass


def _lib():
    global _bound
    lib = h5.load()
    if not _bound:
        for name, restype, argtypes in SYMBOLS:
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = restype, argtypes
        _bound = True
    return lib


def _check(rc):
    if rc < 0:
        err = BamError(_lib().pa_bam_last_error().decode())
        err.code = rc
        raise err
    return rc


class ReadSet(object):
    """Clipped reads of one region, structure-of-arrays."""

    def __init__(self, pos, pos_end, reverse, mapq, flags, hp, seq_offset, seq, qual, cigar_offset, cigar_op, cigar_len, names):
        self.pos, self.pos_end, self.reverse, self.mapq, self.flags, self.hp = pos, pos_end, reverse, mapq, flags, hp
        self.seq_offset, self.seq, self.qual = seq_offset, seq, qual
        self.cigar_offset, self.cigar_op, self.cigar_len = cigar_offset, cigar_op, cigar_len
        self.names = names

    def __len__(self):
        return len(self.pos)

    def take(self, indices):
        """Subset / reorder (reservoir sampling in AlignmentSummarizer keeps slot order)."""
        idx = np.asarray(indices, dtype=np.int64)
        so, co = self.seq_offset, self.cigar_offset
        seq_len, cig_len = (so[1:] - so[:-1])[idx], (co[1:] - co[:-1])[idx]
        new_so = np.zeros(len(idx) + 1, np.int64)
        new_co = np.zeros(len(idx) + 1, np.int64)
        np.cumsum(seq_len, out=new_so[1:])
        np.cumsum(cig_len, out=new_co[1:])

        def gather(starts, lens, total):
            if total == 0:
                return np.zeros(0, np.int64)
            reps = np.repeat(starts - np.concatenate(([0], np.cumsum(lens)[:-1])), lens)
            return reps + np.arange(total)
        si = gather(so[:-1][idx], seq_len, int(new_so[-1]))
        ci = gather(co[:-1][idx], cig_len, int(new_co[-1]))
        return ReadSet(self.pos[idx], self.pos_end[idx], self.reverse[idx], self.mapq[idx], self.flags[idx], self.hp[idx],
                       new_so, self.seq[si], self.qual[si], new_co, self.cigar_op[ci], self.cigar_len[ci],
                       [self.names[i] for i in idx.tolist()])

    def as_pileup(self):
        """The dict PEPPER_VARIANT.RegionalSummaryGenerator.generate_summary_arrays takes (pa_pileup fields)."""
        pad8 = np.zeros(1, np.uint8)
        pad32 = np.zeros(1, np.int32)
        return dict(read_pos=self.pos, read_reverse=self.reverse, read_mapq=self.mapq, seq_offset=self.seq_offset,
                    seq=np.concatenate((self.seq, pad8)), qual=np.concatenate((self.qual, pad8)),
                    cigar_offset=self.cigar_offset, cigar_op=np.concatenate((self.cigar_op, pad32)),
                    cigar_len=np.concatenate((self.cigar_len, pad32)), n_reads=len(self))

    def __getitem__(self, i):
        if isinstance(i, slice):
            return self.take(range(*i.indices(len(self))))
        if i < 0:
            i += len(self)
        s0, s1 = int(self.seq_offset[i]), int(self.seq_offset[i + 1])
        c0, c1 = int(self.cigar_offset[i]), int(self.cigar_offset[i + 1])
        flag = int(self.flags[i])
        return SimpleNamespace(
            query_name=self.names[i], pos=int(self.pos[i]), pos_end=int(self.pos_end[i]),
            sequence=self.seq[s0:s1].tobytes().decode("latin-1"), base_qualities=self.qual[s0:s1].astype(np.int64).tolist(),
            mapping_quality=int(self.mapq[i]), hp_tag=int(self.hp[i]),
            cigar_tuples=[SimpleNamespace(cigar_op=int(o), cigar_len=int(n)) for o, n in zip(self.cigar_op[c0:c1], self.cigar_len[c0:c1])],
            flags=SimpleNamespace(is_paired=bool(flag & 0x1), is_proper_pair=bool(flag & 0x2), is_unmapped=bool(flag & 0x4),
                                  is_mate_unmapped=bool(flag & 0x8), is_reverse=bool(flag & 0x10),
                                  is_mate_is_reverse=bool(flag & 0x20), is_read1=bool(flag & 0x40), is_read2=bool(flag & 0x80),
                                  is_secondary=bool(flag & 0x100), is_qc_failed=bool(flag & 0x200),
                                  is_duplicate=bool(flag & 0x400), is_supplementary=bool(flag & 0x800)))

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]


PACKED_READ = np.dtype([("data_off", np.int64), ("pos", np.int32), ("n_cigar", np.int32), ("l_seq", np.int32),
                        ("flags", np.int32)])          # pa_packed_read (include/pepper_amd_io.h)
RECORD_HEADER = np.dtype([("data_off", np.int64), ("ref_id", np.int32), ("pos", np.int32), ("l_seq", np.int32), ("n_cigar", np.int32),
                          ("flags", np.int32), ("ref_len", np.int32), ("state", np.int32), ("block_size", np.int32)])      # pa_record_header


class BAM_handler(object):
    def __init__(self, path):
        self._h = c_void_p()
        self.path = path
        _check(_lib().pa_bam_open(str(path).encode(), ctypes.byref(self._h)))

    def close(self):
        if self._h:
            _lib().pa_bam_close(self._h)
            self._h = c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def has_index(self):
        return bool(_lib().pa_bam_has_index(self._h))

    def get_chromosome_sequence_names(self):
        lib = _lib()
        out = []
        for i in range(lib.pa_bam_n_targets(self._h)):
            n = _check(lib.pa_bam_target(self._h, i, None, 0, None))
            buf = ctypes.create_string_buffer(n + 1)
            _check(lib.pa_bam_target(self._h, i, buf, n + 1, None))
            out.append(buf.value.decode())
        return out

    def get_chromosome_sequence_names_with_length(self):
        lib = _lib()
        out = []
        for i, name in enumerate(self.get_chromosome_sequence_names()):
            length = c_int64()
            _check(lib.pa_bam_target(self._h, i, None, 0, ctypes.byref(length)))
            out.append(SimpleNamespace(sequence_name=name, sequence_length=length.value))
        return out

    def get_header_text(self):
        lib = _lib()
        needed = c_int64()
        _check(lib.pa_bam_header_text(self._h, None, 0, ctypes.byref(needed)))
        buf = ctypes.create_string_buffer(needed.value)
        _check(lib.pa_bam_header_text(self._h, buf, needed.value, ctypes.byref(needed)))
        return buf.value.decode()

    def get_sample_names(self):
        """SM values of the @RG lines (bam_handler.cpp:30-54)."""
        samples = set()
        for line in self.get_header_text().split("\n"):
            fields = line.split("\t")
            if fields and fields[0] == "@RG":
                for token in fields[1:]:
                    parts = token.split(":")
                    if parts[0] == "SM" and len(parts) > 1:
                        samples.add(parts[1])
        return samples

    def pack_regions(self, chromosome, starts, stops, include_supplementary, min_mapq, arena, reads, pair_read):
        """pa_bam_pack_regions: the reads of the regions [starts[r], stops[r]] (ascending, one contig) in the packed form of the
        GPU encoder, written into the caller's buffers -- arena: uint8 array (the encoder's page-locked arena), reads: array
        of PACKED_READ, pair_read: int32 array.  -> (n_done, region_pairs int32 [n + 1], (n_reads, n_pairs, arena_bytes)):
        the first n_done regions are complete; the caller goes on with the rest when n_done < len(starts)."""
        starts = np.ascontiguousarray(starts, np.int64)
        stops = np.ascontiguousarray(stops, np.int64)
        n = len(starts)
        region_pairs = np.zeros(n + 1, np.int32)
        counts = np.zeros(3, np.int64)
        n_done = c_int32()
        _check(_lib().pa_bam_pack_regions(self._h, str(chromosome).encode(), n, starts.ctypes.data, stops.ctypes.data,
                                          int(bool(include_supplementary)), int(min_mapq), arena.ctypes.data, arena.nbytes,
                                          reads.ctypes.data, len(reads), pair_read.ctypes.data, len(pair_read),
                                          region_pairs.ctypes.data, ctypes.byref(n_done), counts.ctypes.data))
        return n_done.value, region_pairs, (int(counts[0]), int(counts[1]), int(counts[2]))

    # ---- the packed form over spans inflated on the device (include/pepper_amd_io.h) ----
    def region_span(self, chromosome, start, stop, lookahead_windows=4):
        """-> (begin_coffset, begin_uoffset, end_coffset, to_contig_end): the file span of the records reaching [start, stop)."""
        b, u, e, f = c_int64(), c_int32(), c_int64(), c_int32()
        _check(_lib().pa_bam_region_span(self._h, str(chromosome).encode(), int(start), int(stop), int(lookahead_windows),
                                         ctypes.byref(b), ctypes.byref(u), ctypes.byref(e), ctypes.byref(f)))
        return b.value, u.value, e.value, bool(f.value)

    def read_span(self, begin, end_min, buf, tables, extra_members=1):
        """Members of the file span into `buf` (uint8 array) and their inflate tables into `tables` = (comp_off int64, comp_len
        int32, out_off int64, out_len int32) -> (n_blocks, comp_bytes, out_bytes, complete, at_eof)."""
        comp_off, comp_len, out_off, out_len = tables
        n, cb, ob, done = c_int32(), c_int64(), c_int64(), c_int32()
        _check(_lib().pa_bam_read_span(self._h, int(begin), int(end_min), int(extra_members), buf.ctypes.data, buf.nbytes,
                                       comp_off.ctypes.data, comp_len.ctypes.data, out_off.ctypes.data, out_len.ctypes.data,
                                       min(len(comp_off), len(comp_len), len(out_off), len(out_len)), ctypes.byref(n),
                                       ctypes.byref(cb), ctypes.byref(ob), ctypes.byref(done)))
        return n.value, cb.value, ob.value, bool(done.value & 1), bool(done.value & 2)

    def span_entries(self, chromosome, first_record, out_off, n_blocks, entries):
        """Record starts inside the last read_span's inflated bytes (first_record, then the linear index's entries) into the
        int64 array `entries` -> their number."""
        n = c_int32()
        _check(_lib().pa_bam_span_entries(self._h, str(chromosome).encode(), int(first_record), out_off.ctypes.data, int(n_blocks),
                                          entries.ctypes.data, len(entries), ctypes.byref(n)))
        return n.value

    def pack_headers(self, headers, n_headers, data_is_final, chromosome, starts, stops, include_supplementary, min_mapq, reads,
                     pair_read):
        """pa_bam_pack_headers: pack_inflated's tables from the record headers the device read out (array of RECORD_HEADER)."""
        starts = np.ascontiguousarray(starts, np.int64)
        stops = np.ascontiguousarray(stops, np.int64)
        n = len(starts)
        region_pairs = np.zeros(n + 1, np.int32)
        counts = np.zeros(3, np.int64)
        n_done = c_int32()
        _check(_lib().pa_bam_pack_headers(self._h, headers.ctypes.data, int(n_headers), int(bool(data_is_final)), str(chromosome).encode(),
                                          n, starts.ctypes.data, stops.ctypes.data, int(bool(include_supplementary)), int(min_mapq),
                                          reads.ctypes.data, len(reads), pair_read.ctypes.data, len(pair_read),
                                          region_pairs.ctypes.data, ctypes.byref(n_done), counts.ctypes.data))
        return n_done.value, region_pairs, (int(counts[0]), int(counts[1]), int(counts[2]))

    def pack_inflated(self, data, data_bytes, first_record, data_is_final, chromosome, starts, stops, include_supplementary,
                      min_mapq, reads, pair_read):
        """pa_bam_pack_inflated: pack_regions' tables over an inflated span (`data`: uint8 array), records left in place."""
        starts = np.ascontiguousarray(starts, np.int64)
        stops = np.ascontiguousarray(stops, np.int64)
        n = len(starts)
        region_pairs = np.zeros(n + 1, np.int32)
        counts = np.zeros(3, np.int64)
        n_done = c_int32()
        _check(_lib().pa_bam_pack_inflated(self._h, data.ctypes.data, int(data_bytes), int(first_record), int(bool(data_is_final)),
                                           str(chromosome).encode(), n, starts.ctypes.data, stops.ctypes.data,
                                           int(bool(include_supplementary)), int(min_mapq), reads.ctypes.data, len(reads),
                                           pair_read.ctypes.data, len(pair_read), region_pairs.ctypes.data, ctypes.byref(n_done),
                                           counts.ctypes.data))
        return n_done.value, region_pairs, (int(counts[0]), int(counts[1]), int(counts[2]))

    def get_reads(self, chromosome, start, stop, include_supplementary, min_mapq=0, min_baseq=0):
        lib = _lib()
        n, nb, nc, nn = c_int64(), c_int64(), c_int64(), c_int64()
        _check(lib.pa_bam_get_reads(self._h, str(chromosome).encode(), int(start), int(stop), int(bool(include_supplementary)),
                                    int(min_mapq), int(min_baseq), ctypes.byref(n), ctypes.byref(nb), ctypes.byref(nc),
                                    ctypes.byref(nn)))
        n, nb, nc, nn = n.value, nb.value, nc.value, nn.value
        pos, pos_end = np.empty(n, np.int64), np.empty(n, np.int64)
        reverse = np.empty(n, np.uint8)
        mapq, flags, hp = np.empty(n, np.int32), np.empty(n, np.int32), np.empty(n, np.int32)
        seq_offset, cigar_offset = np.empty(n + 1, np.int64), np.empty(n + 1, np.int64)
        seq, qual = np.empty(nb, np.uint8), np.empty(nb, np.uint8)
        cigar_op, cigar_len = np.empty(nc, np.int32), np.empty(nc, np.int32)
        names = ctypes.create_string_buffer(max(1, nn))
        _check(lib.pa_bam_copy_reads(self._h, pos.ctypes.data, pos_end.ctypes.data, reverse.ctypes.data, mapq.ctypes.data,
                                     flags.ctypes.data, hp.ctypes.data, seq_offset.ctypes.data, seq.ctypes.data,
                                     qual.ctypes.data, cigar_offset.ctypes.data, cigar_op.ctypes.data, cigar_len.ctypes.data,
                                     ctypes.cast(names, c_void_p)))
        name_list = [s.decode("latin-1") for s in names.raw[:nn].split(b"\0")[:n]]
        return ReadSet(pos, pos_end, reverse, mapq, flags, hp, seq_offset, seq, qual, cigar_offset, cigar_op, cigar_len,
                       name_list)
