"""Python shim with the names of the reference's pybind11 module for the polish summary encoder.

Mirrors `from pepper.build import PEPPER`'s SummaryGenerator
(/root/reference/pepper/modules/headers/pybind_api.h:18-25; call site
/root/reference/pepper/modules/python/AlignmentSummarizer.py:340-347): after
`generate_summary(reads, start, end)` the object exposes `.image` (uint8 [rows,10]),
`.genomic_pos` ([(position, insert_index)]), `.labels`, `.bad_label_positions`.
Encoded by libpepper_amd.so (include/pepper_amd_encoder.h, pa_polish_encoder_*).
"""
import ctypes

import numpy as np

from pepper_amd import _lib
from pepper_amd.variant.PEPPER_VARIANT import (CigarOp, _Pileup, _encoder, flatten_reads,  # noqa: F401
                                               type_read, type_read_flags)


class SummaryGenerator(object):
    def __init__(self, reference_sequence, chromosome_name, ref_start, ref_end, device=0):
        self.reference_sequence = reference_sequence
        self.chromosome_name = chromosome_name
        self.ref_start = int(ref_start)
        self.ref_end = int(ref_end)
        self.device = device
        self.image = np.zeros((0, 10), np.uint8)
        self.positions_array = np.zeros((0, 2), np.int64)
        self._genomic_pos = []
        self.labels = []
        self.bad_label_positions = []

    @property
    def genomic_pos(self):
        if self._genomic_pos is None:
            self._genomic_pos = [tuple(x) for x in self.positions_array.tolist()]
        return self._genomic_pos

    @genomic_pos.setter
    def genomic_pos(self, value):
        self._genomic_pos = value
        self.positions_array = np.asarray(value, dtype=np.int64).reshape(-1, 2)

    def generate_summary(self, reads, start_pos, end_pos):
        flat = reads if isinstance(reads, dict) else flatten_reads(reads)
        lib, enc = _encoder(self.device)
        ref = self.reference_sequence.encode("latin-1") if isinstance(self.reference_sequence, str) else bytes(self.reference_sequence)
        p = _Pileup(self.ref_start, self.ref_end, ref, len(ref), flat["n_reads"],
                    flat["read_pos"].ctypes.data, flat["read_reverse"].ctypes.data, flat["read_mapq"].ctypes.data,
                    flat["seq_offset"].ctypes.data, flat["seq"].ctypes.data, flat["qual"].ctypes.data,
                    flat["cigar_offset"].ctypes.data, flat["cigar_op"].ctypes.data, flat["cigar_len"].ctypes.data)
        n = ctypes.c_int64()
        _lib.check(lib.pa_polish_encoder_generate_summary(enc, ctypes.cast(ctypes.pointer(p), ctypes.c_void_p),
                                                          int(start_pos), int(end_pos), ctypes.byref(n)))
        rows = n.value
        image = np.zeros((rows, 10), np.uint8)
        pos = np.zeros((rows, 2), np.int64)
        _lib.check(lib.pa_polish_encoder_get_results(enc, image.ctypes.data, pos.ctypes.data))
        self.image = image
        self.positions_array = pos          # [rows, 2] int64; `genomic_pos` (list of tuples) is built on first access
        self._genomic_pos = None


def generate_summaries(generators, reads_list, spans):
    """Many regions through one set of launches (pa_polish_encoder_generate_summary_batch): generators[i] (SummaryGenerator
    objects) with reads_list[i] (type_read-like objects or flatten_reads arrays) and spans[i] = (start_pos, end_pos).  Fills
    every generator's .image / .positions_array as generate_summary does."""
    if not generators:
        return
    lib, enc = _encoder(generators[0].device)
    n = len(generators)
    flats = [r if isinstance(r, dict) else flatten_reads(r) for r in reads_list]
    keep = []
    piles = (_Pileup * n)()
    for i, (g, flat) in enumerate(zip(generators, flats)):
        ref = g.reference_sequence.encode("latin-1") if isinstance(g.reference_sequence, str) else bytes(g.reference_sequence)
        keep.append(ref)
        piles[i] = _Pileup(g.ref_start, g.ref_end, ref, len(ref), flat["n_reads"], flat["read_pos"].ctypes.data,
                           flat["read_reverse"].ctypes.data, flat["read_mapq"].ctypes.data, flat["seq_offset"].ctypes.data,
                           flat["seq"].ctypes.data, flat["qual"].ctypes.data, flat["cigar_offset"].ctypes.data,
                           flat["cigar_op"].ctypes.data, flat["cigar_len"].ctypes.data)
    starts = np.array([int(s) for s, _ in spans], np.int64)
    ends = np.array([int(e) for _, e in spans], np.int64)
    rows = np.zeros(n, np.int64)
    _lib.check(lib.pa_polish_encoder_generate_summary_batch(enc, n, ctypes.cast(piles, ctypes.c_void_p), starts.ctypes.data,
                                                            ends.ctypes.data, rows.ctypes.data))
    total = int(rows.sum())
    image = np.zeros((total, 10), np.uint8)
    pos = np.zeros((total, 2), np.int64)
    _lib.check(lib.pa_polish_encoder_get_results(enc, image.ctypes.data, pos.ctypes.data))
    at = 0
    for g, k in zip(generators, rows):
        k = int(k)
        g.image = image[at:at + k]
        g.positions_array = pos[at:at + k]
        g._genomic_pos = None
        at += k


class StagedSummaries(object):
    """A batch of polish regions uploaded once (pa_polish_encoder_stage_batch) and encoded any number of times
    (pa_polish_encoder_run_staged): what bench.py --model polish-encoder times with the pileups resident in HBM."""

    def __init__(self, generators, flats, spans):
        self.lib, self.enc = _encoder(generators[0].device if generators else 0)
        self.n = len(generators)
        self._keep = [g.reference_sequence.encode("latin-1") if isinstance(g.reference_sequence, str) else bytes(g.reference_sequence)
                      for g in generators]
        self.flats = flats
        self.piles = (_Pileup * max(1, self.n))(*[
            _Pileup(g.ref_start, g.ref_end, ref, len(ref), flat["n_reads"], flat["read_pos"].ctypes.data,
                    flat["read_reverse"].ctypes.data, flat["read_mapq"].ctypes.data, flat["seq_offset"].ctypes.data,
                    flat["seq"].ctypes.data, flat["qual"].ctypes.data, flat["cigar_offset"].ctypes.data,
                    flat["cigar_op"].ctypes.data, flat["cigar_len"].ctypes.data)
            for g, ref, flat in zip(generators, self._keep, flats)])
        self.starts = np.array([int(s) for s, _ in spans], np.int64)
        self.ends = np.array([int(e) for _, e in spans], np.int64)
        self.rows = np.zeros(max(1, self.n), np.int64)
        _lib.check(self.lib.pa_polish_encoder_stage_batch(self.enc, self.n, ctypes.cast(self.piles, ctypes.c_void_p),
                                                          self.starts.ctypes.data, self.ends.ctypes.data))

    def run(self):
        _lib.check(self.lib.pa_polish_encoder_run_staged(self.enc, self.rows.ctypes.data))
        return self.rows[:self.n]

    def timing(self):
        ms = np.zeros(4, np.float64)
        _lib.check(self.lib.pa_polish_encoder_last_timing(self.enc, ms.ctypes.data, 4))
        return dict(records_ms=ms[0], tile_ms=ms[1], insert_rows_ms=ms[2])

    def stats(self):
        v = np.zeros(6, np.int64)
        _lib.check(self.lib.pa_polish_encoder_batch_stats(self.enc, v.ctypes.data, 6))
        return dict(bases=int(v[0]), rows=int(v[1]), reads=int(v[2]), cigar_ops=int(v[3]), tiles=int(v[4]), regions=int(v[5]))

    def results(self):
        total = int(self.rows[:self.n].sum())
        image = np.zeros((total, 10), np.uint8)
        pos = np.zeros((total, 2), np.int64)
        _lib.check(self.lib.pa_polish_encoder_get_results(self.enc, image.ctypes.data, pos.ctypes.data))
        return image, pos


class PolishChain(object):
    """BAM records -> image chunks of a run of regions on the device (include/pepper_amd_encoder.h, pa_polish_chain_run): what
    AlignmentSummarizer.create_summary does per region -- get_reads, ReadAligner, SummaryGenerator, chunk_images
    (/root/reference/pepper/modules/python/AlignmentSummarizer.py:296-358) -- for N regions per call, with the reads clipped,
    re-aligned, summarised and cut into chunks without leaving the GPU.  `packed` is a pepper_amd.variant.PEPPER_VARIANT.
    PackedEncoder: its pack_device() / pack() fill the tables this object hands on (one object per worker thread)."""

    def __init__(self, packed):
        self.packed = packed
        self.lib = packed.lib
        self.n_chunks = 0
        self.chunk_size = 0

    def run(self, regions, windows, region_pairs, counts, realign=True, resident=False, chunk_size=1000, chunk_overlap=50):
        """regions: [(start, end)] of the packed run -- or of a stretch of it: region_pairs[0] may be > 0, the pairs of the
        regions given are then pair_read[region_pairs[0] .. region_pairs[n]) (one pack_device() span serves several chain calls);
        windows[r]: the draft from start to end + 20 (bytes; shorter at the contig's end); region_pairs / counts: what
        pack_device() / pack() returned.  -> (rows per region, reads per region, chunks per region)."""
        from pepper_amd.variant.PEPPER_VARIANT import _PackedRegion
        pe = self.packed
        n = len(regions)
        refs = [w if isinstance(w, bytes) else bytes(w) for w in windows]
        regs = (_PackedRegion * max(1, n))(*[_PackedRegion(int(a), int(b), ref, len(ref)) for (a, b), ref in zip(regions, refs)])
        first_pair = int(region_pairs[0])
        region_pairs = np.ascontiguousarray(np.asarray(region_pairs[:n + 1], np.int64) - first_pair, np.int32)
        n_reads, _n_pairs, arena_bytes = counts
        rows, live, chunks = np.zeros(max(1, n), np.int64), np.zeros(max(1, n), np.int32), np.zeros(max(1, n), np.int32)
        total = ctypes.c_int64()
        _lib.check(self.lib.pa_polish_chain_run(
            pe.enc, n, ctypes.cast(regs, ctypes.c_void_p), None if (resident and n_reads > 0) else pe.arena.ctypes.data,
            int(arena_bytes), pe.reads.ctypes.data, int(n_reads), pe.pair_read.ctypes.data + 4 * first_pair, region_pairs.ctypes.data,
            1 if realign else 0, int(chunk_size), int(chunk_overlap), rows.ctypes.data, live.ctypes.data, chunks.ctypes.data,
            ctypes.byref(total)))
        self.n_chunks, self.chunk_size = total.value, int(chunk_size)
        return rows[:n], live[:n], chunks[:n]

    def chunk_pointers(self):
        """Addresses of the last run's chunks in the handle's page-locked memory: images uint8 [n_chunks, chunk_size, 10],
        position / index int64 [n_chunks, chunk_size]; valid until the next run."""
        img, pos, idx = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        _lib.check(self.lib.pa_polish_chain_chunks(self.packed.enc, ctypes.byref(img), ctypes.byref(pos), ctypes.byref(idx)))
        return img.value, pos.value, idx.value

    def device_chunks(self):
        """Device address of the last run's images ([n_chunks, chunk_size, 10] uint8), valid until the next run."""
        img = ctypes.c_void_p()
        _lib.check(self.lib.pa_polish_chain_device_chunks(self.packed.enc, ctypes.byref(img)))
        return img.value

    def chunk_arrays(self):
        """The same as numpy views (copy them to keep them past the next run)."""
        img, pos, idx = self.chunk_pointers()
        n, c = self.n_chunks, self.chunk_size
        if n == 0:
            return np.zeros((0, c, 10), np.uint8), np.zeros((0, c), np.int64), np.zeros((0, c), np.int64)
        return (np.ctypeslib.as_array(ctypes.cast(img, ctypes.POINTER(ctypes.c_uint8)), shape=(n, c, 10)),
                np.ctypeslib.as_array(ctypes.cast(pos, ctypes.POINTER(ctypes.c_int64)), shape=(n, c)),
                np.ctypeslib.as_array(ctypes.cast(idx, ctypes.POINTER(ctypes.c_int64)), shape=(n, c)))

    def timing(self):
        ms, counts = np.zeros(8, np.float64), np.zeros(5, np.int64)
        _lib.check(self.lib.pa_polish_chain_last_timing(self.packed.enc, ms.ctypes.data, 8, counts.ctypes.data, 5))
        return dict(unpack_ms=ms[0], realign_ms=ms[1], encode_ms=ms[2], chunk_ms=ms[3], score_kernel_ms=ms[5], band_kernel_ms=ms[6],
                    pairs=int(counts[0]), realigned=int(counts[1]), cigar_ops=int(counts[2]), rows=int(counts[3]),
                    proven_overflows=int(counts[4]))


_realigners = {}


def _realigner(device):
    """One native re-aligner (stream + workspace) per device per thread."""
    import threading
    lib = _lib.load()
    key = (device, threading.get_ident())
    if key not in _realigners:
        h = ctypes.c_void_p()
        _lib.check(lib.pa_realigner_create(device, None, ctypes.byref(h)))
        _realigners[key] = h
    return lib, _realigners[key]


def align_windows(windows, read_window, read_pos, seq_offset, seq, device=0, collapse_eqx=True):
    """Reads of several regions in one call (include/pepper_amd_realign.h, pa_realigner_align_windows).
    windows: [(ref_start, ref_seq)]; read_window[k] = index of read k's window.  Returns the flat result arrays:
    status (1 aligned, 0 kept, -1 dropped), score, pos, pos_end, query span and CIGAR arrays."""
    lib, h = _realigner(device)
    read_pos = np.ascontiguousarray(read_pos, dtype=np.int64)
    seq_offset = np.ascontiguousarray(seq_offset, dtype=np.int64)
    seq = np.ascontiguousarray(seq, dtype=np.uint8)
    read_window = np.ascontiguousarray(read_window, dtype=np.int32)
    n = len(read_pos)
    texts = [(w[1].encode("latin-1") if isinstance(w[1], str) else bytes(w[1])) for w in windows]
    window_offset = np.zeros(len(windows) + 1, np.int64)
    np.cumsum([len(t) for t in texts], out=window_offset[1:])
    window_start = np.array([int(w[0]) for w in windows], np.int64)
    status, score = np.zeros(n, np.int32), np.zeros(n, np.int32)
    pos, pos_end = np.zeros(n, np.int64), np.zeros(n, np.int64)
    qbeg, qend = np.zeros(n, np.int32), np.zeros(n, np.int32)
    total = ctypes.c_int64()
    seq_arg = seq if len(seq) else np.zeros(1, np.uint8)
    _lib.check(lib.pa_realigner_align_windows(h, len(windows), b"".join(texts), window_offset.ctypes.data,
                                              window_start.ctypes.data, n, read_window.ctypes.data, read_pos.ctypes.data,
                                              seq_offset.ctypes.data, seq_arg.ctypes.data, status.ctypes.data,
                                              score.ctypes.data, pos.ctypes.data, pos_end.ctypes.data, qbeg.ctypes.data,
                                              qend.ctypes.data, ctypes.byref(total)))
    cigar_offset = np.zeros(n + 1, np.int64)
    cigar_op, cigar_len = np.zeros(max(1, total.value), np.int32), np.zeros(max(1, total.value), np.int32)
    _lib.check(lib.pa_realigner_copy_cigars(h, int(bool(collapse_eqx)), cigar_offset.ctypes.data, cigar_op.ctypes.data,
                                            cigar_len.ctypes.data))
    return dict(status=status, score=score, pos=pos, pos_end=pos_end, query_begin=qbeg, query_end=qend,
                cigar_offset=cigar_offset, cigar_op=cigar_op[:total.value], cigar_len=cigar_len[:total.value])


def apply_alignment(reads, out, first=0):
    """ReadSet `reads` = reads first .. first + len(reads) of a result of align_windows -> the ReadSet the reference's
    ReadAligner returns for them: dropped reads removed, aligned reads with the new position / end / CIGAR."""
    from pepper_amd.variant.bam import ReadSet
    n = len(reads)
    if n == 0:
        return reads
    status = out["status"][first:first + n]
    keep = np.nonzero(status >= 0)[0]
    base = reads if len(keep) == n else reads.take(keep)
    aligned = status[keep] == 1
    pos = np.where(aligned, out["pos"][first:first + n][keep], base.pos)
    pos_end = np.where(aligned, out["pos_end"][first:first + n][keep], base.pos_end)
    old_n = base.cigar_offset[1:] - base.cigar_offset[:-1]
    new_off = out["cigar_offset"][first:first + n + 1]
    counts = np.where(aligned, (new_off[1:] - new_off[:-1])[keep], old_n)
    offsets = np.zeros(len(keep) + 1, np.int64)
    np.cumsum(counts, out=offsets[1:])
    ops, lens = np.empty(int(offsets[-1]), np.int32), np.empty(int(offsets[-1]), np.int32)
    for i, k in enumerate(keep.tolist()):
        a, b = int(offsets[i]), int(offsets[i + 1])
        if aligned[i]:
            s0 = int(new_off[k])
            ops[a:b], lens[a:b] = out["cigar_op"][s0:s0 + b - a], out["cigar_len"][s0:s0 + b - a]
        else:
            s0 = int(base.cigar_offset[i])
            ops[a:b], lens[a:b] = base.cigar_op[s0:s0 + b - a], base.cigar_len[s0:s0 + b - a]
    return ReadSet(pos, pos_end, base.reverse, base.mapq, base.flags, base.hp, base.seq_offset, base.seq, base.qual,
                   offsets, ops, lens, base.names)


class ReadAligner(object):
    """`PEPPER.ReadAligner(ref_start, ref_end, ref_seq).align_reads_to_reference(reads)`

    /root/reference/pepper/modules/src/local_reassembly/simple_aligner.cpp:60-106: every read is aligned (SSW, match 4,
    mismatch 6, gap open 8, gap extend 2) against the reference suffix that starts at its mapped position; with a score
    > 1 it gets the new position / end position / CIGAR ('=' and 'X' runs both become MATCH operations, not merged),
    otherwise it is passed through; reads that start before the region are dropped.  The alignments run on the GPU
    (include/pepper_amd_realign.h).  `reads` is a ReadSet (pepper_amd.variant.bam) -> ReadSet, or a list of
    type_read-like objects -> list of type_read.
    """

    def __init__(self, ref_start, ref_end, ref_seq, device=0):
        self.region_start = int(ref_start)
        self.region_end = int(ref_end)
        self.reference_sequence = ref_seq
        self.device = device

    def align_arrays(self, read_pos, seq_offset, seq, collapse_eqx=True):
        """Flat form: per-read status (1 aligned, 0 kept, -1 dropped), score, pos, pos_end, query span and CIGAR arrays."""
        return align_windows([(self.region_start, self.reference_sequence)], np.zeros(len(read_pos), np.int32), read_pos,
                             seq_offset, seq, device=self.device, collapse_eqx=collapse_eqx)

    def align_reads_to_reference(self, reads):
        from pepper_amd.variant.bam import ReadSet
        if isinstance(reads, ReadSet):
            if len(reads) == 0:
                return reads
            return apply_alignment(reads, self.align_arrays(reads.pos, reads.seq_offset, reads.seq))
        reads = list(reads)
        if not reads:
            return []
        seqs = [r.sequence.encode("latin-1") for r in reads]
        seq_offset = np.zeros(len(reads) + 1, np.int64)
        np.cumsum([len(s) for s in seqs], out=seq_offset[1:])
        out = self.align_arrays([r.pos for r in reads], seq_offset, np.frombuffer(b"".join(seqs), np.uint8))
        result = []
        for k, read in enumerate(reads):
            st = int(out["status"][k])
            if st < 0:
                continue
            if st == 0:
                result.append(read)
                continue
            new = type_read()
            for name in ("query_name", "read_id", "flags", "hp_tag", "sequence", "mapping_quality", "base_qualities", "bad_indicies"):
                if hasattr(read, name):
                    setattr(new, name, getattr(read, name))
            a, b = int(out["cigar_offset"][k]), int(out["cigar_offset"][k + 1])
            new.cigar_tuples = [CigarOp(int(o), int(n)) for o, n in zip(out["cigar_op"][a:b], out["cigar_len"][a:b])]
            new.pos, new.pos_end = int(out["pos"][k]), int(out["pos_end"][k])
            result.append(new)
        return result
