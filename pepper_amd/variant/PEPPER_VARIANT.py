"""Python shim with the names of the reference's pybind11 module for the summary encoder.

Mirrors the part of `from pepper_variant.build import PEPPER_VARIANT` that image generation uses
(/root/reference/pepper_variant/modules/cpp/pybind_api.h:55-62 RegionalSummaryGenerator, :73-101
CandidateImageSummary, :187-221 CigarOp / type_read_flags / type_read; call site
/root/reference/pepper_variant/modules/python/AlignmentSummarizer.py:220-238).  Reads may be any
objects with the type_read attributes (pos, flags.is_reverse, sequence, cigar_tuples[.cigar_op,
.cigar_len], mapping_quality, base_qualities); they are flattened once and encoded by
libpepper_amd.so (include/pepper_amd_encoder.h).  `generate_summary_arrays` returns the bulk
struct-of-arrays form (images already int8-packed on the device) for callers that do not need
per-candidate Python objects.
"""
import ctypes
import os

import numpy as np

from pepper_amd import _lib


class CigarOp(object):
    def __init__(self, cigar_op=-1, cigar_len=0):
        self.cigar_op = cigar_op
        self.cigar_len = cigar_len


class type_read_flags(object):
    def __init__(self):
        for name in ("is_paired", "is_proper_pair", "is_unmapped", "is_mate_unmapped", "is_reverse",
                     "is_mate_is_reverse", "is_read1", "is_read2", "is_secondary", "is_qc_failed",
                     "is_duplicate", "is_supplementary"):
            setattr(self, name, False)


class type_read(object):
    def __init__(self):
        self.pos = 0
        self.pos_end = 0
        self.query_name = ""
        self.read_id = 0
        self.flags = type_read_flags()
        self.hp_tag = 0
        self.sequence = ""
        self.cigar_tuples = []
        self.mapping_quality = 0
        self.base_qualities = []
        self.bad_indicies = []

    def set_read_id(self, read_id):
        self.read_id = read_id

    def __lt__(self, other):
        return (self.pos, self.pos_end) < (other.pos, other.pos_end)


class CandidateImageSummary(object):
    def __init__(self, contig="", position=0, depth=0, candidates=None, candidate_frequency=None,
                 image_matrix=None, base_label=0, type_label=0):
        self.contig = contig
        self.position = position
        self.depth = depth
        self.candidates = candidates if candidates is not None else []
        self.candidate_frequency = candidate_frequency if candidate_frequency is not None else []
        self.image_matrix = image_matrix if image_matrix is not None else []
        self.base_label = base_label
        self.type_label = type_label

    def __getstate__(self):
        return (self.contig, self.position, self.depth, self.candidates, self.candidate_frequency,
                self.image_matrix, self.base_label, self.type_label)

    def __setstate__(self, t):
        if len(t) != 8:
            raise RuntimeError("Invalid state!")
        (self.contig, self.position, self.depth, self.candidates, self.candidate_frequency,
         self.image_matrix, self.base_label, self.type_label) = t


class _Pileup(ctypes.Structure):
    _fields_ = [("region_start", ctypes.c_int64), ("region_end", ctypes.c_int64),
                ("reference", ctypes.c_char_p), ("reference_len", ctypes.c_int64), ("n_reads", ctypes.c_int32),
                ("read_pos", ctypes.c_void_p), ("read_reverse", ctypes.c_void_p), ("read_mapq", ctypes.c_void_p),
                ("seq_offset", ctypes.c_void_p), ("seq", ctypes.c_void_p), ("qual", ctypes.c_void_p),
                ("cigar_offset", ctypes.c_void_p), ("cigar_op", ctypes.c_void_p), ("cigar_len", ctypes.c_void_p)]


class _Params(ctypes.Structure):
    _fields_ = [("min_snp_baseq", ctypes.c_double), ("min_indel_baseq", ctypes.c_double),
                ("snp_freq_threshold", ctypes.c_double), ("insert_freq_threshold", ctypes.c_double),
                ("delete_freq_threshold", ctypes.c_double), ("min_coverage_threshold", ctypes.c_double),
                ("snp_candidate_freq_threshold", ctypes.c_double),
                ("indel_candidate_freq_threshold", ctypes.c_double),
                ("candidate_support_threshold", ctypes.c_double), ("skip_indels", ctypes.c_int32),
                ("candidate_region_start", ctypes.c_int64), ("candidate_region_end", ctypes.c_int64),
                ("candidate_window_size", ctypes.c_int32), ("feature_size", ctypes.c_int32)]


_encoders = {}


def _encoder(device):
    """One native encoder (stream + workspace) per device per THREAD, created on first use: a handle holds the
    results of its last call, so image-generation worker threads must not share one."""
    import threading
    lib = _lib.load()
    key = (device, threading.get_ident())
    if key not in _encoders:
        h = ctypes.c_void_p()
        _lib.check(lib.pa_encoder_create(device, None, ctypes.byref(h)))
        _encoders[key] = h
    return lib, _encoders[key]


def flatten_reads(reads):
    """type_read-like objects -> the flat arrays of pa_pileup."""
    n = len(reads)
    read_pos = np.fromiter((r.pos for r in reads), np.int64, n)
    read_reverse = np.fromiter((1 if r.flags.is_reverse else 0 for r in reads), np.uint8, n)
    read_mapq = np.fromiter((r.mapping_quality for r in reads), np.int32, n)
    seq_offset = np.zeros(n + 1, np.int64)
    cigar_offset = np.zeros(n + 1, np.int64)
    np.cumsum([len(r.sequence) for r in reads], out=seq_offset[1:])
    np.cumsum([len(r.cigar_tuples) for r in reads], out=cigar_offset[1:])
    seq = np.frombuffer(("".join(r.sequence for r in reads)).encode("latin-1") + b"\0", np.uint8)
    qual = np.zeros(int(seq_offset[-1]) + 1, np.uint8)
    for i, r in enumerate(reads):
        qual[seq_offset[i]:seq_offset[i + 1]] = np.clip(np.asarray(r.base_qualities, np.int64), 0, 255)
    cigar_op = np.fromiter((c.cigar_op for r in reads for c in r.cigar_tuples), np.int32, int(cigar_offset[-1]))
    cigar_len = np.fromiter((c.cigar_len for r in reads for c in r.cigar_tuples), np.int32, int(cigar_offset[-1]))
    return dict(read_pos=read_pos, read_reverse=read_reverse, read_mapq=read_mapq, seq_offset=seq_offset,
                seq=seq, qual=qual, cigar_offset=cigar_offset, cigar_op=np.append(cigar_op, 0).astype(np.int32),
                cigar_len=np.append(cigar_len, 0).astype(np.int32), n_reads=n)


class RegionalSummaryGenerator(object):
    def __init__(self, contig, region_start, region_end, reference_sequence, device=0):
        self.contig = contig
        self.ref_start = int(region_start)
        self.ref_end = int(region_end)
        self.reference_sequence = reference_sequence
        self.device = device
        # GENERATE_INDELS == false in the reference (region_summary.h:50): no insert columns
        self.total_observered_insert_bases = 0

    def generate_max_insert_summary(self, reads):
        """Axes of the region.  With GENERATE_INDELS false every max_observed_insert entry is 0, so
        positions[i] = ref_start + i and index[i] = 0 (region_summary.cpp:19-96); nothing to compute."""
        return None

    def generate_summary_arrays(self, reads, min_snp_baseq, min_indel_baseq, snp_freq_threshold,
                                insert_freq_threshold, delete_freq_threshold, min_coverage_threshold,
                                snp_candidate_freq_threshold, indel_candidate_freq_threshold,
                                candidate_support_threshold, skip_indels, candidate_region_start,
                                candidate_region_end, candidate_window_size, feature_size, train_mode=False,
                                want_int32=False):
        return generate_summary_arrays_batch([self], [reads], min_snp_baseq, min_indel_baseq, snp_freq_threshold,
                                             insert_freq_threshold, delete_freq_threshold, min_coverage_threshold,
                                             snp_candidate_freq_threshold, indel_candidate_freq_threshold,
                                             candidate_support_threshold, skip_indels,
                                             [(candidate_region_start, candidate_region_end)], candidate_window_size,
                                             feature_size, train_mode, want_int32)[0]

    def generate_summary(self, reads, *args):
        """-> list[CandidateImageSummary], as the pybind method (region_summary.h:191-206)."""
        out = self.generate_summary_arrays(reads, *args, want_int32=True)
        res = []
        for i in range(len(out["candidates"])):
            res.append(CandidateImageSummary(self.contig, int(out["positions"][i]), int(out["depths"][i]),
                                             [out["candidates"][i]], [int(out["candidate_frequency"][i])],
                                             out["images_int32"][i].tolist(), 0, 0))
        return res


def _pileup_struct(gen, flat, keep):
    ref = gen.reference_sequence.encode("latin-1") if isinstance(gen.reference_sequence, str) else bytes(gen.reference_sequence)
    keep.append(ref)
    return _Pileup(gen.ref_start, gen.ref_end, ref, len(ref), flat["n_reads"],
                   flat["read_pos"].ctypes.data, flat["read_reverse"].ctypes.data, flat["read_mapq"].ctypes.data,
                   flat["seq_offset"].ctypes.data, flat["seq"].ctypes.data, flat["qual"].ctypes.data,
                   flat["cigar_offset"].ctypes.data, flat["cigar_op"].ctypes.data, flat["cigar_len"].ctypes.data)


class StagedBatch(object):
    """A batch of regions uploaded once (pa_encoder_stage_batch) and encoded any number of times
    (pa_encoder_run_staged): what bench.py times with the inputs resident in HBM."""

    def __init__(self, generators, reads_list, params, candidate_regions, candidate_window_size=32, feature_size=26):
        device = generators[0].device if generators else 0
        self.lib, self.enc = _encoder(device)
        self.n_regions = len(generators)
        self.window, self.features = candidate_window_size + 1, feature_size
        self._keep = []
        self.flats = [r if isinstance(r, dict) else flatten_reads(r) for r in reads_list]
        self.piles = (_Pileup * max(1, self.n_regions))(*[_pileup_struct(g, f, self._keep) for g, f in zip(generators, self.flats)])
        (min_snp_baseq, min_indel_baseq, snp_freq_threshold, insert_freq_threshold, delete_freq_threshold,
         min_coverage_threshold, snp_candidate_freq_threshold, indel_candidate_freq_threshold,
         candidate_support_threshold, skip_indels) = params
        self.params = (_Params * max(1, self.n_regions))(*[
            _Params(min_snp_baseq, min_indel_baseq, snp_freq_threshold, insert_freq_threshold, delete_freq_threshold,
                    min_coverage_threshold, snp_candidate_freq_threshold, indel_candidate_freq_threshold,
                    candidate_support_threshold, 1 if skip_indels else 0, int(lo), int(hi), int(candidate_window_size),
                    int(feature_size)) for lo, hi in candidate_regions])
        _lib.check(self.lib.pa_encoder_stage_batch(self.enc, self.n_regions, ctypes.cast(self.piles, ctypes.c_void_p),
                                                   ctypes.cast(self.params, ctypes.c_void_p)))
        self.counts = np.zeros(max(1, self.n_regions), np.int64)

    def run(self):
        """-> candidates per region"""
        _lib.check(self.lib.pa_encoder_run_staged(self.enc, self.counts.ctypes.data))
        return self.counts[:self.n_regions]

    def timing(self):
        ms = np.zeros(12, np.float64)
        _lib.check(self.lib.pa_encoder_last_timing(self.enc, ms.ctypes.data, 12))
        return dict(records_ms=ms[0], tile_count_ms=ms[1], compact_votes_ms=ms[2], gather_windows_ms=ms[3],
                    host_enumeration_ms=ms[4], run_ms=ms[5], host_bucket_ms=ms[6], host_bucket_and_threads_ms=ms[7],
                    upload_ms=ms[8], unpack_clip_ms=ms[9])

    def stats(self):
        v = np.zeros(6, np.int64)
        _lib.check(self.lib.pa_encoder_batch_stats(self.enc, v.ctypes.data, 6))
        return dict(bases=int(v[0]), rows=int(v[1]), reads=int(v[2]), cigar_ops=int(v[3]), tiles=int(v[4]), regions=int(v[5]))

    def results(self, want_int32=False):
        """-> one dict per region (the arrays of generate_summary_arrays)"""
        n = int(self.counts[:self.n_regions].sum())
        W, F = self.window, self.features
        positions = np.zeros(n, np.int64)
        depths = np.zeros(n, np.int32)
        freqs = np.zeros(n, np.int32)
        img8 = np.zeros((n, W, F), np.int8)
        img32 = np.zeros((n, W, F), np.int32) if want_int32 else None
        needed = ctypes.c_int64()
        lib, enc = self.lib, self.enc
        _lib.check(lib.pa_encoder_get_results(enc, None, None, None, None, None, None, 0, ctypes.byref(needed)))
        names = ctypes.create_string_buffer(max(1, needed.value))
        _lib.check(lib.pa_encoder_get_results(enc, positions.ctypes.data, depths.ctypes.data, freqs.ctypes.data,
                                              img32.ctypes.data if want_int32 else None, img8.ctypes.data,
                                              ctypes.cast(names, ctypes.c_void_p), needed.value, ctypes.byref(needed)))
        raw = names.raw[:needed.value]
        cands = [s.decode("latin-1") for s in raw.split(b"\0")[:n]]
        # the same strings as the library left them (NUL-terminated, back to back) with their offsets: what the image writer
        # takes without going through n Python strings again (DataStore.write_summary_packed)
        ends = np.flatnonzero(np.frombuffer(raw, np.uint8) == 0)[:n].astype(np.int64) + 1
        starts = np.concatenate([[0], ends]) if n else np.zeros(1, np.int64)
        out, at = [], 0
        for k in self.counts[:self.n_regions]:
            k = int(k)
            lo, hi = int(starts[at]), int(starts[at + k])
            out.append(dict(positions=positions[at:at + k], depths=depths[at:at + k], candidate_frequency=freqs[at:at + k],
                            images=img8[at:at + k], images_int32=img32[at:at + k] if want_int32 else None,
                            candidates=cands[at:at + k], candidates_blob=raw[lo:hi], candidates_offsets=starts[at:at + k + 1] - lo))
            at += k
        return out


class _PackedRegion(ctypes.Structure):
    _fields_ = [("region_start", ctypes.c_int64), ("region_end", ctypes.c_int64), ("reference", ctypes.c_char_p),
                ("reference_len", ctypes.c_int64)]


class PackedEncoder(object):
    """The packed form of a batch (pa_encoder_stage_packed): the reads of a run of regions as pa_bam_pack_regions leaves them
    in this object's page-locked arena -- CIGAR words, 4-bit bases, qualities, once per read -- clipped to each region and
    decoded by the device.  One object per worker thread: it owns its encoder handle (stream, workspace, arena)."""

    _idle = []                      # encoders returned by release(): a later job's workers take them instead of pinning new arenas
    _idle_lock = __import__("threading").Lock()

    @classmethod
    def acquire(cls, device=0, arena_bytes=192 << 20, host_threads=1, torch_stream=False):
        """An encoder from the process-wide pool (or a new one): pinning a 256 MB arena and the first device allocations cost
        ~0.15 s per worker, which a long-running process pays once.  Under a memlock / cgroup limit the page-locked arena may
        not be had at that size: the request is halved down to 16 MB before the error is passed on (a smaller arena means
        fewer intervals per call, nothing else); the callers fall back to the host-clipped form when even that fails."""
        with cls._idle_lock:
            for k, enc in enumerate(cls._idle):
                if (enc.device == device and enc.arena is not None and enc.arena.nbytes <= arena_bytes and enc.arena.nbytes >= min(arena_bytes, 16 << 20)
                        and (enc.stream is not None) == bool(torch_stream)):
                    return cls._idle.pop(k)
        size = arena_bytes
        while True:
            try:
                return cls(device, size, host_threads=host_threads, torch_stream=torch_stream)
            except _lib.PepperAmdError:
                if size <= 16 << 20:
                    raise
                size >>= 1

    def release(self):
        with self._idle_lock:
            self._idle.append(self)

    def __init__(self, device=0, arena_bytes=192 << 20, max_reads=1 << 18, max_pairs=1 << 19, host_threads=0, torch_stream=False):
        from pepper_amd.variant.bam import PACKED_READ
        self.lib = _lib.load()
        self.device = device
        self.enc = ctypes.c_void_p()
        # torch_stream: the encoder works on a stream torch made (self.stream) instead of one of its own, so that a caller can queue
        # torch operations -- a copy of the results the encoder left on the device -- right behind the encoder's kernels, in the same
        # hardware queue (polish/fused.py: a copy on any other stream waits behind whatever shares that stream's queue)
        self.stream = None
        if torch_stream:
            import torch
            self.stream = torch.cuda.Stream(device=device)
        _lib.check(self.lib.pa_encoder_create(device, ctypes.c_void_p(self.stream.cuda_stream) if self.stream is not None else None,
                                              ctypes.byref(self.enc)))
        _lib.check(self.lib.pa_encoder_set_host_threads(self.enc, host_threads))
        ptr = self.lib.pa_encoder_host_arena(self.enc, arena_bytes)
        if not ptr:
            raise _lib.PepperAmdError("page-locked arena of %d bytes could not be allocated" % arena_bytes)
        self.arena = np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_uint8)), shape=(arena_bytes,))
        self.reads = np.zeros(max_reads, PACKED_READ)
        self.pair_read = np.zeros(max_pairs, np.int32)
        self.span = None                # page-locked block for a file span's BGZF members (pack_device), allocated on first use
        self.tables = None
        self.headers = None             # record headers of a span as the device's walk returns them
        self.entries = None
        self.inflate_ms = 0.0           # device time of the inflate kernels of this object's pack_device calls
        self.inflated_bytes = 0

    def close(self):
        if self.enc:
            self.arena = None
            self.lib.pa_encoder_destroy(self.enc)
            self.enc = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def pack(self, bam_handler, contig, starts, stops, include_supplementary, min_mapq):
        """-> (n_done, region_pairs, (n_reads, n_pairs, arena_bytes)): BAM_handler.pack_regions into this object's buffers."""
        return bam_handler.pack_regions(contig, starts, stops, include_supplementary, min_mapq, self.arena, self.reads, self.pair_read)

    def pack_device(self, bam_handler, contig, starts, stops, include_supplementary, min_mapq, lookahead_windows=4, laps=None):
        """The same tables with the BGZF members inflated ON THE DEVICE (pa_encoder_inflate_bgzf) into the encoder's arena and
        the records left in place there: the file span of the regions' reads (BAM index) is read as it is, uploaded, inflated
        one wavefront per member, and walked on the host in a downloaded copy (headers, filters, region test -- no inflate, no
        copy).  -> (n_done, region_pairs, counts) for encode(..., resident=True), or None when the batch has to take pack():
        no index, a span larger than the arena even for one region, a record with its CIGAR in the CG tag, reads longer than
        the span's lookahead."""
        import time
        from pepper_amd.variant.bam import BamError
        if not bam_handler.has_index():
            return None
        if self.span is None:
            cap = self.arena.nbytes + (1 << 20)
            ptr = self.lib.pa_encoder_host_span(self.enc, cap)
            if not ptr:
                return None                              # (no second page-locked block under this memlock limit: the host packer's form)
            self.span = np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_uint8)), shape=(cap,))
            nb = max(4096, self.arena.nbytes // 4096)
            self.tables = (np.zeros(nb, np.int64), np.zeros(nb, np.int32), np.zeros(nb, np.int64), np.zeros(nb, np.int32))
        t0 = time.perf_counter()
        n = len(starts)
        while n >= 1:
            begin, first, end, final = bam_handler.region_span(contig, int(starts[0]), int(stops[n - 1]), lookahead_windows)
            if end <= begin:                         # no record of the contig: every region is done, with nothing in it
                return n, np.zeros(n + 1, np.int32), (0, 0, 0)
            n_blocks, comp_bytes, out_bytes, complete, at_eof = bam_handler.read_span(begin, end, self.span, self.tables, 1)
            final = final or at_eof
            if complete and out_bytes + 256 <= self.arena.nbytes:
                break
            n //= 2
        else:
            return None
        if laps is not None:
            laps["bam_span_read"] = laps.get("bam_span_read", 0.0) + time.perf_counter() - t0
            t0 = time.perf_counter()
        comp_off, comp_len, out_off, out_len = self.tables
        device_walk = os.environ.get("PEPPER_AMD_DEVICE_WALK", "1") != "0"

        def inflate(host_copy):
            _lib.check(self.lib.pa_encoder_inflate_bgzf(self.enc, self.span.ctypes.data, comp_bytes, n_blocks, comp_off.ctypes.data,
                                                        comp_len.ctypes.data, out_off.ctypes.data, out_len.ctypes.data, out_bytes,
                                                        self.arena.ctypes.data if host_copy else None))
            ms = np.zeros(12, np.float64)
            _lib.check(self.lib.pa_encoder_last_timing(self.enc, ms.ctypes.data, 12))
            self.inflate_ms += float(ms[10])
            self.inflated_bytes += int(out_bytes)
        inflate(not device_walk)
        if laps is not None:
            laps["bam_inflate_device"] = laps.get("bam_inflate_device", 0.0) + time.perf_counter() - t0
            t0 = time.perf_counter()
        headers = None
        if device_walk:
            # the record headers read out on the device (40 bytes per record come back instead of the span)
            if self.headers is None:
                from pepper_amd.variant.bam import RECORD_HEADER
                self.headers = np.zeros(max(1 << 16, self.arena.nbytes // 512), RECORD_HEADER)
                self.entries = np.zeros(8192, np.int64)
            n_entries = bam_handler.span_entries(contig, first, out_off, n_blocks, self.entries)
            n_headers, flags = ctypes.c_int64(), np.zeros(2, np.int32)
            # slots per entry: 2 048 records of one 16 kb window, fewer when a span has very many windows (low coverage): the
            # device keeps two 40-byte tables of entries x slots; a window that overflows its slots takes the host walk
            slots = max(64, min(2048, (32 << 20) // (40 * max(1, n_entries))))
            try:
                _lib.check(self.lib.pa_encoder_walk_records(self.enc, out_bytes, self.entries.ctypes.data, n_entries, slots,
                                                            self.headers.ctypes.data, len(self.headers), ctypes.byref(n_headers),
                                                            flags.ctypes.data))
            except _lib.PepperAmdError as err:
                if getattr(err, "code", 0) != _lib.PA_ERR_INVALID:
                    raise
                return None                              # (entries outside the span: a stale index -- the host packer reads the file itself)
            if flags[0] == 0:
                headers = n_headers.value
            else:
                inflate(True)                        # (a window with more records than a lane's slots, ...: the span to the host after all)
            if laps is not None:
                laps["bam_walk_device"] = laps.get("bam_walk_device", 0.0) + time.perf_counter() - t0
                t0 = time.perf_counter()
        try:
            if headers is not None:
                n_done, region_pairs, counts = bam_handler.pack_headers(self.headers, headers, final, contig, starts[:n], stops[:n],
                                                                        include_supplementary, min_mapq, self.reads, self.pair_read)
            else:
                n_done, region_pairs, counts = bam_handler.pack_inflated(self.arena, out_bytes, first, final, contig, starts[:n], stops[:n],
                                                                         include_supplementary, min_mapq, self.reads, self.pair_read)
        except BamError as err:
            if getattr(err, "code", 0) in (-7, -8, -9):
                return None
            raise
        finally:
            if laps is not None:
                laps["bam_walk"] = laps.get("bam_walk", 0.0) + time.perf_counter() - t0
        return n_done, region_pairs, (counts[0], counts[1], int(out_bytes))

    def encode(self, regions, references, region_pairs, counts, params, candidate_regions, candidate_window_size=32, feature_size=26,
               want_int32=False, resident=False):
        """regions: [(ref_start, ref_end)] of the packed run (the fetch ranges), references: their sequences (bytes / str),
        region_pairs / counts: what pack() returned, params: the ten thresholds of generate_summary in order,
        candidate_regions: [(start, end)].  -> (one dict of arrays per region as generate_summary_arrays, reads per region)."""
        n = len(regions)
        refs = [r.encode("latin-1") if isinstance(r, str) else bytes(r) for r in references]
        regs = (_PackedRegion * max(1, n))(*[_PackedRegion(int(a), int(b), ref, len(ref)) for (a, b), ref in zip(regions, refs)])
        (min_snp_baseq, min_indel_baseq, snp_freq_threshold, insert_freq_threshold, delete_freq_threshold,
         min_coverage_threshold, snp_candidate_freq_threshold, indel_candidate_freq_threshold,
         candidate_support_threshold, skip_indels) = params
        pars = (_Params * max(1, n))(*[
            _Params(min_snp_baseq, min_indel_baseq, snp_freq_threshold, insert_freq_threshold, delete_freq_threshold,
                    min_coverage_threshold, snp_candidate_freq_threshold, indel_candidate_freq_threshold,
                    candidate_support_threshold, 1 if skip_indels else 0, int(lo), int(hi), int(candidate_window_size),
                    int(feature_size)) for lo, hi in candidate_regions])
        region_pairs = np.ascontiguousarray(region_pairs[:n + 1], np.int32)
        n_reads, _n_pairs, arena_bytes = counts
        # resident: the arena is the inflated span pack_device left on the device
        _lib.check(self.lib.pa_encoder_stage_packed(self.enc, n, ctypes.cast(regs, ctypes.c_void_p), ctypes.cast(pars, ctypes.c_void_p),
                                                    None if (resident and n_reads > 0) else self.arena.ctypes.data,
                                                    int(arena_bytes), self.reads.ctypes.data, int(n_reads),
                                                    self.pair_read.ctypes.data, region_pairs.ctypes.data))
        batch = StagedBatch.__new__(StagedBatch)
        batch.lib, batch.enc, batch.n_regions = self.lib, self.enc, n
        batch.window, batch.features = candidate_window_size + 1, feature_size
        batch.counts = np.zeros(max(1, n), np.int64)
        batch._keep = (regs, pars, refs)
        batch.run()
        live = np.zeros(max(1, n), np.int32)
        _lib.check(self.lib.pa_encoder_region_reads(self.enc, live.ctypes.data, n))
        self.last = batch
        return batch.results(want_int32), live[:n]


def generate_summary_arrays_batch(generators, reads_list, min_snp_baseq, min_indel_baseq, snp_freq_threshold,
                                  insert_freq_threshold, delete_freq_threshold, min_coverage_threshold,
                                  snp_candidate_freq_threshold, indel_candidate_freq_threshold,
                                  candidate_support_threshold, skip_indels, candidate_regions, candidate_window_size,
                                  feature_size, train_mode=False, want_int32=False):
    """Many regions through one set of launches (pa_encoder_generate_summary_batch): generators[i] with reads_list[i]
    (type_read-like objects or the flat arrays of flatten_reads) and candidate_regions[i] = (start, end); the other
    arguments are those of RegionalSummaryGenerator.generate_summary.  -> one dict of arrays per region."""
    if train_mode:
        raise NotImplementedError("train_mode label generation (truth VCF haplotypes) is outside the inference path")
    batch = StagedBatch(generators, reads_list,
                        (min_snp_baseq, min_indel_baseq, snp_freq_threshold, insert_freq_threshold, delete_freq_threshold,
                         min_coverage_threshold, snp_candidate_freq_threshold, indel_candidate_freq_threshold,
                         candidate_support_threshold, skip_indels), candidate_regions, candidate_window_size, feature_size)
    batch.run()
    return batch.results(want_int32)
