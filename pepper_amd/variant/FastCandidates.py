"""predictions HDF5 -> the five candidate VCFs, column-wise.

The same rules as CandidateFinder.small_chunk_stitch / find_candidates and VcfWriter.write_vcf_records
(/root/reference/pepper_variant/modules/python/CandidateFinder.py:356-581, VcfWriter.py:48-218), which build one
Python tuple per selected allele, one dict entry per site, and format and write one record at a time (~35 us per
candidate).  Here a prediction batch is handled as numpy columns (genotype, "non alt" probability, reference base and
low-complexity flag of every row at once), the selected rows of all batches are ordered and grouped by site with one stable
lexsort, the records of sites that carry ONE allele record -- nearly all of them -- are formatted in a single pass over plain
Python lists -- or, for batches of the form this package's image generation writes (one contig, one allele per row), inside
the I/O library (pa_candidates_select_format, candidates.cpp: selection and record text of a 512-row batch in ~0.1 ms;
PEPPER_AMD_CANDIDATES_PYTHON=1 keeps the Python loops) --, and each file is written with one join, its virtual offsets and
its tabix index computed arithmetically.
Sites with several allele records go through the reference-shaped merge (VCFWriter.candidate_list_to_variant) unchanged.
The phasing ("margin") list of find_candidates is not built: the writer never reads it (FindCandidates.py:170-176).

tests/test_candidate_finder.py holds this path to the tuple path: same bytes in every .vcf.gz once decompressed, same
index content.  Text rendering stays UNPINNED against pysam / htslib (absent from this image), as for the tuple path.
"""
import ctypes
import math
import os

import numpy as np

from pepper_amd import h5
from pepper_amd.variant.CandidateFinder import (_BASES, _fasta, _in_repeat_many, _parse_list_field, _ReferenceWindow,
                                                _select_site)
from pepper_amd.variant.VcfWriter import _F32, VCFWriter

_FORMAT = "GT:AP:GQ:DP:AD:VAF:REP"
_GT_TEXT = ("0/0", "0/1", "1/1")


def _g(value):
    """_fmt_float for a finite value that is already a float32 number or an integer."""
    return "%g" % value


def _f32(value):
    return _F32.unpack(_F32.pack(value))[0]


class _Columns(object):
    """Selected allele records, one list per field (the fields of the calling tuple of _select_site)."""

    __slots__ = ("contig", "pos", "ref", "alt", "gt", "depth", "support", "pv", "p0", "p1", "p2", "non_alt", "rep")

    def __init__(self):
        for name in self.__slots__:
            setattr(self, name, [])

    def __len__(self):
        return len(self.pos)

    def record(self, i):
        """The calling tuple CandidateFinder._select_site would have built for row i."""
        prediction = [self.p0[i], self.p1[i], self.p2[i]]
        return (self.contig[i], self.pos[i], self.pos[i] + len(self.ref[i]), self.ref[i], [self.alt[i]],
                ([0, 0], [0, 1], [1, 1])[self.gt[i]], self.depth[i], [self.support[i]], self.pv[i], prediction,
                [self.non_alt[i]], self.rep[i])


def _select_batch(options, fasta_handler, file_name, batch_key, cols, leftovers):
    """Rows of one prediction batch -> cols (rows whose candidate list holds one allele: what the pipeline writes) or
    leftovers (calling tuples of rows with several alleles, through _select_site)."""
    with h5.File(file_name, "r") as hdf5_file:
        if "predictions" not in hdf5_file.keys():
            return
        base = "predictions/" + batch_key + "/"
        contigs = hdf5_file[base + "contigs"]
        positions = hdf5_file[base + "positions"]
        depths = hdf5_file[base + "depths"]
        candidates = hdf5_file[base + "candidates"]
        candidate_frequencies = hdf5_file[base + "candidate_frequency"]
        predictions = np.asarray(hdf5_file[base + "base_prediction"]).astype(np.float32)
    n = len(contigs)
    if n == 0:
        return
    names = [c.decode("UTF-8") if isinstance(c, bytes) else str(c) for c in (contigs.tolist() if hasattr(contigs, "tolist") else contigs)]
    pos = np.asarray(positions, dtype=np.int64).reshape(n)
    ref_bases, in_repeats = [""] * n, [False] * n
    one_contig = len(set(names)) == 1
    for contig in dict.fromkeys(names):
        rows = np.arange(n) if one_contig else np.array([k for k in range(n) if names[k] == contig], dtype=np.int64)
        p_rows = pos[rows]
        window = _ReferenceWindow(fasta_handler, contig, int(p_rows.min()) - 16, int(p_rows.max()) + 16)
        bases, flags = _in_repeat_many(window, contig, p_rows)
        if one_contig:
            ref_bases, in_repeats = bases, flags
        else:
            for k, b, f in zip(rows.tolist(), bases, flags):
                ref_bases[k], in_repeats[k] = b, f
    predictions = predictions.reshape(n, -1)
    if predictions.shape[1] != 3:
        raise ValueError("base_prediction of %s/%s has %d classes, expected 3" % (file_name, batch_key, predictions.shape[1]))
    gt = predictions.argmax(axis=1)                                        # first maximum, as the per-site code
    pv = predictions[np.arange(n), gt].tolist()
    non_alt = np.maximum(predictions[:, 1], predictions[:, 2]).tolist()
    p0, p1, p2 = predictions[:, 0].tolist(), predictions[:, 1].tolist(), predictions[:, 2].tolist()
    gt = gt.tolist()
    depth = np.asarray(depths).reshape(n).astype(np.int64).tolist()
    pos = pos.tolist()
    cand = np.asarray(candidates, dtype=object) if not isinstance(candidates, np.ndarray) else candidates
    freq = np.asarray(candidate_frequencies)
    single = cand.ndim == 2 and cand.shape == (n, 1) and freq.ndim == 2 and freq.shape == (n, 1) and cand.dtype.kind in "OU" and \
        freq.dtype.kind in "iu"
    if single:
        alleles = cand[:, 0].tolist()
        supports = freq[:, 0].astype(np.int64).tolist()
        single = all(isinstance(a, str) and a and not any(c in a for c in " ,'\"[]\n") for a in alleles)
    thresholds = {
        "1": (options.snp_p_value, options.snp_p_value_in_lc, options.report_snp_above_freq),
        "2": (options.insert_p_value, options.insert_p_value_in_lc, options.report_indel_above_freq),
        "3": (options.delete_p_value, options.delete_p_value_in_lc, options.report_indel_above_freq),
    }
    if not single:
        for i in range(n):
            rb = ref_bases[i]
            if rb not in _BASES or len(rb) != 1:
                continue
            _, calling = _select_site(options, names[i], pos[i], int(depth[i]), _parse_list_field(candidates[i]),
                                      [int(x) for x in _parse_list_field(candidate_frequencies[i])],
                                      [p0[i], p1[i], p2[i]], rb, in_repeats[i])
            if calling is not None:
                leftovers.append(calling)
        return
    bases = _BASES
    c_contig, c_pos, c_ref, c_alt, c_gt, c_depth, c_support = cols.contig, cols.pos, cols.ref, cols.alt, cols.gt, cols.depth, cols.support
    c_pv, c_p0, c_p1, c_p2, c_non_alt, c_rep = cols.pv, cols.p0, cols.p1, cols.p2, cols.non_alt, cols.rep
    for i in range(n):
        rb = ref_bases[i]
        if rb not in bases:                       # (one upper-cased character or "")
            continue
        code = alleles[i]
        allele = code[1:]
        if not set(allele) <= bases:
            continue
        vaf = float(supports[i]) / float(depth[i])    # for every valid allele, whatever its type: CandidateFinder.py:478 (depth 0 raises)
        entry = thresholds.get(code[0:1])
        if entry is None:
            continue
        rep = in_repeats[i]
        na = non_alt[i]
        by_probability = na >= (entry[1] if rep else entry[0])
        if not by_probability:
            if not 0 < entry[2] <= vaf:
                continue
        if code[0] == "3" and by_probability:
            c_ref.append(allele)                  # a deletion swaps roles: the deleted stretch is REF, the anchor base ALT (:490-501)
            c_alt.append(rb)
        else:
            c_ref.append(rb)
            c_alt.append(allele)
        c_contig.append(names[i])
        c_pos.append(pos[i])
        c_gt.append(gt[i])
        c_depth.append(depth[i])
        c_support.append(supports[i])
        c_pv.append(pv[i])
        c_p0.append(p0[i])
        c_p1.append(p1[i])
        c_p2.append(p2[i])
        c_non_alt.append(na)
        c_rep.append(rep)


def _format_single(options, cols, rows):
    """VCFWriter.format_sites for allele records taken as sites of their own: per row (len(REF), line, is_snp, selected)."""
    ref_lens, lines, snps, sels = [], [], [], []
    log10 = math.log10
    snp_cut, snp_cut_lc = options.snp_q_cutoff, options.snp_q_cutoff_in_lc
    indel_cut, indel_cut_lc = options.indel_q_cutoff, options.indel_q_cutoff_in_lc
    for i in rows:
        g = cols.gt[i]
        ref, alt = cols.ref[i], cols.alt[i]
        # genotype quality (VcfWriter.py:83-90 with one candidate), QUAL (:153)
        gq = cols.pv[i] if g != 0 else max(cols.p1[i], cols.p2[i])
        qual = max(1, int(-10 * log10(max(0.000000001, 1.0 - gq))))
        rep = cols.rep[i]
        is_snp = max(len(ref), len(alt)) == 1
        cutoff = (snp_cut_lc if rep else snp_cut) if is_snp else (indel_cut_lc if rep else indel_cut)
        depth, support = cols.depth[i], cols.support[i]
        line = "%s\t%d\t.\t%s\t%s\t%d\t%s\t.\t%s\t%s:%s:%d:%d:%d:%s:%s\n" % (
            cols.contig[i], cols.pos[i] + 1, ref, alt, qual, "refCall" if g == 0 else "PASS", _FORMAT, _GT_TEXT[g],
            _g(cols.non_alt[i]), qual, depth, support, _g(_f32(round(support / max(1, depth), 3))), "1" if rep else "0")
        ref_lens.append(len(ref))
        lines.append(line.encode())
        snps.append(is_snp)
        sels.append(g == 0 or qual <= cutoff)
    return ref_lens, lines, snps, sels


def _plain_options(options):
    from types import SimpleNamespace
    names = ("allowed_multiallelics", "snp_q_cutoff", "snp_q_cutoff_in_lc", "indel_q_cutoff", "indel_q_cutoff_in_lc",
             "snp_p_value", "snp_p_value_in_lc", "insert_p_value", "insert_p_value_in_lc", "delete_p_value", "delete_p_value_in_lc",
             "report_snp_above_freq", "report_indel_above_freq", "fasta")
    return SimpleNamespace(**{n: getattr(options, n) for n in names})


class _Rules(ctypes.Structure):      # pa_candidate_rules (include/pepper_amd_io.h)
    _fields_ = [("p_value", ctypes.c_double * 3), ("p_value_in_lc", ctypes.c_double * 3), ("report_above_freq", ctypes.c_double * 3),
                ("snp_q_cutoff", ctypes.c_double), ("snp_q_cutoff_in_lc", ctypes.c_double),
                ("indel_q_cutoff", ctypes.c_double), ("indel_q_cutoff_in_lc", ctypes.c_double)]


def _rules(options):
    try:
        return _Rules((ctypes.c_double * 3)(options.snp_p_value, options.insert_p_value, options.delete_p_value),
                      (ctypes.c_double * 3)(options.snp_p_value_in_lc, options.insert_p_value_in_lc, options.delete_p_value_in_lc),
                      (ctypes.c_double * 3)(options.report_snp_above_freq, options.report_indel_above_freq,
                                            options.report_indel_above_freq),
                      options.snp_q_cutoff, options.snp_q_cutoff_in_lc, options.indel_q_cutoff, options.indel_q_cutoff_in_lc)
    except TypeError:
        return None                  # thresholds that are not numbers: the Python path has the reference's behaviour for them


class _Segment(object):
    """The records selected from one prediction batch (plain attributes: a worker process returns these).
    contig: one name for all rows, or a list per row; pos int64 [m]; ref_len int32 [m]; snp / sel bool [m]; lines: list of
    bytes; pair(k) -> (REF, first ALT) and record(k) -> the calling tuple of CandidateFinder._select_site, for the few sites
    that carry several allele records."""

    def __init__(self, contig, pos, ref_len, snp, sel, lines, cols=None, raw=None):
        self.contig, self.pos, self.ref_len, self.snp, self.sel, self.lines = contig, pos, ref_len, snp, sel, lines
        self.cols, self.raw = cols, raw

    def __len__(self):
        return len(self.lines)

    def record(self, k):
        if self.cols is not None:
            return self.cols.record(k)
        row, flags, pos, depth, support, pred, letters, rep, blob, starts = self.raw
        i = int(row[k])
        allele = blob[starts[i] + 1:starts[i + 1] - 1].decode()
        base = chr(letters[i])
        ref, alt = (allele, base) if flags[k] & 4 else (base, allele)
        p = [float(x) for x in pred[i]]
        g = int(flags[k]) >> 4
        return (self.contig, int(pos[i]), int(pos[i]) + len(ref), ref, [alt], ([0, 0], [0, 1], [1, 1])[g], int(depth[i]),
                [int(support[i])], p[g], p, [max(p[1], p[2])], bool(rep[i]))

    def pair(self, k):
        if self.cols is not None:
            return self.cols.ref[k], self.cols.alt[k]
        record = self.record(k)
        return record[3], record[4][0]


def _native_batch(options, rules, fasta_handler, file_name, batch_key, files=None):
    """One prediction batch through pa_candidates_select_format (candidates.cpp): -> a _Segment, or None when the batch is not
    of the form that path takes (several contigs, candidate lists with several alleles, a reference window too wide, NaN
    probabilities, a zero depth the reference would divide by): _select_batch + _format_single then.  files: the caller's
    {file name: open h5.File} (one at a time)."""
    own = files is None
    if own:
        files = {}
    try:
        return _native_batch_open(options, rules, fasta_handler, file_name, batch_key, files)
    finally:
        if own:
            for f in files.values():
                f.close()


def _native_batch_open(options, rules, fasta_handler, file_name, batch_key, files):
    f = files.get(file_name)
    if f is None:
        for other in files.values():                       # one prediction file open at a time: the batches come file by file
            other.close()
        files.clear()
        f = files[file_name] = h5.File(file_name, "r")
        f.has_predictions = "predictions" in f.keys()
    if not f.has_predictions:
        return _Segment("", np.zeros(0, np.int64), np.zeros(0, np.int32), np.zeros(0, bool), np.zeros(0, bool), [])
    base = "predictions/" + batch_key + "/"
    whole = f.read_prediction_batch(base[:-1])
    if whole is not None:
        # the six datasets in one call through the locator (files of this package's writers): the same conditions as below
        contigs, blob, positions, depths, freq, pred = whole
        n = len(positions)
        if n == 0:
            return _Segment("", np.zeros(0, np.int64), np.zeros(0, np.int32), np.zeros(0, bool), np.zeros(0, bool), [])
        if (contigs != contigs[0]).any():
            return None
        first = contigs[0].tobytes()
        first = first[:first.index(b"\0")] if b"\0" in first else first
        return native_batch_arrays(options, rules, fasta_handler, first, n, positions, depths, freq, pred, blob, file_name + "/" + batch_key)
    contig_shape, contig_blob = f.read_strings_shaped(base + "contigs")
    shape, blob = f.read_strings_shaped(base + "candidates")
    positions = f[base + "positions"]
    depths = f[base + "depths"]
    freq = f[base + "candidate_frequency"]
    pred = f[base + "base_prediction"]
    n = int(np.prod(contig_shape)) if contig_shape else 1
    if n == 0:
        return _Segment("", np.zeros(0, np.int64), np.zeros(0, np.int32), np.zeros(0, bool), np.zeros(0, bool), [])
    first = contig_blob[:contig_blob.index(b"\0")]
    if contig_blob != (first + b"\0") * n or tuple(shape) != (n, 1) or np.shape(freq) != (n, 1) or np.asarray(freq).dtype.kind not in "iu":
        return None
    return native_batch_arrays(options, rules, fasta_handler, first, n, positions, depths, freq, pred, blob, file_name + "/" + batch_key)


def native_batch_arrays(options, rules, fasta_handler, first, n, positions, depths, freq, pred, blob, where="(memory)"):
    """Selection + record text of ONE prediction batch from its arrays (what _native_batch_open reads from a file; the fused
    call_variant hands over a batch as it writes it): first = the batch's one contig name (bytes), blob = its n candidate strings,
    each followed by a NUL.  -> _Segment, or None when the batch needs the per-row Python path."""
    file_name, batch_key = where, ""
    text = np.frombuffer(blob, np.uint8)
    ends = np.flatnonzero(text == 0)
    if len(ends) != n or _LIST_LUT[text].any():
        return None
    starts = np.empty(n + 1, np.int64)
    starts[0] = 0
    starts[1:] = ends + 1
    if (np.diff(starts) < 2).any():                       # an empty candidate string
        return None
    pred = np.ascontiguousarray(np.asarray(pred).astype(np.float32)).reshape(n, -1)
    if pred.shape[1] != 3:
        raise ValueError("base_prediction of %s/%s has %d classes, expected 3" % (file_name, batch_key, pred.shape[1]))
    contig = first.decode("UTF-8")
    pos = np.ascontiguousarray(np.asarray(positions, dtype=np.int64).reshape(n))
    window = _ReferenceWindow(fasta_handler, contig, int(pos.min()) - 16, int(pos.max()) + 16)
    if not window.text:
        return None
    text = window.text.encode("latin-1")
    letters, rep = np.empty(n, np.uint8), np.empty(n, np.uint8)
    if h5.load().pa_candidates_reference_flags(text, len(text), window.lo, n, pos.ctypes.data, letters.ctypes.data, rep.ctypes.data) < 0:
        raise h5.H5Error(h5.load().pa_h5_last_error().decode())
    depth = np.ascontiguousarray(np.asarray(depths).reshape(n).astype(np.int64))
    support = np.ascontiguousarray(np.asarray(freq)[:, 0].astype(np.int64))
    row, ref_len, flags = np.empty(n, np.int32), np.empty(n, np.int32), np.empty(n, np.uint8)
    offsets = np.empty(n + 1, np.int64)
    cap = len(blob) + n * (len(first) + 200)
    lines = np.empty(cap, np.uint8)                       # (not zero-filled, and only the written part is copied out)
    m = h5.load().pa_candidates_select_format(
        ctypes.byref(rules), first, n, pos.ctypes.data, depth.ctypes.data, support.ctypes.data, pred.ctypes.data,
        letters.ctypes.data, rep.ctypes.data, blob, starts.ctypes.data, 1, row.ctypes.data, ref_len.ctypes.data,
        flags.ctypes.data, ctypes.c_void_p(lines.ctypes.data), cap, offsets.ctypes.data)
    if m == -2:
        return None
    if m < 0:
        raise h5.H5Error(h5.load().pa_h5_last_error().decode())
    cut = offsets[:m + 1].tolist()
    raw_lines = lines[:cut[m]].tobytes()
    row, flags = row[:m], flags[:m]
    return _Segment(contig, pos[row], ref_len[:m].copy(), (flags & 1).astype(bool), (flags & 2).astype(bool),
                    [raw_lines[cut[k]:cut[k + 1]] for k in range(m)],
                    raw=(row.copy(), flags.copy(), pos, depth, support, pred, letters, rep, blob, starts))


_LIST_BYTES = np.frombuffer(b" ,'\"[]\n", np.uint8)
_LIST_LUT = np.zeros(256, bool)                              # the same set as a look-up (np.isin sorts per call)
_LIST_LUT[_LIST_BYTES] = True


def _python_batch(options, fasta_handler, file_name, batch_key, leftovers):
    cols = _Columns()
    _select_batch(options, fasta_handler, file_name, batch_key, cols, leftovers)
    ref_lens, lines, snps, sels = _format_single(options, cols, range(len(cols)))
    return _Segment(list(cols.contig), np.asarray(cols.pos, dtype=np.int64).reshape(len(cols)), np.asarray(ref_lens, dtype=np.int32),
                    np.asarray(snps, dtype=bool), np.asarray(sels, dtype=bool), lines, cols=cols)


def _part(options, pairs):
    """Selection + single-site formatting of some prediction batches: what a worker process returns."""
    fasta_handler = _fasta(options)
    rules = None if os.environ.get("PEPPER_AMD_CANDIDATES_PYTHON") == "1" else _rules(options)
    segments, leftovers, files = [], [], {}
    try:
        for file_name, batch_key in pairs:
            segment = _native_batch(options, rules, fasta_handler, file_name, batch_key, files) if rules is not None else None
            if segment is None:
                segment = _python_batch(options, fasta_handler, file_name, batch_key, leftovers)
            if len(segment):
                segments.append(segment)
    finally:
        for f in files.values():
            f.close()
    return segments, leftovers


def _parts(options, all_prediction_pair):
    """The batches in order, cut into one part per worker process (options.threads; one process below ~100 k rows per worker
    or when the FASTA reader is injected -- a factory need not survive pickling).  With the selection and the record text inside
    the I/O library a batch of 512 rows takes ~1 ms, so worker processes (0.3 s to start) only pay for very big jobs."""
    threads = max(1, int(getattr(options, "threads", 1) or 1))
    pairs = list(all_prediction_pair)
    python_form = os.environ.get("PEPPER_AMD_CANDIDATES_PYTHON") == "1"
    # (a batch read in one call and selected inside the library is ~0.4 ms: below ~2 000 batches one process is done before a pool
    # of spawned workers has started)
    if (threads == 1 or len(pairs) < (8 * threads if python_form else min(256 * threads, 2048)) or
            getattr(options, "fasta_handler_factory", None) is not None):
        return [_part(options, pairs)]
    import sys
    from multiprocessing import get_context
    plain = _plain_options(options)
    cut = [pairs[k * len(pairs) // threads:(k + 1) * len(pairs) // threads] for k in range(threads)]
    # spawned workers that do not re-import the caller's main module (pepper_amd.hostpipe._start_all's trick): the caller may
    # hold a HIP context (call_variant runs inference first), which must not be forked
    main = sys.modules.get("__main__")
    saved_spec, saved_file = getattr(main, "__spec__", None), getattr(main, "__file__", None)
    had_file = main is not None and hasattr(main, "__file__")
    try:
        if main is not None:
            main.__spec__ = None
            if had_file:
                del main.__file__
        pool = get_context("spawn").Pool(threads)
    finally:
        if main is not None:
            main.__spec__ = saved_spec
            if had_file:
                main.__file__ = saved_file
    with pool:
        futures = [pool.apply_async(_part, (plain, c)) for c in cut]
        return [f.get() for f in futures]           # in batch order; worker errors propagate


def _write_all(vcf, names, contig_code, starts, ref_lens, lines, is_snp, selected):
    """The sequential part of write_vcf_records (:150-218) as array operations -- the duplicate-start rule (a record whose
    start equals the start of the record before it is dropped: `last_position` only moves on records that are kept, so
    that is what the loop does), the routing into the five files -- then one bulk write per file."""
    if len(lines) == 0:
        return (0, 0, 0, 0, 0)
    lengths = np.fromiter(map(len, lines), np.int64, len(lines))
    keep = np.concatenate([[True], starts[1:] != starts[:-1]])           # (sic: compared across contigs too, :150-151)
    calling = keep & selected
    masks = (keep, keep & ~selected, calling, calling & is_snp, calling & ~is_snp)
    files = (vcf.vcf_file_full, vcf.vcf_file_pepper, vcf.vcf_file_variant_calling, vcf.vcf_file_variant_calling_snp,
             vcf.vcf_file_variant_calling_indel)
    totals = []
    for f, mask in zip(files, masks):
        idx = np.flatnonzero(mask)
        totals.append(len(idx))
        if len(idx):
            picked = lines if len(idx) == len(lines) else [lines[k] for k in idx.tolist()]
            f.write_columns(names, contig_code[idx], starts[idx], ref_lens[idx], lengths[idx], picked)
    return tuple(totals)


def process(options, all_prediction_pair, vcf, precomputed=None):
    """all_prediction_pair: [(prediction file, batch key)] as FindCandidates.candidate_finder lists them; vcf: an open
    VCFWriter.  -> (contigs, totals) with totals as write_vcf_records returns them."""
    segments, leftovers = [], []
    if precomputed is None:
        for part_segments, part_left in _parts(options, all_prediction_pair):
            segments.extend(part_segments)
            leftovers.extend(part_left)
    else:
        # the fused call_variant computed most batches' segments while it wrote them (fused.py); the batches keep the order of the
        # listing (which record of a duplicated site survives depends on it), the missing ones are done here from the file
        todo = []

        def flush():
            part_segments, part_left = _part(options, todo)
            segments.extend(part_segments)
            leftovers.extend(part_left)
            del todo[:]
        for pair in all_prediction_pair:
            seg = precomputed.get(pair)
            if seg is None:
                todo.append(pair)
                continue
            if todo:
                flush()
            if len(seg):
                segments.append(seg)
        if todo:
            flush()
    plain = _plain_options(options)
    sizes = np.fromiter(map(len, segments), np.int64, len(segments))
    n = int(sizes.sum())
    if leftovers:
        # files this package did not write (several alleles in a row's candidate list): the whole job through the tuple path,
        # whose record order (batch by batch) decides which duplicate of a site survives
        from pepper_amd.variant.CandidateFinder import find_candidates
        contigs, _, sites = find_candidates(options, None, all_prediction_pair)
        return contigs, vcf.write_vcf_records(sites, plain)
    if n == 0:
        return [], (0, 0, 0, 0, 0)
    # one stable order by (contig name, position), as _by_site / write_vcf_records sort; sites = runs of equal keys
    contig_names = sorted({c for seg in segments for c in ([seg.contig] if isinstance(seg.contig, str) else seg.contig)})
    rank = {c: k for k, c in enumerate(contig_names)}
    key_c = np.concatenate([np.full(len(seg), rank[seg.contig], np.int64) if isinstance(seg.contig, str) else
                            np.fromiter(map(rank.__getitem__, seg.contig), np.int64, len(seg)) for seg in segments])
    key_p = np.concatenate([seg.pos for seg in segments])
    lines = [line for seg in segments for line in seg.lines]
    order = np.lexsort((key_p, key_c))
    sc, sp = key_c[order], key_p[order]
    first = np.concatenate([[True], (sc[1:] != sc[:-1]) | (sp[1:] != sp[:-1])])
    site_at = np.flatnonzero(first)                        # sorted-order index of every site's first row
    site_size = np.diff(np.concatenate([site_at, [n]]))
    rows = order[site_at]                                  # original row of every site's first record
    row_list = rows.tolist()
    site_lines = [lines[k] for k in row_list]
    site_ref_len = np.concatenate([seg.ref_len for seg in segments]).astype(np.int64)[rows]
    site_snp = np.concatenate([seg.snp for seg in segments])[rows]
    site_sel = np.concatenate([seg.sel for seg in segments])[rows]
    seg_first = np.concatenate([[0], np.cumsum(sizes)])    # global row of every segment's first record
    order_list = None
    for s in np.flatnonzero(site_size > 1).tolist():
        # several allele records at one site: keep the first record of each (REF, first ALT) (:552-573), then the reference's merge
        if order_list is None:
            order_list = order.tolist()
        group, seen = [], []
        for k in order_list[site_at[s]:site_at[s] + site_size[s]]:
            si = int(np.searchsorted(seg_first, k, side="right")) - 1
            local = k - int(seg_first[si])
            pair = segments[si].pair(local)
            if pair not in seen:
                seen.append(pair)
                group.append(segments[si].record(local))
        record, is_snp, selected = VCFWriter.format_sites([group], plain)[0]
        site_lines[s], site_ref_len[s], site_snp[s], site_sel[s] = record[3], record[2], is_snp, selected
    totals = _write_all(vcf, contig_names, sc[site_at], sp[site_at], site_ref_len, site_lines, site_snp, site_sel)
    return contig_names, totals
