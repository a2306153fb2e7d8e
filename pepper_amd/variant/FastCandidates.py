"""predictions HDF5 -> the five candidate VCFs, column-wise.

The same rules as CandidateFinder.small_chunk_stitch / find_candidates and VcfWriter.write_vcf_records
(/root/reference/pepper_variant/modules/python/CandidateFinder.py:356-581, VcfWriter.py:48-218), which build one
Python tuple per selected allele, one dict entry per site, and format and write one record at a time (~35 us per
candidate).  Here a prediction batch is handled as numpy columns (genotype, "non alt" probability, reference base and
low-complexity flag of every row at once), the selected rows of all batches are ordered and grouped by site with one stable
lexsort, the records of sites that carry ONE allele record -- nearly all of them -- are formatted in a single pass over plain
Python lists, and each file is written with one join, its virtual offsets and its tabix index computed arithmetically.
Sites with several allele records go through the reference-shaped merge (VCFWriter.candidate_list_to_variant) unchanged.
The phasing ("margin") list of find_candidates is not built: the writer never reads it (FindCandidates.py:170-176).

tests/test_candidate_finder.py holds this path to the tuple path: same bytes in every .vcf.gz once decompressed, same
index content.  Text rendering stays UNPINNED against pysam / htslib (absent from this image), as for the tuple path.
"""
import math

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
        entry = thresholds.get(code[0:1])
        if entry is None:
            continue
        allele = code[1:]
        if not set(allele) <= bases:
            continue
        rep = in_repeats[i]
        na = non_alt[i]
        by_probability = na >= (entry[1] if rep else entry[0])
        if not by_probability:
            if not 0 < entry[2] <= float(supports[i]) / float(depth[i]):
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


def _part(options, pairs):
    """Selection + single-site formatting of some prediction batches: what a worker process returns."""
    fasta_handler = _fasta(options)
    cols, leftovers = _Columns(), []
    for file_name, batch_key in pairs:
        _select_batch(options, fasta_handler, file_name, batch_key, cols, leftovers)
    return cols, leftovers, _format_single(options, cols, range(len(cols)))


def _parts(options, all_prediction_pair):
    """The batches in order, cut into one part per worker process (options.threads; one process below ~100 k rows or when the
    FASTA reader is injected -- a factory need not survive pickling)."""
    threads = max(1, int(getattr(options, "threads", 1) or 1))
    pairs = list(all_prediction_pair)
    if threads == 1 or len(pairs) < 8 * threads or getattr(options, "fasta_handler_factory", None) is not None:
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


def process(options, all_prediction_pair, vcf):
    """all_prediction_pair: [(prediction file, batch key)] as FindCandidates.candidate_finder lists them; vcf: an open
    VCFWriter.  -> (contigs, totals) with totals as write_vcf_records returns them."""
    parts = _parts(options, all_prediction_pair)
    cols, leftovers = _Columns(), []
    ref_lens, lines, snps, sels = [], [], [], []
    for part_cols, part_left, (r, ln, sn, se) in parts:
        for name in _Columns.__slots__:
            getattr(cols, name).extend(getattr(part_cols, name))
        leftovers.extend(part_left)
        ref_lens.extend(r)
        lines.extend(ln)
        snps.extend(sn)
        sels.extend(se)
    n = len(cols)
    plain = _plain_options(options)
    if leftovers:
        # files this package did not write (several alleles in a row's candidate list): every record through the tuple path
        from pepper_amd.variant.CandidateFinder import _by_site
        contigs, sites = _by_site([cols.record(i) for i in range(n)] + leftovers)
        return contigs, vcf.write_vcf_records(sites, plain)
    if n == 0:
        return [], (0, 0, 0, 0, 0)
    # one stable order by (contig name, position), as _by_site / write_vcf_records sort; sites = runs of equal keys
    contig_names = sorted(set(cols.contig))
    rank = {c: k for k, c in enumerate(contig_names)}
    key_c = np.fromiter(map(rank.__getitem__, cols.contig), np.int64, n)
    key_p = np.asarray(cols.pos, dtype=np.int64).reshape(n)
    order = np.lexsort((key_p, key_c))
    sc, sp = key_c[order], key_p[order]
    first = np.concatenate([[True], (sc[1:] != sc[:-1]) | (sp[1:] != sp[:-1])])
    site_at = np.flatnonzero(first)                        # sorted-order index of every site's first row
    site_size = np.diff(np.concatenate([site_at, [n]]))
    rows = order[site_at]                                  # original row of every site's first record
    row_list = rows.tolist()
    site_lines = [lines[k] for k in row_list]
    site_ref_len = np.asarray(ref_lens, dtype=np.int64)[rows]
    site_snp = np.asarray(snps, dtype=bool)[rows]
    site_sel = np.asarray(sels, dtype=bool)[rows]
    order_list = None
    for s in np.flatnonzero(site_size > 1).tolist():
        # several allele records at one site: keep the first record of each (REF, first ALT) (:552-573), then the reference's merge
        if order_list is None:
            order_list = order.tolist()
        group, seen = [], []
        for k in order_list[site_at[s]:site_at[s] + site_size[s]]:
            pair = (cols.ref[k], cols.alt[k])
            if pair not in seen:
                seen.append(pair)
                group.append(cols.record(k))
        record, is_snp, selected = VCFWriter.format_sites([group], plain)[0]
        site_lines[s], site_ref_len[s], site_snp[s], site_sel[s] = record[3], record[2], is_snp, selected
    totals = _write_all(vcf, contig_names, sc[site_at], sp[site_at], site_ref_len, site_lines, site_snp, site_sel)
    return contig_names, totals
