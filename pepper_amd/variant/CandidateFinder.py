"""Candidate selection from the predictions HDF5.

replaces: /root/reference/pepper_variant/modules/python/CandidateFinder.py
    small_chunk_stitch :356-529   per-site allele selection (margin SNP list + re-genotyping list)
    find_candidates    :532-581   sort, per-(contig, position) de-duplication on (ref, first alt)
    repeat_annotation  :279-297   (kmer_size = 1 only, as the live code calls it)
Same names, arguments and returned tuples, so VcfWriter / FindCandidates read as the reference's.

How it differs inside: predictions are handled per batch with numpy (argmax, the "non alt"
probability), the homopolymer test is a run-length pass over the 20-base context instead of the
quadratic k-mer scan, and the work is done in-process (options.threads > 1 fans file chunks out
over a process pool exactly as the reference partitions them).  Selection rules, thresholds,
string handling and ordering are the reference's.  prediction values go through float32, as they
do in the reference when they pass through the pybind struct CandidateImagePrediction
(region_summary.h:114-136, vector<float>).
"""
import concurrent.futures
import sys
from collections import defaultdict

import numpy as np

from pepper_amd import h5
from pepper_amd.variant.fasta import FASTA_handler

_BASES = frozenset("ACGT")


def chunks(file_names, threads):
    """Consecutive slices of `threads` items (CandidateFinder.py:14-19 in spirit; the caller passes
    the slice length, not a thread count)."""
    return [file_names[i:i + threads] for i in range(0, len(file_names), threads)]


def repeat_annotation(sequence, kmer_size=1):
    """Length of the homopolymer run each position sits in (within `sequence` only)."""
    if kmer_size != 1:
        raise NotImplementedError("only homopolymer annotation is used by the candidate finder")
    n = len(sequence)
    runs = [1] * n
    i = 0
    while i < n:
        j = i
        while j + 1 < n and sequence[j + 1] == sequence[i]:
            j += 1
        for k in range(i, j + 1):
            runs[k] = j - i + 1
        i = j + 1
    return runs


def _parse_list_field(value):
    """"['1A' '2AT']" / "['1A', '2AT']" / "[12 3]" -> ['1A', '2AT'] / ['12', '3'] (:376-383)."""
    if isinstance(value, np.ndarray) and value.ndim == 1 and value.size < 1000:
        # what str(array) + the parsing below yields for the rows the pipeline writes (one-dimensional, strings without
        # blanks / quotes / commas, or integers), without numpy's array printer: 20x less time per candidate
        if value.dtype.kind in "iu":
            return [str(v) for v in value.tolist()]
        if value.dtype.kind in "OU":
            items = value.tolist()
            if all(isinstance(v, str) and v and not any(c in v for c in " ,'\"[]\n") for v in items):
                return items
    text = str(value).strip("][").replace(",", " ")
    return [tok.strip("'") for tok in text.split()]


def _fasta(options):
    factory = getattr(options, "fasta_handler_factory", None)
    return factory(options.fasta) if factory is not None else FASTA_handler(options.fasta)


class _ReferenceWindow(object):
    """get_reference_sequence for the positions of one prediction batch out of ONE fetch of the span they cover (the
    batch's candidates sit within a few kb of each other; three seek + read calls per candidate otherwise)."""

    def __init__(self, fasta_handler, contig, lo, hi):
        self.contig, self.fasta = contig, fasta_handler
        self.lo = max(0, int(lo))
        self.text = fasta_handler.get_reference_sequence(contig, self.lo, int(hi)) if hi - lo <= 4000000 else None
        self.hi = self.lo + len(self.text) if self.text is not None else self.lo

    def get_reference_sequence(self, contig, start, stop):
        start = max(0, int(start))
        if self.text is None or contig != self.contig or start < self.lo or stop > self.hi:
            return self.fasta.get_reference_sequence(contig, start, stop)       # outside the window (or clamped by the contig end)
        return self.text[start - self.lo:max(start, int(stop)) - self.lo]


def _in_repeat(fasta_handler, contig, position):
    """The reference's low-complexity flag: a homopolymer run >= 5 touching [position-5, position+4)
    inside the context ref[position-10, position+10)  (:397-418)."""
    after = fasta_handler.get_reference_sequence(contig, position, position + 10).upper()
    before = fasta_handler.get_reference_sequence(contig, max(0, position - 10), position).upper()
    context = before + after
    runs = repeat_annotation(context, 1)
    at = len(before)
    window = runs[max(0, at - 5):min(len(runs), at + 4)]
    return bool(window) and max(window) >= 5


def _in_repeat_arrays(window, contig, positions):
    """_in_repeat and the upper-cased reference base for many positions of one contig out of the window's text, as arrays:
    (letters uint8 [n], 0 where the position lies outside the contig; low-complexity flags bool [n]) -- run lengths of the
    whole window once, then for each position the longest run among [position-5, position+4) as the 20-base context
    ref[position-10, position+10) sees it (a run is cut at the context's edges, which are also cut at the contig's).
    None when the window holds no text (a span too wide to fetch at once)."""
    positions = np.asarray(positions, dtype=np.int64)
    text = window.text
    if text is None or len(text) == 0:
        return None
    t = np.frombuffer(text.upper().encode("latin-1"), np.uint8)
    m = len(t)
    change = np.empty(m, bool)
    change[0] = True
    np.not_equal(t[1:], t[:-1], out=change[1:])
    starts = np.flatnonzero(change)
    run_id = np.cumsum(change) - 1
    run_start = starts[run_id]
    run_end = np.append(starts[1:], m)[run_id]
    q = positions - window.lo
    inside = (q >= 0) & (q < m)
    qc = np.clip(q, 0, m - 1)
    ctx_lo = np.maximum(np.maximum(positions - 10, 0) - window.lo, 0)
    ctx_hi = np.minimum(q + 10, m)
    flag = np.zeros(len(positions), bool)
    for k in range(-5, 4):
        idx = q + k
        ok = (idx >= ctx_lo) & (idx < ctx_hi)
        ic = np.clip(idx, 0, m - 1)
        run = np.minimum(run_end[ic], ctx_hi) - np.maximum(run_start[ic], ctx_lo)
        flag |= ok & (run >= 5)
    return np.where(inside, t[qc], 0).astype(np.uint8), flag & inside


def _in_repeat_many(window, contig, positions):
    """_in_repeat_arrays as lists: (upper-cased reference base or "" per position, low-complexity flag per position)."""
    arrays = _in_repeat_arrays(window, contig, positions)
    if arrays is None:
        positions = np.asarray(positions, dtype=np.int64)
        bases = [window.get_reference_sequence(contig, int(p), int(p) + 1).upper() for p in positions]
        return bases, [_in_repeat(window, contig, int(p)) for p in positions]
    letters, flags = arrays
    return [chr(c) if c else "" for c in letters.tolist()], flags.tolist()


def _select_site(options, contig, position, depth, alleles, supports, prediction, reference_base, in_repeat):
    """One site -> (margin tuple or None, re-genotyping tuple or None)."""
    predicted_genotype = max(range(len(prediction)), key=prediction.__getitem__)      # first maximum, as numpy.argmax
    genotype = ([0, 0], [0, 1], [1, 1])[predicted_genotype]
    prediction_value = prediction[predicted_genotype]
    non_alt_prediction = max(prediction[1], prediction[2])

    thresholds = {
        "1": (options.snp_p_value, options.snp_p_value_in_lc, options.report_snp_above_freq),
        "2": (options.insert_p_value, options.insert_p_value_in_lc, options.report_indel_above_freq),
        "3": (options.delete_p_value, options.delete_p_value_in_lc, options.report_indel_above_freq),
    }

    margin_alts, margin_support = [], []
    alts, alt_support, non_alt_predictions = [], [], []
    reference_allele = reference_base
    for allele_code, support in zip(alleles, supports):
        alt_type, allele = allele_code[0:1], allele_code[1:]
        if not set(allele) <= _BASES:
            continue
        # phasing list: SNPs at sites not called hom-ref (:432-436)
        if alt_type == "1" and predicted_genotype != 0:
            margin_alts.append(allele)
            margin_support.append(support)

        vaf = float(support) / float(depth)
        non_alt_predictions.append(non_alt_prediction)
        if alt_type not in thresholds:
            continue
        p_value, p_value_lc, report_above = thresholds[alt_type]
        by_probability = non_alt_prediction >= (p_value_lc if in_repeat else p_value)
        by_frequency = 0 < report_above <= vaf
        if not (by_probability or by_frequency):
            continue
        if alt_type == "3" and by_probability:
            # a deletion swaps roles: the deleted stretch becomes REF, the anchor base the ALT (:490-501)
            alts.append(reference_allele)
            reference_allele = allele
        else:
            # (a deletion admitted by frequency alone keeps the allele as ALT, as :502-504 does)
            alts.append(allele)
        alt_support.append(support)

    margin = None
    if margin_alts:
        margin = (contig, position, position + 1, reference_base, margin_alts, genotype, depth, margin_support,
                  prediction_value, prediction)
    calling = None
    if alts:
        calling = (contig, position, position + len(reference_allele), reference_allele, alts, genotype, depth,
                   alt_support, prediction_value, prediction, non_alt_predictions, in_repeat)
    return margin, calling


def small_chunk_stitch(options, file_chunks):
    fasta_handler = _fasta(options)
    selected_candidate_list_margin = []
    selected_candidate_list_deepvariant = []
    for file_name, batch_key in file_chunks:
        with h5.File(file_name, "r") as hdf5_file:
            if "predictions" not in hdf5_file.keys():
                continue
            base = "predictions/" + batch_key + "/"
            contigs = hdf5_file[base + "contigs"]
            positions = hdf5_file[base + "positions"]
            depths = hdf5_file[base + "depths"]
            candidates = hdf5_file[base + "candidates"]
            candidate_frequencies = hdf5_file[base + "candidate_frequency"]
            base_predictions = np.asarray(hdf5_file[base + "base_prediction"]).astype(np.float32)

        n = len(contigs)
        if n == 0:
            continue
        # per batch, with numpy: the contig groups, one reference fetch per group, the reference base and the low-complexity
        # flag of every position (_in_repeat_many); the per-site rules then run on plain Python scalars
        names = [c.decode("UTF-8") if isinstance(c, bytes) else str(c) for c in (contigs.tolist() if hasattr(contigs, "tolist") else contigs)]
        pos = np.asarray(positions, dtype=np.int64).reshape(n)
        ref_bases = [""] * n
        in_repeats = [False] * n
        for contig in dict.fromkeys(names):
            rows = np.array([k for k in range(n) if names[k] == contig], dtype=np.int64) if len(set(names)) > 1 else np.arange(n)
            p_rows = pos[rows]
            window = _ReferenceWindow(fasta_handler, contig, int(p_rows.min()) - 16, int(p_rows.max()) + 16)
            bases, flags = _in_repeat_many(window, contig, p_rows)
            for k, base, flag in zip(rows.tolist(), bases, flags):
                ref_bases[k] = base
                in_repeats[k] = flag
        depth_list = np.asarray(depths).reshape(n).tolist()
        prediction_rows = base_predictions.reshape(n, -1).tolist()           # float32 values as Python floats
        position_list = pos.tolist()
        for i in range(n):
            reference_base = ref_bases[i]
            if reference_base not in _BASES or len(reference_base) != 1:
                continue
            alleles = _parse_list_field(candidates[i])
            supports = [int(x) for x in _parse_list_field(candidate_frequencies[i])]
            margin, calling = _select_site(options, names[i], position_list[i], int(depth_list[i]), alleles, supports,
                                           prediction_rows[i], reference_base, in_repeats[i])
            if margin is not None:
                selected_candidate_list_margin.append(margin)
            if calling is not None:
                selected_candidate_list_deepvariant.append(calling)
    return selected_candidate_list_margin, selected_candidate_list_deepvariant


def _by_site(selected):
    """Sorted by (contig, position); per site keep the first record of each (ref, first alt) (:552-573)."""
    sites, seen = defaultdict(list), defaultdict(list)
    contigs = []
    for candidate in sorted(selected, key=lambda x: (x[0], x[1])):
        if candidate[0] not in contigs:
            contigs.append(candidate[0])
        key = (candidate[0], candidate[1])
        pair = (candidate[3], candidate[4][0])
        if pair in seen[key]:
            continue
        seen[key].append(pair)
        sites[key].append(candidate)
    return contigs, sites


def find_candidates(options, input_dir, all_prediction_pair):
    threads = max(1, int(getattr(options, "threads", 1) or 1))
    file_chunks = chunks(all_prediction_pair, max(2, int(len(all_prediction_pair) / threads) + 1))
    phasing, calling = [], []
    if threads == 1 or len(file_chunks) <= 1:
        for file_chunk in file_chunks:
            margin_part, calling_part = small_chunk_stitch(options, file_chunk)
            phasing.extend(margin_part)
            calling.extend(calling_part)
    else:
        with concurrent.futures.ProcessPoolExecutor(max_workers=threads) as executor:
            futures = [executor.submit(small_chunk_stitch, options, file_chunk) for file_chunk in file_chunks]
            for fut in futures:            # submission order: deterministic, unlike as_completed
                margin_part, calling_part = fut.result()    # worker errors propagate (reference logs and drops)
                phasing.extend(margin_part)
                calling.extend(calling_part)
    _, phasing_sites = _by_site(phasing)
    contigs, calling_sites = _by_site(calling)
    if not calling and not phasing:
        sys.stderr.write("INFO: NO CANDIDATES SELECTED FROM " + str(input_dir) + "\n")
    return contigs, phasing_sites, calling_sites
