"""Test infrastructure: synthetic pileups (reads + CIGARs over a random reference), the flat
`oracle_pileup` marshalling shared by the oracle libraries, and loaders for them."""
import ctypes
import os
import subprocess

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(REPO, "oracle")

OP_M, OP_I, OP_D, OP_N, OP_S, OP_H, OP_P, OP_EQ, OP_X = range(9)

c_i64p = ctypes.POINTER(ctypes.c_int64)


class Pileup(ctypes.Structure):
    _fields_ = [("region_start", ctypes.c_int64), ("region_end", ctypes.c_int64),
                ("reference", ctypes.c_char_p), ("reference_len", ctypes.c_int64),
                ("n_reads", ctypes.c_int32),
                ("read_pos", ctypes.c_void_p), ("read_reverse", ctypes.c_void_p), ("read_mapq", ctypes.c_void_p),
                ("seq_offset", ctypes.c_void_p), ("seq", ctypes.c_void_p), ("qual", ctypes.c_void_p),
                ("cigar_offset", ctypes.c_void_p), ("cigar_op", ctypes.c_void_p), ("cigar_len", ctypes.c_void_p)]


class SummaryParams(ctypes.Structure):
    _fields_ = [("min_snp_baseq", ctypes.c_double), ("min_indel_baseq", ctypes.c_double),
                ("snp_freq_threshold", ctypes.c_double), ("insert_freq_threshold", ctypes.c_double),
                ("delete_freq_threshold", ctypes.c_double), ("min_coverage_threshold", ctypes.c_double),
                ("snp_candidate_freq_threshold", ctypes.c_double),
                ("indel_candidate_freq_threshold", ctypes.c_double),
                ("candidate_support_threshold", ctypes.c_double), ("skip_indels", ctypes.c_int32),
                ("candidate_region_start", ctypes.c_int64), ("candidate_region_end", ctypes.c_int64),
                ("candidate_window_size", ctypes.c_int32), ("feature_size", ctypes.c_int32)]


class SummaryResult(ctypes.Structure):
    _fields_ = [("n", ctypes.c_int64), ("positions", ctypes.POINTER(ctypes.c_int64)),
                ("depths", ctypes.POINTER(ctypes.c_int32)), ("candidate_frequency", ctypes.POINTER(ctypes.c_int32)),
                ("images", ctypes.POINTER(ctypes.c_int32)), ("candidates", ctypes.POINTER(ctypes.c_char)),
                ("candidates_bytes", ctypes.c_int64)]


# ONT R9 guppy5 sup preset thresholds (SetParameters.py:12-60 ont_r9_guppy5_sup)
ONT_PARAMS = dict(min_snp_baseq=1, min_indel_baseq=1, snp_freq_threshold=0.10, insert_freq_threshold=0.15,
                  delete_freq_threshold=0.15, min_coverage_threshold=3, snp_candidate_freq_threshold=0.10,
                  indel_candidate_freq_threshold=0.12, candidate_support_threshold=2, skip_indels=0)


# image-generation thresholds of the reference's other presets (pepper_variant/modules/argparse/SetParameters.py:
# ont_r9_guppy5_sup :20-36 = ONT_PARAMS above with indel_candidate 0.10 -- the tests keep 0.12 from round 1 as one more
# variant --, ont_r10_q20 :122-146, hifi :176-203, clr :229-256)
PRESET_PARAMS = {
    "ont_r10_q20": dict(min_snp_baseq=1, min_indel_baseq=1, snp_freq_threshold=0.10, insert_freq_threshold=0.10,
                        delete_freq_threshold=0.10, min_coverage_threshold=3, snp_candidate_freq_threshold=0.10,
                        indel_candidate_freq_threshold=0.10, candidate_support_threshold=2),
    "hifi": dict(min_snp_baseq=10, min_indel_baseq=10, snp_freq_threshold=0.10, insert_freq_threshold=0.12,
                 delete_freq_threshold=0.10, min_coverage_threshold=2, snp_candidate_freq_threshold=0.10,
                 indel_candidate_freq_threshold=0.10, candidate_support_threshold=2),
    "clr": dict(min_snp_baseq=0, min_indel_baseq=0, snp_freq_threshold=0.10, insert_freq_threshold=0.12,
                delete_freq_threshold=0.12, min_coverage_threshold=3, snp_candidate_freq_threshold=0.10,
                indel_candidate_freq_threshold=0.12, candidate_support_threshold=2),
}


class FlatPileup(object):
    """Owns the numpy buffers behind a Pileup struct."""

    def __init__(self, region_start, region_end, reference, reads):
        """reads: list of dict(pos, reverse, mapq, seq(str), qual(list/array), cigar [(op,len),...])."""
        self.region_start, self.region_end = int(region_start), int(region_end)
        self.reference = reference.encode() if isinstance(reference, str) else bytes(reference)
        n = len(reads)
        self.read_pos = np.array([r["pos"] for r in reads], np.int64).reshape(n)
        self.read_reverse = np.array([1 if r["reverse"] else 0 for r in reads], np.uint8).reshape(n)
        self.read_mapq = np.array([r["mapq"] for r in reads], np.int32).reshape(n)
        self.seq_offset = np.zeros(n + 1, np.int64)
        self.cigar_offset = np.zeros(n + 1, np.int64)
        for i, r in enumerate(reads):
            self.seq_offset[i + 1] = self.seq_offset[i] + len(r["seq"])
            self.cigar_offset[i + 1] = self.cigar_offset[i] + len(r["cigar"])
        self.seq = np.frombuffer("".join(r["seq"] for r in reads).encode() + b"\0", np.uint8).copy()
        self.qual = np.concatenate([np.asarray(r["qual"], np.uint8) for r in reads] + [np.zeros(1, np.uint8)])
        ops = [c for r in reads for c in r["cigar"]]
        self.cigar_op = np.array([c[0] for c in ops] + [0], np.int32)
        self.cigar_len = np.array([c[1] for c in ops] + [0], np.int32)
        self.n_reads = n

    def struct(self):
        return Pileup(self.region_start, self.region_end, self.reference, len(self.reference), self.n_reads,
                      self.read_pos.ctypes.data, self.read_reverse.ctypes.data, self.read_mapq.ctypes.data,
                      self.seq_offset.ctypes.data, self.seq.ctypes.data, self.qual.ctypes.data,
                      self.cigar_offset.ctypes.data, self.cigar_op.ctypes.data, self.cigar_len.ctypes.data)


def make_params(candidate_start, candidate_end, **over):
    d = dict(ONT_PARAMS)
    d.update(over)
    return SummaryParams(d["min_snp_baseq"], d["min_indel_baseq"], d["snp_freq_threshold"],
                         d["insert_freq_threshold"], d["delete_freq_threshold"], d["min_coverage_threshold"],
                         d["snp_candidate_freq_threshold"], d["indel_candidate_freq_threshold"],
                         d["candidate_support_threshold"], int(d["skip_indels"]), int(candidate_start),
                         int(candidate_end), 32, 26)


def result_to_dict(res, window=33, features=26):
    n = int(res.n)
    names = ctypes.string_at(res.candidates, res.candidates_bytes).split(b"\0")[:n] if n else []
    return dict(positions=np.ctypeslib.as_array(res.positions, (max(n, 1),))[:n].copy(),
                depths=np.ctypeslib.as_array(res.depths, (max(n, 1),))[:n].copy(),
                candidate_frequency=np.ctypeslib.as_array(res.candidate_frequency, (max(n, 1),))[:n].copy(),
                images=np.ctypeslib.as_array(res.images, (max(n, 1) * window * features,))[:n * window * features]
                .reshape(n, window, features).copy(),
                candidates=[s.decode() for s in names])


def _build_oracle():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR, "all"], check=True)


def load_restatement():
    path = os.path.join(ORACLE_DIR, "libpileup_oracle.so")
    if not os.path.exists(path):
        _build_oracle()
    lib = ctypes.CDLL(path)
    lib.oracle_variant_generate_summary.argtypes = [ctypes.POINTER(Pileup), ctypes.POINTER(SummaryParams),
                                                    ctypes.POINTER(SummaryResult)]
    lib.oracle_free_summary.argtypes = [ctypes.POINTER(SummaryResult)]
    lib.oracle_polish_generate_summary.restype = ctypes.c_int64
    lib.oracle_polish_generate_summary.argtypes = [ctypes.POINTER(Pileup), ctypes.c_int64, ctypes.c_int64,
                                                   ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
    return lib


def load_reference_encoder():
    """oracle/_ref/libref_variant_encoder.so (the reference's own C++), or None if not built."""
    path = os.path.join(ORACLE_DIR, "_ref", "libref_variant_encoder.so")
    if not os.path.exists(path):
        if not os.path.exists("/root/reference"):
            return None
        _build_oracle()
    lib = ctypes.CDLL(path)
    lib.ref_variant_generate_summary.argtypes = [ctypes.POINTER(Pileup), ctypes.POINTER(SummaryParams),
                                                 ctypes.POINTER(SummaryResult)]
    lib.ref_variant_free.argtypes = [ctypes.POINTER(SummaryResult)]
    return lib


def load_reference_polish_encoder():
    """oracle/_ref/libref_polish_encoder.so (the reference's own SummaryGenerator), or None if not built."""
    path = os.path.join(ORACLE_DIR, "_ref", "libref_polish_encoder.so")
    if not os.path.exists(path):
        if not os.path.exists("/root/reference"):
            return None
        _build_oracle()
    lib = ctypes.CDLL(path)
    lib.ref_polish_generate_summary.restype = ctypes.c_int64
    lib.ref_polish_generate_summary.argtypes = [ctypes.POINTER(Pileup), ctypes.c_int64, ctypes.c_int64,
                                                ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
    return lib


def run_polish_reference(lib, pileup, start_pos, end_pos):
    p = pileup.struct()
    rows = lib.ref_polish_generate_summary(ctypes.byref(p), start_pos, end_pos, None, None, 0)
    img = np.zeros((rows, 10), np.uint8)
    pos = np.zeros((rows, 2), np.int64)
    lib.ref_polish_generate_summary(ctypes.byref(p), start_pos, end_pos, img.ctypes.data, pos.ctypes.data, rows)
    return img, pos


def run_variant(lib, pileup, params, reference_impl=False):
    res = SummaryResult()
    p = pileup.struct()
    fn, free = ((lib.ref_variant_generate_summary, lib.ref_variant_free) if reference_impl
                else (lib.oracle_variant_generate_summary, lib.oracle_free_summary))
    rc = fn(ctypes.byref(p), ctypes.byref(params), ctypes.byref(res))
    assert rc == 0
    out = result_to_dict(res)
    free(ctypes.byref(res))
    return out


def run_polish_oracle(lib, pileup, start_pos, end_pos):
    p = pileup.struct()
    rows = lib.oracle_polish_generate_summary(ctypes.byref(p), start_pos, end_pos, None, None, 0)
    img = np.zeros((rows, 10), np.uint8)
    pos = np.zeros((rows, 2), np.int64)
    lib.oracle_polish_generate_summary(ctypes.byref(p), start_pos, end_pos, img.ctypes.data, pos.ctypes.data, rows)
    return img, pos


# ---- synthetic pileups --------------------------------------------------------------------------
def random_reference(rng, length, n_frac=0.0, lower_frac=0.0):
    ref = rng.choice(list("ACGT"), size=length)
    if n_frac:
        ref[rng.random(length) < n_frac] = "N"
    s = "".join(ref)
    if lower_frac:
        s = "".join(c.lower() if rng.random() < lower_frac else c for c in s)
    return s


def simulate_reads(rng, ref_seq, ref_offset, n_reads, read_len=(200, 900), snp_sites=None, err=0.03,
                   ins_rate=0.01, del_rate=0.012, clip_rate=0.3, skip_rate=0.0, low_q_rate=0.05,
                   mapq_zero_rate=0.03, eqx=False, max_indel=8, long_indel_rate=0.0, indel_sites=None):
    """Reads aligned to ref_seq (genomic coordinate of ref_seq[0] = ref_offset).

    snp_sites: {genomic pos: (alt base, fraction of reads carrying it)} planted variants.
    indel_sites: {genomic anchor pos: ("I", inserted string, fraction) | ("D", length, fraction)}: the
    indel follows the aligned base at the anchor.
    Returns list of read dicts with consistent seq / qual / cigar."""
    reads = []
    L = len(ref_seq)
    snp_sites = snp_sites or {}
    indel_sites = indel_sites or {}
    for _ in range(n_reads):
        length = int(rng.integers(read_len[0], read_len[1]))
        start = int(rng.integers(-length // 2, L - 10))
        pos = max(0, start)
        reverse = bool(rng.random() < 0.5)
        seq, cigar = [], []

        def push(op, n):
            if n <= 0:
                return
            if cigar and cigar[-1][0] == op:
                cigar[-1] = (op, cigar[-1][1] + n)
            else:
                cigar.append((op, n))

        if rng.random() < clip_rate:
            n = int(rng.integers(1, 30))
            seq.extend(rng.choice(list("ACGT"), size=n))
            push(OP_S, n)
        rp = pos
        carried = {p: rng.random() < frac for p, (alt, frac) in snp_sites.items()}
        while rp < min(L, pos + length):
            u = rng.random()
            g = ref_offset + rp
            if u < ins_rate and cigar and cigar[-1][0] in (OP_M, OP_EQ, OP_X):
                n = int(rng.integers(1, max_indel)) if rng.random() >= long_indel_rate else int(rng.integers(55, 70))
                seq.extend(rng.choice(list("ACGT"), size=n))
                push(OP_I, n)
            elif u < ins_rate + del_rate and cigar and cigar[-1][0] in (OP_M, OP_EQ, OP_X):
                n = int(rng.integers(1, max_indel)) if rng.random() >= long_indel_rate else int(rng.integers(55, 70))
                push(OP_D, n)
                rp += n
            elif u < ins_rate + del_rate + skip_rate and cigar:
                n = int(rng.integers(1, 20))
                op = OP_N if rng.random() < 0.5 else OP_P
                push(op, n)
                seq.extend(rng.choice(list("ACGT"), size=n))   # N/P also consume read bases in the reference
                rp += n
            else:
                rb = ref_seq[rp].upper()
                b = rb if rb in "ACGT" else "A"
                if g in snp_sites and carried[g]:
                    b = snp_sites[g][0]
                elif rng.random() < err:
                    b = rng.choice([c for c in "ACGT" if c != b])
                seq.append(b)
                push((OP_EQ if b == rb else OP_X) if eqx else OP_M, 1)
                rp += 1
                if g in indel_sites and rng.random() < indel_sites[g][2] and rp < L - 80:
                    kind, payload, _ = indel_sites[g]
                    if kind == "I":
                        seq.extend(payload)
                        push(OP_I, len(payload))
                    else:
                        push(OP_D, int(payload))
                        rp += int(payload)
        if rng.random() < clip_rate:
            n = int(rng.integers(1, 30))
            seq.extend(rng.choice(list("ACGT"), size=n))
            push(OP_S, n)
        if rng.random() < 0.1:
            push(OP_H, int(rng.integers(1, 50)))
        if not seq:
            continue
        qual = rng.integers(2, 40, size=len(seq))
        qual[rng.random(len(seq)) < low_q_rate] = 0
        reads.append(dict(pos=ref_offset + pos, reverse=reverse,
                          mapq=0 if rng.random() < mapq_zero_rate else int(rng.integers(1, 61)),
                          seq="".join(seq), qual=qual.astype(np.uint8), cigar=cigar))
    reads.sort(key=lambda r: r["pos"])
    return reads
