"""TEST / BENCH INFRASTRUCTURE -- never imported by pepper_amd/.

The CPU side of `bench.py --model encoder`'s `cpu_baseline`: the reference's own RegionalSummaryGenerator
(oracle/_ref/libref_variant_encoder.so, compiled by oracle/Makefile from /root/reference/pepper_variant/modules/cpp/
region_summary.cpp where it lies) -- or, where that build did not travel, the oracle's C++ restatement
(libpileup_oracle.so) -- timed on E-syn regions (pepper_amd.synthetic.encoder_region), one region per call, the way
AlignmentSummarizer.create_summary calls it (AlignmentSummarizer.py:220-238).

    python oracle/encoder_cpu.py --seed S --seconds T     one worker: loops the encoder on one region, prints
                                                          {"bases", "regions", "seconds", "kind"}
The reference's image generation is one such single-thread worker per core (ImageGenerationUI.py:262-274), so the
all-core figure is the sum over concurrent workers (bench.py starts them).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

ONT = dict(min_snp_baseq=1, min_indel_baseq=1, snp_freq_threshold=0.10, insert_freq_threshold=0.15,
           delete_freq_threshold=0.15, min_coverage_threshold=3, snp_candidate_freq_threshold=0.10,
           indel_candidate_freq_threshold=0.12, candidate_support_threshold=2, skip_indels=0)


class Pileup(ctypes.Structure):     # oracle/pileup_abi.h oracle_pileup
    _fields_ = [("region_start", ctypes.c_int64), ("region_end", ctypes.c_int64), ("reference", ctypes.c_char_p),
                ("reference_len", ctypes.c_int64), ("n_reads", ctypes.c_int32), ("read_pos", ctypes.c_void_p),
                ("read_reverse", ctypes.c_void_p), ("read_mapq", ctypes.c_void_p), ("seq_offset", ctypes.c_void_p),
                ("seq", ctypes.c_void_p), ("qual", ctypes.c_void_p), ("cigar_offset", ctypes.c_void_p),
                ("cigar_op", ctypes.c_void_p), ("cigar_len", ctypes.c_void_p)]


class Params(ctypes.Structure):     # oracle_summary_params
    _fields_ = [("min_snp_baseq", ctypes.c_double), ("min_indel_baseq", ctypes.c_double),
                ("snp_freq_threshold", ctypes.c_double), ("insert_freq_threshold", ctypes.c_double),
                ("delete_freq_threshold", ctypes.c_double), ("min_coverage_threshold", ctypes.c_double),
                ("snp_candidate_freq_threshold", ctypes.c_double), ("indel_candidate_freq_threshold", ctypes.c_double),
                ("candidate_support_threshold", ctypes.c_double), ("skip_indels", ctypes.c_int32),
                ("candidate_region_start", ctypes.c_int64), ("candidate_region_end", ctypes.c_int64),
                ("candidate_window_size", ctypes.c_int32), ("feature_size", ctypes.c_int32)]


class Result(ctypes.Structure):     # oracle_summary_result
    _fields_ = [("n", ctypes.c_int64), ("positions", ctypes.c_void_p), ("depths", ctypes.c_void_p),
                ("candidate_frequency", ctypes.c_void_p), ("images", ctypes.c_void_p), ("candidates", ctypes.c_void_p),
                ("candidates_bytes", ctypes.c_int64)]


def load():
    """-> (kind, run(pileup, params) -> n candidates)"""
    ref = os.path.join(HERE, "_ref", "libref_variant_encoder.so")
    if os.path.exists(ref):
        lib = ctypes.CDLL(ref)
        fn, free, kind = lib.ref_variant_generate_summary, lib.ref_variant_free, "reference"
    else:
        lib = ctypes.CDLL(os.path.join(HERE, "libpileup_oracle.so"))
        fn, free, kind = lib.oracle_variant_generate_summary, lib.oracle_free_summary, "port"
    fn.argtypes = [ctypes.POINTER(Pileup), ctypes.POINTER(Params), ctypes.POINTER(Result)]
    free.argtypes = [ctypes.POINTER(Result)]

    def run(p, q):
        res = Result()
        if fn(ctypes.byref(p), ctypes.byref(q), ctypes.byref(res)) != 0:
            raise RuntimeError("encoder oracle failed")
        n = int(res.n)
        free(ctypes.byref(res))
        return n
    return kind, run


def region_structs(region):
    ref, flat, rs, re_ = region
    p = Pileup(rs, re_, ref, len(ref), flat["n_reads"], flat["read_pos"].ctypes.data, flat["read_reverse"].ctypes.data,
               flat["read_mapq"].ctypes.data, flat["seq_offset"].ctypes.data, flat["seq"].ctypes.data,
               flat["qual"].ctypes.data, flat["cigar_offset"].ctypes.data, flat["cigar_op"].ctypes.data,
               flat["cigar_len"].ctypes.data)
    q = Params(*[ONT[k] for k in ("min_snp_baseq", "min_indel_baseq", "snp_freq_threshold", "insert_freq_threshold",
                                  "delete_freq_threshold", "min_coverage_threshold", "snp_candidate_freq_threshold",
                                  "indel_candidate_freq_threshold", "candidate_support_threshold")],
               int(ONT["skip_indels"]), rs + 100, re_ - 100, 32, 26)
    return p, q


def time_regions(regions, seconds):
    """Loop the CPU encoder over `regions` until `seconds` have passed (at least one region)."""
    kind, run = load()
    structs = [region_structs(r) for r in regions]
    bases = done = 0
    t0 = time.perf_counter()
    while done < 1 or time.perf_counter() - t0 < seconds:
        p, q = structs[done % len(structs)]
        run(p, q)
        bases += int(regions[done % len(regions)][1]["seq_offset"][-1])
        done += 1
    return {"bases": bases, "regions": done, "seconds": time.perf_counter() - t0, "kind": kind}


def load_polish():
    """-> (kind, run(pileup, start_pos, end_pos) -> rows): the reference's own SummaryGenerator (oracle/_ref/
    libref_polish_encoder.so, built from pepper/modules/src/pileup_summary/summary_generator.cpp as it lies) or, where that
    build did not travel, the restatement."""
    ref = os.path.join(HERE, "_ref", "libref_polish_encoder.so")
    if os.path.exists(ref):
        fn, kind = ctypes.CDLL(ref).ref_polish_generate_summary, "reference"
    else:
        fn, kind = ctypes.CDLL(os.path.join(HERE, "libpileup_oracle.so")).oracle_polish_generate_summary, "port"
    fn.restype = ctypes.c_int64
    fn.argtypes = [ctypes.POINTER(Pileup), ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]

    def run(p, start_pos, end_pos, image=None, positions=None, cap=0):
        return int(fn(ctypes.byref(p), start_pos, end_pos, image, positions, cap))
    return kind, run


def time_polish_regions(regions, seconds):
    """Loop the CPU polish encoder over `regions` ((reference, flat, start, end) of pepper_amd.synthetic.encoder_region) until
    `seconds` have passed: one call per region with room for its rows, as AlignmentSummarizer.py:340-347 makes it."""
    kind, run = load_polish()
    structs = [region_structs(r)[0] for r in regions]
    img = np.zeros((1 << 20, 10), np.uint8)
    pos = np.zeros((1 << 20, 2), np.int64)
    bases = done = rows = 0
    t0 = time.perf_counter()
    while done < 1 or time.perf_counter() - t0 < seconds:
        k = done % len(structs)
        rows = run(structs[k], regions[k][2], regions[k][3], img.ctypes.data, pos.ctypes.data, len(img))
        bases += int(regions[k][1]["seq_offset"][-1])
        done += 1
    return {"bases": bases, "regions": done, "seconds": time.perf_counter() - t0, "kind": kind, "rows_last": rows}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--polish", action="store_true", help="the polish SummaryGenerator on --region-size + 2 x 100 positions")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--region-size", type=int, default=100_000)
    args = ap.parse_args()
    from pepper_amd import synthetic
    kw = dict(read_len=2000, depth=140) if args.polish else {}      # (bench.py polish_encoder_bench: ~60x over the 1.2 kb window)
    region = synthetic.encoder_region(synthetic.ESYN_SEED + args.seed, region=args.region_size, **kw)
    print(json.dumps((time_polish_regions if args.polish else time_regions)([region], args.seconds)))


if __name__ == "__main__":
    main()
