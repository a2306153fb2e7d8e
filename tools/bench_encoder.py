"""Throughput of the variant summary encoder on one synthetic ONT-like region (not the headline
bench): 100 kb (+2x100 flank), ~60x depth, ~8 kb reads, 2 % indels, 4 % substitutions.
    python tools/bench_encoder.py [--reps 5]
Prints one JSON line: regions/s, aligned bases/s end to end (host CIGAR pass + H2D + kernels + D2H),
and the time of the same region through the oracle's C++ restatement on one host core."""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))


def make_region(seed=7, region=100_000, flank=100, depth=60, read_len=8000):
    rng = np.random.default_rng(seed)
    L = region + 2 * flank + 1
    ref = rng.integers(0, 4, size=L).astype(np.uint8)
    alphabet = np.frombuffer(b"ACGT", np.uint8)
    n_reads = int(depth * L / read_len) + 8
    seqs, quals, cig_ops, cig_lens, pos, rev, mapq, soff, coff = [], [], [], [], [], [], [], [0], [0]
    for _ in range(n_reads):
        start = int(rng.integers(-read_len // 2, L - 50))
        s, e = max(0, start), min(L, start + int(rng.normal(read_len, read_len / 4)))
        if e - s < 100:
            continue
        span = e - s
        n_ev = rng.poisson(span * 0.02)
        cuts = np.unique(rng.integers(20, span - 20, size=n_ev)) if n_ev else np.zeros(0, np.int64)
        ops, lens, parts = [], [], []
        prev = 0
        for c in cuts:
            if c - prev < 2:
                continue
            seg = ref[s + prev:s + c].copy()
            ops.append(0); lens.append(c - prev); parts.append(seg)
            n = int(rng.integers(1, 6))
            if rng.random() < 0.5:
                ops.append(1); lens.append(n); parts.append(rng.integers(0, 4, size=n).astype(np.uint8))
                prev = c
            else:
                ops.append(2); lens.append(n)
                prev = min(span - 1, c + n)
        seg = ref[s + prev:e].copy()
        ops.append(0); lens.append(len(seg)); parts.append(seg)
        seq = np.concatenate(parts)
        sub = rng.random(len(seq)) < 0.04
        seq[sub] = (seq[sub] + rng.integers(1, 4, size=int(sub.sum()))) % 4
        seqs.append(alphabet[seq]); quals.append(rng.integers(3, 40, size=len(seq)).astype(np.uint8))
        cig_ops.extend(ops); cig_lens.extend(lens)
        pos.append(10_000 + s); rev.append(int(rng.random() < 0.5)); mapq.append(60)
        soff.append(soff[-1] + len(seq)); coff.append(coff[-1] + len(ops))
    flat = dict(read_pos=np.array(pos, np.int64), read_reverse=np.array(rev, np.uint8), read_mapq=np.array(mapq, np.int32),
                seq_offset=np.array(soff, np.int64), seq=np.concatenate(seqs + [np.zeros(1, np.uint8)]),
                qual=np.concatenate(quals + [np.zeros(1, np.uint8)]), cigar_offset=np.array(coff, np.int64),
                cigar_op=np.array(cig_ops + [0], np.int32), cigar_len=np.array(cig_lens + [0], np.int32), n_reads=len(pos))
    return alphabet[ref].tobytes().decode(), flat, 10_000, 10_000 + L - 1, region, flank


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    import pileup_utils as pu
    from pepper_amd.variant.PEPPER_VARIANT import RegionalSummaryGenerator
    ref, flat, rs, re_, region, flank = make_region()
    bases = int(flat["seq_offset"][-1])
    gen = RegionalSummaryGenerator("chr20", rs, re_, ref)
    call = lambda: gen.generate_summary_arrays(flat, 1, 1, 0.10, 0.15, 0.15, 3, 0.10, 0.12, 2, False, rs + flank,
                                               rs + flank + region, 32, 26, False)
    out = call()
    t0 = time.perf_counter()
    for _ in range(args.reps):
        out = call()
    dt = (time.perf_counter() - t0) / args.reps
    # oracle restatement on one host core, same region
    oracle = pu.load_restatement()
    pile = pu.FlatPileup.__new__(pu.FlatPileup)
    pile.region_start, pile.region_end, pile.reference = rs, re_, ref.encode()
    for k, v in flat.items():
        setattr(pile, k, v)
    params = pu.make_params(rs + flank, rs + flank + region)
    t0 = time.perf_counter()
    want = pu.run_variant(oracle, pile, params)
    dt_cpu = time.perf_counter() - t0
    same = want["candidates"] == out["candidates"] and np.array_equal(
        want["images"].astype(np.int64).astype(np.int8), out["images"])
    print(json.dumps({"metric": "variant summary encoder, one 100 kb region at ~60x", "reads": flat["n_reads"],
                      "aligned_bases": bases, "candidates": len(out["candidates"]), "gpu_path_ms": dt * 1e3,
                      "gpu_path_bases_per_s": bases / dt, "oracle_cpp_1core_ms": dt_cpu * 1e3,
                      "bit_exact_vs_oracle": bool(same)}))


if __name__ == "__main__":
    main()
