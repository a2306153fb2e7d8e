"""Throughput of the variant summary encoder on a batch of synthetic ONT-like regions (E-syn, pepper_amd.synthetic:
100 kb + 2 x 100 flank, ~60x, ~8 kb reads, 2 % indel events, 4 % substitutions, planted SNP / indel sites).
    python tools/bench_encoder.py [--regions 64] [--reps 5] [--check 2]
Prints one JSON line: aligned bases/s of pa_encoder_run_staged (inputs resident in HBM: kernels + host candidate
enumeration + window gather), per-kernel HIP-event times, the rate of the one-call form (H2D included), and the first
`--check` regions compared with the oracle restatement.  bench.py --model encoder is the line of record."""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

ONT = (1, 1, 0.10, 0.15, 0.15, 3, 0.10, 0.12, 2, False)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--regions", type=int, default=64)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--check", type=int, default=2)
    ap.add_argument("--region-size", type=int, default=100_000)
    ap.add_argument("--cache", default="", help="pickle of the generated regions: written if absent, read if present (profiling passes)")
    args = ap.parse_args()
    from pepper_amd import synthetic
    from pepper_amd.variant.PEPPER_VARIANT import RegionalSummaryGenerator, StagedBatch
    t0 = time.perf_counter()
    if args.cache and os.path.exists(args.cache):
        import pickle
        with open(args.cache, "rb") as fh:
            regions = pickle.load(fh)
    else:
        regions = synthetic.encoder_regions(args.regions, region=args.region_size)
        if args.cache:
            import pickle
            with open(args.cache, "wb") as fh:
                pickle.dump(regions, fh, protocol=4)
    t_gen = time.perf_counter() - t0
    gens = [RegionalSummaryGenerator("chr20", rs, re_, ref) for ref, _, rs, re_ in regions]
    flats = [flat for _, flat, _, _ in regions]
    cand = [(rs + 100, re_ - 100) for _, _, rs, re_ in regions]
    t0 = time.perf_counter()
    batch = StagedBatch(gens, flats, ONT, cand)
    t_stage = time.perf_counter() - t0
    batch.run()
    stats = batch.stats()
    times = []
    t0 = time.perf_counter()
    for _ in range(args.reps):
        counts = batch.run()
        times.append(batch.timing())
    dt = (time.perf_counter() - t0) / args.reps
    t0 = time.perf_counter()
    batch2 = StagedBatch(gens, flats, ONT, cand)
    batch2.run()
    dt_one_call = time.perf_counter() - t0
    out = batch2.results(want_int32=True)
    same = None
    if args.check:
        import pileup_utils as pu
        oracle = pu.load_restatement()
        same = True
        for k in range(min(args.check, args.regions)):
            ref, flat, rs, re_ = regions[k]
            pile = pu.FlatPileup.__new__(pu.FlatPileup)
            pile.region_start, pile.region_end, pile.reference = rs, re_, ref
            for key, v in flat.items():
                setattr(pile, key, v)
            want = pu.run_variant(oracle, pile, pu.make_params(*cand[k]))
            same = same and want["candidates"] == out[k]["candidates"] and np.array_equal(want["images"], out[k]["images_int32"])
    avg = {k: float(np.mean([t[k] for t in times])) for k in times[0]}
    alg_bytes = 2 * stats["bases"] + 104 * stats["rows"]
    print(json.dumps({"metric": "variant summary encoder, aligned bases/s (inputs resident in HBM)", "regions": args.regions,
                      "stats": stats, "candidates": int(counts.sum()), "run_ms": dt * 1e3, "bases_per_s": stats["bases"] / dt,
                      "timing_ms": avg, "tile_count_GBps": alg_bytes / (avg["tile_count_ms"] * 1e-3) / 1e9,
                      "tile_count_bases_per_s": stats["bases"] / (avg["tile_count_ms"] * 1e-3),
                      "one_call_incl_h2d_ms": dt_one_call * 1e3, "one_call_bases_per_s": stats["bases"] / dt_one_call,
                      "stage_ms": t_stage * 1e3, "generate_s": t_gen, "bit_exact_vs_oracle": same}))


if __name__ == "__main__":
    main()
