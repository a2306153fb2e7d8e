"""The polish summary encoder (encoder_polish.hip behind PEPPER.SummaryGenerator / generate_summaries) on synthetic 1.2 kb
regions of ~60x long reads: one region per call, and `--batch` regions per call.
    python tools/bench_polish_encoder.py [--reps 20] [--batch 256]"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--batch", type=int, default=256)
    args = ap.parse_args()
    import bam_utils as bu
    import pileup_utils as pu
    from test_gpu_encoder import R
    from pepper_amd.polish.PEPPER import SummaryGenerator, generate_summaries
    from pepper_amd.variant.PEPPER_VARIANT import flatten_reads
    rng = np.random.default_rng(11)
    ref = pu.random_reference(rng, 1201)
    reads = pu.simulate_reads(rng, ref, 7000, 90, read_len=(600, 1200), ins_rate=0.03, del_rate=0.03)
    reads = [r for r in reads if not any(op in (3, 6) for op, _ in r["cigar"])]
    clipped = bu.restated_get_reads(reads, 7000, 8200, False, 0)
    objs = [R(d) for d in clipped]
    bases = sum(len(d["seq"]) for d in clipped)
    gen = SummaryGenerator(ref, "contig_1", 7000, 8200)
    gen.generate_summary(objs, 7000, 8200)
    t0 = time.perf_counter()
    for _ in range(args.reps):
        gen = SummaryGenerator(ref, "contig_1", 7000, 8200)
        gen.generate_summary(objs, 7000, 8200)
    dt = (time.perf_counter() - t0) / args.reps
    # many regions per call: the same flat arrays behind every region (the content does not change the rate)
    flat = flatten_reads(objs)
    gens = [SummaryGenerator(ref, "contig_1", 7000, 8200) for _ in range(args.batch)]
    generate_summaries(gens, [flat] * args.batch, [(7000, 8200)] * args.batch)
    same = bool(np.array_equal(gens[-1].image, gen.image) and np.array_equal(gens[0].positions_array, gen.positions_array))
    t0 = time.perf_counter()
    for _ in range(max(2, args.reps // 4)):
        generate_summaries(gens, [flat] * args.batch, [(7000, 8200)] * args.batch)
    dtb = (time.perf_counter() - t0) / max(2, args.reps // 4)
    print(json.dumps({"metric": "polish summary encoder, one 1.2 kb region", "reads": len(clipped), "aligned_bases": bases,
                      "rows": int(gen.image.shape[0]), "ms_per_region": dt * 1e3, "bases_per_s": bases / dt,
                      "batch": {"regions": args.batch, "ms_per_call": dtb * 1e3, "regions_per_s": args.batch / dtb,
                                "bases_per_s": args.batch * bases / dtb, "equals_single": same}}))


if __name__ == "__main__":
    main()
