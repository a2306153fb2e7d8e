"""Polish image generation rate (BAM + draft -> image HDF5: BAM reader, GPU re-aligner, GPU encoder, HDF5 writer) for a
few worker-thread counts.   python tools/bench_polish_images.py make <dir>   writes a synthetic 60x / 120 kb data set
(CPU only; make_fast <dir> [bases]: any size, numpy);   python tools/bench_polish_images.py run <dir>   times make_images on
it (GPU; PROFILE=1: cProfile of the single-thread run)."""
import json
import os
import shutil
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))


def make(out):
    import bam_utils as bu
    import pileup_utils as pu
    os.makedirs(out, exist_ok=True)
    rng = np.random.default_rng(2026)
    draft = pu.random_reference(rng, 120000)
    reads = []
    bases = np.array(list("ACGT"))
    for _ in range(1400):
        length = int(rng.integers(2000, 9000))
        pos = int(rng.integers(0, len(draft) - 500))
        seq, cigar, rp = [], [], pos

        def push(op, n):
            if cigar and cigar[-1][0] == op:
                cigar[-1] = (op, cigar[-1][1] + n)
            else:
                cigar.append((op, n))
        while rp < min(len(draft), pos + length):
            u = rng.random()
            if u < 0.03 and cigar and cigar[-1][0] == 0:
                n = int(rng.integers(1, 5))
                seq.extend(bases[rng.integers(0, 4, n)])
                push(1, n)
            elif u < 0.07 and cigar and cigar[-1][0] == 0:
                n = int(rng.integers(1, 5))
                push(2, n)
                rp += n
            else:
                b = draft[rp]
                seq.append(b if rng.random() > 0.04 else bases[int(rng.integers(4))])
                push(0, 1)
                rp += 1
        while cigar and cigar[-1][0] != 0:            # end on an aligned base
            op, n = cigar.pop()
            if op == 1:
                del seq[-n:]
        reads.append(dict(pos=pos, reverse=bool(rng.random() < 0.5), mapq=60, seq="".join(seq),
                          qual=np.full(len(seq), 20, np.uint8), cigar=cigar))
    reads.sort(key=lambda r: r["pos"])
    for i, r in enumerate(reads):
        r["name"] = "read%d" % i
    bu.write_bam(os.path.join(out, "reads.bam"), [("ctg1", len(draft))], {0: reads})
    with open(os.path.join(out, "draft.fa"), "w") as fh:
        fh.write(">ctg1\n" + draft + "\n")
    print("wrote", len(reads), "reads,", sum(len(r["seq"]) for r in reads) // len(draft), "x coverage")


def make_fast(out, length=1200000, coverage=60, seed=2027):
    """The same kind of data set, any size, generated with numpy per read (make() walks every base in Python): reads of
    2-9 kb, ~3 % inserts and ~4 % deletions of 1-4 bases, 4 % substitutions."""
    import bam_utils as bu
    os.makedirs(out, exist_ok=True)
    rng = np.random.default_rng(seed)
    draft = rng.integers(0, 4, length).astype(np.uint8)
    letters = np.frombuffer(b"ACGT", np.uint8)
    n_reads = int(coverage * length / 5500)
    starts = np.sort(rng.integers(0, length - 500, n_reads))
    reads = []
    for i, pos in enumerate(starts.tolist()):
        n = min(length - pos, int(rng.integers(2000, 9000)))
        u = rng.random(n)
        dstart = np.flatnonzero(u < 0.04)
        deleted = np.zeros(n + 8, np.int32)
        if len(dstart):
            dl = rng.integers(1, 5, len(dstart))
            np.add.at(deleted, dstart, 1)
            np.add.at(deleted, dstart + dl, -1)
        deleted = np.cumsum(deleted)[:n] > 0
        deleted[:2] = False
        deleted[-2:] = False
        m = ~deleted
        ins = np.where((u >= 0.04) & (u < 0.07) & m & np.roll(m, -1), rng.integers(1, 5, n), 0)
        ins[-1] = 0
        # the op stream: per reference position M (or D), then an I run after it
        kinds = np.where(m, 0, 2)
        has_ins = ins > 0
        total = n + int(has_ins.sum())
        at = np.arange(n) + np.concatenate([[0], np.cumsum(has_ins)[:-1]])
        ops = np.full(total, 1, np.int64)
        lens = np.ones(total, np.int64)
        ops[at] = kinds
        lens[at[has_ins] + 1] = ins[has_ins]
        cut = np.flatnonzero(np.concatenate([[True], ops[1:] != ops[:-1]]))
        run_len = np.add.reduceat(lens, cut)
        cigar = list(zip(ops[cut].tolist(), run_len.tolist()))
        # the bases: per M position the draft base (4 % substituted), followed by its inserted bases (random)
        base = draft[pos:pos + n].copy()
        sub = rng.random(n) < 0.04
        base[sub] = rng.integers(0, 4, int(sub.sum()))
        counts = np.where(m, 1 + ins, 0)
        seq = np.repeat(base, counts)
        first = np.concatenate([[0], np.cumsum(counts)[:-1]])
        extra = np.ones(len(seq), bool)
        extra[first[m]] = False
        seq[extra] = rng.integers(0, 4, int(extra.sum()))
        reads.append(dict(pos=pos, reverse=bool(rng.random() < 0.5), mapq=60, seq=letters[seq].tobytes().decode(),
                          qual=np.full(len(seq), 20, np.uint8), cigar=cigar, name="read%d" % i))
    bu.write_bam(os.path.join(out, "reads.bam"), [("ctg1", length)], {0: reads})
    with open(os.path.join(out, "draft.fa"), "w") as fh:
        fh.write(">ctg1\n" + letters[draft].tobytes().decode() + "\n")
    print("wrote", len(reads), "reads,", sum(len(r["seq"]) for r in reads) // length, "x coverage of", length, "bases")


def run(data):
    from pepper_amd.polish.make_images import make_images
    out = []
    n_regions = -(-os.path.getsize(os.path.join(data, "draft.fa")) // 1000)
    for threads in (1, 4, 8, 16):
        tmp = os.path.join(data, "images_t%d" % threads)
        shutil.rmtree(tmp, ignore_errors=True)
        if threads == 1:
            make_images(os.path.join(data, "reads.bam"), os.path.join(data, "draft.fa"), "ctg1:0-9999", tmp + "_warm", 1)
        profile = threads == 1 and os.environ.get("PROFILE") == "1"
        if profile:
            import cProfile
            import pstats
            pr = cProfile.Profile()
            pr.enable()
        t0 = time.perf_counter()
        make_images(os.path.join(data, "reads.bam"), os.path.join(data, "draft.fa"), None, tmp, threads)
        dt = time.perf_counter() - t0
        if profile:
            pr.disable()
            pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
        out.append({"threads": threads, "seconds": round(dt, 3), "regions_per_s": round(n_regions / dt, 1)})
        shutil.rmtree(tmp, ignore_errors=True)
    print(json.dumps({"metric": "polish make_images, %d regions of ~1.2 kb at ~60x, one GPU" % n_regions, "runs": out}))


if __name__ == "__main__":
    if sys.argv[1] == "make_fast":
        make_fast(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 1200000)
    else:
        (make if sys.argv[1] == "make" else run)(sys.argv[2])
