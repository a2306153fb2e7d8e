"""Polish image generation rate (BAM + draft -> image HDF5: BAM reader, GPU re-aligner, GPU encoder, HDF5 writer) for a
few worker-thread counts.   python tools/bench_polish_images.py make <dir>   writes a synthetic 60x / 120 kb data set
(CPU only);   python tools/bench_polish_images.py run <dir>   times make_images on it (GPU)."""
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


def run(data):
    from pepper_amd.polish.make_images import make_images
    out = []
    for threads in (1, 4, 8, 16):
        tmp = os.path.join(data, "images_t%d" % threads)
        shutil.rmtree(tmp, ignore_errors=True)
        if threads == 1:
            make_images(os.path.join(data, "reads.bam"), os.path.join(data, "draft.fa"), "ctg1:0-9999", tmp + "_warm", 1)
        t0 = time.perf_counter()
        make_images(os.path.join(data, "reads.bam"), os.path.join(data, "draft.fa"), None, tmp, threads)
        dt = time.perf_counter() - t0
        out.append({"threads": threads, "seconds": round(dt, 3), "regions_per_s": round(120 / dt, 1)})
        shutil.rmtree(tmp, ignore_errors=True)
    print(json.dumps({"metric": "polish make_images, 120 regions of ~1.2 kb at ~60x, one GPU", "runs": out}))


if __name__ == "__main__":
    (make if sys.argv[1] == "make" else run)(sys.argv[2])
