"""Where one read's time goes inside the re-aligner kernels (s_memtime stamps kept in the job table):
python tools/realign_stages.py [n_reads]   -- prints percentiles of the four stages in microseconds."""
import ctypes
import sys

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from pepper_amd import _lib  # noqa: E402
from pepper_amd.polish import PEPPER  # noqa: E402


def main():
    n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
    rng = np.random.default_rng(5)
    bases = "ACGT"
    reference = "".join(bases[k] for k in rng.integers(0, 4, 1020))
    pos, seqs = [], []
    for _ in range(n_reads):
        full = rng.random() < 0.85
        a = 0 if full else int(rng.integers(0, 800))
        out = []
        for ch in reference[a:]:
            u = rng.random()
            if u < 0.04:
                continue
            if u < 0.08:
                ch = bases[int(rng.integers(4))]
            out.append(ch)
            while rng.random() < 0.03:
                out.append(bases[int(rng.integers(4))])
        pos.append(a)
        seqs.append("".join(out))
    blob = [q.encode() for q in seqs]
    off = np.zeros(len(seqs) + 1, np.int64)
    np.cumsum([len(b) for b in blob], out=off[1:])
    aligner = PEPPER.ReadAligner(0, len(reference), reference)
    for _ in range(3):
        out = aligner.align_arrays(pos, off, np.frombuffer(b"".join(blob), np.uint8))
    lib, h = PEPPER._realigner(0)
    ticks = np.zeros((n_reads, 4), np.int32)
    _lib.check(lib.pa_realigner_stage_ticks(h, ticks.ctypes.data))
    us = ticks / 100.0
    a, b, c = ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
    _lib.check(lib.pa_realigner_last_timing(h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
    print("reads", n_reads, "kernel ms: score", round(a.value, 3), "band", round(b.value, 3))
    for name, col in zip(("score passes", "band DP", "trace-back", "emission"), us.T):
        print("%-13s us: p50 %8.1f  p90 %8.1f  max %8.1f" % (name, np.percentile(col, 50), np.percentile(col, 90), col.max()))


if __name__ == "__main__":
    main()
