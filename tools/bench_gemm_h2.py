"""Timing of the split-precision GEMM at the variant model's two big shapes (not a test)."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_gemm_h2 import run

from pepper_amd import _lib
rng = np.random.default_rng(0)
M, N, K = 4096, 4096, 16384      # one tile per CU, 512 k-tiles: main-loop efficiency only
A = rng.uniform(-1, 1, size=(M, K)).astype(np.float32)
W = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
for e in range(8):
    _lib.load().pa_debug_gemm_h2_experiment(e)
    _, ms = run(A, W, None, M, N, iters=5)
    print("experiment %d (skip: %s%s%s)  %8.3f ms  %7.1f TFLOP/s" % (e, "vmem " if e & 1 else "", "ds_write " if e & 2 else "", "ds_read" if e & 4 else "", ms, 2.0 * M * N * K / ms / 1e9))
_lib.load().pa_debug_gemm_h2_experiment(0)
for name, M, N, K, T, nb in (("decoder in-proj /4", 33 * 4096, 2048, 512, 33, 4096), ("linear_1", 16384, 512, 16896, 0, 0),
                             ("square", 8192, 8192, 1024, 0, 0)):
    A = rng.uniform(-1, 1, size=(M, K)).astype(np.float32)
    W = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
    _, ms = run(A, W, None, M, N, frag_T=T, frag_nb=nb, iters=5)
    print("%-20s M %7d N %5d K %6d  %8.3f ms  %7.1f TFLOP/s (f32-equivalent)" % (name, M, N, K, ms, 2.0 * M * N * K / ms / 1e9))
