"""Does operand data (switching activity -> power -> clock) set the h2 GEMM rate?  Same kernel, same
shape (one 256x256 tile per CU, K = 16384), different operand values."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from test_gpu_gemm_h2 import run
from pepper_amd import _lib

rng = np.random.default_rng(0)
M, N, K = 4096, 4096, 16384
cases = {
    "zeros": (np.zeros((M, K), np.float32), np.zeros((N, K), np.float32)),
    "ones (lo = 0)": (np.ones((M, K), np.float32), np.ones((N, K), np.float32)),
    "int8-valued A (lo = 0), random W": (rng.integers(-128, 128, size=(M, K)).astype(np.float32), (rng.standard_normal((N, K)) * 0.05).astype(np.float32)),
    "random": (rng.uniform(-1, 1, size=(M, K)).astype(np.float32), (rng.standard_normal((N, K)) * 0.05).astype(np.float32)),
}
for e in (0, 7):
    _lib.load().pa_debug_gemm_h2_experiment(e)
    for name, (A, W) in cases.items():
        _, ms = run(A, W, None, M, N, iters=8)
        print("exp %d  %-34s %8.3f ms  %7.1f TFLOP/s" % (e, name, ms, 2.0 * M * N * K / ms / 1e9))
_lib.load().pa_debug_gemm_h2_experiment(0)
