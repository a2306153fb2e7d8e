"""Split-precision (h2 = f16 hi/lo pairs, 3 MFMAs per product) GEMM against float64 numpy: the kernel
must be as accurate as an f32 GEMM (that is the premise that lets it replace the f32 MFMA path under
the 1e-4 parity budget), in both output forms, with ragged M/N and sub-normal lo parts."""
import ctypes

import numpy as np
import pytest

from pepper_amd import _lib

pytestmark = pytest.mark.gpu

F = ctypes.POINTER(ctypes.c_float)


def run(A, W, bias, M, N, act=0, frag_T=0, frag_nb=0, iters=1):
    lib = _lib.load()
    fn = lib.pa_debug_gemm_h2
    fn.restype = ctypes.c_int
    fn.argtypes = [F, F, F, F] + [ctypes.c_int] * 8 + [F]
    A = np.ascontiguousarray(A, np.float32)
    W = np.ascontiguousarray(W, np.float32)
    C = np.zeros((M, N), np.float32)
    ms = ctypes.c_float()
    bp = bias.ctypes.data_as(F) if bias is not None else None
    rc = fn(A.ctypes.data_as(F), W.ctypes.data_as(F), bp, C.ctypes.data_as(F), A.shape[0], M, N, A.shape[1], act,
            frag_T, frag_nb, iters, ctypes.byref(ms))
    assert rc == 0, _lib.last_error()
    return C, ms.value


def selu(x):
    return 1.0507009873554805 * np.where(x > 0, x, 1.6732632423543772 * np.expm1(x))


@pytest.mark.parametrize("M,N,K,act", [(300, 200, 64, 0), (512, 512, 512, 1), (257, 513, 1024, 0), (1, 3, 32, 0),
                                       (1024, 256, 4096, 1)])
def test_row_major(M, N, K, act):
    rng = np.random.default_rng(M + N + K)
    A = rng.uniform(-1, 1, size=(M, K)).astype(np.float32)
    A[:, ::7] *= 1e-3                                   # lo parts deep in the f16 sub-normal range
    A[0, :] = 0
    W = (rng.standard_normal((N, K)) * 0.05).astype(np.float32)
    W[:, 1::5] *= 40                                    # mixed magnitudes
    bias = rng.uniform(-0.5, 0.5, size=N).astype(np.float32)
    C, _ = run(A, W, bias, M, N, act=act)
    ref = A.astype(np.float64) @ W.astype(np.float64).T + bias
    f32 = (A @ W.T + bias).astype(np.float64)
    if act:
        ref, f32 = selu(ref), selu(f32)
    scale = np.abs(ref).max()
    err = np.abs(C - ref).max() / scale
    err32 = np.abs(f32 - ref).max() / scale
    assert err < max(2 * err32, 2e-7 * np.sqrt(K)), (err, err32)


def test_fragment_order_output():
    """A = [nb, T, K] sequences; logical rows ordered (32-batch block, step, batch in block); C in MFMA
    fragment order [M/32][N/32][4][64][4] with pad batches clamped to the last real one."""
    rng = np.random.default_rng(5)
    nb, T, K, N = 70, 5, 96, 320
    M = ((nb + 31) // 32 * 32) * T
    A = rng.uniform(-1, 1, size=(nb * T, K)).astype(np.float32)
    W = (rng.standard_normal((N, K)) * 0.1).astype(np.float32)
    bias = rng.uniform(-0.5, 0.5, size=N).astype(np.float32)
    C, _ = run(A, W, bias, M, N, frag_T=T, frag_nb=nb)
    ref_rows = A.astype(np.float64) @ W.astype(np.float64).T + bias          # [nb*T, N]
    C = C.reshape(M // 32, N // 32, 4, 64, 4)
    lane = np.arange(64)
    for rt in range(M // 32):
        block, t = rt // T, rt % T
        for qd in range(4):
            for j in range(4):
                r = 4 * qd + j
                rows_in_tile = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)      # [64]
                b = np.minimum(block * 32 + rows_in_tile, nb - 1)
                want = ref_rows[b * T + t][:, None, :].reshape(64, N)        # [64 lanes, N]
                for ct in range(N // 32):
                    got = C[rt, ct, qd, :, j]
                    exp = want[lane, ct * 32 + (lane & 31)]
                    assert np.abs(got - exp).max() < 2e-5, (rt, ct, qd, j)


def test_large_values_and_zero_rows():
    rng = np.random.default_rng(11)
    M, N, K = 256, 256, 256
    A = rng.integers(-128, 128, size=(M, K)).astype(np.float32)          # int8-valued activations: exact hi
    W = rng.uniform(-3, 3, size=(N, K)).astype(np.float32)
    C, _ = run(A, W, None, M, N)
    ref = A.astype(np.float64) @ W.astype(np.float64).T
    assert np.abs(C - ref).max() / np.abs(ref).max() < 1e-6
