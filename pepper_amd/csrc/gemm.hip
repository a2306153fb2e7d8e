// C[M,N] = act(A[M,K] * W[N,K]^T + bias[N])  -- "NT" GEMM on v_mfma_f32_32x32x2_f32.
//
// Used for every input projection of the recurrent layers (torch.nn.LSTM/GRU's W_ih x + b,
// reference call sites pepper_variant/.../simple_model.py:51,54 and pepper/.../simple_model.py:30,32)
// and for the variant MLP head's Linear+SELU layers (simple_model.py:58-75).
//
// Tile: 128x128x32 per 256-thread workgroup, 4 waves as 2x2, each wave 64x64 = 2x2 MFMA
// tiles (64 accumulator VGPRs).  Both operands are K-contiguous (PyTorch weight layout is
// [out,in]), staged global -> registers -> LDS with one barrier per k-tile (loads for tile
// k+2 are issued, and tile k+1 is written to the other LDS buffer, before the MFMAs of tile k).
// LDS rows are padded to 36 floats: ds_read_b128 fragment reads are bank-conflict free
// (row stride 144 B -> 16 distinct 16-B slots per 16-lane group).
#include "common.h"
#include "kernels.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 32, LDT = BK + 4;

template <int AT> struct AType { typedef float type; };
template <> struct AType<pa::A_I8> { typedef int8_t type; };
template <> struct AType<pa::A_U8> { typedef uint8_t type; };

// Load 4 consecutive k of one row.  Rows beyond M/N are clamped to the last valid row by the
// caller (their results are never stored), so only the K tail needs predication; KFULL (K % 32
// == 0) removes even that and the loop carries no exec-mask branches.
template <int AT, bool KFULL>
PA_DEV void load_a4(const typename AType<AT>::type* __restrict__ row, int k, int K, float (&out)[4]) {
    if constexpr (AT == pa::A_F32) {
        // requires 16-byte aligned rows and K % 4 == 0 or zero-padded rows (checked on the host)
        if (KFULL || k < K) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(row + k);
            out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
        } else {
            out[0] = out[1] = out[2] = out[3] = 0.0f;
        }
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            out[e] = (KFULL || k + e < K) ? (float)row[k + e] : 0.0f;
    }
}

template <int AT, int ACT, bool KFULL>
__global__ __launch_bounds__(256) void gemm_nt_kernel(const typename AType<AT>::type* __restrict__ A, int lda,
                                                      const float* __restrict__ W, int ldw,
                                                      const float* __restrict__ bias,
                                                      float* __restrict__ C, int ldc,
                                                      int M, int N, int K, int tiles_n, int nwg,
                                                      int a_rpb, int64_t a_bstride, int frag_T, int frag_nb) {
    __shared__ __attribute__((aligned(16))) float lds[2 * 2 * BM * LDT];
    float* As = lds;                    // [2][BM][LDT]
    float* Bs = lds + 2 * BM * LDT;     // [2][BN][LDT]

    // Tile order inside an XCD's contiguous run: groups of GM row panels, row panel fastest, so the
    // ~64 tiles resident on an XCD at a time form a GM x 8 block sharing GM A panels and 8 W panels
    // in that XCD's L2 (n fastest re-streamed all of W for every 4 row panels: 4-9x A over-fetch).
    const int tile = xcd_swizzle(blockIdx.x, nwg);
    constexpr int GM = 8;
    const int tiles_m = nwg / tiles_n;
    const int group = tile / (GM * tiles_n), within = tile - group * (GM * tiles_n);
    const int gm = min(GM, tiles_m - group * GM);
    const int m0 = (group * GM + within % gm) * BM;
    const int n0 = (within / gm) * BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int li = lane & 31, hf = lane >> 5;

    // staging role: 8 threads cover one 128-byte row segment, 32 rows per pass, 4 passes
    const int kq = tid & 7, r0 = tid >> 3;

    float ra[4][4], rb[4][4];
    f32x16 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;

    const int nk = (K + BK - 1) / BK;

    // Row pointers are fixed per thread for the whole k loop.  A rows may be remapped: logical
    // row m = (batch m / a_rpb, step m % a_rpb) -> A + batch * a_bstride + step * lda, which lets
    // one window of a longer [B, S, F] tensor be projected without a gather copy.
    const typename AType<AT>::type* arow[4];
    const float* wrow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + r0 + 32 * i, n = n0 + r0 + 32 * i;
        const int mc = m < M ? m : M - 1, nc = n < N ? n : N - 1;   // clamp: tail rows are not stored
        size_t off;
        if (frag_T > 0) {
            // fragment-packed output: logical row m = (32-batch block, step t, batch-in-block) so that
            // every 32-row MFMA tile is one step of 32 consecutive sequences (what the recurrent
            // kernel's accumulator tile is); physical A row = batch * T + t, pad batches clamped
            int b = (mc / (32 * frag_T)) * 32 + (mc & 31);
            const int t = (mc >> 5) % frag_T;
            b = b < frag_nb ? b : frag_nb - 1;
            off = (size_t)b * (a_bstride > 0 ? (size_t)a_bstride : (size_t)frag_T * lda) + (size_t)t * lda;
        } else {
            off = a_rpb > 0 ? (size_t)(mc / a_rpb) * a_bstride + (size_t)(mc % a_rpb) * lda : (size_t)mc * lda;
        }
        arow[i] = A + off;
        wrow[i] = W + (size_t)nc * ldw;
    }

    auto gload = [&](int kt) {
        const int k = kt * BK + kq * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            load_a4<AT, KFULL>(arow[i], k, K, ra[i]);
            load_a4<pa::A_F32, KFULL>(wrow[i], k, K, rb[i]);
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x4 va = {ra[i][0], ra[i][1], ra[i][2], ra[i][3]};
            f32x4 vb = {rb[i][0], rb[i][1], rb[i][2], rb[i][3]};
            *reinterpret_cast<f32x4*>(&As[(buf * BM + r0 + 32 * i) * LDT + kq * 4]) = va;
            *reinterpret_cast<f32x4*>(&Bs[(buf * BN + r0 + 32 * i) * LDT + kq * 4]) = vb;
        }
    };

    // Software pipeline, one barrier per k-tile, with every LDS fragment read issued one
    // half-tile (32 MFMAs = 2048 pipe cycles) ahead of its use:
    //   iteration kt:  read H1(kt) | write tile kt+1 to the other buffer | request tile kt+2 |
    //                  MFMA H0(kt) | barrier | read H0(kt+1) | MFMA H1(kt)
    // H0/H1 = k-blocks {0,1} / {2,3} of the 32-wide tile.  sched_barrier(0) pins the order the
    // compiler would otherwise collapse into read-then-immediately-wait.
    f32x4 fa[2][2][2], fb[2][2][2];   // [half parity][kk in half][m or n]
    auto read_half = [&](int buf, int half, f32x4 (&a)[2][2], f32x4 (&b)[2][2]) {
        const float* Ab = As + (buf * BM + wm * 64 + li) * LDT + hf * 4 + half * 16;
        const float* Bb = Bs + (buf * BN + wn * 64 + li) * LDT + hf * 4 + half * 16;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
            for (int m = 0; m < 2; ++m) a[kk][m] = *reinterpret_cast<const f32x4*>(Ab + m * 32 * LDT + kk * 8);
#pragma unroll
            for (int n = 0; n < 2; ++n) b[kk][n] = *reinterpret_cast<const f32x4*>(Bb + n * 32 * LDT + kk * 8);
        }
    };
    auto mma_half = [&](const f32x4 (&a)[2][2], const f32x4 (&b)[2][2]) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n) acc[m][n] = mfma32(a[kk][m][s], b[kk][n][s], acc[m][n]);
    };

    gload(0);
    lstore(0);
    if (nk > 1) gload(1);
    __syncthreads();
    read_half(0, 0, fa[0], fb[0]);

    // The loop body is branch free (tail iterations redo harmless loads / LDS writes into buffers
    // nobody reads) so each half is one scheduling region, and sched_group_barrier spreads its memory
    // instructions ONE PER MFMA GAP: a VMEM issue costs the wave ~60 cycles, an LDS write ~13; issued
    // as a block they stall this wave's MFMA stream, and the co-resident workgroup -- running the
    // same code in lockstep -- does not fill the hole.
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        read_half(buf, 1, fa[1], fb[1]);
        lstore(buf ^ 1);
        gload(kt + 2 < nk ? kt + 2 : nk - 1);
        mma_half(fa[0], fb[0]);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read  (fragments of half 1)
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   // DS write (tile kt+1 -> other buffer)
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // VMEM read (tile kt+2)
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        read_half(buf ^ 1, 0, fa[0], fb[0]);
        mma_half(fa[1], fb[1]);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 24, 0);
        __builtin_amdgcn_sched_barrier(0);
    }

    if (frag_T > 0) {
        // C in MFMA fragment order: tile (row/32, col/32) = 4 chunks of [64 lanes][4 regs] floats, so
        // the producer stores and the consumer (rnn.hip accumulator seed) loads 16 bytes per lane
        const int ct_n = N >> 5;
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int col = n0 + wn * 64 + n * 32 + li;
            const float bv = (bias != nullptr && col < N) ? bias[col] : 0.0f;
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int rt = (m0 + wm * 64 + m * 32) >> 5, ct = (n0 + wn * 64 + n * 32) >> 5;
                if ((rt << 5) < M && (ct << 5) < N) {
                    f32x4* dst = reinterpret_cast<f32x4*>(C + ((size_t)rt * ct_n + ct) * 1024) + lane;
#pragma unroll
                    for (int qd = 0; qd < 4; ++qd) {
                        f32x4 v = {acc[m][n][4 * qd] + bv, acc[m][n][4 * qd + 1] + bv, acc[m][n][4 * qd + 2] + bv,
                                   acc[m][n][4 * qd + 3] + bv};
                        dst[qd * 64] = v;
                    }
                }
            }
        }
        return;
    }
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int col = n0 + wn * 64 + n * 32 + li;
        const float bv = (bias != nullptr && col < N) ? bias[col] : 0.0f;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + m * 32 + crow32(r, lane);
                if (row < M && col < N) {
                    float v = acc[m][n][r] + bv;
                    if (ACT == 1) v = selu_f(v);
                    C[(size_t)row * ldc + col] = v;
                }
            }
        }
    }
}

template <int AT>
hipError_t launch_typed(const typename AType<AT>::type* A, int lda, const float* W, int ldw, const float* bias, float* C,
                        int ldc, int M, int N, int K, int act, int a_rpb, int64_t a_bstride, int frag_T, int frag_nb,
                        hipStream_t stream) {
    const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
    const int nwg = tiles_m * tiles_n;
    if (nwg == 0) return hipSuccess;
#define PA_GEMM_LAUNCH(ACT_, KF_)                                                                   \
    hipLaunchKernelGGL((gemm_nt_kernel<AT, ACT_, KF_>), dim3(nwg), dim3(256), 0, stream, A, lda, W, ldw, \
                       bias, C, ldc, M, N, K, tiles_n, nwg, a_rpb, a_bstride, frag_T, frag_nb)
    const bool kfull = (K % BK) == 0;
    if (act == 1) {
        if (kfull) PA_GEMM_LAUNCH(1, true); else PA_GEMM_LAUNCH(1, false);
    } else {
        if (kfull) PA_GEMM_LAUNCH(0, true); else PA_GEMM_LAUNCH(0, false);
    }
#undef PA_GEMM_LAUNCH
    return hipGetLastError();
}

}  // namespace

namespace pa {

hipError_t launch_gemm_nt(int a_type, const void* A, int lda, const float* W, int ldw,
                          const float* bias, float* C, int ldc, int M, int N, int K, int act,
                          int a_rpb, int64_t a_bstride, int frag_T, int frag_nb, hipStream_t stream) {
    if (frag_T > 0 && ((M & 31) || (N & 31) || act != 0)) return hipErrorInvalidValue;
    // W rows are read 4 floats at a time: they must be 16-byte aligned and zero-padded to a
    // multiple of 4 columns (ldw >= round_up(K, 4)); the packer in api.hip guarantees this.
    const bool w_ok = !(ldw & 3) && !((uintptr_t)W & 15) && ldw >= ((K + 3) & ~3);
    switch (a_type) {
        case A_F32:
            if (!w_ok || (lda & 3) || (K & 3) || (a_bstride & 3) || ((uintptr_t)A & 15))
                return hipErrorInvalidValue;
            return launch_typed<A_F32>((const float*)A, lda, W, ldw, bias, C, ldc, M, N, K, act, a_rpb, a_bstride, frag_T, frag_nb, stream);
        case A_F32_SCALAR:  // unaligned / K % 4 != 0 float rows (e.g. [B,33,26] float images)
            if (!w_ok) return hipErrorInvalidValue;
            return launch_typed<A_F32_SCALAR>((const float*)A, lda, W, ldw, bias, C, ldc, M, N, K, act, a_rpb, a_bstride, frag_T, frag_nb, stream);
        case A_I8:
            if (!w_ok) return hipErrorInvalidValue;
            return launch_typed<A_I8>((const int8_t*)A, lda, W, ldw, bias, C, ldc, M, N, K, act, a_rpb, a_bstride, frag_T, frag_nb, stream);
        case A_U8:
            if (!w_ok) return hipErrorInvalidValue;
            return launch_typed<A_U8>((const uint8_t*)A, lda, W, ldw, bias, C, ldc, M, N, K, act, a_rpb, a_bstride, frag_T, frag_nb, stream);
        default:
            return hipErrorInvalidValue;
    }
}

}  // namespace pa
