// C[M,N] = act(A[M,K] * W[N,K]^T + bias[N]) with both operands in the "h2" split format, on
// v_mfma_f32_32x32x16_f16 (f32 accumulate).
//
// h2 format: an f32 value v is carried as two halves  hi = f16(v), lo = f16(v - hi)  (RNE); eight
// consecutive k of a row are stored as 16 bytes of hi followed by 16 bytes of lo, so a tensor has
// the same footprint and the same row strides as its f32 form (4 bytes per element) and one lane's
// MFMA operand for a 16-wide k step is 32 contiguous bytes.  The product is evaluated as
//     a*w ~= a_hi*w_hi + a_hi*w_lo + a_lo*w_hi          (3 MFMAs into ONE f32 accumulator)
// dropping only lo*lo (<= 2^-22 relative).  f16 sub-normals are honoured by the gfx950 matrix
// pipe (tools/ubench_f16.hip), so lo needs no scaling; |v| < 65504 is required and checked where
// the operands are produced.  Measured against the f32 MFMA path the end-to-end model error is
// unchanged (4e-6 on the reference golden vectors, budget 1e-4) while the matrix pipe runs the
// 3-MFMA product 5.3x faster than v_mfma_f32_32x32x2_f32.
//
// Call sites: the decoder input projection (A = encoder LSTM output, written in h2 by rnn.hip) and
// linear_1 (A = decoder LSTM output) of /root/reference/pepper_variant/modules/python/models/
// simple_model.py:54,58-60, and the polish decoder projection (pepper/.../simple_model.py:32).
//
// Tile: 256x256x32 per 256-thread workgroup, 4 waves as 2x2, each wave 128x128 = 4x4 MFMA tiles
// (256 accumulator registers, one wave per SIMD).  At the f16 rate the LDS port is the scarce
// resource: fragment reads scale with (1/wave_m + 1/wave_n), so the wave tile is as large as the
// register file allows (32 ds_read_b128 per 96 MFMAs; 128x64 tiles would need 48).  LDS holds two
// stages of 512 rows x 144 B (128 B = 32 k, padded so the 16-lane read groups hit 16 distinct
// 16-byte slots) = 147 KB of the CU's 160 KB.
#include "common.h"
#include "kernels.h"

#include <cstdlib>
#include <vector>

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));

constexpr int HM = 256, HN = 256, HK = 32;
constexpr int HROW = 36;                       // LDS row stride in dwords (32 data + 4 pad)
constexpr int HSTAGE = (HM + HN) * HROW;       // dwords per stage

PA_DEV f32x16 mfma_h(h8 a, h8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

template <int ACT, int EXP = 0>
__global__ __launch_bounds__(256) void gemm_h2_kernel(const uint32_t* __restrict__ A, int lda, uint32_t a_bytes,
                                                      const uint32_t* __restrict__ W, int ldw, uint32_t w_bytes,
                                                      const float* __restrict__ bias, float* __restrict__ C, int ldc,
                                                      int M, int N, int K, int tiles_n, int nwg, int a_rpb,
                                                      int64_t a_bstride, int frag_T, int frag_nb) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[2 * HSTAGE];

    // same XCD-aware tile order as gemm.hip: groups of GM row panels, row panel fastest
    const int tile = xcd_swizzle(blockIdx.x, nwg);
    constexpr int GM = 4;
    const int tiles_m = nwg / tiles_n;
    const int group = tile / (GM * tiles_n), within = tile - group * (GM * tiles_n);
    const int gm = min(GM, tiles_m - group * GM);
    const int m0 = (group * GM + within % gm) * HM;
    const int n0 = (within / gm) * HN;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;
    const int li = lane & 31, hf = lane >> 5;
    const int kq = tid & 7, r0 = tid >> 3;      // staging: 8 threads per 128-byte row segment, 32 rows per pass

    const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(A), 0, a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(W), 0, w_bytes, 0x00020000);

    // byte offsets of this thread's 8 A rows and 8 W rows (+ its 16-byte slot); fixed for the k loop
    uint32_t aoff[8], woff[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int m = m0 + r0 + 32 * i, n = n0 + r0 + 32 * i;
        const int mc = m < M ? m : M - 1, nc = n < N ? n : N - 1;   // clamp: tail rows are never stored
        size_t off;
        if (frag_T > 0) {
            int b = (mc / (32 * frag_T)) * 32 + (mc & 31);
            const int t = (mc >> 5) % frag_T;
            b = b < frag_nb ? b : frag_nb - 1;
            off = (size_t)b * (a_bstride > 0 ? (size_t)a_bstride : (size_t)frag_T * lda) + (size_t)t * lda;
        } else {
            off = a_rpb > 0 ? (size_t)(mc / a_rpb) * a_bstride + (size_t)(mc % a_rpb) * lda : (size_t)mc * lda;
        }
        aoff[i] = (uint32_t)(off * 4) + kq * 16;
        woff[i] = (uint32_t)((size_t)nc * ldw * 4) + kq * 16;
    }

    u32x4 ra[8], rb[8];
    auto gload_a = [&](int kt, int lo, int hi) {
#pragma unroll
        for (int i = lo; i < hi; ++i) ra[i] = __builtin_amdgcn_raw_buffer_load_b128(arsrc, aoff[i], kt * (HK * 4), 0);
    };
    auto gload_w = [&](int kt, int lo, int hi) {
#pragma unroll
        for (int i = lo; i < hi; ++i) rb[i] = __builtin_amdgcn_raw_buffer_load_b128(wrsrc, woff[i], kt * (HK * 4), 0);
    };
    auto lstore = [&](int buf) {
        uint32_t* base = lds + buf * HSTAGE + r0 * HROW + kq * 4;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            *reinterpret_cast<u32x4*>(base + (32 * i) * HROW) = ra[i];
            *reinterpret_cast<u32x4*>(base + (HM + 32 * i) * HROW) = rb[i];
        }
    };

    f32x16 acc[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;

    // fragments of one 16-wide k step: [m or n][hi, lo]
    struct Frag { h8 a[4][2], b[4][2]; };
    auto read_step = [&](int buf, int step, Frag& f) {
        const uint32_t* Ab = lds + buf * HSTAGE + (wm * 128 + li) * HROW + (2 * step + hf) * 8;
        const uint32_t* Bb = lds + buf * HSTAGE + (HM + wn * 128 + li) * HROW + (2 * step + hf) * 8;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            f.a[m][0] = *reinterpret_cast<const h8*>(Ab + m * 32 * HROW);
            f.a[m][1] = *reinterpret_cast<const h8*>(Ab + m * 32 * HROW + 4);
        }
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            f.b[n][0] = *reinterpret_cast<const h8*>(Bb + n * 32 * HROW);
            f.b[n][1] = *reinterpret_cast<const h8*>(Bb + n * 32 * HROW + 4);
        }
    };
    auto mma_step = [&](const Frag& f) {
        // the two small terms first, then hi*hi; consecutive MFMAs never share an accumulator
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[m][n] = mfma_h(f.a[m][1], f.b[n][0], acc[m][n]);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[m][n] = mfma_h(f.a[m][0], f.b[n][1], acc[m][n]);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[m][n] = mfma_h(f.a[m][0], f.b[n][0], acc[m][n]);
    };

    // split-K: gridDim.y slices of the k-tile range; slice y writes its partial sums to C + y * M * ldc
    // (bias / activation are applied by splitk_finish_kernel)
    const int nk_all = K / HK;
    const int kbeg = (int)((long long)nk_all * blockIdx.y / gridDim.y);
    const int nk = (int)((long long)nk_all * (blockIdx.y + 1) / gridDim.y) - kbeg;
    C += (size_t)blockIdx.y * M * ldc;
    Frag f0, f1;

    gload_a(kbeg, 0, 8);
    gload_w(kbeg, 0, 8);
    lstore(0);
    gload_a(kbeg + (nk > 1 ? 1 : 0), 0, 8);
    gload_w(kbeg + (nk > 1 ? 1 : 0), 0, 8);
    __syncthreads();
    read_step(0, 0, f0);
    if (EXP & 4) read_step(0, 1, f1);

    // One barrier per k-tile.  Stage kt:
    //   step 0: MFMAs on f0 | read f1 (step 1 of this tile) | write tile kt+1 (staging regs) to the
    //           other LDS buffer | request the A half of tile kt+2
    //   barrier (tile kt+1 visible; nobody reads the other buffer before it)
    //   step 1: MFMAs on f1 | read f0 of tile kt+1 | request the W half of tile kt+2
    // Every memory instruction rides in its own MFMA gap (sched_group_barrier); the loop body is
    // branch free (tail iterations re-request the last tile and write a buffer nobody reads).
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        const int nxt2 = kbeg + (kt + 2 < nk ? kt + 2 : nk - 1);
        if (!(EXP & 4)) read_step(buf, 1, f1);
        if (!(EXP & 2)) lstore(buf ^ 1);
        if (!(EXP & 1)) gload_a(nxt2, 0, 8);
        mma_step(f0);
        // 16 x [MFMA, DS read (f1)] [MFMA, DS write (tile kt+1)], then 8 x [MFMA, VMEM read (A of tile kt+2)]:
        // LDS writes alternate with reads instead of being issued as one block (ds_write_b128 is the slow LDS
        // port, ~79 B/clk; +5 % on the loop)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   // DS write
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // VMEM read
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        if (!(EXP & 4)) read_step(buf ^ 1, 0, f0);
        if (!(EXP & 1)) gload_w(nxt2, 0, 8);
        mma_step(f1);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read (f0 of tile kt+1)
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // VMEM read (W of tile kt+2)
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 24, 0);
        __builtin_amdgcn_sched_barrier(0);
    }

    // Everything the epilogue derives from the lane index is computed AFTER the k loop: left alone, hipcc hoists the four
    // 64-bit bias addresses and the row offsets above the loop, where all 512 registers are taken, and spills them to
    // scratch (9 VGPRs in round 1's build).  The empty asm makes the lane index opaque at this point.
    int lane_ep = lane;
    asm volatile("" : "+v"(lane_ep));
    const int li_ep = lane_ep & 31;
    if (frag_T > 0) {
        // C in MFMA fragment order (see gemm.hip / rnn.hip): 16-byte stores, bias folded in
        const int ct_n = N >> 5;
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int col = n0 + wn * 128 + n * 32 + li_ep;
            const float bv = (bias != nullptr && col < N) ? bias[col] : 0.0f;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int rt = (m0 + wm * 128 + m * 32) >> 5, ct = (n0 + wn * 128 + n * 32) >> 5;
                if ((rt << 5) < M && (ct << 5) < N) {
                    f32x4* dst = reinterpret_cast<f32x4*>(C + ((size_t)rt * ct_n + ct) * 1024) + lane_ep;
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
    for (int n = 0; n < 4; ++n) {
        const int col = n0 + wn * 128 + n * 32 + li_ep;
        const float bv = (bias != nullptr && col < N) ? bias[col] : 0.0f;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 128 + m * 32 + crow32(r, lane_ep);
                if (row < M && col < N) {
                    float v = acc[m][n][r] + bv;
                    if (ACT == 1) v = selu_f(v);
                    C[(size_t)row * ldc + col] = v;
                }
            }
        }
    }
}

// C = act(sum of split-K partials + bias); partials [S][M][N] dense, 4 columns per thread.
template <int ACT>
__global__ __launch_bounds__(256) void splitk_finish_kernel(const float* __restrict__ part, int S, const float* __restrict__ bias,
                                                            float* __restrict__ C, int ldc, int M, int N) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int n4 = N >> 2;
    if (idx >= (size_t)M * n4) return;
    const int row = (int)(idx / n4), c = (int)(idx % n4) * 4;
    f32x4 v = *reinterpret_cast<const f32x4*>(part + (size_t)row * N + c);
    for (int s = 1; s < S; ++s) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(part + ((size_t)s * M + row) * N + c);
        v += w;
    }
    if (bias != nullptr) v += *reinterpret_cast<const f32x4*>(bias + c);
    if (ACT == 1) { v.x = selu_f(v.x); v.y = selu_f(v.y); v.z = selu_f(v.z); v.w = selu_f(v.w); }
    *reinterpret_cast<f32x4*>(C + (size_t)row * ldc + c) = v;
}

// f32 rows -> h2 rows (same strides).  One thread per group of 8 k.
__global__ __launch_bounds__(256) void f32_to_h2_kernel(const float* __restrict__ src, uint32_t* __restrict__ dst,
                                                        int64_t rows, int K, int64_t ld) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int gpr = K >> 3;
    if (idx >= rows * gpr) return;
    const int64_t r = idx / gpr;
    const int g = (int)(idx - r * gpr);
    const float* s = src + r * ld + g * 8;
    h8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        hi[e] = (_Float16)s[e];
        lo[e] = (_Float16)(s[e] - (float)hi[e]);
    }
    h8* d = reinterpret_cast<h8*>(dst + r * ld + g * 8);
    d[0] = hi;
    d[1] = lo;
}

}  // namespace

namespace pa {

int g_h2_exp = 0;   // tuning experiments (tools/bench_gemm_h2.py): bit0 no global loads, bit1 no LDS writes, bit2 no LDS reads

void split_h2_host(const float* src, uint32_t* dst, int64_t rows, int K, int64_t ld_src, int64_t ld_dst) {
    for (int64_t r = 0; r < rows; ++r) {
        const float* s = src + r * ld_src;
        _Float16* d = reinterpret_cast<_Float16*>(dst + r * ld_dst);
        for (int g = 0; g < K / 8; ++g)
            for (int e = 0; e < 8; ++e) {
                const float v = s[g * 8 + e];
                const _Float16 hi = (_Float16)v;
                d[g * 16 + e] = hi;
                d[g * 16 + 8 + e] = (_Float16)(v - (float)hi);
            }
    }
}

hipError_t launch_f32_to_h2(const float* src, void* dst, int64_t rows, int K, int64_t ld, hipStream_t stream) {
    if ((K & 7) || (ld & 7)) return hipErrorInvalidValue;
    const int64_t n = rows * (K >> 3);
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(f32_to_h2_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, src,
                       reinterpret_cast<uint32_t*>(dst), rows, K, ld);
    return hipGetLastError();
}

hipError_t launch_gemm_h2(const void* A, int lda, size_t a_bytes, const void* W, int ldw, size_t w_bytes,
                          const float* bias, float* C, int ldc, int M, int N, int K, int act, int a_rpb,
                          int64_t a_bstride, int frag_T, int frag_nb, hipStream_t stream, float* splitk_ws,
                          int splits) {
    if (frag_T > 0 && ((M & 31) || (N & 31) || act != 0)) return hipErrorInvalidValue;
    if ((K % HK) || (lda & 7) || (ldw & 7) || (a_bstride & 7) || ((uintptr_t)A & 15) || ((uintptr_t)W & 15))
        return hipErrorInvalidValue;
    if (a_bytes >= (1ull << 32) || w_bytes >= (1ull << 32)) return hipErrorInvalidValue;   // 32-bit buffer offsets
    const int tiles_m = (M + HM - 1) / HM, tiles_n = (N + HN - 1) / HN;
    const int nwg = tiles_m * tiles_n;
    if (nwg == 0) return hipSuccess;
    const uint32_t* Au = reinterpret_cast<const uint32_t*>(A);
    const uint32_t* Wu = reinterpret_cast<const uint32_t*>(W);
    if (splits > 1 && splitk_ws != nullptr && frag_T == 0 && !(N & 3) && K / HK >= splits) {
        hipLaunchKernelGGL((gemm_h2_kernel<0>), dim3(nwg, splits), dim3(256), 0, stream, Au, lda, (uint32_t)a_bytes, Wu, ldw,
                           (uint32_t)w_bytes, (const float*)nullptr, splitk_ws, N, M, N, K, tiles_n, nwg, a_rpb, a_bstride,
                           0, 0);
        const size_t n4 = (size_t)M * (N >> 2);
        if (act == 1)
            hipLaunchKernelGGL((splitk_finish_kernel<1>), dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream,
                               splitk_ws, splits, bias, C, ldc, M, N);
        else
            hipLaunchKernelGGL((splitk_finish_kernel<0>), dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream,
                               splitk_ws, splits, bias, C, ldc, M, N);
        return hipGetLastError();
    }
    if (g_h2_exp > 0 && g_h2_exp < 8) {
#define PA_EXP(E_) case E_: hipLaunchKernelGGL((gemm_h2_kernel<0, E_>), dim3(nwg), dim3(256), 0, stream, Au, lda, (uint32_t)a_bytes, Wu, ldw, (uint32_t)w_bytes, bias, C, ldc, M, N, K, tiles_n, nwg, a_rpb, a_bstride, frag_T, frag_nb); break;
        switch (g_h2_exp) { PA_EXP(1) PA_EXP(2) PA_EXP(3) PA_EXP(4) PA_EXP(5) PA_EXP(6) PA_EXP(7) }
#undef PA_EXP
        return hipGetLastError();
    }
    if (act == 1)
        hipLaunchKernelGGL((gemm_h2_kernel<1>), dim3(nwg), dim3(256), 0, stream, Au, lda, (uint32_t)a_bytes, Wu, ldw,
                           (uint32_t)w_bytes, bias, C, ldc, M, N, K, tiles_n, nwg, a_rpb, a_bstride, frag_T, frag_nb);
    else
        hipLaunchKernelGGL((gemm_h2_kernel<0>), dim3(nwg), dim3(256), 0, stream, Au, lda, (uint32_t)a_bytes, Wu, ldw,
                           (uint32_t)w_bytes, bias, C, ldc, M, N, K, tiles_n, nwg, a_rpb, a_bstride, frag_T, frag_nb);
    return hipGetLastError();
}

}  // namespace pa

// Test / tuning hook (tests/test_gpu_gemm_h2.py, tools/): host f32 operands -> h2 (A on the device
// kernel, W on the host packer) -> gemm_h2 -> host C; the launch is repeated `iters` times between
// two HIP events.  frag_T > 0 exercises the recurrent-seed form (A = [frag_nb, frag_T, K]).
extern "C" void pa_debug_gemm_h2_experiment(int e) { pa::g_h2_exp = e; }

extern "C" int pa_debug_gemm_h2(const float* A, const float* W, const float* bias, float* C, int a_rows, int M, int N,
                                int K, int act, int frag_T, int frag_nb, int iters, float* ms_out) {
    void *dA = nullptr, *dA2 = nullptr, *dW = nullptr, *dB = nullptr, *dC = nullptr;
    const size_t a_bytes = (size_t)a_rows * K * 4, w_bytes = (size_t)N * K * 4;
    const size_t c_elems = frag_T > 0 ? (size_t)M * N : (size_t)M * N;
    std::vector<uint32_t> wh((size_t)N * K);
    pa::split_h2_host(W, wh.data(), N, K, K, K);
    hipError_t e = hipSuccess;
    auto ok = [&](hipError_t r) { if (e == hipSuccess) e = r; return r == hipSuccess; };
    ok(hipMalloc(&dA, a_bytes)); ok(hipMalloc(&dA2, a_bytes)); ok(hipMalloc(&dW, w_bytes));
    ok(hipMalloc(&dB, (size_t)N * 4)); ok(hipMalloc(&dC, c_elems * 4));
    if (e == hipSuccess) {
        ok(hipMemcpy(dA, A, a_bytes, hipMemcpyHostToDevice));
        ok(hipMemcpy(dW, wh.data(), w_bytes, hipMemcpyHostToDevice));
        if (bias) ok(hipMemcpy(dB, bias, (size_t)N * 4, hipMemcpyHostToDevice));
        ok(hipMemset(dC, 0, c_elems * 4));
        ok(pa::launch_f32_to_h2((const float*)dA, dA2, a_rows, K, K, nullptr));
        hipEvent_t e0, e1;
        ok(hipEventCreate(&e0)); ok(hipEventCreate(&e1));
        ok(pa::launch_gemm_h2(dA2, K, a_bytes, dW, K, w_bytes, bias ? (const float*)dB : nullptr, (float*)dC, N, M, N, K,
                              act, 0, 0, frag_T, frag_nb, nullptr));
        ok(hipDeviceSynchronize());
        ok(hipEventRecord(e0, nullptr));
        for (int i = 0; i < iters; ++i)
            ok(pa::launch_gemm_h2(dA2, K, a_bytes, dW, K, w_bytes, bias ? (const float*)dB : nullptr, (float*)dC, N, M, N,
                                  K, act, 0, 0, frag_T, frag_nb, nullptr));
        ok(hipEventRecord(e1, nullptr));
        ok(hipEventSynchronize(e1));
        float ms = 0.0f;
        ok(hipEventElapsedTime(&ms, e0, e1));
        if (ms_out) *ms_out = iters > 0 ? ms / iters : 0.0f;
        ok(hipMemcpy(C, dC, c_elems * 4, hipMemcpyDeviceToHost));
        (void)hipEventDestroy(e0);
        (void)hipEventDestroy(e1);
    }
    for (void* p : {(void*)dA, (void*)dA2, (void*)dW, (void*)dB, (void*)dC}) (void)hipFree(p);
    if (e != hipSuccess) return pa::set_error((int)e, std::string("pa_debug_gemm_h2: ") + hipGetErrorString(e));
    return 0;
}
