// Classification heads: a C-class (C <= 8) Linear on top of the last hidden layer, with the
// softmax done by wavefront-level reductions (one 64-lane wave per row, butterfly __shfl_xor).
//
//  * variant:  output_layer_type (512 -> 3) + Softmax(dim=1)
//        /root/reference/pepper_variant/modules/python/models/simple_model.py:76-82
//  * polish :  dense1 (256 -> 5), then per window softmax(dim=2) zero-padded to the chunk and
//              added into the [B,1000,5] accumulator
//        /root/reference/pepper/modules/python/models/simple_model.py:34
//        /root/reference/pepper/modules/python/models/predict_distributed_cpu.py:62-81
//  * polish finalize: max over classes -> label, phred = -10 log10(1 - value / counts)
//        /root/reference/pepper/modules/python/models/predict_distributed_cpu.py:83-90
#include "common.h"
#include "kernels.h"

namespace {

constexpr int MAXC = 8;

PA_DEV float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// MODE 0: write probs (and optionally logits) [B,C].
// MODE 1: accumulate softmax into acc[(row / T) * S + off + row % T][C]   (polish overlap-add)
// MODE 2: write logits only [B,C]
template <int MODE>
__global__ __launch_bounds__(256) void dense_small_kernel(const float* __restrict__ X, int ldx,
                                                          const float* __restrict__ W,
                                                          const float* __restrict__ bias,
                                                          float* __restrict__ out0,
                                                          float* __restrict__ out1, int rows, int K,
                                                          int C, int T, int S, int off) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float part[MAXC];
#pragma unroll
    for (int c = 0; c < MAXC; ++c) part[c] = 0.0f;
    const float* x = X + (size_t)row * ldx;
    for (int k = lane * 4; k < K; k += 256) {
        const f32x4 xv = *reinterpret_cast<const f32x4*>(x + k);
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
            if (c < C) {
                const f32x4 wv = *reinterpret_cast<const f32x4*>(W + (size_t)c * K + k);
                part[c] += xv.x * wv.x + xv.y * wv.y + xv.z * wv.z + xv.w * wv.w;
            }
    }
    float logit[MAXC];
    float mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        logit[c] = c < C ? wave_sum(part[c]) + bias[c] : -INFINITY;
        mx = fmaxf(mx, logit[c]);
    }
    float den = 0.0f, e[MAXC];
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        e[c] = c < C ? expf(logit[c] - mx) : 0.0f;
        den += e[c];
    }
    // lanes 0..C-1 each own one class (select without dynamic register indexing)
    float my_logit = 0.0f, my_e = 0.0f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
        if (lane == c) { my_logit = logit[c]; my_e = e[c]; }
    if (lane < C) {
        const float p = my_e / den;
        if (MODE == 0) {
            out0[(size_t)row * C + lane] = p;
            if (out1 != nullptr) out1[(size_t)row * C + lane] = my_logit;
        } else if (MODE == 1) {
            const size_t dst = ((size_t)(row / T) * S + off + row % T) * C + lane;
            out0[dst] += p;
        } else {
            out0[(size_t)row * C + lane] = my_logit;
        }
    }
}

// Polish head, bandwidth-shaped: 16 lanes per row, 4 rows per wave; every load instruction moves
// four 256-byte row segments (coalesced), the class weights live in registers, and the reduction
// is a 4-step butterfly inside each 16-lane group (5 classes x 4 shuffles per 4 rows instead of
// 5 x 6 per row).  acc[(row / T) * S + off + row % T][c] += softmax(x W^T + b)[c].
template <int K>
__global__ __launch_bounds__(256) void polish_dense_acc_kernel(const float* __restrict__ X, int ldx,
                                                               const float* __restrict__ W,
                                                               const float* __restrict__ bias,
                                                               float* __restrict__ acc, int rows, int C, int T,
                                                               int S, int off, int rows_per_block) {
    constexpr int J = K / 64;
    const int lane = threadIdx.x & 63, l16 = lane & 15, rsel = lane >> 4, wave = threadIdx.x >> 6;
    f32x4 w[5][J];
#pragma unroll
    for (int c = 0; c < 5; ++c)
#pragma unroll
        for (int j = 0; j < J; ++j)
            w[c][j] = c < C ? *reinterpret_cast<const f32x4*>(W + (size_t)c * K + l16 * 4 + 64 * j) : f32x4{0, 0, 0, 0};
    float b[5];
#pragma unroll
    for (int c = 0; c < 5; ++c) b[c] = c < C ? bias[c] : 0.0f;
    const int row0 = blockIdx.x * rows_per_block;
    for (int base = row0 + wave * 4; base < row0 + rows_per_block; base += 16) {
        const int row = base + rsel;
        const bool ok = row < rows;
        const float* x = X + (size_t)(ok ? row : rows - 1) * ldx + l16 * 4;
        f32x4 xv[J];
#pragma unroll
        for (int j = 0; j < J; ++j) xv[j] = *reinterpret_cast<const f32x4*>(x + 64 * j);
        float logit[5];
#pragma unroll
        for (int c = 0; c < 5; ++c) {
            float p = 0.0f;
#pragma unroll
            for (int j = 0; j < J; ++j) p += xv[j].x * w[c][j].x + xv[j].y * w[c][j].y + xv[j].z * w[c][j].z + xv[j].w * w[c][j].w;
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) p += __shfl_xor(p, o, 64);
            logit[c] = c < C ? p + b[c] : -INFINITY;
        }
        float mx = logit[0];
#pragma unroll
        for (int c = 1; c < 5; ++c) mx = fmaxf(mx, logit[c]);
        float e[5], den = 0.0f;
#pragma unroll
        for (int c = 0; c < 5; ++c) {
            e[c] = expf(logit[c] - mx);
            den += e[c];
        }
        float mine = 0.0f;
#pragma unroll
        for (int c = 0; c < 5; ++c)
            if (l16 == c) mine = e[c];
        if (ok && l16 < C) acc[((size_t)(row / T) * S + off + row % T) * C + l16] += mine / den;
    }
}

// The same head reading X in the h2 split format written by rnn_h2.hip (x = hi + lo): 16 lanes per
// row, each lane owns K/128 groups of 8 columns (32 contiguous bytes: 16 B hi, 16 B lo).
template <int K>
__global__ __launch_bounds__(256) void polish_dense_acc_h2_kernel(const uint32_t* __restrict__ X, int ldx,
                                                                  const float* __restrict__ W,
                                                                  const float* __restrict__ bias,
                                                                  float* __restrict__ acc, int rows, int C, int T,
                                                                  int S, int off, int rows_per_block) {
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    constexpr int J = K / 128;
    const int lane = threadIdx.x & 63, l16 = lane & 15, rsel = lane >> 4, wave = threadIdx.x >> 6;
    float w[5][J][8];
#pragma unroll
    for (int c = 0; c < 5; ++c)
#pragma unroll
        for (int j = 0; j < J; ++j)
#pragma unroll
            for (int e = 0; e < 8; ++e) w[c][j][e] = c < C ? W[(size_t)c * K + (l16 + 16 * j) * 8 + e] : 0.0f;
    float b[5];
#pragma unroll
    for (int c = 0; c < 5; ++c) b[c] = c < C ? bias[c] : 0.0f;
    const int row0 = blockIdx.x * rows_per_block;
    for (int base = row0 + wave * 4; base < row0 + rows_per_block; base += 16) {
        const int row = base + rsel;
        const bool ok = row < rows;
        const uint32_t* x = X + (size_t)(ok ? row : rows - 1) * ldx;
        float xv[J][8];
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const h8 hi = *reinterpret_cast<const h8*>(x + (l16 + 16 * j) * 8);
            const h8 lo = *reinterpret_cast<const h8*>(x + (l16 + 16 * j) * 8 + 4);
#pragma unroll
            for (int e = 0; e < 8; ++e) xv[j][e] = (float)hi[e] + (float)lo[e];
        }
        float logit[5];
#pragma unroll
        for (int c = 0; c < 5; ++c) {
            float p = 0.0f;
#pragma unroll
            for (int j = 0; j < J; ++j)
#pragma unroll
                for (int e = 0; e < 8; ++e) p += xv[j][e] * w[c][j][e];
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) p += __shfl_xor(p, o, 64);
            logit[c] = c < C ? p + b[c] : -INFINITY;
        }
        float mx = logit[0];
#pragma unroll
        for (int c = 1; c < 5; ++c) mx = fmaxf(mx, logit[c]);
        float e[5], den = 0.0f;
#pragma unroll
        for (int c = 0; c < 5; ++c) {
            e[c] = expf(logit[c] - mx);
            den += e[c];
        }
        float mine = 0.0f;
#pragma unroll
        for (int c = 0; c < 5; ++c)
            if (l16 == c) mine = e[c];
        if (ok && l16 < C) acc[((size_t)(row / T) * S + off + row % T) * C + l16] += mine / den;
    }
}

// Second half of the fused polish head (first half: gru_rec_h2_kernel<.., DENSE> in rnn_h2.hip).  P holds, per
// direction, the partial logits of the last decoder layer laid out [dir][batch tile of 128][t][class 5][128 rows];
// logits = P[0] + P[1] + bias, then softmax over the classes and the overlap-add of
// predict_distributed_cpu.py:62-81: acc[(b * S + off + t) * C + c] += p.  One block = 16 chunks x all T steps of one
// batch tile: the [t][c][row] slab is read in 64-byte runs, transposed through LDS, and the accumulator -- T * C
// consecutive floats per chunk -- is updated with coalesced 4-byte read-modify-writes.
constexpr int CMB_ROWS = 16, CMB_TILE = 128, CMB_C = 5;
__global__ __launch_bounds__(256) void polish_combine_kernel(const float* __restrict__ P, const float* __restrict__ bias,
                                                             float* __restrict__ acc, int B, int T, int C, int S, int off,
                                                             int nbt) {
    extern __shared__ float cmb[];                      // [T][CMB_C][CMB_ROWS] logits, then [CMB_ROWS][T * C] probabilities
    float* lg = cmb;
    float* pr = cmb + (size_t)T * CMB_C * CMB_ROWS;
    const int bt = blockIdx.x / (CMB_TILE / CMB_ROWS), slab = blockIdx.x % (CMB_TILE / CMB_ROWS);
    const int r0 = slab * CMB_ROWS;
    const size_t dstride = (size_t)nbt * T * CMB_C * CMB_TILE;
    const float* p0 = P + (size_t)bt * T * CMB_C * CMB_TILE + r0;
    // phase 1: 16-byte loads, four lanes cover one (t, c) run of 16 rows
    const int n4 = T * CMB_C * (CMB_ROWS / 4);
    for (int i = threadIdx.x; i < n4; i += 256) {
        const int tc = i >> 2, q = i & 3;
        const f32x4 a = *reinterpret_cast<const f32x4*>(p0 + (size_t)tc * CMB_TILE + 4 * q);
        const f32x4 b = *reinterpret_cast<const f32x4*>(p0 + dstride + (size_t)tc * CMB_TILE + 4 * q);
        const float bv = bias[tc % CMB_C];
        f32x4 v = {a.x + b.x + bv, a.y + b.y + bv, a.z + b.z + bv, a.w + b.w + bv};
        *reinterpret_cast<f32x4*>(lg + tc * CMB_ROWS + 4 * q) = v;
    }
    __syncthreads();
    // phase 2: one (row, t) per thread and pass: softmax over the C classes
    for (int i = threadIdx.x; i < CMB_ROWS * T; i += 256) {
        const int row = i % CMB_ROWS, t = i / CMB_ROWS;
        float l[CMB_C], mx = -INFINITY;
#pragma unroll
        for (int c = 0; c < CMB_C; ++c) {
            l[c] = c < C ? lg[(t * CMB_C + c) * CMB_ROWS + row] : -INFINITY;
            mx = fmaxf(mx, l[c]);
        }
        float e[CMB_C], den = 0.0f;
#pragma unroll
        for (int c = 0; c < CMB_C; ++c) {
            e[c] = c < C ? expf(l[c] - mx) : 0.0f;
            den += e[c];
        }
#pragma unroll
        for (int c = 0; c < CMB_C; ++c)
            if (c < C) pr[(size_t)row * T * C + t * C + c] = e[c] / den;
    }
    __syncthreads();
    // phase 3: acc rows of one chunk are T * C consecutive floats
    const int per = T * C;
    for (int i = threadIdx.x; i < CMB_ROWS * per; i += 256) {
        const int row = i / per, e = i - row * per;
        const int b = bt * CMB_TILE + r0 + row;
        if (b < B) acc[((size_t)b * S + off) * C + e] += pr[(size_t)row * per + e];
    }
}

// h2 rows -> f32 rows in place (hi + lo), for callers that want the layer output itself
__global__ __launch_bounds__(256) void h2_to_f32_kernel(uint32_t* __restrict__ buf, int64_t rows, int K, int64_t ld) {
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int gpr = K >> 3;
    if (idx >= rows * gpr) return;
    const int64_t r = idx / gpr;
    const int g = (int)(idx - r * gpr);
    uint32_t* p = buf + r * ld + g * 8;
    const h8 hi = *reinterpret_cast<const h8*>(p);
    const h8 lo = *reinterpret_cast<const h8*>(p + 4);
    f32x4 a = {(float)hi[0] + (float)lo[0], (float)hi[1] + (float)lo[1], (float)hi[2] + (float)lo[2], (float)hi[3] + (float)lo[3]};
    f32x4 b = {(float)hi[4] + (float)lo[4], (float)hi[5] + (float)lo[5], (float)hi[6] + (float)lo[6], (float)hi[7] + (float)lo[7]};
    *reinterpret_cast<f32x4*>(p) = a;
    *reinterpret_cast<f32x4*>(p + 4) = b;
}

__global__ __launch_bounds__(256) void polish_finalize_kernel(const float* __restrict__ acc,
                                                              uint8_t* __restrict__ labels,
                                                              uint8_t* __restrict__ phred,
                                                              size_t total, int S, int C, int overlap) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const float* a = acc + idx * C;
    float best = a[0];
    int lab = 0;
    for (int c = 1; c < C; ++c)
        if (a[c] > best) { best = a[c]; lab = c; }   // first maximum wins, as torch.max on CPU
    const int p = (int)(idx % S);
    const float counts = (p < overlap || p >= S - overlap) ? 1.0f : 2.0f;
    float ph = -10.0f * log10f(1.0f - best / counts);
    if (isinf(ph)) ph = 100.0f;
    labels[idx] = (uint8_t)lab;
    // numpy float32 -> uint8 cast: truncation; NaN / negative are undefined there, map to 0
    phred[idx] = (ph >= 0.0f && ph < 256.0f) ? (uint8_t)ph : (uint8_t)0;
}

}  // namespace

namespace pa {

hipError_t launch_dense_small(int mode, const float* X, int ldx, const float* W, const float* bias,
                              float* out0, float* out1, int rows, int K, int C, int T, int S, int off,
                              hipStream_t stream) {
    if (rows <= 0) return hipSuccess;
    if (C > MAXC || (K & 3) || (ldx & 3)) return hipErrorInvalidValue;
    if (mode == 1 && C <= 5 && (K == 256 || K == 512)) {
        const int rpb = 256;   // rows per block: 16 rows in flight, 16 passes
        const dim3 g((rows + rpb - 1) / rpb), b(256);
        if (K == 256)
            hipLaunchKernelGGL((polish_dense_acc_kernel<256>), g, b, 0, stream, X, ldx, W, bias, out0, rows, C, T, S, off, rpb);
        else
            hipLaunchKernelGGL((polish_dense_acc_kernel<512>), g, b, 0, stream, X, ldx, W, bias, out0, rows, C, T, S, off, rpb);
        return hipGetLastError();
    }
    const dim3 grid((rows + 3) / 4), block(256);
    switch (mode) {
        case 0: hipLaunchKernelGGL((dense_small_kernel<0>), grid, block, 0, stream, X, ldx, W, bias, out0, out1, rows, K, C, T, S, off); break;
        case 1: hipLaunchKernelGGL((dense_small_kernel<1>), grid, block, 0, stream, X, ldx, W, bias, out0, out1, rows, K, C, T, S, off); break;
        case 2: hipLaunchKernelGGL((dense_small_kernel<2>), grid, block, 0, stream, X, ldx, W, bias, out0, out1, rows, K, C, T, S, off); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_polish_dense_acc_h2(const void* X, int ldx, const float* W, const float* bias, float* acc, int rows,
                                      int K, int C, int T, int S, int off, hipStream_t stream) {
    if (rows <= 0) return hipSuccess;
    if (C > 5 || K != 256 || (ldx & 7)) return hipErrorInvalidValue;
    const int rpb = 256;
    hipLaunchKernelGGL((polish_dense_acc_h2_kernel<256>), dim3((rows + rpb - 1) / rpb), dim3(256), 0, stream,
                       static_cast<const uint32_t*>(X), ldx, W, bias, acc, rows, C, T, S, off, rpb);
    return hipGetLastError();
}

hipError_t launch_polish_combine(const float* P, const float* bias, float* acc, int B, int T, int C, int S, int off,
                                 hipStream_t stream) {
    if (B <= 0) return hipSuccess;
    if (C > CMB_C || T <= 0) return hipErrorInvalidValue;
    const int nbt = (B + CMB_TILE - 1) / CMB_TILE;
    const size_t lds = ((size_t)T * CMB_C * CMB_ROWS + (size_t)CMB_ROWS * T * C) * sizeof(float);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    static bool attr_set = false;
    if (!attr_set && lds > 64 * 1024) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(polish_combine_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(polish_combine_kernel, dim3(nbt * (CMB_TILE / CMB_ROWS)), dim3(256), lds, stream, P, bias, acc, B, T, C, S,
                       off, nbt);
    return hipGetLastError();
}

hipError_t launch_h2_to_f32(void* buf, int64_t rows, int K, int64_t ld, hipStream_t stream) {
    if ((K & 7) || (ld & 7)) return hipErrorInvalidValue;
    const int64_t n = rows * (K >> 3);
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(h2_to_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream,
                       static_cast<uint32_t*>(buf), rows, K, ld);
    return hipGetLastError();
}

hipError_t launch_polish_finalize(const float* acc, uint8_t* labels, uint8_t* phred, int64_t B, int S,
                                  int C, int overlap, hipStream_t stream) {
    const size_t total = (size_t)B * S;
    if (total == 0) return hipSuccess;
    hipLaunchKernelGGL(polish_finalize_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       stream, acc, labels, phred, total, S, C, overlap);
    return hipGetLastError();
}

}  // namespace pa
