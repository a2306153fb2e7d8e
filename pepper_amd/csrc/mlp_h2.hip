// Tail of the variant classifier in one kernel: linear_2 .. linear_5 (512 -> 512, SELU each), the
// output layer (512 -> C) and the softmax, for 64 windows per workgroup with the activations resident
// in LDS between layers.
//   /root/reference/pepper_variant/modules/python/models/simple_model.py:62-82
// Replaces four 16384x512x512 GEMM launches (each too small to fill 256x256-tile workgroups and each
// round-tripping its 32 MB activation through HBM) plus the head launch.  Arithmetic is the split-f16
// scheme of gemm_h2.hip / rnn_h2.hip: activations are kept in LDS as h2 rows (16 B hi + 16 B lo per 8
// columns), weights are packed on the host as per-lane h2 fragments [n tile][k step][hi, lo][64][16 B]
// and streamed from L2, three v_mfma_f32_32x32x16_f16 per product, f32 accumulate, bias + SELU in f32.
// Workgroup = 8 waves; wave w owns output columns [64w, 64w+64) of every layer for both 32-row tiles
// (4 accumulators).  Between layers: barrier (everyone finished reading the activations), write the
// new activations in place (neighbouring lanes swap one half through DPP so each lane writes dwords),
// barrier.
#include "common.h"
#include "kernels.h"

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));

constexpr int MT = 64, D = 512, KS = D / 16, NTILES = D / 32;
constexpr int ROWB = D * 4 + 16, ROWD = ROWB / 4;      // 2064-byte rows: odd number of 16-byte slots

PA_DEV f32x16 mfma_h(h8 a, h8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

template <int NL>
__global__ __launch_bounds__(512, 1) void mlp_tail_h2_kernel(const float* __restrict__ X, int ldx,
                                                             const uint32_t* __restrict__ Wp,      // [NL][16][32][2][64][4] dwords
                                                             const float* __restrict__ bias,       // [NL][512] x scale | [NL] 1 / scale | [NL][512] as given
                                                             const float* __restrict__ Wout,       // [C][512]
                                                             const float* __restrict__ bout, int C,
                                                             float* __restrict__ probs, float* __restrict__ logits,
                                                             int n, const float* const* __restrict__ W32,  // [NL] f32 [512][512]: exact re-run
                                                             int* __restrict__ overflow_rows) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];      // [MT][ROWD]
    static_assert((ROWB / 16) % 2 == 1, "row stride must be an odd number of 16-byte slots");
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, hf = lane >> 5;
    const int b0 = blockIdx.x * MT;

    // An activation of 65504 or more (or a NaN) does not fit the hi half of the h2 format: every conversion below checks its
    // value, and a workgroup that saw one re-runs its 64 rows with plain f32 arithmetic at the end (simple_model.py:60-78 is
    // f32 throughout): never a silent inf / NaN.
    bool bad = false;
    // ---- stage the f32 input rows as h2: thread = (row, eight groups of 8 columns) ----
    {
        const int row = tid >> 3, gsel = tid & 7;              // 64 rows x 8 threads
        int grow = b0 + row;
        grow = grow < n ? grow : n - 1;
        const float* src = X + (size_t)grow * ldx;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int g = gsel + 8 * j;                        // 64 groups per row
            const f32x4 a = *reinterpret_cast<const f32x4*>(src + g * 8);
            const f32x4 b = *reinterpret_cast<const f32x4*>(src + g * 8 + 4);
            const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
            h8 hi, lo;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                bad |= !(fabsf(v[e]) < 65504.0f);
                hi[e] = (_Float16)v[e];
                lo[e] = (_Float16)(v[e] - (float)hi[e]);
            }
            *reinterpret_cast<h8*>(lds + row * ROWD + g * 8) = hi;
            *reinterpret_cast<h8*>(lds + row * ROWD + g * 8 + 4) = lo;
        }
    }
    __syncthreads();

    const uint32_t* arow = lds + li * ROWD + hf * 8;
    const bool odd = li & 1;
    struct Frag { h8 b[2][2], a[2][2]; };       // [n tile][hi, lo], [row tile][hi, lo]

#pragma unroll 1
    for (int layer = 0; layer < NL; ++layer) {
        // fragment (nt, s, hl): dword offset ((nt * KS + s) * 2 + hl) * 256 + lane * 4
        const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<uint32_t*>(Wp + ((size_t)layer * NTILES + 2 * w) * KS * 512), 0, 0x7fffffff, 0x00020000);
        const unsigned woff = lane * 16u;
        auto load_step = [&](int s, Frag& fr) {
#pragma unroll
            for (int nn = 0; nn < 2; ++nn)
#pragma unroll
                for (int hl = 0; hl < 2; ++hl)
                    fr.b[nn][hl] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(
                                                              wrs, woff, (unsigned)((nn * KS + s) * 2 + hl) * 1024u, 0));
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                fr.a[m][0] = *reinterpret_cast<const h8*>(arow + m * 32 * ROWD + s * 16);
                fr.a[m][1] = *reinterpret_cast<const h8*>(arow + m * 32 * ROWD + s * 16 + 4);
            }
        };
        f32x16 acc[2][2];
#pragma unroll
        for (int nn = 0; nn < 2; ++nn) {
            const float bv = bias[layer * D + 64 * w + 32 * nn + li];
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][nn][r] = bv;
        }
        Frag ring[2];
        load_step(0, ring[0]);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int p = s & 1;
            if (s + 1 < KS) load_step(s + 1, ring[p ^ 1]);
#pragma unroll
            for (int term = 0; term < 3; ++term)
#pragma unroll
                for (int nn = 0; nn < 2; ++nn)
#pragma unroll
                    for (int m = 0; m < 2; ++m)
                        acc[m][nn] = mfma_h(ring[p].a[m][term == 0 ? 1 : 0], ring[p].b[nn][term == 1 ? 1 : 0], acc[m][nn]);
            if (s + 1 < KS) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // B fragment
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // A fragment
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        lds_barrier();                      // every wave has finished reading this layer's input
        const float unscale = bias[NL * D + layer];      // the layer's weights and bias were packed times a power of two
#pragma unroll
        for (int nn = 0; nn < 2; ++nn) {
            const int col = 64 * w + 32 * nn + li;
            uint32_t* dst = lds + 4 * hf * ROWD + (col >> 3) * 8 + (odd ? 4 : 0) + ((col & 7) >> 1);
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = selu_f(acc[m][nn][r] * unscale);
                    bad |= !(fabsf(v) < 65504.0f);
                    dst[(32 * m + (r & 3) + 8 * (r >> 2)) * ROWD] = h2_word_of(v, h2_select(odd));
                }
        }
        lds_barrier();                      // next layer's input visible
    }

    // ---- out-of-range activations somewhere in these 64 rows: the layers again, in f32, from the f32 input ----
    const bool rerun = __syncthreads_or(bad ? 1 : 0) != 0;
    float* frow = reinterpret_cast<float*>(lds);       // the same 2064-byte rows, now [64][516] f32
    if (rerun) {
        if (tid == 0 && overflow_rows != nullptr) atomicAdd(overflow_rows, n - b0 < MT ? n - b0 : MT);
        for (int i = tid; i < MT * D; i += 512) {
            const int row = i / D, col = i - row * D;
            const int grow = b0 + row < n ? b0 + row : n - 1;
            frow[row * ROWD + col] = X[(size_t)grow * ldx + col];
        }
        __syncthreads();
#pragma unroll 1
        for (int layer = 0; layer < NL; ++layer) {
            const float* wr = W32[layer] + (size_t)tid * D;      // thread = output unit, all 64 rows
            float y[MT];
            const float bv = bias[NL * D + NL + layer * D + tid];
#pragma unroll
            for (int r = 0; r < MT; ++r) y[r] = bv;
#pragma unroll 1
            for (int k = 0; k < D; k += 4) {
                const f32x4 wv = *reinterpret_cast<const f32x4*>(wr + k);
#pragma unroll
                for (int r = 0; r < MT; ++r) {
                    const f32x4 xv = *reinterpret_cast<const f32x4*>(frow + r * ROWD + k);     // one address per wave: broadcast
                    y[r] = fmaf(xv.x, wv.x, y[r]);
                    y[r] = fmaf(xv.y, wv.y, y[r]);
                    y[r] = fmaf(xv.z, wv.z, y[r]);
                    y[r] = fmaf(xv.w, wv.w, y[r]);
                }
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < MT; ++r) frow[r * ROWD + tid] = selu_f(y[r]);
            __syncthreads();
        }
    }

    // ---- output layer + softmax: wave w takes rows 8w .. 8w+7, lane = one group of 8 columns ----
    float wo[8][8];
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int e = 0; e < 8; ++e) wo[c][e] = c < C ? Wout[(size_t)c * D + lane * 8 + e] : 0.0f;
    for (int rr = 0; rr < 8; ++rr) {
        const int row = 8 * w + rr;
        const h8 hi = *reinterpret_cast<const h8*>(lds + row * ROWD + lane * 8);
        const h8 lo = *reinterpret_cast<const h8*>(lds + row * ROWD + lane * 8 + 4);
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) x[e] = rerun ? frow[row * ROWD + lane * 8 + e] : (float)hi[e] + (float)lo[e];
        float logit[8], mx = -INFINITY;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float p = 0.0f;
#pragma unroll
            for (int e = 0; e < 8; ++e) p += x[e] * wo[c][e];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) p += __shfl_xor(p, o, 64);
            logit[c] = c < C ? p + bout[c < C ? c : 0] : -INFINITY;
            mx = fmaxf(mx, logit[c]);
        }
        float e[8], den = 0.0f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            e[c] = c < C ? expf(logit[c] - mx) : 0.0f;
            den += e[c];
        }
        float my_logit = 0.0f, my_e = 0.0f;
#pragma unroll
        for (int c = 0; c < 8; ++c)
            if (lane == c) { my_logit = logit[c]; my_e = e[c]; }
        const int grow = b0 + row;
        if (grow < n && lane < C) {
            probs[(size_t)grow * C + lane] = my_e / den;
            if (logits != nullptr) logits[(size_t)grow * C + lane] = my_logit;
        }
    }
}

}  // namespace

namespace pa {

// NL matrices W [512][512] (row = output unit) -> per-lane h2 fragments [layer][n tile 16][k step 32][hi, lo][64][8 halves].
// Every layer is packed times a power of two that puts its largest weight near 2^14: the lo half of an h2 value is an f16 too,
// so a weight below 6e-5 would otherwise lose its low bits to the f16 sub-normal spacing (6e-8 absolute) -- harmless beside
// activations of order 1, not beside activations of order 1e4.  scale[l] receives the factor; the kernel multiplies the
// accumulator (bias packed times the same factor) by its reciprocal, exactly.
void pack_mlp_weights_h2(const float* const* W, int NL, uint32_t* out, float* scale) {
    _Float16* o = reinterpret_cast<_Float16*>(out);
    for (int l = 0; l < NL; ++l) {
        float mx = 0.0f;
        for (size_t i = 0; i < (size_t)D * D; ++i) mx = fmaxf(mx, fabsf(W[l][i]));
        int e = 0;
        if (mx > 0.0f) {
            (void)frexpf(mx, &e);                        // mx = f * 2^e, f in [0.5, 1)
            e = 14 - e;                                  // largest weight -> [2^13, 2^14)
            e = e > 60 ? 60 : (e < -60 ? -60 : e);
        }
        const float sc = ldexpf(1.0f, e);
        scale[l] = sc;
        for (int nt = 0; nt < NTILES; ++nt)
            for (int s = 0; s < KS; ++s)
                for (int ln = 0; ln < 64; ++ln)
                    for (int e8 = 0; e8 < 8; ++e8) {
                        const float v = sc * W[l][(size_t)(nt * 32 + (ln & 31)) * D + 16 * s + 8 * (ln >> 5) + e8];
                        const _Float16 hi = (_Float16)v;
                        const size_t base = ((((size_t)l * NTILES + nt) * KS + s) * 2) * 512 + (size_t)ln * 8 + e8;
                        o[base] = hi;
                        o[base + 512] = (_Float16)(v - (float)hi);
                    }
    }
}

size_t mlp_weights_h2_words(int NL) { return (size_t)NL * NTILES * KS * 2 * 256; }

hipError_t launch_mlp_tail_h2(const float* X, int ldx, const void* Wp, const float* bias, int NL, const float* Wout,
                              const float* bout, int C, float* probs, float* logits, int n, hipStream_t stream,
                              const float* const* W32, int* overflow_rows) {
    if (n <= 0) return hipSuccess;
    if (NL != 4 || C > 8 || C <= 0 || (ldx & 3) || W32 == nullptr) return hipErrorInvalidValue;
    const size_t lds = (size_t)MT * ROWB;
    hipLaunchKernelGGL((mlp_tail_h2_kernel<4>), dim3((n + MT - 1) / MT), dim3(512), lds, stream, X, ldx,
                       static_cast<const uint32_t*>(Wp), bias, Wout, bout, C, probs, logits, n, W32, overflow_rows);
    return hipGetLastError();
}

}  // namespace pa
