// LSTM step loop on the f16 matrix pipe with split (h2) operands -- the fast form of
// rnn.hip:lstm_rec_kernel for the variant model's four H = 256 directions
// (/root/reference/pepper_variant/modules/python/models/simple_model.py:51,54).
//
// Same decomposition and lockstep phase structure as the f32 kernel (workgroup = 64 batch rows x one
// direction, wave u owns hidden units [32u, 32u+32) of all four gates and both row tiles, MFMA phase
// | barrier | gate phase | barrier), but the contraction h_{t-1} W_hh^T (+ x_t W_ih^T when the input
// projection is fused) runs as three v_mfma_f32_32x32x16_f16 per 16-wide k step on operands held as
// f16 hi/lo pairs (gemm_h2.hip describes the format and why it is as accurate as f32):
//   * h lives in LDS in h2 form (32 bytes per 8 units: 16 B hi, 16 B lo); the gate phase rounds
//     the new h to (hi, lo), neighbouring lanes swap one half through DPP so each lane still writes
//     one dword per element;
//   * y is that LDS image copied out 16 bytes per lane during the MFMA phase, i.e. the layer output
//     is ALREADY in the h2 format the next projection GEMM / linear_1 consumes -- no conversion pass;
//   * W_hh (and W_ih for the fused first layer, whose int8 inputs are exact in the hi half) is packed
//     on the host in per-lane fragment order [gate tile][k step][hi, lo][64 lanes][16 B], streamed
//     from L2 through a 2-deep register ring.
// Per step a wave issues 24 MFMAs x 32 cycles per k step instead of 128 x 64: the MFMA phase shrinks
// ~5x and the weight stream (1 MiB per step per workgroup, same bytes as f32) becomes the thing to
// watch (~40 B/clk/CU of the ~64 B/clk/CU L2 path).
#include <cstdlib>

#include "common.h"
#include "kernels.h"

// measured r02 (16384 windows, device-resident pass): 2.566 -> 2.596 M windows/s with nt on the once-through streams
#ifndef PA_NT_DEFAULT
#define PA_NT_DEFAULT true
#endif

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));

constexpr int MT = 64;

PA_DEV f32x16 mfma_h(h8 a, h8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

PA_DEV void decode_block(int bid, int& dir, int& btile) {
    const int xcd = bid & 7, q = bid >> 3;   // XCDs 0-3 forward, 4-7 reverse (rnn.hip)
    dir = xcd >> 2;
    btile = q * 4 + (xcd & 3);
}

// value of lane ^ 1 (quad_perm [1,0,3,2])


// PRE: the packed weights, the bias and Xp arrive pre-multiplied per gate row by -log2(e) (i, f, o) or
// +2 log2(e) (g), so the accumulators ARE the exp2 arguments of sigmoid / tanh (api.hip build_rec_layer).
// XG (decoder layers): the layer input x_t is the previous layer's output, an h2 tensor Xh [B*T, KX]; its
// projection is contracted inside the step loop (K = H + KX).  No projection GEMM, no Xp round trip (4.4 GB
// written + read per 16384 windows).  The x slab of a step (64 rows x 2 KB = 128 KB) does not fit in LDS next to
// h and c, so it streams through a two-slot LDS ring, two k steps (64 rows x 128 B) per slot: every thread
// loads one 16-byte chunk of iteration j+2 into a staging register while iteration j is contracted, writes it
// to the free slot, and one LDS barrier per iteration publishes it; all waves then read their x fragments from
// LDS like they read h.  (First form of this kernel: every wave fetched its x fragments straight from global
// memory -- 8x redundant through L1, +0.7 ms per 16384 windows, and 4 MB of x per step per XCD pushed the
// weights out of L2.)
// SAUX: cache policy of the streams that pass through once -- the x slab loads of the XG form and the y stores (2 = nt:
// the lines are not kept in the XCD's L2, which has to hold this direction's 3 MB of weight fragments that every
// workgroup re-reads every step; 0 = default policy).
// BC (fused int8 first layer with F < KX): the bias lives in column H + F of the packed weights and x carries a constant
// 1.0 there, so the first MFMA of a step starts every accumulator from the inline constant 0: no bias loads and no 128
// register moves per step in the gate phase, which is bound by VALU issue (two waves per SIMD, ~1300 instructions each).
// MTILES (row tiles of 32 per workgroup; 2 = the 64-row workgroup everything above describes, 1 = the small-call schedule):
// a step is one CU's affair -- 8 waves x KS k steps x 12 MTILES MFMAs of 32 cycles on 4 SIMDs, then the gate phase of
// 16 MTILES elements per lane -- whatever the number of workgroups, so a call of a few hundred windows, which fills a small
// part of the chip either way, takes half the time per step with 32-row workgroups (twice as many of them, each still
// streaming the direction's weight fragments from L2).  Big calls keep 64 rows: half the weight stream per window.
template <int H, int KX, bool PRE, bool XG = false, int SAUX = 0, bool BC = false, int MTILES = 2>
__global__ __launch_bounds__(H / 32 * 64, 1) void lstm_rec_h2_kernel(const float* __restrict__ Xp, int ldx,
                                                                     const int8_t* __restrict__ Xi, int F,
                                                                     const float* __restrict__ bias,
                                                                     const uint32_t* __restrict__ Wp,
                                                                     uint32_t* __restrict__ Y, int ldy, int B, int T,
                                                                     unsigned long long* __restrict__ dbg,
                                                                     const uint32_t* __restrict__ Xh = nullptr, int ldxh = 0) {
    constexpr int KT = H + KX, KS = KT / 16, KSH = H / 16, NT = H / 32, NW = H / 32;
    constexpr int MTL = 32 * MTILES;             // rows of this workgroup
    static_assert(MTILES == 2 || (MTILES == 1 && !XG), "the x ring of the XG form is laid out for 64 rows");
    constexpr int KL = XG ? H : KT;              // columns kept in the LDS rows
    constexpr int ROWB = KL * 4 + 16;            // bytes per LDS row: h2 image of [h | x] + 16 pad (odd 16-B count)
    constexpr int ROWD = ROWB / 4;               // in dwords
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];   // [MTL][ROWD] h2 rows, then c (f32)
    static_assert(KS % 2 == 0, "the weight prefetch assumes an even number of k steps");
    static_assert((ROWB / 16) % 2 == 1, "row stride must be an odd number of 16-byte slots");

    int dir, btile;
    decode_block(blockIdx.x, dir, btile);
    const int b0 = btile * MTL;
    if (b0 >= B) return;

    const int tid = threadIdx.x, lane = tid & 63;
    const int u = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, hf = lane >> 5;

    float* cs = reinterpret_cast<float*>(lds + MTL * ROWD) + u * (MTILES * 16 * 64) + lane;     // [wave][m][r][lane]
    for (int idx = tid; idx < MTL * ROWD + NW * MTILES * 16 * 64; idx += blockDim.x) lds[idx] = 0u;
    // XG: x ring after the cell state: 2 slots x [MTL rows][XRD dwords] (128 B of x + 16 B pad per row)
    constexpr int XRD = 36, XSLOT = MTL * XRD, NXI = XG ? KX / 32 : 0;   // NXI iterations of two k steps
    uint32_t* xring = lds + MTL * ROWD + NW * MTILES * 16 * 64;

    f32x16 acc[MTILES][4];   // [row tile][gate]

    const size_t urow = (size_t)b0 * T;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(KX ? bias + dir * 4 * H + 32 * u : Xp + (size_t)(b0 >> 5) * T * (ldx >> 5) * 1024), 0,
        0x7fffffff, 0x00020000);
    // fragment (g, s, hi/lo) of this wave lives at byte (((g*NT + u) * KS + s) * 2 + hl) * 1024 + lane * 16 of the direction's
    // block.  The wave's own term (u) is part of the descriptor's base, so that every fragment's scalar offset is a
    // compile-time constant the compiler re-materialises with one s_mov where it needs it: with `u` inside the offset the
    // 8 KS sums were loop-invariant VALUES, kept live across the time loop and spilled -- 315 scalar registers parked in vector
    // lanes and 314 v_readlane_b32 per time step in the K = 768 form.
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint32_t*>(Wp + ((size_t)dir * (4 * NT) + u) * KS * 512), 0, 0x7fffffff, 0x00020000);
    const unsigned xoff = KX ? li * 4u : lane * 16u;
    const unsigned woff = lane * 16u;

    // accumulator seed of (row tile m, register chunk qd) for step t: the unfused form starts from the precomputed input
    // projection (identical to rnn.hip).  Fused forms (KX > 0) seed nothing: the first product into every accumulator starts from the inline constant 0 and the
    // four biases of the lane's column are added in the gate phase (packed adds; with the bias column they are zero).
    auto seed_chunk = [&](int m, int qd, int t) {
        if (KX) return;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            {
                const unsigned ct = (unsigned)(dir * (4 * NT) + g * NT + u);
                const unsigned so = (((unsigned)(m * T + t) * (ldx >> 5) + ct) * 4u + qd) * 1024u;
                const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, xoff, so, 0));
                acc[m][g][4 * qd] = v.x;
                acc[m][g][4 * qd + 1] = v.y;
                acc[m][g][4 * qd + 2] = v.z;
                acc[m][g][4 * qd + 3] = v.w;
            }
        }
    };

    // fused only: x_t (int8, exact in f16) -> hi halves of LDS columns [H, H+KX); lo halves stay 0.
    // One thread per pair of features: 64 rows x KX/2 pairs over 512 threads.
    constexpr bool XI8 = KX > 0 && !XG;
    constexpr int XN = XI8 ? (MTL * KX / 2) / (NW * 64) : 1;
    unsigned xv[XN];
    auto x_load = [&](int t) {
        if (XI8) {
#pragma unroll
            for (int k = 0; k < XN; ++k) {
                const int e = tid + k * (NW * 64);
                const int row = e / (KX / 2), f = (e % (KX / 2)) * 2;
                int brow = b0 + row;
                brow = brow < B ? brow : B - 1;
                const int8_t* src = Xi + ((size_t)brow * T + t) * F;
                const _Float16 one = BC ? (_Float16)1.0f : (_Float16)0.0f;     // the bias column's input
                const _Float16 h0 = f < F ? (_Float16)(float)src[f] : (f == F ? one : (_Float16)0.0f);
                const _Float16 h1 = f + 1 < F ? (_Float16)(float)src[f + 1] : (f + 1 == F ? one : (_Float16)0.0f);
                xv[k] = (unsigned)__builtin_bit_cast(unsigned short, h0) |
                        ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
            }
        }
    };
    auto x_store = [&]() {
        if (XI8) {
#pragma unroll
            for (int k = 0; k < XN; ++k) {
                const int e = tid + k * (NW * 64);
                const int row = e / (KX / 2), f = (e % (KX / 2)) * 2;
                lds[row * ROWD + ((H + f) >> 3) * 8 + ((f & 7) >> 1)] = xv[k];
            }
        }
    };
    // XG staging: thread -> (row = tid / 8, 16-byte chunk c = tid % 8) of an iteration's 64 x 128 B slab
    const __amdgpu_buffer_rsrc_t xgrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint32_t*>(XG ? Xh + (size_t)b0 * T * ldxh : Wp), 0, 0x7fffffff, 0x00020000);
    const unsigned xg_off = XG ? ((unsigned)((tid >> 3) * T) * ldxh) * 4u + (tid & 7) * 16u : 0u;
    uint32_t* xst_dst = xring + (tid >> 3) * XRD + (tid & 7) * 4;
    u32x4 xstage[2];
    auto xg_load = [&](int j, int t) {      // iteration j of step t -> staging register j & 1
        xstage[j & 1] = __builtin_amdgcn_raw_buffer_load_b128(xgrs, xg_off, ((unsigned)t * ldxh) * 4u + (unsigned)j * 128u, SAUX);
    };
    auto xg_store = [&](int j) { *reinterpret_cast<u32x4*>(xst_dst + (j & 1) * XSLOT) = xstage[j & 1]; };

    struct Frag { h8 b[4][2], a[MTILES][2]; };     // [gate][hi, lo], [row tile][hi, lo]
    const uint32_t* arow = lds + li * ROWD + hf * 8;
    bool exp_loads = true;       // PA_EXP_NO_BLOAD (energy experiment, wrong results): weight fragments loaded once, not per step
    auto load_b = [&](int s, Frag& fr) {
#ifdef PA_EXP_NO_BLOAD
        if (!exp_loads) return;
#endif
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int hl = 0; hl < 2; ++hl)
                fr.b[g][hl] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(
                                                         wrs, woff, (unsigned)((g * NT * KS + s) * 2 + hl) * 1024u, 0));
    };
    auto load_a = [&](int s, Frag& fr, int) {
        if (XG && s >= KSH) {
            const int xs = s - KSH;
            const uint32_t* src = xring + ((xs >> 1) & 1) * XSLOT + li * XRD + ((xs & 1) * 2 + hf) * 8;
#pragma unroll
            for (int m = 0; m < MTILES; ++m) {
                fr.a[m][0] = *reinterpret_cast<const h8*>(src + m * 32 * XRD);
                fr.a[m][1] = *reinterpret_cast<const h8*>(src + m * 32 * XRD + 4);
            }
            return;
        }
#pragma unroll
        for (int m = 0; m < MTILES; ++m) {
            fr.a[m][0] = *reinterpret_cast<const h8*>(arow + m * 32 * ROWD + s * 16);
            fr.a[m][1] = *reinterpret_cast<const h8*>(arow + m * 32 * ROWD + s * 16 + 4);
        }
    };

    // y copy: the h part of the LDS rows (H * 4 bytes = H/4 16-byte chunks per row) -> Y, as is
    constexpr int CPR = H / 4;                                // 16-byte chunks per row
    constexpr int YROWS = (NW * 64) / CPR;                    // rows per pass (8)
    constexpr int YC = MTL / YROWS;                            // passes (8)
    const int yc_row = tid / CPR, yc_c = tid % CPR;
    const uint32_t* yc_src = lds + yc_row * ROWD + yc_c * 4;
    const __amdgpu_buffer_rsrc_t ycrs =
        __builtin_amdgcn_make_buffer_rsrc(Y + urow * ldy + dir * H, 0, 0x7fffffff, 0x00020000);
    const unsigned yc_off = ((unsigned)(yc_row * T) * ldy + yc_c * 4) * 4u;
    auto yc_read = [&](int j) { return *reinterpret_cast<const u32x4*>(yc_src + j * YROWS * ROWD); };
    auto yc_write = [&](int j, int tp, u32x4 v) {
        __builtin_amdgcn_raw_buffer_store_b128(v, ycrs, yc_off, ((unsigned)(j * YROWS * T + tp) * ldy) * 4u, SAUX);
    };

    // gate-phase h write: element (row, col = 32u + li) -> hi half at row*ROWB + (col/8)*32 + (col%8)*2,
    // lo half 16 bytes later: two 16-bit stores from the lane's own pair (h2_store16).
    const int hcol = 32 * u + li;
    unsigned short* hl_dst = reinterpret_cast<unsigned short*>(lds + 4 * hf * ROWD + (hcol >> 3) * 8 + ((hcol & 7) >> 1)) + (hcol & 1);
    const bool odd = li & 1;

    // The weights do not depend on the step: the B fragments of k step 0 are requested during the last
    // k step of the previous time step and fly under the gate phase (ring[0].b stays live across it).
    Frag ring[2];
    __syncthreads();
    {
        const int t0 = dir ? T - 1 : 0;
#pragma unroll
        for (int m = 0; m < MTILES; ++m)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) seed_chunk(m, qd, t0);
        x_load(t0);
        x_store();
        load_b(0, ring[0]);
#ifdef PA_EXP_NO_BLOAD
        load_b(1, ring[1]);
        exp_loads = false;
#endif
    }
    __syncthreads();

    for (int step = 0; step < T; ++step) {
        const int t = dir ? T - 1 - step : step;
        const bool stamp = dbg != nullptr && blockIdx.x == 8 && lane == 0;
        if (stamp) dbg[(u * 80 + 2 * step) * 2] = __builtin_amdgcn_s_memtime();
        // ---------------- MFMA phase ----------------
        {
            load_a(0, ring[0], t);
            const int tp = step > 0 ? (dir ? t + 1 : t - 1) : t;   // time index of h_{s-1} (step 0: zeros, rewritten later)
            u32x4 ycv = {0, 0, 0, 0};
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int p = s & 1;
                if (XG) {
                    // iteration j = k steps (KSH + 2j, KSH + 2j + 1); its slab is loaded at step 2j + 9 (j = 0, 1: steps
                    // 0, 1), written to slot j & 1 at step 2j + 13 (j = 0: step 8) and published by the barrier at the
                    // start of step 2j + 15, the step that pre-reads its first fragments
                    if (s == 0) xg_load(0, t);
                    if (s == 1) xg_load(1, t);
                    if (s >= KSH - 1 && ((s - (KSH - 1)) & 1) == 0 && (s - (KSH - 1)) / 2 < NXI) lds_barrier();
                    if (s == 8) xg_store(0);
                    if (s >= 15 && ((s - 13) & 1) == 0 && (s - 13) / 2 < NXI) xg_store((s - 13) / 2);
                    if (s >= 13 && ((s - 9) & 1) == 0 && (s - 9) / 2 < NXI) xg_load((s - 9) / 2, t);
                }
                if (s + 1 < KS) { load_b(s + 1, ring[p ^ 1]); load_a(s + 1, ring[p ^ 1], t); }
                else load_b(0, ring[p ^ 1]);            // KS is even: ring[p ^ 1] == ring[0]
                if (s >= 1 && s <= YC) yc_write(s - 1, tp, ycv);
                if (s < YC) ycv = yc_read(s);
#pragma unroll
                for (int term = 0; term < 3; ++term) {
                    // int8 summaries are exact in the hi half: their lo half is identically zero, so is lo(a) * hi(w)
                    if (XI8 && s >= KSH && term == 0) continue;
                    const bool fresh = KX && s == 0 && term == 0;       // accumulators start from the inline constant 0
#pragma unroll
                    for (int g = 0; g < 4; ++g)
#pragma unroll
                        for (int m = 0; m < MTILES; ++m)
                            acc[m][g] = mfma_h(ring[p].a[m][term == 0 ? 1 : 0], ring[p].b[g][term == 1 ? 1 : 0],
                                               fresh ? f32x16{} : acc[m][g]);
                }
                if (XG && s + 1 < KS && s > YC) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);   // 2 MFMA
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // 1 VMEM read (B fragment)
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read (A fragment: h rows or x ring)
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);       // x slab chunk -> ring
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);       // x slab chunk of a later iteration
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                } else if (MTILES == 1 && s + 1 < KS) {
                    // 12 MFMAs per k step: 8 weight fragments, 2 A fragments, the y copy's read / write in the first steps
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // 1 VMEM read (B fragment)
                    }
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read (A fragment)
                    }
                    if (s <= YC) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);   // 1 VMEM write (y copy)
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read (y copy)
                    } else {
                        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    }
                } else if (s + 1 < KS) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);   // 2 MFMA
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // 1 VMEM read (B fragment)
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read (A fragment)
                    }
                    if (s <= YC) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);   // 1 VMEM write (y copy)
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read (y copy)
                        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    } else {
                        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                    }
                } else if (MTILES == 1) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // next time step's first B fragments
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                } else {
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // next time step's first B fragments
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // the lane's four biases, re-read every step (in flight across the barrier) rather than held through the MFMA phase
        float cb[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (KX && !BC) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
                cb[g] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, xoff, (unsigned)(g * H) * 4u, 0));
        }
        if (stamp) dbg[(u * 80 + 2 * step) * 2 + 1] = __builtin_amdgcn_s_memtime();

        lds_barrier();                    // every wave has finished reading h_{t-1}
        if (stamp) dbg[(u * 80 + 2 * step + 1) * 2] = __builtin_amdgcn_s_memtime();

        // ---------------- gate phase ----------------
        // (Starting a wave's gate math before the barrier, beside the other wave's MFMAs, was measured:
        // the VALU / transcendental issue slows that MFMA stream by about what it hides.  The phase is
        // transcendental-bound: 10 quarter-rate ops per element.)
        const int tn = dir ? t - 1 : t + 1;
        if (step + 1 < T) x_load(tn);
#pragma unroll
        for (int m = 0; m < MTILES; ++m)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
#pragma unroll
                for (int e = 0; e < 4; e += 2) {
                    // two elements per pass: the plain arithmetic is written on 2-vectors so it can issue as
                    // packed f32 instructions (the matrix pipe is idle in this phase)
                    const int r = 4 * qd + e;
                    typedef float f32x2 __attribute__((ext_vector_type(2)));
                    const f32x2 one = {1.0f, 1.0f};
                    f32x2 ai = {acc[m][0][r], acc[m][0][r + 1]}, af = {acc[m][1][r], acc[m][1][r + 1]},
                          ag = {acc[m][2][r], acc[m][2][r + 1]}, ao = {acc[m][3][r], acc[m][3][r + 1]};
                    if (KX && !BC) { ai += cb[0]; af += cb[1]; ag += cb[2]; ao += cb[3]; }
                    if (!PRE) {
                        ai *= -1.4426950408889634f; af *= -1.4426950408889634f;
                        ag *= 2.8853900817779268f;  ao *= -1.4426950408889634f;
                    }
                    auto ex2 = [](f32x2 v) { return f32x2{__builtin_amdgcn_exp2f(v.x), __builtin_amdgcn_exp2f(v.y)}; };
                    auto rcp = [](f32x2 v) { return f32x2{__builtin_amdgcn_rcpf(v.x), __builtin_amdgcn_rcpf(v.y)}; };
                    const f32x2 ig = rcp(one + ex2(ai));
                    const f32x2 fg = rcp(one + ex2(af));
                    const f32x2 gg = one - 2.0f * rcp(one + ex2(ag));
                    const f32x2 og = rcp(one + ex2(ao));
                    const f32x2 cold = {cs[(m * 16 + r) * 64], cs[(m * 16 + r + 1) * 64]};
                    const f32x2 cn = fg * cold + ig * gg;
                    cs[(m * 16 + r) * 64] = cn.x;
                    cs[(m * 16 + r + 1) * 64] = cn.y;
                    const f32x2 hv2 = og * (one - 2.0f * rcp(one + ex2(cn * 2.8853900817779268f)));
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const float hv = k ? hv2.y : hv2.x;
                        h2_store16(hl_dst + (32 * m + ((r + k) & 3) + 8 * ((r + k) >> 2)) * ROWD * 2, hv);
                    }
                }
                if (step + 1 < T) seed_chunk(m, qd, tn);
            }
        if (step + 1 < T) x_store();
        if (stamp) dbg[(u * 80 + 2 * step + 1) * 2 + 1] = __builtin_amdgcn_s_memtime();
        lds_barrier();                    // h_t (and x_{t+1}) visible
    }
    {
        const int tl = dir ? 0 : T - 1;
#pragma unroll
        for (int j = 0; j < YC; ++j) yc_write(j, tl, yc_read(j));
    }
}

// ------------------------------------------------------------------------------------------------
// The step loop of a call of at most 1024 windows (PA_UNIT_SPLIT=0: off) with the hidden units of a 32-row tile
// split over EIGHT workgroups (four above 512 windows, NTW = 2) that exchange their slices of h_t through memory every step.
// A step of the 32-row workgroup above is one CU's affair (~10 us: 384 MFMAs on four SIMDs, a 1 MB weight stream, the
// gate phase of 8 waves); with the units split, member j of a (direction, tile) group keeps the recurrent weights of units
// [32j, 32j + 32) in REGISTERS (wave w = gate w: 16 k steps x (hi, lo) fragments = 128 VGPRs, loaded once), issues 48 MFMAs
// per wave and step, applies its gate's activation, and the four waves then share the element-wise update (wave w takes
// accumulator registers 4w .. 4w+3 of every lane: rows 8w .. 8w+7 of the tile) through a 16 KB LDS buffer.  What the split
// costs is the exchange: the member's 4 KB of new h (h2 image, 128 B per row) go to a buffer in memory, the members meet at
// a counter, and every member reads the whole 32 KB row image back into its LDS.  With agent-scope FENCES that exchange
// measured 6.5-19 us per step (tools/microbench/group_exchange.hip); with every exchanged word written and read by an
// sc1 (agent-coherent) access and no fence it measured 2.4-3.2 us (profiles/r03_group_exchange_microbench.txt) -- that is
// what this kernel does.  All eight members of a group must be resident at once: the launcher takes at most 256 workgroups
// (two fit a CU), the spin is bounded (~25 ms), and a group that does not meet sets *failed and returns instead of hanging:
// the host then runs the call again with the ordinary small-call schedule (api.hip variant_forward_chunk).
// PRE form only (weights, Xp and bias pre-multiplied by the exp2 factors; Xp = x W_ih^T + b from the projection GEMM).
constexpr int US_AUX_SC1 = 16;                // cache policy bit sc1 of the gfx940+ buffer instructions

// NTW: unit tiles of 32 per member.  1: eight members (calls of at most 512 windows: 256 workgroups); 2: four members of 64
// units (513-1024 windows), each wave with two accumulators and 256 VGPRs of weights.
template <int H, int NTW>
__global__ __launch_bounds__(256, 1) void lstm_rec_h2_split_kernel(const float* __restrict__ Xp, int ldx,
                                                                   const uint32_t* __restrict__ Wp, uint32_t* __restrict__ Y,
                                                                   int ldy, int B, int T, uint32_t* __restrict__ exch,
                                                                   unsigned* __restrict__ counters, int* __restrict__ failed,
                                                                   int tiles4, int sabotage) {
    static_assert(H == 256 && (NTW == 1 || NTW == 2), "eight members of 32 units or four of 64");
    constexpr int KS = H / 16, NT = H / 32, K = NT / NTW;
    constexpr int ROWB = H * 4 + 16, ROWD = ROWB / 4;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];      // [32][ROWD] h2 rows | gate buffer [4][NTW][16][64] f32 | flag
    float* gbuf = reinterpret_cast<float*>(lds + 32 * ROWD);
    uint32_t* flag = lds + 32 * ROWD + 4 * NTW * 16 * 64;

    // members of a group share an XCD (blockIdx % 8) and follow each other there; XCDs 0-3 forward, 4-7 reverse (rnn.hip)
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int dir = xcd >> 2, member = q % K, btile = (q / K) * 4 + (xcd & 3);
    const int b0 = btile * 32;
    if (b0 >= B) return;                                                  // (the whole group returns)
    const int group = dir * tiles4 + btile;

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);               // = gate: i, f, g, o
    const int li = lane & 31, hf = lane >> 5;
    for (int idx = tid; idx < 32 * ROWD; idx += 256) lds[idx] = 0u;
    if (tid == 0) *flag = 0u;

    // this wave's recurrent weights: fragment (gate w, unit tile u, k step s, hi / lo) at byte
    // (((w * NT + u) * KS + s) * 2 + hl) * 1024 + lane * 16 of the direction's block (pack_rec_weights_h2)
    h8 bw[NTW][KS][2];
    {
        const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<uint32_t*>(Wp + (size_t)dir * (4 * NT) * KS * 512), 0, 0x7fffffff, 0x00020000);
#pragma unroll
        for (int j = 0; j < NTW; ++j)
#pragma unroll
            for (int s = 0; s < KS; ++s)
#pragma unroll
                for (int hl = 0; hl < 2; ++hl)
                    bw[j][s][hl] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(
                                                              wrs, lane * 16u,
                                                              (unsigned)(((w * NT + member * NTW + j) * KS + s) * 2 + hl) * 1024u, 0));
    }
    // accumulator seeds: the projection's tile (row tile, t, column tile ct, register chunk qd), as lstm_rec_h2_kernel reads it
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(Xp + (size_t)btile * T * (ldx >> 5) * 1024), 0, 0x7fffffff, 0x00020000);
    const unsigned ct0 = (unsigned)(dir * (4 * NT) + w * NT + member * NTW);
    f32x16 acc[NTW];
    auto seed = [&](int t) {
#pragma unroll
        for (int j = 0; j < NTW; ++j)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const unsigned so = (((unsigned)t * (ldx >> 5) + ct0 + j) * 4u + qd) * 1024u;
                const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, lane * 16u, so, 0));
                acc[j][4 * qd] = v.x;
                acc[j][4 * qd + 1] = v.y;
                acc[j][4 * qd + 2] = v.z;
                acc[j][4 * qd + 3] = v.w;
            }
    };

    const uint32_t* arow = lds + li * ROWD + hf * 8;
    const int hcol = 32 * member * NTW + li;                              // (+ 32 j: 64 halves further on in the row)
    unsigned short* hl_dst = reinterpret_cast<unsigned short*>(lds + 4 * hf * ROWD + (hcol >> 3) * 8 + ((hcol & 7) >> 1)) + (hcol & 1);
    // exchange geometry: thread = (row, 16-byte chunk xc of every 128 bytes of that row); a member's slice is NTW x 128 bytes
    const int xrow = tid >> 3, xc = tid & 7;
    const __amdgpu_buffer_rsrc_t ers = __builtin_amdgcn_make_buffer_rsrc(
        exch + (size_t)group * 2 * 32 * H, 0, 0x7fffffff, 0x00020000);   // [parity][32 rows][H dwords]
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(
        Y + (size_t)b0 * T * ldy + dir * H, 0, 0x7fffffff, 0x00020000);
    unsigned* cnt = counters + group * 32;                                // own 128-byte line

    float c[NTW][4];
#pragma unroll
    for (int j = 0; j < NTW; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) c[j][e] = 0.0f;
    seed(dir ? T - 1 : 0);
    __syncthreads();

    for (int step = 0; step < T; ++step) {
        const int t = dir ? T - 1 - step : step;
        // ---- contraction h_{t-1} W_hh^T of this wave's gate (three products per k step, lstm_rec_h2_kernel's order) ----
        {
            h8 a0 = *reinterpret_cast<const h8*>(arow), a1 = *reinterpret_cast<const h8*>(arow + 4);
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                h8 n0 = a0, n1 = a1;
                if (s + 1 < KS) {
                    n0 = *reinterpret_cast<const h8*>(arow + (s + 1) * 16);
                    n1 = *reinterpret_cast<const h8*>(arow + (s + 1) * 16 + 4);
                }
#pragma unroll
                for (int j = 0; j < NTW; ++j) acc[j] = mfma_h(a1, bw[j][s][0], acc[j]);
#pragma unroll
                for (int j = 0; j < NTW; ++j) acc[j] = mfma_h(a0, bw[j][s][1], acc[j]);
#pragma unroll
                for (int j = 0; j < NTW; ++j) acc[j] = mfma_h(a0, bw[j][s][0], acc[j]);
                a0 = n0;
                a1 = n1;
            }
        }
        // ---- this wave's activation (the accumulators are exp2 arguments): sigmoid for i, f, o; tanh for g ----
#pragma unroll
        for (int j = 0; j < NTW; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(acc[j][r]));
                gbuf[((w * NTW + j) * 16 + r) * 64 + lane] = w == 2 ? 1.0f - 2.0f * sg : sg;
            }
        if (step + 1 < T) seed(dir ? t - 1 : t + 1);      // the next step's seeds fly under the rest of this one
        lds_barrier();                                    // gates visible; every wave has finished reading h_{t-1}
        // ---- element-wise update, a quarter of the tile per wave: registers 4w .. 4w+3 = rows 8w + e + 4 hf ----
#pragma unroll
        for (int j = 0; j < NTW; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = 4 * w + e;
                const float ig = gbuf[((0 * NTW + j) * 16 + r) * 64 + lane], fg = gbuf[((1 * NTW + j) * 16 + r) * 64 + lane];
                const float gg = gbuf[((2 * NTW + j) * 16 + r) * 64 + lane], og = gbuf[((3 * NTW + j) * 16 + r) * 64 + lane];
                const float cn = fg * c[j][e] + ig * gg;
                c[j][e] = cn;
                const float hv = og * (1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(cn * 2.8853900817779268f)));
                h2_store16(hl_dst + j * 64 + (e + 8 * w) * ROWD * 2, hv);
            }
        lds_barrier();                                    // this member's slice of h_t is complete in LDS
        // ---- publish: y_t and (not after the last step) the exchange buffer of this step's parity ----
        const unsigned pbase = (unsigned)(step & 1) * (32u * H * 4u);
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            const int chunk = (member * NTW + j) * 8 + xc;
            const u32x4 mine = *reinterpret_cast<const u32x4*>(lds + xrow * ROWD + chunk * 4);
            __builtin_amdgcn_raw_buffer_store_b128(mine, yrs, ((unsigned)(xrow * T) * ldy + chunk * 4) * 4u, ((unsigned)t * ldy) * 4u, 0);
            if (step + 1 < T)
                __builtin_amdgcn_raw_buffer_store_b128(mine, ers, (unsigned)(xrow * H + chunk * 4) * 4u, pbase, US_AUX_SC1);
        }
        if (step + 1 == T) break;
        __builtin_amdgcn_s_waitcnt(0);                    // every store of this thread acknowledged ...
        __builtin_amdgcn_s_barrier();                     // ... and of this member
        if (tid == 0) {
            // (sabotage: the tests' way to a group that does not meet -- its last member never arrives)
            if (!(sabotage && group == 0 && member == K - 1))
                __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = (unsigned)K * (unsigned)(step + 1);
            // The first meeting waits for members that may not have been dispatched yet (other queues hold the CUs): ~12 ms.
            // Members that have met once are all resident and stay so (a queue is preempted as a whole, and then nobody's
            // count advances), so a later meeting that takes more than ~3 ms -- a thousand exchanges' worth -- is a lost group.
            const int bound = step == 0 ? (1 << 13) : (1 << 11);
            int spins = 0;
            while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                if (++spins > bound) {                    // the group is not resident together; the host runs the call again
                    *flag = 1u;
                    __hip_atomic_store(failed, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        lds_barrier();
        if (*flag) return;                                // (uniform: the whole member gives up; the host runs the call again)
        // ---- every member's slice of h_t (the own one included: same bytes) -> the LDS rows ----
        {
            u32x4 v[NT];
#pragma unroll
            for (int m2 = 0; m2 < NT; ++m2)
                v[m2] = __builtin_amdgcn_raw_buffer_load_b128(ers, (unsigned)(xrow * H + (m2 * 8 + xc) * 4) * 4u, pbase, US_AUX_SC1);
#pragma unroll
            for (int m2 = 0; m2 < NT; ++m2) *reinterpret_cast<u32x4*>(lds + xrow * ROWD + (m2 * 8 + xc) * 4) = v[m2];
        }
        lds_barrier();                                    // h_t visible
    }
}

// ------------------------------------------------------------------------------------------------
// GRU step loop (polish model, H = 128), same split-operand scheme.  Always 8 waves: four unit tiles
// x two 64-row groups (128 batch rows per workgroup).  Gates r, z, n with the PyTorch definition
// n = tanh(W_in x + b_in + r * (W_hn h + b_hn)); the hidden half of the n gate has its own accumulator
// (and, when the uint8 input projection is fused, so has the input half).  h is carried exactly in f32
// registers for the z * h_{t-1} term and the final state hand-off; the LDS / y image is its h2 split.
//   /root/reference/pepper/modules/python/models/simple_model.py:30,32
// XG: the layer input is an h2 layer output Xh [B*T, KX] streamed through a two-slot LDS ring exactly as in
// lstm_rec_h2_kernel (two k steps = 128 rows x 128 B per slot, two 16-byte chunks per thread per iteration).
// DENSE (with XG; the polish model's last decoder layer): the layer output never leaves the workgroup.  Its only
// consumer is dense1 (2H -> C classes, /root/reference/pepper/modules/python/models/simple_model.py:34), so while step
// s+1 contracts, the h_s rows still in LDS are multiplied with this direction's half of dense1's weights on
// v_mfma_f32_16x16x32_f16 (same three-term split product; wave (u, rg) takes k in [32u, 32u+32) of row group rg: 12 small
// MFMAs per step), the four k-quarter partials are summed through an LDS scratch behind a barrier the x ring needs
// anyway, and 5 x 128 floats per step go to P[dir][batch tile][t][class][128 rows] (2.5 KB instead of the 128 KB of y).
// polish_combine_kernel (head.hip) adds the two directions and the bias, takes the softmax and overlap-adds it.
// BC: as in lstm_rec_h2_kernel, for the fused uint8 first layer with F < KX = 16: b_r, b_z and the input half of the n
// gate's bias come out of the matrix pipe (bias column of the packed weights x constant 1.0 input); b_hn is added in the
// gate phase.  No fused form seeds an accumulator (see seed_chunk below).
template <int H, int KX, bool XG = false, int SAUX = 0, bool DENSE = false, bool BC = false>
__global__ __launch_bounds__(512, 1) void gru_rec_h2_kernel(const float* __restrict__ Xp, int ldx,
                                                            const uint8_t* __restrict__ Xi, int F, int64_t xi_bstride,
                                                            const float* __restrict__ bias,
                                                            const uint32_t* __restrict__ Wp,
                                                            const float* __restrict__ bhn,
                                                            const float* __restrict__ h0, int ldh0,
                                                            float* __restrict__ hn, int ldhn,
                                                            uint32_t* __restrict__ Y, int ldy, int B, int T,
                                                            const uint32_t* __restrict__ Xh = nullptr, int ldxh = 0,
                                                            const uint32_t* __restrict__ Wd = nullptr,
                                                            float* __restrict__ P = nullptr,
                                                            unsigned long long* __restrict__ dbg = nullptr) {
    static_assert(!DENSE || (XG && H == 128), "the fused head belongs to the H = 128 layer fed by an h2 layer output");
    constexpr int DC = 5;                        // classes of the fused head (columns of the 16-wide MFMA tile in use)
    constexpr int KT = H + KX, KS = KT / 16, KSH = H / 16, NT = H / 32, RG = 8 / NT, MTG = MT * RG;
    constexpr int NA = KX ? 4 : 3;
    constexpr int KL = XG ? H : KT;
    constexpr int ROWB = KL * 4 + 16, ROWD = ROWB / 4;
    constexpr int XRD = 36, XSLOT = MTG * XRD, NXI = XG ? KX / 32 : 0;
    static_assert(!XG || KSH >= 7, "ring schedule needs KSH >= 7");
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];   // [MTG][ROWD] h2 rows of [h | x] (+ x ring when XG)
    static_assert((ROWB / 16) % 2 == 1, "row stride must be an odd number of 16-byte slots");

    int dir, btile;
    decode_block(blockIdx.x, dir, btile);
    const int b0 = btile * MTG;
    if (b0 >= B) return;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int u = wave % NT, rg = wave / NT;
    const int li = lane & 31, hf = lane >> 5;
    const int col = u * 32 + li;
    const int r0 = b0 + rg * MT;

    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(KX ? bhn : Xp + (size_t)(r0 >> 5) * T * (ldx >> 5) * 1024), 0, 0x7fffffff, 0x00020000);
    // fragment (g, s, hi/lo) of this wave: byte (((g*NT + u) * KS + s) * 2 + hl) * 1024 + lane * 16
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint32_t*>(Wp + ((size_t)dir * (3 * NT) + u) * KS * 512), 0, 0x7fffffff, 0x00020000);
    const unsigned xoff = lane * 16u, woff = lane * 16u;
    const size_t lb = (size_t)(r0 + 4 * hf);
    const bool odd = li & 1;
    // the lane's column in the hi half of its k chunk, as a 16-bit address (h2_store16)
    unsigned short* hl_dst = reinterpret_cast<unsigned short*>(lds + (rg * MT + 4 * hf) * ROWD + (col >> 3) * 8 + ((col & 7) >> 1)) + (col & 1);
    const uint32_t* arow = lds + (rg * MT + li) * ROWD + hf * 8;

    for (int idx = tid; idx < MTG * ROWD; idx += 512) lds[idx] = 0u;
    __syncthreads();
    uint32_t* xring = lds + MTG * ROWD;
    // DENSE: after the ring, this direction's dense1 fragments [k quarter 4][hi, lo][64 lanes][16 B] (8 KB, copied once)
    // and the partial-logit scratch [k quarter 4][class DC][MTG rows] f32 (10 KB)
    uint32_t* dw_lds = xring + 2 * XSLOT;
    float* dsc = reinterpret_cast<float*>(dw_lds + 4 * 2 * 256);
    if (DENSE) {
        for (int idx = tid; idx < 4 * 2 * 256; idx += 512) dw_lds[idx] = Wd[(size_t)dir * (4 * 2 * 256) + idx];
        __syncthreads();
    }
    const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(
        DENSE ? P + ((size_t)dir * ((B + MTG - 1) / MTG) + btile) * T * (DC * MTG) : const_cast<float*>(bhn), 0, 0x7fffffff,
        0x00020000);
    auto dense_partials = [&]() {
        typedef float f32x4v __attribute__((ext_vector_type(4)));
        int dl = lane;                    // opaque: the addresses below are recomputed per call, not held in registers
        asm volatile("" : "+v"(dl));
        const int i16 = dl & 15, kg = dl >> 4;
        const uint32_t* drow = lds + (rg * MT + i16) * ROWD + (u * 4 + kg) * 8;
        const h8 bd_hi = *reinterpret_cast<const h8*>(dw_lds + (u * 2 + 0) * 256 + dl * 4);
        const h8 bd_lo = *reinterpret_cast<const h8*>(dw_lds + (u * 2 + 1) * 256 + dl * 4);
        h8 ah[4], al[4];
        f32x4v dacc[4];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
            ah[rt] = *reinterpret_cast<const h8*>(drow + rt * 16 * ROWD);
            al[rt] = *reinterpret_cast<const h8*>(drow + rt * 16 * ROWD + 4);
            dacc[rt] = f32x4v{0.0f, 0.0f, 0.0f, 0.0f};
        }
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) dacc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[rt], bd_hi, dacc[rt], 0, 0, 0);
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) dacc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[rt], bd_lo, dacc[rt], 0, 0, 0);
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) dacc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[rt], bd_hi, dacc[rt], 0, 0, 0);
        // D: column lane & 15 = class, rows 4 * (lane >> 4) + r of row tile rt
        if (i16 < DC) {
#pragma unroll
            for (int rt = 0; rt < 4; ++rt)
                *reinterpret_cast<f32x4v*>(dsc + (u * DC + i16) * MTG + rg * MT + rt * 16 + kg * 4) = dacc[rt];
        }
    };
    // behind a barrier: sum the four k quarters, 4-byte stores coalesced over the 128 rows
    auto dense_store = [&](int tp) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            // value index = class * MTG + row; the second round has 128 values for 512 threads: the surplus threads
            // repeat the last one (same address, same bits) so the unrolled k step stays free of control flow
            int v = tid + 512 * k;
            v = v < DC * MTG ? v : DC * MTG - 1;
            const float sum = (dsc[v] + dsc[DC * MTG + v]) + (dsc[2 * DC * MTG + v] + dsc[3 * DC * MTG + v]);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, sum), prs, (unsigned)v * 4u,
                                                  (unsigned)tp * (unsigned)(DC * MTG * 4), 0);
        }
    };
    const __amdgpu_buffer_rsrc_t xgrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint32_t*>(XG ? Xh + (size_t)b0 * T * ldxh : Wp), 0, 0x7fffffff, 0x00020000);
    // staging: chunk q of thread tid covers row (tid / 8) + 64 q, 16-byte chunk tid % 8
    const unsigned xg_off = XG ? ((unsigned)((tid >> 3) * T) * ldxh) * 4u + (tid & 7) * 16u : 0u;
    uint32_t* xst_dst = xring + (tid >> 3) * XRD + (tid & 7) * 4;
    u32x4 xstage[2][2];
    auto xg_load = [&](int j, int t) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
            xstage[j & 1][q] = __builtin_amdgcn_raw_buffer_load_b128(
                xgrs, xg_off, ((unsigned)(q * 64 * T + t) * ldxh) * 4u + (unsigned)j * 128u, SAUX);
    };
    auto xg_store = [&](int j) {
#pragma unroll
        for (int q = 0; q < 2; ++q) *reinterpret_cast<u32x4*>(xst_dst + (j & 1) * XSLOT + q * 64 * XRD) = xstage[j & 1][q];
    };

    auto load_xp4 = [&](int m, int t, int g, int qd) {
        const unsigned ct = (unsigned)(dir * (3 * NT) + g * NT + u);
        const unsigned so = (((unsigned)(m * T + t) * (ldx >> 5) + ct) * 4u + qd) * 1024u;
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, xoff, so, 0));
    };
    // packed h2 word of one element (see lstm_rec_h2_kernel)

    // exact f32 h_{t-1} of this lane's 32 elements (the z * h term and the final state).  The forms that sit at the
    // 256-register limit of two waves per SIMD keep the last HL of them in a per-wave private LDS strip [HL][64 lanes]
    // instead of registers (the gate phase has LDS slots to spare; same trick as the LSTM's cell state): all of row
    // tile 1 in the XG forms, half of it in the wide-input form.
    constexpr int HL = XG ? 16 : (KX >= 64 ? 15 : 0);
    float* const hls_wave = reinterpret_cast<float*>(lds + MTG * ROWD + (XG ? 2 * XSLOT : 0) + (DENSE ? 4 * 2 * 256 + 4 * DC * MTG : 0)) +
                            wave * (HL * 64);
    float* hls = hls_wave + lane;
    f32x16 hreg[2], acc[2][NA];
    auto h_get = [&](int m, int r) { return (m == 1 && r >= 16 - HL) ? hls[(r - (16 - HL)) * 64] : hreg[m][r]; };
    auto h_set = [&](int m, int r, float v) {
        if (m == 1 && r >= 16 - HL) hls[(r - (16 - HL)) * 64] = v;
        else hreg[m][r] = v;
    };
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int dr = 32 * m + (r & 3) + 8 * (r >> 2);
            const float hv = h0 != nullptr ? h0[(lb + dr) * ldh0 + dir * H + col] : 0.0f;
            h_set(m, r, hv);
            h2_store16(hl_dst + dr * ROWD * 2, hv);
        }
    const float bn0 = KX ? 0.0f : bhn[dir * H + col];          // unfused form: seeds the hidden half of n
    const __amdgpu_buffer_rsrc_t bhrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(bhn + dir * H), 0, H * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t birs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(bias != nullptr ? bias + dir * 3 * H : bhn), 0, bias != nullptr ? 3 * H * 4 : 0, 0x00020000);
    // Fused forms (KX > 0): no accumulator is seeded.  The first product into each one starts from the inline constant 0
    // and the biases enter the gate arithmetic where they cost nothing -- b_r, b_z, b_in as the addend of the multiply that
    // scales the exponent argument anyway (cr, cz, cn; zero with the bias column, which carries them through the
    // MFMAs), b_hn as one add.  Unfused form: r, z start from the precomputed input projection, the hidden half of n from b_hn.
    // (The four per-column constants are re-read every step, just ahead of the gate phase, instead of living in
    // registers through the MFMA phase: the XG forms sit at the 256-register limit.)
    constexpr float L2E = 1.4426950408889634f;
    auto seed_chunk = [&](int m, int qd, int t) {
        if (KX) return;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const f32x4 v = load_xp4(m, t, g, qd);
            acc[m][g][4 * qd] = v.x;
            acc[m][g][4 * qd + 1] = v.y;
            acc[m][g][4 * qd + 2] = v.z;
            acc[m][g][4 * qd + 3] = v.w;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[m][2][4 * qd + e] = bn0;
    };
    // fused: uint8 x_t (exact in f16) -> hi halves of columns [H, H+KX).  KX = 16: one thread per feature pair,
    // byte loads.  KX >= 64 (wide summaries; the launcher requires F % 4 == 0 and 4-byte aligned rows): one thread
    // per four features, one dword load and one 8-byte LDS write each.
    constexpr bool XU8 = KX > 0 && !XG;
    constexpr int XPT = KX >= 64 ? 4 : 2;                               // features per thread slot
    constexpr int XN = XU8 ? (MTG * KX / XPT) / 512 : 1;
    unsigned xv[XN];
    auto x_load = [&](int t) {
        if (XU8) {
#pragma unroll
            for (int k = 0; k < XN; ++k) {
                const int e = tid + k * 512;
                const int row = e / (KX / XPT), f = (e % (KX / XPT)) * XPT;
                int brow = b0 + row;
                brow = brow < B ? brow : B - 1;
                const uint8_t* src = Xi + (size_t)brow * xi_bstride + (size_t)t * F;
                if (XPT == 4) {
                    xv[k] = f < F ? *reinterpret_cast<const uint32_t*>(src + f) : 0u;
                } else {
                    const _Float16 one = BC ? (_Float16)1.0f : (_Float16)0.0f;     // the bias column's input
                    const _Float16 v0 = f < F ? (_Float16)(float)src[f] : (f == F ? one : (_Float16)0.0f);
                    const _Float16 v1 = f + 1 < F ? (_Float16)(float)src[f + 1] : (f + 1 == F ? one : (_Float16)0.0f);
                    xv[k] = (unsigned)__builtin_bit_cast(unsigned short, v0) | ((unsigned)__builtin_bit_cast(unsigned short, v1) << 16);
                }
            }
        }
    };
    auto x_store = [&]() {
        if (XU8) {
#pragma unroll
            for (int k = 0; k < XN; ++k) {
                const int e = tid + k * 512;
                const int row = e / (KX / XPT), f = (e % (KX / XPT)) * XPT;
                uint32_t* dst = lds + row * ROWD + ((H + f) >> 3) * 8 + ((f & 7) >> 1);
                if (XPT == 4) {
                    auto half_bits = [](unsigned byte) { return (unsigned)__builtin_bit_cast(unsigned short, (_Float16)(float)byte); };
                    const unsigned w = xv[k];
                    u32x2 pk;
                    pk.x = half_bits(w & 0xffu) | (half_bits((w >> 8) & 0xffu) << 16);
                    pk.y = half_bits((w >> 16) & 0xffu) | (half_bits(w >> 24) << 16);
                    *reinterpret_cast<u32x2*>(dst) = pk;
                } else {
                    *dst = xv[k];
                }
            }
        }
    };
    {
        const int t0 = dir ? T - 1 : 0;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) seed_chunk(m, qd, t0);
        x_load(t0);
        x_store();
    }
    __syncthreads();

    // y copy: h part of the LDS rows (H/4 16-byte chunks per row), 512 threads
    constexpr int CPR = H / 4, YROWS = 512 / CPR, YC = MTG / YROWS;   // 32 chunks/row, 16 rows/pass, 8 passes
    constexpr int YI = YC < KS - 1 ? YC : KS - 1;                      // passes interleaved with the MFMAs
    const int yc_row = tid / CPR, yc_c = tid % CPR;
    const uint32_t* yc_src = lds + yc_row * ROWD + yc_c * 4;
    const __amdgpu_buffer_rsrc_t ycrs = __builtin_amdgcn_make_buffer_rsrc(
        DENSE ? const_cast<uint32_t*>(Wp) : Y + (size_t)b0 * T * ldy + dir * H, 0, 0x7fffffff, 0x00020000);
    const unsigned yc_off = ((unsigned)(yc_row * T) * ldy + yc_c * 4) * 4u;
    auto yc_read = [&](int j) { return *reinterpret_cast<const u32x4*>(yc_src + j * YROWS * ROWD); };
    auto yc_write = [&](int j, int tp, u32x4 v) {
        __builtin_amdgcn_raw_buffer_store_b128(v, ycrs, yc_off, ((unsigned)(j * YROWS * T + tp) * ldy) * 4u, SAUX);
    };

    struct Frag { h8 b[3][2], a[2][2]; };
    auto load_step = [&](int s, Frag& fr) {
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int hl = 0; hl < 2; ++hl)
                fr.b[g][hl] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(
                                                         wrs, woff, (unsigned)((g * NT * KS + s) * 2 + hl) * 1024u, 0));
        if (XG && s >= KSH) {
            const int xs = s - KSH;
            const uint32_t* src = xring + ((xs >> 1) & 1) * XSLOT + (rg * MT + li) * XRD + ((xs & 1) * 2 + hf) * 8;
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                fr.a[m][0] = *reinterpret_cast<const h8*>(src + m * 32 * XRD);
                fr.a[m][1] = *reinterpret_cast<const h8*>(src + m * 32 * XRD + 4);
            }
            return;
        }
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            fr.a[m][0] = *reinterpret_cast<const h8*>(arow + m * 32 * ROWD + s * 16);
            fr.a[m][1] = *reinterpret_cast<const h8*>(arow + m * 32 * ROWD + s * 16 + 4);
        }
    };

    // PA_DEBUG_TIMING=1 (tools/phase_timing_gru.py): cycles one workgroup's waves spend per phase, summed over the steps
    // (compiled in only with -DPA_GRU_PHASE_TIMING -- PEPPER_AMD_EXTRA_HIPCC_FLAGS for pepper_amd.build -- because the ten
    // extra registers push the XG forms over their 256-register budget)
#ifdef PA_GRU_PHASE_TIMING
    const bool stamp = dbg != nullptr && blockIdx.x == 8 && lane == 0;
    unsigned long long tsum[5] = {0, 0, 0, 0, 0}, tq = 0;
    auto tick = [&](int k) {
        if (stamp) {
            const unsigned long long now = __builtin_amdgcn_s_memtime();
            if (k >= 0) tsum[k] += now - tq;
            tq = now;
        }
    };
#else
    auto tick = [](int) {};
#endif
    for (int step = 0; step < T; ++step) {
        const int t = dir ? T - 1 - step : step;
        const int tp = step > 0 ? (dir ? t + 1 : t - 1) : t;   // time index of h_{s-1} (step 0: h0, rewritten later)
        tick(-1);
        // ---------------- MFMA phase ----------------
        {
            if (DENSE) {
                dense_partials();                       // head partials of h_{s-1} (step 0: of h0, rewritten by step 1)
                __builtin_amdgcn_sched_barrier(0);
            }
            tick(0);
            Frag ring[2];
            load_step(0, ring[0]);
            u32x4 ycv = {0, 0, 0, 0};
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int p = s & 1;
                if (XG) {
                    // x ring schedule (see lstm_rec_h2_kernel): iteration j = k steps (KSH + 2j, KSH + 2j + 1) is loaded at
                    // step KSH - 7 + 2j (j = 0, 1: steps 0, 1), stored at step KSH - 3 + 2j, published by the barrier at the
                    // start of step KSH - 1 + 2j
                    if (s == 0) xg_load(0, t);
                    if (s == 1) xg_load(1, t);
                    if (s >= KSH - 1 && ((s - (KSH - 1)) & 1) == 0 && (s - (KSH - 1)) / 2 < NXI) lds_barrier();
                    if (DENSE && s == KSH - 1) dense_store(tp);     // the scratch is complete behind that barrier
                    if (s >= KSH - 3 && ((s - (KSH - 3)) & 1) == 0 && (s - (KSH - 3)) / 2 < NXI) xg_store((s - (KSH - 3)) / 2);
                    if (s >= KSH - 3 && ((s - (KSH - 7)) & 1) == 0 && (s - (KSH - 7)) / 2 < NXI) xg_load((s - (KSH - 7)) / 2, t);
                }
                if (s + 1 < KS) load_step(s + 1, ring[p ^ 1]);
                if (!DENSE && s >= 1 && s <= YI) yc_write(s - 1, tp, ycv);
                if (!DENSE && s < YI) ycv = yc_read(s);
#pragma unroll
                for (int term = 0; term < 3; ++term) {
                    if (XU8 && s >= KSH && term == 0) continue;    // uint8 summaries: lo(a) == 0
#pragma unroll
                    for (int g = 0; g < 3; ++g)
#pragma unroll
                        for (int m = 0; m < 2; ++m) {
                            const int ai = (g == 2 && s >= KSH) ? NA - 1 : g;
                            // fused forms: the first product into r / z / the hidden half of n (k step 0) and into the input
                            // half of n (first x step; uint8 rows skip its lo(a) term) starts from the inline constant 0
                            const bool fresh = KX && ((ai < 3 && s == 0 && term == 0) || (ai == 3 && s == KSH && term == (XU8 ? 1 : 0)));
                            acc[m][ai] = mfma_h(ring[p].a[m][term == 0 ? 1 : 0], ring[p].b[g][term == 1 ? 1 : 0],
                                                fresh ? f32x16{} : acc[m][ai]);
                        }
                }
                if (s + 1 < KS) {
#pragma unroll
                    for (int q = 0; q < 6; ++q) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);   // 2 MFMA
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // 1 VMEM read (B fragment)
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read (A fragment)
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);       // y copy store
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);       // y copy read
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (!DENSE) {
#pragma unroll
                for (int j = YI; j < YC; ++j) yc_write(j, tp, yc_read(j));     // passes that did not fit a k step
            }
        }
        // unfused: x part of the n gate for this step, in flight across the barrier
        f32x4 xn[2][4];
        if (!KX) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) xn[m][qd] = load_xp4(m, t, 2, qd);
        }
        // The per-lane addresses and selectors of the gate phase are recomputed from the lane id every step (made opaque
        // so the recomputation is not hoisted): loop-invariant otherwise, they would hold registers through the MFMA phase.
        int gl = lane;
        asm volatile("" : "+v"(gl));
        const int gcol = u * 32 + (gl & 31);
        float bn = 0.0f, cr = 0.0f, cz = 0.0f, cn = 0.0f;
        if (KX) {
            const unsigned coff = (unsigned)gcol * 4u;
            auto ldf = [&](const __amdgpu_buffer_rsrc_t& rs, int g) {
                return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, coff, (unsigned)(g * H * 4), 0));
            };
            bn = ldf(bhrs, 0);
            if (!BC) {
                cr = -L2E * ldf(birs, 0);
                cz = -L2E * ldf(birs, 1);
                cn = 2.0f * L2E * ldf(birs, 2);
            }
        }
        tick(1);
        lds_barrier();
        tick(2);

        // ---------------- gate phase ----------------
        {
            hl_dst = reinterpret_cast<unsigned short*>(lds + (rg * MT + 4 * (gl >> 5)) * ROWD + (gcol >> 3) * 8 + ((gcol & 7) >> 1)) + (gcol & 1);
            hls = hls_wave + gl;
        }
        const int tn = dir ? t - 1 : t + 1;
        if (step + 1 < T) x_load(tn);
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
#pragma unroll
                for (int e = 0; e < 4; e += 2) {
                    // two elements per pass on 2-vectors, so the plain arithmetic issues as packed f32 instructions (as in
                    // lstm_rec_h2_kernel); h' = n + z (h - n)
                    const int r = 4 * qd + e;
                    typedef float f32x2 __attribute__((ext_vector_type(2)));
                    auto ex2 = [](f32x2 v) { return f32x2{__builtin_amdgcn_exp2f(v.x), __builtin_amdgcn_exp2f(v.y)}; };
                    auto rcp = [](f32x2 v) { return f32x2{__builtin_amdgcn_rcpf(v.x), __builtin_amdgcn_rcpf(v.y)}; };
                    auto fma2 = [](f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); };
                    auto dup = [](float v) { return f32x2{v, v}; };
                    const f32x2 one = {1.0f, 1.0f};
                    const f32x2 ar = {acc[m][0][r], acc[m][0][r + 1]}, az = {acc[m][1][r], acc[m][1][r + 1]};
                    f32x2 anh = {acc[m][2][r], acc[m][2][r + 1]};
                    const f32x2 anx = KX ? f32x2{acc[m][NA - 1][r], acc[m][NA - 1][r + 1]} : f32x2{xn[m][qd][e], xn[m][qd][e + 1]};
                    const f32x2 rgate = rcp(one + ex2(fma2(ar, dup(-L2E), dup(cr))));
                    const f32x2 zgate = rcp(one + ex2(fma2(az, dup(-L2E), dup(cz))));
                    if (KX) anh += dup(bn);
                    const f32x2 narg = fma2(rgate, anh, anx);
                    const f32x2 ngate = fma2(dup(-2.0f), rcp(one + ex2(fma2(narg, dup(2.0f * L2E), dup(cn)))), one);
                    const f32x2 hold = {h_get(m, r), h_get(m, r + 1)};
                    const f32x2 hv2 = fma2(zgate, hold - ngate, ngate);
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const float hv = k ? hv2.y : hv2.x;
                        h_set(m, r + k, hv);
                        h2_store16(hl_dst + (32 * m + ((r + k) & 3) + 8 * ((r + k) >> 2)) * ROWD * 2, hv);
                    }
                }
                if (step + 1 < T) seed_chunk(m, qd, tn);
            }
        if (step + 1 < T) x_store();
        tick(3);
        lds_barrier();
        tick(4);
    }
#ifdef PA_GRU_PHASE_TIMING
    if (stamp) {
#pragma unroll
        for (int k = 0; k < 5; ++k) dbg[wave * 8 + k] = tsum[k];
    }
#endif

    {
        const int tl = dir ? 0 : T - 1;
        if (DENSE) {
            dense_partials();
            __syncthreads();
            dense_store(tl);
        } else {
#pragma unroll
            for (int j = 0; j < YC; ++j) yc_write(j, tl, yc_read(j));
        }
    }
    if (hn != nullptr) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                hn[(lb + 32 * m + (r & 3) + 8 * (r >> 2)) * ldhn + dir * H + col] = h_get(m, r);
    }
}


// ------------------------------------------------------------------------------------------------
// GRU step loop for SMALL calls (polish model, H = 128; the reference's default batch is 128 chunks,
// pepper/modules/python/models/predict_distributed_gpu.py:40-47).  A call of n chunks runs 19 windows x
// (100 + 100) DEPENDENT steps whatever n is, so a small call is bound by the latency of one step, not by
// throughput: gru_rec_h2_kernel's 128-row workgroup takes ~14 us per step (8 waves x 36-72 MFMAs of 32
// cycles, a 200-600 KB weight stream from L2 and a gate phase of 32 elements per lane) with two of 256 CUs
// busy.  Here ONE workgroup of four waves owns 16 batch rows of one direction:
//   * wave u owns hidden units [32u, 32u + 32) of all three gates, so r, z and n of an element meet in one lane and
//     nothing but h_t itself crosses waves;
//   * the wave's recurrent weights -- 3 gates x 32 units x K = 128, (hi, lo) halves -- stay in REGISTERS for the
//     whole loop (192 VGPRs as B fragments of v_mfma_f32_16x16x32_f16): no weight stream at all;
//   * h lives in LDS in h2 form in TWO row images (step parity): the gate phase writes h_t into one while slower
//     waves may still read h_{t-1} from the other, so a step has ONE workgroup barrier;
//   * the input projections (+ b_ih, + b_hr / b_hz) come precomputed in Xp (one GEMM over all T steps on the whole
//     chip in front of the loop), fetched one step ahead; the layer output is the LDS image copied out 16 bytes
//     per lane while the next step contracts.
// Per step and wave: 72 MFMAs of 16 cycles + a gate phase of 8 elements per lane.  n chunks = 2 * ceil(n / 16)
// workgroups: the reference's 128-chunk batch spreads over 16 CUs instead of 2, 2048 chunks fill the chip.
//   /root/reference/pepper/modules/python/models/simple_model.py:30,32
template <int H>
__global__ __launch_bounds__(256, 1) void gru_small_h2_kernel(const float* __restrict__ Xp, int ldx,
                                                              const uint32_t* __restrict__ Wp,
                                                              const float* __restrict__ bhn,
                                                              const float* __restrict__ h0, int ldh0,
                                                              float* __restrict__ hn, int ldhn,
                                                              uint32_t* __restrict__ Y, int ldy, int B, int T) {
    static_assert(H == 128, "four waves x 32 units");
    typedef float f32x4v __attribute__((ext_vector_type(4)));
    constexpr int RT = 16, NT = H / 32, KS = H / 32;            // rows per workgroup, unit tiles = waves, k steps of 32
    constexpr int ROWB = H * 4 + 16, ROWD = ROWB / 4;
    constexpr float L2E = 1.4426950408889634f;
    __shared__ __attribute__((aligned(16))) uint32_t lds[2 * RT * ROWD];      // two h2 row images (step parity)
    const int dir = blockIdx.x & 1, btile = blockIdx.x >> 1;
    const int b0 = btile * RT;
    if (b0 >= B) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int u = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c16 = lane & 15, q4 = lane >> 4;

    // ---- the wave's weights: [dir][u][gate 3][sub 2][k step KS][hi, lo][64 lanes][16 B] ----
    h8 w[3][2][KS][2];
    {
        const uint32_t* wsrc = Wp + ((size_t)(dir * NT + u) * (3 * 2 * KS * 2)) * 256 + lane * 4;
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                for (int s = 0; s < KS; ++s)
#pragma unroll
                    for (int hl = 0; hl < 2; ++hl)
                        w[g][sb][s][hl] = *reinterpret_cast<const h8*>(wsrc + (size_t)(((g * 2 + sb) * KS + s) * 2 + hl) * 256);
    }
    // ---- h0 -> registers (exact f32, for the z * h term and the final state) and the first LDS image ----
    float hreg[2][4];
#pragma unroll
    for (int sb = 0; sb < 2; ++sb)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int row = 4 * q4 + e, col = u * 32 + sb * 16 + c16;
            const float hv = h0 != nullptr ? h0[(size_t)(b0 + row) * ldh0 + dir * H + col] : 0.0f;
            hreg[sb][e] = hv;
            h2_store16(reinterpret_cast<unsigned short*>(lds + row * ROWD + (col >> 3) * 8 + ((col & 7) >> 1)) + (col & 1), hv);
        }
    float bn[2];
#pragma unroll
    for (int sb = 0; sb < 2; ++sb) bn[sb] = bhn[dir * H + u * 32 + sb * 16 + c16];

    // Xp: C fragments of the projection GEMM's 32 x 32 tiles, [32-row batch tile][t][column tile][qd 4][lane 64][4 floats]
    // (lane l of such a tile: column l & 31, rows 8 qd + 4 (l >> 5) + e); this workgroup is half hh of its batch tile, and
    // a 16 x 16 accumulator lane (column c16, rows 4 q4 + e) finds its four values in ONE 16-byte slot
    const int hh = (b0 >> 4) & 1;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(Xp + (size_t)(b0 >> 5) * T * (ldx >> 5) * 1024), 0, 0x7fffffff, 0x00020000);
    const unsigned xlane = (unsigned)(((2 * hh + (q4 >> 1)) * 64 + (q4 & 1) * 32 + c16) * 16);
    auto load_xp = [&](int t, f32x4v (&dst)[3][2]) {
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            const unsigned tile = ((unsigned)t * (unsigned)(ldx >> 5) + (unsigned)(dir * 3 * NT + g * NT + u)) * 4096u;
#pragma unroll
            for (int sb = 0; sb < 2; ++sb)
                dst[g][sb] = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(xrs, xlane + sb * 256u, tile, 0));
        }
    };
    // y copy: 16 rows x 32 chunks of 16 bytes, two per thread
    const int yrow = tid >> 5, ychunk = tid & 31;
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(Y + (size_t)b0 * T * ldy + dir * H, 0, 0x7fffffff, 0x00020000);
    auto y_copy = [&](const uint32_t* img, int t) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int row = yrow + 8 * k;
            const u32x4 v = *reinterpret_cast<const u32x4*>(img + row * ROWD + ychunk * 4);
            __builtin_amdgcn_raw_buffer_store_b128(v, yrs, (unsigned)((row * T) * ldy + ychunk * 4) * 4u, (unsigned)t * (unsigned)ldy * 4u, 0);
        }
    };

    // Xp two steps ahead in two register sets (steps of even / odd parity): the loads a step issues are consumed two steps on,
    // so no wait in a step refers to what that step itself asked for
    f32x4v xa[3][2], xb[3][2];
    load_xp(dir ? T - 1 : 0, xa);
    if (T > 1) load_xp(dir ? T - 2 : 1, xb);
    __syncthreads();
    auto do_step = [&](int step, f32x4v (&mine)[3][2]) {
        const int t = dir ? T - 1 - step : step;
        const uint32_t* cur = lds + (step & 1) * (RT * ROWD);
        uint32_t* nxt = lds + ((step & 1) ^ 1) * (RT * ROWD);
        // the layer output of the step before: its LDS image is `cur` (published by the barrier that ended that step)
        if (step > 0) y_copy(cur, dir ? t + 1 : t - 1);
        f32x4v ar[2], az[2], anh[2], anx[2];
#pragma unroll
        for (int sb = 0; sb < 2; ++sb) {
            ar[sb] = mine[0][sb];
            az[sb] = mine[1][sb];
            anx[sb] = mine[2][sb];
            anh[sb] = f32x4v{bn[sb], bn[sb], bn[sb], bn[sb]};
        }
        if (step + 2 < T) load_xp(dir ? t - 2 : t + 2, mine);
        // ---------------- MFMA phase: [r z nh] += h_{t-1} W_hh^T, three-term split product ----------------
        const uint32_t* arow = cur + c16 * ROWD + q4 * 8;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const h8 a_hi = *reinterpret_cast<const h8*>(arow + s * 32);
            const h8 a_lo = *reinterpret_cast<const h8*>(arow + s * 32 + 4);
#pragma unroll
            for (int sb = 0; sb < 2; ++sb) {
                ar[sb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_lo, w[0][sb][s][0], ar[sb], 0, 0, 0);
                az[sb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_lo, w[1][sb][s][0], az[sb], 0, 0, 0);
                anh[sb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_lo, w[2][sb][s][0], anh[sb], 0, 0, 0);
            }
#pragma unroll
            for (int sb = 0; sb < 2; ++sb) {
                ar[sb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi, w[0][sb][s][1], ar[sb], 0, 0, 0);
                az[sb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi, w[1][sb][s][1], az[sb], 0, 0, 0);
                anh[sb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi, w[2][sb][s][1], anh[sb], 0, 0, 0);
            }
#pragma unroll
            for (int sb = 0; sb < 2; ++sb) {
                ar[sb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi, w[0][sb][s][0], ar[sb], 0, 0, 0);
                az[sb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi, w[1][sb][s][0], az[sb], 0, 0, 0);
                anh[sb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_hi, w[2][sb][s][0], anh[sb], 0, 0, 0);
            }
        }
        // ---------------- gate phase: the lane's eight elements; h' = n + z (h - n) ----------------
#pragma unroll
        for (int sb = 0; sb < 2; ++sb)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float rg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-L2E * ar[sb][e]));
                const float zg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-L2E * az[sb][e]));
                const float narg = __builtin_fmaf(rg, anh[sb][e], anx[sb][e]);
                const float ng = __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.0f * L2E * narg)), 1.0f);
                const float hv = __builtin_fmaf(zg, hreg[sb][e] - ng, ng);
                hreg[sb][e] = hv;
                const int row = 4 * q4 + e, col = u * 32 + sb * 16 + c16;
                h2_store16(reinterpret_cast<unsigned short*>(nxt + row * ROWD + (col >> 3) * 8 + ((col & 7) >> 1)) + (col & 1), hv);
            }
        lds_barrier();                                      // h_t complete in `nxt`; every wave is past its reads of `cur`
    };
    for (int step = 0; step < T; step += 2) {
        do_step(step, xa);
        if (step + 1 < T) do_step(step + 1, xb);
    }
    y_copy(lds + (T & 1) * (RT * ROWD), dir ? 0 : T - 1);
    if (hn != nullptr) {
#pragma unroll
        for (int sb = 0; sb < 2; ++sb)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                hn[(size_t)(b0 + 4 * q4 + e) * ldhn + dir * H + u * 32 + sb * 16 + c16] = hreg[sb][e];
    }
}

// ------------------------------------------------------------------------------------------------
// GRU decoder layer with its input projection fused (polish model): the layer input x_t is the previous
// layer's output, already an h2 tensor [B*T, KX] with KX = 2H, so the step contracts [h_{t-1} | x_t]
// (K = H + KX = 384) against [W_hh | W_ih] and neither the projection GEMM nor its 2.5 GB-per-window Xp
// round trip exist.  What makes it fit: 64 batch rows per workgroup (LDS rows of 1552 B = 97 KB), four
// waves = one per SIMD with both row tiles each (so every weight fragment is fetched once per workgroup:
// 590 KB per step, ~43 B/clk/CU), a 3-deep fragment ring instead of a second wave to cover L2 latency,
// and the next step's x slab (64 KB) prefetched into registers during the MFMA phase.
//   /root/reference/pepper/modules/python/models/simple_model.py:32
template <int H, int KX>
__global__ __launch_bounds__(256, 1) void gru_dec_h2_kernel(const uint32_t* __restrict__ Xh, int ldxh,
                                                            const float* __restrict__ bias,
                                                            const uint32_t* __restrict__ Wp,
                                                            const float* __restrict__ bhn,
                                                            const float* __restrict__ h0, int ldh0,
                                                            float* __restrict__ hn, int ldhn,
                                                            uint32_t* __restrict__ Y, int ldy, int B, int T) {
    constexpr int KT = H + KX, KS = KT / 16, KSH = H / 16, NT = H / 32, MTG = MT, NTHR = NT * 64;
    constexpr int ROWB = KT * 4 + 16, ROWD = ROWB / 4;
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];   // [MTG][ROWD] h2 rows of [h | x]
    static_assert((ROWB / 16) % 2 == 1, "row stride must be an odd number of 16-byte slots");
    static_assert(NT == 4, "one wave per SIMD");

    int dir, btile;
    decode_block(blockIdx.x, dir, btile);
    const int b0 = btile * MTG;
    if (b0 >= B) return;

    const int tid = threadIdx.x, lane = tid & 63;
    const int u = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, hf = lane >> 5;
    const int col = u * 32 + li;

    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint32_t*>(Wp + ((size_t)dir * (3 * NT) + u) * KS * 512), 0, 0x7fffffff, 0x00020000);
    const unsigned woff = lane * 16u;
    const size_t lb = (size_t)(b0 + 4 * hf);
    const bool odd = li & 1;
    uint32_t* hl_dst = lds + 4 * hf * ROWD + (col >> 3) * 8 + (odd ? 4 : 0) + ((col & 7) >> 1);
    const uint32_t* arow = lds + li * ROWD + hf * 8;

    for (int idx = tid; idx < MTG * ROWD; idx += NTHR) lds[idx] = 0u;
    __syncthreads();

    const unsigned h2sel = h2_select(odd);
    auto h2_word = [&](float hv) { return h2_word_of(hv, h2sel); };

    f32x16 hreg[2], acc[2][4];     // r, z, n(hidden half), n(input half)
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int dr = 32 * m + (r & 3) + 8 * (r >> 2);
            const float hv = h0 != nullptr ? h0[(lb + dr) * ldh0 + dir * H + col] : 0.0f;
            hreg[m][r] = hv;
            hl_dst[dr * ROWD] = h2_word(hv);
        }
    const float bn = bhn[dir * H + col];
    const float b_r = bias[dir * 3 * H + col], b_z = bias[dir * 3 * H + H + col], b_nx = bias[dir * 3 * H + 2 * H + col];
    auto seed_chunk = [&](int m, int qd) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            acc[m][0][4 * qd + e] = b_r;
            acc[m][1][4 * qd + e] = b_z;
            acc[m][2][4 * qd + e] = bn;
            acc[m][3][4 * qd + e] = b_nx;
        }
    };

    // x slab of one step: MTG rows x KX*4 bytes, 16 bytes per thread per pass
    constexpr int XCPR = KX / 4;                       // 16-byte chunks per row (64)
    constexpr int XROWS = NTHR / XCPR;                 // rows per pass (4)
    constexpr int XN = MTG / XROWS;                    // passes (16)
    const int xr = tid / XCPR, xc = tid % XCPR;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint32_t*>(Xh + (size_t)b0 * T * ldxh), 0, 0x7fffffff, 0x00020000);
    const unsigned x_off = ((unsigned)(xr * T) * ldxh + xc * 4) * 4u;
    uint32_t* x_dst = lds + xr * ROWD + H + xc * 4;
    u32x4 xq[XN];
    auto x_load_one = [&](int j, int t) {
        // rows beyond B read the workspace padding (finite garbage in rows nobody reads back)
        xq[j] = __builtin_amdgcn_raw_buffer_load_b128(xrs, x_off, ((unsigned)(j * XROWS * T + t) * ldxh) * 4u, 0);
    };
    auto x_store = [&]() {
#pragma unroll
        for (int j = 0; j < XN; ++j) *reinterpret_cast<u32x4*>(x_dst + j * XROWS * ROWD) = xq[j];
    };
    {
        const int t0 = dir ? T - 1 : 0;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) seed_chunk(m, qd);
#pragma unroll
        for (int j = 0; j < XN; ++j) x_load_one(j, t0);
        x_store();
    }
    __syncthreads();

    constexpr int CPR = H / 4, YROWS = NTHR / CPR, YC = MTG / YROWS;   // 32 chunks/row, 8 rows/pass, 8 passes
    const int yc_row = tid / CPR, yc_c = tid % CPR;
    const uint32_t* yc_src = lds + yc_row * ROWD + yc_c * 4;
    const __amdgpu_buffer_rsrc_t ycrs =
        __builtin_amdgcn_make_buffer_rsrc(Y + (size_t)b0 * T * ldy + dir * H, 0, 0x7fffffff, 0x00020000);
    const unsigned yc_off = ((unsigned)(yc_row * T) * ldy + yc_c * 4) * 4u;
    auto yc_read = [&](int j) { return *reinterpret_cast<const u32x4*>(yc_src + j * YROWS * ROWD); };
    auto yc_write = [&](int j, int tp, u32x4 v) {
        __builtin_amdgcn_raw_buffer_store_b128(v, ycrs, yc_off, ((unsigned)(j * YROWS * T + tp) * ldy) * 4u, 0);
    };

    struct Frag { h8 b[3][2], a[2][2]; };
    auto load_step = [&](int s, Frag& fr) {
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int hl = 0; hl < 2; ++hl)
                fr.b[g][hl] = __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(
                                                         wrs, woff, (unsigned)((g * NT * KS + s) * 2 + hl) * 1024u, 0));
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            fr.a[m][0] = *reinterpret_cast<const h8*>(arow + m * 32 * ROWD + s * 16);
            fr.a[m][1] = *reinterpret_cast<const h8*>(arow + m * 32 * ROWD + s * 16 + 4);
        }
    };

    for (int step = 0; step < T; ++step) {
        const int t = dir ? T - 1 - step : step;
        const int tp = step > 0 ? (dir ? t + 1 : t - 1) : t;
        const int tn = (step + 1 < T) ? (dir ? t - 1 : t + 1) : t;     // last step re-reads its own slab (unused)
        // ---------------- MFMA phase ----------------
        {
            Frag ring[3];
            load_step(0, ring[0]);
            load_step(1, ring[1]);
            u32x4 ycv = {0, 0, 0, 0};
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int p = s % 3;
                if (s + 2 < KS) load_step(s + 2, ring[(p + 2) % 3]);
                if (s < XN) x_load_one(s, tn);
                if (s >= 1 && s <= YC) yc_write(s - 1, tp, ycv);
                if (s < YC) ycv = yc_read(s);
#pragma unroll
                for (int term = 0; term < 3; ++term)
#pragma unroll
                    for (int g = 0; g < 3; ++g)
#pragma unroll
                        for (int m = 0; m < 2; ++m) {
                            const int ai = (g == 2 && s >= KSH) ? 3 : g;
                            acc[m][ai] = mfma_h(ring[p].a[m][term == 0 ? 1 : 0], ring[p].b[g][term == 1 ? 1 : 0], acc[m][ai]);
                        }
                if (s + 2 < KS) {
#pragma unroll
                    for (int q = 0; q < 6; ++q) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // B fragment
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // A fragment
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);       // x slab prefetch
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);       // y copy store
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);       // y copy read
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        lds_barrier();

        // ---------------- gate phase ----------------
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * qd + e;
                    const int dr = 32 * m + (r & 3) + 8 * (r >> 2);
                    const float rgate = fast_sigmoid(acc[m][0][r]);
                    const float zgate = fast_sigmoid(acc[m][1][r]);
                    const float ngate = fast_tanh(acc[m][3][r] + rgate * acc[m][2][r]);
                    const float hv = (1.0f - zgate) * ngate + zgate * hreg[m][r];
                    hreg[m][r] = hv;
                    hl_dst[dr * ROWD] = h2_word(hv);
                }
                seed_chunk(m, qd);
            }
        x_store();
        lds_barrier();
    }

    {
        const int tl = dir ? 0 : T - 1;
#pragma unroll
        for (int j = 0; j < YC; ++j) yc_write(j, tl, yc_read(j));
    }
    if (hn != nullptr) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                hn[(lb + 32 * m + (r & 3) + 8 * (r >> 2)) * ldhn + dir * H + col] = hreg[m][r];
    }
}

// PA_NT=1: nt cache policy on the once-through streams of the step loops (see SAUX)
inline bool stream_nt() {
    static const bool on = [] { const char* e = getenv("PA_NT"); return e ? e[0] != '0' : PA_NT_DEFAULT; }();
    return on;
}

// PA_BIAS_COLUMN=0: the fused first layers take their biases in the gate phase instead of from the bias column
inline bool bias_column() {
    static const bool on = [] { const char* e = getenv("PA_BIAS_COLUMN"); return !e || e[0] != '0'; }();
    return on;
}

inline int rec_grid(int B) {
    const int nbt = (B + MT - 1) / MT;
    return 2 * ((nbt + 3) / 4) * 4;
}

}  // namespace

namespace pa {

unsigned long long* debug_buffer();   // rnn.hip (PA_DEBUG_TIMING=1)

// PA_DEBUG_TIMING=1: [2 kernels][8 waves][8] u64 phase sums of the GRU step loops (decoder+head first, fused encoder
// second), read by pa_debug_dump_gru_timing (tools/phase_timing_gru.py); debug aid, not in the ABI headers
unsigned long long* g_dbg_gru = nullptr;
unsigned long long* gru_debug_buffer() {
    static const bool on = [] { const char* e = getenv("PA_DEBUG_TIMING"); return e && e[0] == '1'; }();
    if (!on) return nullptr;
    if (!g_dbg_gru) {
        if (hipMalloc(&g_dbg_gru, 128 * sizeof(unsigned long long)) != hipSuccess) return nullptr;
        (void)hipMemset(g_dbg_gru, 0, 128 * sizeof(unsigned long long));
    }
    return g_dbg_gru;
}

// W [G*H, K] per direction (K = H, or H + KX with [W_hh | W_ih | 0]) -> per-lane h2 fragments
// [dir][G*H/32][K/16][hi, lo][64 lanes][8 halves]; lane l of tile nt, step s holds
// W[nt*32 + (l&31)][16 s + 8 (l>>5) + e], e = 0..7.
// bias != nullptr (fused first layers with F < KX): column H + F of the packed matrix holds bias[d][n]; the step loop
// feeds a constant 1.0 in that input column, so the matrix pipe adds the bias and the accumulators start from zero.
void pack_rec_weights_h2(const float* const whh[2], const float* const wih[2], int G, int H, int F, int KX,
                         uint32_t* out, const float* const* bias) {
    const int KT = H + KX, NTt = G * H / 32, KS = KT / 16;
    _Float16* o = reinterpret_cast<_Float16*>(out);
    for (int d = 0; d < 2; ++d)
        for (int nt = 0; nt < NTt; ++nt)
            for (int s = 0; s < KS; ++s)
                for (int l = 0; l < 64; ++l)
                    for (int e = 0; e < 8; ++e) {
                        const int n = nt * 32 + (l & 31), k = 16 * s + 8 * (l >> 5) + e;
                        float v = 0.0f;
                        if (k < H) v = whh[d][(size_t)n * H + k];
                        else if (wih[0] != nullptr && k - H < F) v = wih[d][(size_t)n * F + (k - H)];
                        else if (bias != nullptr && k - H == F && F < KX) v = bias[d][n];
                        const _Float16 hi = (_Float16)v;
                        const size_t base = ((((size_t)d * NTt + nt) * KS + s) * 2) * 512 + (size_t)l * 8 + e;
                        o[base] = hi;
                        o[base + 512] = (_Float16)(v - (float)hi);
                    }
}

// Padded input width of the uint8-input GRU step loop for F features (0: not available, the projection runs as a GEMM)
int gru_fused_input_kx(int H, int F) {
    if (H != 128 || F <= 0) return 0;
    if (F <= 16) return 16;
    if (F <= 128 && (F & 3) == 0) return 128;
    return 0;
}

size_t rec_weights_h2_words(int G, int H, int KX) { return (size_t)2 * (G * H / 32) * ((H + KX) / 16) * 2 * 256; }

hipError_t launch_lstm_dec_h2(int H, const void* Xh, int ldxh, const float* bias, const void* Wp, void* Y, int ldy, int B,
                              int T, hipStream_t stream, bool prescaled) {
    if (B <= 0) return hipSuccess;
    if (H != 256 || (ldy & 7) || (ldxh & 7) || ldxh < 512) return hipErrorInvalidValue;
    const size_t lds = (size_t)MT * (256 * 4 + 16) + (size_t)8 * 2 * 16 * 64 * 4 + (size_t)2 * MT * 36 * 4;   // h + c + x ring
    const int grid = rec_grid(B);
#define PA_DEC(PRE_, AUX_)                                                                                             \
    hipLaunchKernelGGL((lstm_rec_h2_kernel<256, 512, PRE_, true, AUX_>), dim3(grid), dim3(512), lds, stream, (const float*)nullptr, 0, \
                       (const int8_t*)nullptr, 0, bias, static_cast<const uint32_t*>(Wp), static_cast<uint32_t*>(Y), ldy, B, T, \
                       debug_buffer(), static_cast<const uint32_t*>(Xh), ldxh)
    if (stream_nt()) { if (prescaled) PA_DEC(true, 2); else PA_DEC(false, 2); }
    else { if (prescaled) PA_DEC(true, 0); else PA_DEC(false, 0); }
#undef PA_DEC
    return hipGetLastError();
}

// Unit-split step loop (lstm_rec_h2_split_kernel): B <= 1024, prescaled weights, Xp from the projection GEMM.
// exch: 2 * tiles4 groups x 64 KB; counters: 2 * tiles4 x 128 B (zeroed here); failed: one int the kernel sets when a group
// did not meet.  tiles4 = row tiles of 32 rounded up to a multiple of four.  Eight members of 32 units up to 512 windows,
// four of 64 units above: at most 256 workgroups either way.
size_t lstm_split_exchange_bytes(int B) { return (size_t)2 * (((B + 31) / 32 + 3) / 4 * 4) * 2 * 32 * 256 * 4; }
size_t lstm_split_counter_bytes(int B) { return (size_t)2 * (((B + 31) / 32 + 3) / 4 * 4) * 128; }

// Workgroups of the split step loop the current device holds at once: the members of a group spin on each other, so a launch
// must fit the chip as a whole (what one CU takes of the kernel x the CUs the device reports -- 256 on a whole MI355X, 32 on
// one of its CPX partitions).  ntw as in the kernel (1: eight members, 2: four).
int lstm_split_resident_workgroups(int ntw) {
    static int cached[2] = {-1, -1};
    int& c = cached[ntw == 2 ? 1 : 0];
    if (c >= 0) return c;
    int dev = 0, per_cu = 0;
    hipDeviceProp_t prop;
    const size_t lds = (size_t)32 * (256 * 4 + 16) + (size_t)4 * ntw * 16 * 64 * 4 + 16;
    hipError_t e = hipGetDevice(&dev);
    if (e == hipSuccess) e = hipGetDeviceProperties(&prop, dev);
    if (e == hipSuccess)
        e = ntw == 2 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, lstm_rec_h2_split_kernel<256, 2>, 256, lds)
                     : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, lstm_rec_h2_split_kernel<256, 1>, 256, lds);
    c = e == hipSuccess ? per_cu * prop.multiProcessorCount : 0;
    return c;
}

int lstm_split_grid(int B) {
    const int tiles4 = ((B + 31) / 32 + 3) / 4 * 4;
    return 8 * (8 / (B <= 512 ? 1 : 2)) * (tiles4 / 4);
}

hipError_t launch_lstm_rec_h2_split(int H, const float* Xp, int ldx, const void* Wp, void* Y, int ldy, int B, int T, void* exch,
                                    void* counters, int* failed, hipStream_t stream, int sabotage) {
    if (B <= 0) return hipSuccess;
    const int tiles4 = ((B + 31) / 32 + 3) / 4 * 4;
    const int ntw = B <= 512 ? 1 : 2;
    const int grid = 8 * (8 / ntw) * (tiles4 / 4);
    if (H != 256 || (ldy & 7) || grid > 256 || !exch || !counters || !failed) return hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(counters, 0, lstm_split_counter_bytes(B), stream);
    if (e != hipSuccess) return e;
    const size_t lds = (size_t)32 * (256 * 4 + 16) + (size_t)4 * ntw * 16 * 64 * 4 + 16;
    if (ntw == 1)
        hipLaunchKernelGGL((lstm_rec_h2_split_kernel<256, 1>), dim3(grid), dim3(256), lds, stream, Xp, ldx, static_cast<const uint32_t*>(Wp),
                           static_cast<uint32_t*>(Y), ldy, B, T, static_cast<uint32_t*>(exch), static_cast<unsigned*>(counters), failed,
                           tiles4, sabotage);
    else
        hipLaunchKernelGGL((lstm_rec_h2_split_kernel<256, 2>), dim3(grid), dim3(256), lds, stream, Xp, ldx, static_cast<const uint32_t*>(Wp),
                           static_cast<uint32_t*>(Y), ldy, B, T, static_cast<uint32_t*>(exch), static_cast<unsigned*>(counters), failed,
                           tiles4, sabotage);
    return hipGetLastError();
}

hipError_t launch_lstm_rec_h2(int H, const float* Xp, int ldx, const int8_t* X, int F, const float* bias,
                              const void* Wp, void* Y, int ldy, int B, int T, hipStream_t stream, bool prescaled, bool small) {
    if (B <= 0) return hipSuccess;
    if (H != 256 || (ldy & 7)) return hipErrorInvalidValue;
    const int grid = rec_grid(B);
    if (small && prescaled) {
        // 32-row workgroups for small calls (see the kernel's MTILES): twice the workgroups, half the time per step
        const int nbt = (B + 31) / 32, grid1 = 2 * ((nbt + 3) / 4) * 4;
        if (X != nullptr && F > 0 && F < 32 && stream_nt() && bias_column()) {
            const size_t lds = (size_t)32 * ((256 + 32) * 4 + 16) + (size_t)8 * 16 * 64 * 4;
            hipLaunchKernelGGL((lstm_rec_h2_kernel<256, 32, true, false, 2, true, 1>), dim3(grid1), dim3(512), lds, stream,
                               (const float*)nullptr, 0, X, F, bias, static_cast<const uint32_t*>(Wp), static_cast<uint32_t*>(Y),
                               ldy, B, T, (unsigned long long*)nullptr);
            return hipGetLastError();
        }
        if (X == nullptr) {
            const size_t lds = (size_t)32 * (256 * 4 + 16) + (size_t)8 * 16 * 64 * 4;
            hipLaunchKernelGGL((lstm_rec_h2_kernel<256, 0, true, false, 0, false, 1>), dim3(grid1), dim3(512), lds, stream, Xp, ldx,
                               (const int8_t*)nullptr, 0, (const float*)nullptr, static_cast<const uint32_t*>(Wp),
                               static_cast<uint32_t*>(Y), ldy, B, T, (unsigned long long*)nullptr);
            return hipGetLastError();
        }
    }
    if (X != nullptr) {
        if (F <= 0 || F > 32) return hipErrorInvalidValue;
        const size_t lds = (size_t)MT * ((256 + 32) * 4 + 16) + (size_t)8 * 2 * 16 * 64 * 4;
        if (prescaled && stream_nt() && F < 32 && bias_column())
            hipLaunchKernelGGL((lstm_rec_h2_kernel<256, 32, true, false, 2, true>), dim3(grid), dim3(512), lds, stream, (const float*)nullptr, 0, X,
                               F, bias, static_cast<const uint32_t*>(Wp), static_cast<uint32_t*>(Y), ldy, B, T,
                               debug_buffer() ? debug_buffer() + 8 * 80 * 2 : nullptr);
        else if (prescaled && stream_nt())
            hipLaunchKernelGGL((lstm_rec_h2_kernel<256, 32, true, false, 2>), dim3(grid), dim3(512), lds, stream, (const float*)nullptr, 0, X,
                               F, bias, static_cast<const uint32_t*>(Wp), static_cast<uint32_t*>(Y), ldy, B, T,
                               debug_buffer() ? debug_buffer() + 8 * 80 * 2 : nullptr);
        else if (prescaled) hipLaunchKernelGGL((lstm_rec_h2_kernel<256, 32, true>), dim3(grid), dim3(512), lds, stream, (const float*)nullptr, 0, X,
                           F, bias, static_cast<const uint32_t*>(Wp), static_cast<uint32_t*>(Y), ldy, B, T,
                           debug_buffer() ? debug_buffer() + 8 * 80 * 2 : nullptr);
        else hipLaunchKernelGGL((lstm_rec_h2_kernel<256, 32, false>), dim3(grid), dim3(512), lds, stream, (const float*)nullptr, 0, X,
                           F, bias, static_cast<const uint32_t*>(Wp), static_cast<uint32_t*>(Y), ldy, B, T,
                           debug_buffer() ? debug_buffer() + 8 * 80 * 2 : nullptr);
    } else {
        const size_t lds = (size_t)MT * (256 * 4 + 16) + (size_t)8 * 2 * 16 * 64 * 4;
        if (prescaled) hipLaunchKernelGGL((lstm_rec_h2_kernel<256, 0, true>), dim3(grid), dim3(512), lds, stream, Xp, ldx,
                           (const int8_t*)nullptr, 0, (const float*)nullptr, static_cast<const uint32_t*>(Wp),
                           static_cast<uint32_t*>(Y), ldy, B, T, debug_buffer());
        else hipLaunchKernelGGL((lstm_rec_h2_kernel<256, 0, false>), dim3(grid), dim3(512), lds, stream, Xp, ldx,
                           (const int8_t*)nullptr, 0, (const float*)nullptr, static_cast<const uint32_t*>(Wp),
                           static_cast<uint32_t*>(Y), ldy, B, T, debug_buffer());
    }
    return hipGetLastError();
}

hipError_t launch_gru_rec_h2(int H, const float* Xp, int ldx, const uint8_t* X, int F, int64_t x_bstride,
                             const float* bias, const void* Wp, const float* bhn, const float* h0, int ldh0, float* hn,
                             int ldhn, void* Y, int ldy, int B, int T, hipStream_t stream) {
    if (B <= 0) return hipSuccess;
    if (H != 128 || (ldy & 7)) return hipErrorInvalidValue;
    const int nbt = (B + 2 * MT - 1) / (2 * MT);
    const int grid = 2 * ((nbt + 3) / 4) * 4;
    if (X != nullptr) {
        const int KX = gru_fused_input_kx(H, F);
        if (KX == 16) {
            const size_t lds = (size_t)2 * MT * ((128 + 16) * 4 + 16);
            if (stream_nt() && F < 16 && bias_column())
                hipLaunchKernelGGL((gru_rec_h2_kernel<128, 16, false, 2, false, true>), dim3(grid), dim3(512), lds, stream, (const float*)nullptr, 0, X,
                                   F, x_bstride, bias, static_cast<const uint32_t*>(Wp), bhn, h0, ldh0, hn, ldhn,
                                   static_cast<uint32_t*>(Y), ldy, B, T, (const uint32_t*)nullptr, 0, (const uint32_t*)nullptr,
                                   (float*)nullptr, gru_debug_buffer() ? gru_debug_buffer() + 64 : nullptr);
            else if (stream_nt())
                hipLaunchKernelGGL((gru_rec_h2_kernel<128, 16, false, 2>), dim3(grid), dim3(512), lds, stream, (const float*)nullptr, 0, X,
                                   F, x_bstride, bias, static_cast<const uint32_t*>(Wp), bhn, h0, ldh0, hn, ldhn,
                                   static_cast<uint32_t*>(Y), ldy, B, T, (const uint32_t*)nullptr, 0, (const uint32_t*)nullptr,
                                   (float*)nullptr, gru_debug_buffer() ? gru_debug_buffer() + 64 : nullptr);
            else
                hipLaunchKernelGGL((gru_rec_h2_kernel<128, 16>), dim3(grid), dim3(512), lds, stream, (const float*)nullptr, 0, X,
                                   F, x_bstride, bias, static_cast<const uint32_t*>(Wp), bhn, h0, ldh0, hn, ldhn,
                                   static_cast<uint32_t*>(Y), ldy, B, T);
        } else if (KX == 128) {
            if ((x_bstride & 3) || (reinterpret_cast<uintptr_t>(X) & 3)) return hipErrorInvalidValue;
            const size_t lds = (size_t)2 * MT * ((128 + 128) * 4 + 16) + (size_t)8 * 15 * 64 * 4;   // + f32 h strip (160 KB in all)
            hipLaunchKernelGGL((gru_rec_h2_kernel<128, 128>), dim3(grid), dim3(512), lds, stream, (const float*)nullptr, 0, X,
                               F, x_bstride, bias, static_cast<const uint32_t*>(Wp), bhn, h0, ldh0, hn, ldhn,
                               static_cast<uint32_t*>(Y), ldy, B, T);
        } else {
            return hipErrorInvalidValue;
        }
    } else {
        const size_t lds = (size_t)2 * MT * (128 * 4 + 16);
        hipLaunchKernelGGL((gru_rec_h2_kernel<128, 0>), dim3(grid), dim3(512), lds, stream, Xp, ldx,
                           (const uint8_t*)nullptr, 0, (int64_t)0, (const float*)nullptr, static_cast<const uint32_t*>(Wp),
                           bhn, h0, ldh0, hn, ldhn, static_cast<uint32_t*>(Y), ldy, B, T);
    }
    return hipGetLastError();
}


// W_hh [3H, H] per direction -> B fragments of v_mfma_f32_16x16x32_f16 in h2 form for gru_small_h2_kernel:
// [dir][unit tile u H/32][gate 3][sub 2][k step H/32][hi, lo][64 lanes][8 halves]; lane l holds
// W[g H + 32 u + 16 sub + (l & 15)][32 s + 8 (l >> 4) + e].
void pack_gru_small_weights_h2(const float* const whh[2], int H, uint32_t* out) {
    const int NTt = H / 32, KS = H / 32;
    _Float16* o = reinterpret_cast<_Float16*>(out);
    for (int d = 0; d < 2; ++d)
        for (int u = 0; u < NTt; ++u)
            for (int g = 0; g < 3; ++g)
                for (int sb = 0; sb < 2; ++sb)
                    for (int s = 0; s < KS; ++s)
                        for (int l = 0; l < 64; ++l)
                            for (int e = 0; e < 8; ++e) {
                                const int n = g * H + 32 * u + 16 * sb + (l & 15), k = 32 * s + 8 * (l >> 4) + e;
                                const float v = whh[d][(size_t)n * H + k];
                                const _Float16 hi = (_Float16)v;
                                const size_t base = ((((((size_t)d * NTt + u) * 3 + g) * 2 + sb) * KS + s) * 2) * 512 + (size_t)l * 8 + e;
                                o[base] = hi;
                                o[base + 512] = (_Float16)(v - (float)hi);
                            }
}

size_t gru_small_weights_h2_words(int H) { return (size_t)2 * (H / 32) * 3 * 2 * (H / 32) * 2 * 256; }

// Rows of the workspace a call of B chunks must have behind Xp / Y / h0 / hn (whole 32-row projection tiles)
hipError_t launch_gru_small_h2(int H, const float* Xp, int ldx, const void* Wp, const float* bhn, const float* h0, int ldh0,
                               float* hn, int ldhn, void* Y, int ldy, int B, int T, hipStream_t stream) {
    if (B <= 0) return hipSuccess;
    if (H != 128 || (ldy & 7) || (ldx & 31) || !Xp || !Wp || !Y) return hipErrorInvalidValue;
    const int grid = 2 * ((B + 15) / 16);
    hipLaunchKernelGGL((gru_small_h2_kernel<128>), dim3(grid), dim3(256), 0, stream, Xp, ldx, static_cast<const uint32_t*>(Wp), bhn, h0,
                       ldh0, hn, ldhn, static_cast<uint32_t*>(Y), ldy, B, T);
    return hipGetLastError();
}

// dense1 [C, 2H] (C <= 5, H = 128) -> per-direction B fragments of v_mfma_f32_16x16x32_f16 in h2 form:
// [dir][k quarter 4][hi, lo][64 lanes][8 halves]; lane l holds W[j = l & 15][dir*H + 32 q + 8 (l >> 4) + e] (0 for j >= C).
void pack_dense_head_h2(const float* W, int C, int H, uint32_t* out) {
    _Float16* o = reinterpret_cast<_Float16*>(out);
    for (int d = 0; d < 2; ++d)
        for (int q = 0; q < H / 32; ++q)
            for (int l = 0; l < 64; ++l)
                for (int e = 0; e < 8; ++e) {
                    const int j = l & 15, k = d * H + 32 * q + 8 * (l >> 4) + e;
                    const float v = j < C ? W[(size_t)j * 2 * H + k] : 0.0f;
                    const _Float16 hi = (_Float16)v;
                    const size_t base = (((size_t)d * (H / 32) + q) * 2) * 512 + (size_t)l * 8 + e;
                    o[base] = hi;
                    o[base + 512] = (_Float16)(v - (float)hi);
                }
}

size_t dense_head_h2_words(int H) { return (size_t)2 * (H / 32) * 2 * 256; }
size_t dense_partials_floats(int B, int T) { return (size_t)2 * ((B + 2 * MT - 1) / (2 * MT)) * T * 5 * (2 * MT); }

hipError_t launch_gru_dec_h2_dense(int H, const void* Xh, int ldxh, const float* bias, const void* Wp, const float* bhn,
                                   const float* h0, int ldh0, float* hn, int ldhn, const void* Wd, float* P, int B, int T,
                                   hipStream_t stream) {
    if (B <= 0) return hipSuccess;
    if (H != 128 || (ldxh & 7) || !Wd || !P) return hipErrorInvalidValue;
    const int nbt = (B + 2 * MT - 1) / (2 * MT);
    const int grid = 2 * ((nbt + 3) / 4) * 4;
    const size_t lds = (size_t)2 * MT * (128 * 4 + 16) + (size_t)2 * 2 * MT * 36 * 4 + (size_t)4 * 2 * 1024 + (size_t)4 * 5 * 2 * MT * 4 +
                       (size_t)8 * 16 * 64 * 4;   // h rows + x ring + dense1 fragments + partial-logit scratch + f32 h strip
#define PA_GDD(AUX_)                                                                                                   \
    hipLaunchKernelGGL((gru_rec_h2_kernel<128, 256, true, AUX_, true>), dim3(grid), dim3(512), lds, stream, (const float*)nullptr, 0, \
                       (const uint8_t*)nullptr, 0, (int64_t)0, bias, static_cast<const uint32_t*>(Wp), bhn, h0, ldh0, hn, ldhn, \
                       (uint32_t*)nullptr, 8, B, T, static_cast<const uint32_t*>(Xh), ldxh, static_cast<const uint32_t*>(Wd), P, \
                       gru_debug_buffer())
    if (stream_nt()) PA_GDD(2); else PA_GDD(0);
#undef PA_GDD
    return hipGetLastError();
}

hipError_t launch_gru_dec_h2(int H, const void* Xh, int ldxh, const float* bias, const void* Wp, const float* bhn,
                             const float* h0, int ldh0, float* hn, int ldhn, void* Y, int ldy, int B, int T,
                             hipStream_t stream) {
    if (B <= 0) return hipSuccess;
    if (H != 128 || (ldy & 7) || (ldxh & 7)) return hipErrorInvalidValue;
    // default: the 8-wave / 128-row step loop with the x slab streamed through the LDS ring (half the weight
    // stream per row of gru_dec_h2_kernel's 64-row form, which PA_GRU_DEC_RING=0 selects)
    static const bool ring = [] { const char* e = getenv("PA_GRU_DEC_RING"); return !e || e[0] != '0'; }();
    if (ring) {
        const int nbt = (B + 2 * MT - 1) / (2 * MT);
        const int grid = 2 * ((nbt + 3) / 4) * 4;
        const size_t lds = (size_t)2 * MT * (128 * 4 + 16) + (size_t)2 * 2 * MT * 36 * 4 + (size_t)8 * 16 * 64 * 4;
        if (stream_nt())
            hipLaunchKernelGGL((gru_rec_h2_kernel<128, 256, true, 2>), dim3(grid), dim3(512), lds, stream, (const float*)nullptr, 0,
                               (const uint8_t*)nullptr, 0, (int64_t)0, bias, static_cast<const uint32_t*>(Wp), bhn, h0, ldh0, hn,
                               ldhn, static_cast<uint32_t*>(Y), ldy, B, T, static_cast<const uint32_t*>(Xh), ldxh);
        else
            hipLaunchKernelGGL((gru_rec_h2_kernel<128, 256, true>), dim3(grid), dim3(512), lds, stream, (const float*)nullptr, 0,
                               (const uint8_t*)nullptr, 0, (int64_t)0, bias, static_cast<const uint32_t*>(Wp), bhn, h0, ldh0, hn,
                               ldhn, static_cast<uint32_t*>(Y), ldy, B, T, static_cast<const uint32_t*>(Xh), ldxh);
        return hipGetLastError();
    }
    const size_t lds = (size_t)MT * ((128 + 256) * 4 + 16);
    hipLaunchKernelGGL((gru_dec_h2_kernel<128, 256>), dim3(rec_grid(B)), dim3(256), lds, stream,
                       static_cast<const uint32_t*>(Xh), ldxh, bias, static_cast<const uint32_t*>(Wp), bhn, h0, ldh0, hn,
                       ldhn, static_cast<uint32_t*>(Y), ldy, B, T);
    return hipGetLastError();
}

}  // namespace pa

extern "C" int pa_debug_dump_gru_timing(unsigned long long* host_out) {   // 128 values
    if (!pa::g_dbg_gru) return 1;
    return hipMemcpy(host_out, pa::g_dbg_gru, 128 * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess ? 0 : 2;
}
