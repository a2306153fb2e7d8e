// Persistent recurrent step-loop kernels (one launch runs all T dependent steps of one
// bidirectional layer): fused  h*W_hh^T  on v_mfma_f32_32x32x2_f32  +  gate nonlinearities  +
// state update, hidden state carried in LDS (as the next step's MFMA A operand) and the cell
// state in registers.
//
// Replaces the recurrent half of torch.nn.LSTM / torch.nn.GRU as called at
//   /root/reference/pepper_variant/modules/python/models/simple_model.py:51,54   (LSTM, H=256)
//   /root/reference/pepper/modules/python/models/simple_model.py:30,32           (GRU,  H=128)
// The input half (W_ih x + b) is a plain GEMM (gemm.hip) whose result Xp seeds the accumulators.
//
// Work decomposition: workgroup = (64 batch rows) x (one direction); wave u of H/32 owns hidden
// units [32u, 32u+32) for ALL gates, so i/f/g/o (or r/z/n) of one (row, unit) sit in the same
// lane and register index of four accumulators and the cell update needs no cross-lane traffic.
// Per step a wave issues (H/8) * 4 * G * 2 MFMAs (G = 4 or 3 gates, 2 row tiles).  W_hh is
// pre-packed in fragment order so each B-operand load is one coalesced 1 KiB global_load_dwordx4
// served from L2 (direction = f(XCD) keeps one direction's weights per XCD L2).
#include <cstdlib>

#include "common.h"
#include "kernels.h"

namespace {

constexpr int MT = 64;  // batch rows per workgroup (2 MFMA row tiles)

PA_DEV void decode_block(int bid, int& dir, int& btile) {
    // workgroup b is observed to run on XCD b % 8: XCDs 0-3 take the forward direction, 4-7 the
    // reverse one (speed only; correctness does not depend on placement).
    const int xcd = bid & 7, q = bid >> 3;
    dir = xcd >> 2;
    btile = q * 4 + (xcd & 3);
}

template <int H>
__global__ __launch_bounds__(H / 32 * 64, 2) void lstm_rec_kernel(const float* __restrict__ Xp, int ldx,
                                                               const float* __restrict__ Wp,
                                                               float* __restrict__ Y, int ldy, int B,
                                                               int T) {
    constexpr int LDH = H + 4, KB = H / 8, NT = H / 32;
    extern __shared__ __attribute__((aligned(16))) float hs[];  // [MT][LDH]

    int dir, btile;
    decode_block(blockIdx.x, dir, btile);
    const int b0 = btile * MT;
    if (b0 >= B) return;

    const int tid = threadIdx.x, lane = tid & 63, u = tid >> 6;
    const int li = lane & 31, hf = lane >> 5;

    for (int idx = tid; idx < MT * LDH; idx += blockDim.x) hs[idx] = 0.0f;

    f32x16 c[2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) c[m][r] = 0.0f;

    // Xp / Y are workspace buffers allocated for a batch padded to a multiple of MT rows, so the
    // tail tile needs no clamping: rows are independent and pad rows are never read back.
    // row(m, r) = b0 + 4*hf + 32*m + (r & 3) + 8*(r >> 2): per-lane base + wave-uniform deltas.
    const int col = u * 32 + li;
    const size_t lrow = (size_t)(b0 + 4 * hf) * T;
    const float* xl = Xp + lrow * ldx + dir * 4 * H + col;
    float* yl = Y + lrow * ldy + dir * H + col;
    float* hl = hs + 4 * hf * LDH + col;
    const f32x4* wp = reinterpret_cast<const f32x4*>(Wp) + (size_t)dir * (4 * NT) * KB * 64 + lane;
    __syncthreads();

    for (int step = 0; step < T; ++step) {
        const int t = dir ? T - 1 - step : step;
        f32x16 acc[2][4];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    acc[m][g][r] = xl[((size_t)(32 * m + (r & 3) + 8 * (r >> 2)) * T + t) * ldx + g * H];

        const float* hrow = hs + li * LDH + hf * 4;
#pragma unroll 2
        for (int kb = 0; kb < KB; ++kb) {
            f32x4 a[2], b[4];
#pragma unroll
            for (int m = 0; m < 2; ++m) a[m] = *reinterpret_cast<const f32x4*>(hrow + m * 32 * LDH + kb * 8);
#pragma unroll
            for (int g = 0; g < 4; ++g) b[g] = wp[((size_t)(g * NT + u) * KB + kb) * 64];
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int m = 0; m < 2; ++m) acc[m][g] = mfma32(a[m][s], b[g][s], acc[m][g]);
        }
        __syncthreads();  // every wave has finished reading h_{t-1}

#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float ig = sigmoid_f(acc[m][0][r]);
                const float fg = sigmoid_f(acc[m][1][r]);
                const float gg = tanhf(acc[m][2][r]);
                const float og = sigmoid_f(acc[m][3][r]);
                const float cn = fg * c[m][r] + ig * gg;
                c[m][r] = cn;
                const float hv = og * tanhf(cn);
                hl[(32 * m + (r & 3) + 8 * (r >> 2)) * LDH] = hv;
                yl[((size_t)(32 * m + (r & 3) + 8 * (r >> 2)) * T + t) * ldy] = hv;
            }
        __syncthreads();  // h_t visible to every wave
    }
}

// ------------------------------------------------------------------------------------------------
// Ping-pong LSTM step loop (the production variant kernel).
//
// The 64 batch rows of a workgroup are split into two independent 32-row halves; waves 0..NW/2-1
// own half 0, the rest half 1.  Time is cut into intervals separated by one workgroup barrier:
// in even intervals half 0 runs its MFMA phase (h_{t-1} W_hh^T) while half 1 runs its gate phase
// (sigmoid/tanh, cell update, h_t -> LDS + HBM, next step's Xp -> accumulators), in odd intervals
// the roles swap.  Every SIMD hosts one wave of each half, so the matrix pipe always has exactly
// one wave feeding it and the VALU/memory work of the gate phase is hidden behind it.
// A wave owns 64 hidden units (two 32-column tiles) x 4 gates for its 32 rows = 8 accumulators.
// W_hh fragments stream from L2 through a 3-deep register ring (two "quads" of 4 gate fragments
// in flight ahead of the one being consumed), so the pipe never waits for an L2 round trip.
// KX > 0 fuses the layer's input projection: the int8 summary row x_t (F <= KX features, zero
// padded) is converted to f32 in the gate phase and stored next to h_{t-1} in the same LDS row, so
// the MFMA phase contracts over K = H + KX against the concatenated [W_hh | W_ih] fragments and
// the accumulators start from the bias -- no Xp round trip through HBM for the first layer.
template <int H, int KX>
__global__ __launch_bounds__(H / 32 * 64, 2) void lstm_rec_pp_kernel(const float* __restrict__ Xp, int ldx,
                                                                     const int8_t* __restrict__ Xi, int F,
                                                                     const float* __restrict__ bias,
                                                                     const float* __restrict__ Wp,
                                                                     float* __restrict__ Y, int ldy,
                                                                     int B, int T, int tune) {
    constexpr int KT = H + KX, LDH = KT + 4, KB = KT / 8, NT = H / 32, NW = H / 32;
    static_assert(KB % 2 == 0, "k-blocks are consumed in pairs");
    extern __shared__ __attribute__((aligned(16))) float hs[];  // [MT][LDH] = [h | x | pad], then c

    int dir, btile;
    decode_block(blockIdx.x, dir, btile);
    const int b0 = btile * MT;
    if (b0 >= B) return;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave / (NW / 2);      // which 32-row half
    const int wq = wave % (NW / 2);       // owns unit tiles 2*wq, 2*wq+1
    const int li = lane & 31, hf = lane >> 5;

    // cell state lives in LDS ([wave][ut][r][lane], lane-contiguous = conflict free): it is only
    // touched in the gate phase, and keeping it out of the VGPR file leaves room for the 128
    // accumulator registers plus the weight prefetch buffers without spilling.
    float* cs = hs + MT * LDH + wave * (2 * 16 * 64) + lane;
    for (int idx = tid; idx < MT * LDH + NW * 2 * 16 * 64; idx += blockDim.x) hs[idx] = 0.0f;

    f32x16 acc[2][4];

    // Global traffic goes through raw buffer descriptors: a wave-uniform base (SGPR resource), ONE
    // 32-bit per-lane byte offset (VGPR) and a wave-uniform byte offset (SGPR soffset) per access,
    // so the 128 + 32 + 32 addresses of a step cost no VGPRs (hipcc otherwise hoists 64-bit
    // per-lane addresses out of the step loop and spills hundreds of registers).
    // row(r) = b0 + 32*grp + 4*hf + (r & 3) + 8*(r >> 2);  col(ut) = 64*wq + 32*ut + li
    const size_t urow = (size_t)(b0 + 32 * grp) * T;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(KX ? bias + dir * 4 * H + 64 * wq : Xp + urow * ldx + dir * 4 * H + 64 * wq), 0,
        0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrs =
        __builtin_amdgcn_make_buffer_rsrc(Y + urow * ldy + dir * H + 64 * wq, 0, 0x7fffffff, 0x00020000);
    // fragment (g, ut, kb) of this wave lives at byte ((g*NT + ut) * KB + kb) * 1024 + lane * 16
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(Wp + ((size_t)dir * (4 * NT) + 2 * wq) * KB * 256), 0, 0x7fffffff, 0x00020000);
    const unsigned xoff = KX ? li * 4u : ((unsigned)(4 * hf * T) * ldx + li) * 4u;
    const unsigned yoff = ((unsigned)(4 * hf * T) * ldy + li) * 4u;
    const unsigned woff = lane * 16u;
    float* hl = hs + (32 * grp + 4 * hf) * LDH + 64 * wq + li;
    const float* hrow = hs + (32 * grp + li) * LDH + hf * 4;

    // accumulator seed of a step: Xp row (unfused) or the per-column bias (fused)
    auto load_seed = [&](int t) {
#pragma unroll
        for (int ut = 0; ut < 2; ++ut)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (KX) {
                    const float bv = __builtin_bit_cast(
                        float, __builtin_amdgcn_raw_buffer_load_b32(xrs, xoff, (unsigned)(g * H + 32 * ut) * 4u, 0));
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[ut][g][r] = bv;
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const unsigned so = ((unsigned)(((r & 3) + 8 * (r >> 2)) * T + t) * ldx + g * H + 32 * ut) * 4u;
                        acc[ut][g][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, xoff, so, 0));
                    }
                }
            }
    };
    // fused only: this half's 32 x KX input slab of time t -> LDS columns [H, H+KX)
    auto stage_x = [&](int t) {
        if (KX) {
            const int gtid = (wave % (NW / 2)) * 64 + lane;
#pragma unroll
            for (int k = 0; k < (32 * KX) / (NW / 2 * 64); ++k) {
                const int e = gtid + k * (NW / 2 * 64);
                const int row = e / KX, f = e % KX;
                int brow = b0 + 32 * grp + row;
                brow = brow < B ? brow : B - 1;
                const float v = f < F ? (float)Xi[((size_t)brow * T + t) * F + f] : 0.0f;
                hs[(32 * grp + row) * LDH + H + f] = v;
            }
        }
    };
    auto load_kb = [&](int kb, f32x4 (&b)[2][4], f32x4& a) {   // both unit tiles of one k-block
#pragma unroll
        for (int ut = 0; ut < 2; ++ut)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                b[ut][g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                         wrs, woff, (unsigned)((g * NT + ut) * KB + kb) * 1024u, 0));
        a = *reinterpret_cast<const f32x4*>(hrow + kb * 8);
    };

    __syncthreads();                      // zero fill complete before x is staged on top of it
    load_seed(dir ? T - 1 : 0);
    stage_x(dir ? T - 1 : 0);
    __syncthreads();

    for (int i = 0; i <= 2 * T; ++i) {
        if ((i & 1) == grp) {
            // ---------------- MFMA phase of step (i - grp) / 2 ----------------
            if (((i - grp) >> 1) < T) {
                if (tune & 2) __builtin_amdgcn_s_setprio(1);   // feed the matrix pipe first
                // double buffer at k-block granularity: the 8 weight fragments + the A fragment of
                // k-block kb+1 are in flight while the 32 MFMAs (2048 pipe cycles) of kb issue
                f32x4 bw[2][2][4], af[2];
                load_kb(0, bw[0], af[0]);
                for (int kb = 0; kb < KB; kb += 2) {
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        if (kb + p + 1 < KB) load_kb(kb + p + 1, bw[p ^ 1], af[p ^ 1]);
                        // pin the prefetch ahead of this k-block's MFMAs: left alone, the scheduler
                        // sinks the loads to just before their use and the lone MFMA-phase wave of
                        // the SIMD stalls on every L2 round trip
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int ut = 0; ut < 2; ++ut)
#pragma unroll
                            for (int s = 0; s < 4; ++s)
#pragma unroll
                                for (int g = 0; g < 4; ++g)
                                    acc[ut][g] = mfma32(af[p][s], bw[p][ut][g][s], acc[ut][g]);
                    }
                }
                if (tune & 2) __builtin_amdgcn_s_setprio(0);
            }
        } else {
            // ---------------- gate phase of step (i - 1 - grp) / 2 ----------------
            const int gs = (i - 1 - grp) >> 1;
            if (i - 1 - grp >= 0 && gs < T) {
                const int t = dir ? T - 1 - gs : gs;
#pragma unroll
                for (int ut = 0; ut < 2; ++ut)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float ig = fast_sigmoid(acc[ut][0][r]);
                        const float fg = fast_sigmoid(acc[ut][1][r]);
                        const float gg = fast_tanh(acc[ut][2][r]);
                        const float og = fast_sigmoid(acc[ut][3][r]);
                        const float cn = fg * cs[(ut * 16 + r) * 64] + ig * gg;
                        cs[(ut * 16 + r) * 64] = cn;
                        const float hv = og * fast_tanh(cn);
                        const int dr = (r & 3) + 8 * (r >> 2);
                        hl[dr * LDH + 32 * ut] = hv;
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, hv), yrs, yoff,
                                                              ((unsigned)(dr * T + t) * ldy + 32 * ut) * 4u, 0);
                    }
                if (gs + 1 < T) {
                    const int tn = dir ? T - 2 - gs : gs + 1;
                    load_seed(tn);
                    stage_x(tn);
                }
            }
        }
        __syncthreads();
    }
}

// GRU: gates r,z,n.  Xp = W_ih x + b_ih (+ b_hr / b_hz folded in for r and z); the n gate keeps
// W_hn h + b_hn separate because it is multiplied by r (PyTorch GRU definition).
template <int H>
__global__ __launch_bounds__(H / 32 * 64, 2) void gru_rec_kernel(const float* __restrict__ Xp, int ldx,
                                                                 const float* __restrict__ Wp,
                                                                 const float* __restrict__ bhn,
                                                                 const float* __restrict__ h0, int ldh0,
                                                                 float* __restrict__ hn, int ldhn,
                                                                 float* __restrict__ Y, int ldy, int B,
                                                                 int T) {
    constexpr int LDH = H + 4, KB = H / 8, NT = H / 32;
    extern __shared__ __attribute__((aligned(16))) float hs[];  // [MT][LDH]

    int dir, btile;
    decode_block(blockIdx.x, dir, btile);
    const int b0 = btile * MT;
    if (b0 >= B) return;

    const int tid = threadIdx.x, lane = tid & 63;
    const int u = __builtin_amdgcn_readfirstlane(tid >> 6);   // this wave's 32-unit tile
    const int li = lane & 31, hf = lane >> 5;
    const int col = u * 32 + li;

    // all buffers are padded to a multiple of MT batch rows (see lstm_rec_kernel); global traffic
    // uses raw buffer descriptors = uniform base + one per-lane offset + uniform soffset
    // (see lstm_rec_pp_kernel).  row(m, r) = b0 + 4*hf + 32*m + (r & 3) + 8*(r >> 2)
    const size_t urow = (size_t)b0 * T;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(Xp + urow * ldx + dir * 3 * H + u * 32), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrs =
        __builtin_amdgcn_make_buffer_rsrc(Y + urow * ldy + dir * H + u * 32, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(Wp + ((size_t)dir * (3 * NT) + u) * KB * 256), 0, 0x7fffffff, 0x00020000);
    const unsigned xoff = ((unsigned)(4 * hf * T) * ldx + li) * 4u;
    const unsigned yoff = ((unsigned)(4 * hf * T) * ldy + li) * 4u;
    const unsigned woff = lane * 16u;
    const size_t lb = (size_t)(b0 + 4 * hf);
    float* hl = hs + 4 * hf * LDH + col;

    f32x16 hreg[2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int dr = 32 * m + (r & 3) + 8 * (r >> 2);
            const float hv = h0 != nullptr ? h0[(lb + dr) * ldh0 + dir * H + col] : 0.0f;
            hreg[m][r] = hv;
            hl[dr * LDH] = hv;
        }
    const float bn = bhn[dir * H + col];
    __syncthreads();

    auto load_w = [&](int kb, f32x4 (&b)[3]) {
#pragma unroll
        for (int g = 0; g < 3; ++g)
            b[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                 wrs, woff, (unsigned)(g * NT * KB + kb) * 1024u, 0));
    };

    for (int step = 0; step < T; ++step) {
        const int t = dir ? T - 1 - step : step;
        f32x16 acc[2][3];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const unsigned so = ((unsigned)((32 * m + (r & 3) + 8 * (r >> 2)) * T + t) * ldx) * 4u;
                acc[m][0][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, xoff, so, 0));
                acc[m][1][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, xoff, so + H * 4u, 0));
                acc[m][2][r] = bn;
            }

        const float* hrow = hs + li * LDH + hf * 4;
        f32x4 bw[2][3], a[2][2];
        load_w(0, bw[0]);
#pragma unroll
        for (int m = 0; m < 2; ++m) a[0][m] = *reinterpret_cast<const f32x4*>(hrow + m * 32 * LDH);
        for (int kb = 0; kb < KB; kb += 2) {
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                if (kb + p + 1 < KB) {
                    load_w(kb + p + 1, bw[p ^ 1]);
#pragma unroll
                    for (int m = 0; m < 2; ++m)
                        a[p ^ 1][m] = *reinterpret_cast<const f32x4*>(hrow + m * 32 * LDH + (kb + p + 1) * 8);
                }
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int g = 0; g < 3; ++g)
#pragma unroll
                        for (int m = 0; m < 2; ++m) acc[m][g] = mfma32(a[p][m][s], bw[p][g][s], acc[m][g]);
            }
        }
        __syncthreads();

#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dr = 32 * m + (r & 3) + 8 * (r >> 2);
                const unsigned so = ((unsigned)(dr * T + t) * ldx + 2 * H) * 4u;
                const float xn = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, xoff, so, 0));
                const float rg = fast_sigmoid(acc[m][0][r]);
                const float zg = fast_sigmoid(acc[m][1][r]);
                const float ng = fast_tanh(xn + rg * acc[m][2][r]);
                const float hv = (1.0f - zg) * ng + zg * hreg[m][r];
                hreg[m][r] = hv;
                hl[dr * LDH] = hv;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, hv), yrs, yoff,
                                                      ((unsigned)(dr * T + t) * ldy) * 4u, 0);
            }
        __syncthreads();
    }

    if (hn != nullptr) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                hn[(lb + 32 * m + (r & 3) + 8 * (r >> 2)) * ldhn + dir * H + col] = hreg[m][r];
    }
}

inline int rec_grid(int B) {
    const int nbt = (B + MT - 1) / MT;
    return 2 * ((nbt + 3) / 4) * 4;  // both directions, batch tiles padded to the 4-XCD groups
}

}  // namespace

namespace pa {

int tune_flags() {
    static const int v = [] {
        const char* e = getenv("PA_TUNE");
        return e ? atoi(e) : 0;
    }();
    return v;
}

hipError_t launch_lstm_rec(int H, const float* Xp, int ldx, const float* Wp, float* Y, int ldy,
                           int B, int T, hipStream_t stream) {
    if (B <= 0) return hipSuccess;
    const int grid = rec_grid(B);
    static const bool use_pp = [] {
        const char* e = getenv("PA_LSTM_PP");
        return !(e && e[0] == '0');
    }();
    if (H == 256 && use_pp) {
        const size_t lds = ((size_t)MT * (256 + 4) + 8 * 2 * 16 * 64) * sizeof(float);  // h + c
        hipLaunchKernelGGL((lstm_rec_pp_kernel<256, 0>), dim3(grid), dim3(512), lds, stream, Xp, ldx,
                           (const int8_t*)nullptr, 0, (const float*)nullptr, Wp, Y, ldy, B, T, tune_flags());
    } else if (H == 256) {
        const size_t lds = (size_t)MT * (256 + 4) * sizeof(float);
        hipLaunchKernelGGL((lstm_rec_kernel<256>), dim3(grid), dim3(512), lds, stream, Xp, ldx, Wp, Y,
                           ldy, B, T);
    } else if (H == 128) {
        const size_t lds = (size_t)MT * (128 + 4) * sizeof(float);
        hipLaunchKernelGGL((lstm_rec_kernel<128>), dim3(grid), dim3(256), lds, stream, Xp, ldx, Wp, Y,
                           ldy, B, T);
    } else {
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_lstm_rec_fused(int H, const int8_t* X, int F, const float* bias, const float* Wcat,
                                 float* Y, int ldy, int B, int T, hipStream_t stream) {
    if (B <= 0) return hipSuccess;
    if (H != 256 || F <= 0 || F > 32) return hipErrorInvalidValue;
    const int grid = rec_grid(B);
    const size_t lds = ((size_t)MT * (256 + 32 + 4) + 8 * 2 * 16 * 64) * sizeof(float);  // [h|x] + c
    hipLaunchKernelGGL((lstm_rec_pp_kernel<256, 32>), dim3(grid), dim3(512), lds, stream,
                       (const float*)nullptr, 0, X, F, bias, Wcat, Y, ldy, B, T, tune_flags());
    return hipGetLastError();
}

hipError_t launch_gru_rec(int H, const float* Xp, int ldx, const float* Wp, const float* bhn,
                          const float* h0, int ldh0, float* hn, int ldhn, float* Y, int ldy,
                          int B, int T, hipStream_t stream) {
    if (B <= 0) return hipSuccess;
    const int grid = rec_grid(B);
    if (H == 128) {
        const size_t lds = (size_t)MT * (128 + 4) * sizeof(float);
        hipLaunchKernelGGL((gru_rec_kernel<128>), dim3(grid), dim3(256), lds, stream, Xp, ldx, Wp, bhn,
                           h0, ldh0, hn, ldhn, Y, ldy, B, T);
    } else if (H == 256) {
        const size_t lds = (size_t)MT * (256 + 4) * sizeof(float);
        hipLaunchKernelGGL((gru_rec_kernel<256>), dim3(grid), dim3(512), lds, stream, Xp, ldx, Wp, bhn,
                           h0, ldh0, hn, ldhn, Y, ldy, B, T);
    } else {
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace pa
