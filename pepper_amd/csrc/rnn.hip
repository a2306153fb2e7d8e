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
        const_cast<float*>(KX ? bias + dir * 4 * H + 64 * wq
                              : Xp + (size_t)((b0 >> 5) + grp) * T * (ldx >> 5) * 1024), 0,
        0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrs =
        __builtin_amdgcn_make_buffer_rsrc(Y + urow * ldy + dir * H + 64 * wq, 0, 0x7fffffff, 0x00020000);
    // fragment (g, ut, kb) of this wave lives at byte ((g*NT + ut) * KB + kb) * 1024 + lane * 16
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(Wp + ((size_t)dir * (4 * NT) + 2 * wq) * KB * 256), 0, 0x7fffffff, 0x00020000);
    const unsigned xoff = KX ? li * 4u : lane * 16u;
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
                    // Xp is in MFMA fragment order (gemm.hip): tile (step t of this 32-row half,
                    // column tile) = 4 x 1 KiB chunks, chunk qd = accumulator registers 4qd..4qd+3
                    const unsigned ct = (unsigned)(dir * (4 * NT) + g * NT + 2 * wq + ut);
#pragma unroll
                    for (int qd = 0; qd < 4; ++qd) {
                        const unsigned so = (((unsigned)t * (ldx >> 5) + ct) * 4u + qd) * 1024u;
                        const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, xoff, so, 0));
                        acc[ut][g][4 * qd] = v.x;
                        acc[ut][g][4 * qd + 1] = v.y;
                        acc[ut][g][4 * qd + 2] = v.z;
                        acc[ut][g][4 * qd + 3] = v.w;
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
    // quad q = (k-block q >> 1, unit tile q & 1): the 4 gate fragments one A fragment meets
    auto load_quad = [&](int q, f32x4 (&b)[4]) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
            b[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                 wrs, woff, (unsigned)((g * NT + (q & 1)) * KB + (q >> 1)) * 1024u, 0));
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
                // 3-deep ring of quads: two quads (2 x 16 MFMAs = 2048 pipe cycles) are in flight
                // ahead of the one being consumed; the A fragment of the next k-block is read one
                // quad ahead.  sched_barrier pins the issue order (the scheduler otherwise sinks the
                // loads next to their use and this lone MFMA-phase wave eats every L2 round trip).
                constexpr int NQ = 2 * KB;
                f32x4 ring[3][4], a_cur, a_nxt;
                load_quad(0, ring[0]);
                load_quad(1, ring[1]);
                a_cur = *reinterpret_cast<const f32x4*>(hrow);
                a_nxt = a_cur;
                for (int q0 = 0; q0 < NQ; q0 += 6) {
#pragma unroll
                    for (int p = 0; p < 6; ++p) {
                        const int q = q0 + p;
                        if (q < NQ) {
                            if (q + 2 < NQ) load_quad(q + 2, ring[(p + 2) % 3]);
                            if ((p & 1) == 0 && q + 2 < NQ)
                                a_nxt = *reinterpret_cast<const f32x4*>(hrow + ((q >> 1) + 1) * 8);
#pragma unroll
                            for (int s = 0; s < 4; ++s)
#pragma unroll
                                for (int g = 0; g < 4; ++g)
                                    acc[p & 1][g] = mfma32(a_cur[s], ring[p % 3][g][s], acc[p & 1][g]);
                            if (p & 1) a_cur = a_nxt;
                            // Issue order inside the quad: ONE memory instruction per MFMA gap.  A
                            // VMEM issue costs this wave ~60 cycles; back to back, only the first
                            // hides in the shadow of the preceding 64-cycle MFMA and the rest stall
                            // the matrix pipe (measured: 75 % -> MfmaUtil with block issue).
                            if (q + 2 < NQ) {
#pragma unroll
                                for (int k = 0; k < 4; ++k) {
                                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
                                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // VMEM read
                                }
                                if ((p & 1) == 0) {
                                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read
                                    __builtin_amdgcn_sched_group_barrier(0x008, 11, 0);
                                } else {
                                    __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
                                }
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
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
    // Xp is in MFMA fragment order (gemm.hip): row tile = (32-batch block, step), 4 x 1 KiB chunks
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(Xp + (size_t)(b0 >> 5) * T * (ldx >> 5) * 1024), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrs =
        __builtin_amdgcn_make_buffer_rsrc(Y + urow * ldy + dir * H + u * 32, 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(Wp + ((size_t)dir * (3 * NT) + u) * KB * 256), 0, 0x7fffffff, 0x00020000);
    const unsigned xoff = lane * 16u;
    const unsigned yoff = ((unsigned)(4 * hf * T) * ldy + li) * 4u;
    const unsigned woff = lane * 16u;
    const size_t lb = (size_t)(b0 + 4 * hf);
    auto load_xp = [&](int m, int t, int g, f32x16& dst) {
        const unsigned ct = (unsigned)(dir * (3 * NT) + g * NT + u);
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            const unsigned so = (((unsigned)(m * T + t) * (ldx >> 5) + ct) * 4u + qd) * 1024u;
            const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, xoff, so, 0));
            dst[4 * qd] = v.x;
            dst[4 * qd + 1] = v.y;
            dst[4 * qd + 2] = v.z;
            dst[4 * qd + 3] = v.w;
        }
    };
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
        for (int m = 0; m < 2; ++m) {
            load_xp(m, t, 0, acc[m][0]);
            load_xp(m, t, 1, acc[m][1]);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][2][r] = bn;
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
        for (int m = 0; m < 2; ++m) {
            f32x16 xnv;
            load_xp(m, t, 2, xnv);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dr = 32 * m + (r & 3) + 8 * (r >> 2);
                const float xn = xnv[r];
                const float rg = fast_sigmoid(acc[m][0][r]);
                const float zg = fast_sigmoid(acc[m][1][r]);
                const float ng = fast_tanh(xn + rg * acc[m][2][r]);
                const float hv = (1.0f - zg) * ng + zg * hreg[m][r];
                hreg[m][r] = hv;
                hl[dr * LDH] = hv;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, hv), yrs, yoff,
                                                      ((unsigned)(dr * T + t) * ldy) * 4u, 0);
            }
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
    if (H != 256) return hipErrorInvalidValue;   // the reference hard-codes lstm_*_hidden_size = 256
    const int grid = rec_grid(B);
    const size_t lds = ((size_t)MT * (256 + 4) + 8 * 2 * 16 * 64) * sizeof(float);  // h + c
    hipLaunchKernelGGL((lstm_rec_pp_kernel<256, 0>), dim3(grid), dim3(512), lds, stream, Xp, ldx,
                       (const int8_t*)nullptr, 0, (const float*)nullptr, Wp, Y, ldy, B, T, tune_flags());
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
