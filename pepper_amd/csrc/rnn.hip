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

    const int tid = threadIdx.x, lane = tid & 63, u = tid >> 6;
    const int li = lane & 31, hf = lane >> 5;
    const int col = u * 32 + li;

    // all buffers are padded to a multiple of MT batch rows (see lstm_rec_kernel)
    const size_t lb = (size_t)(b0 + 4 * hf);
    const float* xl = Xp + lb * T * ldx + dir * 3 * H + col;
    float* yl = Y + lb * T * ldy + dir * H + col;
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
    const f32x4* wp = reinterpret_cast<const f32x4*>(Wp) + (size_t)dir * (3 * NT) * KB * 64 + lane;
    __syncthreads();

    for (int step = 0; step < T; ++step) {
        const int t = dir ? T - 1 - step : step;
        f32x16 acc[2][3];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float* xp = xl + ((size_t)(32 * m + (r & 3) + 8 * (r >> 2)) * T + t) * ldx;
                acc[m][0][r] = xp[0];
                acc[m][1][r] = xp[H];
                acc[m][2][r] = bn;
            }

        const float* hrow = hs + li * LDH + hf * 4;
#pragma unroll 2
        for (int kb = 0; kb < KB; ++kb) {
            f32x4 a[2], b[3];
#pragma unroll
            for (int m = 0; m < 2; ++m) a[m] = *reinterpret_cast<const f32x4*>(hrow + m * 32 * LDH + kb * 8);
#pragma unroll
            for (int g = 0; g < 3; ++g) b[g] = wp[((size_t)(g * NT + u) * KB + kb) * 64];
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int g = 0; g < 3; ++g)
#pragma unroll
                    for (int m = 0; m < 2; ++m) acc[m][g] = mfma32(a[m][s], b[g][s], acc[m][g]);
        }
        __syncthreads();

#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dr = 32 * m + (r & 3) + 8 * (r >> 2);
                const float xn = xl[((size_t)dr * T + t) * ldx + 2 * H];
                const float rg = sigmoid_f(acc[m][0][r]);
                const float zg = sigmoid_f(acc[m][1][r]);
                const float ng = tanhf(xn + rg * acc[m][2][r]);
                const float hv = (1.0f - zg) * ng + zg * hreg[m][r];
                hreg[m][r] = hv;
                hl[dr * LDH] = hv;
                yl[((size_t)dr * T + t) * ldy] = hv;
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

hipError_t launch_lstm_rec(int H, const float* Xp, int ldx, const float* Wp, float* Y, int ldy,
                           int B, int T, hipStream_t stream) {
    if (B <= 0) return hipSuccess;
    const int grid = rec_grid(B);
    if (H == 256) {
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
