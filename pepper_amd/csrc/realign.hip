// Read re-aligner (include/pepper_amd_realign.h): the polish image generator's local re-alignment of every read
// against the reference suffix that starts at its mapped position, as HIP kernels -- one wavefront per read and pass.
//
// What is reproduced, stage by stage (the reference calls the SSW library for each read on one host thread:
// /root/reference/pepper/modules/src/local_reassembly/simple_aligner.cpp:66-106, ssw.c:801-891):
//   sw_ends_kernel   score and end cell of the best local alignment, then the begin cell from the same pass over the
//                    reversed read prefix / reversed reference prefix (ssw.c:161-368 8-bit lanes, 393-569 16-bit
//                    lanes).  The library's striped layout leaves a trace in the arithmetic: the read is cut into 16
//                    (8-bit) or 8 (16-bit) equal segments, and the horizontal-gap state of the next column is opened
//                    from the cell value that only knows the vertical gaps opened inside its own segment.  Both
//                    vertical-gap chains (segment-local and exact) are carried here, so the scores, the first column
//                    reaching the maximum and the smallest row inside it are the library's.  8-bit pass first; a running
//                    maximum >= 249 switches to the 16-bit segmentation (ssw.c:819-824).
//                    Mapping: lane l owns a strip of consecutive read rows in registers (score_pass_reg<R>; reads
//                    beyond 1536 padded rows fall back to one LDS dword per row, score_pass), columns are visited in a
//                    skewed pipeline (lane l works on column t - l at step t) and the strip's bottom cell, both gap
//                    chains and the running column maximum are handed to lane l + 1 by three DPP moves.  Lane 63 sees
//                    each column's complete maximum and runs the sequential part (first column that raises the
//                    maximum, overflow, early stop of the reverse pass at the forward score).  Integer DP, no matrix
//                    cores: 12.5 vector instructions per cell.
//   band_kernel      banded DP between begin and end cell with the library's band slots, its zeroed slot to the right
//                    of the previous row and its tie rules (ssw.c:571-650); three wavefronts per read try three
//                    consecutive widths of the doubling sequence at once, the narrowest that reaches the score wins.
//                    The row's vertical-gap chain is a max-plus prefix scan on the DPP network; direction bits go to a
//                    workspace in HBM (1 byte per cell); the walk back (ssw.c:653-703) probes 64 cells of the diagonal
//                    per round trip, and the final operations -- '=' / 'X' runs from comparing base codes, I, D, soft
//                    clips (ssw_cpp.cpp:56-207) -- are emitted wavefront-parallel into one compacted output.
// Host side of the entry points: job table over one or several reference windows, workspace sizing between the two
// kernels (base text -> codes on the device).
#include "../../include/pepper_amd_realign.h"
#include "../../include/pepper_amd.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "common.h"
#include "encoder_common.h"
#include "kernels.h"

namespace {

constexpr int S_MATCH = 4, S_MIS = 6, GO = 8, GE = 2, BIAS = 6;     // simple_aligner.h:19-25
constexpr int MAX_READ = 4000;                                       // 14-bit cell fields: 4 * m < 16384
constexpr int NEG = -(1 << 28);
enum { ST_NEW = 0, ST_BAND = 1, ST_DONE = 2, ST_KEPT = 3, ST_DROPPED = 4, ST_WIDER = 5, ST_ERR = -1 };
enum { OP_I = 1, OP_D = 2, OP_S = 4, OP_EQ = 7, OP_X = 8 };

struct Job {
    int32_t ref_off, n;            // reference suffix: codes [ref_off, ref_off + n)
    int32_t m, state;
    int64_t seq_off;               // read codes
    int32_t score, wide, ref_begin, ref_end, read_begin, read_end;
    int32_t bw, dir_width;         // band half width to try next; row capacity of the direction workspace
    int64_t dir_off, ops_off;      // direction workspace of this read; first operation in the compacted output
    int32_t ops_cap, n_ops;        // worst-case number of operations (sizes the output buffer); operations written
    int32_t t_ends, t_dp, t_trace, t_emit;   // stage times of this read in 10 ns ticks (s_memtime), for tools/realign_stages.py
};

// base text -> codes in place (ssw_cpp.cpp:10-19: A/a 0, C/c 1, G/g 2, T/t 3, U/u 0, everything else 4)
__global__ void to_codes_kernel(int8_t* __restrict__ buf, int64_t n) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int c = buf[k] & 0xdf;
    buf[k] = (int8_t)(c == 'A' || c == 'U' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : 4);
}

__device__ __forceinline__ int bcast63(int v) { return __builtin_amdgcn_readlane(v, 63); }
__device__ __forceinline__ int uni_value(int v) { return __builtin_amdgcn_readfirstlane(v); }      // the same in every lane: say so

struct PassOut { int score, ref, read, overflow; };

// One striped-semantics score pass over `count` reference columns (first, first + step, ...) and the read rows
// read[rfirst + r * rstep], r < m.  he: this wavefront's LDS, 64 * R dwords.
__device__ PassOut score_pass(uint32_t* he, const int8_t* __restrict__ ref, int first, int step, int count,
                              const int8_t* __restrict__ read, int rfirst, int rstep, int m, int lanes, int terminate) {
    const int lane = threadIdx.x;
    const int L = (m + lanes - 1) / lanes, rows = L * lanes, R = (rows + 63) >> 6;
    for (int k = 0; k < R; ++k) {
        const int r = lane * R + k;
        unsigned q = r < m ? (unsigned)read[rfirst + r * rstep] : (r < rows ? 5u : 7u);   // 5: padding row, 7: none
        if (r < rows && r % L == 0) q |= 8u;                                               // segment start
        he[k * 64 + lane] = q << 28;
    }
    int run_max = 0, end_ref = lanes == 16 ? -1 : 0, end_row = -1, stop = 0, overflow = 0;
    int diag_in = 0;                                    // H(previous column, last row of the lane above)
    int o_h = 0, o_fs = 0, o_ff = 0, o_cm = -1, o_cr = 0;
    const int steps = count + 63;
    for (int t = 0; t < steps; ++t) {
        int i_h = __shfl_up(o_h, 1, 64), i_fs = __shfl_up(o_fs, 1, 64), i_ff = __shfl_up(o_ff, 1, 64);
        int i_cm = __shfl_up(o_cm, 1, 64), i_cr = __shfl_up(o_cr, 1, 64);
        if (lane == 0) { i_h = 0; i_fs = 0; i_ff = 0; i_cm = -1; i_cr = 0; }
        const int c = t - lane;
        if (c >= 0 && c < count) {
            const int rc = ref[first + c * step];
            int dsrc = diag_in, fs = i_fs, ff = i_ff, lmax = -1, lrow = 0, h = 0;
            for (int k = 0; k < R; ++k) {
                const unsigned w = he[k * 64 + lane];
                const unsigned qf = w >> 28, q = qf & 7u;
                if (q == 7u) break;
                const int hp = (int)(w & 0x3fffu), e = (int)((w >> 14) & 0x3fffu);
                const int s = q == 5u ? 0 : (((int)q == rc && q < 4u) ? S_MATCH : -S_MIS);
                const int diag = dsrc + s;
                if (qf & 8u) fs = 0;
                const int hs = max(max(diag, e), fs);        // cell value before the exact vertical-gap correction
                h = max(hs, ff);
                const int e2 = max(max(e - GE, hs - GO), 0);
                he[k * 64 + lane] = (unsigned)h | ((unsigned)e2 << 14) | (qf << 28);
                dsrc = hp;
                fs = max(max(fs - GE, hs - GO), 0);
                ff = max(max(ff - GE, h - GO), 0);
                if (h > lmax) { lmax = h; lrow = lane * R + k; }
            }
            diag_in = i_h;
            o_h = h; o_fs = fs; o_ff = ff;
            if (lmax > i_cm) { o_cm = lmax; o_cr = lrow; } else { o_cm = i_cm; o_cr = i_cr; }
            if (lane == 63) {
                if (o_cm > run_max) {
                    run_max = o_cm;
                    if (lanes == 16 && run_max + BIAS >= 255) { overflow = 1; stop = 1; }
                    else { end_ref = first + c * step; end_row = o_cr; }
                }
                if (!stop && o_cm == terminate) stop = 1;
            }
        }
        if (bcast63(stop)) break;
    }
    PassOut o;
    o.overflow = bcast63(overflow);
    const int rm = bcast63(run_max), er = bcast63(end_row);
    o.score = o.overflow ? 255 : rm;
    o.ref = bcast63(end_ref);
    o.read = m - 1;
    if (rm == 0) { if (m - 1 > 0) o.read = 0; }
    else if (er < m - 1) o.read = er;
    return o;
}

__device__ __forceinline__ int dpp_up1(int v) {          // lane l <- lane l - 1 (v_mov_b32_dpp wave_shr:1), lane 0 <- 0
    return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, false);
}
__device__ __forceinline__ int max3i(int a, int b, int c) { return max(max(a, b), c); }
__device__ __forceinline__ int dpp_up1_or(int v, int lane0) {      // same, lane 0 <- lane0
    return __builtin_amdgcn_update_dpp(lane0, v, 0x138, 0xf, 0xf, false);
}
// inclusive prefix maximum across the wavefront on the DPP network (row shifts inside 16-lane rows, then row broadcasts)
__device__ __forceinline__ int wave_prefix_max(int v, int identity) {
    v = max(v, __builtin_amdgcn_update_dpp(identity, v, 0x111, 0xf, 0xf, false));   // row_shr:1
    v = max(v, __builtin_amdgcn_update_dpp(identity, v, 0x112, 0xf, 0xf, false));   // row_shr:2
    v = max(v, __builtin_amdgcn_update_dpp(identity, v, 0x114, 0xf, 0xf, false));   // row_shr:4
    v = max(v, __builtin_amdgcn_update_dpp(identity, v, 0x118, 0xf, 0xf, false));   // row_shr:8
    v = max(v, __builtin_amdgcn_update_dpp(identity, v, 0x142, 0xa, 0xf, false));   // row_bcast:15 -> rows 1, 3
    v = max(v, __builtin_amdgcn_update_dpp(identity, v, 0x143, 0xc, 0xf, false));   // row_bcast:31 -> rows 2, 3
    return v;
}
// The same with the signed maximum's own identity in the lanes a shift leaves empty: `max(v, shifted-or-INT_MIN)` is then one
// v_max_i32 with the DPP modifier (a lane without a source keeps v) instead of constant + move + maximum.  For values > INT_MIN.
__device__ __forceinline__ int wave_prefix_max_min(int v) {
    constexpr int ID = (int)0x80000000u;
    v = max(v, __builtin_amdgcn_update_dpp(ID, v, 0x111, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(ID, v, 0x112, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(ID, v, 0x114, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(ID, v, 0x118, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(ID, v, 0x142, 0xa, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(ID, v, 0x143, 0xc, 0xf, false));
    return v;
}

// The same pass with the lane's strip of R rows resident in registers (reads of up to 64 R rows): per cell 15 vector
// ops and no LDS traffic; the hand-off to the next lane is three DPP moves (H | segment-local gap chain, exact gap
// chain, column maximum as one key = value << 12 | 4095 - row so that a plain unsigned max keeps the smallest row).
template <int R>
__device__ __noinline__ PassOut score_pass_reg(const int8_t* __restrict__ ref, int first, int step, int count,
                                               const int8_t* __restrict__ read, int rfirst, int rstep, int m, int lanes,
                                               int terminate) {
    const int lane = threadIdx.x;
    const int L = (m + lanes - 1) / lanes, rows = L * lanes;
    int H[R], E[R], Q[R], MIS[R], SEG[R];
    unsigned RINV[R];
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int r = lane * R + k;
        int q = 10;                                       // beyond the padded read: contributes nothing
        if (r < m) {
            q = read[rfirst + r * rstep];
            if (q >= 4) q = 8;                            // N never matches (not even N)
        } else if (r < rows) {
            q = 9;                                        // padding row of the last segment: scores 0 against everything
        }
        Q[k] = q;
        MIS[k] = r < m ? -S_MIS : (r < rows ? 0 : -(1 << 20));
        SEG[k] = (r >= rows || r % L == 0) ? 0 : -1;      // and-mask of the segment-local chain
        RINV[k] = (unsigned)(4095 - r);
        H[k] = 0;
        E[k] = 0;
    }
    int run_max = 0, end_ref = lanes == 16 ? -1 : 0, end_row = -1, stop = 0, overflow = 0, diag_in = 0;
    int p0 = 0, p1 = 0;
    unsigned p2 = 0;
    int rc_next = lane == 0 ? ref[first] : 0;
    const int steps = count + 63;
    for (int t = 0; t < steps; ++t) {
        const int i0 = dpp_up1(p0), i1 = dpp_up1(p1);
        const unsigned i2 = (unsigned)dpp_up1((int)p2);
        const int c = t - lane, rc = rc_next;
        {
            const int cn = c + 1;
            rc_next = (cn >= 0 && cn < count) ? ref[first + cn * step] : 0;
        }
        if (c >= 0 && c < count) {
            int dsrc = diag_in, fs = i0 >> 16, ff = i1;
            unsigned key = 0;
#pragma unroll
            for (int k = 0; k < R; ++k) {
                const int diag = dsrc + (Q[k] == rc ? S_MATCH : MIS[k]);
                dsrc = H[k];
                fs &= SEG[k];
                const int hs = max3i(diag, E[k], fs), h = max(hs, ff), hsgo = hs - GO;
                E[k] = max3i(E[k] - GE, hsgo, 0);
                fs = max3i(fs - GE, hsgo, 0);
                ff = max3i(ff - GE, hsgo, 0);             // = max(ff - GE, h - GO, 0): ff - GO < ff - GE
                H[k] = h;
                key = max(key, ((unsigned)h << 12) | RINV[k]);
            }
            diag_in = i0 & 0xffff;
            p0 = H[R - 1] | (fs << 16);
            p1 = ff;
            p2 = max(i2, key);
            if (lane == 63) {
                const int cm = (int)(p2 >> 12);
                if (cm > run_max) {
                    run_max = cm;
                    if (lanes == 16 && run_max + BIAS >= 255) { overflow = 1; stop = 1; }
                    else { end_ref = first + c * step; end_row = 4095 - (int)(p2 & 0xfffu); }
                }
                if (!stop && cm == terminate) stop = 1;
            }
        }
        if (bcast63(stop)) break;
    }
    PassOut o;
    o.overflow = bcast63(overflow);
    const int rm = bcast63(run_max), er = bcast63(end_row);
    o.score = o.overflow ? 255 : rm;
    o.ref = bcast63(end_ref);
    o.read = m - 1;
    if (rm == 0) { if (m - 1 > 0) o.read = 0; }
    else if (er < m - 1) o.read = er;
    return o;
}

constexpr int REG_ROWS = 64 * 24;          // longest padded read the register-resident pass takes

// MAXN: the longest strip (rows per lane) this instantiation carries in registers.  The strip's arrays set the kernel's
// register count -- 280 with all twelve sizes in one kernel, i.e. ONE wavefront per SIMD for every read, although the reads of
// a polish region (clipped to a ~1.2 kb window: 20 rows per lane) need 2/3 of that; the host picks the instantiation from the
// longest read of the call, so ordinary calls run with two wavefronts per SIMD.
template <int MAXN>
__device__ PassOut score_pass_any(uint32_t* he, const int8_t* __restrict__ ref, int first, int step, int count,
                                  const int8_t* __restrict__ read, int rfirst, int rstep, int m, int lanes, int terminate) {
    const int rows = ((m + lanes - 1) / lanes) * lanes, R = (rows + 63) >> 6;
#define PA_PASS(N) if constexpr (MAXN >= N) return score_pass_reg<N>(ref, first, step, count, read, rfirst, rstep, m, lanes, terminate); else break
    switch ((R + 1) >> 1) {
        case 0: case 1: PA_PASS(2);
        case 2: PA_PASS(4);
        case 3: PA_PASS(6);
        case 4: PA_PASS(8);
        case 5: PA_PASS(10);
        case 6: PA_PASS(12);
        case 7: PA_PASS(14);
        case 8: PA_PASS(16);
        case 9: PA_PASS(18);
        case 10: PA_PASS(20);
        case 11: PA_PASS(22);
        case 12: PA_PASS(24);
        default: break;
    }
#undef PA_PASS
    return score_pass(he, ref, first, step, count, read, rfirst, rstep, m, lanes, terminate);
}

template <int MAXN>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(MAXN <= 20 ? 2 : 1, MAXN <= 20 ? 2 : 1))) void sw_ends_kernel(Job* __restrict__ jobs, const int8_t* __restrict__ ref,
                                                     const int8_t* __restrict__ seq, int r_lo, int r_hi) {
    extern __shared__ uint32_t he[];
    Job& J = jobs[blockIdx.x];
    if (J.state != ST_NEW) return;
    {   // a launch takes the reads whose strip (rows per lane of the 8-bit segmentation, the longer one) lies in (r_lo, r_hi]:
        // the device-fed form launches every instantiation over the whole table, each read runs in the narrowest that holds it
        const int R = ((((J.m + 15) / 16) * 16) + 63) >> 6;
        if (R <= r_lo || R > r_hi) return;
    }
    const int8_t* rf = ref + J.ref_off;
    const int8_t* rd = seq + J.seq_off;
    const int n = J.n, m = J.m;
    // pass 0: 8-bit segmentation; pass 1: 16-bit segmentation if pass 0 overflowed; pass 2: begin cell (reversed)
    PassOut f = {0, 0, 0, 0};
    int wide = 0, ref_begin = -1, read_begin = -1;
    const long long t_start = wall_clock64();
    f.overflow = uni_value(J.wide) != 0;          // the job's builder has proven that the 8-bit pass overflows (overflow_proven): not run
    for (int pass = f.overflow ? 1 : 0; pass < 3; ++pass) {
        if (pass == 1 && !f.overflow) continue;
        if (pass == 2 && !(f.score > 0 && f.ref >= 0)) continue;
        const bool rev = pass == 2;
        const PassOut o = score_pass_any<MAXN>(he, rf, rev ? f.ref : 0, rev ? -1 : 1, rev ? f.ref + 1 : n, rd, rev ? f.read : 0,
                                         rev ? -1 : 1, rev ? f.read + 1 : m, (pass == 1 || (rev && wide)) ? 8 : 16,
                                         rev ? f.score : -1);
        if (rev) {
            ref_begin = o.ref;
            read_begin = f.read - o.read;
        } else {
            f = o;
            wide = pass;
        }
    }
    if (threadIdx.x == 0) {
        J.score = f.score; J.wide = wide; J.ref_end = f.ref; J.read_end = f.read;
        J.ref_begin = ref_begin; J.read_begin = read_begin;
        J.t_ends = (int32_t)(wall_clock64() - t_start);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Two reads per wavefront (round 5).  The cell values of a polish region's reads stay below 2^15 (4 x 4000), so a 32-bit lane
// register holds the same cell of TWO reads -- read A in the low halves, read B in the high halves -- and the packed 16-bit
// instructions (v_pk_add_u16 / v_pk_max_i16 / v_pk_max_u16 / v_pk_sub_u16 clamp) update both with one issue: 15 vector
// instructions per PAIR of cells where score_pass_reg needs 12.5 + 2 per cell.  What changes against score_pass_reg:
//   * substitution score by v_perm_b32: the strip keeps G = H - 6; a column's table word holds 10 in byte `code` of its half
//     (0 elsewhere), a row's selector byte picks byte `base` of its half's table or the constant 0x00 (N, rows past the read):
//     diag = G + {10, 0} = H + {4, -6}.  The library's padding rows of the last segment (score 0 against everything) are
//     treated as never-matching rows too: a padding row can only repeat, never exceed, a value an earlier column held in the
//     read's last row, and every quantity the pass returns changes on a STRICT increase of the running maximum only;
//   * the gap terms are non-negative, so `max(x - GE, hs - GO, 0)` is two saturating unsigned subtractions and one maximum;
//   * the row of the end cell is not carried per cell: every lane keeps the maximum it has seen in its strip over all columns,
//     and when a column's strip maximum exceeds both that and the maximum of the lanes above in this column, it copies its
//     strip to LDS and notes the column -- the lane that does so in the first column reaching the final maximum is the one
//     that holds the end cell (nothing it sees later is larger), and the smallest row is found in its copy after the pass;
//   * the two reads run in lock step over max(columns) steps: the shorter one sees never-matching columns past its end,
//     which lane 63 does not count.  The callers pair reads of like size (sorted by window length).
typedef short pk_s16 __attribute__((ext_vector_type(2)));
typedef unsigned short pk_u16 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk_max_i(unsigned a, unsigned b) {
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(pk_s16, a), __builtin_bit_cast(pk_s16, b)));
}
__device__ __forceinline__ unsigned pk_max_u(unsigned a, unsigned b) {
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(pk_u16, a), __builtin_bit_cast(pk_u16, b)));
}
__device__ __forceinline__ unsigned pk_subs_u(unsigned a, unsigned b) {
    return __builtin_bit_cast(unsigned, __builtin_elementwise_sub_sat(__builtin_bit_cast(pk_u16, a), __builtin_bit_cast(pk_u16, b)));
}
__device__ __forceinline__ unsigned pk_add(unsigned a, unsigned b) {
    return __builtin_bit_cast(unsigned, __builtin_bit_cast(pk_u16, a) + __builtin_bit_cast(pk_u16, b));
}
__device__ __forceinline__ unsigned dpp_up1_u(unsigned v, unsigned lane0) {
    return (unsigned)__builtin_amdgcn_update_dpp((int)lane0, (int)v, 0x138, 0xf, 0xf, false);
}

__device__ __forceinline__ unsigned dpp_up1_z(unsigned v) {          // lane 0 gets 0 (bound_ctrl: no register to preset)
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, true);
}

struct HalfIn {                      // one read's side of a pass
    const int8_t* ref; int first, count;         // reference columns first, first + step, ... (count of them)
    const int8_t* read; int rfirst, m;           // read rows read[rfirst + r * rstep], r < m
    int lanes, terminate;                        // 16 / 8 segments; stop when a column's maximum equals this (-1: never)
};

constexpr unsigned G_ZERO = 0xFFFAFFFAu;         // H = 0 in both halves of a G register

template <int R>
__device__ __forceinline__ void score_pass_pk(uint32_t* snap, const HalfIn a, const HalfIn b, int step, int rstep, PassOut& oa, PassOut& ob) {
    static_assert(R % 2 == 0, "the strip is copied to LDS four (and at the end two) rows at a time");
    constexpr int RP = (R + 3) & ~3;              // a lane's copy starts on a 16-byte boundary
    const int lane = threadIdx.x;
    const int La = a.m > 0 ? (a.m + a.lanes - 1) / a.lanes : 1, Lb = b.m > 0 ? (b.m + b.lanes - 1) / b.lanes : 1;
    const int rows_a = a.m > 0 ? La * a.lanes : 0, rows_b = b.m > 0 ? Lb * b.lanes : 0;
    unsigned G[R], E[R], SEL[R], SEG[R];
    // the strips' base codes: all the loads first (clamped rows: no branch between them), one wait
    int qa[R], qb[R];
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int r = lane * R + k;
        qa[k] = a.read[a.rfirst + (r < a.m ? r : 0) * rstep];       // (m == 0: rfirst == 0, a readable byte of the read text)
        qb[k] = b.read[b.rfirst + (r < b.m ? r : 0) * rstep];
    }
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int r = lane * R + k;
        const unsigned sa = (r < a.m && (unsigned)qa[k] < 4u) ? (unsigned)qa[k] : 0x0Cu;
        const unsigned sb = (r < b.m && (unsigned)qb[k] < 4u) ? 4u + (unsigned)qb[k] : 0x0Cu;
        SEL[k] = sa | (0x0Cu << 8) | (sb << 16) | (0x0Cu << 24);
        SEG[k] = ((r >= rows_a || r % La == 0) ? 0u : 0xFFFFu) | ((r >= rows_b || r % Lb == 0) ? 0u : 0xFFFF0000u);
        G[k] = G_ZERO;
        E[k] = 0;
    }
    uint32_t* my_a = snap + (size_t)lane * RP;
    uint32_t* my_b = snap + (size_t)64 * RP + (size_t)lane * RP;
    auto keep = [&](uint32_t* dst, const unsigned (&g)[R]) {
#pragma unroll
        for (int k = 0; k + 4 <= R; k += 4) *reinterpret_cast<uint4*>(dst + k) = make_uint4(g[k], g[k + 1], g[k + 2], g[k + 3]);
        if constexpr (R % 4 == 2) *reinterpret_cast<uint2*>(dst + R - 2) = make_uint2(g[R - 2], g[R - 1]);
    };
    // lane 63's view of the two reads
    int run_a = 0, run_b = 0, endc_a = -1, endc_b = -1, ovf_a = 0, ovf_b = 0;
    int stop_a = a.count <= 0 || a.m <= 0, stop_b = b.count <= 0 || b.m <= 0;
    unsigned diag_in = G_ZERO, p0 = G_ZERO, p1 = 0, p2 = 0, p3 = 0, best = 0;
    int col_a = -1, col_b = -1;
    const int cmax = a.count > b.count ? a.count : b.count;
    // Every step updates the strip unconditionally: a lane that has not reached its first column yet (c < 0), or has passed the
    // last one, sees a never-matching column -- before the first column that leaves H = E = F = 0 as they are, after the last
    // one nothing reads the strip any more -- so the loop body has no divergent region around the 15 R instructions (the
    // conditional form kept two copies of the strip alive across the branch: 283 registers at 20 rows per lane).
    // the column's base code is fetched one step ahead and turned into the table word when it is used.  The offset of a lane's
    // column moves by `step` while the column index moves inside [0, count - 1] and stays put outside (a clamped index: no branch,
    // every load inside the window text; count == 0: first == 0, a readable byte) -- no multiply, no 64-bit sum per step
    auto table_of = [](const HalfIn& h, bool started, int c, int code) {
        return (started && c < h.count && (unsigned)code < 4u) ? 10u << (8 * code) : 0u;
    };
    unsigned off_a = (unsigned)a.first, off_b = (unsigned)b.first;
    int rawa_next = a.ref[off_a], rawb_next = b.ref[off_b];
    const int steps = cmax + 63;
    for (int t = 0; t < steps; ++t) {
        const unsigned i0 = dpp_up1_u(p0, G_ZERO), i1 = dpp_up1_z(p1), i2 = dpp_up1_z(p2), i3 = dpp_up1_z(p3);
        const int c = t - lane;
        const bool started = c >= 0;
        const unsigned tab_a = table_of(a, started, c, rawa_next), tab_b = table_of(b, started, c, rawb_next);
        off_a += (started && c + 1 < a.count) ? (unsigned)step : 0u;
        off_b += (started && c + 1 < b.count) ? (unsigned)step : 0u;
        rawa_next = a.ref[off_a];
        rawb_next = b.ref[off_b];
        // (row k + 1's diagonal term is formed from G[k] BEFORE row k writes it: the old value dies there and the new one takes its
        // register -- read after the write it cost a register move per row and step)
        unsigned diag = pk_add(diag_in, __builtin_amdgcn_perm(tab_b, tab_a, SEL[0])), fs = i1, ff = i2, lm = 0;
#pragma unroll
        for (int k = 0; k < R; ++k) {
            const unsigned diag_k = diag;
            if (k + 1 < R) diag = pk_add(G[k], __builtin_amdgcn_perm(tab_b, tab_a, SEL[k + 1]));
            fs &= SEG[k];
            const unsigned hs = pk_max_i(pk_max_i(diag_k, E[k]), fs), h = pk_max_i(hs, ff), hsgo = pk_subs_u(hs, 0x00080008u);
            E[k] = pk_max_u(pk_subs_u(E[k], 0x00020002u), hsgo);
            fs = pk_max_u(pk_subs_u(fs, 0x00020002u), hsgo);
            ff = pk_max_u(pk_subs_u(ff, 0x00020002u), hsgo);
            G[k] = pk_add(h, G_ZERO);
            lm = pk_max_u(lm, h);
        }
        diag_in = i0;
        p0 = G[R - 1];
        p1 = fs;
        p2 = ff;
        p3 = pk_max_u(i3, lm);
        // the first lane of this column to exceed everything above it, with a value its own strip never held: keep the strip
        const unsigned cand = pk_subs_u(lm, pk_max_u(i3, best));
        best = pk_max_u(best, lm);
        if (cand & 0xFFFFu) { col_a = c; keep(my_a, G); }
        if (cand >> 16) { col_b = c; keep(my_b, G); }
        // lane 63's column is complete: its maxima as wavefront-uniform values -- the running maxima, the end columns and the stop
        // rules below are scalar-unit work (one v_readlane instead of a block of vector instructions with one lane switched on)
        {
            const unsigned cm = (unsigned)__builtin_amdgcn_readlane((int)p3, 63);
            const int c63 = t - 63, cm_a = (int)(cm & 0xFFFFu), cm_b = (int)(cm >> 16);
            if (!stop_a && c63 >= 0 && c63 < a.count) {
                if (cm_a > run_a) {
                    run_a = cm_a;
                    if (a.lanes == 16 && run_a + BIAS >= 255) { ovf_a = 1; stop_a = 1; }
                    else endc_a = c63;
                }
                if (!stop_a && (cm_a == a.terminate || c63 == a.count - 1)) stop_a = 1;
            }
            if (!stop_b && c63 >= 0 && c63 < b.count) {
                if (cm_b > run_b) {
                    run_b = cm_b;
                    if (b.lanes == 16 && run_b + BIAS >= 255) { ovf_b = 1; stop_b = 1; }
                    else endc_b = c63;
                }
                if (!stop_b && (cm_b == b.terminate || c63 == b.count - 1)) stop_b = 1;
            }
        }
        if (stop_a & stop_b) break;
    }
    // the end cells' rows out of the holders' copies
    auto finish = [&](const HalfIn& h, int run, int endc, int ovf, int col, unsigned best_half, const uint32_t* mine, int shift, PassOut& o) {
        const int rm = run, ec = endc;                  // (wavefront-uniform: lane 63's view, read with v_readlane every step)
        o.overflow = ovf;
        o.score = o.overflow ? 255 : rm;
        o.ref = ec >= 0 ? h.first + ec * step : (h.lanes == 16 ? -1 : 0);
        int er = -1;
        if (rm > 0 && ec >= 0 && !o.overflow) {
            const unsigned long long holders = __ballot((int)best_half == rm && col == ec);
            if (!holders) o.overflow = 2;          // (never, unless the bookkeeping above is wrong: the read is failed, not mis-aligned)
            if (holders) {
                const int l = __ffsll((long long)holders) - 1;
                int kk = R;
                const unsigned want = (unsigned)(rm - 6) & 0xFFFFu;
#pragma unroll
                for (int k = R - 1; k >= 0; --k)
                    if (((mine[k] >> shift) & 0xFFFFu) == want) kk = k;
                er = l * R + __builtin_amdgcn_readlane(kk, l);
            }
        }
        o.read = h.m - 1;
        if (rm == 0) { if (h.m - 1 > 0) o.read = 0; }
        else if (er >= 0 && er < h.m - 1) o.read = er;
    };
    asm volatile("" ::: "memory");                 // (the copies above are read back through another pointer)
    finish(a, run_a, endc_a, ovf_a, col_a, best & 0xFFFFu, my_a, 0, oa);
    finish(b, run_b, endc_b, ovf_b, col_b, best >> 16, my_b, 16, ob);
}

struct JobPair { int32_t a, b; };            // indices into the job table; b < 0: read a alone

// rows per lane of a read's longer segmentation (the 8-bit one), rounded up to the strips the packed pass exists in
__device__ __host__ inline int strip_of(int m) { return (((((m + 15) / 16) * 16) + 63) >> 6); }

// One wavefront per pair of reads (JobPair): both through the 8-bit segmentation, those that overflow through the 16-bit one,
// those with a score through the reversed pass.  One kernel per strip size R (rows per lane): a launch takes the pairs whose
// longer read needs more than r_lo and at most r_hi = R rows per lane (one size per kernel keeps the register count at
// 4 R + ~110: two wavefronts per SIMD at every size, three up to 16 rows per lane -- measured no faster than two; all sizes in
// one kernel cost 342 registers).  LDS: two copies of a strip.
constexpr int pk_lds_bytes(int R) { return 2 * 64 * ((R + 3) & ~3) * 4; }
template <int R, int W>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(W, W))) void sw_ends_pair_kernel(Job* __restrict__ jobs, const JobPair* __restrict__ pairs, const int* __restrict__ n_pairs,
                                                          const int8_t* __restrict__ ref, const int8_t* __restrict__ seq, int r_lo, int r_hi, int adaptive) {
    extern __shared__ uint32_t snap[];
    if ((int)blockIdx.x >= n_pairs[0]) return;
    if (adaptive && n_pairs[1] != R) return;          // (pair_jobs_kernel chose the strip size of this call on the device: one kernel of the ladder runs)
    const JobPair pr = pairs[blockIdx.x];
    Job& JA = jobs[pr.a];
    Job& JB = jobs[pr.b >= 0 ? pr.b : pr.a];
    const bool two = pr.b >= 0;
    {
        const int ra = strip_of(JA.m), rb = two ? strip_of(JB.m) : 0, need = ra > rb ? ra : rb;
        if (need <= r_lo || need > r_hi) return;
    }
    const long long t_start = wall_clock64();
    // (the job fields are the same in every lane: kept in scalar registers, so that the passes' bounds tests are scalar too)
    auto uni = [](int v) { return __builtin_amdgcn_readfirstlane(v); };
    auto uni64 = [&](int64_t v) { return (int64_t)(((uint64_t)(uint32_t)uni((int)(v >> 32)) << 32) | (uint32_t)uni((int)v)); };
    const int na = uni(JA.n), ma = uni(JA.m), nb = uni(JB.n), mb = uni(JB.m);
    const int8_t* rfa = ref + uni(JA.ref_off);
    const int8_t* rda = seq + uni64(JA.seq_off);
    const int8_t* rfb = ref + uni(JB.ref_off);
    const int8_t* rdb = seq + uni64(JB.seq_off);
    // pass 0: the 8-bit segmentation; pass 1: the 16-bit one for the reads that left the 8-bit range; pass 2: the begin cells,
    // from the end cells backwards (one call site: the passes of every strip size are inlined into this loop once)
    PassOut fa = {0, 0, 0, 0}, fb = {0, 0, 0, 0}, o1 = {0, 0, 0, 0}, o2 = {0, 0, 0, 0};
    int wide_a = 0, wide_b = 0, bad = 0;
    int rbeg_a = -1, qbeg_a = -1, rbeg_b = -1, qbeg_b = -1;
    // a read whose 8-bit pass is proven to overflow (overflow_proven, by the job's builder) does not run it: the library throws that
    // pass away whole (ssw.c:819-824); a pair of two such reads starts with the 16-bit segmentation
    fa.overflow = uni(JA.wide) != 0;
    fb.overflow = two && uni(JB.wide) != 0;
    for (int pass = 0; pass < 3; ++pass) {
        bool on_a, on_b;
        if (pass == 0) { on_a = !fa.overflow; on_b = two && !fb.overflow; }
        else if (pass == 1) { on_a = fa.overflow == 1; on_b = two && fb.overflow == 1; }
        else { on_a = fa.score > 0 && fa.ref >= 0; on_b = two && fb.score > 0 && fb.ref >= 0; }
        if (!on_a && !on_b) continue;
        const bool rev = pass == 2;
        const HalfIn ha = {rfa, (rev && on_a) ? fa.ref : 0, !on_a ? 0 : (rev ? fa.ref + 1 : na), rda, (rev && on_a) ? fa.read : 0,
                           !on_a ? 0 : (rev ? fa.read + 1 : ma), (pass == 1 || (rev && wide_a)) ? 8 : 16, (rev && on_a) ? fa.score : -1};
        const HalfIn hb = {rfb, (rev && on_b) ? fb.ref : 0, !on_b ? 0 : (rev ? fb.ref + 1 : nb), rdb, (rev && on_b) ? fb.read : 0,
                           !on_b ? 0 : (rev ? fb.read + 1 : mb), (pass == 1 || (rev && wide_b)) ? 8 : 16, (rev && on_b) ? fb.score : -1};
        score_pass_pk<R>(snap, ha, hb, rev ? -1 : 1, rev ? -1 : 1, o1, o2);
        bad |= (on_a && o1.overflow == 2) | ((on_b && o2.overflow == 2) << 1);
        if (pass == 0) {
            if (on_a) fa = o1;
            if (on_b) fb = o2;
        } else if (pass == 1) {
            if (on_a) { fa = o1; wide_a = 1; }
            if (on_b) { fb = o2; wide_b = 1; }
        } else {
            if (on_a) { rbeg_a = o1.ref; qbeg_a = fa.read - o1.read; }
            if (on_b) { rbeg_b = o2.ref; qbeg_b = fb.read - o2.read; }
        }
    }
    if (threadIdx.x == 0) {
        const int32_t ticks = (int32_t)(wall_clock64() - t_start);
        if (bad & 1) JA.state = ST_ERR;
        if (two && (bad & 2)) JB.state = ST_ERR;
        JA.score = fa.score; JA.wide = wide_a; JA.ref_end = fa.ref; JA.read_end = fa.read; JA.ref_begin = rbeg_a; JA.read_begin = qbeg_a;
        JA.t_ends = ticks;
        if (two) {
            JB.score = fb.score; JB.wide = wide_b; JB.ref_end = fb.ref; JB.read_end = fb.read; JB.ref_begin = rbeg_b; JB.read_begin = qbeg_b;
            JB.t_ends = ticks;
        }
    }
}

// One launch per call (and a catch-all): the kernel of the strip size that holds the call's longest ordinary read takes every
// pair up to that size -- shorter reads leave lanes idle, which costs less than the tail of a launch of their own (measured:
// eight size-sorted launches 11.6 ms, one 7.7 ms for 8 000 reads) -- and the widest kernel takes what is longer.
// adaptive (the device-fed form, where the host knows only a bound of the reads' lengths): every size of the ladder up to that
// bound is launched and all but one return at once -- the one pair_jobs_kernel found to be the smallest that holds the call's
// longest read within the bound (pairs[0].b).  A step of the pass is R rows of instructions whatever the read's length: a
// 1 220-base region's reads fit 20 rows per lane, the bound (region + 10 % + 16) asked for 22.
// the strip sizes the packed pass is instantiated for: the smallest that holds `need` rows per lane
__host__ __device__ inline int ladder_size(int need) {
    return need <= 8 ? 8 : need <= 12 ? 12 : need <= 16 ? 16 : need <= 18 ? 18 : need <= 20 ? 20 : need <= 22 ? 22 : 24;
}
inline void launch_pair_kernels(hipStream_t st, int np, Job* dj, const JobPair* dp, const int* dn, const int8_t* dref, const int8_t* dseq, int need,
                                bool adaptive = false) {
    const int size = ladder_size(need);
#define PA_PAIR_SIZE(N)                                                                                                                          \
    if (adaptive ? N <= size : N == size)                                                                                                        \
        hipLaunchKernelGGL((sw_ends_pair_kernel<N, (N <= 16 ? 3 : 2)>), dim3((unsigned)np), dim3(64), pk_lds_bytes(N), st, dj, dp, dn, dref, dseq, 0, N, adaptive ? 1 : 0);
    PA_PAIR_SIZE(8) PA_PAIR_SIZE(12) PA_PAIR_SIZE(16) PA_PAIR_SIZE(18) PA_PAIR_SIZE(20) PA_PAIR_SIZE(22)
#undef PA_PAIR_SIZE
    hipLaunchKernelGGL((sw_ends_pair_kernel<24, 2>), dim3((unsigned)np), dim3(64), pk_lds_bytes(24), st, dj, dp, dn, dref, dseq, size == 24 ? (adaptive ? 22 : 0) : size, 24, 0);
}

// inclusive prefix sum across the wavefront (same DPP pattern as wave_prefix_max)
__device__ __forceinline__ int wave_prefix_add(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
    return v;
}
// One wavefront per workgroup: LDS instructions of a wavefront execute in issue order, so cross-lane hand-offs through
// LDS only need the compiler to keep the program order.
__device__ __forceinline__ void wave_lds_order() { asm volatile("" ::: "memory"); }

__device__ __forceinline__ int wave_max(int v) {
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return v;
}

// Three wavefronts per read try three consecutive band widths of the doubling sequence at once (the attempts are
// independent; the narrowest one that reaches the score is the library's); that wavefront alone walks back and writes
// the operations.  cap: ints per band array (the H row twice -- hb and hc, read and written in turn -- and eb; one set per
// wavefront); the bytes behind them hold the trace-back steps, the base codes of the aligned windows and the per-step operation classes (3 (m + n) + 4 bytes).
// ops_counter: running number of operations written to opsws.
// BAND_WAVES = 3 races three widths (the lowest latency per read: what a call of a few hundred reads wants); = 1 tries the
// widths one after the other in one wavefront (a third of the wavefronts, LDS and direction bytes, and no work on widths the
// narrower one made unnecessary: what a call that fills the chip anyway wants).  Either way an attempt stops as soon as it
// cannot reach the score any more: a later row gains at most one match over the best cell so far.
template <int BAND_WAVES>
__global__ __launch_bounds__(64 * BAND_WAVES) void band_kernel(Job* __restrict__ jobs, const int8_t* __restrict__ ref,
                                                               const int8_t* __restrict__ seq, uint8_t* __restrict__ dirws,
                                                               uint32_t* __restrict__ opsws,
                                                               unsigned long long* __restrict__ ops_counter, int cap) {
    extern __shared__ int sm[];
    Job& J = jobs[blockIdx.x];
    if (J.state != ST_BAND) return;
    const int lane = threadIdx.x & 63, wave = uni_value((int)(threadIdx.x >> 6));
    // (the read's fields as wavefront-uniform values: the row and chunk loops below branch on the scalar unit)
    const int n = uni_value(J.ref_end - J.ref_begin + 1), m = uni_value(J.read_end - J.read_begin + 1), score = uni_value(J.score);
    int* hb = sm + wave * 3 * cap;
    int* eb = hb + cap;
    int* hc = hb + 2 * cap;
    int* sh_best = sm + BAND_WAVES * 3 * cap;                 // [BAND_WAVES] result of each wavefront's attempt
    const int step_cap = m + n + 2;
    int8_t* lrf = reinterpret_cast<int8_t*>(sh_best + 4);
    int8_t* lrd = lrf + n;
    // the walk back's steps and the operation classes: written and read once per read, by the winning wavefront alone -- in
    // the workspace behind the read's direction bytes, not in LDS (LDS per read is what bounds the wavefronts per SIMD here)
    uint8_t* steps = dirws + J.dir_off + (size_t)BAND_WAVES * m * J.dir_width;
    uint8_t* cls = steps + step_cap;
    {
        const int8_t* rf = ref + J.ref_off + J.ref_begin;
        const int8_t* rd = seq + J.seq_off + J.read_begin;
        for (int k = threadIdx.x; k < n; k += 64 * BAND_WAVES) lrf[k] = rf[k];
        for (int k = threadIdx.x; k < m; k += 64 * BAND_WAVES) lrd[k] = rd[k];
    }
    constexpr int PENDING = -(1 << 30);
    volatile int* vbest = sh_best;
    if (threadIdx.x < BAND_WAVES) sh_best[threadIdx.x] = PENDING;
    __syncthreads();
    const int base_bw = uni_value(J.bw), dir_width = uni_value(J.dir_width);
    uint8_t* dir = dirws + J.dir_off + (size_t)wave * m * dir_width;
    int bw = 0, stride = 0;
    const long long t_start = wall_clock64();
    for (int round = 0;; ++round) {
        const int attempt = round * BAND_WAVES + wave;
        // results: >= 0 banded maximum; -1 the workspace rows are too narrow for this width; -2 the sequence does not
        // get here (an earlier width already covers the whole matrix: the library would double for ever)
        int result;
        bw = attempt < 24 ? base_bw << attempt : INT32_MAX / 8;
        const int width = 2 * bw + 3;
        stride = min(2 * bw + 1, n);
        const int slots = min(width, n + 2) + 1;
        if (attempt >= 24 || (attempt > 0 && (base_bw << (attempt - 1)) > n + m)) {
            result = -2;
        } else if (stride > dir_width || slots > cap) {
            result = -1;
        } else {
            for (int k = lane; k < slots; k += 64) { hb[k] = 0; eb[k] = 0; hc[k] = 0; }
            int best = 0;
            bool overtaken = false;                     // a narrower width of this round has reached the score already
            bool hopeless = false;
            for (int i = 0; i < m; ++i) {
                if (wave > 0 && (i & 7) == 0) {
                    for (int w = 0; w < wave; ++w) overtaken = overtaken || vbest[w] >= score;
                    if (overtaken) break;
                }
                if ((i & 15) == 0 && i > 0) {
                    // rows i .. m - 1 are still to come: each adds at most a match to the best cell of the rows before
                    if (wave_max(best) + S_MATCH * (m - i) < score) { hopeless = true; break; }
                }
                const int x = max(i - bw, 0), xp = max(i - 1 - bw, 0), sh = x - xp;
                const int end = min(n - 1, i + bw), U = end - x + 1, edge = min(end + 1, width - 1);
                // the H row is kept twice: row i reads `hp` (row i - 1) and writes `hw`, and the two swap -- a chunk may not
                // overwrite slots the next chunk still reads, and copying the row back cost a pass over it per row
                int* const hp = (i & 1) ? hc : hb;
                int* const hw = (i & 1) ? hb : hc;
                wave_lds_order();
                if (lane == 0) { hp[0] = 0; eb[0] = 0; hp[edge] = 0; eb[edge] = 0; }
                wave_lds_order();
                const int qi = lrd[i];
                int carry_a = NEG, carry_h = 0, carry_f = 0;
                uint8_t* drow = dir + (size_t)i * stride;
                for (int base = 0; base < U; base += 64) {
                    const int u = 1 + base + lane;
                    const bool valid = u <= U;
                    // (lanes past the row's last slot read its last slot: no branch around the loads; what they compute goes
                    // nowhere -- their term of the chain below is the identity, they store nothing, and the carries are read
                    // from lane 63 only when the next chunk exists, i.e. when lane 63 is a slot of the row)
                    const int uc = min(u, U);
                    const int hbe = hp[uc + sh], ebe = eb[uc + sh], hbd = hp[uc + sh - 1], rj = lrf[x + uc - 1];
                    const int t1 = hbe - GO, t2 = ebe - GE;        // (row 0 reads the zeros both rows start from: -GO, -GE)
                    const int ecur = max(t1, t2), de = t1 > t2;
                    const int diag = hbd + ((rj == qi && qi < 4) ? S_MATCH : -S_MIS);
                    const int e1 = max(ecur, 0), g = max(e1, diag);
                    // vertical-gap chain of the row: f(u) = max(-GE u, max_{v<u} (g(v) + GE v) - GO - GE (u - 1))
                    const int pm = wave_prefix_max_min(valid ? g + u * GE : NEG);
                    const int ex = max(dpp_up1_or(pm, (int)0x80000000u), carry_a);       // (lane 0: carry_a, which is >= NEG)
                    const int f = max(-GE * u, ex - GO - (u - 1) * GE);
                    const int f1 = max(f, 0), hcur = max(g, f1);
                    const int hl = dpp_up1_or(hcur, carry_h), fl = dpp_up1_or(f, carry_f);
                    const int df = (hl - GO) > (fl - GE);
                    const int gap = max(e1, f1);
                    const bool from_e = e1 > f1;                   // (selects, not branches: the code of the cell's source)
                    const int dh = gap <= diag ? 1 : (from_e ? 2 : 4) + ((from_e ? de : df) ? 1 : 0);
                    wave_lds_order();                 // every lane has read the previous row's slots of this chunk
                    if (valid) {
                        eb[u] = ecur;
                        hw[u] = hcur;
                        drow[u - 1] = (uint8_t)(de | (df << 1) | (dh << 2));
                        best = max(best, hcur);
                    }
                    if (base + 64 < U) {
                        carry_a = max(carry_a, bcast63(pm));
                        carry_h = bcast63(hcur);
                        carry_f = bcast63(f);
                    }
                }
            }
            result = overtaken ? -3 : (hopeless ? 0 : wave_max(best));
        }
        if (lane == 0) vbest[wave] = result;
        __syncthreads();
        int winner = -1, verdict = 0;                 // verdict: 1 wider workspace needed, 2 no width reaches the score
        int need_bw = 0;
        for (int w = 0; w < BAND_WAVES; ++w) {
            const int r = sh_best[w];
            if (r >= score) { winner = w; break; }
            if (r == -1) { verdict = 1; need_bw = base_bw << (round * BAND_WAVES + w); break; }
            if (r == -2) { verdict = 2; break; }
        }
        if (verdict) {
            if (threadIdx.x == 0) {
                if (verdict == 1) { J.bw = need_bw; J.state = ST_WIDER; }
                else J.state = ST_ERR;
            }
            return;
        }
        if (winner >= 0) {
            if (wave != winner) return;
            break;
        }
        __syncthreads();                              // sh_best is rewritten by the next round
        if (lane == 0) sh_best[wave] = PENDING;
        __syncthreads();
    }
    __threadfence();
    const long long t_dp = wall_clock64();

    // trace back (ssw.c:653-703): state 2 = H, 0 = E, 1 = F.  Runs of diagonal moves are found 64 cells at a time (every
    // lane probes one cell of the diagonal), gap cells one by one.
    int i = m - 1, j = n - 1, state = 2, ns = 0;
    bool bad = false;
    while (i > 0) {
        int code;
        if (state == 2) {
            const int ii = i - lane, jj = j - lane;
            int probe = 0;
            if (ii > 0) {
                const int off = jj - max(ii - bw, 0);
                if (off >= 0 && off < stride) probe = dir[(size_t)ii * stride + off] >> 2;
            }
            const unsigned long long other = __ballot(probe != 1);
            const int run = other ? (int)__builtin_ctzll(other) : 64;
            if (run > 0) {
                if (ns + run > step_cap) { bad = true; break; }
                if (lane < run) steps[ns + lane] = 0;
                ns += run; i -= run; j -= run;
                if (i <= 0) break;
                if (run == 64) continue;
            }
            code = __shfl(probe, run, 64);
            if (code == 0) { bad = true; break; }
        } else {
            const int off = j - max(i - bw, 0);
            if (off < 0 || off >= stride) { bad = true; break; }
            const int d = dir[(size_t)i * stride + off];
            code = state == 0 ? ((d & 1) ? 3 : 2) : ((d & 2) ? 5 : 4);
        }
        int op = 0;
        switch (code) {
            case 1: --i; --j; state = 2; op = 0; break;
            case 2: --i; state = 0; op = OP_I; break;
            case 3: --i; state = 2; op = OP_I; break;
            case 4: --j; state = 1; op = OP_D; break;
            case 5: --j; state = 2; op = OP_D; break;
            default: bad = true; break;
        }
        if (bad || ns >= step_cap) { bad = true; break; }
        if (lane == 0) steps[ns] = (uint8_t)op;
        ++ns;
    }
    wave_lds_order();
    if (bad) {
        if (lane == 0) J.state = ST_ERR;
        return;
    }
    __threadfence();                                  // (steps: written by some lanes, read by others below)
    const long long t_trace = wall_clock64();

    // Operations in alignment order: soft clip, the first cell, the steps backwards, soft clip; aligned pairs are
    // classified by comparing base codes from the begin cell on (ssw_cpp.cpp:126-207).  Wavefront-parallel: positions
    // from prefix sums, classes to LDS, run starts counted, output range reserved, runs written, lengths filled in.
    const int total = ns + 1;                                  // cells of the path, the begin cell first
    int carry_r = 0, carry_q = 0;
    for (int t0 = 0; t0 < total; t0 += 64) {
        const int t = t0 + lane;
        const bool valid = t < total;
        const int op = (!valid || t == 0) ? 0 : steps[ns - t];
        const int cr = valid && op != OP_I, cq = valid && op != OP_D;
        const int pr = wave_prefix_add(cr), pq = wave_prefix_add(cq);
        if (valid) {
            const int rp = carry_r + pr - cr, qp = carry_q + pq - cq;
            cls[t] = (uint8_t)(op == 0 ? (lrf[rp] == lrd[qp] ? OP_EQ : OP_X) : op);
        }
        carry_r += bcast63(pr);
        carry_q += bcast63(pq);
    }
    wave_lds_order();
    __threadfence();                                  // (cls likewise)
    int nruns = 0;
    for (int t0 = 0; t0 < total; t0 += 64) {
        const int t = t0 + lane;
        const bool start = t < total && (t == 0 || cls[t] != cls[t - 1]);
        nruns += __popcll(__ballot(start));
    }
    const int lead = J.read_begin > 0, trail = J.m - J.read_end - 1 > 0, no = lead + nruns + trail;
    unsigned long long at = 0;
    if (lane == 0) at = atomicAdd(ops_counter, (unsigned long long)no);
    at = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(at >> 32)) << 32) |
         (unsigned)__builtin_amdgcn_readfirstlane((int)(at & 0xffffffffu));
    uint32_t* ops = opsws + at;
    int carry_s = 0;
    for (int t0 = 0; t0 < total; t0 += 64) {
        const int t = t0 + lane;
        const int start = t < total && (t == 0 || cls[t] != cls[t - 1]);
        const int ps = wave_prefix_add(start);
        if (start) ops[lead + carry_s + ps - 1] = (uint32_t)cls[t] | ((uint32_t)t << 4);     // run start for now
        carry_s += bcast63(ps);
    }
    __threadfence();
    for (int k0 = 0; k0 < nruns; k0 += 64) {
        const int k = k0 + lane;
        uint32_t a = 0, next = (uint32_t)total;
        if (k < nruns) {
            a = ops[lead + k];
            if (k + 1 < nruns) next = ops[lead + k + 1] >> 4;
        }
        wave_lds_order();
        if (k < nruns) ops[lead + k] = (a & 15u) | ((next - (a >> 4)) << 4);
    }
    if (lane == 0) {
        if (lead) ops[0] = (uint32_t)OP_S | ((uint32_t)J.read_begin << 4);
        if (trail) ops[no - 1] = (uint32_t)OP_S | ((uint32_t)(J.m - J.read_end - 1) << 4);
        J.ops_off = (int64_t)at;
        J.n_ops = no;
        J.bw = bw;
        J.state = ST_DONE;
        J.t_dp = (int32_t)(t_dp - t_start);
        J.t_trace = (int32_t)(t_trace - t_dp);
        J.t_emit = (int32_t)(wall_clock64() - t_trace);
    }
}

// wavefronts per read of the band stage: three widths at once for a call that does not fill the chip on its own, one after the
// other beyond that (PA_BAND_WAVES overrides)
// slots per row of the direction bytes a read gets first: half width <= 64 for three wavefronts per read; <= 128 for one (its one
// copy of the rows costs less than the three narrow ones, and the reads that outgrow it -- 1.5 % outgrew 64 on nanopore-like
// data -- each cost a launch that waits for a few 1 200-slot wavefronts)
inline int first_band_width(int nw) { return nw == 1 ? 257 : 129; }
inline int band_waves_for(int n_reads) {
    static const int forced = getenv("PA_BAND_WAVES") ? atoi(getenv("PA_BAND_WAVES")) : 0;
    if (forced == 1 || forced == 3) return forced;
    return n_reads >= 3072 ? 1 : 3;
}
inline hipError_t launch_band(hipStream_t st, int nw, int n_reads, size_t lds, Job* dj, const int8_t* dref, const int8_t* dseq, uint8_t* dir,
                              uint32_t* ops, unsigned long long* counter, int cap) {
    if (lds > 64 * 1024) {
        const hipError_t e = nw == 1 ? hipFuncSetAttribute(reinterpret_cast<const void*>(band_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
                                     : hipFuncSetAttribute(reinterpret_cast<const void*>(band_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    if (nw == 1) hipLaunchKernelGGL(band_kernel<1>, dim3(n_reads), dim3(64), lds, st, dj, dref, dseq, dir, ops, counter, cap);
    else hipLaunchKernelGGL(band_kernel<3>, dim3(n_reads), dim3(192), lds, st, dj, dref, dseq, dir, ops, counter, cap);
    return hipGetLastError();
}

struct DBuf {
    void* p = nullptr;
    size_t bytes = 0;
    bool ensure(size_t need) {
        if (need <= bytes) return true;
        static const bool trace = getenv("PA_TRACE_ALLOC") != nullptr;
        if (trace) fprintf(stderr, "[alloc] re-aligner buffer %zu -> %zu bytes\n", bytes, need + need / 4 + 256);
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
        const size_t grow = need + need / 4 + 256;
        if (hipMalloc(&p, grow) != hipSuccess) return false;
        bytes = grow;
        return true;
    }
    ~DBuf() { if (p) (void)hipFree(p); }
};

}  // namespace

struct pa_realigner {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    DBuf d_ref, d_seq, d_jobs, d_dir, d_ops, d_counter, d_pairs, d_order;
    std::vector<JobPair> pairs;                                  // host-fed form: the reads two by two, like sizes together
    pa_enc::HBuf h_meta, h_back;                                 // device-fed form: window text + tables up, counters back
    std::vector<Job> jobs;
    std::vector<uint32_t> ops;
    int64_t total_ops = 0;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};     // around the score kernel, around the band launches
    double ends_ms = 0.0, band_ms = 0.0;
    int64_t cells = 0;                                           // DP cells of the score passes of the last call (n x m per read)
};

#define RA_HIP(expr)                                                                                    \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) return pa::set_error(PA_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)
#define RA_ALLOC(buf, bytes_)                                                                           \
    do {                                                                                                \
        if (!(buf).ensure(bytes_)) return pa::set_error(PA_ERR_HIP, "hipMalloc failed in re-aligner workspace"); \
    } while (0)

namespace {
int band_rounds_host(pa_realigner* r, Job* dj, const int8_t* dref, const int8_t* dseq, int32_t n_reads, int first_round, int aux, int nw);
}

extern "C" {

int pa_realigner_create(int32_t device, void* hip_stream, pa_realigner** out) {
    if (!out) return pa::set_error(PA_ERR_INVALID, "null argument");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return pa::set_error(PA_ERR_NO_DEVICE, "no HIP device visible: the pepper_amd re-aligner has no CPU fallback");
    if (device < 0 || device >= count) return pa::set_error(PA_ERR_INVALID, "device ordinal out of range");
    RA_HIP(hipSetDevice(device));
    auto* r = new pa_realigner();
    r->device = device;
    if (hip_stream) r->stream = static_cast<hipStream_t>(hip_stream);
    else {
        if (hipStreamCreateWithFlags(&r->stream, hipStreamNonBlocking) != hipSuccess) {
            delete r;
            return pa::set_error(PA_ERR_HIP, "hipStreamCreate failed");
        }
        r->own_stream = true;
    }
    for (auto& e : r->ev)
        if (hipEventCreate(&e) != hipSuccess) { pa_realigner_destroy(r); return pa::set_error(PA_ERR_HIP, "hipEventCreate failed"); }
    *out = r;
    return PA_OK;
}

void pa_realigner_destroy(pa_realigner* r) {
    if (!r) return;
    (void)hipSetDevice(r->device);
    if (r->stream) (void)hipStreamSynchronize(r->stream);
    for (auto& e : r->ev)
        if (e) (void)hipEventDestroy(e);
    if (r->own_stream && r->stream) (void)hipStreamDestroy(r->stream);
    delete r;
}

int pa_realigner_align(pa_realigner* r, const char* reference, int64_t reference_len, int64_t region_start,
                       int32_t n_reads, const int64_t* read_pos, const int64_t* seq_offset, const char* seq,
                       int32_t* status, int32_t* sw_score, int64_t* new_pos, int64_t* new_pos_end,
                       int32_t* query_begin, int32_t* query_end, int64_t* n_cigar_ops) {
    if (reference_len < 0) return pa::set_error(PA_ERR_INVALID, "null argument");
    const int64_t offsets[2] = {0, reference_len};
    return pa_realigner_align_windows(r, 1, reference, offsets, &region_start, n_reads, nullptr, read_pos, seq_offset, seq,
                                      status, sw_score, new_pos, new_pos_end, query_begin, query_end, n_cigar_ops);
}

int pa_realigner_align_windows(pa_realigner* r, int32_t n_windows, const char* reference, const int64_t* window_offset,
                               const int64_t* window_start, int32_t n_reads, const int32_t* read_window,
                               const int64_t* read_pos, const int64_t* seq_offset, const char* seq, int32_t* status,
                               int32_t* sw_score, int64_t* new_pos, int64_t* new_pos_end, int32_t* query_begin,
                               int32_t* query_end, int64_t* n_cigar_ops) {
    if (!r || !n_cigar_ops || n_reads < 0 || n_windows <= 0 || !window_offset || !window_start ||
        (n_reads > 0 && (!reference || !read_pos || !seq_offset || !seq || !status || !sw_score || !new_pos || !new_pos_end)))
        return pa::set_error(PA_ERR_INVALID, "null argument");
    for (int32_t w = 0; w < n_windows; ++w)
        if (window_offset[w + 1] < window_offset[w] || window_offset[0] != 0)
            return pa::set_error(PA_ERR_INVALID, "window_offset must start at 0 and be monotonic");
    const int64_t reference_len = window_offset[n_windows];
    if (reference_len > (int64_t)1 << 30) return pa::set_error(PA_ERR_INVALID, "reference window too long");
    RA_HIP(hipSetDevice(r->device));
    static const bool trace = getenv("PA_REALIGN_TRACE") != nullptr;      // host phase times on stderr
    auto t_prev = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!trace) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[realign] %-18s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
        t_prev = now;
    };
    r->jobs.assign((size_t)n_reads, Job());
    r->total_ops = 0;
    r->ends_ms = r->band_ms = 0.0;
    r->cells = 0;
    *n_cigar_ops = 0;
    if (n_reads == 0) return PA_OK;

    const int64_t total_seq = seq_offset[n_reads];
    if (total_seq < 0 || total_seq > (int64_t)1 << 31) return pa::set_error(PA_ERR_INVALID, "bad seq_offset");
    int max_m = 1;
    bool any = false;
    for (int32_t k = 0; k < n_reads; ++k) {
        Job& J = r->jobs[(size_t)k];
        const int32_t w = read_window ? read_window[k] : 0;
        if (w < 0 || w >= n_windows) return pa::set_error(PA_ERR_INVALID, "read " + std::to_string(k) + ": bad window index");
        const int64_t m = seq_offset[k + 1] - seq_offset[k], off = read_pos[k] - window_start[w];
        const int64_t window_len = window_offset[w + 1] - window_offset[w];
        if (m < 0) return pa::set_error(PA_ERR_INVALID, "seq_offset is not monotonic");
        J.seq_off = seq_offset[k];
        J.m = (int32_t)std::min<int64_t>(m, INT32_MAX);
        J.ref_begin = J.read_begin = -1;
        if (off < 0) { J.state = ST_DROPPED; continue; }
        if (off > window_len)
            return pa::set_error(PA_ERR_INVALID, "read " + std::to_string(k) + " starts beyond the reference window");
        J.ref_off = (int32_t)(window_offset[w] + off);
        J.n = (int32_t)(window_len - off);
        if (m == 0 || J.n == 0) { J.state = ST_KEPT; continue; }
        if (m > MAX_READ)
            return pa::set_error(PA_ERR_INVALID, "read " + std::to_string(k) + " has " + std::to_string(m) +
                                 " bases: the re-aligner handles region-clipped reads up to " + std::to_string(MAX_READ));
        J.state = ST_NEW;
        r->cells += (int64_t)J.n * J.m;
        max_m = std::max(max_m, J.m);
        any = true;
    }

    auto finish_outputs = [&]() {
        for (int32_t k = 0; k < n_reads; ++k) {
            const Job& J = r->jobs[(size_t)k];
            status[k] = J.state == ST_DONE ? PA_REALIGN_ALIGNED : (J.state == ST_DROPPED ? PA_REALIGN_DROPPED : PA_REALIGN_KEPT);
            sw_score[k] = J.score;
            new_pos[k] = J.state == ST_DONE ? read_pos[k] + J.ref_begin : read_pos[k];
            new_pos_end[k] = J.state == ST_DONE ? read_pos[k] + J.ref_end : -1;
            if (query_begin) query_begin[k] = J.read_begin;
            if (query_end) query_end[k] = J.state == ST_DONE ? J.read_end : -1;
        }
    };
    if (!any) { finish_outputs(); return PA_OK; }

    lap("job table");
    // upload the text, turn it into base codes on the device
    RA_ALLOC(r->d_ref, (size_t)reference_len + 64);
    RA_ALLOC(r->d_seq, (size_t)total_seq + 64);
    RA_ALLOC(r->d_jobs, sizeof(Job) * (size_t)n_reads);
    RA_HIP(hipMemcpyAsync(r->d_ref.p, reference, (size_t)reference_len, hipMemcpyHostToDevice, r->stream));
    RA_HIP(hipMemcpyAsync(r->d_seq.p, seq, (size_t)total_seq, hipMemcpyHostToDevice, r->stream));
    RA_HIP(hipMemcpyAsync(r->d_jobs.p, r->jobs.data(), sizeof(Job) * (size_t)n_reads, hipMemcpyHostToDevice, r->stream));
    hipLaunchKernelGGL(to_codes_kernel, dim3((unsigned)((reference_len + 255) / 256)), dim3(256), 0, r->stream,
                       static_cast<int8_t*>(r->d_ref.p), reference_len);
    hipLaunchKernelGGL(to_codes_kernel, dim3((unsigned)((total_seq + 255) / 256)), dim3(256), 0, r->stream,
                       static_cast<int8_t*>(r->d_seq.p), total_seq);
    RA_HIP(hipGetLastError());
    Job* dj = static_cast<Job*>(r->d_jobs.p);
    const int8_t* dref = static_cast<const int8_t*>(r->d_ref.p);
    const int8_t* dseq = static_cast<const int8_t*>(r->d_seq.p);
    {
        // LDS only for reads too long for the register-resident pass (one dword per padded row)
        const int rows = ((max_m + 15) / 16) * 16, R = (rows + 63) / 64;
        const size_t lds = rows > REG_ROWS ? (size_t)64 * R * 4 : 0;
        RA_HIP(hipEventRecord(r->ev[0], r->stream));
        // two reads per wavefront halve the instructions per cell but leave half the wavefronts: a call that cannot fill the 1 024
        // SIMDs even one read per wavefront (a region on its own) keeps the one-read kernels, whose wavefronts finish sooner
        static const char* mode = getenv("PA_REALIGN_SINGLE");                   // "1": always one read per wavefront, "0": never (A/B runs)
        const bool single = mode ? mode[0] == '1' : n_reads <= 2048;
        if (single) {
            // the instantiation whose register strip just covers the longest read of the call (fewer registers: more wavefronts
            // per SIMD); a read beyond 24 rows per lane takes the LDS form inside the widest one
            if (R <= 12) hipLaunchKernelGGL(sw_ends_kernel<12>, dim3(n_reads), dim3(64), lds, r->stream, dj, dref, dseq, 0, 1 << 30);
            else if (R <= 16) hipLaunchKernelGGL(sw_ends_kernel<16>, dim3(n_reads), dim3(64), lds, r->stream, dj, dref, dseq, 0, 1 << 30);
            else if (R <= 20) hipLaunchKernelGGL(sw_ends_kernel<20>, dim3(n_reads), dim3(64), lds, r->stream, dj, dref, dseq, 0, 1 << 30);
            else hipLaunchKernelGGL(sw_ends_kernel<24>, dim3(n_reads), dim3(64), lds, r->stream, dj, dref, dseq, 0, 1 << 30);
        } else {
            // two reads per wavefront (score_pass_pk), like sizes together; reads beyond 24 rows per lane one by one in the LDS form
            std::vector<int32_t> order;
            for (int32_t k = 0; k < n_reads; ++k)
                if (r->jobs[(size_t)k].state == ST_NEW && strip_of(r->jobs[(size_t)k].m) <= 24) order.push_back(k);
            std::sort(order.begin(), order.end(), [&](int32_t x, int32_t y) {
                const Job &a = r->jobs[(size_t)x], &b = r->jobs[(size_t)y];
                return a.n != b.n ? a.n < b.n : (a.m != b.m ? a.m < b.m : x < y);
            });
            r->pairs.assign(1 + (order.size() + 1) / 2, JobPair{0, 0});
            const int32_t np = (int32_t)((order.size() + 1) / 2);
            r->pairs[0] = JobPair{np, 0};                                        // [0].a = the number of pairs (read by the kernels)
            for (int32_t i = 0; i < np; ++i)
                r->pairs[(size_t)i + 1] = JobPair{order[(size_t)2 * i], (size_t)2 * i + 1 < order.size() ? order[(size_t)2 * i + 1] : -1};
            RA_ALLOC(r->d_pairs, sizeof(JobPair) * r->pairs.size());
            RA_HIP(hipMemcpyAsync(r->d_pairs.p, r->pairs.data(), sizeof(JobPair) * r->pairs.size(), hipMemcpyHostToDevice, r->stream));
            const JobPair* dp = static_cast<const JobPair*>(r->d_pairs.p);
            const int* dn = reinterpret_cast<const int*>(dp);
            if (np > 0) {
                int hi = 0;
                for (int32_t k : order) hi = std::max(hi, strip_of(r->jobs[(size_t)k].m));
                launch_pair_kernels(r->stream, np, dj, dp + 1, dn, dref, dseq, hi);
            }
            if (R > 24) hipLaunchKernelGGL(sw_ends_kernel<24>, dim3(n_reads), dim3(64), lds, r->stream, dj, dref, dseq, 24, 1 << 30);
        }
        RA_HIP(hipGetLastError());
        RA_HIP(hipEventRecord(r->ev[1], r->stream));
    }
    lap("upload + launch");
    RA_HIP(hipMemcpyAsync(r->jobs.data(), dj, sizeof(Job) * (size_t)n_reads, hipMemcpyDeviceToHost, r->stream));
    RA_HIP(hipStreamSynchronize(r->stream));
    lap("score kernel");

    // band stage: workspace layout, first with rows of at most 129 slots (band half width <= 64), then full rows
    int64_t ops_total = 0;
    int aux = 0;
    for (Job& J : r->jobs) {
        if (J.state != ST_NEW) continue;
        if (J.score <= 1 || J.ref_begin < 0) { J.state = ST_KEPT; continue; }     // simple_aligner.cpp:85
        const int n2 = J.ref_end - J.ref_begin + 1, m2 = J.read_end - J.read_begin + 1;
        J.state = ST_BAND;
        J.bw = std::abs(n2 - m2) + 1;
        J.ops_cap = n2 + m2 + 4;
        ops_total += J.ops_cap;
        aux = std::max(aux, n2 + m2 + 16);                 // (the two base-code windows; steps and classes live in the workspace)
    }
    RA_ALLOC(r->d_ops, sizeof(uint32_t) * (size_t)std::max<int64_t>(ops_total, 1));
    RA_ALLOC(r->d_counter, 8);
    RA_HIP(hipMemsetAsync(r->d_counter.p, 0, 8, r->stream));
    {
        const int rc = band_rounds_host(r, dj, dref, dseq, n_reads, 0, aux, band_waves_for(n_reads));
        if (rc != PA_OK) return rc;
        lap("band stage");
    }
    {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, r->ev[0], r->ev[1]) == hipSuccess) r->ends_ms = ms;
    }
    for (int32_t k = 0; k < n_reads; ++k) {
        const Job& J = r->jobs[(size_t)k];
        if (J.state == ST_ERR || J.state == ST_BAND || J.state == ST_WIDER)
            return pa::set_error(PA_ERR_INVALID, "read " + std::to_string(k) + ": the band stage did not reach the alignment score "
                                 "(the reference library aborts on such an alignment)");
    }
    unsigned long long written = 0;
    RA_HIP(hipMemcpyAsync(&written, r->d_counter.p, 8, hipMemcpyDeviceToHost, r->stream));
    RA_HIP(hipStreamSynchronize(r->stream));
    r->ops.resize((size_t)written);
    if (written)
        RA_HIP(hipMemcpyAsync(r->ops.data(), r->d_ops.p, sizeof(uint32_t) * (size_t)written, hipMemcpyDeviceToHost, r->stream));
    RA_HIP(hipStreamSynchronize(r->stream));
    for (const Job& J : r->jobs)
        if (J.state == ST_DONE) r->total_ops += J.n_ops;
    *n_cigar_ops = r->total_ops;
    finish_outputs();
    lap("results");
    return PA_OK;
}

int pa_realigner_last_timing(pa_realigner* r, double* score_kernel_ms, double* band_kernel_ms, int64_t* cells) {
    if (!r) return pa::set_error(PA_ERR_INVALID, "null argument");
    if (score_kernel_ms) *score_kernel_ms = r->ends_ms;
    if (band_kernel_ms) *band_kernel_ms = r->band_ms;
    if (cells) *cells = r->cells;
    return PA_OK;
}

int pa_realigner_stage_ticks(pa_realigner* r, int32_t* ticks4) {
    if (!r || !ticks4) return pa::set_error(PA_ERR_INVALID, "null argument");
    for (size_t k = 0; k < r->jobs.size(); ++k) {
        const Job& J = r->jobs[k];
        ticks4[4 * k] = J.t_ends; ticks4[4 * k + 1] = J.t_dp; ticks4[4 * k + 2] = J.t_trace; ticks4[4 * k + 3] = J.t_emit;
    }
    return PA_OK;
}

int pa_realigner_copy_cigars(pa_realigner* r, int32_t collapse_eqx, int64_t* cigar_offset, int32_t* cigar_op,
                             int32_t* cigar_len) {
    if (!r || !cigar_offset || (r->total_ops > 0 && (!cigar_op || !cigar_len))) return pa::set_error(PA_ERR_INVALID, "null argument");
    int64_t at = 0;
    for (size_t k = 0; k < r->jobs.size(); ++k) {
        const Job& J = r->jobs[k];
        cigar_offset[k] = at;
        if (J.state != ST_DONE) continue;
        for (int32_t o = 0; o < J.n_ops; ++o) {
            const uint32_t w = r->ops[(size_t)(J.ops_off + o)];
            int32_t op = (int32_t)(w & 15u);
            if (collapse_eqx && (op == OP_EQ || op == OP_X)) op = 0;
            cigar_op[at] = op;
            cigar_len[at] = (int32_t)(w >> 4);
            ++at;
        }
    }
    cigar_offset[r->jobs.size()] = at;
    return PA_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------------
// The device-fed form (encoder_common.h, pa_ra::align_device / apply_device): the reads are where unpack_clip_kernel left
// them (ReadRec table, bases as text), the job table is built and the band stage laid out by kernels, and the results --
// job table and compacted operations -- stay on the device for the summary encoder.  One wait for the common case (counters
// after the first band launch); reads whose band outgrows the first rows (half width > 64) go through the host-laid-out
// rounds of pa_realigner_align_windows afterwards.
namespace {

enum { DC_TOO_LONG = 0, DC_BAD = 1, DC_UNFINISHED = 2, DC_ERR = 3, DC_ALIGNED = 4, DC_PROVEN = 5, DC_N = 8 };     // int32 counters; [6..7] = dir bytes (u64)

// one wave per read: its kept bases (text) -> codes at the same offsets
__global__ __launch_bounds__(256) void codes_of_reads_kernel(const pa_enc::ReadRec* __restrict__ reads, int n_reads,
                                                             const char* __restrict__ text, int8_t* __restrict__ codes) {
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (k >= n_reads) return;
    const pa_enc::ReadRec rd = reads[k];
    if (!(rd.flags & pa_enc::READ_MAPQ_OK)) return;
    for (int i = lane; i < rd.slen; i += 64) {
        const int c = text[rd.s0 + i] & 0xdf;
        codes[rd.s0 + i] = (int8_t)(c == 'A' || c == 'U' ? 0 : c == 'C' ? 1 : c == 'G' ? 2 : c == 'T' ? 3 : 4);
    }
}

// Does the 8-bit pass of this read overflow for certain?  (ssw.c:819-824: a running maximum >= 249 sends the read through the
// 16-bit pass and the 8-bit pass's results are thrown away -- ~6 % of the score stage for nothing, on every long read.)  The BAM
// alignment the read came with is a path through the matrix, and the pass's cell values are bounded from below along ANY path:
//   diagonal  h(i, j) >= hs(i, j) >= h(i-1, j-1) + s(i, j)     (the cell's first term; s = +4 / -6)
//   rows      h(i+k, j) >= h(i, j) - 8 - 2 (k - 1)             (the exact vertical chain ff, which no segment start resets)
//   columns   hs(i, j+k) >= hs(i, j) - 8 - 2 (k - 1)           (E opens from hs: the library's segment-local value, so the bound
//                                                                is carried for hs, and a row move leaves it at 0)
// and every value is >= 0.  A path value >= 249 means a column maximum >= 249, which is the overflow; the walk stops there
// (~75 bases into an ordinary read).  One thread per read: op codes M 0, I 1, D 2, N 3, S 4, = 7, X 8 as unpack_clip_kernel kept them
// (the first kept operation is an M that starts at the read's first base and the window column the job starts at).
__device__ bool overflow_proven(const int32_t* __restrict__ cigar_op, const int32_t* __restrict__ cigar_len, int c0, int ncig,
                                const int8_t* __restrict__ read, int m, const int8_t* __restrict__ ref, int n) {
    constexpr int LIMIT = 255 - BIAS;
    if (S_MATCH * min(m, n) < LIMIT) return false;
    int lh = 0, lhs = 0, i = 0, j = 0;                 // bounds of h and hs at the path's last cell (i - 1, j - 1)
    for (int o = 0; o < ncig; ++o) {
        const int op = cigar_op[c0 + o], len = cigar_len[c0 + o];
        if (len <= 0) continue;
        if (op == 0 || op == 7 || op == 8) {
            const int k_end = min(len, min(m - i, n - j));
            for (int k = 0; k < k_end; ++k) {
                const int q = read[i + k], c = ref[j + k];
                lh = max(lh + ((q == c && (unsigned)q < 4u) ? S_MATCH : -S_MIS), 0);
                if (lh >= LIMIT) return true;
            }
            lhs = lh;
            i += len; j += len;
        } else if (op == 1 || op == 4) {
            lh = max(lh - GO - (len - 1) * GE, 0);
            lhs = 0;
            i += len;
        } else if (op == 2 || op == 3) {
            lhs = max(lhs - GO - (len - 1) * GE, 0);
            lh = lhs;
            j += len;
        } else {
            lh = lhs = 0;                              // (nothing unpack_clip_kernel keeps; claim nothing)
        }
        if (i >= m || j >= n) break;
    }
    return false;
}

// cigar_op / cigar_len: the reads' clipped BAM alignments (ReadRec.c0 / ncig), or null: no read is proven
__global__ __launch_bounds__(256) void jobs_of_reads_kernel(const pa_enc::ReadRec* __restrict__ reads, int n_reads,
                                                            const int64_t* __restrict__ window_off, const int32_t* __restrict__ window_len,
                                                            Job* __restrict__ jobs, int* __restrict__ counters, int max_m,
                                                            const int32_t* __restrict__ cigar_op, const int32_t* __restrict__ cigar_len,
                                                            const int8_t* __restrict__ ref, const int8_t* __restrict__ seq) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n_reads) return;
    const pa_enc::ReadRec rd = reads[k];
    Job J;
    J.ref_off = 0; J.n = 0; J.m = rd.slen; J.state = ST_KEPT; J.seq_off = rd.s0;
    J.score = 0; J.wide = 0; J.ref_begin = -1; J.ref_end = 0; J.read_begin = -1; J.read_end = 0;
    J.bw = 0; J.dir_width = 0; J.dir_off = 0; J.ops_off = 0; J.ops_cap = 0; J.n_ops = 0;
    J.t_ends = J.t_dp = J.t_trace = J.t_emit = 0;
    if ((rd.flags & pa_enc::READ_MAPQ_OK) && rd.slen > 0) {      // (a read the summary skips is not aligned: nothing reads its CIGAR)
        const int wl = window_len[rd.region], off = rd.row0;
        if (off < 0) J.state = ST_DROPPED;                          // simple_aligner.cpp:72-76 (a clipped read never starts there)
        else if (off > wl) atomicMax(&counters[DC_BAD], k + 1);
        else if (rd.slen > max_m) atomicMax(&counters[DC_TOO_LONG], k + 1);
        else {
            J.ref_off = (int32_t)(window_off[rd.region] + off);
            J.n = wl - off;
            if (J.n > 0) {
                J.state = ST_NEW;
                // (J.wide on the way in: the score kernels start this read with the 16-bit segmentation; on the way out it is what it was)
                if (cigar_op && overflow_proven(cigar_op, cigar_len, rd.c0, rd.ncig, seq + rd.s0, rd.slen, ref + J.ref_off, J.n)) {
                    J.wide = 1;
                    atomicAdd(&counters[DC_PROVEN], 1);
                }
            }
        }
    }
    jobs[k] = J;
}

// The reads two by two for sw_ends_pair_kernel, like window lengths together (a counting sort over n >> shift, one workgroup:
// a call has a few thousand reads): pairs[0].a = the number of pairs, pairs[1 ..] the pairs.  Reads beyond 24 rows per lane
// stay out (the one-read kernel's LDS form takes them).
__global__ __launch_bounds__(1024) void pair_jobs_kernel(const Job* __restrict__ jobs, int n_reads, int shift, int32_t* __restrict__ order,
                                                         JobPair* __restrict__ pairs, int strip_cap) {
    __shared__ int hist[2048];
    __shared__ int total, longest;
    for (int i = threadIdx.x; i < 2048; i += 1024) hist[i] = 0;
    if (threadIdx.x == 0) longest = 0;
    __syncthreads();
    auto key_of = [&](const Job& J) { const int k = J.n >> shift; return k < 2047 ? k : 2047; };
    int mine = 0;                                 // the longest strip within the cap (what is longer goes to the widest kernel anyway)
    for (int k = threadIdx.x; k < n_reads; k += 1024)
        if (jobs[k].state == ST_NEW && strip_of(jobs[k].m) <= 24) {
            atomicAdd(&hist[key_of(jobs[k])], 1);
            const int st = strip_of(jobs[k].m);
            if (st <= strip_cap && st > mine) mine = st;
        }
    if (mine) atomicMax(&longest, mine);
    __syncthreads();
    if (threadIdx.x == 0) {                       // exclusive scan of 2048 counters: a few microseconds
        int run = 0;
        for (int i = 0; i < 2048; ++i) { const int c = hist[i]; hist[i] = run; run += c; }
        total = run;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < n_reads; k += 1024)
        if (jobs[k].state == ST_NEW && strip_of(jobs[k].m) <= 24) order[atomicAdd(&hist[key_of(jobs[k])], 1)] = k;
    __syncthreads();
    const int np = (total + 1) / 2;
    if (threadIdx.x == 0) {
        pairs[0] = JobPair{np, ladder_size(longest)};          // (longest <= strip_cap <= 22: a size of launch_pair_kernels' ladder)
    }
    for (int i = threadIdx.x; i < np; i += 1024) pairs[i + 1] = JobPair{order[2 * i], 2 * i + 1 < total ? order[2 * i + 1] : -1};
}

// what the host does between the two stages (pa_realigner_align_windows): which reads go on, their first band, their slice of
// the direction workspace (rows of at most 129 slots: half width <= 64)
__global__ __launch_bounds__(256) void band_layout_kernel(Job* __restrict__ jobs, int n_reads, int* __restrict__ counters,
                                                          unsigned long long dir_capacity, int band_waves, int first_width) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= n_reads) return;
    Job& J = jobs[k];
    if (J.state != ST_NEW) return;
    if (J.score <= 1 || J.ref_begin < 0) { J.state = ST_KEPT; return; }      // simple_aligner.cpp:85
    const int n2 = J.ref_end - J.ref_begin + 1, m2 = J.read_end - J.read_begin + 1;
    J.bw = abs(n2 - m2) + 1;
    J.ops_cap = n2 + m2 + 4;
    J.dir_width = min(n2, first_width);
    bool wider = min(2 * J.bw + 1, n2) > J.dir_width;                         // the first band is wider already
    if (!wider) {
        const unsigned long long need = (unsigned long long)m2 * J.dir_width * band_waves + 2ull * (n2 + m2 + 2);   // + steps, classes
        const unsigned long long at = atomicAdd(reinterpret_cast<unsigned long long*>(counters + 6), need);
        if (at + need > dir_capacity) wider = true;
        else J.dir_off = (int64_t)at;
    }
    J.state = wider ? ST_WIDER : ST_BAND;
}

// after the band launch: reads still to be done (their band outgrew the first rows), failed, aligned
__global__ __launch_bounds__(256) void count_states_kernel(const Job* __restrict__ jobs, int n_reads, int* __restrict__ counters) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    const int state = k < n_reads ? jobs[k].state : ST_KEPT;
    const unsigned long long more = __ballot(state == ST_WIDER || state == ST_BAND), err = __ballot(state == ST_ERR),
                             done = __ballot(state == ST_DONE);
    if ((threadIdx.x & 63) == 0) {
        if (more) atomicAdd(&counters[DC_UNFINISHED], __popcll(more));
        if (err) atomicAdd(&counters[DC_ERR], __popcll(err));
        if (done) atomicAdd(&counters[DC_ALIGNED], __popcll(done));
    }
}

// one wave per newly aligned read: its operations decoded behind the batch's CIGAR arrays, the ReadRec pointed there
__global__ __launch_bounds__(256) void apply_alignment_kernel(const Job* __restrict__ jobs, const uint32_t* __restrict__ ops,
                                                              pa_enc::ReadRec* __restrict__ reads, int n_reads,
                                                              int32_t* __restrict__ cigar_op, int32_t* __restrict__ cigar_len,
                                                              long long ops_base, int collapse_eqx) {
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (k >= n_reads) return;
    const Job& J = jobs[k];
    if (J.state != ST_DONE || J.n_ops <= 0) return;                           // (an empty CIGAR leaves the read as it was: simple_aligner.cpp:96)
    const long long at = ops_base + J.ops_off;
    for (int i = lane; i < J.n_ops; i += 64) {
        const uint32_t w = ops[J.ops_off + i];
        int op = (int)(w & 15u);
        if (collapse_eqx && (op == OP_EQ || op == OP_X)) op = 0;
        cigar_op[at + i] = op;
        cigar_len[at + i] = (int32_t)(w >> 4);
    }
    if (lane == 0) {
        pa_enc::ReadRec& rd = reads[k];
        rd.row0 += J.ref_begin;
        rd.c0 = (int32_t)at;
        rd.ncig = J.n_ops;
    }
}

// the band rounds laid out by the host over r->jobs (states ST_BAND / ST_WIDER pending): round 0 with rows of at most 129 slots,
// later rounds with full rows; dj holds the table on the device before and after
int band_rounds_host(pa_realigner* r, Job* dj, const int8_t* dref, const int8_t* dseq, int32_t n_reads, int first_round, int aux, int nw) {
    for (int round = first_round; round < 3; ++round) {
        int64_t dir_total = 0;
        int cap = 0, pending = 0;
        for (Job& J : r->jobs) {
            if (J.state == ST_WIDER) J.state = ST_BAND;
            if (J.state != ST_BAND) continue;
            const int n2 = J.ref_end - J.ref_begin + 1, m2 = J.read_end - J.read_begin + 1;
            J.dir_width = round == 0 ? std::min(n2, first_band_width(nw)) : n2;
            if (round == 0 && std::min(2 * J.bw + 1, n2) > J.dir_width) J.dir_width = n2;   // first band already wider
            J.dir_off = dir_total;
            dir_total += (int64_t)m2 * J.dir_width * nw + 2ll * (n2 + m2 + 2);
            const int bw_cap = J.dir_width >= n2 ? INT32_MAX / 4 : (J.dir_width - 1) / 2;
            cap = std::max(cap, (int)std::min<int64_t>(2 * (int64_t)bw_cap + 3, n2 + 2) + 2);
            ++pending;
        }
        if (!pending) break;
        const size_t lds = (size_t)cap * 12 * nw + (size_t)aux + 64;
        if (lds > 150 * 1024) return pa::set_error(PA_ERR_INVALID, "alignment too long for the band stage");
        RA_ALLOC(r->d_dir, (size_t)std::max<int64_t>(dir_total, 1));
        RA_HIP(hipMemcpyAsync(dj, r->jobs.data(), sizeof(Job) * (size_t)n_reads, hipMemcpyHostToDevice, r->stream));
        RA_HIP(hipEventRecord(r->ev[2], r->stream));
        RA_HIP(launch_band(r->stream, nw, n_reads, lds, dj, dref, dseq, static_cast<uint8_t*>(r->d_dir.p), static_cast<uint32_t*>(r->d_ops.p),
                           static_cast<unsigned long long*>(r->d_counter.p), cap));
        RA_HIP(hipEventRecord(r->ev[3], r->stream));
        RA_HIP(hipMemcpyAsync(r->jobs.data(), dj, sizeof(Job) * (size_t)n_reads, hipMemcpyDeviceToHost, r->stream));
        RA_HIP(hipStreamSynchronize(r->stream));
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, r->ev[2], r->ev[3]) == hipSuccess) r->band_ms += ms;
    }
    return PA_OK;
}

}  // namespace

int pa_ra::align_device(pa_realigner* r, const char* window_text, int64_t window_bytes, const int64_t* window_off,
                        const int32_t* window_len, int32_t n_windows, const pa_enc::ReadRec* d_reads, int32_t n_reads, const char* d_seq,
                        int64_t seq_bytes, int32_t max_region_len, const int32_t* d_cigar_op, const int32_t* d_cigar_len, DeviceResult* out) {
    if (!r || !out || n_reads < 0 || n_windows < 0 || window_bytes < 0 || seq_bytes < 0 ||
        (n_reads > 0 && (!window_text || !window_off || !window_len || !d_reads || !d_seq || n_windows <= 0)))
        return pa::set_error(PA_ERR_INVALID, "null argument");
    if (window_bytes > (int64_t)1 << 30) return pa::set_error(PA_ERR_INVALID, "reference windows too long");
    RA_HIP(hipSetDevice(r->device));
    *out = DeviceResult();
    r->jobs.clear();
    r->ops.clear();
    r->total_ops = 0;
    r->ends_ms = r->band_ms = 0.0;
    r->cells = 0;
    if (n_reads == 0) return PA_OK;
    hipStream_t st = r->stream;
    int max_wl = 0;
    for (int32_t w = 0; w < n_windows; ++w) {
        if (window_len[w] < 0 || window_off[w] < 0 || window_off[w] + window_len[w] > window_bytes)
            return pa::set_error(PA_ERR_INVALID, "window " + std::to_string(w) + " lies outside the window text");
        max_wl = std::max(max_wl, window_len[w]);
    }
    // a clipped read holds the aligned bases of its region and the inserts between them; beyond 2 L + 64 bases (and beyond the
    // kernels' 14-bit cell fields) the call is refused and the caller takes the host-fed form
    const int max_m = std::min(MAX_READ, 2 * std::max(max_region_len, 1) + 64);
    // [window text -> codes][window_off][window_len] in one page-locked block, one upload
    auto up16 = [](size_t v) { return (v + 15) & ~(size_t)15; };
    const size_t o_off = up16((size_t)window_bytes + 64), o_len = up16(o_off + (size_t)n_windows * 8), meta = up16(o_len + (size_t)n_windows * 4);
    if (!r->h_meta.ensure(meta) || !r->h_back.ensure(256)) return pa::set_error(PA_ERR_HIP, "hipHostMalloc failed");
    char* hm = static_cast<char*>(r->h_meta.p);
    std::memcpy(hm, window_text, (size_t)window_bytes);
    std::memcpy(hm + o_off, window_off, (size_t)n_windows * 8);
    std::memcpy(hm + o_len, window_len, (size_t)n_windows * 4);
    RA_ALLOC(r->d_ref, meta);
    RA_ALLOC(r->d_seq, (size_t)seq_bytes + 64);
    RA_ALLOC(r->d_jobs, sizeof(Job) * (size_t)n_reads);
    RA_ALLOC(r->d_counter, 8 + DC_N * 4);
    // operations: a read's worst case is n2 + m2 + 4 <= its window + its bases + 4
    const int64_t ops_cap_total = seq_bytes + (int64_t)n_reads * (max_wl + 4);
    RA_ALLOC(r->d_ops, sizeof(uint32_t) * (size_t)std::max<int64_t>(ops_cap_total, 1));
    const int nw = band_waves_for(n_reads);
    // direction bytes of the first rows (<= 129 slots): what the reads of a region at ordinary length need; what does not fit
    // there is laid out by the host afterwards
    const int first_width = first_band_width(nw);
    const unsigned long long dir_capacity = std::min<unsigned long long>((unsigned long long)n_reads * ((unsigned long long)(max_region_len + 64) *
                                                                             (unsigned long long)first_width * nw + 2ull * (max_wl + max_m + 2)), 24ull << 30);
    RA_ALLOC(r->d_dir, (size_t)std::max<unsigned long long>(dir_capacity, 1));
    RA_HIP(hipMemcpyAsync(r->d_ref.p, hm, meta, hipMemcpyHostToDevice, st));
    RA_HIP(hipMemsetAsync(r->d_counter.p, 0, 8 + DC_N * 4, st));
    int8_t* dref = static_cast<int8_t*>(r->d_ref.p);
    const int64_t* d_woff = reinterpret_cast<const int64_t*>(dref + o_off);
    const int32_t* d_wlen = reinterpret_cast<const int32_t*>(dref + o_len);
    int8_t* dseq = static_cast<int8_t*>(r->d_seq.p);
    Job* dj = static_cast<Job*>(r->d_jobs.p);
    unsigned long long* d_opsctr = static_cast<unsigned long long*>(r->d_counter.p);
    int* d_ctr = reinterpret_cast<int*>(d_opsctr + 1);
    if (window_bytes > 0)
        hipLaunchKernelGGL(to_codes_kernel, dim3((unsigned)((window_bytes + 255) / 256)), dim3(256), 0, st, dref, window_bytes);
    hipLaunchKernelGGL(codes_of_reads_kernel, dim3((unsigned)((n_reads + 3) / 4)), dim3(256), 0, st, d_reads, n_reads, d_seq, dseq);
    {
        // PA_REALIGN_PROOF=0: every read runs its 8-bit pass (A/B runs and the test that holds the two forms to each other)
        const char* v = getenv("PA_REALIGN_PROOF");
        const bool proof = d_cigar_op && d_cigar_len && !(v && v[0] == '0');
        hipLaunchKernelGGL(jobs_of_reads_kernel, dim3((unsigned)((n_reads + 255) / 256)), dim3(256), 0, st, d_reads, n_reads, d_woff, d_wlen, dj,
                           d_ctr, max_m, proof ? d_cigar_op : nullptr, proof ? d_cigar_len : nullptr, dref, dseq);
    }
    RA_HIP(hipEventRecord(r->ev[0], st));
    {
        // every instantiation over the whole table: a read runs in the narrowest register strip that holds it (two wavefronts per
        // SIMD up to 20 rows per lane); LDS for the rows beyond the widest strip
        const int rows = ((max_m + 15) / 16) * 16, R = (rows + 63) / 64;
        const size_t lds = rows > REG_ROWS ? (size_t)64 * R * 4 : 0;
        const char* mode = getenv("PA_REALIGN_SINGLE");                          // (as in pa_realigner_align_windows; read per call: tests flip it)
        const bool single = mode ? mode[0] == '1' : n_reads <= 2048;
        if (single) {
            hipLaunchKernelGGL(sw_ends_kernel<12>, dim3(n_reads), dim3(64), 0, st, dj, dref, dseq, 0, 12);
            hipLaunchKernelGGL(sw_ends_kernel<16>, dim3(n_reads), dim3(64), 0, st, dj, dref, dseq, 12, 16);
            hipLaunchKernelGGL(sw_ends_kernel<20>, dim3(n_reads), dim3(64), 0, st, dj, dref, dseq, 16, 20);
            hipLaunchKernelGGL(sw_ends_kernel<24>, dim3(n_reads), dim3(64), lds, st, dj, dref, dseq, 20, 1 << 30);
        } else {
            // two reads per wavefront, like window lengths together (pair_jobs_kernel); every instantiation over the pair table
            RA_ALLOC(r->d_pairs, sizeof(JobPair) * ((size_t)n_reads / 2 + 2));
            RA_ALLOC(r->d_order, sizeof(int32_t) * (size_t)n_reads);
            int shift = 0;
            while ((max_wl >> shift) >= 2047) ++shift;
            JobPair* dp = static_cast<JobPair*>(r->d_pairs.p);
            // (the reads' lengths are known on the device only: a region's ordinary read keeps its L aligned bases and some
            // inserts; what is longer than L + L / 10 + 16 goes to the catch-all launch, and within that bound pair_jobs_kernel
            // picks the strip size the call's longest read needs -- PA_REALIGN_ADAPT=0: the bound's size, as before round 6)
            const int bound = strip_of(std::min(max_m, max_region_len + max_region_len / 10 + 16));
            const char* av = getenv("PA_REALIGN_ADAPT");
            const bool adaptive = !(av && av[0] == '0');
            hipLaunchKernelGGL(pair_jobs_kernel, dim3(1), dim3(1024), 0, st, dj, n_reads, shift, static_cast<int32_t*>(r->d_order.p), dp,
                               std::min(ladder_size(bound), 22));
            const int* dn = reinterpret_cast<const int*>(dp);
            const unsigned np = (unsigned)(n_reads + 1) / 2;
            launch_pair_kernels(st, (int)np, dj, dp + 1, dn, dref, dseq, bound, adaptive);
            if (rows > REG_ROWS) hipLaunchKernelGGL(sw_ends_kernel<24>, dim3(n_reads), dim3(64), lds, st, dj, dref, dseq, 24, 1 << 30);
        }
        RA_HIP(hipGetLastError());
    }
    RA_HIP(hipEventRecord(r->ev[1], st));
    hipLaunchKernelGGL(band_layout_kernel, dim3((unsigned)((n_reads + 255) / 256)), dim3(256), 0, st, dj, n_reads, d_ctr, dir_capacity, nw, first_width);
    const int aux = max_wl + max_m + 16;
    {
        const int cap = std::min(2 * ((first_width - 1) / 2) + 3, max_wl + 2) + 2;
        const size_t lds = (size_t)cap * 12 * nw + (size_t)aux + 64;
        if (lds > 150 * 1024) return pa::set_error(PA_ERR_INVALID, "alignment too long for the band stage");
        RA_HIP(hipEventRecord(r->ev[2], st));
        RA_HIP(launch_band(st, nw, n_reads, lds, dj, dref, dseq, static_cast<uint8_t*>(r->d_dir.p), static_cast<uint32_t*>(r->d_ops.p), d_opsctr, cap));
        RA_HIP(hipEventRecord(r->ev[3], st));
    }
    hipLaunchKernelGGL(count_states_kernel, dim3((unsigned)((n_reads + 255) / 256)), dim3(256), 0, st, dj, n_reads, d_ctr);
    int* back = static_cast<int*>(r->h_back.p);
    RA_HIP(hipMemcpyAsync(back, r->d_counter.p, 8 + DC_N * 4, hipMemcpyDeviceToHost, st));
    RA_HIP(hipStreamSynchronize(st));
    {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, r->ev[0], r->ev[1]) == hipSuccess) r->ends_ms = ms;
        if (hipEventElapsedTime(&ms, r->ev[2], r->ev[3]) == hipSuccess) r->band_ms = ms;
    }
    const int* ctr = back + 2;
    if (ctr[DC_BAD] > 0) return pa::set_error(PA_ERR_INVALID, "read " + std::to_string(ctr[DC_BAD] - 1) + " starts beyond the reference window");
    if (ctr[DC_TOO_LONG] > 0)
        return pa::set_error(PA_ERR_UNSUPPORTED, "read " + std::to_string(ctr[DC_TOO_LONG] - 1) + " keeps more than " + std::to_string(max_m) +
                                                     " bases of its region (take the host-fed form for this batch)");
    out->n_aligned = ctr[DC_ALIGNED];
    out->n_proven = ctr[DC_PROVEN];
    // reads whose band left the first rows (ST_WIDER: from the layout or from the band kernel's doubling), and -- never, unless a
    // kernel is wrong -- a band that found nothing (ST_ERR): the table comes to the host and the remaining rounds run as in
    // pa_realigner_align_windows, with full rows
    if (ctr[DC_UNFINISHED] > 0 || ctr[DC_ERR] > 0) {
        const bool more = ctr[DC_UNFINISHED] > 0;
        r->jobs.resize((size_t)n_reads);
        RA_HIP(hipMemcpyAsync(r->jobs.data(), dj, sizeof(Job) * (size_t)n_reads, hipMemcpyDeviceToHost, st));
        RA_HIP(hipStreamSynchronize(st));
        if (more) {
            const int rc = band_rounds_host(r, dj, dref, dseq, n_reads, 1, aux, nw);
            if (rc != PA_OK) return rc;
        }
        int aligned = 0;
        for (int32_t k = 0; k < n_reads; ++k) {
            const Job& J = r->jobs[(size_t)k];
            if (J.state == ST_ERR || J.state == ST_BAND || J.state == ST_WIDER)
                return pa::set_error(PA_ERR_INVALID, "read " + std::to_string(k) + ": the band stage did not reach the alignment score "
                                     "(the reference library aborts on such an alignment)");
            aligned += J.state == ST_DONE;
        }
        out->n_aligned = aligned;
        r->jobs.clear();
        RA_HIP(hipMemcpyAsync(back, r->d_counter.p, 8, hipMemcpyDeviceToHost, st));
        RA_HIP(hipStreamSynchronize(st));
    }
    static const bool trace = getenv("PA_REALIGN_TRACE") != nullptr;      // what the band stage needed, on stderr
    if (trace) {
        std::vector<Job> jj((size_t)n_reads);
        RA_HIP(hipMemcpyAsync(jj.data(), dj, sizeof(Job) * (size_t)n_reads, hipMemcpyDeviceToHost, st));
        RA_HIP(hipStreamSynchronize(st));
        long hist_bw[12] = {0}, hist_att[12] = {0}, n_done = 0, n_kept = 0;
        double t_ends = 0, t_dp = 0, t_trace = 0, t_emit = 0;
        for (const Job& J : jj) {
            if (J.state != ST_DONE) { n_kept += 1; continue; }
            ++n_done;
            const int n2 = J.ref_end - J.ref_begin + 1, m2 = J.read_end - J.read_begin + 1, base = std::abs(n2 - m2) + 1;
            int a = 0, b = 0;
            while ((base << a) < J.bw && a < 11) ++a;
            while ((1 << b) < J.bw && b < 11) ++b;
            hist_att[a]++; hist_bw[b]++;
            t_ends += J.t_ends; t_dp += J.t_dp; t_trace += J.t_trace; t_emit += J.t_emit;
        }
        fprintf(stderr, "[realign-device] reads %d aligned %ld other %ld | score %.3f ms band %.3f ms | winning attempt:", n_reads, n_done, n_kept,
                r->ends_ms, r->band_ms);
        for (int k = 0; k < 12; ++k) fprintf(stderr, " %ld", hist_att[k]);
        fprintf(stderr, " | log2(bw):");
        for (int k = 0; k < 12; ++k) fprintf(stderr, " %ld", hist_bw[k]);
        if (n_done) fprintf(stderr, " | per read us: ends %.1f dp %.1f trace %.1f emit %.1f", t_ends / n_done / 100, t_dp / n_done / 100, t_trace / n_done / 100, t_emit / n_done / 100);
        fprintf(stderr, "\n");
    }
    out->jobs = dj;
    out->ops = static_cast<const uint32_t*>(r->d_ops.p);
    out->ops_written = (int64_t)*reinterpret_cast<unsigned long long*>(back);
    if (out->ops_written > ops_cap_total) return pa::set_error(PA_ERR_HIP, "re-aligner: more operations than the output holds");
    return PA_OK;
}

int pa_ra::apply_device(pa_realigner* r, pa_enc::ReadRec* d_reads, int32_t n_reads, int32_t* cigar_op, int32_t* cigar_len,
                        int64_t ops_base, int32_t collapse_eqx) {
    if (!r || n_reads < 0 || (n_reads > 0 && (!d_reads || !cigar_op || !cigar_len || !r->d_jobs.p || !r->d_ops.p)))
        return pa::set_error(PA_ERR_INVALID, "null argument");
    if (n_reads == 0) return PA_OK;
    RA_HIP(hipSetDevice(r->device));
    hipLaunchKernelGGL(apply_alignment_kernel, dim3((unsigned)((n_reads + 3) / 4)), dim3(256), 0, r->stream, static_cast<const Job*>(r->d_jobs.p),
                       static_cast<const uint32_t*>(r->d_ops.p), d_reads, n_reads, cigar_op, cigar_len, (long long)ops_base, (int)collapse_eqx);
    RA_HIP(hipGetLastError());
    return PA_OK;
}
