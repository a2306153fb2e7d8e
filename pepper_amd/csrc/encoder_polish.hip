// Polish summary encoder (include/pepper_amd_encoder.h): pileups of many regions -> uint8 summary rows, one launch set.
// Reference: /root/reference/pepper/modules/src/pileup_summary/summary_generator.cpp:16-32 (feature index), 47-121
// (per-read walk), 274-306 (pixels), 370-393 (row order).
//
// Round 2 walked the CIGAR strings on the host and uploaded segment lists; here the whole walk is on the device, with the
// machinery of the variant encoder (encoder_common.h):
//   polish_segment_kernel<false>  one wave per read: prefix sums over its operations; longest insert per anchor (atomicMax),
//                                 tile record counts
//   polish_rows_kernel            one workgroup per region: exclusive scan of the longest inserts = where every position's
//                                 insert rows sit among the region's insert slots and among its output rows; region totals
//   polish_bases_kernel           prefix sums of the totals over the regions: every region's first insert slot / output row --
//                                 the totals never visit the host in the middle of a call; outputs are sized by a bound that
//                                 grows with what a handle has seen, and a run that outgrows it is repeated with room
//   polish_segment_kernel<true>   the (read, tile) records into their tiles' slices
//   polish_tile_kernel            one workgroup owns 512 positions: 10 feature counts + coverage privatised in LDS, the tile's
//                                 records walked one row per lane; insert bases go to the (sparse) insert-slot counts with global
//                                 atomics; then the base rows' pixels uint8(count / max(1, coverage) * 254) are stored, once
//   polish_insert_rows_kernel     the insert rows' pixels (normalised by the anchor's coverage) and every row's (position, index)
// Reference quirks kept: a deletion credits coverage to its FIRST position once per deleted position inside the region
// (:105-110); REF_SKIP and PAD count as deletions; the walk of a read stops at the first operation that starts past end_pos,
// but an operation that started before it is counted to its end; no quality or mapping-quality-beyond->0 filter.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/pepper_amd_realign.h"
#include "encoder_common.h"

using namespace pa_enc;

namespace {

constexpr int TP = 512, NW = 8, NT = 64 * NW, UNR = 8;   // rows per tile = threads per workgroup; rows per lane in flight
constexpr int NF = 10, K_COV = 10, NCNT = 11;            // features, coverage slot, counters per row
constexpr int PROW = 16;                                 // int32 per insert slot in HBM (10 used)
static_assert(NT == TP && 64 * UNR == TP, "one thread per row");
enum { PC_ERR = 0, PC_INS = 1, PC_OUT = 2, PC_BUG = 3, PC_N = 4 };   // PC_BUG: an index left its buffer (never, unless a kernel is wrong)     // error code | insert rows of the batch | output rows of the batch

struct PRegRec {
    int64_t row_base;            // first row of the region in longest / ins_base / coverage
    int64_t seq_base;
    int64_t pos0;                // region_start
    int32_t L;                   // region_end - region_start + 1
    int32_t tile0, n_tiles;
    int32_t stop_row;            // end_pos - region_start: operations that start past it are not walked
    int32_t out_lo, out_hi;      // start_pos / end_pos as rows (may lie outside [0, L))
    int32_t pad[2];
};
struct TileRec { int32_t read, op, row, ri; };

__host__ __device__ inline int up(char c) { return (c >= 'a' && c <= 'z') ? c - 32 : c; }
__host__ __device__ inline int polish_feature(char b, bool rev) {   // summary_generator.cpp:16-32
    int k;
    switch (up(b)) {
        case 'A': k = 0; break;
        case 'C': k = 1; break;
        case 'G': k = 2; break;
        case 'T': k = 3; break;
        default: return rev ? 8 : 9;
    }
    return rev ? k : k + 4;
}

// exclusive scan of the per-tile record counts (one workgroup)
__global__ __launch_bounds__(1024) void polish_tile_offsets_kernel(const int* __restrict__ tile_count, int n_tiles, int* __restrict__ tile_off) {
    __shared__ int wsum[16];
    __shared__ int carry;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < n_tiles; base += 1024) {
        const int i = base + threadIdx.x;
        const int v = i < n_tiles ? tile_count[i] : 0;
        const int inc = wave_inclusive_sum(v);
        if (lane == 63) wsum[w] = inc;
        __syncthreads();
        int before = carry;
        for (int k = 0; k < w; ++k) before += wsum[k];
        if (i < n_tiles) tile_off[i] = before + inc - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = before + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) tile_off[n_tiles] = carry;
}

// FILL = false: longest insert per anchor + record counts; FILL = true: the records
template <bool FILL>
__global__ __launch_bounds__(256) void polish_segment_kernel(const ReadRec* __restrict__ reads, int n_reads,
                                                             const PRegRec* __restrict__ regions,
                                                             const int32_t* __restrict__ cigar_op, const int32_t* __restrict__ cigar_len,
                                                             int* __restrict__ longest, int* __restrict__ tile_count,
                                                             const int* __restrict__ tile_off, int* __restrict__ tile_fill,
                                                             TileRec* __restrict__ recs, int rec_cap, int* __restrict__ counters) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= n_reads) return;
    const ReadRec rd = reads[r];
    if (!(rd.flags & READ_MAPQ_OK)) return;
    const PRegRec* reg = regions + rd.region;
    const int L = reg->L, tile0 = reg->tile0, stop = reg->stop_row;
    auto put = [&](int tile, int op, int row, int ri) {
        if (!FILL) {
            atomicAdd(&tile_count[tile], 1);
        } else {
            const int slot = tile_off[tile] + atomicAdd(&tile_fill[tile], 1);
            if (slot < rec_cap) recs[slot] = TileRec{r, op, row, ri};
        }
    };
    int pos = rd.row0, ri = 0;
    if (pos > stop) return;
    if (pos >= 0 && pos <= L - 1 && lane == 0) put(tile0 + pos / TP, rd.c0, pos, 0);
    // an insert in front of the read's first aligned base is anchored on the row before it: when that row is the last one of
    // the previous tile, that tile walks the read's leading operations too
    if (pos >= 1 && pos <= L && (pos % TP == 0 || pos == L) && lane == 0) put(tile0 + (pos - 1) / TP, rd.c0, pos, 0);
    for (int cb = 0; cb < rd.ncig; cb += 64) {
        const int i = cb + lane;
        const bool valid = i < rd.ncig;
        const int op = valid ? cigar_op[rd.c0 + i] : OP_H;
        const int len = valid ? cigar_len[rd.c0 + i] : 0;
        const int radv = polish_ref_advance(op, len), qadv = polish_read_advance(op, len);
        const int rinc = wave_inclusive_sum(radv), qinc = wave_inclusive_sum(qadv);
        const int first = pos + rinc - radv, after = pos + rinc, rfirst = ri + qinc - qadv;
        const bool walked = valid && first <= stop;
        if (!FILL && walked && op == OP_I) {
            const int anchor = first - 1;
            if (anchor >= 0 && anchor <= L - 1) {
                if (rfirst + len > rd.slen) atomicMax(&counters[PC_ERR], 2 * (r + 1) + 1);   // insert runs past the sequence
                else atomicMax(&longest[reg->row_base + anchor], len);
            }
        }
        if (walked && radv > 0 && first <= L - 1) {
            int lo = first > rd.row0 + 1 ? first : rd.row0 + 1;
            if (lo < 0) lo = 0;
            const int last_row = after - 1 < L - 1 ? after - 1 : L - 1;
            for (int k = (lo + TP - 1) / TP; k * TP <= last_row; ++k) put(tile0 + k, rd.c0 + i, first, rfirst);
        }
        pos += wave_total(rinc);
        ri += wave_total(qinc);
        if (pos > stop) break;
    }
}

// one workgroup per region: ins_base[row] = insert rows of the output positions before `row` (exclusive scan of the longest
// inserts over the rows that are output positions); totals[r] = the region's insert rows
__global__ __launch_bounds__(1024) void polish_rows_kernel(const PRegRec* __restrict__ regions, const int* __restrict__ longest,
                                                           int* __restrict__ ins_base, int* __restrict__ totals) {
    __shared__ int wsum[16];
    __shared__ int carry;
    const PRegRec reg = regions[blockIdx.x];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int lo = reg.out_lo > 0 ? reg.out_lo : 0, hi = reg.out_hi < reg.L - 1 ? reg.out_hi : reg.L - 1;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < reg.L; base += 1024) {
        const int i = base + threadIdx.x;
        const int v = (i >= lo && i <= hi) ? longest[reg.row_base + i] : 0;
        const int inc = wave_inclusive_sum(v);
        if (lane == 63) wsum[w] = inc;
        __syncthreads();
        int before = carry;
        for (int k = 0; k < w; ++k) before += wsum[k];
        if (i < reg.L) ins_base[reg.row_base + i] = before + inc - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = before + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) totals[blockIdx.x] = carry;
}

// one workgroup: slot_base[r] = insert rows of the regions before r, out_base[r] = out_pos_base[r] + slot_base[r]; the batch's
// totals for the host
__global__ __launch_bounds__(1024) void polish_bases_kernel(const int* __restrict__ totals, const int64_t* __restrict__ out_pos_base,
                                                            int n_regions, int64_t* __restrict__ slot_base, int64_t* __restrict__ out_base,
                                                            int* __restrict__ counters) {
    __shared__ int wsum[16];
    __shared__ int carry;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < n_regions; base += 1024) {
        const int i = base + threadIdx.x;
        const int v = i < n_regions ? totals[i] : 0;
        const int inc = wave_inclusive_sum(v);
        if (lane == 63) wsum[w] = inc;
        __syncthreads();
        int before = carry;
        for (int k = 0; k < w; ++k) before += wsum[k];
        if (i < n_regions) {
            slot_base[i] = before + inc - v;
            out_base[i] = out_pos_base[i] + before + inc - v;
        }
        __syncthreads();
        if (threadIdx.x == 1023) carry = before + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        counters[PC_INS] = carry;
        counters[PC_OUT] = (int)(out_pos_base[n_regions] + carry);
    }
}

struct PTileArgs {
    const ReadRec* reads; const PRegRec* regions; const int32_t* tile_region; const int32_t* cigar_op; const int32_t* cigar_len;
    const char* seq; const TileRec* recs; const int* tile_off; int rec_cap;
    const int* longest; const int* ins_base; const int64_t* slot_base; const int64_t* out_base;
    int* ins_counts; int* coverage; uint8_t* pixels; int* counters;
    int64_t ins_cap, out_cap;        // insert slots / output rows the buffers hold: a batch that needs more is run again with room
};

__global__ __launch_bounds__(NT, 4) void polish_tile_kernel(PTileArgs a) {
    __shared__ int cnt[NCNT * TP];
    __shared__ int s_first[NW][65], s_ri[NW][64], s_op[NW][64];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int tile = blockIdx.x;
    const int region = a.tile_region[tile];
    const PRegRec reg = a.regions[region];
    const int L = reg.L;
    const int tile_lo = (tile - reg.tile0) * TP, tile_hi = tile_lo + TP - 1;
    const int live_max = tile_hi + 1 < reg.stop_row ? tile_hi + 1 : reg.stop_row;
    const int out_lo = reg.out_lo > 0 ? reg.out_lo : 0, out_hi = reg.out_hi < L - 1 ? reg.out_hi : L - 1;
    for (int i = tid; i < NCNT * TP; i += NT) cnt[i] = 0;
    __syncthreads();
    const int rec0 = a.tile_off[tile], rec1 = a.tile_off[tile + 1] < a.rec_cap ? a.tile_off[tile + 1] : a.rec_cap;
    int* first_s = s_first[w];
    int* ri_s = s_ri[w];
    int* op_s = s_op[w];
    const int64_t slot0 = a.slot_base[region];
    TileRec rec{0, 0, 0, 0};
    if (rec0 + w < rec1) rec = a.recs[rec0 + w];
    for (int k = rec0 + w; k < rec1; k += NW) {
        const ReadRec rd_v = a.reads[rec.read];
        int op_ld = a.cigar_op[rec.op + lane], len_ld = a.cigar_len[rec.op + lane];
        TileRec rec_next = rec;
        if (k + NW < rec1) rec_next = a.recs[k + NW];
        const int64_t s0 = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(rd_v.s0 >> 32)) << 32) |
                                     (uint32_t)__builtin_amdgcn_readfirstlane((int)rd_v.s0));
        const int slen = __builtin_amdgcn_readfirstlane(rd_v.slen);
        const int c_end = __builtin_amdgcn_readfirstlane(rd_v.c0) + __builtin_amdgcn_readfirstlane(rd_v.ncig);
        const bool rev = __builtin_amdgcn_readfirstlane(rd_v.flags) & READ_REV;
        const int rec_read = __builtin_amdgcn_readfirstlane(rec.read), rec_op = __builtin_amdgcn_readfirstlane(rec.op);
        const char* seq0 = a.seq + s0;
        int pos = __builtin_amdgcn_readfirstlane(rec.row), ri = __builtin_amdgcn_readfirstlane(rec.ri);
        for (int c = rec_op; c < c_end; c += 64) {
            const int i = c + lane;
            if (c != rec_op) {
                op_ld = a.cigar_op[i];
                len_ld = a.cigar_len[i];
            }
            const bool valid = i < c_end;
            const int op = valid ? op_ld : OP_H;
            const int len = valid ? len_ld : 0;
            const int radv = polish_ref_advance(op, len), qadv = polish_read_advance(op, len);
            const int rinc = wave_inclusive_sum(radv), qinc = wave_inclusive_sum(qadv);
            const int first = pos + rinc - radv, rfirst = ri + qinc - qadv;
            const int total_r = wave_total(rinc);
            const bool live = valid && first <= live_max;
            first_s[lane] = valid ? first : 0x7fffffff;
            if (lane == 63) first_s[64] = valid ? pos + total_r : 0x7fffffff;
            ri_s[lane] = rfirst;
            op_s[lane] = live ? op : OP_H;                 // an operation that starts past end_pos is not walked: its rows count nothing
            __builtin_amdgcn_wave_barrier();
            const int span_lo = pos > tile_lo ? pos : tile_lo;
            int span_hi = pos + total_r - 1;
            if (span_hi > tile_hi) span_hi = tile_hi;
            if (span_hi > L - 1) span_hi = L - 1;

            // -- row phase loads first
            bool is_m[UNR], is_gap[UNR];
            int pl[UNR];
            char bv[UNR];
            bool past = false;
            {
                unsigned si[UNR];
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    const int p = span_lo + lane + 64 * u;
                    const bool act = p <= span_hi;
                    const int pc = act ? p : tile_lo + ((lane + 64 * u) & (TP - 1));
                    const int j = owner_of_row(first_s, pc);
                    const int oj = op_s[j];
                    const int rp = ri_s[j] + (pc - first_s[j]);
                    const bool m = act && (oj == OP_M || oj == OP_EQ || oj == OP_X);
                    const bool inb = (unsigned)rp < (unsigned)slen;
                    past |= m && !inb;
                    is_m[u] = m && inb;
                    is_gap[u] = act && (oj == OP_D || oj == OP_N || oj == OP_P);
                    pl[u] = pc - tile_lo;
                    si[u] = is_m[u] ? (unsigned)rp : 0u;
                }
#pragma unroll
                for (int u = 0; u < UNR; ++u) bv[u] = __builtin_nontemporal_load(seq0 + si[u]);
            }
            if (past) atomicMax(&a.counters[PC_ERR], 2 * (rec_read + 1));   // CIGAR runs past the sequence

            // -- one operation per lane: insert bases into the insert slots (:83-97), the coverage credit of gaps (:105-110)
            const int anchor = first - 1;
            if (live && op == OP_I && anchor >= tile_lo && anchor <= tile_hi && anchor >= out_lo && anchor <= out_hi && rfirst + len <= slen) {
                const int64_t s_at = slot0 + a.ins_base[reg.row_base + anchor];
                int* slot = a.ins_counts + s_at * PROW;
                if (s_at + len <= a.ins_cap)
                    for (int q = 0; q < len; ++q) atomicAdd(&slot[(int64_t)q * PROW + polish_feature(seq0[(unsigned)(rfirst + q)], rev)], 1);
            }
            if (live && radv > 0 && (op == OP_D || op == OP_N || op == OP_P) && first >= tile_lo && first <= tile_hi && first <= L - 1) {
                const int lo_r = first, hi_r = first + len - 1 < L - 1 ? first + len - 1 : L - 1;     // (first >= tile_lo >= 0)
                atomicAdd(&cnt[K_COV * TP + first - tile_lo], hi_r - lo_r + 1);
            }

            // -- the rows: a matched base counts its feature and coverage, a deleted position the strand's gap feature
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const unsigned bu = (unsigned char)bv[u] & 0xDFu;
                const unsigned bidx = (bu >> 1) & 3u;
                const bool letter = ((0x47544341u >> (bidx * 8)) & 0xFFu) == bu;
                const int acgt = (int)(bidx ^ (bidx >> 1));
                const int feat = is_gap[u] ? (rev ? 8 : 9) : (letter ? (rev ? acgt : acgt + 4) : (rev ? 8 : 9));
                atomicAdd(&cnt[feat * TP + pl[u]], (is_m[u] || is_gap[u]) ? 1 : 0);
                atomicAdd(&cnt[K_COV * TP + pl[u]], is_m[u] ? 1 : 0);
            }
            __builtin_amdgcn_wave_barrier();
            pos += total_r;
            ri += wave_total(qinc);
            if (pos > live_max) break;
        }
        rec = rec_next;
    }
    __syncthreads();

    // ---- pixels of the base rows (:274-306), coverage kept for the insert rows
    const int idx = tile_lo + tid;
    if (idx < L) {
        const int cov = cnt[K_COV * TP + tid];
        a.coverage[reg.row_base + idx] = cov;
        const int64_t o_at = a.out_base[region] + (idx - reg.out_lo) + a.ins_base[reg.row_base + idx];
        if (idx >= reg.out_lo && idx <= reg.out_hi && o_at < a.out_cap) {
            const double c = cov > 1 ? (double)cov : 1.0;
            uint8_t* out = a.pixels + o_at * NF;
#pragma unroll
            for (int j = 0; j < NF; ++j) {
                const double v = ((double)cnt[j * TP + tid] / c) * 254.0;
                out[j] = (uint8_t)((long long)v & 0xff);       // double -> uint8 as x86-64 gcc truncates
            }
        }
    }
}

// one thread per output position of every region: its (position, 0) row and its insert rows' positions and pixels
__global__ __launch_bounds__(256) void polish_insert_rows_kernel(const PRegRec* __restrict__ regions, const int64_t* __restrict__ out_pos_base,
                                                                 int n_regions, int64_t total_positions, const int* __restrict__ longest,
                                                                 const int* __restrict__ ins_base, const int64_t* __restrict__ slot_base,
                                                                 const int64_t* __restrict__ out_base, const int* __restrict__ ins_counts,
                                                                 const int* __restrict__ coverage, uint8_t* __restrict__ pixels,
                                                                 int64_t* __restrict__ positions, int64_t out_cap, int64_t ins_cap,
                                                                 int64_t total_rows, int* __restrict__ counters) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= total_positions) return;
    int lo = 0, hi = n_regions - 1;                         // region of output position g: out_pos_base is ascending
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (out_pos_base[mid] <= g) lo = mid; else hi = mid - 1;
    }
    const PRegRec reg = regions[lo];
    const int idx = reg.out_lo + (int)(g - out_pos_base[lo]);
    const bool in = idx >= 0 && idx <= reg.L - 1;
    const int n_ins = in ? longest[reg.row_base + idx] : 0;
    // rows before it: one per output position + the insert rows of the in-region positions before it
    int before_ins = 0;
    if (in) before_ins = ins_base[reg.row_base + idx];
    else if (idx > reg.L - 1 && reg.L > 0) {
        const int last = reg.out_hi < reg.L - 1 ? reg.out_hi : reg.L - 1;
        before_ins = last >= 0 && last >= reg.out_lo ? ins_base[reg.row_base + last] + longest[reg.row_base + last] : 0;
    }
    const int64_t row = out_base[lo] + (idx - reg.out_lo) + before_ins;
    if (row < 0 || n_ins < 0 || before_ins < 0 || (in && reg.row_base + idx >= total_rows)) {
        atomicMax(&counters[PC_BUG], 1 + (row < 0 ? 1 : 0) + (n_ins < 0 ? 2 : 0) + (before_ins < 0 ? 4 : 0));
        return;
    }
    if (row + n_ins >= out_cap || slot_base[lo] + before_ins + n_ins > ins_cap) return;      // (the run is repeated with room)
    positions[2 * row] = reg.pos0 + idx;
    positions[2 * row + 1] = 0;
    if (n_ins > 0) {
        const int cov = coverage[reg.row_base + idx];
        const double c = cov > 1 ? (double)cov : 1.0;
        for (int k = 0; k < n_ins; ++k) {
            const int* src = ins_counts + (slot_base[lo] + before_ins + k) * PROW;
            uint8_t* out = pixels + (row + 1 + k) * NF;
#pragma unroll
            for (int j = 0; j < NF; ++j) {
                const double v = ((double)src[j] / c) * 254.0;
                out[j] = (uint8_t)((long long)v & 0xff);
            }
            positions[2 * (row + 1 + k)] = reg.pos0 + idx;
            positions[2 * (row + 1 + k) + 1] = k + 1;
        }
    }
}

struct PRegHost {
    pa_pileup p;
    int64_t start_pos = 0, end_pos = 0, row_base = 0, seq_base = 0, op_base = 0, read_base = 0;
    int L = 0;
};

}  // namespace

struct pa_polish_batch {
    std::vector<PRegHost> regs;
    std::vector<int64_t> region_rows;        // output rows per region of the last run
    int64_t total_bases = 0, total_ops = 0, total_reads = 0, total_rows = 0, total_out = 0, total_positions = 0;
    int n_tiles = 0, rec_cap = 0;
    int64_t ins_cap = 0;                     // insert slots the output buffers hold (grows with what the handle has seen)
    bool staged = false;
    DBuf d_seq, d_cig_op, d_cig_len, d_meta, d_zero, d_tile_off, d_sorted, d_ins_base, d_totals,
        d_bases, d_ins_counts, d_coverage, d_pixels, d_positions;
    HBuf h_seq, h_cig_op, h_cig_len, h_meta, h_back;
    size_t o_reads = 0, o_regions = 0, o_tile = 0, o_pos = 0;     // the tables inside d_meta
    double ms[4] = {0, 0, 0, 0};
    // device-fed form (the image chain): the reads are where unpack_clip_kernel / the re-aligner left them
    const ReadRec* x_reads = nullptr;
    const char* x_seq = nullptr;
    const int32_t* x_cig_op = nullptr;
    const int32_t* x_cig_len = nullptr;
    // the image chain's chunked output (pa_polish_chain_run): [n_chunks, chunk_size, 10] pixels, [n_chunks, chunk_size] position / index
    DBuf d_chunk_tab, d_chunk_img, d_chunk_pos, d_chunk_idx;
    HBuf h_chunk_tab, h_chunk_img, h_chunk_pos, h_chunk_idx, h_windows;
    std::vector<int32_t> chunk_count;        // per region
    int64_t n_chunks = 0;
    int32_t chunk_size = 0;
    double chain_ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int64_t chain_counts[5] = {0, 0, 0, 0, 0};   // pairs, reads re-aligned, operations written, rows, 8-bit passes proven away
};

void pa_polish_batch_free(pa_polish_batch* b) { delete b; }

namespace {

// validate + upload: the reads of all regions gathered into page-locked blocks (one copy to the device per array instead of
// three pageable ones per region), the tables in one block; nothing here waits for the device
int polish_stage(pa_encoder* e, int32_t n_regions, const pa_pileup* pileups, const int64_t* start_pos, const int64_t* end_pos) {
    if (!e || n_regions < 0 || (n_regions > 0 && (!pileups || !start_pos || !end_pos))) return pa::set_error(PA_ERR_INVALID, "null argument");
    ENC_HIP(hipSetDevice(e->device));
    if (!e->polish) e->polish = new pa_polish_batch();
    pa_polish_batch& b = *e->polish;
    hipStream_t st = e->stream;
    b.staged = false;
    b.x_reads = nullptr;
    b.x_seq = nullptr;
    b.x_cig_op = b.x_cig_len = nullptr;
    b.regs.assign((size_t)n_regions, PRegHost());
    b.region_rows.assign((size_t)n_regions, 0);
    b.total_bases = b.total_ops = b.total_reads = b.total_rows = b.total_out = b.total_positions = 0;
    b.n_tiles = 0;
    if (n_regions == 0) {
        b.staged = true;
        return PA_OK;
    }
    for (int r = 0; r < n_regions; ++r) {
        const pa_pileup& p = pileups[r];
        if (p.region_end < p.region_start || p.region_end - p.region_start > (int64_t)1 << 28 || end_pos[r] < start_pos[r] ||
            end_pos[r] - start_pos[r] > (int64_t)1 << 28 || start_pos[r] < p.region_start - ((int64_t)1 << 28) ||
            end_pos[r] > p.region_end + ((int64_t)1 << 28))
            return pa::set_error(PA_ERR_INVALID, "bad region");
        if (p.n_reads < 0) return pa::set_error(PA_ERR_INVALID, "negative count");
        PRegHost& rh = b.regs[(size_t)r];
        rh.p = p;
        rh.start_pos = start_pos[r];
        rh.end_pos = end_pos[r];
        rh.L = (int)(p.region_end - p.region_start + 1);
        rh.row_base = b.total_rows;
        rh.seq_base = b.total_bases;
        rh.op_base = b.total_ops;
        rh.read_base = b.total_reads;
        b.total_rows += rh.L;
        b.total_reads += p.n_reads;
        b.total_bases += p.n_reads > 0 ? p.seq_offset[p.n_reads] : 0;
        b.total_ops += p.n_reads > 0 ? p.cigar_offset[p.n_reads] : 0;
        b.n_tiles += (rh.L + TP - 1) / TP;
        b.total_positions += end_pos[r] - start_pos[r] + 1;
        if (b.total_rows > ((int64_t)1 << 30) || b.total_ops > 0x7ffffff0 || b.total_reads > 0x3ffffff0 || b.total_positions > ((int64_t)1 << 30))
            return pa::set_error(PA_ERR_INVALID, "batch too large");
    }
    // one block: [ReadRec x reads][PRegRec x regions][tile_region x tiles][out_pos_base x (regions + 1)]
    auto up16 = [](size_t v) { return (v + 15) & ~(size_t)15; };
    b.o_reads = 0;
    b.o_regions = up16((size_t)b.total_reads * sizeof(ReadRec));
    b.o_tile = up16(b.o_regions + (size_t)n_regions * sizeof(PRegRec));
    b.o_pos = up16(b.o_tile + (size_t)b.n_tiles * 4);
    const size_t meta_bytes = up16(b.o_pos + ((size_t)n_regions + 1) * 8);
    if (!b.h_meta.ensure(meta_bytes) || !b.h_seq.ensure((size_t)b.total_bases + 64) || !b.h_cig_op.ensure((size_t)b.total_ops * 4 + 64) ||
        !b.h_cig_len.ensure((size_t)b.total_ops * 4 + 64))
        return pa::set_error(PA_ERR_HIP, "hipHostMalloc failed");
    char* hm = b.h_meta.as<char>();
    ReadRec* reads = reinterpret_cast<ReadRec*>(hm + b.o_reads);
    PRegRec* regrecs = reinterpret_cast<PRegRec*>(hm + b.o_regions);
    int32_t* tile_region = reinterpret_cast<int32_t*>(hm + b.o_tile);
    int64_t* out_pos_base = reinterpret_cast<int64_t*>(hm + b.o_pos);
    int tile0 = 0;
    out_pos_base[0] = 0;
    for (int r = 0; r < n_regions; ++r) {
        const PRegHost& rh = b.regs[(size_t)r];
        const pa_pileup& p = rh.p;
        PRegRec& g = regrecs[r];
        g.row_base = rh.row_base;
        g.seq_base = rh.seq_base;
        g.pos0 = p.region_start;
        g.L = rh.L;
        g.tile0 = tile0;
        g.n_tiles = (rh.L + TP - 1) / TP;
        g.stop_row = (int32_t)(rh.end_pos - p.region_start);
        g.out_lo = (int32_t)(rh.start_pos - p.region_start);
        g.out_hi = (int32_t)(rh.end_pos - p.region_start);
        g.pad[0] = g.pad[1] = 0;
        out_pos_base[r + 1] = out_pos_base[r] + (rh.end_pos - rh.start_pos + 1);
        for (int t = 0; t < g.n_tiles; ++t) tile_region[tile0 + t] = r;
        tile0 += g.n_tiles;
        for (int32_t k = 0; k < p.n_reads; ++k) {
            ReadRec& rd = reads[(size_t)(rh.read_base + k)];
            const int64_t slen = p.seq_offset[k + 1] - p.seq_offset[k], ncig = p.cigar_offset[k + 1] - p.cigar_offset[k];
            const int64_t row0 = p.read_pos[k] - p.region_start;
            if (slen < 0 || ncig < 0 || slen > 0x7ffffff0) return pa::set_error(PA_ERR_INVALID, "offsets of read " + std::to_string(k) + " are not ascending");
            rd.s0 = rh.seq_base + p.seq_offset[k];
            rd.c0 = (int32_t)(rh.op_base + p.cigar_offset[k]);
            rd.ncig = (int32_t)ncig;
            rd.slen = (int32_t)slen;
            rd.row0 = (int32_t)std::max<int64_t>(-(1 << 30), std::min<int64_t>(row0, 1 << 30));
            rd.region = r;
            rd.flags = (p.read_reverse[k] ? READ_REV : 0) | ((p.read_mapq[k] > 0 && row0 > -(1 << 30)) ? READ_MAPQ_OK : 0);
        }
        const int64_t nb = p.n_reads > 0 ? p.seq_offset[p.n_reads] : 0, no = p.n_reads > 0 ? p.cigar_offset[p.n_reads] : 0;
        if (nb > 0) std::memcpy(b.h_seq.as<char>() + rh.seq_base, p.seq, (size_t)nb);
        if (no > 0) {
            std::memcpy(b.h_cig_op.as<int32_t>() + rh.op_base, p.cigar_op, (size_t)no * 4);
            std::memcpy(b.h_cig_len.as<int32_t>() + rh.op_base, p.cigar_len, (size_t)no * 4);
        }
    }
    ENC_ALLOC(b.d_seq, (size_t)b.total_bases + 64);
    ENC_ALLOC(b.d_cig_op, (size_t)b.total_ops * 4 + 1024);
    ENC_ALLOC(b.d_cig_len, (size_t)b.total_ops * 4 + 1024);
    ENC_ALLOC(b.d_meta, meta_bytes);
    if (b.total_bases > 0) ENC_HIP(hipMemcpyAsync(b.d_seq.p, b.h_seq.p, (size_t)b.total_bases, hipMemcpyHostToDevice, st));
    if (b.total_ops > 0) {
        ENC_HIP(hipMemcpyAsync(b.d_cig_op.p, b.h_cig_op.p, (size_t)b.total_ops * 4, hipMemcpyHostToDevice, st));
        ENC_HIP(hipMemcpyAsync(b.d_cig_len.p, b.h_cig_len.p, (size_t)b.total_ops * 4, hipMemcpyHostToDevice, st));
    }
    ENC_HIP(hipMemcpyAsync(b.d_meta.p, hm, meta_bytes, hipMemcpyHostToDevice, st));
    b.staged = true;
    return PA_OK;
}

// the kernels of the staged batch; one wait, at the end (sizes, errors, rows per region)
int polish_run(pa_encoder* e, int64_t* n_rows) {
    if (!e || !e->polish || !e->polish->staged) return pa::set_error(PA_ERR_INVALID, "no staged batch");
    ENC_HIP(hipSetDevice(e->device));
    pa_polish_batch& b = *e->polish;
    hipStream_t st = e->stream;
    const int n_regions = (int)b.regs.size();
    b.total_out = 0;
    if (n_regions == 0) return PA_OK;
    const char* dm = b.d_meta.as<char>();
    const ReadRec* d_reads = b.x_reads ? b.x_reads : reinterpret_cast<const ReadRec*>(dm + b.o_reads);
    const int32_t* d_cig_op = b.x_reads ? b.x_cig_op : b.d_cig_op.as<int32_t>();
    const int32_t* d_cig_len = b.x_reads ? b.x_cig_len : b.d_cig_len.as<int32_t>();
    const char* d_seq = b.x_reads ? b.x_seq : b.d_seq.as<char>();
    const PRegRec* d_regions = reinterpret_cast<const PRegRec*>(dm + b.o_regions);
    const int32_t* d_tile_region = reinterpret_cast<const int32_t*>(dm + b.o_tile);
    const int64_t* d_out_pos_base = reinterpret_cast<const int64_t*>(dm + b.o_pos);
    // zeroed per run: counters | longest [rows] | tile_count | tile_fill
    const size_t n_zero = (size_t)PC_N + (size_t)b.total_rows + 2 * (size_t)b.n_tiles;
    ENC_ALLOC(b.d_zero, n_zero * 4);
    ENC_ALLOC(b.d_tile_off, ((size_t)b.n_tiles + 1) * 4);
    ENC_ALLOC(b.d_ins_base, (size_t)b.total_rows * 4 + 64);
    ENC_ALLOC(b.d_totals, (size_t)n_regions * 4);
    ENC_ALLOC(b.d_coverage, (size_t)b.total_rows * 4 + 64);
    ENC_ALLOC(b.d_bases, (size_t)(2 * (size_t)n_regions + 1) * 8);
    if (!b.h_back.ensure(((size_t)PC_N + 1 + (size_t)n_regions) * 4)) return pa::set_error(PA_ERR_HIP, "hipHostMalloc failed");
    b.rec_cap = std::max<int>(std::max(b.rec_cap, 1024), (int)std::min<int64_t>(0x7ffffff0, b.total_bases / TP + 2 * b.total_reads + 1024));
    // insert rows: one per position and base of the longest insert anchored there -- known on the device only.  Room for two
    // per position to begin with (nanopore pileups at 60x: ~1.3), and for what the handle has needed before
    b.ins_cap = std::max<int64_t>(b.ins_cap, 2 * b.total_rows + 4096);
    int* counters = b.d_zero.as<int>();
    int* longest = counters + PC_N;
    int* tile_count = longest + b.total_rows;
    int* tile_fill = tile_count + b.n_tiles;
    int* back = b.h_back.as<int>();            // [PC_N] counters | records | totals per region
    for (int attempt = 0;; ++attempt) {
        const int64_t out_cap = b.total_positions + b.ins_cap;
        ENC_ALLOC(b.d_sorted, (size_t)b.rec_cap * sizeof(TileRec));
        ENC_ALLOC(b.d_ins_counts, (size_t)(b.ins_cap + 1) * PROW * 4);
        ENC_ALLOC(b.d_pixels, (size_t)out_cap * NF + 64);
        ENC_ALLOC(b.d_positions, (size_t)out_cap * 16 + 64);
        ENC_HIP(hipMemsetAsync(b.d_zero.p, 0, n_zero * 4, st));
        ENC_HIP(hipMemsetAsync(b.d_ins_counts.p, 0, (size_t)(b.ins_cap + 1) * PROW * 4, st));
        ENC_HIP(hipMemsetAsync(b.d_pixels.p, 0, (size_t)out_cap * NF + 64, st));     // (rows of positions outside the region stay zero)
        ENC_HIP(hipEventRecord(e->ev[0], st));
        const dim3 seg_grid((unsigned)((b.total_reads + 3) / 4));
        if (b.total_reads > 0)
            hipLaunchKernelGGL(polish_segment_kernel<false>, seg_grid, dim3(256), 0, st, d_reads, (int)b.total_reads, d_regions,
                               d_cig_op, d_cig_len, longest, tile_count, (const int*)nullptr, (int*)nullptr,
                               (TileRec*)nullptr, 0, counters);
        hipLaunchKernelGGL(polish_rows_kernel, dim3((unsigned)n_regions), dim3(1024), 0, st, d_regions, longest, b.d_ins_base.as<int>(),
                           b.d_totals.as<int>());
        int64_t* slot_base = b.d_bases.as<int64_t>();
        int64_t* out_base = slot_base + n_regions;
        hipLaunchKernelGGL(polish_bases_kernel, dim3(1), dim3(1024), 0, st, b.d_totals.as<int>(), d_out_pos_base, n_regions, slot_base, out_base,
                           counters);
        if (b.n_tiles > 0)
            hipLaunchKernelGGL(polish_tile_offsets_kernel, dim3(1), dim3(1024), 0, st, tile_count, b.n_tiles, b.d_tile_off.as<int>());
        if (b.total_reads > 0)
            hipLaunchKernelGGL(polish_segment_kernel<true>, seg_grid, dim3(256), 0, st, d_reads, (int)b.total_reads, d_regions,
                               d_cig_op, d_cig_len, longest, tile_count, b.d_tile_off.as<int>(), tile_fill,
                               b.d_sorted.as<TileRec>(), b.rec_cap, counters);
        ENC_HIP(hipEventRecord(e->ev[1], st));
        if (b.n_tiles > 0) {
            PTileArgs ta;
            ta.reads = d_reads;
            ta.regions = d_regions;
            ta.tile_region = d_tile_region;
            ta.cigar_op = d_cig_op;
            ta.cigar_len = d_cig_len;
            ta.seq = d_seq;
            ta.recs = b.d_sorted.as<TileRec>();
            ta.tile_off = b.d_tile_off.as<int>();
            ta.rec_cap = b.rec_cap;
            ta.longest = longest;
            ta.ins_base = b.d_ins_base.as<int>();
            ta.slot_base = slot_base;
            ta.out_base = out_base;
            ta.ins_counts = b.d_ins_counts.as<int>();
            ta.coverage = b.d_coverage.as<int>();
            ta.pixels = b.d_pixels.as<uint8_t>();
            ta.counters = counters;
            ta.ins_cap = b.ins_cap;
            ta.out_cap = out_cap;
            hipLaunchKernelGGL(polish_tile_kernel, dim3((unsigned)b.n_tiles), dim3(NT), 0, st, ta);
        }
        ENC_HIP(hipEventRecord(e->ev[2], st));
        hipLaunchKernelGGL(polish_insert_rows_kernel, dim3((unsigned)((b.total_positions + 255) / 256)), dim3(256), 0, st, d_regions,
                           d_out_pos_base, n_regions, b.total_positions, longest, b.d_ins_base.as<int>(), slot_base, out_base,
                           b.d_ins_counts.as<int>(), b.d_coverage.as<int>(), b.d_pixels.as<uint8_t>(), b.d_positions.as<int64_t>(), out_cap,
                           b.ins_cap, b.total_rows, counters);
        ENC_HIP(hipEventRecord(e->ev[3], st));
        ENC_HIP(hipGetLastError());
        ENC_HIP(hipMemcpyAsync(back, counters, PC_N * 4, hipMemcpyDeviceToHost, st));
        back[PC_N] = 0;
        if (b.n_tiles > 0) ENC_HIP(hipMemcpyAsync(back + PC_N, b.d_tile_off.as<int>() + b.n_tiles, 4, hipMemcpyDeviceToHost, st));
        ENC_HIP(hipMemcpyAsync(back + PC_N + 1, b.d_totals.p, (size_t)n_regions * 4, hipMemcpyDeviceToHost, st));
        ENC_HIP(hipStreamSynchronize(st));
        const int n_recs = back[PC_N];
        if (n_recs > b.rec_cap || back[PC_INS] > b.ins_cap) {
            if (attempt >= 2) return pa::set_error(PA_ERR_HIP, "encoder buffers could not be sized");
            b.rec_cap = std::max(b.rec_cap, n_recs + 1024);
            b.ins_cap = std::max<int64_t>(b.ins_cap, (int64_t)back[PC_INS] + back[PC_INS] / 4 + 1024);
            continue;
        }
        break;
    }
    if (back[PC_BUG] != 0) return pa::set_error(PA_ERR_HIP, "polish encoder: an index left its buffer (code " + std::to_string(back[PC_BUG]) + ")");
    if (back[PC_ERR] > 0) {
        const int code = back[PC_ERR];
        const int64_t g = code / 2 - 1;
        size_t r = 0;
        while (r + 1 < b.regs.size() && b.regs[r + 1].read_base <= g) ++r;
        return pa::set_error(PA_ERR_INVALID, std::string(code & 1 ? "insert" : "CIGAR") + " of read " + std::to_string(g - b.regs[r].read_base) +
                                                 (n_regions > 1 ? " of region " + std::to_string(r) : "") + " runs past its sequence");
    }
    for (int r = 0; r < n_regions; ++r) {
        b.region_rows[(size_t)r] = (b.regs[(size_t)r].end_pos - b.regs[(size_t)r].start_pos + 1) + back[PC_N + 1 + r];
        b.total_out += b.region_rows[(size_t)r];
        if (n_rows) n_rows[r] = b.region_rows[(size_t)r];
    }
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e->ev[0], e->ev[1]); b.ms[0] = ms;     // segment passes + scans
    (void)hipEventElapsedTime(&ms, e->ev[1], e->ev[2]); b.ms[1] = ms;     // polish_tile_kernel
    (void)hipEventElapsedTime(&ms, e->ev[2], e->ev[3]); b.ms[2] = ms;     // polish_insert_rows_kernel
    return PA_OK;
}

// The tables of a batch whose reads are already on the device (the image chain): regions from their bounds, the ReadRec table,
// bases and CIGAR arrays where unpack_clip_kernel and the re-aligner left them.  Nothing here waits for the device.
int polish_stage_device(pa_encoder* e, int32_t n_regions, const int64_t* region_start, const int64_t* region_end,
                        const int32_t* region_pairs, const UnpackedReads& u) {
    ENC_HIP(hipSetDevice(e->device));
    if (!e->polish) e->polish = new pa_polish_batch();
    pa_polish_batch& b = *e->polish;
    hipStream_t st = e->stream;
    b.staged = false;
    b.regs.assign((size_t)n_regions, PRegHost());
    b.region_rows.assign((size_t)n_regions, 0);
    b.total_bases = u.total_bases;
    b.total_ops = u.total_ops + u.extra_ops;
    b.total_reads = u.n_pairs;
    b.total_rows = b.total_out = b.total_positions = 0;
    b.n_tiles = 0;
    b.x_reads = u.reads;
    b.x_seq = u.seq;
    b.x_cig_op = u.cigar_op;
    b.x_cig_len = u.cigar_len;
    if (n_regions == 0) {
        b.staged = true;
        return PA_OK;
    }
    for (int r = 0; r < n_regions; ++r) {
        if (region_end[r] < region_start[r] || region_end[r] - region_start[r] > (int64_t)1 << 28) return pa::set_error(PA_ERR_INVALID, "bad region");
        PRegHost& rh = b.regs[(size_t)r];
        rh.p = pa_pileup{};
        rh.p.region_start = region_start[r];
        rh.p.region_end = region_end[r];
        rh.p.n_reads = region_pairs[r + 1] - region_pairs[r];
        rh.start_pos = region_start[r];
        rh.end_pos = region_end[r];
        rh.L = (int)(region_end[r] - region_start[r] + 1);
        rh.row_base = b.total_rows;
        rh.read_base = region_pairs[r];
        b.total_rows += rh.L;
        b.n_tiles += (rh.L + TP - 1) / TP;
        b.total_positions += rh.L;
        if (b.total_rows > ((int64_t)1 << 30)) return pa::set_error(PA_ERR_INVALID, "batch too large");
    }
    auto up16 = [](size_t v) { return (v + 15) & ~(size_t)15; };
    b.o_reads = 0;
    b.o_regions = 0;
    b.o_tile = up16(b.o_regions + (size_t)n_regions * sizeof(PRegRec));
    b.o_pos = up16(b.o_tile + (size_t)b.n_tiles * 4);
    const size_t meta_bytes = up16(b.o_pos + ((size_t)n_regions + 1) * 8);
    if (!b.h_meta.ensure(meta_bytes)) return pa::set_error(PA_ERR_HIP, "hipHostMalloc failed");
    char* hm = b.h_meta.as<char>();
    PRegRec* regrecs = reinterpret_cast<PRegRec*>(hm + b.o_regions);
    int32_t* tile_region = reinterpret_cast<int32_t*>(hm + b.o_tile);
    int64_t* out_pos_base = reinterpret_cast<int64_t*>(hm + b.o_pos);
    int tile0 = 0;
    out_pos_base[0] = 0;
    for (int r = 0; r < n_regions; ++r) {
        const PRegHost& rh = b.regs[(size_t)r];
        PRegRec& g = regrecs[r];
        g.row_base = rh.row_base;
        g.seq_base = 0;
        g.pos0 = rh.p.region_start;
        g.L = rh.L;
        g.tile0 = tile0;
        g.n_tiles = (rh.L + TP - 1) / TP;
        g.stop_row = rh.L - 1;
        g.out_lo = 0;
        g.out_hi = rh.L - 1;
        g.pad[0] = g.pad[1] = 0;
        out_pos_base[r + 1] = out_pos_base[r] + rh.L;
        for (int t = 0; t < g.n_tiles; ++t) tile_region[tile0 + t] = r;
        tile0 += g.n_tiles;
    }
    ENC_ALLOC(b.d_meta, meta_bytes);
    ENC_HIP(hipMemcpyAsync(b.d_meta.p, hm, meta_bytes, hipMemcpyHostToDevice, st));
    b.staged = true;
    return PA_OK;
}

// chunk_images (pepper AlignmentSummarizer.py:18-56) for every region of the batch: one thread per row of a chunk.  tab[c] =
// {first source row (among all output rows of the batch), rows taken from there}; the rows behind them are zero pixels with
// position / index -1.
struct ChunkRec { int64_t src; int32_t n, pad; };
__global__ __launch_bounds__(256) void polish_chunk_rows_kernel(const ChunkRec* __restrict__ tab, int chunk_size, const uint8_t* __restrict__ pixels,
                                                                const int64_t* __restrict__ positions, uint8_t* __restrict__ img,
                                                                int64_t* __restrict__ pos, int64_t* __restrict__ idx) {
    const int c = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= chunk_size) return;
    const ChunkRec t = tab[c];
    const int64_t o = (int64_t)c * chunk_size + i;
    uint8_t* dst = img + o * NF;
    if (i < t.n) {
        const uint8_t* src = pixels + (t.src + i) * NF;
#pragma unroll
        for (int j = 0; j < NF; ++j) dst[j] = src[j];
        pos[o] = positions[2 * (t.src + i)];
        idx[o] = positions[2 * (t.src + i) + 1];
    } else {
#pragma unroll
        for (int j = 0; j < NF; ++j) dst[j] = 0;
        pos[o] = -1;
        idx[o] = -1;
    }
}

}  // namespace

extern "C" {

int pa_polish_encoder_generate_summary_batch(pa_encoder* e, int32_t n_regions, const pa_pileup* pileups, const int64_t* start_pos,
                                             const int64_t* end_pos, int64_t* n_rows) {
    const int rc = polish_stage(e, n_regions, pileups, start_pos, end_pos);
    return rc != PA_OK ? rc : polish_run(e, n_rows);
}

int pa_polish_encoder_stage_batch(pa_encoder* e, int32_t n_regions, const pa_pileup* pileups, const int64_t* start_pos,
                                  const int64_t* end_pos) {
    return polish_stage(e, n_regions, pileups, start_pos, end_pos);
}

int pa_polish_encoder_run_staged(pa_encoder* e, int64_t* n_rows) { return polish_run(e, n_rows); }

int pa_polish_encoder_batch_stats(pa_encoder* e, int64_t* out, int32_t n) {
    if (!e || !out || n < 0) return pa::set_error(PA_ERR_INVALID, "null argument");
    const pa_polish_batch* b = e->polish;
    const int64_t v[6] = {b ? b->total_bases : 0, b ? b->total_out : 0, b ? b->total_reads : 0, b ? b->total_ops : 0,
                          b ? (int64_t)b->n_tiles : 0, b ? (int64_t)b->regs.size() : 0};
    for (int i = 0; i < n; ++i) out[i] = i < 6 ? v[i] : 0;
    return PA_OK;
}

int pa_polish_encoder_generate_summary(pa_encoder* e, const pa_pileup* p, int64_t start_pos, int64_t end_pos, int64_t* n_rows) {
    if (!e || !p || !n_rows) return pa::set_error(PA_ERR_INVALID, "null argument");
    const int rc = polish_stage(e, 1, p, &start_pos, &end_pos);
    return rc != PA_OK ? rc : polish_run(e, n_rows);
}

int pa_polish_encoder_get_results(pa_encoder* e, uint8_t* image, int64_t* positions) {
    if (!e || !e->polish) return pa::set_error(PA_ERR_INVALID, "null encoder or no polish summary");
    ENC_HIP(hipSetDevice(e->device));
    pa_polish_batch& b = *e->polish;
    if (b.total_out > 0) {
        if (image) ENC_HIP(hipMemcpyAsync(image, b.d_pixels.p, (size_t)b.total_out * NF, hipMemcpyDeviceToHost, e->stream));
        if (positions) ENC_HIP(hipMemcpyAsync(positions, b.d_positions.p, (size_t)b.total_out * 16, hipMemcpyDeviceToHost, e->stream));
        ENC_HIP(hipStreamSynchronize(e->stream));
    }
    return PA_OK;
}

int pa_polish_encoder_last_timing(pa_encoder* e, double* ms, int32_t n) {
    if (!e || !ms || n < 0) return pa::set_error(PA_ERR_INVALID, "null argument");
    for (int i = 0; i < n; ++i) ms[i] = (e->polish && i < 4) ? e->polish->ms[i] : 0.0;
    return PA_OK;
}

// ---- the polish image chain ------------------------------------------------------------------------------------------------
int pa_polish_chain_run(pa_encoder* e, int32_t n_regions, const pa_packed_region* regions, const uint8_t* arena, int64_t arena_bytes,
                        const pa_packed_read* reads, int32_t n_reads, const int32_t* pair_read, const int32_t* region_pairs,
                        int32_t realign, int32_t chunk_size, int32_t chunk_overlap, int64_t* n_rows, int32_t* region_reads,
                        int32_t* n_chunks, int64_t* total_chunks) {
    if (!e || n_regions < 0 || (n_regions > 0 && (!regions || !region_pairs)) || chunk_size <= 0 || chunk_overlap < 0 || chunk_overlap >= chunk_size)
        return pa::set_error(PA_ERR_INVALID, "null argument");
    ENC_HIP(hipSetDevice(e->device));
    if (!e->polish) e->polish = new pa_polish_batch();
    pa_polish_batch& b = *e->polish;
    hipStream_t st = e->stream;
    b.n_chunks = 0;
    b.chunk_size = chunk_size;
    b.chunk_count.assign((size_t)n_regions, 0);
    for (double& v : b.chain_ms) v = 0;
    for (int64_t& v : b.chain_counts) v = 0;
    if (total_chunks) *total_chunks = 0;
    if (n_regions == 0) return PA_OK;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto since = [](std::chrono::steady_clock::time_point t0) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
    auto t0 = now();
    std::vector<int64_t> rs((size_t)n_regions), re((size_t)n_regions);
    int max_L = 1, max_wl = 0;
    int64_t window_bytes = 0;
    for (int r = 0; r < n_regions; ++r) {
        rs[(size_t)r] = regions[r].region_start;
        re[(size_t)r] = regions[r].region_end;
        if (re[(size_t)r] < rs[(size_t)r] || re[(size_t)r] - rs[(size_t)r] > (int64_t)1 << 24) return pa::set_error(PA_ERR_INVALID, "bad region");
        max_L = std::max(max_L, (int)(re[(size_t)r] - rs[(size_t)r] + 1));
        if (realign) {
            if (regions[r].reference_len < 0 || regions[r].reference_len > (int64_t)1 << 26 || (regions[r].reference_len > 0 && !regions[r].reference))
                return pa::set_error(PA_ERR_INVALID, "bad reference window");
            max_wl = std::max(max_wl, (int)regions[r].reference_len);
            window_bytes += regions[r].reference_len;
        }
    }
    // 1. clip + decode the pairs on the device
    UnpackedReads u;
    int rc = unpack_packed_regions(e, n_regions, rs.data(), re.data(), arena, arena_bytes, reads, n_reads, pair_read, region_pairs,
                                   realign ? max_wl + 4 : -1, &u);
    if (rc != PA_OK) return rc;
    b.chain_counts[0] = u.n_pairs;
    b.chain_ms[0] = since(t0);
    t0 = now();
    // 2. every read re-aligned to its region's window of the draft (simple_aligner.cpp:66-106); the new CIGARs behind the old ones
    if (realign && u.n_pairs > 0) {
        if (!e->realigner) {
            rc = pa_realigner_create(e->device, st, &e->realigner);
            if (rc != PA_OK) return rc;
        }
        if (!b.h_windows.ensure((size_t)window_bytes + (size_t)n_regions * 12 + 64)) return pa::set_error(PA_ERR_HIP, "hipHostMalloc failed");
        char* text = b.h_windows.as<char>();
        std::vector<int64_t> woff((size_t)n_regions);
        std::vector<int32_t> wlen((size_t)n_regions);
        int64_t at = 0;
        for (int r = 0; r < n_regions; ++r) {
            woff[(size_t)r] = at;
            wlen[(size_t)r] = (int32_t)regions[r].reference_len;
            if (regions[r].reference_len > 0) std::memcpy(text + at, regions[r].reference, (size_t)regions[r].reference_len);
            at += regions[r].reference_len;
        }
        pa_ra::DeviceResult res;
        rc = pa_ra::align_device(e->realigner, text, window_bytes, woff.data(), wlen.data(), n_regions, u.reads, (int32_t)u.n_pairs, u.seq,
                                 u.total_bases, max_L, u.cigar_op, u.cigar_len, &res);
        if (rc != PA_OK) return rc;
        b.chain_counts[1] = res.n_aligned;
        b.chain_counts[2] = res.ops_written;
        b.chain_counts[4] = res.n_proven;
        if (res.ops_written > u.extra_ops) return pa::set_error(PA_ERR_HIP, "polish chain: more re-aligned operations than the CIGAR arrays hold");
        (void)pa_realigner_last_timing(e->realigner, &b.chain_ms[5], &b.chain_ms[6], nullptr);
        rc = pa_ra::apply_device(e->realigner, u.reads, (int32_t)u.n_pairs, u.cigar_op, u.cigar_len, u.total_ops, 1);
        if (rc != PA_OK) return rc;
    }
    b.chain_ms[1] = since(t0);
    t0 = now();
    // 3. the summary of every region from those reads
    rc = polish_stage_device(e, n_regions, rs.data(), re.data(), region_pairs, u);
    if (rc != PA_OK) return rc;
    std::vector<int64_t> rows((size_t)n_regions, 0);
    rc = polish_run(e, rows.data());
    if (rc != PA_OK) return rc;
    // (the stream has been waited for: what unpack_clip_kernel reported is here)
    if (u.h_live[n_regions] > 0)
        return pa::set_error(PA_ERR_INVALID, "packed read " + std::to_string(u.h_live[n_regions] - 1) + ": its CIGAR walks over more bases than the record holds");
    if (u.h_live[n_regions + 1] > 0)
        return pa::set_error(PA_ERR_UNSUPPORTED, "packed read " + std::to_string(u.h_live[n_regions + 1] - 1) +
                                                     ": an operation of 2^24 bases or more, or more kept bases than a region's pair holds (take the host-clipped form)");
    b.chain_ms[2] = since(t0);
    t0 = now();
    // 4. chunks: a region without reads has none (AlignmentSummarizer.py:311-312)
    int64_t total = 0, first_row = 0;
    for (int r = 0; r < n_regions; ++r) {
        const int64_t nr = rows[(size_t)r];
        int32_t c = 0;
        if (u.h_live[r] > 0 && nr > 0) c = nr <= chunk_size ? 1 : 1 + (int32_t)((nr - chunk_size + (chunk_size - chunk_overlap) - 1) / (chunk_size - chunk_overlap));
        b.chunk_count[(size_t)r] = c;
        total += c;
        if (n_rows) n_rows[r] = nr;
        if (region_reads) region_reads[r] = u.h_live[r];
        if (n_chunks) n_chunks[r] = c;
        b.chain_counts[3] += nr;
    }
    b.n_chunks = total;
    if (total_chunks) *total_chunks = total;
    if (total > 0) {
        if (!b.h_chunk_tab.ensure((size_t)total * sizeof(ChunkRec)) || !b.h_chunk_img.ensure((size_t)total * chunk_size * NF) ||
            !b.h_chunk_pos.ensure((size_t)total * chunk_size * 8) || !b.h_chunk_idx.ensure((size_t)total * chunk_size * 8))
            return pa::set_error(PA_ERR_HIP, "hipHostMalloc failed");
        ChunkRec* tab = b.h_chunk_tab.as<ChunkRec>();
        int64_t k = 0;
        for (int r = 0; r < n_regions; ++r) {
            const int64_t nr = rows[(size_t)r];
            int64_t start = 0, end = std::min<int64_t>(nr, chunk_size);
            for (int32_t c = 0; c < b.chunk_count[(size_t)r]; ++c) {
                tab[k++] = ChunkRec{first_row + start, (int32_t)(end - start), 0};
                start = end - chunk_overlap;
                end = std::min<int64_t>(nr, start + chunk_size);
            }
            first_row += nr;
        }
        ENC_ALLOC(b.d_chunk_tab, (size_t)total * sizeof(ChunkRec));
        ENC_ALLOC(b.d_chunk_img, (size_t)total * chunk_size * NF);
        ENC_ALLOC(b.d_chunk_pos, (size_t)total * chunk_size * 8);
        ENC_ALLOC(b.d_chunk_idx, (size_t)total * chunk_size * 8);
        ENC_HIP(hipMemcpyAsync(b.d_chunk_tab.p, tab, (size_t)total * sizeof(ChunkRec), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(polish_chunk_rows_kernel, dim3((unsigned)((chunk_size + 255) / 256), (unsigned)total), dim3(256), 0, st,
                           b.d_chunk_tab.as<ChunkRec>(), chunk_size, b.d_pixels.as<uint8_t>(), b.d_positions.as<int64_t>(),
                           b.d_chunk_img.as<uint8_t>(), b.d_chunk_pos.as<int64_t>(), b.d_chunk_idx.as<int64_t>());
        ENC_HIP(hipGetLastError());
        ENC_HIP(hipMemcpyAsync(b.h_chunk_img.p, b.d_chunk_img.p, (size_t)total * chunk_size * NF, hipMemcpyDeviceToHost, st));
        ENC_HIP(hipMemcpyAsync(b.h_chunk_pos.p, b.d_chunk_pos.p, (size_t)total * chunk_size * 8, hipMemcpyDeviceToHost, st));
        ENC_HIP(hipMemcpyAsync(b.h_chunk_idx.p, b.d_chunk_idx.p, (size_t)total * chunk_size * 8, hipMemcpyDeviceToHost, st));
        ENC_HIP(hipStreamSynchronize(st));
    } else {
        for (int r = 0; r < n_regions; ++r) first_row += rows[(size_t)r];
    }
    b.chain_ms[3] = since(t0);
    return PA_OK;
}

int pa_polish_chain_chunks(pa_encoder* e, const uint8_t** images, const int64_t** position, const int64_t** index) {
    if (!e || !e->polish) return pa::set_error(PA_ERR_INVALID, "no chain run");
    const pa_polish_batch& b = *e->polish;
    if (images) *images = b.n_chunks > 0 ? b.h_chunk_img.as<uint8_t>() : nullptr;
    if (position) *position = b.n_chunks > 0 ? b.h_chunk_pos.as<int64_t>() : nullptr;
    if (index) *index = b.n_chunks > 0 ? b.h_chunk_idx.as<int64_t>() : nullptr;
    return PA_OK;
}

int pa_polish_chain_device_chunks(pa_encoder* e, const uint8_t** images) {
    if (!e || !e->polish || !images) return pa::set_error(PA_ERR_INVALID, "no chain run");
    *images = e->polish->n_chunks > 0 ? e->polish->d_chunk_img.as<uint8_t>() : nullptr;
    return PA_OK;
}

int pa_polish_chain_last_timing(pa_encoder* e, double* ms, int32_t n_ms, int64_t* counts, int32_t n_counts) {
    if (!e || n_ms < 0 || n_counts < 0) return pa::set_error(PA_ERR_INVALID, "null argument");
    for (int i = 0; ms && i < n_ms; ++i) ms[i] = (e->polish && i < 8) ? e->polish->chain_ms[i] : 0.0;
    for (int i = 0; counts && i < n_counts; ++i) counts[i] = (e->polish && i < 5) ? e->polish->chain_counts[i] : 0;
    return PA_OK;
}

}  // extern "C"
