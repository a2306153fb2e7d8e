// Variant summary encoder (include/pepper_amd_encoder.h): pileup -> candidate images.
//
// Split of the reference's RegionalSummaryGenerator::generate_summary
// (/root/reference/pepper_variant/modules/cpp/region_summary.cpp:337-916):
//   GPU   : cigar_walk_kernel      the whole per-read walk, one wave per read: coverage, strand coverage, base
//                                  columns, SNP counts and SNP allele tallies (:366-428), insert quality sums and the
//                                  sparse indel updates (:431-540), '*' columns of deleted bases (:541-551), one
//                                  (row, type, strand, length, byte source) record per indel allele vote
//                                                                                       [atomics, HBM bound]
//           (PA_ENCODER_HOST_CIGAR=1: round 1's split -- host pass over CIGAR ops -> segment / event lists ->
//            pileup_count_kernel + apply_events_kernel)
//   host  : what needs strings: ordered per-site maps of insert / delete allele keys, built only for the sites
//           that pass the thresholds, from the votes compact_votes_kernel leaves for them;
//           site_threshold_kernel  per-position fractions vs thresholds in fp64, clamp of columns
//                                  11..24 (:634-654), compaction of passing sites
//           gather_windows_kernel  33 x 26 window copy + candidate-specific overwrite (:828-905),
//                                  int32 image_matrix and the int8 wrap DataStore.py:68 applies
//   host  : candidate enumeration in the reference's std::set order with its filters (:669-712).
// The matrix lives on the device as int32 [L+1][32]: columns 0..25 = the image, 26 coverage,
// 27 snp_count, 28 insert_count, 29 delete_count (128-byte rows).
#include "../../include/pepper_amd_encoder.h"
#include "../../include/pepper_amd.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "kernels.h"

namespace {

enum { OP_M = 0, OP_I = 1, OP_D = 2, OP_N = 3, OP_S = 4, OP_H = 5, OP_P = 6, OP_EQ = 7, OP_X = 8 };
constexpr int MAXC = 125, ROW = 32, C_COV = 26, C_SNP = 27, C_INS = 28, C_DEL = 29;
constexpr uint32_t SEG_REV = 1, SEG_DEL = 2, SEG_ANCHOR = 4;

struct Seg {          // a run of consecutive reference positions touched by one CIGAR op
    int64_t seq0;     // offset of the first base in the concatenated seq / qual arrays
    int32_t idx0;     // first row (pos - region_start)
    int32_t n;        // rows
    uint32_t flags;   // SEG_REV | SEG_DEL | SEG_ANCHOR (last base anchors an indel: no strand coverage)
    int32_t pad;
};
struct Event { int32_t row, col, delta, pad; };
struct SiteRec { int32_t idx, cov, flags, fwd[4], rev[4]; };
struct CandDesc {
    int32_t idx, type;        // row of the candidate site; 1 SNP, 2 insert, 3 delete
    int32_t vcol, vval;       // columns 1/2/3 <- alt base code / allele length
    int32_t fwd, rev;         // strand allele depths (<= 125) for columns 5..7 / 16..18
    int32_t neg_f, neg_r;     // columns negated on the centre row (-1: none)
    int32_t last;             // delete: last spill row of the window (else -1)
    int32_t star_f, star_r;   // delete: '*' columns negated on spill rows
    int32_t pad;
};

__host__ __device__ inline bool is_acgt(char c) {
    c &= ~0x20;
    return c == 'A' || c == 'C' || c == 'G' || c == 'T';
}
__host__ __device__ inline int up(char c) { return (c >= 'a' && c <= 'z') ? c - 32 : c; }
// column of `symbol` for a strand, -1 if the reference base is not A/C/G/T (region_summary.cpp:201-230)
__host__ __device__ inline int symbol_column(char ref_base, char symbol, bool reverse) {
    if (!is_acgt(ref_base)) return -1;
    const int first = reverse ? 19 : 8;
    switch (up(symbol)) {
        case 'A': return first;
        case 'C': return first + 1;
        case 'G': return first + 2;
        case 'T': return first + 3;
        case 'I': return first + 4;
        case 'D': return first + 5;
        default: return first + 6;
    }
}
__host__ __device__ inline int base_code(char c) {
    switch (up(c)) {
        case 'A': return 1;
        case 'C': return 2;
        case 'G': return 3;
        case 'T': return 4;
        default: return 5;
    }
}

__global__ __launch_bounds__(256) void init_matrix_kernel(int* __restrict__ mat, const char* __restrict__ ref,
                                                          int64_t ref_len, int L) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i > L) return;
    int4* row = reinterpret_cast<int4*>(mat + (size_t)i * ROW);
#pragma unroll
    for (int k = 0; k < ROW / 4; ++k) row[k] = make_int4(0, 0, 0, 0);
    if (i < L) mat[(size_t)i * ROW] = base_code(i < ref_len ? ref[i] : 'N');
}

// One wave per segment chunk of <= 64 consecutive positions, one lane per base (the host splits
// longer runs): 64 lanes hit 64 distinct matrix rows, so the int32 atomics of a wave never collide
// and the per-base work is fully parallel (a thread-per-segment version walked ~25 bases serially
// and took 1.07 ms for 5.7 M bases).
__global__ __launch_bounds__(256) void pileup_count_kernel(const Seg* __restrict__ segs, int nseg,
                                                           const char* __restrict__ seq,
                                                           const uint8_t* __restrict__ qual,
                                                           const char* __restrict__ ref, int64_t ref_len,
                                                           int* __restrict__ mat, int* __restrict__ snp_tab,
                                                           int* __restrict__ ovf_count, int4* __restrict__ ovf,
                                                           int ovf_cap, double min_snp_q) {
    const int s = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int i = threadIdx.x & 63;
    if (s >= nseg) return;
    const Seg sg = segs[s];
    if (i >= sg.n) return;
    const bool rev = sg.flags & SEG_REV;
    const int idx = sg.idx0 + i;
    const char rb = idx < ref_len ? ref[idx] : 'N';
    if (sg.flags & SEG_DEL) {
        const int col = symbol_column(rb, '*', rev);
        if (col >= 0) atomicSub(&mat[(size_t)idx * ROW + col], 1);
        return;
    }
    if (!((double)qual[sg.seq0 + i] >= min_snp_q)) return;
    const char base = seq[sg.seq0 + i];
    int* row = mat + (size_t)idx * ROW;
    atomicAdd(&row[C_COV], 1);
    if (!((sg.flags & SEG_ANCHOR) && i == sg.n - 1)) atomicSub(&row[rev ? 15 : 4], 1);
    const int col = symbol_column(rb, base, rev);
    if (col >= 0) atomicSub(&row[col], 1);
    if (rb != base) {                       // case-sensitive, as the reference compares
        atomicAdd(&row[C_SNP], 1);
        const int k = base == 'A' ? 0 : base == 'C' ? 1 : base == 'G' ? 2 : base == 'T' ? 3 : -1;
        if (k >= 0) {
            atomicAdd(&snp_tab[((size_t)idx * 2 + (rev ? 1 : 0)) * 4 + k], 1);
        } else {                            // rare alphabet (N, IUPAC, lower case): exact key kept on host
            const int slot = atomicAdd(ovf_count, 1);
            if (slot < ovf_cap) ovf[slot] = make_int4(idx, (int)(unsigned char)base, rev ? 1 : 0, 0);
        }
    }
}

// The whole per-read walk of populate_summary_matrix (region_summary.cpp:337-566) on the device: one wave per read steps
// through its CIGAR operations (wave-uniform scalars), the lanes take the bases of a match run / the rows of a deletion
// 64 at a time with the same per-base arithmetic as pileup_count_kernel, insert quality sums are wave reductions, the
// sparse indel updates go straight into the matrix, and every indel allele vote is appended as (row, type, strand,
// length, where the allele's bytes live) for the host, which only ever builds strings for the sites that pass the
// thresholds.  Replaces the host pass over CIGAR operations + Seg / Event uploads (0.9 of 1.5 ms per 10 kb interval of
// 60x long reads in round 1).  Integer atomics commute, so the matrix is bit-identical whatever the order.
struct DVote { int32_t idx, type_rev, len, from_ref; int64_t off; };   // type_rev = type char | strand << 8

struct WalkArgs {
    const int64_t* read_pos; const uint8_t* read_reverse; const int32_t* read_mapq; const int64_t* seq_offset;
    const int64_t* cigar_offset; const int32_t* cigar_op; const int32_t* cigar_len;
    const char* seq; const uint8_t* qual; const char* ref;
    int64_t ref_len, start, end;
    int n_reads;
    int* mat; int* snp_tab; int* counters; int4* ovf; int ovf_cap; DVote* votes; int vote_cap;
    double min_snp_q, min_indel_q;
};

__global__ __launch_bounds__(256) void cigar_walk_kernel(WalkArgs a) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= a.n_reads) return;
    if (a.read_mapq[r] <= 0) return;
    const bool rev = a.read_reverse[r] != 0;
    const int64_t s0 = a.seq_offset[r], read_len = a.seq_offset[r + 1] - s0;
    const int64_t c0 = a.cigar_offset[r], c1 = a.cigar_offset[r + 1];
    const int64_t start = a.start, end = a.end;
    int64_t ri = 0, pos = a.read_pos[r];
    auto refc = [&](int64_t idx) { return idx >= 0 && idx < a.ref_len ? a.ref[idx] : 'N'; };
    for (int64_t c = c0; c < c1; ++c) {
        if (pos > end) break;
        const int op = a.cigar_op[c];
        const int64_t len = a.cigar_len[c];
        if (op == OP_M || op == OP_EQ || op == OP_X) {
            const int64_t lo = pos > start ? pos : start, hi = pos + len - 1 < end ? pos + len - 1 : end;
            if (lo <= hi) {
                if (ri + (hi - pos) >= read_len) {          // CIGAR runs past the sequence: reported by the host
                    if (lane == 0) atomicMax(&a.counters[3], r + 1);
                    return;
                }
                bool anchor = false;
                if (hi == pos + len - 1 && c != c1 - 1) {
                    const int nop = a.cigar_op[c + 1];
                    anchor = (nop == OP_I || nop == OP_D);
                }
                for (int64_t q = lo + lane; q <= hi; q += 64) {
                    const int64_t si = s0 + ri + (q - pos);
                    if (!((double)a.qual[si] >= a.min_snp_q)) continue;
                    const int idx = (int)(q - start);
                    const char rb = refc(idx);
                    const char base = a.seq[si];
                    int* row = a.mat + (size_t)idx * ROW;
                    atomicAdd(&row[C_COV], 1);
                    if (!(anchor && q == hi)) atomicSub(&row[rev ? 15 : 4], 1);
                    const int col = symbol_column(rb, base, rev);
                    if (col >= 0) atomicSub(&row[col], 1);
                    if (rb != base) {
                        atomicAdd(&row[C_SNP], 1);
                        const int k = base == 'A' ? 0 : base == 'C' ? 1 : base == 'G' ? 2 : base == 'T' ? 3 : -1;
                        if (k >= 0) {
                            atomicAdd(&a.snp_tab[((size_t)idx * 2 + (rev ? 1 : 0)) * 4 + k], 1);
                        } else {
                            const int slot = atomicAdd(&a.counters[0], 1);
                            if (slot < a.ovf_cap) a.ovf[slot] = make_int4(idx, (int)(unsigned char)base, rev ? 1 : 0, 0);
                        }
                    }
                }
            }
            ri += len;
            pos += len;
        } else if (op == OP_I) {
            const int64_t anchor = pos - 1;
            if (anchor >= start && anchor <= end && ri - 1 >= 0) {
                const int idx = (int)(anchor - start);
                const int64_t n = len + 1;
                const int64_t avail = n < read_len - (ri - 1) ? n : (read_len - (ri - 1) > 0 ? read_len - (ri - 1) : 0);
                // sum of integer qualities: exact, and equal to the reference's double accumulation
                long long part = 0;
                for (int64_t k = ri - 1 + lane; k < ri - 1 + n; k += 64) part += k < read_len ? a.qual[s0 + k] : 0;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
                const bool passes = (double)part >= a.min_indel_q * (double)n;
                if (lane == 0) {
                    if (passes && (double)a.qual[s0 + ri - 1] < a.min_snp_q) atomicAdd(&a.mat[(size_t)idx * ROW + C_COV], 1);
                    if (avail + 1 <= 61 && passes) {
                        const int col = symbol_column(refc(idx), 'I', rev);
                        if (col >= 0) atomicSub(&a.mat[(size_t)idx * ROW + col], 1);
                        atomicAdd(&a.mat[(size_t)idx * ROW + C_INS], 1);
                        const int slot = atomicAdd(&a.counters[2], 1);
                        if (slot < a.vote_cap) a.votes[slot] = DVote{idx, (int)'2' | (rev ? 256 : 0), (int)avail, 0, s0 + ri - 1};
                    }
                }
            }
            ri += len;
        } else if (op == OP_D) {
            const int64_t anchor = pos - 1;
            if (anchor >= start && anchor <= end && lane == 0) {
                const int idx = (int)(anchor - start);
                const int col = symbol_column(refc(idx), 'D', rev);
                if (col >= 0) atomicSub(&a.mat[(size_t)idx * ROW + col], 1);
                int64_t avail = len + 1 < a.ref_len - idx ? len + 1 : a.ref_len - idx;
                if (avail < 0) avail = 0;
                if (avail + 1 <= 61) {
                    atomicAdd(&a.mat[(size_t)idx * ROW + C_DEL], 1);
                    const int slot = atomicAdd(&a.counters[2], 1);
                    if (slot < a.vote_cap) a.votes[slot] = DVote{idx, (int)'3' | (rev ? 256 : 0), (int)avail, 1, (int64_t)idx};
                }
            }
            const int64_t lo = pos > start ? pos : start, hi = pos + len - 1 < end ? pos + len - 1 : end;
            for (int64_t q = lo + lane; q <= hi; q += 64) {
                const int idx = (int)(q - start);
                const int col = symbol_column(refc(idx), '*', rev);
                if (col >= 0) atomicSub(&a.mat[(size_t)idx * ROW + col], 1);
            }
            pos += len;
        } else if (op == OP_N || op == OP_P) {
            pos += len;
            ri += len;      // the reference falls through into the soft-clip case (region_summary.cpp:556-561)
        } else if (op == OP_S) {
            ri += len;
        }
    }
}

// votes of the sites that passed the thresholds, compacted for the host (a few per cent of all votes)
// (the number of votes is only known on the device: the grid covers the capacity, counters[2] bounds it)
__global__ __launch_bounds__(256) void compact_votes_kernel(const DVote* __restrict__ votes, int* __restrict__ counters, int cap,
                                                            const uint8_t* __restrict__ pass, DVote* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int n = counters[2] < cap ? counters[2] : cap;
    if (i >= n) return;
    const DVote v = votes[i];
    if (pass[v.idx]) out[atomicAdd(&counters[4], 1)] = v;
}

__global__ __launch_bounds__(256) void apply_events_kernel(const Event* __restrict__ ev, int n, int* __restrict__ mat) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) atomicAdd(&mat[(size_t)ev[i].row * ROW + ev[i].col], ev[i].delta);
}

__global__ __launch_bounds__(256) void site_threshold_kernel(int* __restrict__ mat, const int* __restrict__ snp_tab, int L,
                                                             int64_t region_start, int64_t cand_start, int64_t cand_end,
                                                             double snp_thr, double ins_thr, double del_thr,
                                                             double min_cov, int* __restrict__ site_count,
                                                             SiteRec* __restrict__ sites, int site_cap,
                                                             uint8_t* __restrict__ pass = nullptr) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= L) return;
    if (pass) pass[idx] = 0;
    int* row = mat + (size_t)idx * ROW;
    const int cov = row[C_COV];
    const double c = cov > 1 ? (double)cov : 1.0;
    const bool s = (double)row[C_SNP] / c >= snp_thr;
    const bool n = (double)row[C_INS] / c >= ins_thr;
    const bool d = (double)row[C_DEL] / c >= del_thr;
    const int64_t pos = region_start + idx;
    if ((s || n || d) && pos >= cand_start && pos <= cand_end && (double)cov >= min_cov) {
        const int slot = atomicAdd(site_count, 1);
        if (pass) pass[idx] = 1;
        if (slot < site_cap) {
            SiteRec r;
            r.idx = idx;
            r.cov = cov;
            r.flags = (s ? 1 : 0) | (n ? 2 : 0) | (d ? 4 : 0);
            for (int k = 0; k < 4; ++k) {
                r.fwd[k] = snp_tab[((size_t)idx * 2 + 0) * 4 + k];
                r.rev[k] = snp_tab[((size_t)idx * 2 + 1) * 4 + k];
            }
            sites[slot] = r;
        }
    }
    for (int col = 11; col < 25; ++col) row[col] = max(-MAXC, min(MAXC, row[col]));
}

// one 64-lane workgroup per candidate: 33 x 26 = 858 cells
__global__ __launch_bounds__(64) void gather_windows_kernel(const int* __restrict__ mat, const CandDesc* __restrict__ cands,
                                                            int L, int W, int F, int mid, int* __restrict__ out32,
                                                            int8_t* __restrict__ out8) {
    const CandDesc cd = cands[blockIdx.x];
    const size_t base = (size_t)blockIdx.x * W * F;
    for (int e = threadIdx.x; e < W * F; e += 64) {
        const int r = e / F, f = e - r * F;
        const int row = cd.idx - mid + r;
        int v = (row >= 0 && row <= L && f < 26) ? mat[(size_t)row * ROW + f] : 0;
        if (r == mid) {
            if (f == cd.vcol) v = cd.vval;
            else if (f == cd.type + 4) v = cd.fwd;        // 5 / 6 / 7
            else if (f == cd.type + 15) v = cd.rev;       // 16 / 17 / 18
            else if (f == cd.neg_f || f == cd.neg_r) v = -v;
        } else if (r > mid && r <= cd.last) {
            if (f == 3) v = cd.vval;
            else if (f == 7) v = cd.fwd;
            else if (f == 18) v = cd.rev;
            else if (f == cd.star_f || f == cd.star_r) v = -v;
        }
        out32[base + e] = v;
        out8[base + e] = (int8_t)v;                        // two's-complement wrap, as numpy 1.22's int8 cast
    }
}

// ---- polish encoder -----------------------------------------------------------------------------
constexpr uint32_t PSEG_REV = 1, PSEG_GAP = 2, PSEG_INS = 4;
constexpr int PROW = 16, PC_COV = 10;       // base counts int32 [L][16]: 10 features + coverage

struct PSeg {
    int64_t seq0;
    int32_t idx0, n;      // MATCH/GAP: first position row; INS: first insert-slot row
    uint32_t flags;
    int32_t cov_idx;      // GAP: row credited with coverage (deletion start), -1 if outside the region
};
struct PRow { int32_t idx, slot; };   // output row -> (position row, 0 = base row / k = insert slot k)

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

__global__ __launch_bounds__(256) void polish_count_kernel(const PSeg* __restrict__ segs, int nseg,
                                                           const char* __restrict__ seq, int* __restrict__ base,
                                                           int* __restrict__ ins) {
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= nseg) return;
    const PSeg sg = segs[s];
    const bool rev = sg.flags & PSEG_REV;
    if (sg.flags & PSEG_GAP) {
        const int col = rev ? 8 : 9;
        for (int i = 0; i < sg.n; ++i) atomicAdd(&base[(size_t)(sg.idx0 + i) * PROW + col], 1);
        if (sg.cov_idx >= 0) atomicAdd(&base[(size_t)sg.cov_idx * PROW + PC_COV], sg.n);
    } else if (sg.flags & PSEG_INS) {
        for (int i = 0; i < sg.n; ++i)
            atomicAdd(&ins[(size_t)(sg.idx0 + i) * PROW + polish_feature(seq[sg.seq0 + i], rev)], 1);
    } else {
        for (int i = 0; i < sg.n; ++i) {
            int* row = base + (size_t)(sg.idx0 + i) * PROW;
            atomicAdd(&row[polish_feature(seq[sg.seq0 + i], rev)], 1);
            atomicAdd(&row[PC_COV], 1);
        }
    }
}

__global__ __launch_bounds__(256) void polish_pixels_kernel(const PRow* __restrict__ rows, int nrows,
                                                            const int* __restrict__ base, const int* __restrict__ ins,
                                                            const int* __restrict__ ins_row0, uint8_t* __restrict__ out) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= nrows) return;
    const PRow pr = rows[r];
    const int cov = pr.idx >= 0 ? base[(size_t)pr.idx * PROW + PC_COV] : 0;
    const double c = cov > 1 ? (double)cov : 1.0;
    const int* src = pr.idx < 0 ? nullptr
                                : (pr.slot == 0 ? base + (size_t)pr.idx * PROW
                                                : ins + (size_t)(ins_row0[pr.idx] + pr.slot - 1) * PROW);
#pragma unroll
    for (int j = 0; j < 10; ++j) {
        const double v = src ? ((double)src[j] / c) * 254.0 : 0.0;
        out[(size_t)r * 10 + j] = (uint8_t)((long long)v & 0xff);   // double -> uint8 as x86-64 gcc truncates
    }
}

struct DBuf {
    void* p = nullptr;
    size_t bytes = 0;
    bool ensure(size_t need) {
        if (need <= bytes) return true;
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

struct Tally { int total = 0, fwd = 0, rev = 0; };

}  // namespace

struct pa_encoder {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    DBuf d_seq, d_qual, d_ref, d_segs, d_events, d_mat, d_snp, d_ovf, d_counters, d_sites, d_cands, d_img32, d_img8;
    DBuf d_reads, d_cig, d_votes, d_votes_out, d_pass;     // device CIGAR walk: read tables, operations, allele votes
    DBuf d_pbase, d_pins, d_prow0, d_prows, d_ppix;
    int64_t p_rows = 0;
    std::vector<int64_t> p_positions;
    // results of the last call
    int64_t n = 0;
    int W = 33, F = 26;
    std::vector<int64_t> positions;
    std::vector<int32_t> depths, freqs;
    std::string names;
};

#define ENC_HIP(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) return pa::set_error(PA_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)
#define ENC_ALLOC(buf, bytes_)                                                                          \
    do {                                                                                                \
        if (!(buf).ensure(bytes_)) return pa::set_error(PA_ERR_HIP, "hipMalloc failed in encoder workspace"); \
    } while (0)

extern "C" {

int pa_encoder_create(int32_t device, void* hip_stream, pa_encoder** out) {
    if (!out) return pa::set_error(PA_ERR_INVALID, "null argument");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return pa::set_error(PA_ERR_NO_DEVICE, "no HIP device visible: the pepper_amd encoder has no CPU fallback");
    if (device < 0 || device >= count) return pa::set_error(PA_ERR_INVALID, "device ordinal out of range");
    ENC_HIP(hipSetDevice(device));
    auto* e = new pa_encoder();
    e->device = device;
    if (hip_stream) e->stream = static_cast<hipStream_t>(hip_stream);
    else {
        if (hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking) != hipSuccess) {
            delete e;
            return pa::set_error(PA_ERR_HIP, "hipStreamCreate failed");
        }
        e->own_stream = true;
    }
    *out = e;
    return PA_OK;
}

void pa_encoder_destroy(pa_encoder* e) {
    if (!e) return;
    (void)hipSetDevice(e->device);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    if (e->own_stream && e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
}

int pa_encoder_generate_summary(pa_encoder* e, const pa_pileup* p, const pa_summary_params* q, int64_t* n_candidates) {
    if (!e || !p || !q || !n_candidates) return pa::set_error(PA_ERR_INVALID, "null argument");
    if (p->region_end < p->region_start || p->region_end - p->region_start > (int64_t)1 << 28)
        return pa::set_error(PA_ERR_INVALID, "bad region");
    if (q->feature_size < 26 || q->candidate_window_size < 2 || q->candidate_window_size > 254)
        return pa::set_error(PA_ERR_INVALID, "feature_size must be >= 26 and 2 <= candidate_window_size <= 254");
    ENC_HIP(hipSetDevice(e->device));
    const int64_t start = p->region_start, end = p->region_end;
    const int L = (int)(end - start + 1);
    const int W = q->candidate_window_size + 1, F = q->feature_size, mid = q->candidate_window_size / 2;
    e->W = W;
    e->F = F;
    e->n = 0;
    e->positions.clear();
    e->depths.clear();
    e->freqs.clear();
    e->names.clear();
    auto refc = [&](int64_t idx) { return idx >= 0 && idx < p->reference_len ? p->reference[idx] : 'N'; };
    static const bool trace = getenv("PA_ENCODER_TRACE") != nullptr;      // host phase times on stderr
    auto t_prev = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!trace) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[encoder] %-22s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
        t_prev = now;
    };

    // PA_ENCODER_HOST_CIGAR=1: round 1's split (host pass over CIGAR operations -> Seg / Event lists -> pileup_count_kernel +
    // apply_events_kernel); default: the whole walk on the device (cigar_walk_kernel)
    const char* host_cigar_env = getenv("PA_ENCODER_HOST_CIGAR");       // read per call: the tests run both paths
    const bool host_cigar = host_cigar_env && host_cigar_env[0] != '0';
    const int64_t n_ops = p->n_reads > 0 ? p->cigar_offset[p->n_reads] : 0;
    // ---- host pass over CIGAR ops -----------------------------------------------------------
    std::vector<Seg> segs;
    std::vector<Event> events;
    if (host_cigar) {
        segs.reserve((size_t)n_ops + 1024);
        events.reserve((size_t)n_ops * 2 + 1024);
    }
    // Indel allele votes are only recorded here (site, type, where the allele's bytes live); the ordered
    // per-site maps of allele strings the reference keeps for EVERY position (region_summary.cpp:431-555)
    // are built later and only for the few sites that pass the thresholds -- one string allocation and two
    // map look-ups per indel event were most of the host time of a region.
    struct IndelVote { int32_t idx; char type; bool rev; int32_t len; const char* src; };
    std::vector<IndelVote> votes;
    auto vote = [&](int32_t idx, char type, const char* src, int64_t len, bool rev) {
        votes.push_back({idx, type, rev, (int32_t)len, src});
    };
    const int64_t total_bases = p->n_reads > 0 ? p->seq_offset[p->n_reads] : 0;
    for (int32_t r = 0; host_cigar && r < p->n_reads; ++r) {
        if (p->read_mapq[r] <= 0) continue;
        const bool rev = p->read_reverse[r] != 0;
        const int64_t s0 = p->seq_offset[r], read_len = p->seq_offset[r + 1] - s0;
        const char* seq = p->seq + s0;
        const uint8_t* ql = p->qual + s0;
        const int64_t c0 = p->cigar_offset[r], c1 = p->cigar_offset[r + 1];
        int64_t ri = 0, pos = p->read_pos[r];
        for (int64_t c = c0; c < c1; ++c) {
            if (pos > end) break;
            const int op = p->cigar_op[c];
            const int64_t len = p->cigar_len[c];
            if (op == OP_M || op == OP_EQ || op == OP_X) {
                const int64_t lo = std::max(pos, start), hi = std::min(pos + len - 1, end);   // touched positions
                if (lo <= hi) {
                    if (ri + (hi - pos) >= read_len)
                        return pa::set_error(PA_ERR_INVALID, "CIGAR of read " + std::to_string(r) + " runs past its sequence");
                    bool anchor = false;
                    if (hi == pos + len - 1 && c != c1 - 1) {
                        const int nop = p->cigar_op[c + 1];
                        anchor = (nop == OP_I || nop == OP_D);
                    }
                    for (int64_t q0 = lo; q0 <= hi; q0 += 64) {       // one wave per <= 64 positions
                        const int64_t q1 = std::min(hi, q0 + 63);
                        Seg sg;
                        sg.seq0 = s0 + ri + (q0 - pos);
                        sg.idx0 = (int32_t)(q0 - start);
                        sg.n = (int32_t)(q1 - q0 + 1);
                        sg.flags = (rev ? SEG_REV : 0) | ((anchor && q1 == hi) ? SEG_ANCHOR : 0);
                        sg.pad = 0;
                        segs.push_back(sg);
                    }
                }
                ri += len;
                pos += len;
            } else if (op == OP_I) {
                const int64_t anchor = pos - 1;
                if (anchor >= start && anchor <= end && ri - 1 >= 0) {
                    const int32_t idx = (int32_t)(anchor - start);
                    const int64_t n = len + 1;
                    const int64_t avail = std::max<int64_t>(0, std::min<int64_t>(n, read_len - (ri - 1)));
                    double qsum = 0;
                    for (int64_t k = ri - 1; k < ri - 1 + n; ++k) qsum += k < read_len ? ql[k] : 0;
                    const bool passes = qsum >= q->min_indel_baseq * (double)n;
                    if (passes && (double)ql[ri - 1] < q->min_snp_baseq) events.push_back({idx, C_COV, 1, 0});
                    if (avail + 1 <= 61 && passes) {
                        const int col = symbol_column(refc(idx), 'I', rev);
                        if (col >= 0) events.push_back({idx, col, -1, 0});
                        events.push_back({idx, C_INS, 1, 0});
                        vote(idx, '2', seq + (ri - 1), avail, rev);
                    }
                }
                ri += len;
            } else if (op == OP_D) {
                const int64_t anchor = pos - 1;
                if (anchor >= start && anchor <= end) {
                    const int32_t idx = (int32_t)(anchor - start);
                    const int col = symbol_column(refc(idx), 'D', rev);
                    if (col >= 0) events.push_back({idx, col, -1, 0});
                    const int64_t avail = std::max<int64_t>(0, std::min<int64_t>(len + 1, p->reference_len - idx));
                    if (avail + 1 <= 61) {
                        events.push_back({idx, C_DEL, 1, 0});
                        vote(idx, '3', p->reference + idx, avail, rev);
                    }
                }
                const int64_t lo = std::max(pos, start), hi = std::min(pos + len - 1, end);
                for (int64_t q0 = lo; q0 <= hi; q0 += 64)
                    segs.push_back({0, (int32_t)(q0 - start), (int32_t)(std::min(hi, q0 + 63) - q0 + 1),
                                    (rev ? SEG_REV : 0) | SEG_DEL, 0});
                pos += len;
            } else if (op == OP_N || op == OP_P) {
                pos += len;
                ri += len;      // the reference falls through into the soft-clip case (region_summary.cpp:556-561)
            } else if (op == OP_S) {
                ri += len;
            }
        }
    }

    // ---- device: counts -----------------------------------------------------------------------
    const int ovf_cap = 1 << 16;
    const int site_cap = L;
    ENC_ALLOC(e->d_seq, (size_t)total_bases + 16);
    ENC_ALLOC(e->d_qual, (size_t)total_bases + 16);
    ENC_ALLOC(e->d_ref, (size_t)p->reference_len + 16);
    ENC_ALLOC(e->d_segs, segs.size() * sizeof(Seg) + 16);
    lap("cigar pass");
    ENC_ALLOC(e->d_events, events.size() * sizeof(Event) + 16);
    ENC_ALLOC(e->d_mat, (size_t)(L + 1) * ROW * sizeof(int));
    ENC_ALLOC(e->d_snp, (size_t)L * 8 * sizeof(int));
    ENC_ALLOC(e->d_ovf, (size_t)ovf_cap * sizeof(int4));
    ENC_ALLOC(e->d_counters, 64);
    ENC_ALLOC(e->d_sites, (size_t)site_cap * sizeof(SiteRec));
    hipStream_t st = e->stream;
    if (total_bases > 0) {
        ENC_HIP(hipMemcpyAsync(e->d_seq.p, p->seq, (size_t)total_bases, hipMemcpyHostToDevice, st));
        ENC_HIP(hipMemcpyAsync(e->d_qual.p, p->qual, (size_t)total_bases, hipMemcpyHostToDevice, st));
    }
    if (p->reference_len > 0)
        ENC_HIP(hipMemcpyAsync(e->d_ref.p, p->reference, (size_t)p->reference_len, hipMemcpyHostToDevice, st));
    if (!segs.empty())
        ENC_HIP(hipMemcpyAsync(e->d_segs.p, segs.data(), segs.size() * sizeof(Seg), hipMemcpyHostToDevice, st));
    if (!events.empty())
        ENC_HIP(hipMemcpyAsync(e->d_events.p, events.data(), events.size() * sizeof(Event), hipMemcpyHostToDevice, st));
    ENC_HIP(hipMemsetAsync(e->d_snp.p, 0, (size_t)L * 8 * sizeof(int), st));
    ENC_HIP(hipMemsetAsync(e->d_counters.p, 0, 64, st));
    int* mat = static_cast<int*>(e->d_mat.p);
    int* counters = static_cast<int*>(e->d_counters.p);
    hipLaunchKernelGGL(init_matrix_kernel, dim3((L + 1 + 255) / 256), dim3(256), 0, st, mat,
                       static_cast<const char*>(e->d_ref.p), p->reference_len, L);
    uint8_t* d_pass = nullptr;
    if (!host_cigar) {
        // read tables and operations: [pos i64 n][seq_offset i64 n+1][cigar_offset i64 n+1][mapq i32 n][reverse u8 n]
        const size_t n = (size_t)p->n_reads;
        const size_t o_pos = 0, o_soff = o_pos + 8 * n, o_coff = o_soff + 8 * (n + 1), o_mapq = o_coff + 8 * (n + 1),
                     o_rev = o_mapq + 4 * n, reads_bytes = o_rev + n;
        ENC_ALLOC(e->d_reads, reads_bytes + 64);
        ENC_ALLOC(e->d_cig, (size_t)n_ops * 8 + 64);
        ENC_ALLOC(e->d_votes, (size_t)(n_ops + 1) * sizeof(DVote));
        ENC_ALLOC(e->d_votes_out, (size_t)(n_ops + 1) * sizeof(DVote));
        ENC_ALLOC(e->d_pass, (size_t)L + 64);
        d_pass = static_cast<uint8_t*>(e->d_pass.p);
        char* dr = static_cast<char*>(e->d_reads.p);
        if (n > 0) {
            ENC_HIP(hipMemcpyAsync(dr + o_pos, p->read_pos, 8 * n, hipMemcpyHostToDevice, st));
            ENC_HIP(hipMemcpyAsync(dr + o_soff, p->seq_offset, 8 * (n + 1), hipMemcpyHostToDevice, st));
            ENC_HIP(hipMemcpyAsync(dr + o_coff, p->cigar_offset, 8 * (n + 1), hipMemcpyHostToDevice, st));
            ENC_HIP(hipMemcpyAsync(dr + o_mapq, p->read_mapq, 4 * n, hipMemcpyHostToDevice, st));
            ENC_HIP(hipMemcpyAsync(dr + o_rev, p->read_reverse, n, hipMemcpyHostToDevice, st));
        }
        char* dc = static_cast<char*>(e->d_cig.p);
        if (n_ops > 0) {
            ENC_HIP(hipMemcpyAsync(dc, p->cigar_op, (size_t)n_ops * 4, hipMemcpyHostToDevice, st));
            ENC_HIP(hipMemcpyAsync(dc + (size_t)n_ops * 4, p->cigar_len, (size_t)n_ops * 4, hipMemcpyHostToDevice, st));
        }
        if (n > 0) {
            WalkArgs wa;
            wa.read_pos = reinterpret_cast<const int64_t*>(dr + o_pos);
            wa.read_reverse = reinterpret_cast<const uint8_t*>(dr + o_rev);
            wa.read_mapq = reinterpret_cast<const int32_t*>(dr + o_mapq);
            wa.seq_offset = reinterpret_cast<const int64_t*>(dr + o_soff);
            wa.cigar_offset = reinterpret_cast<const int64_t*>(dr + o_coff);
            wa.cigar_op = reinterpret_cast<const int32_t*>(dc);
            wa.cigar_len = reinterpret_cast<const int32_t*>(dc + (size_t)n_ops * 4);
            wa.seq = static_cast<const char*>(e->d_seq.p);
            wa.qual = static_cast<const uint8_t*>(e->d_qual.p);
            wa.ref = static_cast<const char*>(e->d_ref.p);
            wa.ref_len = p->reference_len;
            wa.start = start;
            wa.end = end;
            wa.n_reads = p->n_reads;
            wa.mat = mat;
            wa.snp_tab = static_cast<int*>(e->d_snp.p);
            wa.counters = counters;
            wa.ovf = static_cast<int4*>(e->d_ovf.p);
            wa.ovf_cap = ovf_cap;
            wa.votes = static_cast<DVote*>(e->d_votes.p);
            wa.vote_cap = (int)n_ops;
            wa.min_snp_q = q->min_snp_baseq;
            wa.min_indel_q = q->min_indel_baseq;
            hipLaunchKernelGGL(cigar_walk_kernel, dim3((p->n_reads + 3) / 4), dim3(256), 0, st, wa);
        }
    }
    if (!segs.empty())
        hipLaunchKernelGGL(pileup_count_kernel, dim3(((int)segs.size() + 3) / 4), dim3(256), 0, st,
                           static_cast<const Seg*>(e->d_segs.p), (int)segs.size(), static_cast<const char*>(e->d_seq.p),
                           static_cast<const uint8_t*>(e->d_qual.p), static_cast<const char*>(e->d_ref.p),
                           p->reference_len, mat, static_cast<int*>(e->d_snp.p), counters,
                           static_cast<int4*>(e->d_ovf.p), ovf_cap, q->min_snp_baseq);
    if (!events.empty())
        hipLaunchKernelGGL(apply_events_kernel, dim3(((int)events.size() + 255) / 256), dim3(256), 0, st,
                           static_cast<const Event*>(e->d_events.p), (int)events.size(), mat);
    hipLaunchKernelGGL(site_threshold_kernel, dim3((L + 255) / 256), dim3(256), 0, st, mat,
                       static_cast<const int*>(e->d_snp.p), L, start, q->candidate_region_start,
                       q->candidate_region_end, q->snp_freq_threshold, q->insert_freq_threshold,
                       q->delete_freq_threshold, q->min_coverage_threshold, counters + 1,
                       static_cast<SiteRec*>(e->d_sites.p), site_cap, d_pass);
    if (!host_cigar && n_ops > 0)      // votes of the passing sites only (vote count read on the device: grid over the capacity)
        hipLaunchKernelGGL(compact_votes_kernel, dim3(((int)n_ops + 255) / 256), dim3(256), 0, st,
                           static_cast<const DVote*>(e->d_votes.p), counters, (int)n_ops, d_pass,
                           static_cast<DVote*>(e->d_votes_out.p));
    ENC_HIP(hipGetLastError());
    int host_counters[5] = {0, 0, 0, 0, 0};
    lap("uploads + launches");
    ENC_HIP(hipMemcpyAsync(host_counters, counters, sizeof(host_counters), hipMemcpyDeviceToHost, st));
    ENC_HIP(hipStreamSynchronize(st));
    lap("count kernels");
    if (host_counters[3] > 0)
        return pa::set_error(PA_ERR_INVALID, "CIGAR of read " + std::to_string(host_counters[3] - 1) + " runs past its sequence");
    const int n_ovf = host_counters[0], n_sites = std::min(host_counters[1], site_cap);
    if (n_ovf > ovf_cap)
        return pa::set_error(PA_ERR_INVALID, "more than 65536 mismatching bases outside ACGT in one region");
    std::vector<SiteRec> sites((size_t)n_sites);
    std::vector<int4> ovf((size_t)n_ovf);
    if (n_sites) ENC_HIP(hipMemcpyAsync(sites.data(), e->d_sites.p, sites.size() * sizeof(SiteRec), hipMemcpyDeviceToHost, st));
    if (n_ovf) ENC_HIP(hipMemcpyAsync(ovf.data(), e->d_ovf.p, ovf.size() * sizeof(int4), hipMemcpyDeviceToHost, st));
    std::vector<DVote> dvotes;
    if (!host_cigar) {
        if (host_counters[2] > n_ops) return pa::set_error(PA_ERR_INVALID, "more indel votes than CIGAR operations (corrupt pileup)");
        dvotes.resize((size_t)host_counters[4]);
        if (!dvotes.empty())
            ENC_HIP(hipMemcpyAsync(dvotes.data(), e->d_votes_out.p, dvotes.size() * sizeof(DVote), hipMemcpyDeviceToHost, st));
    }
    ENC_HIP(hipStreamSynchronize(st));
    for (const DVote& v : dvotes)
        votes.push_back({v.idx, (char)(v.type_rev & 0xff), (v.type_rev >> 8) != 0, v.len,
                         (v.from_ref ? p->reference : p->seq) + v.off});
    std::sort(sites.begin(), sites.end(), [](const SiteRec& a, const SiteRec& b) { return a.idx < b.idx; });
    std::map<int32_t, std::map<char, Tally>> rare;       // SNP alleles outside ACGT
    for (const int4& o : ovf) {
        Tally& t = rare[o.x][(char)o.y];
        t.total += 1;
        (o.z ? t.rev : t.fwd) += 1;
    }

    // votes bucketed by site (counting sort, stable: order of arrival does not matter for the tallies)
    std::vector<int32_t> vote_begin((size_t)L + 2, 0);
    for (const IndelVote& v : votes) vote_begin[(size_t)v.idx + 1] += 1;
    for (int i = 0; i <= L; ++i) vote_begin[(size_t)i + 1] += vote_begin[i];
    std::vector<int32_t> vote_order(votes.size());
    {
        std::vector<int32_t> cursor(vote_begin.begin(), vote_begin.end() - 1);
        for (size_t k = 0; k < votes.size(); ++k) vote_order[(size_t)cursor[votes[k].idx]++] = (int32_t)k;
    }

    // ---- host: candidates in the reference's order (std::set<string> per site) -------------------
    std::vector<CandDesc> cands;
    for (const SiteRec& s : sites) {
        const int depth = std::min(s.cov, MAXC);
        const char rb = refc(s.idx);
        auto accept = [&](char type, const Tally& t) {
            const double freq = (double)t.total / std::max(1.0, (double)depth);
            if ((double)t.total < q->candidate_support_threshold) return false;
            if (type != '1' && freq < q->indel_candidate_freq_threshold) return false;
            if (type == '1' && freq < q->snp_candidate_freq_threshold) return false;
            if (type != '1' && q->skip_indels) return false;
            if ((type == '1' && !(s.flags & 1)) || (type == '2' && !(s.flags & 2)) || (type == '3' && !(s.flags & 4)))
                return false;
            return true;
        };
        auto emit = [&](const std::string& key, const Tally& t, const CandDesc& d) {
            cands.push_back(d);
            e->positions.push_back(start + s.idx);
            e->depths.push_back(depth);
            e->freqs.push_back(std::min(t.total, MAXC));
            e->names += key;
            e->names.push_back('\0');
        };
        // SNP alleles: "1" + base, ordered by the raw base character
        std::map<char, Tally> snps;
        const char acgt[4] = {'A', 'C', 'G', 'T'};
        for (int k = 0; k < 4; ++k)
            if (s.fwd[k] + s.rev[k] > 0) snps[acgt[k]] = Tally{s.fwd[k] + s.rev[k], s.fwd[k], s.rev[k]};
        const auto rit = rare.find(s.idx);
        if (rit != rare.end())
            for (const auto& kv : rit->second) {
                Tally& t = snps[kv.first];
                t.total += kv.second.total;
                t.fwd += kv.second.fwd;
                t.rev += kv.second.rev;
            }
        for (const auto& kv : snps) {
            if (!accept('1', kv.second)) continue;
            CandDesc d{};
            d.idx = s.idx; d.type = 1; d.vcol = 1; d.vval = base_code(kv.first);
            d.fwd = std::min(kv.second.fwd, MAXC); d.rev = std::min(kv.second.rev, MAXC);
            d.neg_f = symbol_column(rb, kv.first, false); d.neg_r = symbol_column(rb, kv.first, true);
            d.last = -1; d.star_f = d.star_r = -1;
            emit(std::string("1") + kv.first, kv.second, d);
        }
        if (vote_begin[(size_t)s.idx] == vote_begin[(size_t)s.idx + 1]) continue;
        std::map<std::string, Tally> site_indels;          // ordered allele keys ("2..." < "3...")
        for (int32_t k = vote_begin[(size_t)s.idx]; k < vote_begin[(size_t)s.idx + 1]; ++k) {
            const IndelVote& v = votes[(size_t)vote_order[(size_t)k]];
            std::string key(1, v.type);
            key.append(v.src, (size_t)v.len);
            Tally& t = site_indels[key];
            t.total += 1;
            (v.rev ? t.rev : t.fwd) += 1;
        }
        for (const auto& kv : site_indels) {
            const char type = kv.first[0];
            if (!accept(type, kv.second)) continue;
            CandDesc d{};
            d.idx = s.idx; d.type = type - '0';
            d.fwd = std::min(kv.second.fwd, MAXC); d.rev = std::min(kv.second.rev, MAXC);
            const int alen = (int)kv.first.size() - 1;
            d.vval = std::min(alen, MAXC);
            d.star_f = d.star_r = -1;
            if (type == '2') {
                d.vcol = 2; d.last = -1;
                d.neg_f = symbol_column(rb, 'I', false); d.neg_r = symbol_column(rb, 'I', true);
            } else {
                d.vcol = 3; d.last = std::min(mid + alen - 1, q->candidate_window_size - 1);
                d.neg_f = symbol_column(rb, 'D', false); d.neg_r = symbol_column(rb, 'D', true);
                d.star_f = symbol_column(rb, '*', false); d.star_r = symbol_column(rb, '*', true);
            }
            emit(kv.first, kv.second, d);
        }
    }

    lap("candidate enumeration");
    // ---- device: window gather ----------------------------------------------------------------------
    e->n = (int64_t)cands.size();
    *n_candidates = e->n;
    if (e->n > 0) {
        ENC_ALLOC(e->d_cands, cands.size() * sizeof(CandDesc));
        ENC_ALLOC(e->d_img32, (size_t)e->n * W * F * sizeof(int));
        ENC_ALLOC(e->d_img8, (size_t)e->n * W * F);
        ENC_HIP(hipMemcpyAsync(e->d_cands.p, cands.data(), cands.size() * sizeof(CandDesc), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(gather_windows_kernel, dim3((unsigned)e->n), dim3(64), 0, st, mat,
                           static_cast<const CandDesc*>(e->d_cands.p), L, W, F, mid, static_cast<int*>(e->d_img32.p),
                           static_cast<int8_t*>(e->d_img8.p));
        ENC_HIP(hipGetLastError());
        ENC_HIP(hipStreamSynchronize(st));
    }
    lap("window gather");
    return PA_OK;
}

int pa_encoder_get_results(pa_encoder* e, int64_t* positions, int32_t* depths, int32_t* candidate_frequency,
                           int32_t* images_i32, int8_t* images_i8, char* candidates, int64_t candidates_cap,
                           int64_t* candidates_needed) {
    if (!e) return pa::set_error(PA_ERR_INVALID, "null encoder");
    ENC_HIP(hipSetDevice(e->device));
    const size_t n = (size_t)e->n;
    if (positions) std::copy(e->positions.begin(), e->positions.end(), positions);
    if (depths) std::copy(e->depths.begin(), e->depths.end(), depths);
    if (candidate_frequency) std::copy(e->freqs.begin(), e->freqs.end(), candidate_frequency);
    if (candidates_needed) *candidates_needed = (int64_t)e->names.size();
    if (candidates && candidates_cap >= (int64_t)e->names.size()) std::memcpy(candidates, e->names.data(), e->names.size());
    if (n > 0 && images_i32)
        ENC_HIP(hipMemcpyAsync(images_i32, e->d_img32.p, n * e->W * e->F * sizeof(int), hipMemcpyDeviceToHost, e->stream));
    if (n > 0 && images_i8)
        ENC_HIP(hipMemcpyAsync(images_i8, e->d_img8.p, n * e->W * e->F, hipMemcpyDeviceToHost, e->stream));
    ENC_HIP(hipStreamSynchronize(e->stream));
    return PA_OK;
}

int pa_polish_encoder_generate_summary(pa_encoder* e, const pa_pileup* p, int64_t start_pos, int64_t end_pos,
                                       int64_t* n_rows) {
    if (!e || !p || !n_rows) return pa::set_error(PA_ERR_INVALID, "null argument");
    if (p->region_end < p->region_start || p->region_end - p->region_start > (int64_t)1 << 28 || end_pos < start_pos)
        return pa::set_error(PA_ERR_INVALID, "bad region");
    ENC_HIP(hipSetDevice(e->device));
    const int64_t start = p->region_start, end = p->region_end;
    const int L = (int)(end - start + 1);
    std::vector<PSeg> segs;
    std::vector<int32_t> longest((size_t)L, 0);
    struct InsOp { int32_t idx; int32_t len; int64_t seq0; bool rev; };
    std::vector<InsOp> ins_ops;
    const int64_t total_bases = p->n_reads > 0 ? p->seq_offset[p->n_reads] : 0;
    for (int32_t r = 0; r < p->n_reads; ++r) {
        if (p->read_mapq[r] <= 0) continue;
        const bool rev = p->read_reverse[r] != 0;
        const int64_t s0 = p->seq_offset[r], read_len = p->seq_offset[r + 1] - s0;
        int64_t ri = 0, pos = p->read_pos[r];
        for (int64_t c = p->cigar_offset[r]; c < p->cigar_offset[r + 1]; ++c) {
            if (pos > end_pos) break;
            const int op = p->cigar_op[c];
            const int64_t len = p->cigar_len[c];
            if (op == OP_M || op == OP_EQ || op == OP_X) {
                const int64_t lo = std::max(pos, start), hi = std::min(pos + len - 1, end);
                if (lo <= hi) {
                    if (ri + (hi - pos) >= read_len)
                        return pa::set_error(PA_ERR_INVALID, "CIGAR of read " + std::to_string(r) + " runs past its sequence");
                    segs.push_back({s0 + ri + (lo - pos), (int32_t)(lo - start), (int32_t)(hi - lo + 1), rev ? PSEG_REV : 0u, -1});
                }
                ri += len;
                pos += len;
            } else if (op == OP_I) {
                const int64_t anchor = pos - 1;
                if (anchor >= start && anchor <= end) {
                    if (ri + len > read_len)
                        return pa::set_error(PA_ERR_INVALID, "insert of read " + std::to_string(r) + " runs past its sequence");
                    const int32_t idx = (int32_t)(anchor - start);
                    ins_ops.push_back({idx, (int32_t)len, s0 + ri, rev});
                    longest[(size_t)idx] = std::max<int32_t>(longest[(size_t)idx], (int32_t)len);
                }
                ri += len;
            } else if (op == OP_D || op == OP_N || op == OP_P) {
                const int64_t lo = std::max(pos, start), hi = std::min(pos + len - 1, end);
                if (lo <= hi)
                    segs.push_back({0, (int32_t)(lo - start), (int32_t)(hi - lo + 1), (rev ? PSEG_REV : 0u) | PSEG_GAP,
                                    (pos >= start && pos <= end) ? (int32_t)(pos - start) : -1});
                pos += len;
            } else if (op == OP_S) {
                ri += len;
            }
        }
    }
    // insert-slot rows: prefix sum of the longest insert per anchor
    std::vector<int32_t> ins_row0((size_t)L + 1, 0);
    for (int i = 0; i < L; ++i) ins_row0[(size_t)i + 1] = ins_row0[(size_t)i] + longest[(size_t)i];
    const int total_ins_rows = ins_row0[(size_t)L];
    for (const InsOp& io : ins_ops)
        segs.push_back({io.seq0, ins_row0[(size_t)io.idx], io.len, (io.rev ? PSEG_REV : 0u) | PSEG_INS, -1});
    // output rows in the reference's order: position, then its insert slots
    std::vector<PRow> rows;
    e->p_positions.clear();
    for (int64_t pos = start_pos; pos <= end_pos; ++pos) {
        const bool in = pos >= start && pos <= end;
        const int32_t idx = in ? (int32_t)(pos - start) : -1;
        rows.push_back({idx, 0});
        e->p_positions.push_back(pos);
        e->p_positions.push_back(0);
        const int32_t n_ins = in ? longest[(size_t)idx] : 0;
        for (int32_t k = 1; k <= n_ins; ++k) {
            rows.push_back({idx, k});
            e->p_positions.push_back(pos);
            e->p_positions.push_back(k);
        }
    }
    e->p_rows = (int64_t)rows.size();
    *n_rows = e->p_rows;

    hipStream_t st = e->stream;
    ENC_ALLOC(e->d_seq, (size_t)total_bases + 16);
    ENC_ALLOC(e->d_segs, segs.size() * sizeof(PSeg) + 16);
    ENC_ALLOC(e->d_pbase, (size_t)L * PROW * sizeof(int));
    ENC_ALLOC(e->d_pins, (size_t)(total_ins_rows + 1) * PROW * sizeof(int));
    ENC_ALLOC(e->d_prow0, (size_t)(L + 1) * sizeof(int));
    ENC_ALLOC(e->d_prows, rows.size() * sizeof(PRow) + 16);
    ENC_ALLOC(e->d_ppix, rows.size() * 10 + 16);
    if (total_bases > 0) ENC_HIP(hipMemcpyAsync(e->d_seq.p, p->seq, (size_t)total_bases, hipMemcpyHostToDevice, st));
    if (!segs.empty()) ENC_HIP(hipMemcpyAsync(e->d_segs.p, segs.data(), segs.size() * sizeof(PSeg), hipMemcpyHostToDevice, st));
    ENC_HIP(hipMemcpyAsync(e->d_prow0.p, ins_row0.data(), (size_t)(L + 1) * sizeof(int), hipMemcpyHostToDevice, st));
    ENC_HIP(hipMemcpyAsync(e->d_prows.p, rows.data(), rows.size() * sizeof(PRow), hipMemcpyHostToDevice, st));
    ENC_HIP(hipMemsetAsync(e->d_pbase.p, 0, (size_t)L * PROW * sizeof(int), st));
    ENC_HIP(hipMemsetAsync(e->d_pins.p, 0, (size_t)(total_ins_rows + 1) * PROW * sizeof(int), st));
    if (!segs.empty())
        hipLaunchKernelGGL(polish_count_kernel, dim3(((int)segs.size() + 255) / 256), dim3(256), 0, st,
                           static_cast<const PSeg*>(e->d_segs.p), (int)segs.size(), static_cast<const char*>(e->d_seq.p),
                           static_cast<int*>(e->d_pbase.p), static_cast<int*>(e->d_pins.p));
    hipLaunchKernelGGL(polish_pixels_kernel, dim3(((int)rows.size() + 255) / 256), dim3(256), 0, st,
                       static_cast<const PRow*>(e->d_prows.p), (int)rows.size(), static_cast<const int*>(e->d_pbase.p),
                       static_cast<const int*>(e->d_pins.p), static_cast<const int*>(e->d_prow0.p),
                       static_cast<uint8_t*>(e->d_ppix.p));
    ENC_HIP(hipGetLastError());
    ENC_HIP(hipStreamSynchronize(st));
    return PA_OK;
}

int pa_polish_encoder_get_results(pa_encoder* e, uint8_t* image, int64_t* positions) {
    if (!e) return pa::set_error(PA_ERR_INVALID, "null encoder");
    ENC_HIP(hipSetDevice(e->device));
    if (positions) std::copy(e->p_positions.begin(), e->p_positions.end(), positions);
    if (image && e->p_rows > 0) {
        ENC_HIP(hipMemcpyAsync(image, e->d_ppix.p, (size_t)e->p_rows * 10, hipMemcpyDeviceToHost, e->stream));
        ENC_HIP(hipStreamSynchronize(e->stream));
    }
    return PA_OK;
}

const int8_t* pa_encoder_device_images(pa_encoder* e) {
    return (e && e->n > 0) ? static_cast<const int8_t*>(e->d_img8.p) : nullptr;
}

}  // extern "C"
