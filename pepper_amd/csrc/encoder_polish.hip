// Polish summary encoder (include/pepper_amd_encoder.h): pileup -> uint8 summary rows.
// Reference: /root/reference/pepper/modules/src/pileup_summary/summary_generator.cpp:16-32 (feature index), 47-121
// (per-read walk), 274-306 (pixels), 370-393 (row order).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "encoder_common.h"

using namespace pa_enc;

namespace {

__host__ __device__ inline int up(char c) { return (c >= 'a' && c <= 'z') ? c - 32 : c; }

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
}  // namespace

struct pa_polish_batch {
    DBuf d_seq, d_segs, d_pbase, d_pins, d_prow0, d_prows, d_ppix;
    int64_t p_rows = 0;
    std::vector<int64_t> p_positions;
};

void pa_polish_batch_free(pa_polish_batch* b) { delete b; }

extern "C" {

int pa_polish_encoder_generate_summary(pa_encoder* e, const pa_pileup* p, int64_t start_pos, int64_t end_pos,
                                       int64_t* n_rows) {
    if (!e || !p || !n_rows) return pa::set_error(PA_ERR_INVALID, "null argument");
    if (p->region_end < p->region_start || p->region_end - p->region_start > (int64_t)1 << 28 || end_pos < start_pos)
        return pa::set_error(PA_ERR_INVALID, "bad region");
    ENC_HIP(hipSetDevice(e->device));
    if (!e->polish) e->polish = new pa_polish_batch();
    pa_polish_batch& b = *e->polish;
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
    b.p_positions.clear();
    for (int64_t pos = start_pos; pos <= end_pos; ++pos) {
        const bool in = pos >= start && pos <= end;
        const int32_t idx = in ? (int32_t)(pos - start) : -1;
        rows.push_back({idx, 0});
        b.p_positions.push_back(pos);
        b.p_positions.push_back(0);
        const int32_t n_ins = in ? longest[(size_t)idx] : 0;
        for (int32_t k = 1; k <= n_ins; ++k) {
            rows.push_back({idx, k});
            b.p_positions.push_back(pos);
            b.p_positions.push_back(k);
        }
    }
    b.p_rows = (int64_t)rows.size();
    *n_rows = b.p_rows;

    hipStream_t st = e->stream;
    ENC_ALLOC(b.d_seq, (size_t)total_bases + 16);
    ENC_ALLOC(b.d_segs, segs.size() * sizeof(PSeg) + 16);
    ENC_ALLOC(b.d_pbase, (size_t)L * PROW * sizeof(int));
    ENC_ALLOC(b.d_pins, (size_t)(total_ins_rows + 1) * PROW * sizeof(int));
    ENC_ALLOC(b.d_prow0, (size_t)(L + 1) * sizeof(int));
    ENC_ALLOC(b.d_prows, rows.size() * sizeof(PRow) + 16);
    ENC_ALLOC(b.d_ppix, rows.size() * 10 + 16);
    if (total_bases > 0) ENC_HIP(hipMemcpyAsync(b.d_seq.p, p->seq, (size_t)total_bases, hipMemcpyHostToDevice, st));
    if (!segs.empty()) ENC_HIP(hipMemcpyAsync(b.d_segs.p, segs.data(), segs.size() * sizeof(PSeg), hipMemcpyHostToDevice, st));
    ENC_HIP(hipMemcpyAsync(b.d_prow0.p, ins_row0.data(), (size_t)(L + 1) * sizeof(int), hipMemcpyHostToDevice, st));
    ENC_HIP(hipMemcpyAsync(b.d_prows.p, rows.data(), rows.size() * sizeof(PRow), hipMemcpyHostToDevice, st));
    ENC_HIP(hipMemsetAsync(b.d_pbase.p, 0, (size_t)L * PROW * sizeof(int), st));
    ENC_HIP(hipMemsetAsync(b.d_pins.p, 0, (size_t)(total_ins_rows + 1) * PROW * sizeof(int), st));
    if (!segs.empty())
        hipLaunchKernelGGL(polish_count_kernel, dim3(((int)segs.size() + 255) / 256), dim3(256), 0, st,
                           static_cast<const PSeg*>(b.d_segs.p), (int)segs.size(), static_cast<const char*>(b.d_seq.p),
                           static_cast<int*>(b.d_pbase.p), static_cast<int*>(b.d_pins.p));
    hipLaunchKernelGGL(polish_pixels_kernel, dim3(((int)rows.size() + 255) / 256), dim3(256), 0, st,
                       static_cast<const PRow*>(b.d_prows.p), (int)rows.size(), static_cast<const int*>(b.d_pbase.p),
                       static_cast<const int*>(b.d_pins.p), static_cast<const int*>(b.d_prow0.p),
                       static_cast<uint8_t*>(b.d_ppix.p));
    ENC_HIP(hipGetLastError());
    ENC_HIP(hipStreamSynchronize(st));
    return PA_OK;
}

int pa_polish_encoder_get_results(pa_encoder* e, uint8_t* image, int64_t* positions) {
    if (!e || !e->polish) return pa::set_error(PA_ERR_INVALID, "null encoder or no polish summary");
    ENC_HIP(hipSetDevice(e->device));
    pa_polish_batch& b = *e->polish;
    if (positions) std::copy(b.p_positions.begin(), b.p_positions.end(), positions);
    if (image && b.p_rows > 0) {
        ENC_HIP(hipMemcpyAsync(image, b.d_ppix.p, (size_t)b.p_rows * 10, hipMemcpyDeviceToHost, e->stream));
        ENC_HIP(hipStreamSynchronize(e->stream));
    }
    return PA_OK;
}
}  // extern "C"
