// Shared pieces of the two summary encoders (encoder.hip: variant, encoder_polish.hip: polish).
//
// Both walk CIGAR strings on the device the same way.  A wave takes 64 operations at a time, one per lane; a wave-wide
// prefix sum of the reference / read advances gives every operation its first reference row and its first read index
// (no serial walk); the sparse per-operation work (insert anchors, deletion anchors) is done one operation per lane, the
// dense per-base work one REFERENCE ROW per lane: lane l takes row span_lo + l + 64 k and finds the operation that owns
// it by a 6-step binary search over the 64 first-rows the wave has just written to its LDS scratch.  Consecutive lanes
// therefore touch consecutive rows (conflict-free LDS atomics on a [counter][row] tile) and consecutive read bytes
// (coalesced loads), whatever the lengths of the match runs -- a lane-per-operation walk idles on the longest run of the 64,
// a wave-per-operation walk (round 2) used 10-20 lanes of 64 on nanopore CIGARs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/pepper_amd.h"
#include "../../include/pepper_amd_encoder.h"
#include "kernels.h"

namespace pa_enc {

enum { OP_M = 0, OP_I = 1, OP_D = 2, OP_N = 3, OP_S = 4, OP_H = 5, OP_P = 6, OP_EQ = 7, OP_X = 8 };

// inclusive prefix sum over the 64 lanes of a wave on the DPP network (row_shr 1/2/4/8 inside each row of 16 lanes, then
// row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3); lanes without a source add the `old` operand, 0
__device__ __forceinline__ int wave_inclusive_sum(int x) {
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);
    return x;
}
__device__ __forceinline__ int wave_total(int inclusive) { return __builtin_amdgcn_readlane(inclusive, 63); }
__device__ __forceinline__ int wave_sum(int x) { return wave_total(wave_inclusive_sum(x)); }
// minimum / maximum over the wave on the same network (identity in the lanes without a source), the same in every lane
__device__ __forceinline__ int wave_min(int x) {
    const int id = 0x7fffffff;
    x = min(x, __builtin_amdgcn_update_dpp(id, x, 0x111, 0xf, 0xf, false));
    x = min(x, __builtin_amdgcn_update_dpp(id, x, 0x112, 0xf, 0xf, false));
    x = min(x, __builtin_amdgcn_update_dpp(id, x, 0x114, 0xf, 0xf, false));
    x = min(x, __builtin_amdgcn_update_dpp(id, x, 0x118, 0xf, 0xf, false));
    x = min(x, __builtin_amdgcn_update_dpp(id, x, 0x142, 0xa, 0xf, false));
    x = min(x, __builtin_amdgcn_update_dpp(id, x, 0x143, 0xc, 0xf, false));
    return __builtin_amdgcn_readlane(x, 63);
}
__device__ __forceinline__ int wave_max(int x) { return -wave_min(-x); }       // (callers keep |x| < 2^31 - 1)

// reference / read advance of one operation as populate_summary_matrix steps (region_summary.cpp:357-563): N and P advance
// the reference AND, through the missing break, the read (:556-561)
__device__ __forceinline__ int variant_ref_advance(int op, int len) {
    return (op == OP_M || op == OP_EQ || op == OP_X || op == OP_D || op == OP_N || op == OP_P) ? len : 0;
}
__device__ __forceinline__ int variant_read_advance(int op, int len) {
    return (op == OP_M || op == OP_EQ || op == OP_X || op == OP_I || op == OP_S || op == OP_N || op == OP_P) ? len : 0;
}
// the polish walk (summary_generator.cpp:47-121): D, N and P are gaps that advance only the reference
__device__ __forceinline__ int polish_ref_advance(int op, int len) {
    return (op == OP_M || op == OP_EQ || op == OP_X || op == OP_D || op == OP_N || op == OP_P) ? len : 0;
}
__device__ __forceinline__ int polish_read_advance(int op, int len) {
    return (op == OP_M || op == OP_EQ || op == OP_X || op == OP_I || op == OP_S) ? len : 0;
}

// last j in [0, 64) with first_row[j] <= row; first_row is non-decreasing (INT_MAX past the read's last operation) and
// first_row[0] <= row
__device__ __forceinline__ int owner_of_row(const int* first_row, int row) {
    int j = 0;
#pragma unroll
    for (int step = 32; step > 0; step >>= 1)
        if (first_row[j + step] <= row) j += step;
    return j;
}

struct DBuf {
    void* p = nullptr;
    size_t bytes = 0;
    bool ensure(size_t need) {
        if (need <= bytes) return true;
        static const bool trace = getenv("PA_TRACE_ALLOC") != nullptr;      // every (re)allocation on stderr: a hipFree waits for the device
        if (trace) fprintf(stderr, "[alloc] device buffer %zu -> %zu bytes\n", bytes, need + need / 4 + 256);
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
        const size_t grow = need + need / 4 + 256;
        if (hipMalloc(&p, grow) != hipSuccess) return false;
        bytes = grow;
        return true;
    }
    template <typename T> T* as() const { return static_cast<T*>(p); }
    ~DBuf() { if (p) (void)hipFree(p); }
};

// page-locked host buffer that only grows (results of a run: one asynchronous copy each, no page faults per run)
struct HBuf {
    void* p = nullptr;
    size_t bytes = 0;
    bool ensure(size_t need) {
        if (need <= bytes) return true;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        bytes = 0;
        const size_t grow = need + need / 2 + 4096;
        if (hipHostMalloc(&p, grow, hipHostMallocDefault) != hipSuccess) return false;
        bytes = grow;
        return true;
    }
    template <typename T> T* as() const { return static_cast<T*>(p); }
    ~HBuf() { if (p) (void)hipHostFree(p); }
};


// one read of a batch: where its bases, qualities and operations start in the concatenated device arrays
struct ReadRec {
    int64_t s0;        // first base in seq / qual
    int32_t c0;        // first operation in cigar_op / cigar_len
    int32_t ncig;
    int32_t slen;
    int32_t row0;      // type_read.pos - region_start
    int32_t region;
    int32_t flags;     // 1 reverse strand, 2 mapping quality > 0
};
constexpr int READ_REV = 1, READ_MAPQ_OK = 2;

}  // namespace pa_enc

#define ENC_HIP(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) return pa::set_error(PA_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)
#define ENC_ALLOC(buf, bytes_)                                                                          \
    do {                                                                                                \
        if (!(buf).ensure(bytes_)) return pa::set_error(PA_ERR_HIP, "hipMalloc failed in encoder workspace"); \
    } while (0)

struct pa_variant_batch;     // encoder.hip
struct pa_polish_batch;      // encoder_polish.hip

struct pa_encoder {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipEvent_t ev[12] = {};
    pa_variant_batch* variant = nullptr;
    pa_polish_batch* polish = nullptr;
    struct pa_realigner* realigner = nullptr;      // the polish image chain's re-aligner on this encoder's stream (made on first use)
};

// ---- the polish image chain (encoder_polish.hip: pa_polish_chain_*) crosses the three files; the pieces it calls --------------
namespace pa_enc {
// what unpack_clip_kernel (encoder.hip) left ON THE DEVICE for the (read, region) pairs of a packed batch: every pair's clipped
// read as the host-clipped form would have uploaded it (ReadRec, CIGAR arrays, bases as text).  The CIGAR arrays have room
// for extra_ops more operations behind the total_ops the pairs may use (the re-aligned CIGARs go there): total_bases +
// n_pairs x extra_ops_per_pair of them (extra_ops_per_pair < 0: none).
struct UnpackedReads {
    ReadRec* reads = nullptr;
    int32_t* cigar_op = nullptr;
    int32_t* cigar_len = nullptr;
    const char* seq = nullptr;
    int64_t n_pairs = 0, total_bases = 0, total_ops = 0, extra_ops = 0;      // slots the pairs were given (bounds of what they use)
    const int* h_live = nullptr;     // page-locked [n_regions + 2], valid once the stream has been waited for: reads with a base
                                     // inside each region | first inconsistent read + 1 | first unsupported read + 1
};
int unpack_packed_regions(pa_encoder* e, int32_t n_regions, const int64_t* region_start, const int64_t* region_end,
                          const uint8_t* arena, int64_t arena_bytes, const pa_packed_read* reads, int32_t n_reads,
                          const int32_t* pair_read, const int32_t* region_pairs, int32_t extra_ops_per_pair, UnpackedReads* out);
}  // namespace pa_enc

namespace pa_ra {
// The re-aligner over reads that are already on the device (realign.hip).  reads[k] (ReadRec of pair k: s0, slen, row0 = read
// position - region start, region, flags) against window `region`: codes [window_off[region], + window_len[region]) of the
// window text uploaded by this call.  Reads without READ_MAPQ_OK are not aligned (the summary skips them).  On return the
// re-aligner's stream (the one it was created on) has been waited for once or more; the results are on the device: jobs (the kernels' table: state, ref_begin, ops_off,
// n_ops per read) and the compacted operations (len << 4 | op).  d_cigar_op / d_cigar_len: the reads' clipped BAM alignments
// (ReadRec.c0 / ncig; null: none) -- only used to prove that a read's 8-bit score pass overflows, which is then not run.
struct DeviceResult {
    const void* jobs = nullptr;          // Job [n_reads] (realign.hip)
    const uint32_t* ops = nullptr;
    int64_t ops_written = 0;
    int32_t n_aligned = 0;
    int32_t n_proven = 0;                // reads whose 8-bit score pass was proven to overflow from their BAM alignment and not run
};
int align_device(pa_realigner* r, const char* window_text, int64_t window_bytes, const int64_t* window_off,
                 const int32_t* window_len, int32_t n_windows, const pa_enc::ReadRec* d_reads, int32_t n_reads, const char* d_seq,
                 int64_t seq_bytes, int32_t max_region_len, const int32_t* d_cigar_op, const int32_t* d_cigar_len, DeviceResult* out);
// per read k with a new alignment: reads[k].row0 += ref_begin, its CIGAR = the operations decoded into cigar_op / cigar_len at
// ops_base + ops_off ('=' and 'X' as MATCH when collapse_eqx), c0 / ncig pointing there
int apply_device(pa_realigner* r, pa_enc::ReadRec* d_reads, int32_t n_reads, int32_t* cigar_op,
                 int32_t* cigar_len, int64_t ops_base, int32_t collapse_eqx);
}  // namespace pa_ra

// the two halves free their own state (defined next to the structs)
void pa_variant_batch_free(pa_variant_batch*);
void pa_polish_batch_free(pa_polish_batch*);
