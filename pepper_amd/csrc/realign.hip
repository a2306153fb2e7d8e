// Read re-aligner (include/pepper_amd_realign.h): the polish image generator's local re-alignment of every read
// against the reference suffix that starts at its mapped position, as HIP kernels -- one wavefront per read.
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
//                    Mapping: lane l owns a strip of consecutive read rows (state in LDS, one dword per row:
//                    H | E | base code), columns are visited in a skewed pipeline (lane l works on column t - l at
//                    step t) and the strip's bottom cell, both gap chains and the running column maximum are handed to
//                    lane l + 1 through the cross-lane network.  Lane 63 sees each column's complete maximum and runs
//                    the sequential part (first column that raises the maximum, overflow, early stop of the reverse
//                    pass at the forward score).  Integer DP, no matrix cores: ~25 VALU ops + one LDS read / write
//                    per cell.
//   band_kernel      banded DP between begin and end cell with the library's band slots, its zeroed slot to the right
//                    of the previous row and its tie rules (ssw.c:571-650), band doubled until the score is reached;
//                    the row's vertical-gap chain is a max-plus prefix scan across the wavefront; direction bits go to
//                    a workspace in HBM (1 byte per cell); lane 0 walks them back (ssw.c:653-703) and writes the
//                    final operations: '=' / 'X' runs from comparing base codes, I, D, soft clips
//                    (ssw_cpp.cpp:56-207).
// Host side of the entry points: base text -> codes, job table, workspace sizing between the two kernels.
#include "../../include/pepper_amd_realign.h"
#include "../../include/pepper_amd.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

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
    int64_t dir_off, ops_off, steps_off;
    int32_t ops_cap, n_ops;
};

__device__ __forceinline__ int bcast63(int v) { return __builtin_amdgcn_readlane(v, 63); }

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

__global__ __launch_bounds__(64) void sw_ends_kernel(Job* __restrict__ jobs, const int8_t* __restrict__ ref,
                                                     const int8_t* __restrict__ seq) {
    extern __shared__ uint32_t he[];
    Job& J = jobs[blockIdx.x];
    if (J.state != ST_NEW) return;
    const int8_t* rf = ref + J.ref_off;
    const int8_t* rd = seq + J.seq_off;
    const int n = J.n, m = J.m;
    PassOut f = score_pass(he, rf, 0, 1, n, rd, 0, 1, m, 16, -1);
    int wide = 0;
    if (f.overflow) {
        f = score_pass(he, rf, 0, 1, n, rd, 0, 1, m, 8, -1);
        wide = 1;
    }
    int ref_begin = -1, read_begin = -1;
    if (f.score > 0 && f.ref >= 0) {
        const PassOut r = score_pass(he, rf, f.ref, -1, f.ref + 1, rd, f.read, -1, f.read + 1, wide ? 8 : 16, f.score);
        ref_begin = r.ref;
        read_begin = f.read - r.read;
    }
    if (threadIdx.x == 0) {
        J.score = f.score; J.wide = wide; J.ref_end = f.ref; J.read_end = f.read;
        J.ref_begin = ref_begin; J.read_begin = read_begin;
    }
}

__device__ __forceinline__ int wave_max(int v) {
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return v;
}

__global__ __launch_bounds__(64) void band_kernel(Job* __restrict__ jobs, const int8_t* __restrict__ ref,
                                                  const int8_t* __restrict__ seq, uint8_t* __restrict__ dirws,
                                                  uint8_t* __restrict__ stepws, uint32_t* __restrict__ opsws, int cap) {
    extern __shared__ int sm[];
    Job& J = jobs[blockIdx.x];
    if (J.state != ST_BAND) return;
    const int lane = threadIdx.x;
    const int n = J.ref_end - J.ref_begin + 1, m = J.read_end - J.read_begin + 1, score = J.score;
    const int8_t* rf = ref + J.ref_off + J.ref_begin;
    const int8_t* rd = seq + J.seq_off + J.read_begin;
    int* hb = sm;
    int* eb = sm + cap;
    int* hc = sm + 2 * cap;
    uint8_t* dir = dirws + J.dir_off;
    int bw = J.bw, stride = 0;
    for (;;) {
        const int width = 2 * bw + 3;
        stride = min(2 * bw + 1, n);
        const int slots = min(width, n + 2) + 1;
        if (stride > J.dir_width || slots > cap) {
            if (lane == 0) { J.bw = bw; J.state = ST_WIDER; }
            return;
        }
        for (int k = lane; k < slots; k += 64) { hb[k] = 0; eb[k] = 0; hc[k] = 0; }
        int best = 0;
        for (int i = 0; i < m; ++i) {
            const int x = max(i - bw, 0), xp = max(i - 1 - bw, 0), sh = x - xp;
            const int end = min(n - 1, i + bw), U = end - x + 1, edge = min(end + 1, width - 1);
            __syncthreads();
            if (lane == 0) { hb[0] = 0; eb[0] = 0; hb[edge] = 0; eb[edge] = 0; hc[0] = 0; }
            __syncthreads();
            const int qi = rd[i];
            int carry_a = NEG, carry_h = 0, carry_f = 0;
            uint8_t* drow = dir + (size_t)i * stride;
            for (int base = 0; base < U; base += 64) {
                const int u = 1 + base + lane;
                const bool valid = u <= U;
                int hbe = 0, ebe = 0, hbd = 0, rj = 4;
                if (valid) {
                    hbe = hb[u + sh];
                    ebe = eb[u + sh];
                    hbd = hb[u + sh - 1];
                    rj = rf[x + u - 1];
                }
                const int t1 = i == 0 ? -GO : hbe - GO, t2 = i == 0 ? -GE : ebe - GE;
                const int ecur = max(t1, t2), de = t1 > t2;
                const int diag = hbd + ((rj == qi && qi < 4) ? S_MATCH : -S_MIS);
                const int e1 = max(ecur, 0), g = max(e1, diag);
                // vertical-gap chain of the row: f(u) = max(-GE u, max_{v<u} (g(v) + GE v) - GO - GE (u - 1))
                int pm = valid ? g + u * GE : NEG;
                for (int o = 1; o < 64; o <<= 1) {
                    const int v = __shfl_up(pm, o, 64);
                    if (lane >= o) pm = max(pm, v);
                }
                int ex = __shfl_up(pm, 1, 64);
                ex = lane == 0 ? carry_a : max(ex, carry_a);
                const int f = max(-GE * u, ex - GO - (u - 1) * GE);
                const int f1 = max(f, 0), hcur = max(g, f1);
                int hl = __shfl_up(hcur, 1, 64), fl = __shfl_up(f, 1, 64);
                if (lane == 0) { hl = carry_h; fl = carry_f; }
                const int df = (hl - GO) > (fl - GE);
                const int gap = max(e1, f1);
                const int dh = gap <= diag ? 1 : (e1 > f1 ? (de ? 3 : 2) : (df ? 5 : 4));
                if (valid) {
                    eb[u] = ecur;
                    hc[u] = hcur;
                    drow[u - 1] = (uint8_t)(de | (df << 1) | (dh << 2));
                    best = max(best, hcur);
                }
                carry_a = max(carry_a, bcast63(pm));
                carry_h = bcast63(hcur);
                carry_f = bcast63(f);
            }
            __syncthreads();
            for (int u = 1 + lane; u <= U; u += 64) hb[u] = hc[u];
        }
        best = wave_max(best);
        if (best >= score) break;
        if (bw > n + m) {                       // the library would double for ever here
            if (lane == 0) J.state = ST_ERR;
            return;
        }
        bw *= 2;
    }
    __threadfence();
    __syncthreads();
    if (lane != 0) return;

    // trace back (ssw.c:653-703): state 2 = H, 0 = E (read gap open/extend), 1 = F
    uint8_t* steps = stepws + J.steps_off;
    const int step_cap = m + n + 2;
    int i = m - 1, j = n - 1, state = 2, ns = 0;
    while (i > 0) {
        const int off = j - max(i - bw, 0);
        if (off < 0 || off >= stride || ns >= step_cap) { J.state = ST_ERR; return; }
        const int d = dir[(size_t)i * stride + off];
        const int code = state == 2 ? (d >> 2) : (state == 0 ? ((d & 1) ? 3 : 2) : ((d & 2) ? 5 : 4));
        int op;
        switch (code) {
            case 1: --i; --j; state = 2; op = 0; break;
            case 2: --i; state = 0; op = OP_I; break;
            case 3: --i; state = 2; op = OP_I; break;
            case 4: --j; state = 1; op = OP_D; break;
            case 5: --j; state = 2; op = OP_D; break;
            default: J.state = ST_ERR; return;
        }
        steps[ns++] = (uint8_t)op;
    }
    // operations in alignment order: soft clip, the first cell, the steps backwards, soft clip; aligned pairs are
    // classified by comparing base codes from the begin cell on (ssw_cpp.cpp:126-207)
    uint32_t* ops = opsws + J.ops_off;
    const int8_t* rfull = ref + J.ref_off;
    const int8_t* rdfull = seq + J.seq_off;
    int no = 0, cur = -1, len = 0, rp = J.ref_begin, qp = J.read_begin;
    const int ocap = J.ops_cap;
    auto put = [&](int op, int l) { if (no < ocap) ops[no] = ((uint32_t)l << 4) | (uint32_t)op; ++no; };
    auto emit = [&](int op) {
        if (op == cur) { ++len; return; }
        if (len) put(cur, len);
        cur = op; len = 1;
    };
    if (J.read_begin > 0) put(OP_S, J.read_begin);
    for (int s = ns; s >= 0; --s) {
        const int op = s == ns ? 0 : steps[s];
        if (op == 0) {
            emit(rfull[rp] == rdfull[qp] ? OP_EQ : OP_X);
            ++rp; ++qp;
        } else if (op == OP_I) {
            emit(OP_I); ++qp;
        } else {
            emit(OP_D); ++rp;
        }
    }
    if (len) put(cur, len);
    if (J.m - J.read_end - 1 > 0) put(OP_S, J.m - J.read_end - 1);
    J.n_ops = no;
    J.bw = bw;
    J.state = no <= ocap ? ST_DONE : ST_ERR;
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

inline int8_t base_code(char c) {                 // ssw_cpp.cpp:10-19
    switch (c) {
        case 'A': case 'a': case 'U': case 'u': return 0;   // the table sends U to 0 as well
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': return 3;
        default: return 4;
    }
}

}  // namespace

struct pa_realigner {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    DBuf d_ref, d_seq, d_jobs, d_dir, d_steps, d_ops;
    std::vector<Job> jobs;
    std::vector<uint32_t> ops;
    int64_t total_ops = 0;
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
    *out = r;
    return PA_OK;
}

void pa_realigner_destroy(pa_realigner* r) {
    if (!r) return;
    (void)hipSetDevice(r->device);
    if (r->stream) (void)hipStreamSynchronize(r->stream);
    if (r->own_stream && r->stream) (void)hipStreamDestroy(r->stream);
    delete r;
}

int pa_realigner_align(pa_realigner* r, const char* reference, int64_t reference_len, int64_t region_start,
                       int32_t n_reads, const int64_t* read_pos, const int64_t* seq_offset, const char* seq,
                       int32_t* status, int32_t* sw_score, int64_t* new_pos, int64_t* new_pos_end,
                       int32_t* query_begin, int32_t* query_end, int64_t* n_cigar_ops) {
    if (!r || !n_cigar_ops || n_reads < 0 || reference_len < 0 || (n_reads > 0 && (!reference || !read_pos || !seq_offset ||
        !seq || !status || !sw_score || !new_pos || !new_pos_end)))
        return pa::set_error(PA_ERR_INVALID, "null argument");
    if (reference_len > (int64_t)1 << 30) return pa::set_error(PA_ERR_INVALID, "reference window too long");
    RA_HIP(hipSetDevice(r->device));
    r->jobs.assign((size_t)n_reads, Job());
    r->total_ops = 0;
    *n_cigar_ops = 0;
    if (n_reads == 0) return PA_OK;

    const int64_t total_seq = seq_offset[n_reads];
    if (total_seq < 0 || total_seq > (int64_t)1 << 31) return pa::set_error(PA_ERR_INVALID, "bad seq_offset");
    int max_m = 1;
    bool any = false;
    for (int32_t k = 0; k < n_reads; ++k) {
        Job& J = r->jobs[(size_t)k];
        const int64_t m = seq_offset[k + 1] - seq_offset[k], off = read_pos[k] - region_start;
        if (m < 0) return pa::set_error(PA_ERR_INVALID, "seq_offset is not monotonic");
        J.seq_off = seq_offset[k];
        J.m = (int32_t)std::min<int64_t>(m, INT32_MAX);
        J.ref_begin = J.read_begin = -1;
        if (off < 0) { J.state = ST_DROPPED; continue; }
        if (off > reference_len)
            return pa::set_error(PA_ERR_INVALID, "read " + std::to_string(k) + " starts beyond the reference window");
        J.ref_off = (int32_t)off;
        J.n = (int32_t)(reference_len - off);
        if (m == 0 || J.n == 0) { J.state = ST_KEPT; continue; }
        if (m > MAX_READ)
            return pa::set_error(PA_ERR_INVALID, "read " + std::to_string(k) + " has " + std::to_string(m) +
                                 " bases: the re-aligner handles region-clipped reads up to " + std::to_string(MAX_READ));
        J.state = ST_NEW;
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

    // base text -> codes, upload
    std::vector<int8_t> codes((size_t)reference_len + (size_t)total_seq);
    for (int64_t k = 0; k < reference_len; ++k) codes[(size_t)k] = base_code(reference[k]);
    for (int64_t k = 0; k < total_seq; ++k) codes[(size_t)(reference_len + k)] = base_code(seq[k]);
    RA_ALLOC(r->d_ref, (size_t)reference_len + 64);
    RA_ALLOC(r->d_seq, (size_t)total_seq + 64);
    RA_ALLOC(r->d_jobs, sizeof(Job) * (size_t)n_reads);
    RA_HIP(hipMemcpyAsync(r->d_ref.p, codes.data(), (size_t)reference_len, hipMemcpyHostToDevice, r->stream));
    RA_HIP(hipMemcpyAsync(r->d_seq.p, codes.data() + reference_len, (size_t)total_seq, hipMemcpyHostToDevice, r->stream));
    RA_HIP(hipMemcpyAsync(r->d_jobs.p, r->jobs.data(), sizeof(Job) * (size_t)n_reads, hipMemcpyHostToDevice, r->stream));
    Job* dj = static_cast<Job*>(r->d_jobs.p);
    const int8_t* dref = static_cast<const int8_t*>(r->d_ref.p);
    const int8_t* dseq = static_cast<const int8_t*>(r->d_seq.p);
    {
        const int rows = ((max_m + 15) / 16) * 16, R = (rows + 63) / 64;
        hipLaunchKernelGGL(sw_ends_kernel, dim3(n_reads), dim3(64), (size_t)64 * R * 4, r->stream, dj, dref, dseq);
        RA_HIP(hipGetLastError());
    }
    RA_HIP(hipMemcpyAsync(r->jobs.data(), dj, sizeof(Job) * (size_t)n_reads, hipMemcpyDeviceToHost, r->stream));
    RA_HIP(hipStreamSynchronize(r->stream));

    // band stage: workspace layout, first with rows of at most 129 slots (band half width <= 64), then full rows
    int64_t ops_total = 0, steps_total = 0;
    for (Job& J : r->jobs) {
        if (J.state != ST_NEW) continue;
        if (J.score <= 1 || J.ref_begin < 0) { J.state = ST_KEPT; continue; }     // simple_aligner.cpp:85
        const int n2 = J.ref_end - J.ref_begin + 1, m2 = J.read_end - J.read_begin + 1;
        J.state = ST_BAND;
        J.bw = std::abs(n2 - m2) + 1;
        J.ops_off = ops_total;
        J.ops_cap = n2 + m2 + 4;
        ops_total += J.ops_cap;
        J.steps_off = steps_total;
        steps_total += n2 + m2 + 2;
    }
    RA_ALLOC(r->d_ops, sizeof(uint32_t) * (size_t)std::max<int64_t>(ops_total, 1));
    RA_ALLOC(r->d_steps, (size_t)std::max<int64_t>(steps_total, 1));
    for (int round = 0; round < 3; ++round) {
        int64_t dir_total = 0;
        int cap = 0, pending = 0;
        for (Job& J : r->jobs) {
            if (J.state == ST_WIDER) J.state = ST_BAND;
            if (J.state != ST_BAND) continue;
            const int n2 = J.ref_end - J.ref_begin + 1, m2 = J.read_end - J.read_begin + 1;
            J.dir_width = round == 0 ? std::min(n2, 129) : n2;
            if (round == 0 && std::min(2 * J.bw + 1, n2) > J.dir_width) J.dir_width = n2;   // first band already wider
            J.dir_off = dir_total;
            dir_total += (int64_t)m2 * J.dir_width;
            const int bw_cap = J.dir_width >= n2 ? INT32_MAX / 4 : (J.dir_width - 1) / 2;
            cap = std::max(cap, (int)std::min<int64_t>(2 * (int64_t)bw_cap + 3, n2 + 2) + 2);
            ++pending;
        }
        if (!pending) break;
        if ((size_t)cap * 12 > 150 * 1024) return pa::set_error(PA_ERR_INVALID, "alignment too long for the band stage");
        RA_ALLOC(r->d_dir, (size_t)std::max<int64_t>(dir_total, 1));
        RA_HIP(hipMemcpyAsync(dj, r->jobs.data(), sizeof(Job) * (size_t)n_reads, hipMemcpyHostToDevice, r->stream));
        if ((size_t)cap * 12 > 64 * 1024)
            RA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(band_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       cap * 12));
        hipLaunchKernelGGL(band_kernel, dim3(n_reads), dim3(64), (size_t)cap * 12, r->stream, dj, dref, dseq,
                           static_cast<uint8_t*>(r->d_dir.p), static_cast<uint8_t*>(r->d_steps.p),
                           static_cast<uint32_t*>(r->d_ops.p), cap);
        RA_HIP(hipGetLastError());
        RA_HIP(hipMemcpyAsync(r->jobs.data(), dj, sizeof(Job) * (size_t)n_reads, hipMemcpyDeviceToHost, r->stream));
        RA_HIP(hipStreamSynchronize(r->stream));
    }
    for (int32_t k = 0; k < n_reads; ++k) {
        const Job& J = r->jobs[(size_t)k];
        if (J.state == ST_ERR || J.state == ST_BAND || J.state == ST_WIDER)
            return pa::set_error(PA_ERR_INVALID, "read " + std::to_string(k) + ": the band stage did not reach the alignment score "
                                 "(the reference library aborts on such an alignment)");
    }
    r->ops.resize((size_t)ops_total);
    if (ops_total)
        RA_HIP(hipMemcpyAsync(r->ops.data(), r->d_ops.p, sizeof(uint32_t) * (size_t)ops_total, hipMemcpyDeviceToHost, r->stream));
    RA_HIP(hipStreamSynchronize(r->stream));
    for (const Job& J : r->jobs)
        if (J.state == ST_DONE) r->total_ops += J.n_ops;
    *n_cigar_ops = r->total_ops;
    finish_outputs();
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
