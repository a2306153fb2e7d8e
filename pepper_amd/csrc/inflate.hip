// BGZF blocks inflated on the device: one wavefront per block.
//
// What it stands in for: the block-by-block inflate under every BAM read of image generation (the reference reaches it
// through htslib's bgzf_read_block under sam_itr_next, /root/reference/pepper_variant/modules/src/dataio/bam_handler.cpp:
// 341-372; this repository's host form is Bgzf::read_block in bamio.cpp).  After the packed read form of round 4 that host
// inflate is 89 % of an image-generation worker's time, and a BGZF file is thousands of independent
// <= 64 KiB DEFLATE streams -- the unit the chip parallelises over.
//
// DEFLATE is RFC 1951 (stored, fixed and dynamic blocks; the length / distance base-and-extra-bit tables below are the
// RFC's, written as their closed forms).  A symbol is a chain of dependent steps (peek bits -> table -> consume), so a
// wavefront decodes its block as UNIFORM code: bit buffer, table index, positions all live in scalar registers, the
// 64 lanes serve as
//   - the input window: lane l of `cur` holds compressed word 64 c + l, consumed with v_readlane (one global load per
//     256 bytes of input, the next window already in flight),
//   - the literal staging register: v_writelane puts literal k into lane k, one 64-byte store per 64 literals,
//   - the match copier: a match of up to 258 bytes is one or a few byte-per-lane load/store pairs (overlapping matches,
//     distance < length, read source byte (i mod distance): every source byte exists before the match starts),
//   - the table builders (counting, the stable order of symbols by code length through ballots, the primary table filled
//     one entry per lane).
// Huffman tables in LDS: a primary table of 2^10 (literal/length) and 2^8 (distance) 16-bit entries (symbol << 4 | length),
// longer codes through the canonical count/symbol arrays bit by bit (RFC 1951 3.2.2's numbering; rare by construction:
// a code longer than 10 bits has probability < 2^-10).  3.9 KB of LDS per wavefront, so the 32 wavefront slots of a CU
// all hold a block.
//
// Checks: over-subscribed code sets, codes without a symbol, distances beyond the produced output, output beyond the
// block's ISIZE, a stored block's LEN/NLEN complement, input consumed beyond the block -> a non-zero status word per
// block (the host call fails with the first one).  Like the host reader the block CRC32 is not verified.
#include "../../include/pepper_amd.h"
#include "../../include/pepper_amd_io_device.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "common.h"
#include "kernels.h"

namespace pa {

namespace {

constexpr int LIT_BITS = 10, DIST_BITS = 8, CL_BITS = 7;
constexpr int MAX_LIT = 288, MAX_DIST = 32;

enum InflateStatus : int32_t {
    INF_OK = 0, INF_BAD_BLOCK_TYPE = 1, INF_STORED_LEN = 2, INF_OVERSUBSCRIBED = 3, INF_NO_END_CODE = 4, INF_BAD_CODE = 5,
    INF_BAD_REPEAT = 6, INF_DISTANCE = 7, INF_OUTPUT = 8, INF_LENGTH = 9, INF_INPUT = 10, INF_BAD_COUNTS = 11
};

struct Tables {
    uint16_t lit_table[1 << LIT_BITS];
    uint16_t dist_table[1 << DIST_BITS];
    uint16_t cl_table[1 << CL_BITS];
    uint16_t lit_sym[MAX_LIT];
    uint16_t dist_sym[MAX_DIST];
    uint16_t cl_sym[20];
    int lit_count[16], dist_count[16], cl_count[16];
    uint8_t lens[MAX_LIT + MAX_DIST + 16];        // literal/length lengths, then the distance lengths
    uint8_t cl_lens[20];
};

PA_DEV int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// the compressed bytes of one block as a stream of bits
struct Bits {
    const uint32_t* words;      // the word holding the block's first byte
    int n_words;                // words that hold bytes of the block
    int chunk;                  // `cur` holds words [64 chunk, 64 chunk + 64)
    int widx;                   // the next word to take
    uint32_t cur, nxt;          // per lane
    uint64_t bb;                // bits not yet consumed, the next one lowest
    int bc;

    PA_DEV uint32_t fetch(int c) const {
        const int i = c * 64 + (int)threadIdx.x;
        return i < n_words ? words[i] : 0u;
    }
    PA_DEV void seek(int byte_off) {            // position at a byte of the block (relative to `words`)
        widx = byte_off >> 2;
        chunk = widx >> 6;
        cur = fetch(chunk);
        nxt = fetch(chunk + 1);
        bb = 0;
        bc = 0;
        refill();
        const int skip = (byte_off & 3) * 8;
        bb >>= skip;
        bc -= skip;
        refill();
    }
    PA_DEV void refill() {                      // at least 33 bits afterwards (zeros beyond the end of the block)
        while (bc <= 32) {
            if ((widx >> 6) != chunk) {
                cur = nxt;
                ++chunk;
                nxt = fetch(chunk + 1);
            }
            const uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)cur, widx & 63);
            bb |= (uint64_t)w << bc;
            bc += 32;
            ++widx;
        }
    }
    PA_DEV uint32_t peek() const { return (uint32_t)bb; }
    PA_DEV void drop(int n) { bb >>= n; bc -= n; }
    PA_DEV uint32_t take(int n) {               // n <= 16
        const uint32_t v = (uint32_t)bb & ((1u << n) - 1u);
        drop(n);
        return v;
    }
    PA_DEV long long consumed_bits() const { return (long long)widx * 32 - bc; }
};

// Canonical code of `n` symbols from their lengths (RFC 1951 3.2.2): count[], the symbols in (length, symbol) order and
// the primary table of 2^tbits entries.  Returns false for an over-subscribed set.  All lanes call it.
PA_DEV bool build_table(const uint8_t* lens, int n, int* count, uint16_t* syms, uint16_t* table, int tbits) {
    const int lane = threadIdx.x;
    if (lane < 16) count[lane] = 0;
    __syncthreads();
    for (int s = lane; s < n; s += 64) atomicAdd(&count[lens[s]], 1);
    __syncthreads();
    int offs[16];
    int left = 1, total = 0;
    bool ok = true;
    offs[0] = 0;
#pragma unroll
    for (int len = 1; len < 16; ++len) {
        const int c = uni(count[len]);
        offs[len] = total;
        total += c;
        left = (left << 1) - c;
        if (left < 0) ok = false;
    }
    __syncthreads();
    if (lane == 0) count[0] = 0;
    if (!ok) return false;
    for (int base = 0; base < n; base += 64) {
        const int s = base + lane;
        const int l = s < n ? lens[s] : 0;
#pragma unroll
        for (int len = 1; len < 16; ++len) {
            const unsigned long long mask = __ballot(l == len);
            if (l == len) syms[offs[len] + __popcll(mask & ((1ull << lane) - 1ull))] = (uint16_t)s;
            offs[len] += __popcll(mask);
        }
    }
    __syncthreads();
    for (int t = lane; t < (1 << tbits); t += 64) {
        int code = 0, first = 0, index = 0;
        uint16_t e = 0;
        for (int len = 1; len <= tbits; ++len) {
            code |= (t >> (len - 1)) & 1;
            const int c = count[len];
            if (code - c < first) {
                e = (uint16_t)((syms[index + (code - first)] << 4) | len);
                break;
            }
            index += c;
            first = (first + c) << 1;
            code <<= 1;
        }
        table[t] = e;
    }
    __syncthreads();
    return true;
}

// One symbol: the primary table, or bit by bit for a longer code.  -1: the bits are no code of the set.
PA_DEV int decode(Bits& in, const uint16_t* table, int tbits, const int* count, const uint16_t* syms) {
    const uint32_t bits = in.peek();
    const int e = uni(table[bits & ((1u << tbits) - 1u)]);
    if (e) {
        in.drop(e & 15);
        return e >> 4;
    }
    int code = 0, first = 0, index = 0;
    for (int len = 1; len < 16; ++len) {
        code |= (bits >> (len - 1)) & 1;
        const int c = uni(count[len]);
        if (code - c < first) {
            in.drop(len);
            return uni(syms[index + (code - first)]);
        }
        index += c;
        first = (first + c) << 1;
        code <<= 1;
    }
    return -1;
}

__global__ __launch_bounds__(64) void bgzf_inflate_kernel(const uint8_t* __restrict__ comp, const int64_t* __restrict__ comp_off,
                                                         const int32_t* __restrict__ comp_len, const int64_t* __restrict__ out_off,
                                                         const int32_t* __restrict__ out_len, uint8_t* out_base,
                                                         int32_t* __restrict__ status) {
    __shared__ Tables T;
    const int lane = threadIdx.x;
    const int blk = blockIdx.x;
    const int64_t coff = comp_off[blk];
    const int clen = comp_len[blk], olen = out_len[blk];
    uint8_t* out = out_base + out_off[blk];
    const int mis = (int)(coff & 3);
    Bits in;
    in.words = reinterpret_cast<const uint32_t*>(comp + (coff - mis));
    in.n_words = (mis + clen + 3) >> 2;
    in.seek(mis);
    const uint8_t* in_bytes = comp + (coff - mis);

    int pos = 0;                 // bytes written or staged
    int staged = 0;              // literals in `stage` (lane k: the byte for out[pos - staged + k])
    int stage = 0;
    int err = INF_OK;
    auto flush = [&]() {
        if (staged) {
            if (lane < staged) out[pos - staged + lane] = (uint8_t)stage;
            staged = 0;
        }
    };

    bool last = olen == 0 && clen == 0;        // nothing at all: an empty block without a stream
    while (!last && !err) {
        in.refill();
        last = in.take(1) != 0;
        const int type = (int)in.take(2);
        if (type == 0) {
            // stored: to the next byte, LEN, ~LEN, the bytes
            in.drop(in.bc & 7);
            in.refill();
            const uint32_t len = in.take(16), nlen = in.take(16);
            if ((len ^ nlen) != 0xffffu) { err = INF_STORED_LEN; break; }
            flush();
            const int from = (int)(in.consumed_bits() >> 3);
            if (from + (int)len > mis + clen) { err = INF_INPUT; break; }
            if (pos + (int)len > olen) { err = INF_OUTPUT; break; }
            for (int k = lane; k < (int)len; k += 64) out[pos + k] = in_bytes[from + k];
            pos += (int)len;
            in.seek(from + (int)len);
            continue;
        }
        if (type == 3) { err = INF_BAD_BLOCK_TYPE; break; }
        int hlit = 288, hdist = 32;
        if (type == 1) {
            for (int s = lane; s < 288; s += 64) T.lens[s] = s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8;
            if (lane < 32) T.lens[288 + lane] = 5;
        } else {
            in.refill();
            hlit = (int)in.take(5) + 257;
            hdist = (int)in.take(5) + 1;
            const int hclen = (int)in.take(4) + 4;
            if (hlit > 286 || hdist > 30) { err = INF_BAD_COUNTS; break; }
            if (lane < 19) T.cl_lens[lane] = 0;
            __syncthreads();
            for (int k = 0; k < hclen; ++k) {
                in.refill();
                const int v = (int)in.take(3);
                // the order of the code length code lengths (RFC 1951 3.2.7), 5 bits each
                const unsigned long long lo = 16ull | 17ull << 5 | 18ull << 10 | 0ull << 15 | 8ull << 20 | 7ull << 25 | 9ull << 30 |
                                              6ull << 35 | 10ull << 40 | 5ull << 45 | 11ull << 50 | 4ull << 55;
                const unsigned long long hi = 12ull | 3ull << 5 | 13ull << 10 | 2ull << 15 | 14ull << 20 | 1ull << 25 | 15ull << 30;
                const int idx = k < 12 ? (int)((lo >> (5 * k)) & 31) : (int)((hi >> (5 * (k - 12))) & 31);
                if (lane == 0) T.cl_lens[idx] = (uint8_t)v;
            }
            __syncthreads();
            if (!build_table(T.cl_lens, 19, T.cl_count, T.cl_sym, T.cl_table, CL_BITS)) { err = INF_OVERSUBSCRIBED; break; }
            const int total = hlit + hdist;
            int i = 0, prev = 0;
            while (i < total) {
                in.refill();
                const int sym = decode(in, T.cl_table, CL_BITS, T.cl_count, T.cl_sym);
                if (sym < 0) { err = INF_BAD_CODE; break; }
                if (sym < 16) {
                    if (lane == 0) T.lens[i] = (uint8_t)sym;
                    prev = sym;
                    ++i;
                    continue;
                }
                int rep, val = 0;
                if (sym == 16) {
                    if (i == 0) { err = INF_BAD_REPEAT; break; }
                    val = prev;
                    rep = 3 + (int)in.take(2);
                } else if (sym == 17) {
                    rep = 3 + (int)in.take(3);
                    prev = 0;
                } else {
                    rep = 11 + (int)in.take(7);
                    prev = 0;
                }
                if (i + rep > total) { err = INF_BAD_REPEAT; break; }
                for (int k = lane; k < rep; k += 64) T.lens[i + k] = (uint8_t)val;
                i += rep;
            }
            if (err) break;
            __syncthreads();
            if (uni(T.lens[256]) == 0) { err = INF_NO_END_CODE; break; }
            // the distance lengths follow the literal/length ones directly: move them to their own place
            uint8_t dl = 0;
            if (lane < hdist) dl = T.lens[hlit + lane];
            __syncthreads();
            if (lane < 32) T.lens[288 + lane] = lane < hdist ? dl : 0;
            for (int s = hlit + lane; s < 288; s += 64) T.lens[s] = 0;
        }
        __syncthreads();
        if (!build_table(T.lens, 288, T.lit_count, T.lit_sym, T.lit_table, LIT_BITS) ||
            !build_table(T.lens + 288, 32, T.dist_count, T.dist_sym, T.dist_table, DIST_BITS)) {
            err = INF_OVERSUBSCRIBED;
            break;
        }
        (void)hlit;
        // the symbols of the block
        for (;;) {
            in.refill();
            int sym = decode(in, T.lit_table, LIT_BITS, T.lit_count, T.lit_sym);
            if (sym < 256) {
                if (sym < 0) { err = INF_BAD_CODE; break; }
                if (pos >= olen) { err = INF_OUTPUT; break; }
                asm("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(stage) : "s"(sym), "s"(staged) : "m0");
                ++staged;
                ++pos;
                if (staged == 64) {
                    out[pos - 64 + lane] = (uint8_t)stage;
                    staged = 0;
                }
                continue;
            }
            if (sym == 256) break;
            if (sym > 285) { err = INF_LENGTH; break; }
            sym -= 257;
            int len;
            if (sym < 8) len = sym + 3;
            else if (sym == 28) len = 258;
            else {
                const int eb = (sym - 4) >> 2;
                len = ((4 + (sym & 3)) << eb) + 3 + (int)in.take(eb);
            }
            in.refill();
            const int dsym = decode(in, T.dist_table, DIST_BITS, T.dist_count, T.dist_sym);
            if (dsym < 0 || dsym > 29) { err = INF_BAD_CODE; break; }
            int dist;
            if (dsym < 4) dist = dsym + 1;
            else {
                const int eb = (dsym >> 1) - 1;
                dist = ((2 + (dsym & 1)) << eb) + 1 + (int)in.take(eb);
            }
            if (dist > pos) { err = INF_DISTANCE; break; }
            if (pos + len > olen) { err = INF_OUTPUT; break; }
            flush();
            const uint8_t* src = out + pos - dist;
            if (dist >= len) {
                for (int k = lane; k < len; k += 64) out[pos + k] = src[k];
            } else {
                for (int k = lane; k < len; k += 64) out[pos + k] = src[k % dist];
            }
            pos += len;
        }
    }
    flush();
    if (!err && pos != olen) err = INF_OUTPUT;
    if (!err && in.consumed_bits() > (long long)(mis + clen) * 8) err = INF_INPUT;
    if (lane == 0) status[blk] = err;
}

}  // namespace

void launch_bgzf_inflate(hipStream_t stream, const uint8_t* comp, const int64_t* comp_off, const int32_t* comp_len,
                         const int64_t* out_off, const int32_t* out_len, uint8_t* out, int32_t* status, int n_blocks) {
    if (n_blocks <= 0) return;
    hipLaunchKernelGGL(bgzf_inflate_kernel, dim3(n_blocks), dim3(64), 0, stream, comp, comp_off, comp_len, out_off, out_len, out,
                       status);
}

const char* inflate_status_text(int32_t s) {
    switch (s) {
        case INF_OK: return "ok";
        case INF_BAD_BLOCK_TYPE: return "reserved DEFLATE block type";
        case INF_STORED_LEN: return "stored block: LEN and NLEN do not match";
        case INF_OVERSUBSCRIBED: return "over-subscribed Huffman code";
        case INF_NO_END_CODE: return "no end-of-block code";
        case INF_BAD_CODE: return "bits that are no code of the block's set";
        case INF_BAD_REPEAT: return "code length repeat without a previous length or beyond the table";
        case INF_DISTANCE: return "match distance beyond the start of the block";
        case INF_OUTPUT: return "output does not have the block's ISIZE";
        case INF_LENGTH: return "invalid length symbol";
        case INF_INPUT: return "stream runs beyond the block's compressed bytes";
        case INF_BAD_COUNTS: return "HLIT or HDIST out of range";
    }
    return "?";
}

}  // namespace pa

// ---------------------------------------------------------------------------------------------------------------------------
// C ABI (include/pepper_amd_io_device.h): a handle with its own stream and buffers -- the test and bench entry; image
// generation inflates into the encoder's arena through pa_encoder_inflate_bgzf (encoder.hip) with the same kernel.
struct pa_inflater {
    int device = 0;
    hipStream_t stream = nullptr;
    void* d_comp = nullptr; size_t comp_cap = 0;
    void* d_out = nullptr; size_t out_cap = 0;
    void* d_table = nullptr; size_t table_cap = 0;
    hipEvent_t ev[2] = {nullptr, nullptr};
    double kernel_ms = 0.0;
    int64_t n_blocks = 0, out_bytes = 0;
};

namespace {
bool grow(void** p, size_t* cap, size_t need) {
    if (need <= *cap) return true;
    if (*p) (void)hipFree(*p);
    *p = nullptr;
    *cap = 0;
    const size_t want = need + need / 4 + 4096;
    if (hipMalloc(p, want) != hipSuccess) return false;
    *cap = want;
    return true;
}
}  // namespace

#define INF_HIP(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) return pa::set_error(PA_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

extern "C" {

int pa_inflater_create(int32_t device, pa_inflater** out) {
    if (!out) return pa::set_error(PA_ERR_INVALID, "null argument");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return pa::set_error(PA_ERR_NO_DEVICE, "no HIP device visible: the device inflate has no CPU fallback");
    if (device < 0 || device >= count) return pa::set_error(PA_ERR_INVALID, "device ordinal out of range");
    INF_HIP(hipSetDevice(device));
    auto* h = new pa_inflater();
    h->device = device;
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess || hipEventCreate(&h->ev[0]) != hipSuccess ||
        hipEventCreate(&h->ev[1]) != hipSuccess) {
        delete h;
        return pa::set_error(PA_ERR_HIP, "stream / event creation failed");
    }
    *out = h;
    return PA_OK;
}

void pa_inflater_destroy(pa_inflater* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->d_comp) (void)hipFree(h->d_comp);
    if (h->d_out) (void)hipFree(h->d_out);
    if (h->d_table) (void)hipFree(h->d_table);
    for (auto& e : h->ev) if (e) (void)hipEventDestroy(e);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

int pa_inflater_inflate(pa_inflater* h, const uint8_t* comp, int64_t comp_bytes, int32_t n_blocks, const int64_t* comp_off,
                        const int32_t* comp_len, const int64_t* out_off, const int32_t* out_len, uint8_t* out, int64_t out_bytes,
                        int32_t repeats) {
    if (!h || (n_blocks > 0 && (!comp || !comp_off || !comp_len || !out_off || !out_len)) || n_blocks < 0 || comp_bytes < 0 ||
        out_bytes < 0 || (out_bytes > 0 && !out))
        return pa::set_error(PA_ERR_INVALID, "null or negative argument");
    for (int32_t b = 0; b < n_blocks; ++b) {
        if (comp_off[b] < 0 || comp_len[b] < 0 || comp_off[b] + comp_len[b] > comp_bytes || out_off[b] < 0 || out_len[b] < 0 ||
            out_off[b] + out_len[b] > out_bytes)
            return pa::set_error(PA_ERR_INVALID, "block " + std::to_string(b) + " lies outside the buffers");
    }
    INF_HIP(hipSetDevice(h->device));
    h->kernel_ms = 0.0;
    h->n_blocks = n_blocks;
    h->out_bytes = out_bytes;
    if (n_blocks == 0) return PA_OK;
    const size_t nb = (size_t)n_blocks;
    const size_t table_bytes = nb * (8 + 4 + 8 + 4 + 4);
    if (!grow(&h->d_comp, &h->comp_cap, (size_t)comp_bytes + 16) || !grow(&h->d_out, &h->out_cap, (size_t)out_bytes + 16) ||
        !grow(&h->d_table, &h->table_cap, table_bytes))
        return pa::set_error(PA_ERR_HIP, "hipMalloc failed in the inflate workspace");
    auto* d_coff = static_cast<int64_t*>(h->d_table);
    auto* d_ooff = d_coff + nb;
    auto* d_clen = reinterpret_cast<int32_t*>(d_ooff + nb);
    auto* d_olen = d_clen + nb;
    auto* d_status = d_olen + nb;
    INF_HIP(hipMemcpyAsync(h->d_comp, comp, (size_t)comp_bytes, hipMemcpyHostToDevice, h->stream));
    INF_HIP(hipMemcpyAsync(d_coff, comp_off, nb * 8, hipMemcpyHostToDevice, h->stream));
    INF_HIP(hipMemcpyAsync(d_ooff, out_off, nb * 8, hipMemcpyHostToDevice, h->stream));
    INF_HIP(hipMemcpyAsync(d_clen, comp_len, nb * 4, hipMemcpyHostToDevice, h->stream));
    INF_HIP(hipMemcpyAsync(d_olen, out_len, nb * 4, hipMemcpyHostToDevice, h->stream));
    const int reps = std::max(1, (int)repeats);
    INF_HIP(hipEventRecord(h->ev[0], h->stream));
    for (int r = 0; r < reps; ++r)
        pa::launch_bgzf_inflate(h->stream, static_cast<const uint8_t*>(h->d_comp), d_coff, d_clen, d_ooff, d_olen,
                                static_cast<uint8_t*>(h->d_out), d_status, n_blocks);
    INF_HIP(hipGetLastError());
    INF_HIP(hipEventRecord(h->ev[1], h->stream));
    std::vector<int32_t> status(nb);
    INF_HIP(hipMemcpyAsync(status.data(), d_status, nb * 4, hipMemcpyDeviceToHost, h->stream));
    if (out_bytes) INF_HIP(hipMemcpyAsync(out, h->d_out, (size_t)out_bytes, hipMemcpyDeviceToHost, h->stream));
    INF_HIP(hipStreamSynchronize(h->stream));
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, h->ev[0], h->ev[1]) == hipSuccess) h->kernel_ms = (double)ms / reps;
    for (int32_t b = 0; b < n_blocks; ++b)
        if (status[(size_t)b] != 0)
            return pa::set_error(PA_ERR_INVALID, "BGZF block " + std::to_string(b) + ": " + pa::inflate_status_text(status[(size_t)b]));
    return PA_OK;
}

int pa_inflater_last_kernel_ms(pa_inflater* h, double* ms) {
    if (!h || !ms) return pa::set_error(PA_ERR_INVALID, "null argument");
    *ms = h->kernel_ms;
    return PA_OK;
}

}  // extern "C"
