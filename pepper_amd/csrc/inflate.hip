// BGZF blocks inflated on the device: one wavefront per block.
//
// What it stands in for: the block-by-block inflate under every BAM read of image generation (the reference reaches it
// through htslib's bgzf_read_block under sam_itr_next, /root/reference/pepper_variant/modules/cpp/bam_handler.cpp:
// 341-372; this repository's host form is Bgzf::read_block in bamio.cpp).  After the packed read form of round 4 that host
// inflate is 89 % of an image-generation worker's time, and a BGZF file is thousands of independent
// <= 64 KiB DEFLATE streams -- the unit the chip parallelises over.
//
// DEFLATE is RFC 1951 (stored, fixed and dynamic blocks; the length / distance base-and-extra-bit tables below are the
// RFC's, written as their closed forms).  A symbol is a chain of dependent steps (peek bits -> table -> consume); decoded
// as uniform scalar code a member costs ~40 scalar instructions per symbol and the chip runs at the scalar units' issue
// rate (14 GB/s, the first version of this file).  What is here instead, per step of one wavefront:
//   1. 64 bit offsets at once: lane k looks up whatever would start at bit p + k of the stream -- literal/length code from
//      the primary table, for a length symbol also its extra bits, the distance code behind them and that code's extra
//      bits, all from the lane's own 64 bits (three words of a 512-byte window of the stream kept in LDS).  Most lanes
//      sit inside a symbol: their result is never looked at.
//   2. the symbols that do start in the window: follow the lengths from offset 0 (v_readlane per symbol, 6-7 per step).
//   3. the step's OUTPUT, one lane per byte: the symbols mark their first byte in a 64-byte LDS scratch, a prefix maximum
//      spreads the mark, ds_bpermute fetches the symbol's (value, distance); a match byte's source is memory in front of
//      the step or an earlier byte of the same step, followed back between lanes to a literal or a memory byte.  One
//      gather load and one store of up to 64 bytes per step.
//   4. the step is completed one step later: its loads are in flight while the next window is decoded (a step whose
//      sources reach into the pending bytes completes them first).
// Steps of more than 64 output bytes (long matches) and codes longer than the primary tables take a per-symbol path.
// Huffman tables in LDS: a primary table of 2^10 literal/length entries (16 bits: code length, extra-bit count, the
// literal or the length's base) and 2^8 distance entries (32 bits: code length, extra-bit count, base) -- no lane does
// arithmetic on symbol numbers; longer codes through the canonical count/symbol arrays bit by bit (RFC 1951 3.2.2's
// numbering; rare by construction: a code longer than 10 bits has probability < 2^-10).  Tables are built by all lanes:
// counts by LDS atomics, the stable order of symbols by code length through ballots, the primary table one entry per
// lane.  5.1 KB of LDS per wavefront.  A workgroup is one wavefront: hand-offs through LDS need program order only
// (wave_order), no barrier.
// What binds the kernel is the CU's ONE scalar unit, shared by its four SIMDs (measured before the diet: 185 scalar
// against 169 vector instructions per step -- the scalar unit needs 4 x 185 = 740 cycles for the four SIMDs' steps, a
// SIMD's vector unit 4 x 169 = 676 for its own): the symbol chain is written out in eight scalar instructions per symbol, every lane select hangs on ONE comparison (combining two lane
// masks is scalar work), and the register budget is six wavefronts per SIMD (79 VGPRs; at eight, spills cost more than
// the two wavefronts bring).
//
// Checks: over-subscribed code sets, codes without a symbol, distances beyond the produced output, output beyond the
// block's ISIZE, a stored block's LEN/NLEN complement, input consumed beyond the block -> a non-zero status word per
// block (the host call fails with the first one).  The member's CRC-32 is verified in the kernel's epilogue (wave_crc32).
#include "../../include/pepper_amd.h"
#include "../../include/pepper_amd_io_device.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "common.h"
#include "kernels.h"

namespace pa {

namespace {

constexpr int LIT_BITS = 10, DIST_BITS = 8, CL_BITS = 7;
constexpr int MAX_LIT = 288, MAX_DIST = 32;
constexpr int RING_WORDS = 128;                  // two chunks of 64 words of the compressed stream
constexpr int RING_MIRROR = 8;                   // ... and its first words once more behind them: a read of up to 5 words from any word
                                                 // of the ring is then five consecutive addresses (no wrap per word: two ds_read2 + one
                                                 // ds_read from ONE address instead of five masked ones)

enum InflateStatus : int32_t {
    INF_OK = 0, INF_BAD_BLOCK_TYPE = 1, INF_STORED_LEN = 2, INF_OVERSUBSCRIBED = 3, INF_NO_END_CODE = 4, INF_BAD_CODE = 5,
    INF_BAD_REPEAT = 6, INF_DISTANCE = 7, INF_OUTPUT = 8, INF_LENGTH = 9, INF_INPUT = 10, INF_BAD_COUNTS = 11, INF_CRC = 12
};

struct Tables {
    uint16_t lit_table[1 << LIT_BITS];
    uint32_t dist_table[1 << DIST_BITS];
    uint16_t cl_table[1 << CL_BITS];
    uint16_t lit_sym[MAX_LIT];
    uint16_t dist_sym[MAX_DIST];
    uint16_t cl_sym[20];
    int lit_count[16], dist_count[16], cl_count[16];
    int lit_first[16], lit_index[16], dist_first[16], dist_index[16];     // per code length: the first canonical code, its symbol's place in *_sym
    uint8_t lens[MAX_LIT + MAX_DIST + 16];        // literal/length lengths, then the distance lengths
    uint8_t cl_lens[20];
    uint32_t ring[RING_WORDS + RING_MIRROR];
    uint8_t scratch[128];                         // per output byte of a step: the lane of the symbol that starts there (+ 64 bytes
                                                  // nobody reads: where the lanes that are no symbol write, so that no lane is masked)
};

PA_DEV int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// A workgroup here is ONE wavefront, and a wavefront's LDS instructions execute in issue order: hand-offs between lanes
// through LDS need the compiler to keep the program order, nothing else.  (__syncthreads() would also drain vmcnt -- the
// gather of the step before and its store, which are meant to stay in flight.)
PA_DEV void wave_order() { asm volatile("" ::: "memory"); }

// inclusive prefix sum across the wavefront (DPP row shifts, then the row broadcasts)
PA_DEV int wave_scan_add(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
    return v;
}

// inclusive prefix maximum across the wavefront of values >= -1
PA_DEV int wave_scan_max(int v) {
    v = max(v, __builtin_amdgcn_update_dpp(-1, v, 0x111, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(-1, v, 0x112, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(-1, v, 0x114, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(-1, v, 0x118, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(-1, v, 0x142, 0xa, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(-1, v, 0x143, 0xc, 0xf, false));
    return v;
}

// The compressed bytes of one member as a stream of bits, a window of it in LDS: chunk c (64 words) lives in
// ring[(c & 1) * 64 ..], chunks `loaded - 2` and `loaded - 1` are there, chunk `loaded` is in flight in `pre`.
struct Stream {
    const uint32_t* words;      // the word holding the member's first byte
    int n_words;                // words that hold bytes of the member
    uint32_t* ring;
    int loaded;
    uint32_t pre;               // per lane
    int p;                      // bit position of the next symbol, from `words`
    int reach = 160;            // every read at or after p reaches at most this many bits further (lane 63's three words; 64 more
                                // with a second window)

    PA_DEV void store_chunk(int c, uint32_t v) {
        ring[(c & 1) * 64 + threadIdx.x] = v;
        if (!(c & 1) && threadIdx.x < RING_MIRROR) ring[RING_WORDS + threadIdx.x] = v;      // (the mirror of the ring's first words)
    }
    PA_DEV uint32_t fetch(int c) const {
        const int i = c * 64 + (int)threadIdx.x;
        return i < n_words ? words[i] : 0u;
    }
    PA_DEV void seek(int bit) {
        p = bit;
        const int c = bit >> 11;
        wave_order();
        store_chunk(c, fetch(c));
        store_chunk(c + 1, fetch(c + 1));
        loaded = c + 2;
        pre = fetch(loaded);
        wave_order();
    }
    PA_DEV void ensure() {
        while (((p + reach) >> 11) >= loaded) {
            wave_order();
            store_chunk(loaded, pre);
            ++loaded;
            pre = fetch(loaded);
            wave_order();
        }
    }
    // 64 bits from bit position q (the lane's own, or a uniform one): three consecutive words from the ring (its mirror makes them so)
    PA_DEV uint64_t bits_at(int q) const {
        const int sh = q & 31;
        const uint32_t* w = reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(ring) + ((q >> 3) & ((RING_WORDS - 1) << 2)));
        const uint32_t w0 = w[0], w1 = w[1], w2 = w[2];
        const uint32_t lo = __builtin_amdgcn_alignbit(w1, w0, sh), hi = __builtin_amdgcn_alignbit(w2, w1, sh);
        return ((uint64_t)hi << 32) | lo;
    }
    // ... and the 64 bits from q + 64 with them: five consecutive words
    PA_DEV void bits2_at(int q, uint64_t& a, uint64_t& b) const {
        const int sh = q & 31;
        const uint32_t* w = reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(ring) + ((q >> 3) & ((RING_WORDS - 1) << 2)));
        const uint32_t w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3], w4 = w[4];
        a = ((uint64_t)__builtin_amdgcn_alignbit(w2, w1, sh) << 32) | __builtin_amdgcn_alignbit(w1, w0, sh);
        b = ((uint64_t)__builtin_amdgcn_alignbit(w4, w3, sh) << 32) | __builtin_amdgcn_alignbit(w3, w2, sh);
    }
    PA_DEV uint64_t peek() {                    // uniform
        ensure();
        const uint64_t v = bits_at(p);
        return ((uint64_t)(uint32_t)uni((int)(v >> 32)) << 32) | (uint32_t)uni((int)v);
    }
    PA_DEV uint32_t take(int n) {               // n <= 32
        const uint64_t v = peek();
        p += n;
        return (uint32_t)(v & ((1ull << n) - 1ull));
    }
};

// Canonical code of `n` symbols from their lengths (RFC 1951 3.2.2): count[], the symbols in (length, symbol) order and
// the primary table of 2^tbits entries.  Returns false for an over-subscribed set.  All lanes call it.
// What a primary-table entry holds (0 = no code of at most `tbits` bits starts like this index):
//   PLAIN   uint16  length | symbol << 4                                              (the code length code)
//   LITLEN  uint16  length | extra bits << 4 | is-length << 7 | value << 8            value = the literal, or a length
//                   symbol's base - 3; the end-of-block code is "a length with 7 extra bits"; symbols 286 / 287: 0
//   DIST    uint32  length | extra bits << 4 | base << 8                              distance symbols 30 / 31: 0
// so that a lane of the window decode needs no arithmetic on symbol numbers (RFC 1951 3.2.5's tables in closed form here).
enum TableFormat { PLAIN, LITLEN, DIST };

PA_DEV uint32_t table_entry(TableFormat format, int sym, int len) {
    if (format == PLAIN) return (uint32_t)((sym << 4) | len);
    if (format == LITLEN) {
        if (sym < 256) return (uint32_t)(len | sym << 8);
        if (sym == 256) return (uint32_t)(len | 7 << 4 | 1 << 7);
        if (sym > 285) return 0u;
        const int s = sym - 257;
        const bool longer = s >= 8 && s != 28;
        const int eb = longer ? (s - 4) >> 2 : 0;
        const int base = longer ? ((4 + (s & 3)) << eb) + 3 : (s == 28 ? 258 : s + 3);
        return (uint32_t)(len | eb << 4 | 1 << 7 | (base - 3) << 8);
    }
    if (sym > 29) return 0u;
    const int eb = sym < 4 ? 0 : (sym >> 1) - 1;
    const int base = sym < 4 ? sym + 1 : ((2 + (sym & 1)) << eb) + 1;
    return (uint32_t)(len | eb << 4 | base << 8);
}

template <TableFormat FORMAT, typename Entry>
PA_DEV bool build_table(const uint8_t* lens, int n, int* count, uint16_t* syms, Entry* table, int tbits, int* first_of = nullptr,
                        int* index_of = nullptr) {
    const int lane = threadIdx.x;
    if (lane < 16) count[lane] = 0;
    wave_order();
    for (int s = lane; s < n; s += 64) atomicAdd(&count[lens[s]], 1);
    wave_order();
    int offs[16];
    int left = 1, total = 0, first = 0;
    bool ok = true;
    offs[0] = 0;
#pragma unroll
    for (int len = 1; len < 16; ++len) {
        const int c = uni(count[len]);
        offs[len] = total;
        if (first_of && lane == 0) {             // (what long_code() needs of the codes beyond the primary table)
            first_of[len] = first;
            index_of[len] = total;
        }
        total += c;
        first = (first + c) << 1;
        left = (left << 1) - c;
        if (left < 0) ok = false;
    }
    wave_order();
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
    wave_order();
    // The primary table, one entry per lane and pass: index t holds the code whose bits (first bit = bit 0 of t) t starts with.
    // Per length L the canonical codes are the L-bit values first .. first + count - 1 (most significant bit first = the
    // bit-reversed low L bits of t); the counts are uniform, so the walk over the lengths keeps (count, first, index) in scalar
    // registers and every lane does compares only -- the per-lane loop over LDS reads of count[] this replaces was 10 dependent
    // round trips per entry, 16 entries per lane, per block.
    int cnt[16];
#pragma unroll
    for (int len = 1; len < 16; ++len) cnt[len] = uni(count[len]);
    for (int t = lane; t < (1 << tbits); t += 64) {
        const uint32_t rev = __builtin_bitreverse32((uint32_t)t);
        int first = 0, index = 0, where = -1, hit_len = 0;
#pragma unroll
        for (int len = 1; len < 16; ++len) {
            if (len <= tbits) {
                const int rel = (int)(rev >> (32 - len)) - first;
                const bool hit = where < 0 && (unsigned)rel < (unsigned)cnt[len];
                where = hit ? index + rel : where;
                hit_len = hit ? len : hit_len;
                index += cnt[len];
                first = (first + cnt[len]) << 1;
            }
        }
        table[t] = where >= 0 ? (Entry)table_entry(FORMAT, syms[where], hit_len) : (Entry)0;
    }
    wave_order();
    return true;
}

// One symbol from uniform bits: the primary table, or bit by bit for a longer code.  -1: the bits are no code of the set.
// *used: the code's length.
PA_DEV int decode(uint32_t bits, const uint16_t* table, int tbits, const int* count, const uint16_t* syms, int* used) {
    const int e = uni(table[bits & ((1u << tbits) - 1u)]);
    if (e) {
        *used = e & 15;
        return e >> 4;
    }
    int code = 0, first = 0, index = 0;
#pragma unroll 1
    for (int len = 1; len < 16; ++len) {
        code |= (bits >> (len - 1)) & 1;
        const int c = uni(count[len]);
        if (code - c < first) {
            *used = len;
            return uni(syms[index + (code - first)]);
        }
        index += c;
        first = (first + c) << 1;
        code <<= 1;
    }
    *used = 0;
    return -1;
}

// One symbol bit by bit from the canonical arrays alone (RFC 1951 3.2.2) -- the per-symbol path behind the window decode.
PA_DEV int decode_canonical(uint32_t bits, const int* count, const uint16_t* syms, int* used) {
    int code = 0, first = 0, index = 0;
#pragma unroll 1
    for (int len = 1; len < 16; ++len) {
        code |= (bits >> (len - 1)) & 1;
        const int c = uni(count[len]);
        if (code - c < first) {
            *used = len;
            return uni(syms[index + (code - first)]);
        }
        index += c;
        first = (first + c) << 1;
        code <<= 1;
    }
    *used = 0;
    return -1;
}

PA_DEV uint32_t load32(const uint8_t* p) {
    typedef uint32_t __attribute__((aligned(1))) word_any;
    return *reinterpret_cast<const word_any*>(p);
}

// ---- CRC-32 of every inflated member against its trailer --------------------------------------------------------------
// htslib's inflate_block compares crc32() of the inflated block with the member's trailer and fails the read on a mismatch
// (bgzf.c, under sam_itr_next: bam_handler.cpp:341-372); without it a flipped literal in the file is a silently different
// base.  Round 5 did this in a second kernel (one wavefront per member, lane l on its own 1 KiB slice): a pass of its own over
// bytes the inflate wavefront had just written, every load instruction touching 64 different cache lines -- +12 % on the leg at
// 0.04 of HBM.  Now the wavefront that inflated the member checks it in its epilogue, in the LDS its Huffman tables no longer
// need, with COALESCED loads: the member's 4-byte words are dealt round robin from the END (lane l: the words that end 4 (l +
// 64 k) bytes before the member's end), so one load instruction reads 256 contiguous bytes.  A lane's words are 256 bytes
// apart; between two of them its CRC state is advanced over the 252 bytes that belong to the other lanes -- which is linear in
// the state, so "absorb a word, then advance 256 bytes" is ONE slice-by-4 style lookup in four tables built for a 256-byte
// advance instead of a 4-byte one (IEEE 802.3 polynomial, reflected: 0xEDB88320; advancing a state over n zero bytes is a
// multiplication by x^(8n) mod P, zlib's crc32_combine restated).  The lane's last word ends 4 l bytes before the end: one
// multiplication by the constant x^(32 l) brings every lane's state to the member's end, where the 64 states are XORed.  The
// initial state 0xFFFFFFFF and the (len mod 4) head bytes go to the lane that owns the first word.  ~6 k vector instructions
// per member beside the inflate's ~700 k, no second launch, no second pass over HBM.
constexpr uint32_t CRC_POLY = 0xEDB88320u;
constexpr uint32_t crc_multmodp(uint32_t a, uint32_t b) {       // a * b mod P, bit 31 = x^0
    uint32_t m = 1u << 31, p = 0;
    for (;;) {
        if (a & m) {
            p ^= b;
            if ((a & (m - 1)) == 0) break;
        }
        m >>= 1;
        b = (b & 1) ? (b >> 1) ^ CRC_POLY : b >> 1;
    }
    return p;
}
constexpr uint32_t crc_x2nmodp(uint64_t n, unsigned k) {          // x^(n * 2^k) mod P
    uint32_t table = 1u << 30;                                    // x^1
    for (unsigned i = 0; i < k; ++i) table = crc_multmodp(table, table);
    uint32_t p = 1u << 31;                                        // x^0
    while (n) {
        if (n & 1) p = crc_multmodp(table, p);
        n >>= 1;
        table = crc_multmodp(table, table);
    }
    return p;
}
static_assert(crc_x2nmodp(0, 3) == 0x80000000u && crc_multmodp(0x80000000u, 0x12345678u) == 0x12345678u, "x^0 is the identity");

constexpr int CRC_STRIDE = 256;                  // bytes between two words of one lane (64 lanes x 4 bytes)
struct CrcConsts {
    uint32_t bit[4][8];                          // bit[b][j] = x^(8 (CRC_STRIDE - b)) * (the state with only bit 7 - j of its low byte set)
    uint32_t lane_shift[64];                     // x^(32 l): a state advanced over 4 l zero bytes
};
constexpr CrcConsts crc_consts() {
    CrcConsts c{};
    for (int b = 0; b < 4; ++b)
        for (int j = 0; j < 8; ++j) c.bit[b][j] = crc_multmodp(crc_x2nmodp((uint64_t)(CRC_STRIDE - b), 3), 0x80u >> j);
    for (int l = 0; l < 64; ++l) c.lane_shift[l] = crc_x2nmodp((uint64_t)(4 * l), 3);
    return c;
}
__constant__ const CrcConsts CRC_CONSTS = crc_consts();

struct CrcTables {
    uint32_t adv[4][256];                        // adv[b][v]: the state v << 8 b advanced over CRC_STRIDE bytes
    uint32_t one[256];                           // the state v advanced over one byte (the ordinary CRC-32 table)
};

__device__ __forceinline__ uint32_t crc_multmodp_dev(uint32_t a, uint32_t b) {
    uint32_t p = 0;
#pragma unroll 4
    for (int i = 0; i < 32; ++i) {
        p ^= (a & (0x80000000u >> i)) ? b : 0u;
        b = (b >> 1) ^ ((b & 1u) ? CRC_POLY : 0u);
    }
    return p;
}

// CRC-32 of data[0 .. len) by the calling wavefront (all 64 lanes call; the result is uniform).  `T` is LDS the wavefront
// owns; its own earlier stores to `data` are complete (the caller waits for them).
__device__ __forceinline__ uint32_t wave_crc32(const uint8_t* __restrict__ data, int len, CrcTables& T) {
    const int lane = threadIdx.x;
    for (int i = lane; i < 256; i += 64) {
        uint32_t c = (uint32_t)i;
#pragma unroll
        for (int k = 0; k < 8; ++k) c = (c >> 1) ^ ((c & 1u) ? CRC_POLY : 0u);
        T.one[i] = c;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            uint32_t e = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) e ^= (i & (0x80 >> j)) ? CRC_CONSTS.bit[b][j] : 0u;
            T.adv[b][i] = e;
        }
    }
    wave_order();
    const int nw = len >> 2, hb = len & 3;
    const int owner = nw > 0 ? (nw - 1) & 63 : 0;            // the lane whose first word is the member's first whole word
    uint32_t c = 0;
    if (lane == owner) {
        c = 0xFFFFFFFFu;
        for (int i = 0; i < hb; ++i) c = T.one[(c ^ data[i]) & 0xffu] ^ (c >> 8);
    }
    if (lane < nw) {
        const uint8_t* last = data + len - 4 * (lane + 1);      // this lane's last word; its k-th from the end lies CRC_STRIDE k in front
        int k = (nw - 1 - lane) >> 6;
        auto absorb = [&](uint32_t w) {
            c ^= w;
            c = T.adv[0][c & 0xffu] ^ T.adv[1][(c >> 8) & 0xffu] ^ T.adv[2][(c >> 16) & 0xffu] ^ T.adv[3][c >> 24];
        };
        for (; k >= 4; k -= 4) {                                 // four loads in flight in front of the dependent lookups
            const uint32_t w0 = load32(last - (size_t)CRC_STRIDE * k), w1 = load32(last - (size_t)CRC_STRIDE * (k - 1));
            const uint32_t w2 = load32(last - (size_t)CRC_STRIDE * (k - 2)), w3 = load32(last - (size_t)CRC_STRIDE * (k - 3));
            absorb(w0);
            absorb(w1);
            absorb(w2);
            absorb(w3);
        }
        for (; k >= 1; --k) absorb(load32(last - (size_t)CRC_STRIDE * k));
        c ^= load32(last);
#pragma unroll
        for (int i = 0; i < 4; ++i) c = T.one[c & 0xffu] ^ (c >> 8);
        c = crc_multmodp_dev(CRC_CONSTS.lane_shift[lane], c);   // over the 4 l bytes of the lanes behind this one
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c ^= (uint32_t)__shfl_xor((int)c, o, 64);
    return c ^ 0xFFFFFFFFu;
}

// A code LONGER than the primary table, from uniform bits, all candidate lengths at once: lane j asks whether the first
// min_len + j bits are a code of that length (canonical codes of length L are the L-bit values first[L] .. first[L] + count[L]
// - 1; a smaller value continues a shorter code) -- the shortest length that says yes is the code.  Two LDS round trips instead
// of decode_canonical's one per bit.  -> the symbol, *used = its length; -1: the bits are no code of the set.
PA_DEV int long_code(uint32_t bits, const int* count, const uint16_t* syms, const int* first_of, const int* index_of, int min_len,
                     int* used) {
    const int j = threadIdx.x, L = min(min_len + j, 15);
    const int code = (int)(__builtin_bitreverse32(bits) >> (32 - L));
    const int rel = code - first_of[L];
    const bool hit = j < 16 - min_len && (unsigned)rel < (unsigned)count[L];
    const unsigned long long mask = __ballot(hit);
    if (!mask) { *used = 0; return -1; }
    const int w = __builtin_ctzll(mask);
    const int sym = hit ? (int)syms[index_of[L] + rel] : 0;
    *used = min_len + w;
    return __builtin_amdgcn_readlane(sym, w);
}

// what a lane found at its bit offset: the walk's advance (the symbol's total bits: code, extra bits, for a match also the distance's)
constexpr int A_END = 64, A_INVALID = 128;      // (an ordinary symbol advances by its bits: at most 15 + 5 + 15 + 13 = 48)

// WIDE: a step looks at TWO windows of 64 bit offsets while the steps are small (the last one wrote at most `wide_below` bytes; both
// windows' output must fit the step's 64 output lanes, else the second one is dropped and decoded again by the next step): the
// lookups, the walk and the prefix sums run twice, everything per step -- the stream's window, the output placement, the gather, the
// pending store -- once for twice the symbols.  Measured (profiles/r06_inflate_wide_ab.txt, counters beside it): the steps of the
// level-1 bench members halve (60.3 M -> 32.8 M), vector instructions fall by 12 % (8.51 G -> 7.52 G per launch), scalar ones not
// at all (7.57 G -> 7.55 G: they are per SYMBOL -- the walk -- not per step), time by 5.6 %: per-step work was a tenth of this
// kernel.  What binds it is what the header says: the CU's one scalar unit (7.55 G instructions / 256 CUs = 0.78 of its issue slots
// over the launch) beside the vector units at the same 0.78.
constexpr int WIDE_BELOW = 48;
#ifndef PA_INFLATE_WIDE_DEFAULT
#define PA_INFLATE_WIDE_DEFAULT 1
#endif
// STATS: the step / match / long-code counts of PA_INFLATE_DEBUG (a handful of scalar instructions per step on the unit that binds the
// kernel: compiled in only for the launches that ask for them).
template <int WAVES, bool WIDE, bool STATS>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) void bgzf_inflate_kernel(const uint8_t* __restrict__ comp, const int64_t* __restrict__ comp_off,
                                                         const int32_t* __restrict__ comp_len, const int64_t* __restrict__ out_off,
                                                         const int32_t* __restrict__ out_len, uint8_t* out_base,
                                                         int32_t* __restrict__ status, unsigned long long* dbg,
                                                         int64_t comp_bytes, int wide_below) {
    __shared__ union Lds { Tables t; CrcTables crc; } lds;      // (the CRC tables take the Huffman tables' place in the epilogue)
    Tables& T = lds.t;
    int n_steps = 0, n_match = 0, n_fallback = 0, n_blocks_in = 0;      // statistics for PA_INFLATE_DEBUG
    const int lane = threadIdx.x;
    const int blk = blockIdx.x;
    const int64_t coff = comp_off[blk];
    const int clen = comp_len[blk], olen = out_len[blk];
    uint8_t* out = out_base + out_off[blk];
    const int mis = (int)(coff & 3);
    Stream in;
    in.words = reinterpret_cast<const uint32_t*>(comp + (coff - mis));
    in.n_words = (mis + clen + 3) >> 2;
    in.ring = T.ring;
    in.seek(mis * 8);
    const uint8_t* in_bytes = comp + (coff - mis);
    const unsigned long long below = (1ull << lane) - 1ull;

    int pos = 0;                 // bytes written
    int err = INF_OK;
    bool wide = WIDE;            // the next step takes two windows
    int bad_dist = 0;            // per lane: a match of this member reached in front of it
    in.reach = WIDE ? 224 : 160;
    // (A wavefront's vector memory operations reach its L1 in program order: a load issued after a store of the same
    // wavefront to the same bytes returns them -- the ordinary single-thread guarantee every in-place loop relies on; the
    // tests run every RLE-like pattern, where a match reads what the instruction before it wrote.)
    auto copy_match = [&](int dst, int len, int dist) {
        const uint8_t* src = out + dst - dist;
        if (dist >= len) {
            for (int k = lane; k < len; k += 64) out[dst + k] = src[k];
        } else {
            for (int k = lane; k < len; k += 64) out[dst + k] = src[k % dist];
        }
    };

    // The bytes of a step whose matches read memory are completed one step later: their loads are in flight while the next
    // window is decoded.  pend_n bytes at pend_pos: lane j's byte is the value of lane pend_root[j] -- a literal (pend_val
    // >= 0) or what that lane loaded (pend_byte).
    int pend_n = 0, pend_pos = 0;
    int pend_val = 0, pend_root = 0;
    uint8_t pend_byte = 0;
    auto complete = [&]() {          // (with nothing pending no lane stores: no test of pend_n on the scalar unit)
        const int own = pend_val >= 0 ? pend_val : (int)pend_byte;
        const int v = __builtin_amdgcn_ds_bpermute(pend_root << 2, own);
        if (lane < pend_n) out[pend_pos + lane] = (uint8_t)v;
        pend_n = 0;
    };

    bool last = olen == 0 && clen == 0;        // nothing at all: an empty member without a stream
    while (!last && !err) {
        last = in.take(1) != 0;
        const int type = (int)in.take(2);
        if (type == 0) {
            // stored: to the next byte, LEN, ~LEN, the bytes
            complete();
            in.p = (in.p + 7) & ~7;
            const uint32_t len = in.take(16), nlen = in.take(16);
            if ((len ^ nlen) != 0xffffu) { err = INF_STORED_LEN; break; }
            const int from = in.p >> 3;
            if (from + (int)len > mis + clen) { err = INF_INPUT; break; }
            if (pos + (int)len > olen) { err = INF_OUTPUT; break; }
            for (int k = lane; k < (int)len; k += 64) out[pos + k] = in_bytes[from + k];
            pos += (int)len;
            in.seek(in.p + 8 * (int)len);
            continue;
        }
        if (type == 3) { err = INF_BAD_BLOCK_TYPE; break; }
        if (type == 1) {
            for (int s = lane; s < 288; s += 64) T.lens[s] = s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8;
            if (lane < 32) T.lens[288 + lane] = 5;
        } else {
            const int hlit = (int)in.take(5) + 257, hdist = (int)in.take(5) + 1, hclen = (int)in.take(4) + 4;
            if (hlit > 286 || hdist > 30) { err = INF_BAD_COUNTS; break; }
            if (lane < 19) T.cl_lens[lane] = 0;
            wave_order();
            // the order of the code length code lengths (RFC 1951 3.2.7), 5 bits each
            const unsigned long long lo = 16ull | 17ull << 5 | 18ull << 10 | 0ull << 15 | 8ull << 20 | 7ull << 25 | 9ull << 30 |
                                          6ull << 35 | 10ull << 40 | 5ull << 45 | 11ull << 50 | 4ull << 55;
            const unsigned long long hi = 12ull | 3ull << 5 | 13ull << 10 | 2ull << 15 | 14ull << 20 | 1ull << 25 | 15ull << 30;
            {
                // 19 x 3 bits = 57 bits: one peek, every lane its own code length
                const uint64_t v = in.peek();
                if (lane < hclen) {
                    const int idx = lane < 12 ? (int)((lo >> (5 * lane)) & 31) : (int)((hi >> (5 * (lane - 12))) & 31);
                    T.cl_lens[idx] = (uint8_t)((v >> (3 * lane)) & 7);
                }
                in.p += 3 * hclen;
            }
            wave_order();
            if (!build_table<PLAIN>(T.cl_lens, 19, T.cl_count, T.cl_sym, T.cl_table, CL_BITS)) { err = INF_OVERSUBSCRIBED; break; }
            const int total = hlit + hdist;
            int i = 0, prev = 0;
            while (i < total) {
                const uint32_t v = (uint32_t)in.peek();
                int used;
                const int sym = decode(v, T.cl_table, CL_BITS, T.cl_count, T.cl_sym, &used);
                if (sym < 0) { err = INF_BAD_CODE; break; }
                if (sym < 16) {
                    if (lane == 0) T.lens[i] = (uint8_t)sym;
                    prev = sym;
                    ++i;
                    in.p += used;
                    continue;
                }
                int rep, val = 0;
                if (sym == 16) {
                    if (i == 0) { err = INF_BAD_REPEAT; break; }
                    val = prev;
                    rep = 3 + (int)((v >> used) & 3);
                    used += 2;
                } else if (sym == 17) {
                    rep = 3 + (int)((v >> used) & 7);
                    used += 3;
                    prev = 0;
                } else {
                    rep = 11 + (int)((v >> used) & 127);
                    used += 7;
                    prev = 0;
                }
                in.p += used;
                if (i + rep > total) { err = INF_BAD_REPEAT; break; }
                for (int k = lane; k < rep; k += 64) T.lens[i + k] = (uint8_t)val;
                i += rep;
            }
            if (err) break;
            wave_order();
            if (uni(T.lens[256]) == 0) { err = INF_NO_END_CODE; break; }
            // the distance lengths follow the literal/length ones directly: move them to their own place
            uint8_t dl = 0;
            if (lane < hdist) dl = T.lens[hlit + lane];
            wave_order();
            if (lane < 32) T.lens[288 + lane] = lane < hdist ? dl : 0;
            for (int s = hlit + lane; s < 288; s += 64) T.lens[s] = 0;
        }
        wave_order();
        if (!build_table<LITLEN>(T.lens, 288, T.lit_count, T.lit_sym, T.lit_table, LIT_BITS, T.lit_first, T.lit_index) ||
            !build_table<DIST>(T.lens + 288, 32, T.dist_count, T.dist_sym, T.dist_table, DIST_BITS, T.dist_first, T.dist_index)) {
            err = INF_OVERSUBSCRIBED;
            break;
        }
        // ---- the symbols of the block: 64 bit offsets at a time ----
        // Lane k decodes whatever starts at bit p + k (most offsets are inside a symbol: their result is never looked at);
        // the symbols that do start in the window are found by following the lengths from offset 0.
        if (STATS) ++n_blocks_in;
        for (bool more = true; more && !err;) {
            if (STATS) ++n_steps;
            in.ensure();
            // what starts at a bit offset: the symbol's bits | flags, its literal / length, its distance
            auto lookup = [&](uint32_t lo, uint32_t hi, int& adv, int& kind, int& val, int& dist) {
                const uint32_t e = T.lit_table[lo & ((1u << LIT_BITS) - 1u)];
                const int len = (int)(e & 15u), eb = (int)((e >> 4) & 7u);
                // as if it were a length symbol: its extra bits, the distance code behind them, that code's extra bits
                const uint32_t behind = __builtin_amdgcn_alignbit(hi, lo, (uint32_t)(len + eb));     // (at most 15 + 7 bits in)
                const uint32_t de = T.dist_table[behind & ((1u << DIST_BITS) - 1u)];
                const int dlen = (int)(de & 15u), deb = (int)((de >> 4) & 15u);
                // Selects, not branches, and every select on ONE comparison: a lane mask is a scalar register pair, and combining
                // two of them is work for the scalar unit -- the unit this kernel is bound by.
                // adv: what the walk adds to its offset -- the symbol's bits (at most 48), A_END | its code length for the end-of-block
                // code, A_INVALID for bits the tables do not hold: both of those end the walk by themselves (>= 64).  kind: 1 a
                // literal, 2 a match, 0 neither.
                const int total_m = len + eb + dlen + deb;
                const int adv_m = de == 0u ? A_INVALID : total_m;
                const int adv_l = eb == 7 ? (len | A_END) : adv_m;
                const int adv_e = (e & 128u) ? adv_l : len;
                adv = e == 0u ? A_INVALID : adv_e;
                const int kind_l = adv_l < 64 ? 2 : 0;
                const int kind_e = (e & 128u) ? kind_l : 1;
                kind = e == 0u ? 0 : kind_e;
                const int xlen = (int)__builtin_amdgcn_ubfe(lo, (uint32_t)len, (uint32_t)eb);
                val = (int)(e >> 8) + ((e & 128u) ? 3 + xlen : 0);
                dist = (int)(de >> 8) + (int)__builtin_amdgcn_ubfe(behind, (uint32_t)dlen, (uint32_t)deb);
            };
            // Follow the symbols of a window of 64 bit offsets from offset k on (eight scalar instructions per symbol, written out:
            // the scalar unit is shared by the CU's four SIMDs and is the unit this kernel keeps busiest -- the compiler's version of
            // the loop took fourteen).  An offset whose code is longer than a primary table is worked out from the stream's bits at
            // that offset (long_code: all candidate lengths at once), written into lane k, and the walk goes on.  Ends with k >= 64,
            // or with f = what stopped it at offset k (the end-of-block code, or bits that are no code of the block's set).
            auto walk = [&](int base, int& adv, int& kind, int& val, int& dist, unsigned long long& chain, int& k, int& f) {
                for (;;) {
                    // four scalar instructions per symbol: the end-of-block code and undecodable bits carry an advance of 64 or more, so
                    // ONE test ends the walk for them as for the window's end; which it was is sorted out behind the loop
                    asm volatile("1:\n\t"
                                 "v_readlane_b32 %[f], %[adv], %[k]\n\t"
                                 "s_bitset1_b64 %[chain], %[k]\n\t"
                                 "s_add_i32 %[k], %[k], %[f]\n\t"
                                 "s_cmpk_lt_i32 %[k], 64\n\t"
                                 "s_cbranch_scc1 1b\n\t"
                                 : [chain] "+s"(chain), [k] "+s"(k), [f] "=&s"(f)
                                 : [adv] "v"(adv)
                                 : "scc");
                    if (f < 64) return;                      // the window ran out behind an ordinary symbol: k >= 64
                    k -= f;                                  // the offset of what stopped the walk; it is no symbol of the step
                    chain &= ~(1ull << k);
                    if (f < A_INVALID) return;               // the end-of-block code: f = A_END | its length
                    if (STATS) ++n_fallback;
                    const uint64_t bits_k = in.bits_at(in.p + base + k);            // (uniform: every lane reads the same three words)
                    const uint32_t lo_k = (uint32_t)uni((int)(uint32_t)bits_k), hi_k = (uint32_t)uni((int)(uint32_t)(bits_k >> 32));
                    uint32_t e_k = (uint32_t)uni((int)T.lit_table[lo_k & ((1u << LIT_BITS) - 1u)]);
                    if (e_k == 0u) {
                        int used;
                        const int sym = long_code(lo_k, T.lit_count, T.lit_sym, T.lit_first, T.lit_index, LIT_BITS + 1, &used);
                        if (sym < 0) return;
                        e_k = table_entry(LITLEN, sym, used);
                        if (e_k == 0u) return;                             // (symbols 286 / 287: the per-symbol path names the error)
                    }
                    const int len_k = (int)(e_k & 15u), eb_k = (int)((e_k >> 4) & 7u);
                    int adv_k, kind_k, val_k, dist_k = 0;
                    if (!(e_k & 128u)) {
                        adv_k = len_k;
                        kind_k = 1;
                        val_k = (int)(e_k >> 8);
                    } else if (eb_k == 7) {
                        adv_k = len_k | A_END;
                        kind_k = 0;
                        val_k = 0;
                    } else {
                        const uint32_t behind_k = __builtin_amdgcn_alignbit(hi_k, lo_k, (uint32_t)(len_k + eb_k));
                        uint32_t de_k = (uint32_t)uni((int)T.dist_table[behind_k & ((1u << DIST_BITS) - 1u)]);
                        if (de_k == 0u) {
                            int used;
                            const int ds = long_code(behind_k, T.dist_count, T.dist_sym, T.dist_first, T.dist_index, DIST_BITS + 1, &used);
                            if (ds < 0) return;
                            de_k = table_entry(DIST, ds, used);
                            if (de_k == 0u) return;                        // (distance symbols 30 / 31)
                        }
                        const int dlen_k = (int)(de_k & 15u), deb_k = (int)((de_k >> 4) & 15u);
                        adv_k = len_k + eb_k + dlen_k + deb_k;
                        kind_k = 2;
                        val_k = (int)(e_k >> 8) + 3 + (int)__builtin_amdgcn_ubfe(lo_k, (uint32_t)len_k, (uint32_t)eb_k);
                        dist_k = (int)(de_k >> 8) + (int)__builtin_amdgcn_ubfe(behind_k, (uint32_t)dlen_k, (uint32_t)deb_k);
                    }
                    const bool here = lane == k;
                    adv = here ? adv_k : adv;
                    kind = here ? kind_k : kind;
                    val = here ? val_k : val;
                    dist = here ? dist_k : dist;
                }
            };
            // what a lane is as a symbol of the step: 0 nothing (or the end code: it writes nothing), 1 a literal, 2 a match
            auto kind_of = [&](unsigned long long chain, int kind) {
                const int on = (int)((chain >> lane) & 1ull);
                return on ? kind : 0;
            };
            // ---- window A: the 64 bit offsets from p; window B (WIDE, when the steps are small): the 64 behind them ----
            int adv, kindv, val, dist;
            int adv_b = A_INVALID, kindv_b = 0, val_b = 0, dist_b = 0;
            const bool two = WIDE && wide;                  // (uniform)
            if (two) {
                uint64_t bits, bits_b;
                in.bits2_at(in.p + lane, bits, bits_b);
                lookup((uint32_t)bits, (uint32_t)(bits >> 32), adv, kindv, val, dist);
                lookup((uint32_t)bits_b, (uint32_t)(bits_b >> 32), adv_b, kindv_b, val_b, dist_b);
            } else {
                const uint64_t bits = in.bits_at(in.p + lane);
                lookup((uint32_t)bits, (uint32_t)(bits >> 32), adv, kindv, val, dist);
            }
            unsigned long long chain = 0, chain_b = 0;
            int k = 0, f, stop = 0;
            walk(0, adv, kindv, val, dist, chain, k, f);
            if (k < 64) {                                   // the end-of-block code, or something that is no code of the block's set
                if (f < A_INVALID) { k += f & 15; stop = 1; }
                else stop = 2;
            }
            int consumed = k, k_b = 0, stop_b = 0;
            bool with_b = two && stop == 0;                 // (window A ran through: the next symbol starts at offset k - 64 of B)
            if (with_b) {
                k_b = k - 64;
                walk(64, adv_b, kindv_b, val_b, dist_b, chain_b, k_b, f);
                if (k_b < 64) {
                    if (f < A_INVALID) { k_b += f & 15; stop_b = 1; }
                    else stop_b = 2;
                }
            }
            const int kind = kind_of(chain, kindv);
            const bool is_lit = kind == 1, is_match = kind == 2;
            unsigned long long matches = __ballot(is_match);
            const unsigned long long lits = __ballot(is_lit);
            int kind_b = 0, off_b = 0, produced_b = 0;
            bool is_lit_b = false, is_match_b = false;
            unsigned long long matches_b = 0, lits_b = 0;
            if (with_b) {
                kind_b = kind_of(chain_b, kindv_b);
                is_lit_b = kind_b == 1;
                is_match_b = kind_b == 2;
                matches_b = __ballot(is_match_b);
                lits_b = __ballot(is_lit_b);
            }
            int off, produced;
            if ((matches | matches_b) == 0) {
                off = __popcll(lits & below);
                produced = __popcll(lits);
                if (with_b) {
                    off_b = produced + __popcll(lits_b & below);
                    produced_b = __popcll(lits_b);
                }
            } else {
                const int mine = is_lit ? 1 : (is_match ? val : 0);
                const int incl = wave_scan_add(mine);
                off = incl - mine;
                produced = __builtin_amdgcn_readlane(incl, 63);
                if (with_b) {
                    const int mine_b = is_lit_b ? 1 : (is_match_b ? val_b : 0);
                    const int incl_b = wave_scan_add(mine_b);
                    off_b = produced + incl_b - mine_b;
                    produced_b = __builtin_amdgcn_readlane(incl_b, 63);
                }
            }
            if (with_b && produced + produced_b > 64) {
                // too much for one step's 64 output lanes: window B is dropped (its symbols are decoded again by the next step)
                with_b = false;
                matches_b = 0;
                produced_b = 0;
            }
            if (with_b) {
                consumed = 64 + k_b;
                stop = stop_b;
                produced += produced_b;
            } else {
                is_lit_b = is_match_b = false;
                kind_b = 0;
            }
            if (WIDE) wide = produced <= wide_below;         // (the next step takes two windows while the steps stay small)
            k = consumed;
            if (pos + produced > olen) { err = INF_OUTPUT; break; }
            if ((matches | matches_b) == 0) {
                if (is_lit) out[pos + off] = (uint8_t)val;            // (no read of memory: whatever is pending stays pending)
                if (is_lit_b) out[pos + off_b] = (uint8_t)val_b;
            } else if (produced <= 64) {
                // Every output byte of the step on its own lane: which symbol it belongs to (the symbols mark their first
                // byte; a prefix maximum spreads the mark), then a literal's value or a match byte's source -- memory in front
                // of the step, or an earlier byte of this same step, followed back to a literal or a memory byte.
                T.scratch[lane] = 0xff;
                wave_order();
                T.scratch[kind != 0 ? off : 64 + lane] = (uint8_t)lane;
                T.scratch[kind_b != 0 ? off_b : 64 + lane] = (uint8_t)(64 + lane);
                wave_order();
                const int mark = T.scratch[lane];
                const int packed = wave_scan_max(mark != 0xff ? (lane << 8 | mark) : -1);
                const int start = packed >> 8;
                int word = __builtin_amdgcn_ds_bpermute((packed & 63) << 2, is_match ? (val | dist << 9 | 1 << 25) : val);
                if (with_b) {
                    const int word_b = __builtin_amdgcn_ds_bpermute((packed & 63) << 2, is_match_b ? (val_b | dist_b << 9 | 1 << 25) : val_b);
                    word = (packed & 64) ? word_b : word;
                }
                const int mi = lane < produced ? (word >> 25) & 1 : 0;            // this byte comes from a match
                const bool m = mi != 0;
                const int sval = word & 511, d = (word >> 9) & 0xffff;
                // (a distance that reaches in front of the member: remembered per lane and reported when the block ends -- one vector
                // instruction here instead of a test and a branch on the scalar unit; the source is clamped into the member meanwhile)
                bad_dist |= (m ? d : 0) > pos + start ? 1 : 0;
                int rel = lane - start;
                const bool wraps = (m ? d : 0x7fffffff) < sval;                  // distance < length: the source repeats
                if (__ballot(wraps)) {
                    if (wraps) rel %= d;
                }
                const int src = max(pos + start - d + rel, 0);
                const bool in_step = (m ? src : -1) >= pos, from_mem = (m ? src : 0x7fffffff) < pos;
                int root = in_step ? src - pos : lane;
                if (__ballot(in_step)) {
                    for (;;) {
                        const int next = __builtin_amdgcn_ds_bpermute(root << 2, root);
                        if (!__ballot(next != root)) break;
                        root = next;
                    }
                }
                if (STATS) n_match += __popcll(matches) + __popcll(matches_b);
                if (pend_n && __ballot((from_mem ? src : -1) >= pend_pos)) complete();
                const uint8_t b = out[from_mem ? src : 0];              // (every lane loads: the others the member's first byte, one line)
                complete();                                            // the step before: its loads were issued a step ago
                pend_byte = b;
                pend_val = lane < produced ? (m ? -1 : sval) : -1;
                pend_root = root;
                pend_n = produced;
                pend_pos = pos;
                matches = 0;
            } else {
                // a step of more than 64 bytes (long matches; window A alone): the literals, then match by match in order
                complete();
                if (is_lit) out[pos + off] = (uint8_t)val;
                while (matches) {
                    const int m = __builtin_ctzll(matches);
                    matches &= matches - 1;
                    const int mlen = __builtin_amdgcn_readlane(val, m), md = __builtin_amdgcn_readlane(dist, m);
                    const int dst = pos + __builtin_amdgcn_readlane(off, m);
                    if (md > dst) { err = INF_DISTANCE; break; }
                    if (STATS) ++n_match;
                    copy_match(dst, mlen, md);
                }
            }
            if (err) break;
            pos += produced;
            in.p += k;
            if (stop == 1) more = false;
            else if (stop == 2) {
                complete();
                // bits the walk could not resolve (no code of the set, an invalid symbol): this one symbol from uniform bits, which names
                // the error
                const uint64_t v = in.peek();
                int used;
                int s = decode_canonical((uint32_t)v, T.lit_count, T.lit_sym, &used);
                if (s < 0) { err = INF_BAD_CODE; break; }
                if (s < 256) {
                    if (pos >= olen) { err = INF_OUTPUT; break; }
                    if (lane == 0) out[pos] = (uint8_t)s;
                    ++pos;
                    in.p += used;
                } else if (s == 256) {
                    in.p += used;
                    more = false;
                } else {
                    if (s > 285) { err = INF_LENGTH; break; }
                    s -= 257;
                    uint64_t after = v >> used;
                    int mlen;
                    if (s < 8) mlen = s + 3;
                    else if (s == 28) mlen = 258;
                    else {
                        const int eb = (s - 4) >> 2;
                        mlen = ((4 + (s & 3)) << eb) + 3 + (int)((uint32_t)after & ((1u << eb) - 1u));
                        after >>= eb;
                        used += eb;
                    }
                    int dused;
                    const int dsym = decode_canonical((uint32_t)after, T.dist_count, T.dist_sym, &dused);
                    if (dsym < 0 || dsym > 29) { err = INF_BAD_CODE; break; }
                    after >>= dused;
                    used += dused;
                    int md;
                    if (dsym < 4) md = dsym + 1;
                    else {
                        const int deb = (dsym >> 1) - 1;
                        md = ((2 + (dsym & 1)) << deb) + 1 + (int)((uint32_t)after & ((1u << deb) - 1u));
                        used += deb;
                    }
                    if (md > pos) { err = INF_DISTANCE; break; }
                    if (pos + mlen > olen) { err = INF_OUTPUT; break; }
                    copy_match(pos, mlen, md);
                    pos += mlen;
                    in.p += used;
                }
            }
        }
        if (!err && __ballot(bad_dist != 0)) err = INF_DISTANCE;
    }
    complete();
    if (!err && pos != olen) err = INF_OUTPUT;
    if (!err && in.p > (mis + clen) * 8) err = INF_INPUT;
    // The member's CRC-32 against its trailer (the 4 bytes behind its DEFLATE bytes), by the wavefront that wrote it: as the host
    // reader, a span that ends before the trailer is not checked (a caller that hands over bare DEFLATE streams).
    // (the member's place in `comp` is read again here rather than kept in scalar registers through the symbol loop)
    const int64_t trailer = *const_cast<const volatile int64_t*>(&comp_off[blk]) + *const_cast<const volatile int32_t*>(&comp_len[blk]);
    if (!err && trailer + 4 <= comp_bytes) {
        __builtin_amdgcn_s_waitcnt(0x0f70);       // vmcnt(0): this wavefront's stores of the member have been taken by the memory system
        wave_order();
        const uint32_t crc = wave_crc32(out, olen, lds.crc);
        const uint8_t* t = comp + trailer;
        const uint32_t want = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
        if (crc != want) err = INF_CRC;
    }
    if (lane == 0) status[blk] = err;
    if (STATS && dbg && lane == 0) {
        atomicAdd(&dbg[0], (unsigned long long)n_steps);
        atomicAdd(&dbg[1], (unsigned long long)n_match);
        atomicAdd(&dbg[2], (unsigned long long)n_fallback);
        atomicAdd(&dbg[3], (unsigned long long)n_blocks_in);
        atomicAdd(&dbg[5], (unsigned long long)pos);
    }
}

// ---- the BAM records of an inflated span ---------------------------------------------------------------------------
// A BAM record says how long it is and nothing else: finding the records is a pointer chase.  The linear index gives a
// record start per 16 kb window; one lane per such entry follows the block sizes up to the next entry (one dependent
// 4-byte load per record, the 32-byte core read beside it), then one wavefront per record sums the reference bases of its
// operations.  What leaves the device is 40 bytes per record instead of the span.
struct RecHdr { int64_t data_off; int32_t ref_id, pos, l_seq, n_cigar, flags, ref_len, state, block_size; };   // = pa_record_header

// flags[0]: 1 a lane ran out of slots, 2 a record shorter than its core, 8 a lane's walk overshot the next entry (the entries
// are not record starts);  flags[1]: 1 the span ends inside a record
__global__ __launch_bounds__(64) void record_chase_kernel(const uint8_t* __restrict__ data, int64_t data_bytes,
                                                         const int64_t* __restrict__ entries, int n_entries, int cap,
                                                         RecHdr* __restrict__ slots, int32_t* __restrict__ counts, int32_t* flags) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n_entries) return;
    int64_t at = entries[i];
    const int64_t stop = i + 1 < n_entries ? entries[i + 1] : data_bytes;
    RecHdr* mine = slots + (size_t)i * cap;
    int j = 0;
    while (at < stop) {
        if (at + 4 > data_bytes) { atomicOr(&flags[1], 1); break; }
        const uint32_t bs = load32(data + at);
        if (bs < 32) { atomicOr(&flags[0], 2); break; }
        if (at + 4 + (int64_t)bs > data_bytes) { atomicOr(&flags[1], 1); break; }
        if (j >= cap) { atomicOr(&flags[0], 1); break; }
        const uint8_t* R = data + at + 4;
        const uint32_t w2 = load32(R + 8), w3 = load32(R + 12);
        RecHdr h;
        h.ref_id = (int32_t)load32(R);
        h.pos = (int32_t)load32(R + 4);
        h.l_seq = (int32_t)load32(R + 16);
        h.n_cigar = (int32_t)(w3 & 0xffffu);
        h.flags = (int32_t)((w3 >> 16) | (((w2 >> 8) & 0xffu) << 16));
        const int64_t o_cigar = 32 + (int64_t)(w2 & 0xffu);
        h.data_off = at + 4 + o_cigar;
        h.ref_len = 0;
        h.block_size = (int32_t)bs;
        const int64_t o_aux = o_cigar + 4ll * h.n_cigar + ((int64_t)(uint32_t)h.l_seq + 1) / 2 + (int64_t)(uint32_t)h.l_seq;
        h.state = (h.l_seq < 0 || o_aux > (int64_t)bs) ? 2 : 0;
        mine[j++] = h;
        at += 4 + (int64_t)bs;
    }
    // the walk from an entry must land exactly on the next one: an entry that is not a record start (a stale or foreign index)
    // leaves this lane past it and the next lane parsing garbage as headers -- flag it (8) and the caller takes the host walk
    if (i + 1 < n_entries && at != stop && !(at < stop)) atomicOr(&flags[0], 8);
    counts[i] = j;
}

// exclusive prefix of the lanes' record counts (n_entries is a few hundred at most: one wavefront, 64 at a time)
__global__ __launch_bounds__(64) void record_base_kernel(const int32_t* __restrict__ counts, int n_entries, int32_t* __restrict__ base,
                                                        const int32_t* flags, int32_t* tail) {
    int carry = 0;
    for (int i0 = 0; i0 < n_entries; i0 += 64) {
        const int i = i0 + (int)threadIdx.x;
        const int c = i < n_entries ? counts[i] : 0;
        const int incl = wave_scan_add(c);
        if (i < n_entries) base[i] = carry + incl - c;
        carry += __builtin_amdgcn_readlane(incl, 63);
    }
    if (threadIdx.x == 0) {
        base[n_entries] = carry;
        if (tail) {                              // (page-locked host memory: what the caller reads after the stream's last kernel)
            tail[0] = carry;
            tail[1] = flags[0];
            tail[2] = flags[1];
        }
    }
}

// one wavefront per slot: the reference bases of the record's operations; the placeholder of a CIGAR kept in the CG tag
// (kSmN, SAM specification 4.2.2: first operation S over all bases) marks the record for the host path
__global__ __launch_bounds__(256) void record_finish_kernel(const uint8_t* __restrict__ data, const RecHdr* __restrict__ slots,
                                                           const int32_t* __restrict__ counts, const int32_t* __restrict__ base,
                                                           int cap, int n_entries, RecHdr* __restrict__ out, long long out_cap) {
    // sixteen workgroups = 64 wavefronts per entry, striding over its records (a wavefront per SLOT would be mostly empty ones)
    const int i = blockIdx.x >> 4, lane = threadIdx.x & 63;
    if (i >= n_entries) return;
    const int n = counts[i];
    for (int j = (blockIdx.x & 15) * 4 + ((int)threadIdx.x >> 6); j < n; j += 64) {
    RecHdr h = slots[(size_t)i * cap + j];
    int ref = 0;
    if (h.state == 0) {
        const uint8_t* cig = data + h.data_off;
        for (int k0 = 0; k0 < h.n_cigar; k0 += 64) {
            const int k = k0 + lane;
            const uint32_t c = k < h.n_cigar ? load32(cig + 4ll * k) : 5u;
            const int op = (int)(c & 15u);
            ref += (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) ? (int)(c >> 4) : 0;
        }
        for (int o = 32; o > 0; o >>= 1) ref += __shfl_xor(ref, o, 64);
        if (h.n_cigar >= 1) {
            const uint32_t c0 = load32(cig);
            if ((c0 & 15u) == 4u && (int)(c0 >> 4) == h.l_seq) h.state = 1;
        }
    }
    if (lane == 0 && (long long)base[i] + j < out_cap) {
        h.ref_len = ref;
        out[base[i] + j] = h;
    }
    }
}

}  // namespace

void launch_record_walk(hipStream_t stream, const uint8_t* data, int64_t data_bytes, const int64_t* entries, int n_entries, int cap,
                        void* slots, int32_t* counts, int32_t* base, int32_t* flags, void* out, int64_t out_cap, int32_t* tail) {
    if (n_entries <= 0) return;
    hipLaunchKernelGGL(record_chase_kernel, dim3((unsigned)((n_entries + 63) / 64)), dim3(64), 0, stream, data, data_bytes, entries,
                       n_entries, cap, static_cast<RecHdr*>(slots), counts, flags);
    hipLaunchKernelGGL(record_base_kernel, dim3(1), dim3(64), 0, stream, counts, n_entries, base, flags, tail);
    hipLaunchKernelGGL(record_finish_kernel, dim3((unsigned)n_entries * 16u), dim3(256), 0, stream, data,
                       static_cast<const RecHdr*>(slots), counts, base, cap, n_entries, static_cast<RecHdr*>(out), (long long)out_cap);
}

void launch_bgzf_inflate(hipStream_t stream, const uint8_t* comp, const int64_t* comp_off, const int32_t* comp_len,
                         const int64_t* out_off, const int32_t* out_len, uint8_t* out, int32_t* status, int n_blocks,
                         int64_t comp_bytes, unsigned long long* debug_counts) {
    if (n_blocks <= 0) return;
    // (one launch: inflate, then every member's CRC-32 against its trailer in the same wavefront's epilogue)
    // PA_INFLATE_WIDE=0 / 1: the one-window / two-window step (A/B runs; the default is what measured faster)
    static const int wide = [] { const char* v = getenv("PA_INFLATE_WIDE"); return v ? atoi(v) != 0 : PA_INFLATE_WIDE_DEFAULT; }();
    static const int below = [] { const char* v = getenv("PA_INFLATE_WIDE_BELOW"); return v ? atoi(v) : WIDE_BELOW; }();
#define PA_LAUNCH_INFLATE(W, S)                                                                                                        \
    hipLaunchKernelGGL((bgzf_inflate_kernel<6, W, S>), dim3(n_blocks), dim3(64), 0, stream, comp, comp_off, comp_len, out_off, out_len, out, \
                       status, debug_counts, comp_bytes, below)
    if (debug_counts) {
        if (wide) PA_LAUNCH_INFLATE(true, true);
        else PA_LAUNCH_INFLATE(false, true);
    } else {
        if (wide) PA_LAUNCH_INFLATE(true, false);
        else PA_LAUNCH_INFLATE(false, false);
    }
#undef PA_LAUNCH_INFLATE
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
        case INF_CRC: return "CRC32 of the inflated bytes differs from the member's trailer";
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
    // (a byte the kernel fails to write must not be one a previous call left there: this handle serves the tests)
    if (out_bytes) INF_HIP(hipMemsetAsync(h->d_out, 0xA5, (size_t)out_bytes, h->stream));
    unsigned long long* d_dbg = nullptr;
    if (getenv("PA_INFLATE_DEBUG")) {             // symbol statistics of the call on stderr (a diagnostic, not part of the ABI)
        INF_HIP(hipMalloc(reinterpret_cast<void**>(&d_dbg), 64));
        INF_HIP(hipMemsetAsync(d_dbg, 0, 64, h->stream));
    }
    INF_HIP(hipEventRecord(h->ev[0], h->stream));
    for (int r = 0; r < reps; ++r)
        pa::launch_bgzf_inflate(h->stream, static_cast<const uint8_t*>(h->d_comp), d_coff, d_clen, d_ooff, d_olen,
                                static_cast<uint8_t*>(h->d_out), d_status, n_blocks, comp_bytes, r == 0 ? d_dbg : nullptr);
    INF_HIP(hipGetLastError());
    INF_HIP(hipEventRecord(h->ev[1], h->stream));
    std::vector<int32_t> status(nb);
    INF_HIP(hipMemcpyAsync(status.data(), d_status, nb * 4, hipMemcpyDeviceToHost, h->stream));
    if (out_bytes) INF_HIP(hipMemcpyAsync(out, h->d_out, (size_t)out_bytes, hipMemcpyDeviceToHost, h->stream));
    INF_HIP(hipStreamSynchronize(h->stream));
    if (d_dbg) {
        unsigned long long c[6] = {0, 0, 0, 0, 0, 0};
        (void)hipMemcpy(c, d_dbg, sizeof c, hipMemcpyDeviceToHost);
        (void)hipFree(d_dbg);
        fprintf(stderr, "inflate: %d members, %llu bytes, %llu DEFLATE blocks, %llu window steps, %llu matches, %llu long-code symbols\n",
                n_blocks, c[5], c[3], c[0], c[1], c[2]);
    }
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
