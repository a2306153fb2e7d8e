// Variant summary encoder (include/pepper_amd_encoder.h): pileups of many regions -> candidate images, one launch set.
//
// Split of the reference's RegionalSummaryGenerator::generate_summary
// (/root/reference/pepper_variant/modules/cpp/region_summary.cpp:337-916):
//   GPU   segment_reads_kernel   one wave per read: prefix sums over its CIGAR operations, one 16-byte record per
//                                (read, 512-row tile) the read touches = where the walk of that tile starts
//         tile_offsets_kernel    (run twice around an exclusive scan: count per tile, then fill each tile's slice)
//         tile_count_kernel      ONE WORKGROUP OWNS ONE TILE: 28 int32 counters x 512 rows privatised in LDS (coverage,
//                                SNP / insert / delete counts, the 16 image columns the walk updates, the 8 SNP allele
//                                tallies); its waves walk the tile's records (encoder_common.h: one row per lane, LDS
//                                atomics only, :366-551); then the same workgroup applies the per-position pass of :568-654
//                                (fractions vs thresholds in fp64, passing-site records, clamp of columns 11..24) and
//                                stores the finished tile ONCE, coalesced: int32 [row][26], the algorithmic 104 bytes per
//                                position.  No global atomics on the matrix, no zero-fill pass, no read-modify-write pass
//                                [HBM bound: 2 B per aligned base in, 104 B per position out]
//         compact_votes_kernel   indel allele votes of the sites that passed (a few per cent of all votes)
//   host  what needs strings: ordered per-site maps of allele keys for the passing sites, candidate enumeration in the
//         reference's std::set order with its filters (:669-712)
//   GPU   gather_windows_kernel  33 x 26 window copy + candidate-specific overwrite (:828-905), int32 image_matrix and
//                                the int8 wrap DataStore.py:68 applies
// Round 2 walked one wave per read with global int32 atomics on a [L+1][32] matrix: 559 MB of atomic write traffic for
// 22 MB of algorithmic bytes on one 100 kb / 60x region, 759 waves on a 1024-SIMD chip.  Here parallelism = tiles
// (~200 per 100 kb region, x regions per batch) and every matrix byte is written exactly once.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/pepper_amd_realign.h"
#include "encoder_common.h"

using namespace pa_enc;

namespace {

constexpr int MAXC = 125;
constexpr int TP = 512;             // rows per tile = threads of a tile_count_kernel workgroup
constexpr int MATF = 26;            // int32 per matrix row in HBM (the image columns)
constexpr int NCNT = 28;            // privatised counters per row
constexpr int NW = 8, NT = 64 * NW;  // waves / threads of a tile_count_kernel workgroup: NT == TP, a thread owns a row in the per-position pass
constexpr int UNR = 8;              // rows per lane in flight in the row phase: 64 x UNR = TP
constexpr int VCAP = 512;           // indel votes a tile keeps in LDS (nanopore: ~600 per 512 rows x 60 reads; more spill to the global list)
constexpr int XQ = 128;             // per-wave queue of the bases that are not clean matches (processed 64 at a time)
static_assert(NT == TP && 64 * UNR == TP, "one thread per row, UNR rows per lane");
static_assert(VCAP == NT, "the flush takes one buffered vote per thread");
// counter slots: 0 coverage, 1 snp_count, 2 insert_count, 3 delete_count, 4 = column 4 (forward strand coverage),
// 5..11 = columns 8..14 (forward A C G T I D *), 12 = column 15 (reverse strand coverage), 13..19 = columns 19..25,
// 20..23 / 24..27 = forward / reverse tallies of mismatching A C G T (the SNP allele map of :409-425)
enum { K_COV = 0, K_SNP = 1, K_INS = 2, K_DEL = 3, K_FWD = 4, K_REV = 12, K_TAB = 20 };
// counters of a run
enum { CT_OVF = 0, CT_VOTES = 2, CT_ERR = 3, CT_POOL = 4, CT_N = 8 };
constexpr int POOL_SLOT = 64;       // bytes per pooled allele (an allele is at most 61 bytes, region_summary.cpp:455)
// per region, behind them: [2 r] passing sites, [2 r + 1] votes of passing sites -- both lists are written region by region
// (sites of region r from row_base, votes from vote_base), so the host never has to sort a batch's records by region

struct RegRec {                      // one region of the batch, as the kernels see it
    int64_t ref_off;                 // first byte of its reference in d_ref
    int64_t row_base;                // first row of its matrix (multiple of 16: line-aligned tile stores)
    int64_t seq_base;                // first base of its reads in d_seq / d_qual
    int32_t ref_len, L;              // L = region_end - region_start + 1; the matrix has L + 1 rows (row L stays zero)
    int32_t tile0, n_tiles;
    int32_t cand_lo, cand_hi;        // candidate_region_start / _end as rows (clamped into int32)
    int32_t qmin;                    // smallest integer base quality that is >= min_snp_baseq (256: none)
    int32_t vote_base;               // first slot of the region's slice of the passing-vote list (= its first CIGAR operation)
    double min_snp_q, min_indel_q, snp_thr, ins_thr, del_thr, min_cov;
};
struct TileRec { int32_t read, op, row, ri; };   // walk of `read` enters the tile at operation `op`, whose first row / read index are given
struct SiteRec { int32_t region, idx, cov, flags, fwd[4], rev[4]; };
// meta = type (1 insert, 2 delete) | reverse << 2 | from_ref << 3 | len << 4 | region << 10; prefix = the first 8 bytes of the allele,
// first byte in the top bits (compares like the string), filled for the votes of passing rows: the host orders alleles without touching
// the reads unless two of them agree on 8 bytes
struct Vote { uint32_t idx, meta; int64_t off; uint64_t prefix; };
struct CandDesc {
    int32_t idx, type;        // row of the candidate site; 1 SNP, 2 insert, 3 delete
    int32_t vcol, vval;       // columns 1/2/3 <- alt base code / allele length
    int32_t fwd, rev;         // strand allele depths (<= 125) for columns 5..7 / 16..18
    int32_t neg_f, neg_r;     // columns negated on the centre row (-1: none)
    int32_t last;             // delete: last spill row of the window (else -1)
    int32_t star_f, star_r;   // delete: '*' columns negated on spill rows
    int32_t region;
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
__device__ __forceinline__ int column_slot(int col) { return col < 15 ? col - 3 : col - 6; }   // columns 8..14 / 19..25
__host__ __device__ inline int base_code(char c) {
    switch (up(c)) {
        case 'A': return 1;
        case 'C': return 2;
        case 'G': return 3;
        case 'T': return 4;
        default: return 5;
    }
}

// ---- records: where the walk of each (read, tile) starts -----------------------------------------------------------
// Two passes of the same walk: FILL = false counts the records of every tile (atomics spread over the tiles), an exclusive
// scan turns the counts into offsets, FILL = true writes each record into its tile's slice.  (One pass appending to a
// global list cost 3 ms for 64 regions: ~300 k returning atomics on ONE counter serialise at ~10 ns each.)
template <bool FILL>
__global__ __launch_bounds__(256) void segment_reads_kernel(const ReadRec* __restrict__ reads, int n_reads,
                                                            const RegRec* __restrict__ regions,
                                                            const int32_t* __restrict__ cigar_op,
                                                            const int32_t* __restrict__ cigar_len, int* __restrict__ tile_count,
                                                            const int* __restrict__ tile_off, int* __restrict__ tile_fill,
                                                            TileRec* __restrict__ recs, int rec_cap) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= n_reads) return;
    const ReadRec rd = reads[r];
    if (!(rd.flags & READ_MAPQ_OK)) return;
    const RegRec* reg = regions + rd.region;
    const int L = reg->L, tile0 = reg->tile0;
    auto put = [&](int tile, int op, int row, int ri) {
        if (!FILL) {
            atomicAdd(&tile_count[tile], 1);
        } else {
            const int slot = tile_off[tile] + atomicAdd(&tile_fill[tile], 1);
            if (slot < rec_cap) recs[slot] = TileRec{r, op, row, ri};     // past the capacity: the host repeats the run with room
        }
    };
    int pos = rd.row0, ri = 0;
    if (pos > L - 1) return;
    if (pos >= 0 && lane == 0) {
        put(tile0 + pos / TP, rd.c0, pos, 0);
        // an insert in front of the read's first aligned base (S I M) is anchored on the row before it: when that row is the
        // last one of the previous tile, that tile walks the read's leading operations too
        if (pos > 0 && pos % TP == 0) put(tile0 + pos / TP - 1, rd.c0, pos, 0);
    }
    for (int cb = 0; cb < rd.ncig; cb += 64) {
        const int i = cb + lane;
        const bool valid = i < rd.ncig;
        const int op = valid ? cigar_op[rd.c0 + i] : OP_H;
        const int len = valid ? cigar_len[rd.c0 + i] : 0;
        const int radv = variant_ref_advance(op, len), qadv = variant_read_advance(op, len);
        const int rinc = wave_inclusive_sum(radv), qinc = FILL ? wave_inclusive_sum(qadv) : 0;
        const int first = pos + rinc - radv, after = pos + rinc;
        // the operation that holds row k * TP (the first row of tile k) opens the read's walk of that tile
        if (radv > 0 && first <= L - 1) {
            int lo = first > rd.row0 + 1 ? first : rd.row0 + 1;
            if (lo < 0) lo = 0;
            const int last_row = after - 1 < L - 1 ? after - 1 : L - 1;      // (an operation that ends before row 0 opens no tile)
            for (int k = (lo + TP - 1) / TP; k * TP <= last_row; ++k) put(tile0 + k, rd.c0 + i, first, ri + qinc - qadv);
        }
        pos += wave_total(rinc);
        if (FILL) ri += wave_total(qinc);
        if (pos > L - 1) break;
    }
}


// ---- packed form: clip + decode on the device -----------------------------------------------------------------------
// pa_bam_pack_regions (pepper_amd/csrc/bamio.cpp) ships every record as BAM stores it -- uint32 CIGAR words, 4-bit bases, one
// byte per quality -- once per batch; what the reference's get_reads does per (read, region) on the host (bam_handler.cpp:
// 176-303: walk the CIGAR, keep what lies inside [start, stop], decode the kept stretch of bases) is this kernel: one wave per
// (read, region) pair.  The walk has no running state here: 64 operations at a time, wave prefix sums give every operation its
// reference / read position, and what is kept of it follows from those plus one bit -- has an earlier operation kept an aligned
// base ("anchored": inserts, soft clips, deletions and skips are kept only behind one) [tests/bam_utils.py closed_form_clip is
// the same arithmetic in Python, checked against the sequential walk].  The kept bases are ONE stretch of the read
// (bamio.cpp, pa_bam_get_reads); it is decoded four bases per lane into the byte-per-base arrays tile_count_kernel reads.
struct PackedRead { int64_t data_off; int32_t pos, n_cigar, l_seq, flags; };     // = pa_packed_read
struct PairRec { int64_t s0; int32_t read, region, c0, cap; };                   // where the pair's clipped bases / operations go; cap > 0: room for that many bases only
static_assert(sizeof(PackedRead) == sizeof(pa_packed_read) && sizeof(PackedRead) == 24 && sizeof(PairRec) == 24, "packed tables");

struct UnpackArgs {
    const PairRec* pairs; int n_pairs;
    const PackedRead* preads; const RegRec* regions; const int64_t* region_start; const uint8_t* arena;
    ReadRec* reads; int32_t* cigar_op; int32_t* cigar_len; char* seq; uint8_t* qual;
    int* live;        // [n_regions] reads with a base inside | [n_regions] first inconsistent read + 1 | [n_regions + 1] first unsupported + 1
    int n_regions;
};

__global__ __launch_bounds__(256) void unpack_clip_kernel(UnpackArgs a) {
    const int pi = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (pi >= a.n_pairs) return;
    const PairRec pr = a.pairs[pi];
    const PackedRead rd = a.preads[pr.read];
    const int64_t start = a.region_start[pr.region], stop = start + a.regions[pr.region].L - 1;
    typedef uint32_t __attribute__((aligned(1))) word_any;        // (a slice left in place in an inflated span sits at any byte)
    const word_any* cig = reinterpret_cast<const word_any*>(a.arena + rd.data_off);
    // positions relative to the region's start, clamped far outside it: a read that begins 2^30 bases in front of the region
    // would need operations the check below refuses anyway
    int64_t rel64 = (int64_t)rd.pos - start;
    int rpos = (int)(rel64 < -(1 << 30) ? -(1 << 30) : rel64);      // (rd.pos < stop: the packer's region test)
    const int last = (int)(stop - start);                            // rows 0 .. last are inside
    int ridx = 0, n_out = 0, written = 0, pos_start = 0, first_idx = 0;
    bool anchored = false, unsupported = false;
    for (int cb = 0; cb < rd.n_cigar; cb += 64) {
        const int i = cb + lane;
        const uint32_t c = i < rd.n_cigar ? cig[i] : 5u;             // (H: nothing)
        const int op = (int)(c & 15u), len = (int)(c >> 4);
        const bool isM = op == OP_M || op == OP_EQ || op == OP_X, isIS = op == OP_I || op == OP_S, isDN = op == OP_D || op == OP_N;
        if (__ballot(len >= (1 << 24))) { unsupported = true; break; }          // (prefix sums are 32-bit: 64 x 2^24 fits)
        const int radv = (isM || isDN) ? len : 0, qadv = (isM || isIS) ? len : 0;
        const int rinc = wave_inclusive_sum(radv), qinc = wave_inclusive_sum(qadv);
        const int rb = rpos + rinc - radv, qb = ridx + qinc - qadv;
        const int lo = rb > 0 ? rb : 0, hi = rb + len - 1 < last ? rb + len - 1 : last;
        const int kept_m = (isM && hi >= lo) ? hi - lo + 1 : 0;
        const unsigned long long mm = __ballot(kept_m > 0);
        const int first = mm ? __ffsll((long long)mm) - 1 : 64;
        const bool anch = anchored || lane > first;
        const bool inside = rb >= 0 && rb <= last;
        int kept = kept_m;
        if (isIS) kept = (inside && anch) ? len : 0;
        if (isDN) kept = (inside && anch) ? (len < last - rb + 1 ? len : last - rb + 1) : 0;
        if (!anchored && mm) {
            pos_start = __builtin_amdgcn_readlane(lo, first);
            first_idx = __builtin_amdgcn_readlane(qb + (rb < 0 ? -rb : 0), first);
            anchored = true;
        }
        const unsigned long long km = __ballot(kept > 0);
        if (kept > 0) {
            const int slot = pr.c0 + n_out + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(km >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)km, 0u));
            a.cigar_op[slot] = op;
            a.cigar_len[slot] = kept;
        }
        n_out += __popcll(km);
        written += wave_sum((isM || isIS) ? kept : 0);
        rpos += wave_total(rinc);
        ridx += wave_total(qinc);
        if (rpos > last) break;
    }
    if (pr.cap > 0 && written > pr.cap) unsupported = true;      // (the polish chain gives a pair the room a region can fill)
    const bool bad = !unsupported && written > 0 && ((unsigned)first_idx + (unsigned)written > (unsigned)rd.l_seq);
    if (lane == 0) {
        if (unsupported) atomicMax(&a.live[a.n_regions + 1], pr.read + 1);
        if (bad) atomicMax(&a.live[a.n_regions], pr.read + 1);
    }
    if (unsupported || bad) written = 0;
    if (lane == 0) {
        ReadRec out;
        out.s0 = pr.s0;
        out.c0 = pr.c0;
        out.ncig = written > 0 ? n_out : 0;
        out.slen = written;
        out.row0 = pos_start;
        out.region = pr.region;
        out.flags = ((rd.flags & 0x10) ? READ_REV : 0) | ((written > 0 && ((rd.flags >> 16) & 0xff) > 0) ? READ_MAPQ_OK : 0);
        a.reads[pi] = out;
        if (written > 0) atomicAdd(&a.live[pr.region], 1);
    }
    if (written <= 0) return;
    // the kept stretch: bases first_idx .. first_idx + written - 1 of the record, four per lane and step
    const uint8_t* packed = a.arena + rd.data_off + 4ll * rd.n_cigar;
    const uint8_t* quals = packed + (rd.l_seq + 1) / 2;
    const unsigned long long lut_lo = 0x565352474d43413dull;                // "=ACMGRSV", first letter in the low byte
    const unsigned long long lut_hi = 0x4e42444b48595754ull;                // "TWYHKDBN"
    uint32_t* seq_out = reinterpret_cast<uint32_t*>(a.seq + pr.s0);         // (s0 is a multiple of 4)
    uint32_t* qual_out = reinterpret_cast<uint32_t*>(a.qual + pr.s0);
    auto letter = [&](unsigned code) -> unsigned {
        const unsigned long long t = (code & 8u) ? lut_hi : lut_lo;
        return (unsigned)(t >> ((code & 7u) * 8u)) & 0xffu;
    };
    for (int j = lane; 4 * j < written; j += 64) {
        const int bidx = first_idx + 4 * j;
        const uint8_t* src = packed + (bidx >> 1);
        const unsigned p0 = src[0], p1 = src[1], p2 = src[2];               // (past the record's bases: its qualities / the arena's padding)
        unsigned c0, c1, c2, c3;
        if (bidx & 1) { c0 = p0 & 15u; c1 = p1 >> 4; c2 = p1 & 15u; c3 = p2 >> 4; }
        else { c0 = p0 >> 4; c1 = p0 & 15u; c2 = p1 >> 4; c3 = p1 & 15u; }
        seq_out[j] = letter(c0) | (letter(c1) << 8) | (letter(c2) << 16) | (letter(c3) << 24);
        const uint8_t* q = quals + bidx;
        qual_out[j] = (unsigned)q[0] | ((unsigned)q[1] << 8) | ((unsigned)q[2] << 16) | ((unsigned)q[3] << 24);
    }
}

// exclusive scan of the per-tile record counts (one workgroup; tiles per batch: 12.5 k for 64 regions of 100 kb)
__global__ __launch_bounds__(1024) void tile_offsets_kernel(const int* __restrict__ tile_count, int n_tiles, int* __restrict__ tile_off) {
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

// ---- the tile kernel -----------------------------------------------------------------------------------------------
struct TileArgs {
    const ReadRec* reads; const RegRec* regions; const int32_t* tile_region;
    const int32_t* cigar_op; const int32_t* cigar_len; const char* seq; const uint8_t* qual; const char* ref;
    const TileRec* recs; const int* tile_off; int rec_cap;
    int* mat; uint8_t* pass; SiteRec* sites; Vote* votes; Vote* votes_out; int vote_cap; int4* ovf; int ovf_cap; int* counters;
    int* region_counts;
    int debug;        // PA_TILE_DEBUG (profiling only, tools/tile_ablation.sh): parts of the kernel switched off
};

__device__ __forceinline__ uint64_t allele_prefix(const char* bytes, uint32_t off, uint32_t len) {
    uint64_t p = 0;
    for (uint32_t k = 0; k < 8; ++k) p = (p << 8) | (k < len ? (uint64_t)(unsigned char)bytes[(size_t)off + k] : 0);
    return p;
}

// Append v to the global list `out` (counter `count`) from the lanes where `has`: one global atomic per wave.
__device__ __forceinline__ void wave_append_vote(bool has, const Vote& v, Vote* out, int* count, int cap, int lane) {
    const unsigned long long m = __ballot(has);
    if (!m) return;
    int base = 0;
    const int leader = __ffsll((long long)m) - 1;
    if (lane == leader) base = atomicAdd(count, __popcll(m));
    base = __builtin_amdgcn_readlane(base, leader);
    const int slot = base + __popcll(m & ((1ull << lane) - 1ull));
    if (has && slot < cap) out[slot] = v;
}

#ifndef PA_TILE_MIN_WAVES
#define PA_TILE_MIN_WAVES 4          // measured equal at 4 / 6 / 8 (2 / 3 / 4 workgroups per CU); 6 and 8 spill 7 / 19 registers to scratch
#endif
#ifdef PA_ENC_STAMP
__device__ unsigned long long g_enc_cycles[8];     // debug: shader-clock cycles per phase, summed over the waves
#define ENC_LAP(k) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); lap_acc[k] += t_ - lap_t; lap_t = t_; } while (0)
#else
#define ENC_LAP(k) do { } while (0)
#endif
__global__ __launch_bounds__(NT, PA_TILE_MIN_WAVES) void tile_count_kernel(TileArgs a) {
    __shared__ int cnt[NCNT * TP];                 // [counter][row]; reused as the finished [row][26] tile for the store
    __shared__ char ref_s[TP];
    __shared__ uint8_t refok_s[TP];               // is_acgt(reference base) per row
    __shared__ int4 s_ent[NW][64];                 // per wave: the reference-consuming operations of the batch that touch the span, in order
    __shared__ unsigned s_mask[NW][16];            // per wave: bit k = an operation's share of the span starts at row span_lo + k
    __shared__ uint2 vbuf[VCAP];                   // indel allele votes of this tile: only those of passing rows leave the CU
    __shared__ int vcount;
    // A base of a match run that equals an A/C/G/T reference base and passes the quality test -- 95 % of all bases -- adds one
    // to the coverage, to its strand's coverage (unless it anchors an insert / a deletion) and to its strand's column of that
    // letter.  Those three go in per RUN, as +1 / -1 at the run's ends in a difference array (two LDS atomics per operation
    // instead of three per base; the prefix sums are taken once, when the tile is complete); the row phase only looks at each
    // base (quality, letter, reference letter) and queues the ones that are NOT of that kind, which are then processed 64 at
    // a time with the full rules plus the undoing of what the run assumed for them.
    __shared__ int mcov[2][TP + 8];                // [strand][row]: match runs starting minus match runs ended before this row
    __shared__ int anch[TP];                       // runs whose last base sits here and anchors an insert / a deletion: forward | reverse << 16
    __shared__ unsigned xq[NW][XQ];                // queued bases: row | letter << 9 | quality ok << 17 | anchoring << 18 | reverse << 19
    __shared__ int wtot[2][NW];
    __shared__ int s_out[4];                       // passing sites / votes of the tile; their first slots in the region's lists
    uint8_t* pass_s = reinterpret_cast<uint8_t*>(&xq[0][0]);   // verdicts of the per-position pass: the queues are empty by then (two
                                                               // workgroups per CU need the tile's LDS below 80 KB)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int tile = blockIdx.x;
    const int region = a.tile_region[tile];
    const RegRec reg = a.regions[region];
    const int L = reg.L;
    const int tile_lo = (tile - reg.tile0) * TP, tile_hi = tile_lo + TP - 1;
    const int live_max = tile_hi + 1 < L - 1 ? tile_hi + 1 : L - 1;   // operations starting past this row are not this tile's (pos > end: not walked at all)
    for (int i = tid; i < NCNT * TP; i += NT) cnt[i] = 0;
    for (int i = tid; i < 2 * (TP + 8); i += NT) (&mcov[0][0])[i] = 0;
    if (tid < TP) anch[tid] = 0;
    if (tid < TP) {
        const int idx = tile_lo + tid;
        ref_s[tid] = idx < reg.ref_len ? a.ref[reg.ref_off + idx] : 'N';
        refok_s[tid] = is_acgt(ref_s[tid]) ? 1 : 0;
    }
    if (tid == 0) vcount = 0;
    if (tid < 4) s_out[tid] = 0;
    __syncthreads();

    const int rec0 = a.tile_off[tile], rec1 = (a.debug & 1) ? rec0 : (a.tile_off[tile + 1] < a.rec_cap ? a.tile_off[tile + 1] : a.rec_cap);
    int4* ent_s = s_ent[w];
    unsigned* mask_s = s_mask[w];
    // the wave's queue of bases that need the full rules: `qn` entries (wave-uniform), drained 64 at a time
    unsigned* xq_w = xq[w];
    int qn = 0;
    auto drain = [&](int count) {                  // entries [0, count) of the queue, one per lane
        const bool on = lane < count;
        const unsigned e = on ? xq_w[lane] : 0u;
        const int o = (int)(e & 511u);
        const unsigned b = (e >> 9) & 255u;
        const bool q_ok = on && ((e >> 17) & 1u), anchoring = (e >> 18) & 1u, erev = (e >> 19) & 1u;
        const unsigned rb = (unsigned char)ref_s[o];
        const unsigned ru = rb & 0xDFu, bu = b & 0xDFu;
        const unsigned ridx = (ru >> 1) & 3u, bidx = (bu >> 1) & 3u;
        const bool ref_ok = ((0x47544341u >> (ridx * 8)) & 0xFFu) == ru;            // is_acgt(reference base)
        const unsigned letter = (0x47544341u >> (bidx * 8)) & 0xFFu;
        const int acgt = (int)(bidx ^ (bidx >> 1));                                  // A C G T -> 0 1 2 3
        const int racgt = (int)(ridx ^ (ridx >> 1));
        const int sym = letter == bu ? acgt : (bu == 'I' ? 4 : (bu == 'D' ? 5 : 6));   // symbol_column's switch
        const bool mism = q_ok && rb != b;                                           // case-sensitive, as the reference compares
        const bool plain = letter == b;                                              // an upper-case A C G T: tallied on the device
        const int strand = erev ? K_REV : K_FWD;
        // what the rules give this base (:357-430) minus what its run assumed for it (coverage, strand coverage unless
        // anchoring, the reference letter's column where the reference base is a letter)
        atomicAdd(&cnt[K_COV * TP + o], on ? (q_ok ? 0 : -1) : 0);
        atomicAdd(&cnt[strand * TP + o], (on && !anchoring && !q_ok) ? -1 : 0);
        atomicAdd(&cnt[(strand + 1 + racgt) * TP + o], (on && ref_ok) ? -1 : 0);
        atomicAdd(&cnt[(strand + 1 + sym) * TP + o], (q_ok && ref_ok) ? 1 : 0);
        atomicAdd(&cnt[K_SNP * TP + o], mism ? 1 : 0);
        atomicAdd(&cnt[(K_TAB + (erev ? 4 : 0) + acgt) * TP + o], (mism && plain) ? 1 : 0);
        if (mism && !plain) {                        // rare alphabet (N, IUPAC, lower case): exact key kept on host
            const int slot = atomicAdd(&a.counters[CT_OVF], 1);
            if (slot < a.ovf_cap) a.ovf[slot] = make_int4(o + tile_lo, (int)b, erev ? 1 : 0, region);
        }
    };
    TileRec rec{0, 0, 0, 0};
    if (rec0 + w < rec1) rec = a.recs[rec0 + w];
    for (int k = rec0 + w; k < rec1; k += NW) {
        // three independent loads: the read's table entry, the first 64 operations (the arrays are padded: no bound needed
        // to issue them), and the next record of this wave
        const ReadRec rd_v = a.reads[rec.read];
        int op_ld = a.cigar_op[rec.op + lane], len_ld = a.cigar_len[rec.op + lane], next_ld = a.cigar_op[rec.op + lane + 1];
        TileRec rec_next = rec;
        if (k + NW < rec1) rec_next = a.recs[k + NW];
        // the record and the read's entry are the same in every lane: keep them in scalar registers
        struct { int64_t s0; int32_t c0, ncig, slen, flags; } rd;
        rd.s0 = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(rd_v.s0 >> 32)) << 32) |
                          (uint32_t)__builtin_amdgcn_readfirstlane((int)rd_v.s0));
        rd.c0 = __builtin_amdgcn_readfirstlane(rd_v.c0);
        rd.ncig = __builtin_amdgcn_readfirstlane(rd_v.ncig);
        rd.slen = __builtin_amdgcn_readfirstlane(rd_v.slen);
        rd.flags = __builtin_amdgcn_readfirstlane(rd_v.flags);
        rec.read = __builtin_amdgcn_readfirstlane(rec.read);
        rec.op = __builtin_amdgcn_readfirstlane(rec.op);
        rec.row = __builtin_amdgcn_readfirstlane(rec.row);
        rec.ri = __builtin_amdgcn_readfirstlane(rec.ri);
        const uint8_t* qual0 = a.qual + rd.s0;         // scalar base + 32-bit lane offset
        const char* seq0 = a.seq + rd.s0;
        const bool rev = rd.flags & READ_REV;
        const int c_end = rd.c0 + rd.ncig;
        int pos = rec.row, ri = rec.ri;
        ENC_LAP(0);                                   // record set-up (waits for the prefetched loads)
        for (int c = rec.op; c < c_end; c += 64) {
            const int i = c + lane;
            if (c != rec.op) {
                op_ld = a.cigar_op[i];
                len_ld = a.cigar_len[i];
                next_ld = a.cigar_op[i + 1];
            }
            const bool valid = i < c_end;
            const int op = valid ? op_ld : OP_H;
            const int len = valid ? len_ld : 0;
            const int next_op = i + 1 < c_end ? next_ld : -1;
            const int radv = variant_ref_advance(op, len), qadv = variant_read_advance(op, len);
            const int rinc = wave_inclusive_sum(radv), qinc = wave_inclusive_sum(qadv);
            const int first = pos + rinc - radv, rfirst = ri + qinc - qadv;
            const int total_r = wave_total(rinc);
            const bool live = valid && first <= live_max;
            const int anchor = first - 1;                                   // row an insert / a deletion is credited to
            const bool mine = live && anchor >= tile_lo && anchor <= tile_hi;   // (anchor <= L - 2 follows from first <= L - 1)
            const bool is_ins = mine && op == OP_I && rfirst - 1 >= 0 && !(a.debug & 8);
            const int n_ins = len + 1;
            const int span_lo = pos > tile_lo ? pos : tile_lo;
            int span_hi = pos + total_r - 1;
            if (span_hi > tile_hi) span_hi = tile_hi;
            if (span_hi > L - 1) span_hi = L - 1;

            // -- an insert's qualities (its anchor base and its first seven bases: nearly every insert is that short) as two
            //    unaligned dwords, asked for FIRST: they are back before the row phase's sixteen byte loads, which are asked for
            //    next and which the per-operation section would otherwise wait behind (loads return in order)
            typedef unsigned __attribute__((aligned(1))) u32u;
            const unsigned q_at = is_ins ? (unsigned)(rfirst - 1) : 0u;
            const unsigned qw0 = *reinterpret_cast<const u32u*>(qual0 + q_at), qw1 = *reinterpret_cast<const u32u*>(qual0 + q_at + 4);
            // -- scratch for the row phase.  Every row of the span belongs to the reference-consuming operation that covers it:
            //    those operations are compacted, in order, into {first row, read index there, last row, code | anchoring bit}
            //    entries, and each marks the row where its share of the span starts in a 512-bit mask.  A row's owner is then
            //    the number of marks at or before it (one broadcast read of the mask + mbcnt) instead of a six-step binary
            //    search over the first rows -- the kernel is bound by what it asks of the LDS (16 wave-instructions per 64 rows
            //    before this, each a round trip that four waves per SIMD have little to cover with).
            const int last = first + radv - 1;
            const bool cons = valid && radv > 0 && last >= span_lo && first <= span_hi;
            const unsigned long long cm = __ballot(cons);
            if (lane < 16) mask_s[lane] = 0u;
            __builtin_amdgcn_wave_barrier();
            if (cons) {
                const int cidx = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(cm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)cm, 0u));
                // bit 4: an insert or a deletion follows, so the last base of this match run anchors it (:381-391)
                ent_s[cidx] = make_int4(first, rfirst, last, op | ((next_op == OP_I || next_op == OP_D) ? 16 : 0));
                const int kk = (first > span_lo ? first : span_lo) - span_lo;
                atomicOr(&mask_s[kk >> 5], 1u << (kk & 31));
            }
            __builtin_amdgcn_wave_barrier();

            // -- one match run per lane: its clipped ends into the difference array of its strand, its anchoring last base
            {
                const bool match_op = valid && (op == OP_M || op == OP_EQ || op == OP_X) && len > 0;
                const int lo = first > span_lo ? first : span_lo;
                const int hi = last < span_hi ? last : span_hi;
                if (match_op && lo <= hi && !(a.debug & 64)) {
                    atomicAdd(&mcov[rev ? 1 : 0][lo - tile_lo], 1);
                    atomicAdd(&mcov[rev ? 1 : 0][hi + 1 - tile_lo], -1);
                    if ((next_op == OP_I || next_op == OP_D) && hi == last) atomicAdd(&anch[hi - tile_lo], rev ? 0x10000 : 1);
                }
            }

            // -- the row phase's loads first: one reference row per lane, UNR rows in flight; owners by binary search over the
            //    scratch, then every quality / base byte load of the (read, tile) is issued before anything waits on one
            bool is_m[UNR], is_d[UNR], anchored[UNR];
            int pl[UNR], qv[UNR];
            char bv[UNR];
            bool past = false;
            {
                unsigned si[UNR];
                int before = 0;                           // marks in the blocks in front of this one (wave-uniform)
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    const int p = span_lo + lane + 64 * u;
                    const bool act = p <= span_hi;
                    // a lane past the span keeps a row of its own (its add is a zero): equal addresses would serialise in the LDS
                    const int pc = act ? p : tile_lo + ((lane + 64 * u) & (TP - 1));
                    const unsigned m_lo = __builtin_amdgcn_readfirstlane(mask_s[2 * u]), m_hi = __builtin_amdgcn_readfirstlane(mask_s[2 * u + 1]);
                    const unsigned own = ((lane < 32 ? m_lo >> lane : m_hi >> (lane - 32)) & 1u);
                    int j = before + (int)__builtin_amdgcn_mbcnt_hi(m_hi, __builtin_amdgcn_mbcnt_lo(m_lo, 0u)) + (int)own - 1;
                    before += __popc(m_lo) + __popc(m_hi);
                    j = (act && j > 0) ? j : 0;
                    const int4 e = ent_s[j];
                    const int oj = e.w & 15;
                    const int rp = e.y + (pc - e.x);
                    const bool m = act && (oj == OP_M || oj == OP_EQ || oj == OP_X);
                    const bool inb = (unsigned)rp < (unsigned)rd.slen;
                    past |= m && !inb;
                    is_m[u] = m && inb;
                    is_d[u] = act && oj == OP_D;
                    anchored[u] = (e.w & 16) && pc == e.z;
                    pl[u] = pc - tile_lo;
                    si[u] = is_m[u] ? (unsigned)rp : 0u;       // (a lane without a base loads the read's first byte: no exec mask)
                }
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    qv[u] = (int)__builtin_nontemporal_load(qual0 + si[u]);
                    bv[u] = __builtin_nontemporal_load(seq0 + si[u]);
                }
            }
            if (past) atomicMax(&a.counters[CT_ERR], rec.read + 1);    // CIGAR runs past the sequence: reported by the host
            ENC_LAP(1);                               // scans, scratch, owner search, byte loads issued

            // -- one operation per lane: inserts (:431-490) and deletion anchors (:491-540)
            long long qsum = 0;
            if (is_ins) {
                // bytes 0 .. min(n_ins, bases left in the read, 8) - 1 of the two dwords
                int take = n_ins < rd.slen - (rfirst - 1) ? n_ins : rd.slen - (rfirst - 1);
                take = take < 8 ? (take > 0 ? take : 0) : 8;
                const unsigned m0 = take >= 4 ? 0xffffffffu : ((1u << (8 * take)) - 1u);
                const unsigned m1 = take >= 8 ? 0xffffffffu : (take > 4 ? ((1u << (8 * (take - 4))) - 1u) : 0u);
                qsum = (long long)__builtin_amdgcn_sad_u8(qw0 & m0, 0u, __builtin_amdgcn_sad_u8(qw1 & m1, 0u, 0u));
                if (n_ins <= 33)
                    for (int q = 8; q < n_ins; ++q) qsum += rfirst - 1 + q < rd.slen ? qual0[(unsigned)(rfirst - 1 + q)] : 0;
            }
            for (unsigned long long big = __ballot(is_ins && n_ins > 33); big; big &= big - 1) {   // long inserts: the wave sums
                const int src = __ffsll((long long)big) - 1;
                const int n = __builtin_amdgcn_readlane(n_ins, src), r0 = __builtin_amdgcn_readlane(rfirst, src) - 1;
                int part = 0;
                for (int q = lane; q < n; q += 64) part += r0 + q < rd.slen ? qual0[(unsigned)(r0 + q)] : 0;
                const int total = wave_sum(part);
                if (lane == src) qsum = total;
            }
            bool vote = false;
            unsigned vmeta = 0, voff = 0;
            if (is_ins) {
                const int avail = n_ins < rd.slen - (rfirst - 1) ? n_ins : (rd.slen - (rfirst - 1) > 0 ? rd.slen - (rfirst - 1) : 0);
                const bool passes = (double)qsum >= reg.min_indel_q * (double)n_ins;
                const int al = anchor - tile_lo;
                // (the reference reads the anchor base's quality without a bound; a CIGAR that ends past the sequence is an error here)
                const int anchor_q = rfirst - 1 < rd.slen ? (int)(qw0 & 255u) : 0;
                if (passes && (double)anchor_q < reg.min_snp_q) atomicAdd(&cnt[K_COV * TP + al], 1);
                if (avail + 1 <= 61 && passes) {
                    if (refok_s[al]) atomicAdd(&cnt[((rev ? K_REV : K_FWD) + 5) * TP + al], 1);   // column I
                    atomicAdd(&cnt[K_INS * TP + al], 1);
                    vote = true;
                    vmeta = 1u | (rev ? 4u : 0u) | ((unsigned)avail << 4) | ((unsigned)al << 10);
                    voff = (unsigned)(rd.s0 - reg.seq_base + rfirst - 1);
                }
            }
            if (mine && op == OP_D && !(a.debug & 8)) {
                const int al = anchor - tile_lo;
                if (refok_s[al]) atomicAdd(&cnt[((rev ? K_REV : K_FWD) + 6) * TP + al], 1);       // column D, no quality test
                int avail = len + 1 < reg.ref_len - anchor ? len + 1 : reg.ref_len - anchor;
                if (avail < 0) avail = 0;
                if (avail + 1 <= 61) {
                    atomicAdd(&cnt[K_DEL * TP + al], 1);
                    vote = true;
                    vmeta = 2u | (rev ? 4u : 0u) | 8u | ((unsigned)avail << 4) | ((unsigned)al << 10);
                    voff = (unsigned)anchor;
                }
            }
            {   // allele votes wait in LDS for the verdict on their row; a tile with more than VCAP of them spills the rest
                const unsigned long long m = __ballot(vote);
                if (m) {
                    int base = 0;
                    const int leader = __ffsll((long long)m) - 1;
                    if (lane == leader) base = atomicAdd(&vcount, __popcll(m));
                    base = __builtin_amdgcn_readlane(base, leader);
                    const int slot = base + __popcll(m & ((1ull << lane) - 1ull));
                    const bool spill = vote && slot >= VCAP;
                    if (vote && !spill) vbuf[slot] = make_uint2(voff, vmeta);
                    if (__ballot(spill))
                        wave_append_vote(spill, Vote{(uint32_t)(tile_lo + (int)(vmeta >> 10)), (vmeta & 1023u) | ((uint32_t)region << 10), (int64_t)voff, 0},
                                         a.votes, &a.counters[CT_VOTES], a.vote_cap, lane);
                }
            }

            ENC_LAP(2);                               // per-operation section (inserts, deletions, votes)
            // -- the rows: a base that is a clean match (quality passes, equal to an A/C/G/T reference base) is already counted
            //    by its run; the '*' column of a deletion's rows (:541-551) is one unconditional add; everything else is queued
            const int strand = rev ? K_REV : K_FWD;
            // Which of the lane's rows need the full rules, as a bit per row; the wave's exceptions of the whole pass then enter
            // the queue with ONE prefix sum (round 3 pushed per row slot: eight ballots, rank computations, queue-length updates
            // and drain tests per pass, nearly all of them taken -- 5 % of the bases are exceptions, three per 64 rows)
            unsigned om = 0, okm = 0;
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int o = pl[u];
                const unsigned rb = (unsigned char)ref_s[o], b = (unsigned char)bv[u];
                const bool ref_ok = refok_s[o] != 0;
                const bool q_ok = qv[u] >= reg.qmin;
                atomicAdd(&cnt[(strand + 7) * TP + o], (is_d[u] && ref_ok) ? 1 : 0);
                const bool other = is_m[u] && !(q_ok && ref_ok && rb == b) && !(a.debug & 32);
                om |= other ? 1u << u : 0u;
                okm |= q_ok ? 1u << u : 0u;
            }
            if (__ballot(om != 0u)) {                     // (wave-uniform)
                const int mine_n = __popc(om);
                const int inc = wave_inclusive_sum(mine_n);
                const int total = wave_total(inc);
                if (qn + total <= XQ) {
                    int slot = qn + inc - mine_n;
#pragma unroll
                    for (int u = 0; u < UNR; ++u)
                        if ((om >> u) & 1u) {
                            xq_w[slot++] = (unsigned)pl[u] | ((unsigned)(unsigned char)bv[u] << 9) | (((okm >> u) & 1u) << 17) |
                                           (anchored[u] ? 1u << 18 : 0u) | (rev ? 1u << 19 : 0u);
                        }
                    qn += total;
                    __builtin_amdgcn_wave_barrier();
                    while (qn >= 64) {
                        drain(64);
                        const unsigned moved = lane < qn - 64 ? xq_w[64 + lane] : 0u;
                        __builtin_amdgcn_wave_barrier();
                        if (lane < qn - 64) xq_w[lane] = moved;
                        qn -= 64;
                        __builtin_amdgcn_wave_barrier();
                    }
                } else {
                    // more exceptions in one pass than the queue holds (a read that disagrees with the reference nearly
                    // everywhere): row slot by row slot, draining as it fills
#pragma unroll
                    for (int u = 0; u < UNR; ++u) {
                        const bool other = (om >> u) & 1u;
                        const unsigned long long m = __ballot(other);
                        if (m) {
                            const int slot = qn + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                            if (other)
                                xq_w[slot] = (unsigned)pl[u] | ((unsigned)(unsigned char)bv[u] << 9) | (((okm >> u) & 1u) << 17) |
                                             (anchored[u] ? 1u << 18 : 0u) | (rev ? 1u << 19 : 0u);
                            qn += __popcll(m);
                            __builtin_amdgcn_wave_barrier();
                            if (qn >= 64) {
                                drain(64);
                                const unsigned moved = lane < qn - 64 ? xq_w[64 + lane] : 0u;
                                __builtin_amdgcn_wave_barrier();
                                if (lane < qn - 64) xq_w[lane] = moved;
                                qn -= 64;
                                __builtin_amdgcn_wave_barrier();
                            }
                        }
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            ENC_LAP(3);                               // row phase (first use of the loaded bytes: their latency lands here)
            pos += total_r;
            ri += wave_total(qinc);
            if (pos > live_max) break;
        }
        rec = rec_next;
    }
    if (qn > 0) drain(qn);
    ENC_LAP(4);
    __syncthreads();
    ENC_LAP(5);                                       // waiting for the slowest wave of the tile

    // ---- the runs' share: prefix sums of the difference arrays = match-run coverage per strand and row, minus the anchoring
    //      last bases for the strand columns; the reference letter's column where the reference base is a letter
    {
        const int dF = tid < TP ? mcov[0][tid] : 0, dR = tid < TP ? mcov[1][tid] : 0;
        int sF = wave_inclusive_sum(dF), sR = wave_inclusive_sum(dR);
        if (lane == 63) {
            wtot[0][w] = sF;
            wtot[1][w] = sR;
        }
        __syncthreads();
        for (int k = 0; k < w; ++k) {
            sF += wtot[0][k];
            sR += wtot[1][k];
        }
        if (tid < TP) {
            const int an = anch[tid];
            cnt[K_COV * TP + tid] += sF + sR;
            cnt[K_FWD * TP + tid] += sF - (an & 0xffff);
            cnt[K_REV * TP + tid] += sR - (an >> 16);
            if (refok_s[tid]) {
                const unsigned ru = (unsigned char)ref_s[tid] & 0xDFu, ridx = (ru >> 1) & 3u;
                const int racgt = (int)(ridx ^ (ridx >> 1));
                cnt[(K_FWD + 1 + racgt) * TP + tid] += sF;
                cnt[(K_REV + 1 + racgt) * TP + tid] += sR;
            }
        }
    }
    __syncthreads();

    // ---- the tile is complete: per-position pass of generate_summary (:568-654), then one coalesced store --------
    const int idx = tile_lo + tid;
    int vals[MATF];
#pragma unroll
    for (int f = 0; f < MATF; ++f) vals[f] = 0;
    int cov = 0, n_snp = 0, n_ins = 0, n_del = 0, tab[8];
    const bool row = tid < TP && idx < L;
    if (row) {
        cov = cnt[K_COV * TP + tid];
        n_snp = cnt[K_SNP * TP + tid];
        n_ins = cnt[K_INS * TP + tid];
        n_del = cnt[K_DEL * TP + tid];
#pragma unroll
        for (int t = 0; t < 8; ++t) tab[t] = cnt[(K_TAB + t) * TP + tid];
        vals[0] = base_code(ref_s[tid]);
        vals[4] = -cnt[K_FWD * TP + tid];
        vals[15] = -cnt[K_REV * TP + tid];
#pragma unroll
        for (int s = 0; s < 7; ++s) {
            vals[8 + s] = -cnt[(K_FWD + 1 + s) * TP + tid];
            vals[19 + s] = -cnt[(K_REV + 1 + s) * TP + tid];
        }
    }
    __syncthreads();                               // every counter is in registers: the LDS becomes the output tile
    if (tid < TP) pass_s[tid] = 0;
    bool passes = false;
    SiteRec site{};
    if (row && !(a.debug & 4)) {
        const double c = cov > 1 ? (double)cov : 1.0;
        const bool s = (double)n_snp / c >= reg.snp_thr;
        const bool n = (double)n_ins / c >= reg.ins_thr;
        const bool d = (double)n_del / c >= reg.del_thr;
        passes = (s || n || d) && idx >= reg.cand_lo && idx <= reg.cand_hi && (double)cov >= reg.min_cov;
        a.pass[reg.row_base + idx] = passes ? 1 : 0;
        pass_s[tid] = passes ? 1 : 0;
        if (passes) {
            site.region = region;
            site.idx = idx;
            site.cov = cov;
            site.flags = (s ? 1 : 0) | (n ? 2 : 0) | (d ? 4 : 0);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                site.fwd[t] = tab[t];
                site.rev[t] = tab[4 + t];
            }
        }
#pragma unroll
        for (int col = 11; col < 25; ++col) vals[col] = max(-MAXC, min(MAXC, vals[col]));
    }
    if (tid < TP) {
#pragma unroll
        for (int f = 0; f < MATF; ++f) cnt[tid * MATF + f] = vals[f];
    }
    __syncthreads();
    const int n_rows = L + 1 - tile_lo < TP ? L + 1 - tile_lo : TP;
    const int n_out = n_rows * MATF;
    int* dst = a.mat + (reg.row_base + tile_lo) * (int64_t)MATF;          // 128-byte aligned: row_base % 16 == 0, tile_lo % 512 == 0
    if (!(a.debug & 2)) {
        for (int i = tid; i < n_out / 4; i += NT) reinterpret_cast<int4*>(dst)[i] = reinterpret_cast<const int4*>(cnt)[i];
        for (int i = (n_out & ~3) + tid; i < n_out; i += NT) dst[i] = cnt[i];
    }
    // The sites that passed and the votes of their rows (a few per cent of each) go to the region's lists, each vote with the
    // first bytes of its allele.  ONE update of the region's two counters per TILE: round 3 updated the site counter once per
    // passing row and the vote counter once per wave, and with ~200 tiles of a region at work on the same two words those
    // returning atomics, served one after the other by the L2, were a sixth of the kernel (tools/tile_ablation.sh).
    const int nv = (a.debug & 128) ? 0 : (vcount < VCAP ? vcount : VCAP);       // (VCAP == NT: one vote per thread)
    const uint2 pv = tid < nv ? vbuf[tid] : make_uint2(0, 0);
    const bool has = tid < nv && pass_s[pv.y >> 10];
    Vote v{(uint32_t)(tile_lo + (int)(pv.y >> 10)), (pv.y & 1023u) | ((uint32_t)region << 10), (int64_t)pv.x, 0};
    if (has) v.prefix = allele_prefix((pv.y & 8u) ? a.ref + reg.ref_off : a.seq + reg.seq_base, pv.x, (pv.y >> 4) & 63u);
    const unsigned long long sm = __ballot(passes), vm = __ballot(has);
    int s_at = 0, v_at = 0;
    if (lane == 0) {
        if (sm) s_at = atomicAdd(&s_out[0], __popcll(sm));
        if (vm) v_at = atomicAdd(&s_out[1], __popcll(vm));
    }
    s_at = __builtin_amdgcn_readfirstlane(s_at);      // (with every lane active: lane 0's value)
    v_at = __builtin_amdgcn_readfirstlane(v_at);
    __syncthreads();
    if (tid == 0) {
        if (s_out[0]) s_out[2] = atomicAdd(&a.region_counts[2 * region], s_out[0]);       // < L: one per row at most
        if (s_out[1]) s_out[3] = atomicAdd(&a.region_counts[2 * region + 1], s_out[1]);
    }
    __syncthreads();
    const unsigned long long below = (1ull << lane) - 1ull;
    if (passes) a.sites[reg.row_base + s_out[2] + s_at + __popcll(sm & below)] = site;
    if (has) a.votes_out[reg.vote_base + s_out[3] + v_at + __popcll(vm & below)] = v;
#ifdef PA_ENC_STAMP
    ENC_LAP(6);                                       // flush: prefix sums, per-position pass, store, votes
    if (lane == 0)
        for (int k = 0; k < 8; ++k) atomicAdd(&g_enc_cycles[k], lap_acc[k]);
#endif
}

// votes of the sites that passed the thresholds, compacted for the host
// (the number of votes is only known on the device: a fixed grid strides over counters[CT_VOTES] of them)
__global__ __launch_bounds__(256) void compact_votes_kernel(const Vote* __restrict__ votes, const int* __restrict__ counters, int cap,
                                                            const RegRec* __restrict__ regions, const uint8_t* __restrict__ pass,
                                                            const char* __restrict__ seq, const char* __restrict__ ref,
                                                            int* __restrict__ region_counts, Vote* __restrict__ out) {
    const int n = counters[CT_VOTES] < cap ? counters[CT_VOTES] : cap;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        Vote v = votes[i];
        const int r = (int)(v.meta >> 10);
        if (pass[regions[r].row_base + v.idx]) {
            v.prefix = allele_prefix((v.meta & 8u) ? ref + regions[r].ref_off : seq + regions[r].seq_base, (uint32_t)v.off, (v.meta >> 4) & 63u);
            out[regions[r].vote_base + atomicAdd(&region_counts[2 * r + 1], 1)] = v;
        }
    }
}

// The two region-sliced lists made dense for one copy to the host: region r's sites / votes move to the prefix sums of the
// counts before it (one workgroup per region; the counts of a batch are a few hundred integers).
// An insert allele of more than 8 bytes (its first 8 travel in the vote's prefix) is copied into a pool slot here and the vote's
// `off` becomes the slot: the host orders and names alleles without the reads, which in the packed form (pa_encoder_stage_packed)
// it never holds decoded.
__global__ __launch_bounds__(256) void pack_results_kernel(const RegRec* __restrict__ regions, const int* __restrict__ region_counts,
                                                           const SiteRec* __restrict__ sites, const Vote* __restrict__ votes,
                                                           SiteRec* __restrict__ sites_out, Vote* __restrict__ votes_out,
                                                           const char* __restrict__ seq, char* __restrict__ pool, int pool_cap,
                                                           int* __restrict__ counters) {
    __shared__ int off[2];
    const int r = blockIdx.x;
    if (threadIdx.x < 2) off[threadIdx.x] = 0;
    __syncthreads();
    int s = 0, v = 0;
    for (int q = threadIdx.x; q < r; q += 256) {
        s += region_counts[2 * q];
        v += region_counts[2 * q + 1];
    }
    if (s) atomicAdd(&off[0], s);
    if (v) atomicAdd(&off[1], v);
    __syncthreads();
    const int ns = region_counts[2 * r], nv = region_counts[2 * r + 1];
    const SiteRec* src_s = sites + regions[r].row_base;
    const Vote* src_v = votes + regions[r].vote_base;
    for (int i = threadIdx.x; i < ns; i += 256) sites_out[off[0] + i] = src_s[i];
    const char* rseq = seq + regions[r].seq_base;
    for (int i = threadIdx.x; i < nv; i += 256) {
        Vote v = src_v[i];
        const uint32_t len = (v.meta >> 4) & 63u;
        if (!(v.meta & 8u) && len > 8) {
            const int slot = atomicAdd(&counters[CT_POOL], 1);
            if (slot < pool_cap)
                for (uint32_t k = 0; k < len; ++k) pool[(size_t)slot * POOL_SLOT + k] = rseq[(size_t)v.off + k];
            v.off = slot;
        }
        votes_out[off[1] + i] = v;
    }
}

// one 64-lane workgroup per candidate: 33 x 26 = 858 cells
__global__ __launch_bounds__(64) void gather_windows_kernel(const int* __restrict__ mat, const RegRec* __restrict__ regions,
                                                            const CandDesc* __restrict__ cands, int W, int F, int mid,
                                                            int* __restrict__ out32, int8_t* __restrict__ out8) {
    const CandDesc cd = cands[blockIdx.x];
    const int L = regions[cd.region].L;
    const int* m = mat + regions[cd.region].row_base * MATF;
    const size_t base = (size_t)blockIdx.x * W * F;
    for (int e = threadIdx.x; e < W * F; e += 64) {
        const int r = e / F, f = e - r * F;
        const int row = cd.idx - mid + r;
        int v = (row >= 0 && row <= L && f < MATF) ? m[(size_t)row * MATF + f] : 0;
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

struct Tally { int total = 0, fwd = 0, rev = 0; };

struct RegHost {
    pa_pileup p;
    pa_summary_params q;
    int L = 0;
    int64_t row_base = 0, seq_base = 0, op_base = 0, read_base = 0;
};

}  // namespace

namespace {

// CPUs this process may really use: the hardware's, cut by a cgroup quota (cgroup v2 cpu.max; the project's GPU boxes show 256
// logical CPUs and grant 16 -- 64 busy threads would each run at a quarter of the speed)
int usable_cpus() {
    int n = (int)std::thread::hardware_concurrency();
    if (n <= 0) n = 4;
    if (FILE* fh = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char quota[32] = {0};
        long long period = 0;
        if (fscanf(fh, "%31s %lld", quota, &period) == 2 && strcmp(quota, "max") != 0 && period > 0)
            n = std::min(n, std::max(1, (int)(atoll(quota) / period)));
        fclose(fh);
    }
    return n;
}

// worker threads that live as long as the encoder: the candidate enumeration of a batch is one short task per region
// (a few hundred microseconds of ordered maps and strings), and starting 16 threads per run cost more than the tasks
class RegionPool {
public:
    explicit RegionPool(int n) {
        for (int t = 0; t < n; ++t) threads_.emplace_back([this] { loop(); });
    }
    ~RegionPool() {
        {
            std::lock_guard<std::mutex> g(m_);
            stop_ = true;
        }
        cv_.notify_all();
        for (std::thread& t : threads_) t.join();
    }
    int size() const { return (int)threads_.size(); }
    // fn(i) for i in [0, n): on the workers and on the caller; returns when all are done
    void run(int n, const std::function<void(int)>& fn) {
        {
            std::lock_guard<std::mutex> g(m_);
            fn_ = &fn;
            n_ = n;
            next_.store(0);
            left_ = n;
            ++epoch_;
        }
        cv_.notify_all();
        drain();
        std::unique_lock<std::mutex> g(m_);
        done_.wait(g, [this] { return left_ == 0; });
        fn_ = nullptr;
    }

private:
    void drain() {
        for (;;) {
            const int i = next_.fetch_add(1);
            if (i >= n_) return;
            (*fn_)(i);
            std::lock_guard<std::mutex> g(m_);
            if (--left_ == 0) done_.notify_all();
        }
    }
    void loop() {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> g(m_);
                cv_.wait(g, [&] { return stop_ || epoch_ != seen; });
                if (stop_) return;
                seen = epoch_;
            }
            drain();
        }
    }
    std::vector<std::thread> threads_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    const std::function<void(int)>* fn_ = nullptr;
    std::atomic<int> next_{0};
    int n_ = 0, left_ = 0;
    uint64_t epoch_ = 0;
    bool stop_ = false;
};

}  // namespace

struct pa_variant_batch {
    std::vector<RegHost> regs;
    int64_t total_bases = 0, total_ops = 0, total_reads = 0, total_rows = 0, total_ref = 0;
    int n_tiles = 0, W = 33, F = 26, mid = 16;
    int rec_cap = 0, ovf_cap = 0, pool_cap = 0;
    bool staged = false, packed = false;
    DBuf d_seq, d_qual, d_ref, d_cig_op, d_cig_len, d_reads, d_regions, d_tile_region;
    // the tables of the staged batch as the kernels get them: the buffers above (pa_encoder_stage_batch) or slices of d_meta
    // (pa_encoder_stage_packed, one upload for all of them)
    const RegRec* p_regions = nullptr;
    const int32_t* p_tile_region = nullptr;
    const char* p_ref = nullptr;
    // packed form: the caller's arena as uploaded, the tables, what unpack_clip_kernel reports per region
    DBuf d_arena, d_meta, d_live, d_pool, d_comp, d_inf, d_walk;
    HBuf h_arena, h_meta, h_live, h_pool, h_comp, h_inf, h_walk;
    int64_t resident_bytes = 0;      // inflated bytes pa_encoder_inflate_bgzf left in d_arena (0: none)
    std::vector<int32_t> live;        // reads with a base inside each region (the reference's len(all_reads)) of the last run
    DBuf d_zero;                      // counters [CT_N] | per-region counts [2 n_regions] | tile_count [n_tiles] | tile_fill [n_tiles]: cleared per run
    DBuf d_tile_off, d_sorted, d_mat, d_pass, d_sites, d_votes, d_votes_out, d_sites_dense, d_votes_dense, d_ovf, d_cands, d_img32, d_img8;
    HBuf h_counts, h_sites, h_votes;
    std::unique_ptr<RegionPool> pool;
    int host_threads = 0;             // threads of the candidate enumeration: 0 = the default below, 1 = the calling thread alone
    // results of the last run
    int64_t n = 0;
    std::vector<int64_t> region_n;
    std::vector<int64_t> positions;
    std::vector<int32_t> depths, freqs;
    std::string names;
    double ms[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
};

void pa_variant_batch_free(pa_variant_batch* b) { delete b; }

namespace {

// candidates of one region in the reference's order (std::set<string> per site, region_summary.cpp:667-916)
struct RegionOut {
    std::vector<CandDesc> cands;
    std::vector<int64_t> positions;
    std::vector<int32_t> depths, freqs;
    std::string names;
};

// where the bytes of a vote's allele past the first eight live: deleted bases in the region's reference, inserted ones in the
// pool slot pack_results_kernel filled (the first eight are the vote's prefix)
struct AlleleSrc { const char* reference; const char* pool; };
inline const char* allele_tail(const Vote& v, const AlleleSrc& p) {
    return ((v.meta & 8u) ? p.reference + v.off : p.pool + (size_t)v.off * POOL_SLOT) + 8;
}
inline void append_allele(std::string& key, const Vote& v, const AlleleSrc& p) {
    const uint32_t len = (v.meta >> 4) & 63u;
    for (uint32_t k = 0; k < std::min(len, 8u); ++k) key.push_back((char)(v.prefix >> (56 - 8 * k)));
    if (len > 8) key.append(allele_tail(v, p), (size_t)len - 8);
}
// order of the votes of one region: site, then the allele key as std::map<std::string> orders "2..." / "3..." strings
// (type character, bytes as unsigned chars, the shorter of two that agree first).  The 8-byte prefix decides nearly always.
inline bool vote_less(const Vote& x, const Vote& y, const AlleleSrc& p) {
    if (x.idx != y.idx) return x.idx < y.idx;
    const uint32_t tx = x.meta & 3u, ty = y.meta & 3u;
    if (tx != ty) return tx < ty;
    if (x.prefix != y.prefix) return x.prefix < y.prefix;
    const uint32_t lx = (x.meta >> 4) & 63u, ly = (y.meta >> 4) & 63u;
    if (lx > 8 && ly > 8) {
        const int c = std::memcmp(allele_tail(x, p), allele_tail(y, p), std::min(lx, ly) - 8);
        if (c != 0) return c < 0;
    }
    return lx < ly;
}
inline bool same_allele(const Vote& x, const Vote& y, const AlleleSrc& p) {
    if (x.idx != y.idx || ((x.meta ^ y.meta) & 3u) || x.prefix != y.prefix) return false;
    const uint32_t lx = (x.meta >> 4) & 63u, ly = (y.meta >> 4) & 63u;
    return lx == ly && (lx <= 8 || std::memcmp(allele_tail(x, p), allele_tail(y, p), lx - 8) == 0);
}

void enumerate_region(const RegHost& rh, int region, int mid, const SiteRec* sites, size_t n_sites, const Vote* votes,
                      size_t n_votes, const int4* ovf, size_t n_ovf, const char* pool, RegionOut& out) {
    const pa_pileup& p = rh.p;
    const AlleleSrc src{p.reference, pool};
    const pa_summary_params& q = rh.q;
    auto refc = [&](int64_t idx) { return idx >= 0 && idx < p.reference_len ? p.reference[idx] : 'N'; };
    std::map<int32_t, std::map<char, Tally>> rare;       // SNP alleles outside ACGT
    for (size_t k = 0; k < n_ovf; ++k) {
        Tally& t = rare[ovf[k].x][(char)ovf[k].y];
        t.total += 1;
        (ovf[k].z ? t.rev : t.fwd) += 1;
    }
    out.cands.reserve(2 * n_sites + 16);
    out.positions.reserve(2 * n_sites + 16);
    out.depths.reserve(2 * n_sites + 16);
    out.freqs.reserve(2 * n_sites + 16);
    out.names.reserve(8 * n_sites + 64);
    size_t vk = 0;                                        // votes are sorted by site, like the sites
    for (size_t si = 0; si < n_sites; ++si) {
        const SiteRec& s = sites[si];
        const int depth = std::min(s.cov, MAXC);
        const char rb = refc(s.idx);
        auto accept = [&](char type, const Tally& t) {
            const double freq = (double)t.total / std::max(1.0, (double)depth);
            if ((double)t.total < q.candidate_support_threshold) return false;
            if (type != '1' && freq < q.indel_candidate_freq_threshold) return false;
            if (type == '1' && freq < q.snp_candidate_freq_threshold) return false;
            if (type != '1' && q.skip_indels) return false;
            if ((type == '1' && !(s.flags & 1)) || (type == '2' && !(s.flags & 2)) || (type == '3' && !(s.flags & 4)))
                return false;
            return true;
        };
        auto emit = [&](const std::string& key, const Tally& t, CandDesc d) {
            d.region = region;
            out.cands.push_back(d);
            out.positions.push_back(p.region_start + s.idx);
            out.depths.push_back(depth);
            out.freqs.push_back(std::min(t.total, MAXC));
            out.names += key;
            out.names.push_back('\0');
        };
        // SNP alleles: "1" + base, ordered by the raw base character (A C G T from the device's tallies are already in that
        // order; the rare alphabet, if any at this site, is merged in)
        struct SnpAllele { char base; Tally t; };
        SnpAllele snp[4 + 12];
        int n_snp = 0;
        const char acgt[4] = {'A', 'C', 'G', 'T'};
        for (int k = 0; k < 4; ++k)
            if (s.fwd[k] + s.rev[k] > 0) snp[n_snp++] = SnpAllele{acgt[k], Tally{s.fwd[k] + s.rev[k], s.fwd[k], s.rev[k]}};
        if (!rare.empty()) {
            const auto rit = rare.find(s.idx);
            if (rit != rare.end())
                for (const auto& kv : rit->second) {
                    int at = 0;
                    while (at < n_snp && snp[at].base != kv.first) ++at;
                    if (at == n_snp) {
                        if (n_snp == 16) continue;           // (more than 12 distinct non-ACGT read letters at one site: not a pileup)
                        snp[n_snp++] = SnpAllele{kv.first, Tally{}};
                    }
                    snp[at].t.total += kv.second.total;
                    snp[at].t.fwd += kv.second.fwd;
                    snp[at].t.rev += kv.second.rev;
                }
            std::sort(snp, snp + n_snp, [](const SnpAllele& x, const SnpAllele& y) { return x.base < y.base; });
        }
        for (int k = 0; k < n_snp; ++k) {
            if (!accept('1', snp[k].t)) continue;
            CandDesc d{};
            d.idx = s.idx; d.type = 1; d.vcol = 1; d.vval = base_code(snp[k].base);
            d.fwd = std::min(snp[k].t.fwd, MAXC); d.rev = std::min(snp[k].t.rev, MAXC);
            d.neg_f = symbol_column(rb, snp[k].base, false); d.neg_r = symbol_column(rb, snp[k].base, true);
            d.last = -1; d.star_f = d.star_r = -1;
            const char key[3] = {'1', snp[k].base, 0};
            emit(std::string(key, 2), snp[k].t, d);
        }
        while (vk < n_votes && (int32_t)votes[vk].idx < s.idx) ++vk;
        if (vk >= n_votes || (int32_t)votes[vk].idx != s.idx) continue;
        // allele keys "2" + anchor + inserted bases / "3" + deleted reference bases in std::map<std::string> order: the region's
        // votes arrive ordered by (site, type, allele bytes) -- see vote_less -- so equal keys are runs
        for (size_t k0 = vk; k0 < n_votes && (int32_t)votes[k0].idx == s.idx;) {
            size_t k1 = k0;
            Tally t;
            while (k1 < n_votes && same_allele(votes[k1], votes[k0], src)) {
                t.total += 1;
                ((votes[k1].meta & 4u) ? t.rev : t.fwd) += 1;
                ++k1;
            }
            const Vote& v = votes[k0];
            const char type = (v.meta & 3u) == 1u ? '2' : '3';
            const int alen = (int)((v.meta >> 4) & 63u);
            if (accept(type, t)) {
                CandDesc d{};
                d.idx = s.idx; d.type = type - '0';
                d.fwd = std::min(t.fwd, MAXC); d.rev = std::min(t.rev, MAXC);
                d.vval = std::min(alen, MAXC);
                d.star_f = d.star_r = -1;
                if (type == '2') {
                    d.vcol = 2; d.last = -1;
                    d.neg_f = symbol_column(rb, 'I', false); d.neg_r = symbol_column(rb, 'I', true);
                } else {
                    d.vcol = 3; d.last = std::min(mid + alen - 1, q.candidate_window_size - 1);
                    d.neg_f = symbol_column(rb, 'D', false); d.neg_r = symbol_column(rb, 'D', true);
                    d.star_f = symbol_column(rb, '*', false); d.star_r = symbol_column(rb, '*', true);
                }
                std::string key(1, type);
                append_allele(key, v, src);
                emit(key, t, d);
            }
            k0 = k1;
        }
        while (vk < n_votes && (int32_t)votes[vk].idx == s.idx) ++vk;
    }
}

int stage_batch(pa_encoder* e, int32_t n_regions, const pa_pileup* pileups, const pa_summary_params* params) {
    if (!e || n_regions < 0 || (n_regions > 0 && (!pileups || !params))) return pa::set_error(PA_ERR_INVALID, "null argument");
    if (n_regions >= (1 << 22)) return pa::set_error(PA_ERR_INVALID, "more than 4194303 regions in one batch");
    ENC_HIP(hipSetDevice(e->device));
    if (!e->variant) e->variant = new pa_variant_batch();
    pa_variant_batch& b = *e->variant;
    b.staged = false;
    b.ms[8] = b.ms[9] = 0;
    b.regs.assign((size_t)n_regions, RegHost());
    b.total_bases = b.total_ops = b.total_reads = b.total_rows = b.total_ref = 0;
    b.n_tiles = 0;
    std::vector<RegRec> regrecs((size_t)n_regions);
    for (int r = 0; r < n_regions; ++r) {
        const pa_pileup& p = pileups[r];
        const pa_summary_params& q = params[r];
        if (p.region_end < p.region_start || p.region_end - p.region_start > (int64_t)1 << 28) return pa::set_error(PA_ERR_INVALID, "bad region");
        if (q.feature_size < 26 || q.candidate_window_size < 2 || q.candidate_window_size > 254)
            return pa::set_error(PA_ERR_INVALID, "feature_size must be >= 26 and 2 <= candidate_window_size <= 254");
        if (r > 0 && (q.feature_size != params[0].feature_size || q.candidate_window_size != params[0].candidate_window_size))
            return pa::set_error(PA_ERR_INVALID, "one batch has one window size and one feature size");
        if (p.n_reads < 0 || p.reference_len < 0) return pa::set_error(PA_ERR_INVALID, "negative count");
        if (p.reference_len > 0x7fffffff || (p.n_reads > 0 && p.seq_offset[p.n_reads] > 0xffffffffll))
            return pa::set_error(PA_ERR_INVALID, "a region is limited to 2^32 read bases and 2^31 reference bases");
        RegHost& rh = b.regs[(size_t)r];
        rh.p = p;
        rh.q = q;
        rh.L = (int)(p.region_end - p.region_start + 1);
        rh.row_base = b.total_rows;
        rh.seq_base = b.total_bases;
        rh.op_base = b.total_ops;
        rh.read_base = b.total_reads;
        RegRec& g = regrecs[(size_t)r];
        g.ref_off = b.total_ref;
        g.row_base = rh.row_base;
        g.seq_base = rh.seq_base;
        g.ref_len = (int32_t)std::min<int64_t>(p.reference_len, 0x7fffffff);
        g.L = rh.L;
        g.tile0 = b.n_tiles;
        g.n_tiles = (rh.L + 1 + TP - 1) / TP;
        auto row_of = [&](int64_t pos) { return (int32_t)std::max<int64_t>(-2, std::min<int64_t>(pos - p.region_start, 0x7ffffff0)); };
        g.cand_lo = row_of(q.candidate_region_start);
        g.cand_hi = row_of(q.candidate_region_end);
        // integer base quality Q passes `(double)Q >= min_snp_baseq` iff Q >= qmin
        g.qmin = std::isnan(q.min_snp_baseq) ? 256 : (q.min_snp_baseq <= 0 ? 0 : (q.min_snp_baseq > 255 ? 256 : (int)std::ceil(q.min_snp_baseq)));
        g.vote_base = (int32_t)rh.op_base;
        g.min_snp_q = q.min_snp_baseq;
        g.min_indel_q = q.min_indel_baseq;
        g.snp_thr = q.snp_freq_threshold;
        g.ins_thr = q.insert_freq_threshold;
        g.del_thr = q.delete_freq_threshold;
        g.min_cov = q.min_coverage_threshold;
        b.total_rows += (rh.L + 1 + 15) & ~(int64_t)15;      // 16 rows x 104 B = 13 x 128 B: every tile store starts on a line
        b.total_ref += p.reference_len;
        b.total_reads += p.n_reads;
        b.total_bases += p.n_reads > 0 ? p.seq_offset[p.n_reads] : 0;
        b.total_ops += p.n_reads > 0 ? p.cigar_offset[p.n_reads] : 0;
        b.n_tiles += g.n_tiles;
        if (b.total_rows > ((int64_t)1 << 30) || b.total_ops > 0x7ffffff0 || b.total_reads > 0x7ffffff0)
            return pa::set_error(PA_ERR_INVALID, "batch too large: more than 2^30 rows or 2^31 CIGAR operations / reads");
    }
    b.W = n_regions ? params[0].candidate_window_size + 1 : 33;
    b.F = n_regions ? params[0].feature_size : 26;
    b.mid = n_regions ? params[0].candidate_window_size / 2 : 16;

    // read table
    std::vector<ReadRec> reads((size_t)b.total_reads);
    std::vector<int32_t> tile_region((size_t)b.n_tiles);
    for (int r = 0; r < n_regions; ++r) {
        const RegHost& rh = b.regs[(size_t)r];
        const pa_pileup& p = rh.p;
        for (int t = 0; t < regrecs[(size_t)r].n_tiles; ++t) tile_region[(size_t)(regrecs[(size_t)r].tile0 + t)] = r;
        for (int32_t k = 0; k < p.n_reads; ++k) {
            ReadRec& rd = reads[(size_t)(rh.read_base + k)];
            const int64_t slen = p.seq_offset[k + 1] - p.seq_offset[k], ncig = p.cigar_offset[k + 1] - p.cigar_offset[k];
            const int64_t row0 = p.read_pos[k] - p.region_start;
            if (slen < 0 || ncig < 0 || slen > 0x7ffffff0) return pa::set_error(PA_ERR_INVALID, "offsets of read " + std::to_string(k) + " are not ascending");
            rd.s0 = rh.seq_base + p.seq_offset[k];
            rd.c0 = (int32_t)(rh.op_base + p.cigar_offset[k]);
            rd.ncig = (int32_t)ncig;
            rd.slen = (int32_t)slen;
            // a read further than 2^30 rows from the region cannot reach it (CIGAR lengths are < 2^28 each, but their sum is
            // walked in int32): such a read is dropped here exactly as the walk would never touch a row
            rd.row0 = (int32_t)std::max<int64_t>(-(1 << 30), std::min<int64_t>(row0, 1 << 30));
            rd.region = r;
            rd.flags = (p.read_reverse[k] ? READ_REV : 0) | ((p.read_mapq[k] > 0 && row0 > -(1 << 30)) ? READ_MAPQ_OK : 0);
        }
    }

    hipStream_t st = e->stream;
    ENC_ALLOC(b.d_seq, (size_t)b.total_bases + 64);
    ENC_ALLOC(b.d_qual, (size_t)b.total_bases + 64);
    ENC_ALLOC(b.d_ref, (size_t)b.total_ref + 64);
    ENC_ALLOC(b.d_cig_op, (size_t)b.total_ops * 4 + 1024);      // the tile kernel issues 65-operation loads before it knows the read's end
    ENC_ALLOC(b.d_cig_len, (size_t)b.total_ops * 4 + 1024);
    ENC_ALLOC(b.d_reads, reads.size() * sizeof(ReadRec) + 64);
    ENC_ALLOC(b.d_regions, regrecs.size() * sizeof(RegRec) + 64);
    ENC_ALLOC(b.d_tile_region, tile_region.size() * 4 + 64);
    for (int r = 0; r < n_regions; ++r) {
        const RegHost& rh = b.regs[(size_t)r];
        const pa_pileup& p = rh.p;
        const int64_t nb = p.n_reads > 0 ? p.seq_offset[p.n_reads] : 0, no = p.n_reads > 0 ? p.cigar_offset[p.n_reads] : 0;
        if (nb > 0) {
            ENC_HIP(hipMemcpyAsync(b.d_seq.as<char>() + rh.seq_base, p.seq, (size_t)nb, hipMemcpyHostToDevice, st));
            ENC_HIP(hipMemcpyAsync(b.d_qual.as<char>() + rh.seq_base, p.qual, (size_t)nb, hipMemcpyHostToDevice, st));
        }
        if (no > 0) {
            ENC_HIP(hipMemcpyAsync(b.d_cig_op.as<int32_t>() + rh.op_base, p.cigar_op, (size_t)no * 4, hipMemcpyHostToDevice, st));
            ENC_HIP(hipMemcpyAsync(b.d_cig_len.as<int32_t>() + rh.op_base, p.cigar_len, (size_t)no * 4, hipMemcpyHostToDevice, st));
        }
        if (p.reference_len > 0)
            ENC_HIP(hipMemcpyAsync(b.d_ref.as<char>() + regrecs[(size_t)r].ref_off, p.reference, (size_t)p.reference_len, hipMemcpyHostToDevice, st));
    }
    if (!reads.empty()) ENC_HIP(hipMemcpyAsync(b.d_reads.p, reads.data(), reads.size() * sizeof(ReadRec), hipMemcpyHostToDevice, st));
    if (n_regions) {
        ENC_HIP(hipMemcpyAsync(b.d_regions.p, regrecs.data(), regrecs.size() * sizeof(RegRec), hipMemcpyHostToDevice, st));
        ENC_HIP(hipMemcpyAsync(b.d_tile_region.p, tile_region.data(), tile_region.size() * 4, hipMemcpyHostToDevice, st));
    }
    ENC_HIP(hipStreamSynchronize(st));            // the tables above are locals
    // records: a read enters a tile once per TP rows it spans, plus its first tile; reads with long deletions / skips span
    // more rows than they have bases, so the kernels count what they could not store and the run is repeated with room
    b.rec_cap = (int)std::min<int64_t>(0x7ffffff0, b.total_bases / TP + 2 * b.total_reads + 1024);
    b.ovf_cap = (int)std::min<int64_t>(0x7ffffff0, std::max<int64_t>(1 << 16, b.total_bases / 64));
    b.pool_cap = (int)std::min<int64_t>(0x7ffffff0, std::max<int64_t>(4096, b.total_ops / 64));
    b.p_regions = b.d_regions.as<RegRec>();
    b.p_tile_region = b.d_tile_region.as<int32_t>();
    b.p_ref = b.d_ref.as<char>();
    b.packed = false;
    b.live.assign((size_t)n_regions, 0);
    for (int r = 0; r < n_regions; ++r) b.live[(size_t)r] = pileups[r].n_reads;
    b.staged = true;
    return PA_OK;
}


// The packed form of a batch (include/pepper_amd_encoder.h): tables built in ONE page-locked block and uploaded with one copy,
// the arena with another, then unpack_clip_kernel -- nothing here waits for the device.
int stage_packed(pa_encoder* e, int32_t n_regions, const pa_packed_region* regions, const pa_summary_params* params,
                 const uint8_t* arena, int64_t arena_bytes, const pa_packed_read* reads, int32_t n_reads, const int32_t* pair_read,
                 const int32_t* region_pairs) {
    if (!e || n_regions < 0 || (n_regions > 0 && (!regions || !params || !region_pairs)) || arena_bytes < 0 || n_reads < 0 ||
        (n_reads > 0 && (!reads || !pair_read)))
        return pa::set_error(PA_ERR_INVALID, "null argument");
    if (n_regions >= (1 << 22)) return pa::set_error(PA_ERR_INVALID, "more than 4194303 regions in one batch");
    ENC_HIP(hipSetDevice(e->device));
    if (!e->variant) e->variant = new pa_variant_batch();
    pa_variant_batch& b = *e->variant;
    // arena == NULL: the bytes pa_encoder_inflate_bgzf left on the device
    const bool resident = arena == nullptr && n_reads > 0;
    if (resident && (b.resident_bytes <= 0 || arena_bytes > b.resident_bytes))
        return pa::set_error(PA_ERR_INVALID, "no arena given and no inflated span of that size resident on the device");
    b.staged = false;
    b.ms[8] = b.ms[9] = 0;
    b.regs.assign((size_t)n_regions, RegHost());
    b.total_bases = b.total_ops = b.total_reads = b.total_rows = b.total_ref = 0;
    b.n_tiles = 0;
    const int64_t n_pairs = n_regions ? region_pairs[n_regions] : 0;
    if (n_pairs < 0 || (n_regions && region_pairs[0] != 0)) return pa::set_error(PA_ERR_INVALID, "region_pairs must start at 0 and ascend");
    for (int r = 0; r < n_regions; ++r) {
        const pa_packed_region& p = regions[r];
        const pa_summary_params& q = params[r];
        if (p.region_end < p.region_start || p.region_end - p.region_start > (int64_t)1 << 28) return pa::set_error(PA_ERR_INVALID, "bad region");
        if (q.feature_size < 26 || q.candidate_window_size < 2 || q.candidate_window_size > 254)
            return pa::set_error(PA_ERR_INVALID, "feature_size must be >= 26 and 2 <= candidate_window_size <= 254");
        if (r > 0 && (q.feature_size != params[0].feature_size || q.candidate_window_size != params[0].candidate_window_size))
            return pa::set_error(PA_ERR_INVALID, "one batch has one window size and one feature size");
        if (p.reference_len < 0 || p.reference_len > 0x7fffffff || region_pairs[r + 1] < region_pairs[r])
            return pa::set_error(PA_ERR_INVALID, "negative count");
        b.total_ref += p.reference_len;
        b.total_rows += ((p.region_end - p.region_start + 1) + 1 + 15) & ~(int64_t)15;
        b.n_tiles += (int)((p.region_end - p.region_start + 1 + 1 + TP - 1) / TP);
        if (b.total_rows > ((int64_t)1 << 30)) return pa::set_error(PA_ERR_INVALID, "batch too large: more than 2^30 rows");
    }
    // one block: [RegRec x R][region_start x R][tile_region x tiles][PackedRead x reads][PairRec x pairs][reference bytes]
    auto up16 = [](size_t v) { return (v + 15) & ~(size_t)15; };
    const size_t o_reg = 0, o_start = up16(o_reg + (size_t)n_regions * sizeof(RegRec)), o_tile = up16(o_start + (size_t)n_regions * 8),
                 o_reads = up16(o_tile + (size_t)b.n_tiles * 4), o_pairs = up16(o_reads + (size_t)n_reads * sizeof(PackedRead)),
                 o_ref = up16(o_pairs + (size_t)n_pairs * sizeof(PairRec)), meta_bytes = up16(o_ref + (size_t)b.total_ref + 64);
    if (!b.h_meta.ensure(meta_bytes) || !b.h_live.ensure(((size_t)n_regions + 2) * 4)) return pa::set_error(PA_ERR_HIP, "hipHostMalloc failed");
    char* hm = b.h_meta.as<char>();
    RegRec* regrecs = reinterpret_cast<RegRec*>(hm + o_reg);
    int64_t* rstart = reinterpret_cast<int64_t*>(hm + o_start);
    int32_t* tile_region = reinterpret_cast<int32_t*>(hm + o_tile);
    PairRec* pairs = reinterpret_cast<PairRec*>(hm + o_pairs);
    if (n_reads) std::memcpy(hm + o_reads, reads, (size_t)n_reads * sizeof(PackedRead));
    b.total_rows = 0;
    b.total_ref = 0;
    b.n_tiles = 0;
    for (int r = 0; r < n_regions; ++r) {
        const pa_packed_region& p = regions[r];
        const pa_summary_params& q = params[r];
        RegHost& rh = b.regs[(size_t)r];
        rh.p = pa_pileup{};
        rh.p.region_start = p.region_start;
        rh.p.region_end = p.region_end;
        rh.p.reference = p.reference;
        rh.p.reference_len = p.reference_len;
        rh.p.n_reads = region_pairs[r + 1] - region_pairs[r];
        rh.q = q;
        rh.L = (int)(p.region_end - p.region_start + 1);
        rh.row_base = b.total_rows;
        rh.seq_base = b.total_bases;
        rh.op_base = b.total_ops;
        rh.read_base = region_pairs[r];
        RegRec& g = regrecs[r];
        g.ref_off = b.total_ref;
        g.row_base = rh.row_base;
        g.seq_base = rh.seq_base;
        g.ref_len = (int32_t)p.reference_len;
        g.L = rh.L;
        g.tile0 = b.n_tiles;
        g.n_tiles = (rh.L + 1 + TP - 1) / TP;
        auto row_of = [&](int64_t pos) { return (int32_t)std::max<int64_t>(-2, std::min<int64_t>(pos - p.region_start, 0x7ffffff0)); };
        g.cand_lo = row_of(q.candidate_region_start);
        g.cand_hi = row_of(q.candidate_region_end);
        g.qmin = std::isnan(q.min_snp_baseq) ? 256 : (q.min_snp_baseq <= 0 ? 0 : (q.min_snp_baseq > 255 ? 256 : (int)std::ceil(q.min_snp_baseq)));
        g.vote_base = (int32_t)rh.op_base;
        g.min_snp_q = q.min_snp_baseq;
        g.min_indel_q = q.min_indel_baseq;
        g.snp_thr = q.snp_freq_threshold;
        g.ins_thr = q.insert_freq_threshold;
        g.del_thr = q.delete_freq_threshold;
        g.min_cov = q.min_coverage_threshold;
        rstart[r] = p.region_start;
        for (int t = 0; t < g.n_tiles; ++t) tile_region[g.tile0 + t] = r;
        if (p.reference_len > 0) std::memcpy(hm + o_ref + b.total_ref, p.reference, (size_t)p.reference_len);
        // where each pair's clipped bases and operations go: room for the whole read (what is kept is known on the device only)
        for (int32_t k = region_pairs[r]; k < region_pairs[r + 1]; ++k) {
            const int32_t ri = pair_read[k];
            if (ri < 0 || ri >= n_reads) return pa::set_error(PA_ERR_INVALID, "pair_read out of range");
            const pa_packed_read& rd = reads[ri];
            if (rd.n_cigar < 0 || rd.l_seq < 0 || rd.data_off < 0 ||
                rd.data_off + 4ll * rd.n_cigar + (rd.l_seq + 1) / 2 + rd.l_seq > arena_bytes)
                return pa::set_error(PA_ERR_INVALID, "packed read " + std::to_string(ri) + " lies outside the arena");
            pairs[k] = PairRec{b.total_bases, ri, r, (int32_t)b.total_ops, 0};
            b.total_bases += ((int64_t)rd.l_seq + 3 & ~(int64_t)3) + 4;
            b.total_ops += rd.n_cigar;
            if (b.total_ops > 0x7ffffff0) return pa::set_error(PA_ERR_INVALID, "batch too large: more than 2^31 CIGAR operations");
        }
        if (b.total_bases - rh.seq_base > 0xffffff00ll) return pa::set_error(PA_ERR_INVALID, "a region is limited to 2^32 read bases");
        b.total_rows += (rh.L + 1 + 15) & ~(int64_t)15;
        b.total_ref += p.reference_len;
        b.n_tiles += g.n_tiles;
    }
    b.total_reads = n_pairs;
    b.W = n_regions ? params[0].candidate_window_size + 1 : 33;
    b.F = n_regions ? params[0].feature_size : 26;
    b.mid = n_regions ? params[0].candidate_window_size / 2 : 16;

    hipStream_t st = e->stream;
    if (!resident) {
        ENC_ALLOC(b.d_arena, (size_t)arena_bytes + 256);
        b.resident_bytes = 0;
    }
    ENC_ALLOC(b.d_meta, meta_bytes);
    ENC_ALLOC(b.d_live, ((size_t)n_regions + 2) * 4);
    ENC_ALLOC(b.d_seq, (size_t)b.total_bases + 64);
    ENC_ALLOC(b.d_qual, (size_t)b.total_bases + 64);
    ENC_ALLOC(b.d_cig_op, (size_t)b.total_ops * 4 + 1024);
    ENC_ALLOC(b.d_cig_len, (size_t)b.total_ops * 4 + 1024);
    ENC_ALLOC(b.d_reads, (size_t)n_pairs * sizeof(ReadRec) + 64);
    ENC_HIP(hipEventRecord(e->ev[6], st));
    if (arena_bytes > 0 && !resident) ENC_HIP(hipMemcpyAsync(b.d_arena.p, arena, (size_t)arena_bytes, hipMemcpyHostToDevice, st));
    ENC_HIP(hipMemcpyAsync(b.d_meta.p, hm, meta_bytes, hipMemcpyHostToDevice, st));
    ENC_HIP(hipMemsetAsync(b.d_live.p, 0, ((size_t)n_regions + 2) * 4, st));
    ENC_HIP(hipEventRecord(e->ev[7], st));
    const char* dm = b.d_meta.as<char>();
    b.p_regions = reinterpret_cast<const RegRec*>(dm + o_reg);
    b.p_tile_region = reinterpret_cast<const int32_t*>(dm + o_tile);
    b.p_ref = dm + o_ref;
    if (n_pairs > 0) {
        UnpackArgs ua;
        ua.pairs = reinterpret_cast<const PairRec*>(dm + o_pairs);
        ua.n_pairs = (int)n_pairs;
        ua.preads = reinterpret_cast<const PackedRead*>(dm + o_reads);
        ua.regions = b.p_regions;
        ua.region_start = reinterpret_cast<const int64_t*>(dm + o_start);
        ua.arena = b.d_arena.as<uint8_t>();
        ua.reads = b.d_reads.as<ReadRec>();
        ua.cigar_op = b.d_cig_op.as<int32_t>();
        ua.cigar_len = b.d_cig_len.as<int32_t>();
        ua.seq = b.d_seq.as<char>();
        ua.qual = b.d_qual.as<uint8_t>();
        ua.live = b.d_live.as<int>();
        ua.n_regions = n_regions;
        hipLaunchKernelGGL(unpack_clip_kernel, dim3((unsigned)((n_pairs + 3) / 4)), dim3(256), 0, st, ua);
        ENC_HIP(hipGetLastError());
    }
    ENC_HIP(hipEventRecord(e->ev[8], st));
    ENC_HIP(hipMemcpyAsync(b.h_live.p, b.d_live.p, ((size_t)n_regions + 2) * 4, hipMemcpyDeviceToHost, st));
    b.rec_cap = (int)std::min<int64_t>(0x7ffffff0, b.total_bases / TP + 2 * b.total_reads + 1024);
    b.ovf_cap = (int)std::min<int64_t>(0x7ffffff0, std::max<int64_t>(1 << 16, b.total_bases / 64));
    b.pool_cap = (int)std::min<int64_t>(0x7ffffff0, std::max<int64_t>(4096, b.total_ops / 64));
    b.live.assign((size_t)n_regions, 0);
    b.packed = true;
    b.staged = true;
    return PA_OK;
}

int run_staged(pa_encoder* e, int64_t* n_candidates) {
    if (!e || !e->variant || !e->variant->staged) return pa::set_error(PA_ERR_INVALID, "no staged batch");
    ENC_HIP(hipSetDevice(e->device));
    pa_variant_batch& b = *e->variant;
    hipStream_t st = e->stream;
    const int n_regions = (int)b.regs.size();
    b.n = 0;
    b.region_n.assign((size_t)n_regions, 0);
    b.positions.clear();
    b.depths.clear();
    b.freqs.clear();
    b.names.clear();
    for (int k = 0; k < 8; ++k) b.ms[k] = 0;          // ([8], [9] belong to the staging of this batch)
    if (n_regions == 0) return PA_OK;
    const auto t_begin = std::chrono::steady_clock::now();
    const int vote_cap = (int)std::min<int64_t>(b.total_ops + 1, 0x7ffffff0);
    const size_t n_zero = (size_t)CT_N + 2 * (size_t)n_regions + 2 * (size_t)b.n_tiles;
    ENC_ALLOC(b.d_zero, n_zero * 4);
    ENC_ALLOC(b.d_tile_off, ((size_t)b.n_tiles + 1) * 4);
    ENC_ALLOC(b.d_mat, (size_t)b.total_rows * MATF * 4 + 64);
    ENC_ALLOC(b.d_pass, (size_t)b.total_rows + 64);
    ENC_ALLOC(b.d_sites, (size_t)b.total_rows * sizeof(SiteRec));
    ENC_ALLOC(b.d_votes, (size_t)vote_cap * sizeof(Vote));
    ENC_ALLOC(b.d_votes_out, (size_t)vote_cap * sizeof(Vote));
    ENC_ALLOC(b.d_sites_dense, (size_t)b.total_rows * sizeof(SiteRec));
    ENC_ALLOC(b.d_votes_dense, (size_t)vote_cap * sizeof(Vote));
    if (!b.h_counts.ensure(((size_t)CT_N + 2 * (size_t)n_regions + 1) * 4)) return pa::set_error(PA_ERR_HIP, "hipHostMalloc failed");
    int* host_counters = b.h_counts.as<int>();                  // [CT_N] | per-region counts | records
    const int* host_rc = host_counters + CT_N;
    for (int attempt = 0;; ++attempt) {
        ENC_ALLOC(b.d_sorted, (size_t)b.rec_cap * sizeof(TileRec));
        ENC_ALLOC(b.d_ovf, (size_t)b.ovf_cap * sizeof(int4));
        ENC_ALLOC(b.d_pool, (size_t)b.pool_cap * POOL_SLOT);
        int* counters = b.d_zero.as<int>();
        int* region_counts = counters + CT_N;
        int* tile_count = region_counts + 2 * n_regions;
        int* tile_fill = tile_count + b.n_tiles;
        ENC_HIP(hipMemsetAsync(b.d_zero.p, 0, n_zero * 4, st));
        ENC_HIP(hipEventRecord(e->ev[0], st));
        const dim3 seg_grid((unsigned)((b.total_reads + 3) / 4));
        if (b.total_reads > 0)
            hipLaunchKernelGGL(segment_reads_kernel<false>, seg_grid, dim3(256), 0, st, b.d_reads.as<ReadRec>(), (int)b.total_reads,
                               b.p_regions, b.d_cig_op.as<int32_t>(), b.d_cig_len.as<int32_t>(), tile_count,
                               (const int*)nullptr, (int*)nullptr, (TileRec*)nullptr, 0);
        hipLaunchKernelGGL(tile_offsets_kernel, dim3(1), dim3(1024), 0, st, tile_count, b.n_tiles, b.d_tile_off.as<int>());
        if (b.total_reads > 0)
            hipLaunchKernelGGL(segment_reads_kernel<true>, seg_grid, dim3(256), 0, st, b.d_reads.as<ReadRec>(), (int)b.total_reads,
                               b.p_regions, b.d_cig_op.as<int32_t>(), b.d_cig_len.as<int32_t>(), tile_count,
                               b.d_tile_off.as<int>(), tile_fill, b.d_sorted.as<TileRec>(), b.rec_cap);
        ENC_HIP(hipEventRecord(e->ev[1], st));
        TileArgs ta;
        ta.reads = b.d_reads.as<ReadRec>();
        ta.regions = b.p_regions;
        ta.tile_region = b.p_tile_region;
        ta.cigar_op = b.d_cig_op.as<int32_t>();
        ta.cigar_len = b.d_cig_len.as<int32_t>();
        ta.seq = b.d_seq.as<char>();
        ta.qual = b.d_qual.as<uint8_t>();
        ta.ref = b.p_ref;
        ta.recs = b.d_sorted.as<TileRec>();
        ta.tile_off = b.d_tile_off.as<int>();
        ta.rec_cap = b.rec_cap;
        ta.mat = b.d_mat.as<int>();
        ta.pass = b.d_pass.as<uint8_t>();
        ta.sites = b.d_sites.as<SiteRec>();
        ta.votes = b.d_votes.as<Vote>();
        ta.votes_out = b.d_votes_out.as<Vote>();
        ta.vote_cap = vote_cap;
        ta.ovf = b.d_ovf.as<int4>();
        ta.ovf_cap = b.ovf_cap;
        ta.counters = counters;
        ta.region_counts = region_counts;
        static const int tile_debug = [] { const char* e = getenv("PA_TILE_DEBUG"); return e ? atoi(e) : 0; }();
        ta.debug = tile_debug;
        hipLaunchKernelGGL(tile_count_kernel, dim3((unsigned)b.n_tiles), dim3(NT), 0, st, ta);
        ENC_HIP(hipEventRecord(e->ev[2], st));
        hipLaunchKernelGGL(compact_votes_kernel, dim3((unsigned)std::min(2048, (vote_cap + 255) / 256)), dim3(256), 0, st, b.d_votes.as<Vote>(), counters,
                           vote_cap, b.p_regions, b.d_pass.as<uint8_t>(), b.d_seq.as<char>(), b.p_ref, region_counts,
                           b.d_votes_out.as<Vote>());
        hipLaunchKernelGGL(pack_results_kernel, dim3((unsigned)n_regions), dim3(256), 0, st, b.p_regions, region_counts,
                           b.d_sites.as<SiteRec>(), b.d_votes_out.as<Vote>(), b.d_sites_dense.as<SiteRec>(), b.d_votes_dense.as<Vote>(),
                           b.d_seq.as<char>(), b.d_pool.as<char>(), b.pool_cap, counters);
        ENC_HIP(hipEventRecord(e->ev[3], st));
        ENC_HIP(hipGetLastError());
        ENC_HIP(hipMemcpyAsync(host_counters, counters, ((size_t)CT_N + 2 * (size_t)n_regions) * 4, hipMemcpyDeviceToHost, st));
        ENC_HIP(hipMemcpyAsync(host_counters + CT_N + 2 * n_regions, b.d_tile_off.as<int>() + b.n_tiles, sizeof(int), hipMemcpyDeviceToHost, st));
        ENC_HIP(hipStreamSynchronize(st));
        const int n_recs = host_counters[CT_N + 2 * n_regions];
        if (n_recs > b.rec_cap || host_counters[CT_OVF] > b.ovf_cap || host_counters[CT_POOL] > b.pool_cap) {
            if (attempt >= 2) return pa::set_error(PA_ERR_HIP, "encoder record buffers could not be sized");
            b.rec_cap = std::max(b.rec_cap, n_recs + 1024);
            b.ovf_cap = std::max(b.ovf_cap, host_counters[CT_OVF] + 1024);
            b.pool_cap = std::max(b.pool_cap, host_counters[CT_POOL] + 1024);
            continue;
        }
        break;
    }
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e->ev[0], e->ev[1]); b.ms[0] = ms;       // records: count pass + offsets + fill pass
    (void)hipEventElapsedTime(&ms, e->ev[1], e->ev[2]); b.ms[1] = ms;       // tile_count_kernel
    (void)hipEventElapsedTime(&ms, e->ev[2], e->ev[3]); b.ms[2] = ms;       // compact_votes_kernel + pack_results_kernel
    if (b.packed && b.ms[8] == 0 && b.ms[9] == 0) {
        (void)hipEventElapsedTime(&ms, e->ev[6], e->ev[7]); b.ms[8] = ms;   // upload of the arena and the tables
        (void)hipEventElapsedTime(&ms, e->ev[7], e->ev[8]); b.ms[9] = ms;   // unpack_clip_kernel
    }
    if (b.packed) {               // what unpack_clip_kernel reported while the batch was staged (the copy is long done)
        const int* hl = b.h_live.as<int>();
        if (hl[n_regions] > 0)
            return pa::set_error(PA_ERR_INVALID, "packed read " + std::to_string(hl[n_regions] - 1) + ": its CIGAR walks over more bases than the record holds");
        if (hl[n_regions + 1] > 0)
            return pa::set_error(PA_ERR_UNSUPPORTED, "packed read " + std::to_string(hl[n_regions + 1] - 1) +
                                                         ": a CIGAR operation of 2^24 bases or more (take the host-clipped form for this batch)");
        for (int r = 0; r < n_regions; ++r) b.live[(size_t)r] = hl[r];
    }
    if (host_counters[CT_ERR] > 0) {
        const int64_t g = host_counters[CT_ERR] - 1;
        size_t r = 0;
        while (r + 1 < b.regs.size() && b.regs[r + 1].read_base <= g) ++r;
        return pa::set_error(PA_ERR_INVALID, "CIGAR of read " + std::to_string(g - b.regs[r].read_base) + (n_regions > 1 ? " of region " + std::to_string(r) : "") +
                                                 " runs past its sequence");
    }
    if (host_counters[CT_VOTES] > vote_cap) return pa::set_error(PA_ERR_INVALID, "more indel votes than CIGAR operations (corrupt pileup)");
    // region r's sites / votes are [s0[r], s0[r + 1]) / [v0[r], v0[r + 1]) of the dense lists
    std::vector<size_t> s0((size_t)n_regions + 1, 0), v0((size_t)n_regions + 1, 0), o0((size_t)n_regions + 1, 0);
    for (int r = 0; r < n_regions; ++r) {
        s0[(size_t)r + 1] = s0[(size_t)r] + (size_t)host_rc[2 * r];
        v0[(size_t)r + 1] = v0[(size_t)r] + (size_t)host_rc[2 * r + 1];
    }
    const size_t n_sites = s0[(size_t)n_regions], n_votes = v0[(size_t)n_regions];
    const int n_ovf = host_counters[CT_OVF], n_pool = host_counters[CT_POOL];
    if (!b.h_sites.ensure(n_sites * sizeof(SiteRec) + 64) || !b.h_votes.ensure(n_votes * sizeof(Vote) + 64) ||
        !b.h_pool.ensure((size_t)n_pool * POOL_SLOT + 64))
        return pa::set_error(PA_ERR_HIP, "hipHostMalloc failed");
    const char* pool = b.h_pool.as<char>();
    SiteRec* sites = b.h_sites.as<SiteRec>();
    Vote* votes = b.h_votes.as<Vote>();
    std::vector<int4> ovf((size_t)n_ovf);
    if (n_sites) ENC_HIP(hipMemcpyAsync(sites, b.d_sites_dense.p, n_sites * sizeof(SiteRec), hipMemcpyDeviceToHost, st));
    if (n_votes) ENC_HIP(hipMemcpyAsync(votes, b.d_votes_dense.p, n_votes * sizeof(Vote), hipMemcpyDeviceToHost, st));
    if (n_ovf) ENC_HIP(hipMemcpyAsync(ovf.data(), b.d_ovf.p, ovf.size() * sizeof(int4), hipMemcpyDeviceToHost, st));
    if (n_pool) ENC_HIP(hipMemcpyAsync(b.h_pool.p, b.d_pool.p, (size_t)n_pool * POOL_SLOT, hipMemcpyDeviceToHost, st));
    ENC_HIP(hipStreamSynchronize(st));
    const auto t_host = std::chrono::steady_clock::now();
    if (n_ovf) {                                                 // rare alphabet: a handful per batch, grouped here
        std::sort(ovf.begin(), ovf.end(), [](const int4& x, const int4& y) { return x.w < y.w; });
        for (const int4& o : ovf) o0[(size_t)o.w + 1] += 1;
    }
    for (int r = 0; r < n_regions; ++r) o0[(size_t)r + 1] += o0[(size_t)r];
    b.ms[6] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_host).count();
    std::vector<RegionOut> outs((size_t)n_regions);
    const std::function<void(int)> work = [&](int r) {
        std::sort(sites + s0[(size_t)r], sites + s0[(size_t)r + 1], [](const SiteRec& x, const SiteRec& y) { return x.idx < y.idx; });
        // votes by (site, type, allele): one 64-bit key per vote decides nearly every comparison (28 bits of site, 2 of type, the
        // first 34 bits of the allele); the rare ties fall back to the full order
        const AlleleSrc pile{b.regs[(size_t)r].p.reference, pool};
        Vote* vb = votes + v0[(size_t)r];
        const size_t nv = v0[(size_t)r + 1] - v0[(size_t)r];
        std::vector<std::pair<uint64_t, uint32_t>> order(nv);
        for (size_t k = 0; k < nv; ++k)
            order[k] = {((uint64_t)vb[k].idx << 36) | ((uint64_t)(vb[k].meta & 3u) << 34) | (vb[k].prefix >> 30), (uint32_t)k};
        std::sort(order.begin(), order.end(), [&](const std::pair<uint64_t, uint32_t>& x, const std::pair<uint64_t, uint32_t>& y) {
            return x.first != y.first ? x.first < y.first : vote_less(vb[x.second], vb[y.second], pile);
        });
        std::vector<Vote> sorted(nv);
        for (size_t k = 0; k < nv; ++k) sorted[k] = vb[order[k].second];
        std::copy(sorted.begin(), sorted.end(), vb);
        enumerate_region(b.regs[(size_t)r], r, b.mid, sites + s0[(size_t)r], s0[(size_t)r + 1] - s0[(size_t)r], votes + v0[(size_t)r],
                         v0[(size_t)r + 1] - v0[(size_t)r], ovf.data() + o0[(size_t)r], o0[(size_t)r + 1] - o0[(size_t)r], pool, outs[(size_t)r]);
    };
    if (n_regions < 4 || b.host_threads == 1) {
        for (int r = 0; r < n_regions; ++r) work(r);
    } else {
        if (!b.pool) {
            const char* env = getenv("PA_ENCODER_HOST_THREADS");
            // a burst of ~0.3 ms tasks, one per region: a quarter of the hardware threads even where a cgroup quota grants fewer
            // CPUs on average (the quota is per 100 ms period; measured on the 16-of-256 box: 16 threads 1.45 ms per 64 regions,
            // 32 threads 0.94 ms)
            const int dflt = b.host_threads > 0 ? b.host_threads : std::max(usable_cpus(), (int)std::thread::hardware_concurrency() / 4);
            b.pool.reset(new RegionPool(std::max(1, std::min(env ? atoi(env) : dflt, 64)) - 1));
        }
        b.pool->run(n_regions, work);
    }
    b.ms[7] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_host).count();   // ... + the regions' threads
    std::vector<CandDesc> cands;
    for (int r = 0; r < n_regions; ++r) {
        const RegionOut& o = outs[(size_t)r];
        b.region_n[(size_t)r] = (int64_t)o.cands.size();
        cands.insert(cands.end(), o.cands.begin(), o.cands.end());
        b.positions.insert(b.positions.end(), o.positions.begin(), o.positions.end());
        b.depths.insert(b.depths.end(), o.depths.begin(), o.depths.end());
        b.freqs.insert(b.freqs.end(), o.freqs.begin(), o.freqs.end());
        b.names += o.names;
    }
    b.ms[4] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_host).count();   // host enumeration
    b.n = (int64_t)cands.size();
    if (n_candidates)
        for (int r = 0; r < n_regions; ++r) n_candidates[r] = b.region_n[(size_t)r];
    if (b.n > 0) {
        ENC_ALLOC(b.d_cands, cands.size() * sizeof(CandDesc));
        ENC_ALLOC(b.d_img32, (size_t)b.n * b.W * b.F * sizeof(int));
        ENC_ALLOC(b.d_img8, (size_t)b.n * b.W * b.F);
        ENC_HIP(hipMemcpyAsync(b.d_cands.p, cands.data(), cands.size() * sizeof(CandDesc), hipMemcpyHostToDevice, st));
        ENC_HIP(hipEventRecord(e->ev[4], st));
        hipLaunchKernelGGL(gather_windows_kernel, dim3((unsigned)b.n), dim3(64), 0, st, b.d_mat.as<int>(), b.p_regions,
                           b.d_cands.as<CandDesc>(), b.W, b.F, b.mid, b.d_img32.as<int>(), b.d_img8.as<int8_t>());
        ENC_HIP(hipEventRecord(e->ev[5], st));
        ENC_HIP(hipGetLastError());
        ENC_HIP(hipStreamSynchronize(st));
        (void)hipEventElapsedTime(&ms, e->ev[4], e->ev[5]); b.ms[3] = ms;   // gather_windows_kernel
    }
    b.ms[5] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();  // whole run, host clock
    return PA_OK;
}

}  // namespace

// The packed reads of a batch of regions clipped and decoded on the device WITHOUT the variant encoder's tables: what the polish
// image chain (encoder_polish.hip) starts from.  Same arena / read / pair tables as stage_packed, same kernel; the regions are
// given by their bounds alone.  Nothing here waits for the device.
int pa_enc::unpack_packed_regions(pa_encoder* e, int32_t n_regions, const int64_t* region_start, const int64_t* region_end,
                                  const uint8_t* arena, int64_t arena_bytes, const pa_packed_read* reads, int32_t n_reads,
                                  const int32_t* pair_read, const int32_t* region_pairs, int32_t extra_ops_per_pair, UnpackedReads* out) {
    if (!e || !out || n_regions < 0 || (n_regions > 0 && (!region_start || !region_end || !region_pairs)) || arena_bytes < 0 || n_reads < 0 ||
        (n_reads > 0 && (!reads || !pair_read)))
        return pa::set_error(PA_ERR_INVALID, "null argument");
    ENC_HIP(hipSetDevice(e->device));
    if (!e->variant) e->variant = new pa_variant_batch();
    pa_variant_batch& b = *e->variant;
    const bool resident = arena == nullptr && n_reads > 0;
    if (resident && (b.resident_bytes <= 0 || arena_bytes > b.resident_bytes))
        return pa::set_error(PA_ERR_INVALID, "no arena given and no inflated span of that size resident on the device");
    b.staged = false;               // (the variant encoder's staged batch, if any, shares these buffers)
    const int64_t n_pairs = n_regions ? region_pairs[n_regions] : 0;
    if (n_pairs < 0 || (n_regions && region_pairs[0] != 0)) return pa::set_error(PA_ERR_INVALID, "region_pairs must start at 0 and ascend");
    auto up16 = [](size_t v) { return (v + 15) & ~(size_t)15; };
    const size_t o_reg = 0, o_start = up16(o_reg + (size_t)n_regions * sizeof(RegRec)), o_reads = up16(o_start + (size_t)n_regions * 8),
                 o_pairs = up16(o_reads + (size_t)n_reads * sizeof(PackedRead)), meta_bytes = up16(o_pairs + (size_t)n_pairs * sizeof(PairRec)) + 64;
    if (!b.h_meta.ensure(meta_bytes) || !b.h_live.ensure(((size_t)n_regions + 2) * 4)) return pa::set_error(PA_ERR_HIP, "hipHostMalloc failed");
    char* hm = b.h_meta.as<char>();
    RegRec* regrecs = reinterpret_cast<RegRec*>(hm + o_reg);
    int64_t* rstart = reinterpret_cast<int64_t*>(hm + o_start);
    PairRec* pairs = reinterpret_cast<PairRec*>(hm + o_pairs);
    if (n_reads) std::memcpy(hm + o_reads, reads, (size_t)n_reads * sizeof(PackedRead));
    int64_t total_bases = 0, total_ops = 0;
    for (int r = 0; r < n_regions; ++r) {
        if (region_end[r] < region_start[r] || region_end[r] - region_start[r] > (int64_t)1 << 28) return pa::set_error(PA_ERR_INVALID, "bad region");
        if (region_pairs[r + 1] < region_pairs[r]) return pa::set_error(PA_ERR_INVALID, "region_pairs must ascend");
        RegRec& g = regrecs[r];
        std::memset(&g, 0, sizeof(g));
        g.L = (int32_t)(region_end[r] - region_start[r] + 1);
        rstart[r] = region_start[r];
        for (int32_t k = region_pairs[r]; k < region_pairs[r + 1]; ++k) {
            const int32_t ri = pair_read[k];
            if (ri < 0 || ri >= n_reads) return pa::set_error(PA_ERR_INVALID, "pair_read out of range");
            const pa_packed_read& rd = reads[ri];
            if (rd.n_cigar < 0 || rd.l_seq < 0 || rd.data_off < 0 ||
                rd.data_off + 4ll * rd.n_cigar + (rd.l_seq + 1) / 2 + rd.l_seq > arena_bytes)
                return pa::set_error(PA_ERR_INVALID, "packed read " + std::to_string(ri) + " lies outside the arena");
            // the clipped stretch of a read holds its aligned bases inside the region and the inserts between them: room for the
            // whole read would be 10-50 kb per pair of a 1 kb region, so a pair gets min(l_seq, 2 L + 64) bases; a pair that
            // keeps more is reported as unsupported (h_live[n_regions + 1]) and the caller takes the host-clipped form
            const int64_t cap = std::min<int64_t>(rd.l_seq, 2ll * g.L + 64);
            pairs[k] = PairRec{total_bases, ri, r, (int32_t)total_ops, (int32_t)cap};
            total_bases += ((cap + 3) & ~(int64_t)3) + 4;
            // (kept M / D / N operations each cover a row of the region, kept I / S ones a kept base)
            total_ops += std::min<int64_t>(rd.n_cigar, (int64_t)g.L + cap + 2);
        }
    }
    const int64_t extra_ops = extra_ops_per_pair < 0 ? 0 : total_bases + n_pairs * (int64_t)extra_ops_per_pair;
    if (total_ops + extra_ops > 0x7ffffff0) return pa::set_error(PA_ERR_INVALID, "batch too large: more than 2^31 CIGAR operations");
    hipStream_t st = e->stream;
    if (!resident) {
        ENC_ALLOC(b.d_arena, (size_t)arena_bytes + 256);
        b.resident_bytes = 0;
    }
    ENC_ALLOC(b.d_meta, meta_bytes);
    ENC_ALLOC(b.d_live, ((size_t)n_regions + 2) * 4);
    ENC_ALLOC(b.d_seq, (size_t)total_bases + 64);
    ENC_ALLOC(b.d_qual, (size_t)total_bases + 64);
    ENC_ALLOC(b.d_cig_op, (size_t)(total_ops + extra_ops) * 4 + 1024);
    ENC_ALLOC(b.d_cig_len, (size_t)(total_ops + extra_ops) * 4 + 1024);
    ENC_ALLOC(b.d_reads, (size_t)n_pairs * sizeof(ReadRec) + 64);
    if (arena_bytes > 0 && !resident) ENC_HIP(hipMemcpyAsync(b.d_arena.p, arena, (size_t)arena_bytes, hipMemcpyHostToDevice, st));
    ENC_HIP(hipMemcpyAsync(b.d_meta.p, hm, meta_bytes, hipMemcpyHostToDevice, st));
    ENC_HIP(hipMemsetAsync(b.d_live.p, 0, ((size_t)n_regions + 2) * 4, st));
    const char* dm = b.d_meta.as<char>();
    if (n_pairs > 0) {
        UnpackArgs ua;
        ua.pairs = reinterpret_cast<const PairRec*>(dm + o_pairs);
        ua.n_pairs = (int)n_pairs;
        ua.preads = reinterpret_cast<const PackedRead*>(dm + o_reads);
        ua.regions = reinterpret_cast<const RegRec*>(dm + o_reg);
        ua.region_start = reinterpret_cast<const int64_t*>(dm + o_start);
        ua.arena = b.d_arena.as<uint8_t>();
        ua.reads = b.d_reads.as<ReadRec>();
        ua.cigar_op = b.d_cig_op.as<int32_t>();
        ua.cigar_len = b.d_cig_len.as<int32_t>();
        ua.seq = b.d_seq.as<char>();
        ua.qual = b.d_qual.as<uint8_t>();
        ua.live = b.d_live.as<int>();
        ua.n_regions = n_regions;
        hipLaunchKernelGGL(unpack_clip_kernel, dim3((unsigned)((n_pairs + 3) / 4)), dim3(256), 0, st, ua);
        ENC_HIP(hipGetLastError());
    }
    ENC_HIP(hipMemcpyAsync(b.h_live.p, b.d_live.p, ((size_t)n_regions + 2) * 4, hipMemcpyDeviceToHost, st));
    out->reads = b.d_reads.as<ReadRec>();
    out->cigar_op = b.d_cig_op.as<int32_t>();
    out->cigar_len = b.d_cig_len.as<int32_t>();
    out->seq = b.d_seq.as<char>();
    out->n_pairs = n_pairs;
    out->total_bases = total_bases;
    out->total_ops = total_ops;
    out->extra_ops = extra_ops;
    out->h_live = b.h_live.as<int>();
    return PA_OK;
}

#ifdef PA_ENC_STAMP
extern "C" int pa_encoder_debug_cycles(unsigned long long* out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_enc_cycles), sizeof(unsigned long long) * 8) != hipSuccess) return -1;
    if (reset) {
        unsigned long long zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_enc_cycles), zero, sizeof zero) != hipSuccess) return -1;
    }
    return 0;
}
#endif

extern "C" {

int pa_encoder_create(int32_t device, void* hip_stream, pa_encoder** out) {
    if (!out) return pa::set_error(PA_ERR_INVALID, "null argument");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return pa::set_error(PA_ERR_NO_DEVICE, "no HIP device visible: the pepper_amd encoder has no CPU fallback");
    if (device < 0 || device >= count) return pa::set_error(PA_ERR_INVALID, "device ordinal out of range");
    ENC_HIP(hipSetDevice(device));
    // PEPPER_AMD_BLOCKING_SYNC=1: waits on the device sleep instead of spinning (sixteen image-generation workers spinning in
    // hipStreamSynchronize use up the CPUs the other workers' host stages need); refused once the device is in use: ignored
    if (const char* v = getenv("PEPPER_AMD_BLOCKING_SYNC"))
        if (v[0] == '1') (void)hipSetDeviceFlags(hipDeviceScheduleBlockingSync);
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
    for (hipEvent_t& ev : e->ev)
        if (hipEventCreate(&ev) != hipSuccess) {
            pa_encoder_destroy(e);
            return pa::set_error(PA_ERR_HIP, "hipEventCreate failed");
        }
    *out = e;
    return PA_OK;
}

void pa_encoder_destroy(pa_encoder* e) {
    if (!e) return;
    (void)hipSetDevice(e->device);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    for (hipEvent_t ev : e->ev)
        if (ev) (void)hipEventDestroy(ev);
    pa_variant_batch_free(e->variant);
    pa_polish_batch_free(e->polish);
    if (e->realigner) pa_realigner_destroy(e->realigner);      // (the image chain's, on this encoder's stream)
    if (e->own_stream && e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
}

int pa_encoder_stage_batch(pa_encoder* e, int32_t n_regions, const pa_pileup* pileups, const pa_summary_params* params) {
    return stage_batch(e, n_regions, pileups, params);
}

int pa_encoder_run_staged(pa_encoder* e, int64_t* n_candidates) { return run_staged(e, n_candidates); }

void* pa_encoder_host_arena(pa_encoder* e, int64_t bytes) {
    if (!e || bytes < 0 || hipSetDevice(e->device) != hipSuccess) return nullptr;
    if (!e->variant) e->variant = new pa_variant_batch();
    if (e->stream) (void)hipStreamSynchronize(e->stream);        // (an upload out of the old block may still be running)
    return e->variant->h_arena.ensure((size_t)bytes) ? e->variant->h_arena.p : nullptr;
}

void* pa_encoder_host_span(pa_encoder* e, int64_t bytes) {
    if (!e || bytes < 0 || hipSetDevice(e->device) != hipSuccess) return nullptr;
    if (!e->variant) e->variant = new pa_variant_batch();
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    return e->variant->h_comp.ensure((size_t)bytes) ? e->variant->h_comp.p : nullptr;
}

// BGZF members -> the encoder's device arena (inflate.hip: one wavefront per member), and a copy for the host's record walk
int pa_encoder_inflate_bgzf(pa_encoder* e, const uint8_t* comp, int64_t comp_bytes, int32_t n_blocks, const int64_t* comp_off,
                            const int32_t* comp_len, const int64_t* out_off, const int32_t* out_len, int64_t out_bytes,
                            uint8_t* host_out) {
    if (!e || n_blocks < 0 || comp_bytes < 0 || out_bytes < 0 ||
        (n_blocks > 0 && (!comp || !comp_off || !comp_len || !out_off || !out_len)))
        return pa::set_error(PA_ERR_INVALID, "null or negative argument");
    for (int32_t k = 0; k < n_blocks; ++k)
        if (comp_off[k] < 0 || comp_len[k] < 0 || comp_off[k] + comp_len[k] > comp_bytes || out_off[k] < 0 || out_len[k] < 0 ||
            out_off[k] + out_len[k] > out_bytes)
            return pa::set_error(PA_ERR_INVALID, "BGZF block " + std::to_string(k) + " lies outside the buffers");
    ENC_HIP(hipSetDevice(e->device));
    if (!e->variant) e->variant = new pa_variant_batch();
    pa_variant_batch& b = *e->variant;
    b.resident_bytes = 0;
    b.ms[10] = b.ms[11] = 0;
    if (n_blocks == 0) return PA_OK;
    hipStream_t st = e->stream;
    const size_t nb = (size_t)n_blocks, table_bytes = nb * 28;
    ENC_ALLOC(b.d_arena, (size_t)out_bytes + 256);
    ENC_ALLOC(b.d_comp, (size_t)comp_bytes + 64);
    ENC_ALLOC(b.d_inf, table_bytes);
    if (!b.h_inf.ensure(table_bytes)) return pa::set_error(PA_ERR_HIP, "hipHostMalloc failed in the inflate tables");
    char* hm = b.h_inf.as<char>();
    std::memcpy(hm, comp_off, nb * 8);
    std::memcpy(hm + nb * 8, out_off, nb * 8);
    std::memcpy(hm + nb * 16, comp_len, nb * 4);
    std::memcpy(hm + nb * 20, out_len, nb * 4);
    // The tables are read, and the status words written, in the page-locked block itself (it is mapped into the device's
    // address space): each of a job's small copies otherwise waits for CUs behind other workers' long-lived inflate
    // wavefronts (the runtime's copy kernels: ~1.5 ms apiece in the trace of DESIGN.md 4.4).
    void* mapped = nullptr;
    if (hipHostGetDevicePointer(&mapped, hm, 0) != hipSuccess || !mapped) mapped = nullptr;
    std::memset(hm + nb * 24, 0xff, nb * 4);                  // (a member the kernel never reaches reads as an error)
    char* dm = mapped ? static_cast<char*>(mapped) : b.d_inf.as<char>();
    const auto t0 = std::chrono::steady_clock::now();
    ENC_HIP(hipMemcpyAsync(b.d_comp.p, comp, (size_t)comp_bytes, hipMemcpyHostToDevice, st));
    if (!mapped) ENC_HIP(hipMemcpyAsync(dm, hm, nb * 24, hipMemcpyHostToDevice, st));
    ENC_HIP(hipEventRecord(e->ev[10], st));
    pa::launch_bgzf_inflate(st, b.d_comp.as<uint8_t>(), reinterpret_cast<const int64_t*>(dm), reinterpret_cast<const int32_t*>(dm + nb * 16),
                            reinterpret_cast<const int64_t*>(dm + nb * 8), reinterpret_cast<const int32_t*>(dm + nb * 20),
                            b.d_arena.as<uint8_t>(), reinterpret_cast<int32_t*>(dm + nb * 24), n_blocks, comp_bytes);
    ENC_HIP(hipGetLastError());
    ENC_HIP(hipEventRecord(e->ev[11], st));
    if (!mapped) ENC_HIP(hipMemcpyAsync(hm + nb * 24, dm + nb * 24, nb * 4, hipMemcpyDeviceToHost, st));
    if (host_out && out_bytes > 0) ENC_HIP(hipMemcpyAsync(host_out, b.d_arena.p, (size_t)out_bytes, hipMemcpyDeviceToHost, st));
    ENC_HIP(hipStreamSynchronize(st));
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, e->ev[10], e->ev[11]) == hipSuccess) b.ms[10] = ms;
    b.ms[11] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    const int32_t* status = reinterpret_cast<const int32_t*>(hm + nb * 24);
    for (int32_t k = 0; k < n_blocks; ++k)
        if (status[k] != 0)
            return pa::set_error(PA_ERR_INVALID, "BGZF block " + std::to_string(k) + ": " + pa::inflate_status_text(status[k]));
    b.resident_bytes = out_bytes;
    return PA_OK;
}

// The record headers of the span pa_encoder_inflate_bgzf left on the device (inflate.hip: record_chase_kernel and friends)
int pa_encoder_walk_records(pa_encoder* e, int64_t data_bytes, const int64_t* entries, int32_t n_entries, int32_t cap_per_entry,
                            void* headers, int64_t headers_cap, int64_t* n_headers, int32_t* flags) {
    if (!e || !entries || n_entries < 1 || cap_per_entry < 1 || !headers || headers_cap < 0 || !n_headers || !flags || data_bytes < 0)
        return pa::set_error(PA_ERR_INVALID, "null or invalid argument");
    *n_headers = 0;
    flags[0] = flags[1] = 0;
    if (!e->variant || e->variant->resident_bytes <= 0 || data_bytes > e->variant->resident_bytes)
        return pa::set_error(PA_ERR_INVALID, "no inflated span of that size resident on the device");
    for (int32_t k = 0; k < n_entries; ++k)
        if (entries[k] < 0 || entries[k] > data_bytes || (k > 0 && entries[k] <= entries[k - 1]))
            return pa::set_error(PA_ERR_INVALID, "record entries must ascend inside the span");
    ENC_HIP(hipSetDevice(e->device));
    pa_variant_batch& b = *e->variant;
    hipStream_t st = e->stream;
    const size_t n = (size_t)n_entries, slots = n * (size_t)cap_per_entry;
    const size_t o_counts = n * 8, o_base = o_counts + n * 4, o_flags = o_base + (n + 1) * 4, o_slots = (o_flags + 8 + 63) & ~(size_t)63,
                 o_out = o_slots + slots * 40, total = o_out + slots * 40;
    ENC_ALLOC(b.d_walk, total);
    const size_t h_tail = (n * 8 + 63) & ~(size_t)63;
    if (!b.h_walk.ensure(h_tail + 64)) return pa::set_error(PA_ERR_HIP, "hipHostMalloc failed in the record walk");
    char* dw = b.d_walk.as<char>();
    char* hw = b.h_walk.as<char>();
    std::memcpy(hw, entries, n * 8);
    int32_t* tail = reinterpret_cast<int32_t*>(hw + h_tail);           // [0] the number of records, [1..2] the flags
    tail[0] = tail[1] = tail[2] = 0;
    // entries in and totals out through the page-locked block itself where it is mapped (see pa_encoder_inflate_bgzf)
    void* mapped = nullptr;
    if (hipHostGetDevicePointer(&mapped, hw, 0) != hipSuccess || !mapped) mapped = nullptr;
    char* mw = static_cast<char*>(mapped);
    const auto t0 = std::chrono::steady_clock::now();
    if (!mapped) ENC_HIP(hipMemcpyAsync(dw, hw, n * 8, hipMemcpyHostToDevice, st));
    ENC_HIP(hipMemsetAsync(dw + o_flags, 0, 8, st));
    pa::launch_record_walk(st, b.d_arena.as<uint8_t>(), data_bytes, reinterpret_cast<const int64_t*>(mapped ? mw : dw), n_entries,
                           cap_per_entry, dw + o_slots, reinterpret_cast<int32_t*>(dw + o_counts), reinterpret_cast<int32_t*>(dw + o_base),
                           reinterpret_cast<int32_t*>(dw + o_flags), dw + o_out, (int64_t)slots,
                           mapped ? reinterpret_cast<int32_t*>(mw + h_tail) : nullptr);
    ENC_HIP(hipGetLastError());
    if (!mapped) {
        ENC_HIP(hipMemcpyAsync(tail, dw + o_base + n * 4, 4, hipMemcpyDeviceToHost, st));
        ENC_HIP(hipMemcpyAsync(tail + 1, dw + o_flags, 8, hipMemcpyDeviceToHost, st));
    }
    ENC_HIP(hipStreamSynchronize(st));
    flags[0] = tail[1];
    flags[1] = tail[2];
    const int64_t found = tail[0];
    if (flags[0] == 0 && found > headers_cap) flags[0] |= 4;            // the caller's table is too small
    if (flags[0] == 0 && found > 0) {                                   // (the headers themselves: one copy out of device memory --
        ENC_HIP(hipMemcpyAsync(headers, dw + o_out, (size_t)found * 40, hipMemcpyDeviceToHost, st));      // 40-byte stores of single
        ENC_HIP(hipStreamSynchronize(st));                                                              // lanes across PCIe are slower)
    }
    *n_headers = flags[0] == 0 ? found : 0;
    b.ms[11] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return PA_OK;
}

int pa_encoder_stage_packed(pa_encoder* e, int32_t n_regions, const pa_packed_region* regions, const pa_summary_params* params,
                            const uint8_t* arena, int64_t arena_bytes, const pa_packed_read* reads, int32_t n_reads,
                            const int32_t* pair_read, const int32_t* region_pairs) {
    return stage_packed(e, n_regions, regions, params, arena, arena_bytes, reads, n_reads, pair_read, region_pairs);
}

int pa_encoder_set_host_threads(pa_encoder* e, int32_t n) {
    if (!e || n < 0) return pa::set_error(PA_ERR_INVALID, "null encoder or negative thread count");
    if (!e->variant) e->variant = new pa_variant_batch();
    e->variant->host_threads = n;
    e->variant->pool.reset();
    return PA_OK;
}

int pa_encoder_region_reads(pa_encoder* e, int32_t* n_reads, int32_t n) {
    if (!e || !n_reads || n < 0) return pa::set_error(PA_ERR_INVALID, "null argument");
    for (int i = 0; i < n; ++i) n_reads[i] = (e->variant && i < (int)e->variant->live.size()) ? e->variant->live[(size_t)i] : 0;
    return PA_OK;
}

int pa_encoder_generate_summary_batch(pa_encoder* e, int32_t n_regions, const pa_pileup* pileups, const pa_summary_params* params,
                                      int64_t* n_candidates) {
    const int rc = stage_batch(e, n_regions, pileups, params);
    return rc != PA_OK ? rc : run_staged(e, n_candidates);
}

int pa_encoder_generate_summary(pa_encoder* e, const pa_pileup* p, const pa_summary_params* q, int64_t* n_candidates) {
    if (!e || !p || !q || !n_candidates) return pa::set_error(PA_ERR_INVALID, "null argument");
    return pa_encoder_generate_summary_batch(e, 1, p, q, n_candidates);
}

int pa_encoder_get_results(pa_encoder* e, int64_t* positions, int32_t* depths, int32_t* candidate_frequency,
                           int32_t* images_i32, int8_t* images_i8, char* candidates, int64_t candidates_cap,
                           int64_t* candidates_needed) {
    if (!e) return pa::set_error(PA_ERR_INVALID, "null encoder");
    ENC_HIP(hipSetDevice(e->device));
    if (!e->variant) {
        if (candidates_needed) *candidates_needed = 0;
        return PA_OK;
    }
    const pa_variant_batch& b = *e->variant;
    const size_t n = (size_t)b.n;
    if (positions) std::copy(b.positions.begin(), b.positions.end(), positions);
    if (depths) std::copy(b.depths.begin(), b.depths.end(), depths);
    if (candidate_frequency) std::copy(b.freqs.begin(), b.freqs.end(), candidate_frequency);
    if (candidates_needed) *candidates_needed = (int64_t)b.names.size();
    if (candidates && candidates_cap >= (int64_t)b.names.size()) std::memcpy(candidates, b.names.data(), b.names.size());
    if (n > 0 && images_i32)
        ENC_HIP(hipMemcpyAsync(images_i32, b.d_img32.p, n * b.W * b.F * sizeof(int), hipMemcpyDeviceToHost, e->stream));
    if (n > 0 && images_i8)
        ENC_HIP(hipMemcpyAsync(images_i8, b.d_img8.p, n * b.W * b.F, hipMemcpyDeviceToHost, e->stream));
    ENC_HIP(hipStreamSynchronize(e->stream));
    return PA_OK;
}

const int8_t* pa_encoder_device_images(pa_encoder* e) {
    return (e && e->variant && e->variant->n > 0) ? e->variant->d_img8.as<int8_t>() : nullptr;
}

int pa_encoder_last_timing(pa_encoder* e, double* ms, int32_t n) {
    if (!e || !ms || n < 0) return pa::set_error(PA_ERR_INVALID, "null argument");
    for (int i = 0; i < n; ++i) ms[i] = (e->variant && i < 12) ? e->variant->ms[i] : 0.0;
    return PA_OK;
}

int pa_encoder_batch_stats(pa_encoder* e, int64_t* out, int32_t n) {
    if (!e || !out || n < 0) return pa::set_error(PA_ERR_INVALID, "null argument");
    const int64_t v[6] = {e->variant ? e->variant->total_bases : 0, e->variant ? e->variant->total_rows : 0,
                          e->variant ? e->variant->total_reads : 0, e->variant ? e->variant->total_ops : 0,
                          e->variant ? (int64_t)e->variant->n_tiles : 0, e->variant ? (int64_t)e->variant->regs.size() : 0};
    for (int i = 0; i < n; ++i) out[i] = i < 6 ? v[i] : 0;
    return PA_OK;
}

}  // extern "C"
