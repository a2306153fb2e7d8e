// ORACLE (test infrastructure; never linked or imported by pepper_amd/): scalar restatement of the local
// re-alignment the polish image generator applies to every read before summarising it
//   /root/reference/pepper/modules/python/AlignmentSummarizer.py:159-177,328-332  (realignment_flag defaults to True)
//   /root/reference/pepper/modules/src/local_reassembly/simple_aligner.cpp:66-106 (ReadAligner)
// whose arithmetic is the SSW library vendored with the reference (striped Smith-Waterman, Zhao et al. 2013):
//   /root/reference/pepper/modules/src/local_reassembly/ssw.c:161-368   score + end cell, 8-bit lanes
//   /root/reference/pepper/modules/src/local_reassembly/ssw.c:393-569   score + end cell, 16-bit lanes
//   /root/reference/pepper/modules/src/local_reassembly/ssw.c:571-757   banded DP + trace-back
//   /root/reference/pepper/modules/src/local_reassembly/ssw.c:801-891   ssw_align (forward, reverse, band)
//   /root/reference/pepper/modules/src/local_reassembly/ssw_cpp.cpp:56-207,329-363  cigar text (= X I D S)
// PINNED: tests/test_realign_oracle.py compares this file with the reference's own SSW build
// (oracle/_ref/libref_ssw.so, oracle/ref_ssw_driver.cpp) on seeded read sets, and with the vectors that build
// produced (tests/golden/realign_*.npz, tests/golden/make_golden_realign.py).
//
// What has to be reproduced beyond "affine-gap Smith-Waterman":
//  * The score pass is the striped kernel: the read is cut into `lanes` equal segments of L = ceil(m / lanes)
//    rows (lanes = 16 with 8-bit cells, 8 with 16-bit cells; rows beyond the read score 0 against everything).
//    Vertical gaps are exact, but the horizontal-gap state E of the next column is opened from the cell value
//    *before* the lazy vertical-gap correction, i.e. from H' = max(diagonal, E, F') where F' only sees openings
//    inside the same segment.  H = max(H', F) with the exact F.  (ssw.c:239-262 / 462-485: pvE is stored in
//    the inner loop and never touched by the lazy-F loop.)
//  * 8-bit pass first (bias 6); any running maximum >= 249 abandons it for the 16-bit pass (ssw.c:819-824).
//  * End cell: first column that reaches the maximum, smallest read row inside it, capped at m - 1.
//  * Begin cell: the same pass over the reversed read prefix and the reference walked backwards from the end
//    column, stopped at the first column whose maximum equals the forward score.
//  * Band: |n' - m'| + 1, doubled until the banded maximum reaches the score; slot indexing, the zeroed slot to
//    the right of the previous row and the stale slots are those of ssw.c:606-650; ties: open beats extend only
//    when strictly larger, the diagonal wins ties against gaps, E wins only when strictly larger than F.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

constexpr int MATCH = 4, MISMATCH = 6, GAP_OPEN = 8, GAP_EXT = 2, BIAS = 6;

inline int code_of(char c) {          // ssw_cpp.cpp:10-19: A C G T (and U as A?) -> the table maps U/u to 0 as well
    switch (c) {
        case 'A': case 'a': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': return 3;
        case 'U': case 'u': return 0;
        default: return 4;
    }
}
inline int sub_score(int a, int b) { return (a == b && a < 4) ? MATCH : -MISMATCH; }   // ssw_cpp.cpp:21-40
inline int sat0(int v) { return v > 0 ? v : 0; }

struct EndCell { int score = 0, ref = 0, read = 0; bool overflow = false; };

// One striped score pass.  `ref` is visited from `first` in direction `step` for `count` columns.
EndCell score_pass(const std::vector<int8_t>& ref, int first, int step, int count, const int8_t* read, int m, int lanes,
                   int terminate) {
    const int L = (m + lanes - 1) / lanes, rows = L * lanes;
    std::vector<int> Hprev(rows, 0), H(rows, 0), E(rows, 0);
    EndCell out;
    out.read = m - 1;
    out.ref = lanes == 16 ? -1 : 0;                // ssw.c:185 vs 410
    std::vector<int> best_col(rows, 0);
    int run_max = 0;
    for (int c = 0; c < count; ++c) {
        const int i = first + c * step, rc = ref[(size_t)i];
        int fseg = 0, ffull = 0, hseg_prev = 0, hfull_prev = 0, col_max = 0;
        for (int r = 0; r < rows; ++r) {
            const int s = r < m ? sub_score(read[r], rc) : 0;
            const int diag = (r == 0 ? 0 : Hprev[r - 1]) + s;
            fseg = (r % L == 0) ? 0 : std::max(sat0(fseg - GAP_EXT), sat0(hseg_prev - GAP_OPEN));
            ffull = (r == 0) ? 0 : std::max(sat0(ffull - GAP_EXT), sat0(hfull_prev - GAP_OPEN));
            const int hseg = std::max(std::max(diag, E[r]), fseg);
            const int h = std::max(hseg, ffull);
            E[r] = std::max(sat0(E[r] - GAP_EXT), sat0(hseg - GAP_OPEN));
            H[r] = h;
            hseg_prev = hseg;
            hfull_prev = h;
            col_max = std::max(col_max, h);
        }
        if (col_max > run_max) {
            run_max = col_max;
            if (lanes == 16 && run_max + BIAS >= 255) { out.overflow = true; break; }
            out.ref = i;
            best_col = H;
        }
        Hprev.swap(H);
        if (col_max == terminate) break;
    }
    out.score = out.overflow ? 255 : run_max;
    for (int r = 0; r < rows; ++r)
        if (best_col[r] == run_max && r < out.read) { out.read = r; break; }
    return out;
}

struct RawOp { char op; int len; };
thread_local int g_band_first = 0, g_band_final = 0;     // diagnostics of the last banded_path call

// Banded DP + trace-back over ref[0..n) x read[0..m); returns M/I/D runs in alignment order.
bool banded_path(const int8_t* ref, const int8_t* read, int n, int m, int score, std::vector<RawOp>& ops) {
    int bw = std::abs(n - m) + 1;
    g_band_first = bw;
    std::vector<int> hb, eb, hc;
    std::vector<int8_t> dir;
    int width_d = 0;
    for (;;) {
        const int width = 2 * bw + 3;
        width_d = 2 * bw + 1;
        if ((int64_t)width_d * m * 3 > ((int64_t)1 << 31)) return false;     // ssw.c:611-618 gives up (exit) here
        hb.assign((size_t)width + 1, 0);
        eb.assign((size_t)width + 1, 0);
        hc.assign((size_t)width + 1, 0);
        dir.assign((size_t)width_d * m * 3 + 8, 0);
        int best = 0;
        for (int i = 0; i < m; ++i) {
            const int beg = std::max(0, i - bw), end = std::min(n - 1, i + bw);
            const int edge = std::min(end + 1, width - 1);
            int f = 0;
            hb[0] = eb[0] = hb[(size_t)edge] = eb[(size_t)edge] = hc[0] = 0;
            int8_t* line = dir.data() + (size_t)width_d * i * 3;
            const int x = std::max(i - bw, 0), xp = std::max(i - 1 - bw, 0);
            int u = 0;
            for (int j = beg; j <= end; ++j) {
                u = j - x + 1;
                const int e = j - xp + 1, b = u - 1, d = e - 1, slot = (j - x) * 3;
                int t1 = i == 0 ? -GAP_OPEN : hb[(size_t)e] - GAP_OPEN;
                int t2 = i == 0 ? -GAP_EXT : eb[(size_t)e] - GAP_EXT;
                eb[(size_t)u] = std::max(t1, t2);
                line[slot] = t1 > t2 ? 3 : 2;
                t1 = hc[(size_t)b] - GAP_OPEN;
                t2 = f - GAP_EXT;
                f = std::max(t1, t2);
                line[slot + 1] = t1 > t2 ? 5 : 4;
                const int e1 = sat0(eb[(size_t)u]), f1 = sat0(f);
                const int gap = std::max(e1, f1);
                const int diag = hb[(size_t)d] + sub_score(ref[j], read[i]);
                hc[(size_t)u] = std::max(gap, diag);
                best = std::max(best, hc[(size_t)u]);
                line[slot + 2] = gap <= diag ? 1 : (e1 > f1 ? line[slot] : line[slot + 1]);
            }
            for (int k = 1; k <= u; ++k) hb[(size_t)k] = hc[(size_t)k];
        }
        if (best >= score) break;
        bw *= 2;
    }
    g_band_final = bw;
    // trace back from the bottom-right corner (ssw.c:653-703)
    int i = m - 1, j = n - 1, run = 0, state = 2;
    char op = 'M', prev = 'M';
    std::vector<RawOp> rev;
    while (i > 0) {
        const int x = std::max(i - bw, 0);
        const int64_t at = (int64_t)width_d * i * 3 + (int64_t)(j - x) * 3 + state;
        if (at < 0 || at >= (int64_t)dir.size()) return false;
        switch (dir[(size_t)at]) {
            case 1: --i; --j; state = 2; op = 'M'; break;
            case 2: --i; state = 0; op = 'I'; break;
            case 3: --i; state = 2; op = 'I'; break;
            case 4: --j; state = 1; op = 'D'; break;
            case 5: --j; state = 2; op = 'D'; break;
            default: return false;
        }
        if (op == prev) ++run;
        else { rev.push_back({prev, run}); prev = op; run = 1; }
    }
    if (op == 'M') rev.push_back({'M', run + 1});
    else { rev.push_back({op, run}); rev.push_back({'M', 1}); }
    ops.assign(rev.rbegin(), rev.rend());
    return true;
}

}  // namespace

extern "C" void ssw_oracle_last_band(int32_t* first, int32_t* final_width) {
    *first = g_band_first;
    *final_width = g_band_final;
}

// out[0..5] = score, ref_begin, ref_end, query_begin, query_end, wide (1: the 16-bit pass produced the result)
// returns 1 on success (cigar text written), 0 when the library would not have aligned (empty inputs), -1 on error
extern "C" int ssw_oracle_align(const char* ref_txt, int32_t n, const char* query_txt, int32_t m, int32_t* out, char* cigar,
                                int32_t cigar_cap) {
    for (int k = 0; k < 6; ++k) out[k] = 0;
    if (cigar_cap > 0) cigar[0] = 0;
    if (n <= 0 || m <= 0) return 0;
    std::vector<int8_t> ref((size_t)n), read((size_t)m);
    for (int k = 0; k < n; ++k) ref[(size_t)k] = (int8_t)code_of(ref_txt[k]);
    for (int k = 0; k < m; ++k) read[(size_t)k] = (int8_t)code_of(query_txt[k]);

    int lanes = 16;
    EndCell fwd = score_pass(ref, 0, 1, n, read.data(), m, 16, -1);
    if (fwd.overflow) {
        lanes = 8;
        fwd = score_pass(ref, 0, 1, n, read.data(), m, 8, -1);
    }
    out[0] = fwd.score; out[2] = fwd.ref; out[4] = fwd.read; out[5] = lanes == 8;
    out[1] = -1; out[3] = -1;
    if (fwd.score <= 0 || fwd.ref < 0) return 1;          // nothing aligned; ReadAligner keeps the read (score <= 1)

    std::vector<int8_t> rread((size_t)fwd.read + 1);
    for (int k = 0; k <= fwd.read; ++k) rread[(size_t)k] = read[(size_t)(fwd.read - k)];
    const EndCell rev = score_pass(ref, fwd.ref, -1, fwd.ref + 1, rread.data(), fwd.read + 1, lanes, fwd.score);
    const int ref_begin = rev.ref, read_begin = fwd.read - rev.read;
    out[1] = ref_begin; out[3] = read_begin;

    std::vector<RawOp> ops;
    if (!banded_path(ref.data() + ref_begin, read.data() + read_begin, fwd.ref - ref_begin + 1, fwd.read - read_begin + 1,
                     fwd.score, ops))
        return -1;

    // cigar text: soft clips around the path, M runs split into '=' / 'X' by comparing base codes (N == N counts as '=')
    std::string txt;
    auto emit = [&](int len, char op) { txt += std::to_string(len); txt += op; };
    if (read_begin > 0) emit(read_begin, 'S');
    int rp = ref_begin, qp = read_begin, run_eq = 0, run_x = 0;
    auto flush = [&]() {
        if (run_eq) emit(run_eq, '=');
        else if (run_x) emit(run_x, 'X');
        run_eq = run_x = 0;
    };
    for (const RawOp& o : ops) {
        if (o.op == 'M') {
            for (int k = 0; k < o.len; ++k, ++rp, ++qp) {
                if (ref[(size_t)rp] != read[(size_t)qp]) { if (run_eq) { emit(run_eq, '='); run_eq = 0; } ++run_x; }
                else { if (run_x) { emit(run_x, 'X'); run_x = 0; } ++run_eq; }
            }
        } else {
            flush();
            emit(o.len, o.op);
            if (o.op == 'I') qp += o.len; else rp += o.len;
        }
    }
    flush();
    if (m - fwd.read - 1 > 0) emit(m - fwd.read - 1, 'S');
    if ((int)txt.size() + 1 > cigar_cap) return -1;
    std::memcpy(cigar, txt.c_str(), txt.size() + 1);
    return 1;
}
