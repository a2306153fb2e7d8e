// ORACLE (test infrastructure, never the product path): CPU restatement of PEPPER's two pileup
// summary encoders on flat arrays.  Only tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg may load libpileup_oracle.so; pepper_amd/ never does.
//
//  * variant:  RegionalSummaryGenerator::{generate_max_insert_summary, encode_reference_bases,
//              populate_summary_matrix, generate_summary}
//              /root/reference/pepper_variant/modules/cpp/region_summary.cpp:69-96, 174-191, 337-566, 568-916
//  * polish :  SummaryGenerator::{iterate_over_read, generate_image, generate_summary}
//              /root/reference/pepper/modules/src/pileup_summary/summary_generator.cpp:47-121, 274-306, 370-393
//
// Pinning: the variant restatement is checked against the reference's own implementation compiled
// into oracle/_ref/libref_variant_encoder.so (oracle/Makefile) on randomized and hand-built
// pileups (tests/test_encoder_oracle.py) and against committed golden vectors produced by it
// (tests/golden/encoder_variant_*.npz).  The polish restatement is pinned the same way against
// oracle/_ref/libref_polish_encoder.so -- the reference's SummaryGenerator compiled as it lies, with the
// htslib-backed #include of bam_handler.h dropped and that header's plain read types lifted by sed at build
// time (oracle/Makefile; no stand-in for htslib is written) -- on six pileup families
// (tests/test_encoder_oracle.py) and by golden vectors it produced (tests/golden/encoder_polish_*.npz).
//
// Reference quirks that are reproduced on purpose (each changes output bytes):
//  - GENERATE_INDELS == false (region_summary.h:50): no insert columns, row index = pos - ref_start.
//  - the last base of an M/=/X op that is followed by I or D does not count in the forward /
//    reverse coverage columns 4 / 15 (region_summary.cpp:381-391).
//  - REF_SKIP and PAD fall through into SOFT_CLIP and also advance the read index (:556-561).
//  - only columns 11..24 are clamped to +-125 (:648-653); 4, 8-10 and 25 are not.
//  - comparisons `ref_base != base` are case sensitive while the feature lookup upper-cases.
//  - polish: coverage of a deletion is credited to the deletion START for every deleted base
//    (summary_generator.cpp:105-110), pixels are double -> uint8 truncations of count/cov*254.
#include <algorithm>
#include <cctype>
#include <cstdint>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "pileup_abi.h"

namespace {

enum { OP_M = 0, OP_I = 1, OP_D = 2, OP_N = 3, OP_S = 4, OP_H = 5, OP_P = 6, OP_EQ = 7, OP_X = 8 };
constexpr int kMaxColor = 125;

inline bool is_acgt(char c) {
    c = (char)std::toupper((unsigned char)c);
    return c == 'A' || c == 'C' || c == 'G' || c == 'T';
}

// column of `symbol` for a strand, or -1 when the reference base is not A/C/G/T
// (region_summary.cpp:201-230): forward A..T = 8..11, I 12, D 13, other 14; reverse 19..25.
inline int symbol_column(char ref_base, char symbol, bool reverse) {
    if (!is_acgt(ref_base)) return -1;
    const int first = reverse ? 19 : 8;
    switch (std::toupper((unsigned char)symbol)) {
        case 'A': return first;
        case 'C': return first + 1;
        case 'G': return first + 2;
        case 'T': return first + 3;
        case 'I': return first + 4;
        case 'D': return first + 5;
        default: return first + 6;
    }
}

inline int base_code(char c) {   // region_summary.cpp:165-172
    switch (std::toupper((unsigned char)c)) {
        case 'A': return 1;
        case 'C': return 2;
        case 'G': return 3;
        case 'T': return 4;
        default: return 5;
    }
}

struct AlleleCount {
    int total = 0, fwd = 0, rev = 0;
};

struct VariantState {
    int64_t start, end, L;
    const char* ref;
    int64_t ref_len;
    std::vector<int> image;                   // [(L + 1) * 26]
    std::vector<int> cov, snp, ins, del;
    std::vector<std::map<std::string, AlleleCount>> alleles;   // ordered = the reference's std::set order
    int& px(int64_t row, int col) { return image[(size_t)row * 26 + col]; }
    char refc(int64_t idx) const { return idx >= 0 && idx < ref_len ? ref[idx] : 'N'; }
    void vote(int64_t idx, const std::string& key, bool reverse) {
        AlleleCount& a = alleles[(size_t)idx][key];
        a.total += 1;
        (reverse ? a.rev : a.fwd) += 1;
    }
};

void walk_read(VariantState& st, const oracle_pileup* p, int r, double min_snp_q, double min_indel_q) {
    const bool rev = p->read_reverse[r] != 0;
    const char* seq = p->seq + p->seq_offset[r];
    const uint8_t* q = p->qual + p->seq_offset[r];
    const int64_t read_len = p->seq_offset[r + 1] - p->seq_offset[r];
    const int64_t c0 = p->cigar_offset[r], c1 = p->cigar_offset[r + 1];
    int64_t ri = 0, pos = p->read_pos[r];
    for (int64_t c = c0; c < c1; ++c) {
        if (pos > st.end) break;
        const int op = p->cigar_op[c];
        const int64_t len = p->cigar_len[c];
        if (op == OP_M || op == OP_EQ || op == OP_X) {
            int64_t i = 0;
            if (pos < st.start) {
                i = std::min<int64_t>(st.start - pos, len);
                ri += i;
                pos += i;
            }
            for (; i < len; ++i, ++ri, ++pos) {
                if (pos < st.start || pos > st.end) continue;
                const int64_t idx = pos - st.start;
                const char base = seq[ri], rb = st.refc(idx);
                const bool good = (double)q[ri] >= min_snp_q;
                const int col = symbol_column(rb, base, rev);
                if (good) {
                    st.cov[(size_t)idx] += 1;
                    bool anchors_indel = false;
                    if (i == len - 1 && c != c1 - 1) {
                        const int nop = p->cigar_op[c + 1];
                        anchors_indel = (nop == OP_I || nop == OP_D);
                    }
                    if (!anchors_indel) st.px(idx, rev ? 15 : 4) -= 1;
                    if (col >= 0) st.px(idx, col) -= 1;
                    if (rb != base) {
                        st.snp[(size_t)idx] += 1;
                        st.vote(idx, std::string("1") + base, rev);
                    }
                }
            }
        } else if (op == OP_I) {
            const int64_t anchor = pos - 1;
            if (anchor >= st.start && anchor <= st.end && ri - 1 >= 0) {
                const int64_t idx = anchor - st.start;
                const char rb = st.refc(idx);
                const int64_t n = len + 1;
                const int64_t avail = std::max<int64_t>(0, std::min<int64_t>(n, read_len - (ri - 1)));
                const std::string alt(seq + (ri - 1), (size_t)avail);
                double qsum = 0;
                for (int64_t k = ri - 1; k < ri - 1 + n; ++k) qsum += (k < read_len) ? q[k] : 0;
                const bool passes = qsum >= min_indel_q * (double)n;
                if (passes && (double)q[ri - 1] < min_snp_q) st.cov[(size_t)idx] += 1;
                const std::string key = "2" + alt;
                if (key.size() <= 61 && passes) {
                    const int col = symbol_column(rb, 'I', rev);
                    if (col >= 0) st.px(idx, col) -= 1;
                    st.ins[(size_t)idx] += 1;
                    st.vote(idx, key, rev);
                }
            }
            ri += len;
        } else if (op == OP_D) {
            const int64_t anchor = pos - 1;
            if (anchor >= st.start && anchor <= st.end) {
                const int64_t idx = anchor - st.start;
                const int col = symbol_column(st.refc(idx), 'D', rev);
                if (col >= 0) st.px(idx, col) -= 1;
                const int64_t avail = std::max<int64_t>(0, std::min<int64_t>(len + 1, st.ref_len - idx));
                const std::string key = "3" + std::string(st.ref + idx, (size_t)avail);
                if (key.size() <= 61) {
                    st.del[(size_t)idx] += 1;
                    st.vote(idx, key, rev);
                }
            }
            for (int64_t i = 0; i < len; ++i) {
                const int64_t g = pos + i;
                if (g < st.start || g > st.end) continue;
                const int col = symbol_column(st.refc(g - st.start), '*', rev);
                if (col >= 0) st.px(g - st.start, col) -= 1;
            }
            pos += len;
        } else if (op == OP_N || op == OP_P) {
            pos += len;
            ri += len;          // fall-through into the soft-clip case in the reference
        } else if (op == OP_S) {
            ri += len;
        }
    }
}

}  // namespace

extern "C" {

int oracle_variant_generate_summary(const oracle_pileup* p, const oracle_summary_params* q,
                                    oracle_summary_result* out) {
    VariantState st;
    st.start = p->region_start;
    st.end = p->region_end;
    st.L = st.end - st.start + 1;
    st.ref = p->reference;
    st.ref_len = p->reference_len;
    st.image.assign((size_t)(st.L + 1) * 26, 0);
    st.cov.assign((size_t)st.L, 0);
    st.snp.assign((size_t)st.L, 0);
    st.ins.assign((size_t)st.L, 0);
    st.del.assign((size_t)st.L, 0);
    st.alleles.resize((size_t)st.L + 1);
    for (int64_t i = 0; i < st.L; ++i) st.px(i, 0) = base_code(st.refc(i));
    for (int r = 0; r < p->n_reads; ++r)
        if (p->read_mapq[r] > 0) walk_read(st, p, r, q->min_snp_baseq, q->min_indel_baseq);

    std::vector<int64_t> sites;
    std::vector<uint8_t> pass((size_t)st.L, 0);   // bit0 snp, bit1 insert, bit2 delete
    for (int64_t i = 0; i < st.L; ++i) {
        const double c = std::max(1.0, (double)st.cov[(size_t)i]);
        const bool s = st.snp[(size_t)i] / c >= q->snp_freq_threshold;
        const bool n = st.ins[(size_t)i] / c >= q->insert_freq_threshold;
        const bool d = st.del[(size_t)i] / c >= q->delete_freq_threshold;
        const int64_t gpos = st.start + i;
        if ((s || n || d) && gpos >= q->candidate_region_start && gpos <= q->candidate_region_end &&
            (double)st.cov[(size_t)i] >= q->min_coverage_threshold) {
            sites.push_back(i);
            pass[(size_t)i] = (uint8_t)((s ? 1 : 0) | (n ? 2 : 0) | (d ? 4 : 0));
        }
        for (int col = 11; col < 25; ++col) st.px(i, col) = std::max(-kMaxColor, std::min(kMaxColor, st.px(i, col)));
    }

    const int W = q->candidate_window_size + 1, F = q->feature_size, mid = q->candidate_window_size / 2;
    std::vector<int64_t> positions;
    std::vector<int32_t> depths, freqs, images;
    std::string names;
    for (int64_t idx : sites) {
        for (const auto& kv : st.alleles[(size_t)idx]) {
            const std::string& key = kv.first;
            const AlleleCount& a = kv.second;
            const int depth = std::min(st.cov[(size_t)idx], kMaxColor);
            const double freq = (double)a.total / std::max(1.0, (double)depth);
            const char type = key[0];
            if ((double)a.total < q->candidate_support_threshold) continue;
            if (type != '1' && freq < q->indel_candidate_freq_threshold) continue;
            if (type == '1' && freq < q->snp_candidate_freq_threshold) continue;
            if (type != '1' && q->skip_indels) continue;
            if ((type == '1' && !(pass[(size_t)idx] & 1)) || (type == '2' && !(pass[(size_t)idx] & 2)) ||
                (type == '3' && !(pass[(size_t)idx] & 4)))
                continue;
            const size_t base = images.size();
            images.resize(base + (size_t)W * F, 0);
            int32_t* img = images.data() + base;
            for (int r = 0; r < W; ++r) {
                const int64_t row = idx - mid + r;
                if (row < 0 || row > st.L) continue;                 // row L exists (all zero)
                for (int f = 0; f < F && f < 26; ++f) img[r * F + f] = st.px(row, f);
            }
            const char rb = st.refc(idx);
            const int fwd = std::min(a.fwd, kMaxColor), rv = std::min(a.rev, kMaxColor);
            auto negate = [&](int row, int col) { if (col >= 0) img[row * F + col] = -img[row * F + col]; };
            if (type == '1') {
                img[mid * F + 1] = base_code(key[1]);
                img[mid * F + 5] = fwd;
                img[mid * F + 16] = rv;
                negate(mid, symbol_column(rb, key[1], false));
                negate(mid, symbol_column(rb, key[1], true));
            } else if (type == '2') {
                img[mid * F + 2] = std::min((int)key.size() - 1, kMaxColor);
                img[mid * F + 6] = fwd;
                img[mid * F + 17] = rv;
                negate(mid, symbol_column(rb, 'I', false));
                negate(mid, symbol_column(rb, 'I', true));
            } else {
                const int dlen = (int)key.size() - 1;
                const int last = std::min(mid + dlen - 1, q->candidate_window_size - 1);
                img[mid * F + 3] = std::min(dlen, kMaxColor);
                img[mid * F + 7] = fwd;
                img[mid * F + 18] = rv;
                negate(mid, symbol_column(rb, 'D', false));
                negate(mid, symbol_column(rb, 'D', true));
                for (int r = mid + 1; r <= last; ++r) {
                    img[r * F + 3] = std::min(dlen, kMaxColor);
                    img[r * F + 7] = fwd;
                    img[r * F + 18] = rv;
                    negate(r, symbol_column(rb, '*', false));
                    negate(r, symbol_column(rb, '*', true));
                }
            }
            positions.push_back(st.start + idx);
            depths.push_back(depth);
            freqs.push_back(std::min(a.total, kMaxColor));
            names += key;
            names.push_back('\0');
        }
    }
    const int64_t n = (int64_t)positions.size();
    out->n = n;
    out->positions = new int64_t[(size_t)n + 1];
    out->depths = new int32_t[(size_t)n + 1];
    out->candidate_frequency = new int32_t[(size_t)n + 1];
    out->images = new int32_t[images.size() + 1];
    out->candidates = new char[names.size() + 1];
    out->candidates_bytes = (int64_t)names.size();
    std::copy(positions.begin(), positions.end(), out->positions);
    std::copy(depths.begin(), depths.end(), out->depths);
    std::copy(freqs.begin(), freqs.end(), out->candidate_frequency);
    std::copy(images.begin(), images.end(), out->images);
    std::memcpy(out->candidates, names.data(), names.size());
    return 0;
}

void oracle_free_summary(oracle_summary_result* r) {
    delete[] r->positions;
    delete[] r->depths;
    delete[] r->candidate_frequency;
    delete[] r->images;
    delete[] r->candidates;
    std::memset(r, 0, sizeof(*r));
}

// ---- polish ------------------------------------------------------------------------------------
// rows: out_image uint8 [rows][10], out_pos int64 [rows][2] = (position, insert index); returns rows
// (call with null outputs to size).  start_pos/end_pos = generate_summary's arguments; the pileup's
// region_start/end = the constructor's ref_start/ref_end.
int64_t oracle_polish_generate_summary(const oracle_pileup* p, int64_t start_pos, int64_t end_pos,
                                       uint8_t* out_image, int64_t* out_pos, int64_t cap_rows) {
    auto col = [](char b, bool rev) {   // summary_generator.cpp:16-32
        int k;
        switch (std::toupper((unsigned char)b)) {
            case 'A': k = 0; break;
            case 'C': k = 1; break;
            case 'G': k = 2; break;
            case 'T': k = 3; break;
            default: return rev ? 8 : 9;
        }
        return rev ? k : k + 4;
    };
    std::map<std::pair<int64_t, int>, double> base_sum;
    std::map<std::pair<std::pair<int64_t, int>, int>, double> ins_sum;
    std::map<int64_t, int64_t> longest;
    std::map<int64_t, double> cov;
    for (int r = 0; r < p->n_reads; ++r) {
        if (p->read_mapq[r] <= 0) continue;
        const bool rev = p->read_reverse[r] != 0;
        const char* seq = p->seq + p->seq_offset[r];
        int64_t ri = 0, pos = p->read_pos[r];
        for (int64_t c = p->cigar_offset[r]; c < p->cigar_offset[r + 1]; ++c) {
            if (pos > end_pos) break;
            const int op = p->cigar_op[c];
            const int64_t len = p->cigar_len[c];
            if (op == OP_M || op == OP_EQ || op == OP_X) {
                int64_t i = 0;
                if (pos < p->region_start) {
                    i = std::min<int64_t>(p->region_start - pos, len);
                    ri += i;
                    pos += i;
                }
                for (; i < len; ++i, ++ri, ++pos)
                    if (pos >= p->region_start && pos <= p->region_end) {
                        base_sum[{pos, col(seq[ri], rev)}] += 1.0;
                        cov[pos] += 1.0;
                    }
            } else if (op == OP_I) {
                const int64_t anchor = pos - 1;
                if (anchor >= p->region_start && anchor <= p->region_end) {
                    for (int64_t i = 0; i < len; ++i) ins_sum[{{anchor, (int)i}, col(seq[ri + i], rev)}] += 1.0;
                    longest[anchor] = std::max(longest[anchor], len);
                }
                ri += len;
            } else if (op == OP_D || op == OP_N || op == OP_P) {
                for (int64_t i = 0; i < len; ++i)
                    if (pos + i >= p->region_start && pos + i <= p->region_end) {
                        base_sum[{pos + i, col('*', rev)}] += 1.0;
                        cov[pos] += 1.0;           // credited to the deletion start, as the reference does
                    }
                pos += len;
            } else if (op == OP_S) {
                ri += len;
            }
        }
    }
    int64_t rows = 0;
    auto emit = [&](int64_t pos, int idx, const double* counts, double c) {
        if (out_image && rows < cap_rows)
            for (int j = 0; j < 10; ++j) {
                // double -> uint8_t; out-of-range values (gap counts over zero coverage) are UB in
                // C++ -- x86-64 gcc truncates through a 32-bit integer, which is what is mimicked
                const double v = (counts[j] / std::max(1.0, c)) * 254;
                out_image[rows * 10 + j] = (uint8_t)((int64_t)v & 0xff);
            }
        if (out_pos && rows < cap_rows) {
            out_pos[rows * 2] = pos;
            out_pos[rows * 2 + 1] = idx;
        }
        ++rows;
    };
    for (int64_t pos = start_pos; pos <= end_pos; ++pos) {
        double counts[10];
        const auto cit = cov.find(pos);
        const double c = cit == cov.end() ? 0.0 : cit->second;
        for (int j = 0; j < 10; ++j) {
            const auto it = base_sum.find({pos, j});
            counts[j] = it == base_sum.end() ? 0.0 : it->second;
        }
        emit(pos, 0, counts, c);
        const auto lit = longest.find(pos);
        const int64_t n_ins = lit == longest.end() ? 0 : lit->second;
        for (int64_t ii = 0; ii < n_ins; ++ii) {
            for (int j = 0; j < 10; ++j) {
                const auto it = ins_sum.find({{pos, (int)ii}, j});
                counts[j] = it == ins_sum.end() ? 0.0 : it->second;
            }
            emit(pos, (int)ii + 1, counts, c);
        }
    }
    return rows;
}

}  // extern "C"
