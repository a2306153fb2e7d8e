// Candidate selection and VCF record text for one prediction batch (include/pepper_amd_io.h, pa_candidates_select_format).
// Host-only C++.  The rules are those of the reference's candidate finder for a row whose candidate list holds ONE allele
// (pepper_variant/modules/python/CandidateFinder.py:356-581 small_chunk_stitch -> find_candidates) and of its VCF writer for
// a site with one allele record (VcfWriter.py:48-218); pepper_amd/variant/FastCandidates.py holds the same rules as Python
// loops (_select_batch, _format_single: ~5 us per candidate) and keeps them as the form the tests hold this one to, byte
// for byte.  Number formatting follows CPython exactly: '%g' % x is C's %g; round(x, 3) is the correctly rounded decimal of
// the double (glibc's "%.3f" rounds the exact binary value the same way) read back with strtod; struct.pack('f') is the
// round-to-nearest-even float conversion; int() truncates.
#include "../../include/pepper_amd_io.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

void pa_h5_set_error(const std::string& msg);      // hdf5io.cpp

namespace {

inline bool is_base(unsigned char c) { return c == 'A' || c == 'C' || c == 'G' || c == 'T'; }

inline char* put_int(char* w, long long v) {
    char tmp[24];
    int n = 0;
    const bool neg = v < 0;
    unsigned long long u = neg ? 0ull - (unsigned long long)v : (unsigned long long)v;
    do { tmp[n++] = (char)('0' + u % 10); u /= 10; } while (u);
    if (neg) *w++ = '-';
    while (n) *w++ = tmp[--n];
    return w;
}

inline char* put_str(char* w, const char* s, size_t n) {
    std::memcpy(w, s, n);
    return w + n;
}

}  // namespace

extern "C" {

int pa_candidates_reference_flags(const char* text, int64_t text_len, int64_t window_lo, int64_t n, const int64_t* position,
                                  uint8_t* letters, uint8_t* in_repeat) {
    if (text_len < 0 || n < 0 || (text_len > 0 && !text) || (n > 0 && (!position || !letters || !in_repeat))) {
        pa_h5_set_error("bad argument");
        return -1;
    }
    // Per position, the 20-base context ref[p - 10, p + 10) cut at the contig's start and at the window's end, upper-cased: a
    // homopolymer run of >= 5 INSIDE it that touches [p - 5, p + 4) (CandidateFinder.py:397-418).  The context is scanned where
    // it lies (20 characters per position; run tables over the whole window -- tens of kilobases for 512 positions -- were most
    // of a batch's selection time).
    auto up = [&](int64_t i) -> unsigned char {
        const unsigned char c = (unsigned char)text[i];
        return (unsigned char)((c >= 'a' && c <= 'z') ? c - 32 : c);
    };
    for (int64_t r = 0; r < n; ++r) {
        const int64_t p = position[r], q = p - window_lo;
        const bool inside = q >= 0 && q < text_len;
        letters[r] = inside ? (uint8_t)up(q) : 0;
        bool flag = false;
        if (inside) {
            int64_t ctx_lo = (p - 10 > 0 ? p - 10 : 0) - window_lo;
            if (ctx_lo < 0) ctx_lo = 0;
            const int64_t ctx_hi = q + 10 < text_len ? q + 10 : text_len;
            const int64_t touch_lo = q - 5 > ctx_lo ? q - 5 : ctx_lo, touch_hi = q + 4 < ctx_hi ? q + 4 : ctx_hi;   // [touch_lo, touch_hi)
            int64_t run_start = ctx_lo;
            for (int64_t i = ctx_lo; i < ctx_hi && !flag; ++i) {
                const bool last_of_run = i + 1 == ctx_hi || up(i + 1) != up(i);
                if (last_of_run) {
                    // the run [run_start, i + 1): long enough and sharing an index with the touched stretch
                    if (i + 1 - run_start >= 5 && run_start < touch_hi && i + 1 > touch_lo) flag = true;
                    run_start = i + 1;
                }
            }
        }
        in_repeat[r] = flag ? 1 : 0;
    }
    return 0;
}

int64_t pa_candidates_select_format(const pa_candidate_rules* rules, const char* contig, int64_t n, const int64_t* position,
                                    const int64_t* depth, const int64_t* support, const float* prediction,
                                    const uint8_t* reference_base, const uint8_t* in_repeat, const char* alleles,
                                    const int64_t* allele_offsets, int32_t separator_bytes, int32_t* kept_row, int32_t* ref_len,
                                    uint8_t* flags, char* lines, int64_t lines_cap, int64_t* line_offsets) {
    if (!rules || !contig || n < 0 ||
        (n > 0 && (!position || !depth || !support || !prediction || !reference_base || !in_repeat || !alleles ||
                   !allele_offsets || !kept_row || !ref_len || !flags || !lines || !line_offsets))) {
        pa_h5_set_error("bad argument");
        return -1;
    }
    static const char* const GT_TEXT[3] = {"0/0", "0/1", "1/1"};
    static const char FORMAT[] = "GT:AP:GQ:DP:AD:VAF:REP";
    const size_t contig_len = std::strlen(contig);
    int64_t m = 0;
    char* w = lines;
    char* const end = lines + lines_cap;
    if (line_offsets) line_offsets[0] = 0;
    for (int64_t i = 0; i < n; ++i) {
        const unsigned char rb = reference_base[i];
        if (!is_base(rb)) continue;                                    // (one upper-cased character or "")
        const char* code = alleles + allele_offsets[i];
        const int64_t code_len = allele_offsets[i + 1] - allele_offsets[i] - separator_bytes;
        if (code_len < 1) continue;
        const int kind = code[0] - '1';                                // '1' SNP, '2' insert, '3' delete
        const char* allele = code + 1;
        const int64_t allele_len = code_len - 1;
        bool plain = true;
        for (int64_t k = 0; k < allele_len; ++k) plain = plain && is_base((unsigned char)allele[k]);
        if (!plain) continue;
        // the reference divides support by depth for every valid allele, whatever its type (CandidateFinder.py:478): float
        // division by zero there, so the batch goes to the caller's Python path, which raises it
        if (depth[i] == 0) return -2;
        if (kind < 0 || kind > 2) continue;
        const float p0 = prediction[3 * i], p1 = prediction[3 * i + 1], p2 = prediction[3 * i + 2];
        if (std::isnan(p0) || std::isnan(p1) || std::isnan(p2)) return -2;     // numpy's argmax / maximum rules for NaN: the caller's Python path
        const int g = (p1 > p0) ? ((p2 > p1) ? 2 : 1) : ((p2 > p0) ? 2 : 0);   // first maximum
        const double pv = g == 0 ? p0 : (g == 1 ? p1 : p2);
        const double non_alt = p1 > p2 ? p1 : p2;
        const bool rep = in_repeat[i] != 0;
        const bool by_probability = non_alt >= (rep ? rules->p_value_in_lc[kind] : rules->p_value[kind]);
        if (!by_probability) {
            const double above = rules->report_above_freq[kind];
            if (!(0 < above)) continue;
            if (!(above <= (double)support[i] / (double)depth[i])) continue;
        }
        // a deletion swaps roles: the deleted stretch is REF, the anchor base ALT (CandidateFinder.py:490-501)
        const bool swap = kind == 2 && by_probability;
        const char* ref = swap ? allele : reinterpret_cast<const char*>(&reference_base[i]);
        const int64_t rlen = swap ? allele_len : 1;
        const char* alt = swap ? reinterpret_cast<const char*>(&reference_base[i]) : allele;
        const int64_t alen = swap ? 1 : allele_len;
        // genotype quality (VcfWriter.py:83-90 with one candidate), QUAL (:153)
        const double gq = g != 0 ? pv : non_alt;
        double rest = 1.0 - gq;
        if (!(rest > 0.000000001)) rest = 0.000000001;
        long long qual = (long long)(-10 * std::log10(rest));
        if (qual < 1) qual = 1;
        const bool is_snp = (rlen > alen ? rlen : alen) == 1;
        const double cutoff = is_snp ? (rep ? rules->snp_q_cutoff_in_lc : rules->snp_q_cutoff)
                                     : (rep ? rules->indel_q_cutoff_in_lc : rules->indel_q_cutoff);
        if (end - w < (int64_t)(contig_len + rlen + alen + 192)) {
            pa_h5_set_error("line buffer too small");
            return -1;
        }
        char num[64];
        w = put_str(w, contig, contig_len);
        *w++ = '\t';
        w = put_int(w, position[i] + 1);
        w = put_str(w, "\t.\t", 3);
        w = put_str(w, ref, (size_t)rlen);
        *w++ = '\t';
        w = put_str(w, alt, (size_t)alen);
        *w++ = '\t';
        w = put_int(w, qual);
        *w++ = '\t';
        w = g == 0 ? put_str(w, "refCall", 7) : put_str(w, "PASS", 4);
        w = put_str(w, "\t.\t", 3);
        w = put_str(w, FORMAT, sizeof FORMAT - 1);
        *w++ = '\t';
        w = put_str(w, GT_TEXT[g], 3);
        *w++ = ':';
        w = put_str(w, num, (size_t)std::snprintf(num, sizeof num, "%g", non_alt));
        *w++ = ':';
        w = put_int(w, qual);
        *w++ = ':';
        w = put_int(w, depth[i]);
        *w++ = ':';
        w = put_int(w, support[i]);
        *w++ = ':';
        const double ratio = (double)support[i] / (double)(depth[i] > 1 ? depth[i] : 1);
        std::snprintf(num, sizeof num, "%.3f", ratio);                 // round(x, 3) ...
        const float vaf = (float)std::strtod(num, nullptr);            // ... through struct.pack('f')
        w = put_str(w, num, (size_t)std::snprintf(num, sizeof num, "%g", (double)vaf));
        *w++ = ':';
        *w++ = rep ? '1' : '0';
        *w++ = '\n';
        kept_row[m] = (int32_t)i;
        ref_len[m] = (int32_t)rlen;
        flags[m] = (uint8_t)((is_snp ? 1 : 0) | ((g == 0 || (double)qual <= cutoff) ? 2 : 0) | (swap ? 4 : 0) | (g << 4));
        ++m;
        line_offsets[m] = w - lines;
    }
    return m;
}

}  // extern "C"
