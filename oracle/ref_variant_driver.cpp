// ORACLE (test infrastructure): C-ABI driver around the REFERENCE's own
// RegionalSummaryGenerator, compiled from the sources where they lie under
// /root/reference/pepper_variant/modules/cpp (see oracle/Makefile).  Nothing of the reference is
// copied into this repository: this translation unit only #includes its files and marshals
// flat arrays into its types.  Output: oracle/_ref/libref_variant_encoder.so (git-ignored).
//
// region_summary.cpp is written for a unity build (pybind_api.cpp #includes every .cpp) and takes
// AlleleType::{SNP,INSERT,DELETE}_ALLELE from candidate_finder.h:23-27, a header that also pulls in
// htslib (absent from this image).  The Makefile extracts that one namespace block verbatim into
// oracle/_ref/allele_type_extract.h at build time; no stand-in is written for it.
#include <cstring>
#include <iomanip>
#include <map>
#include <set>
#include <string>
#include <vector>

#include "read.h"                  // reference: type_read, CigarOp (uses namespace std)
#include "allele_type_extract.h"   // generated from candidate_finder.h by the Makefile
#include "region_summary.cpp"      // reference implementation, compiled as-is

#include "pileup_abi.h"

extern "C" {

int ref_variant_generate_summary(const oracle_pileup* p, const oracle_summary_params* q,
                                 oracle_summary_result* out) {
    std::vector<type_read> reads((size_t)p->n_reads);
    for (int32_t i = 0; i < p->n_reads; ++i) {
        type_read& r = reads[(size_t)i];
        r.pos = p->read_pos[i];
        r.pos_end = p->read_pos[i];
        r.flags.is_reverse = p->read_reverse[i] != 0;
        r.mapping_quality = p->read_mapq[i];
        r.hp_tag = 0;
        r.read_id = i;
        const int64_t s0 = p->seq_offset[i], s1 = p->seq_offset[i + 1];
        r.sequence.assign(p->seq + s0, (size_t)(s1 - s0));
        r.base_qualities.assign(p->qual + s0, p->qual + s1);
        for (int64_t c = p->cigar_offset[i]; c < p->cigar_offset[i + 1]; ++c)
            r.cigar_tuples.emplace_back(p->cigar_op[c], p->cigar_len[c]);
    }
    RegionalSummaryGenerator gen("contig", p->region_start, p->region_end,
                                 std::string(p->reference, (size_t)p->reference_len));
    gen.generate_max_insert_summary(reads);
    std::vector<CandidateImageSummary> cands = gen.generate_summary(
        reads, q->min_snp_baseq, q->min_indel_baseq, q->snp_freq_threshold, q->insert_freq_threshold,
        q->delete_freq_threshold, q->min_coverage_threshold, q->snp_candidate_freq_threshold,
        q->indel_candidate_freq_threshold, q->candidate_support_threshold, q->skip_indels != 0,
        q->candidate_region_start, q->candidate_region_end, q->candidate_window_size, q->feature_size, false);

    const int64_t n = (int64_t)cands.size();
    const int W = q->candidate_window_size + 1, F = q->feature_size;
    out->n = n;
    out->positions = new int64_t[(size_t)n + 1];
    out->depths = new int32_t[(size_t)n + 1];
    out->candidate_frequency = new int32_t[(size_t)n + 1];
    out->images = new int32_t[(size_t)n * W * F + 1];
    std::string names;
    for (int64_t i = 0; i < n; ++i) {
        const CandidateImageSummary& c = cands[(size_t)i];
        out->positions[i] = c.position;
        out->depths[i] = c.depth;
        out->candidate_frequency[i] = c.candidate_frequency.empty() ? -1 : c.candidate_frequency[0];
        for (int r = 0; r < W; ++r)
            for (int f = 0; f < F; ++f) out->images[((size_t)i * W + r) * F + f] = c.image_matrix[r][f];
        names += c.candidates.empty() ? std::string() : c.candidates[0];
        names.push_back('\0');
    }
    out->candidates_bytes = (int64_t)names.size();
    out->candidates = new char[names.size() + 1];
    std::memcpy(out->candidates, names.data(), names.size());
    return 0;
}

void ref_variant_free(oracle_summary_result* r) {
    delete[] r->positions;
    delete[] r->depths;
    delete[] r->candidate_frequency;
    delete[] r->images;
    delete[] r->candidates;
    std::memset(r, 0, sizeof(*r));
}

}  // extern "C"
