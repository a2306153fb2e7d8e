/* pepper_amd encoder C ABI -- pileup -> candidate summary images on MI355X.
 *
 * Replaces the pybind11 surface of the reference's variant encoder:
 *   PEPPER_VARIANT.RegionalSummaryGenerator(contig, region_start, region_end, reference_sequence)
 *     .generate_max_insert_summary(reads)
 *     .generate_summary(reads, min_snp_baseq, ..., candidate_window_size, feature_size, train_mode)
 *       -> list[CandidateImageSummary]
 *   pepper_variant/modules/cpp/pybind_api.h:55-62,73-101; region_summary.h:88-111,159-206;
 *   implementation region_summary.cpp:69-96 (axes), 174-191 (reference row), 337-566 (per-read
 *   walk), 568-916 (thresholds, candidate windows).
 * Reads arrive as flat arrays (the fields of type_read / CigarOp, read.h:52-64, cigar.h:30-53)
 * instead of per-read Python objects.  The per-base counting, the threshold/clamp pass and the
 * candidate window gather run as HIP kernels; the allele-string bookkeeping (ordered maps of
 * candidate strings) stays on the host, as in SURVEY.md section 7 step 7.
 */
#ifndef PEPPER_AMD_ENCODER_H
#define PEPPER_AMD_ENCODER_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int64_t region_start, region_end;   /* generator's ref_start / ref_end (inclusive)               */
    const char* reference;              /* reference_sequence covering [region_start, region_end]    */
    int64_t reference_len;
    int32_t n_reads;
    const int64_t* read_pos;            /* type_read.pos                                              */
    const uint8_t* read_reverse;        /* type_read.flags.is_reverse                                 */
    const int32_t* read_mapq;           /* type_read.mapping_quality (reads with mapq <= 0 are skipped) */
    const int64_t* seq_offset;          /* [n_reads+1] offsets into seq / qual                        */
    const char* seq;                    /* concatenated type_read.sequence                            */
    const uint8_t* qual;                /* concatenated type_read.base_qualities                      */
    const int64_t* cigar_offset;        /* [n_reads+1] offsets into cigar_op / cigar_len              */
    const int32_t* cigar_op;            /* CIGAR_OPERATIONS codes (cigar.h:17-27)                     */
    const int32_t* cigar_len;
} pa_pileup;

typedef struct {                        /* arguments of generate_summary, region_summary.h:191-206    */
    double min_snp_baseq, min_indel_baseq;
    double snp_freq_threshold, insert_freq_threshold, delete_freq_threshold;
    double min_coverage_threshold;
    double snp_candidate_freq_threshold, indel_candidate_freq_threshold, candidate_support_threshold;
    int32_t skip_indels;
    int64_t candidate_region_start, candidate_region_end;
    int32_t candidate_window_size;      /* ImageSizeOptions.CANDIDATE_WINDOW_SIZE = 32                */
    int32_t feature_size;               /* ImageSizeOptions.IMAGE_HEIGHT = 26                         */
} pa_summary_params;

typedef struct pa_encoder pa_encoder;

/* One encoder per (thread, GPU): owns a stream (or uses hip_stream) and reusable workspace. */
int pa_encoder_create(int32_t device, void* hip_stream, pa_encoder** out);
void pa_encoder_destroy(pa_encoder* e);

/* Encode one region.  On success *n_candidates = number of CandidateImageSummary the reference
 * would return (train_mode=False); results stay in the handle until the next call. */
int pa_encoder_generate_summary(pa_encoder* e, const pa_pileup* pileup, const pa_summary_params* params,
                                int64_t* n_candidates);

/* Copy results of the last call (HOST pointers, any may be NULL):
 *   positions int64 [n], depths int32 [n], candidate_frequency int32 [n]  (CandidateImageSummary
 *   .position / .depth / .candidate_frequency[0]); images_i32 [n, window+1, feature] = image_matrix;
 *   images_i8 = the same values wrapped to int8 exactly as DataStore.py:68 stores them;
 *   candidates: n NUL-terminated allele strings (.candidates[0]); *candidates_needed = bytes. */
int pa_encoder_get_results(pa_encoder* e, int64_t* positions, int32_t* depths, int32_t* candidate_frequency,
                           int32_t* images_i32, int8_t* images_i8, char* candidates, int64_t candidates_cap,
                           int64_t* candidates_needed);

/* Device pointer to the int8 images of the last call ([n, window+1, feature], valid until the
 * next call) so inference can consume them without a host round trip. */
const int8_t* pa_encoder_device_images(pa_encoder* e);

/* ------------------------------------------------------------------------------------------
 * Polish summary encoder
 * replaces: PEPPER.SummaryGenerator(ref_seq, chr, start, end).generate_summary(reads, start, end)
 *   -> .image (uint8 rows of 10 features), .genomic_pos ((position, insert index) per row)
 *   pepper/modules/headers/pybind_api.h:18-25; pepper/modules/src/pileup_summary/summary_generator.cpp:
 *   16-32 (feature index), 47-121 (per-read walk), 274-306 (pixels), 370-393 (row order).
 * pileup->region_start/end = the constructor's ref_start/ref_end; start_pos/end_pos = the
 * arguments of generate_summary (identical in the reference's caller, AlignmentSummarizer.py:340-347).
 * ------------------------------------------------------------------------------------------ */
int pa_polish_encoder_generate_summary(pa_encoder* e, const pa_pileup* pileup, int64_t start_pos,
                                       int64_t end_pos, int64_t* n_rows);
/* HOST pointers: image uint8 [n_rows, 10], positions int64 [n_rows, 2]; either may be NULL. */
int pa_polish_encoder_get_results(pa_encoder* e, uint8_t* image, int64_t* positions);

#ifdef __cplusplus
}
#endif
#endif /* PEPPER_AMD_ENCODER_H */
