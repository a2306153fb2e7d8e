/* Flat pileup representation shared by the oracle drivers (test infrastructure) -- the same field
 * layout as the product ABI in include/pepper_amd_encoder.h so one ctypes marshaller feeds both.
 * Mirrors type_read / CigarOp (pepper_variant/modules/cpp/read.h:52-64, cigar.h:30-53). */
#ifndef ORACLE_PILEUP_ABI_H
#define ORACLE_PILEUP_ABI_H
#include <stdint.h>

typedef struct {
    int64_t region_start, region_end;   /* ref_start, ref_end (inclusive) of the generator            */
    const char* reference;              /* reference_sequence, length >= region_end-region_start+1     */
    int64_t reference_len;
    int32_t n_reads;
    const int64_t* read_pos;            /* type_read.pos                                               */
    const uint8_t* read_reverse;        /* type_read.flags.is_reverse                                  */
    const int32_t* read_mapq;           /* type_read.mapping_quality                                   */
    const int64_t* seq_offset;          /* [n_reads+1] into seq / qual                                 */
    const char* seq;                    /* concatenated type_read.sequence                             */
    const uint8_t* qual;                /* concatenated type_read.base_qualities                       */
    const int64_t* cigar_offset;        /* [n_reads+1] into cigar_op / cigar_len                       */
    const int32_t* cigar_op;            /* CIGAR_OPERATIONS codes 0..9 (cigar.h:17-27)                 */
    const int32_t* cigar_len;
} oracle_pileup;

typedef struct {                        /* generate_summary arguments, region_summary.h:191-206        */
    double min_snp_baseq, min_indel_baseq;
    double snp_freq_threshold, insert_freq_threshold, delete_freq_threshold;
    double min_coverage_threshold;
    double snp_candidate_freq_threshold, indel_candidate_freq_threshold, candidate_support_threshold;
    int32_t skip_indels;
    int64_t candidate_region_start, candidate_region_end;
    int32_t candidate_window_size, feature_size;
} oracle_summary_params;

/* Result: N candidates; images int32 [N][window+1][feature]; strings NUL-separated. */
typedef struct {
    int64_t n;
    int64_t* positions;
    int32_t* depths;
    int32_t* candidate_frequency;
    int32_t* images;
    char* candidates;        /* N NUL-terminated allele strings, concatenated */
    int64_t candidates_bytes;
} oracle_summary_result;

#endif
