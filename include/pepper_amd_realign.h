/* pepper_amd read re-aligner C ABI -- local re-alignment of the reads of one region on MI355X.
 *
 * Replaces the pybind11 surface of the reference's polish re-aligner:
 *   PEPPER.ReadAligner(ref_start, ref_end, ref_seq).align_reads_to_reference(reads) -> list[type_read]
 *   pepper/modules/headers/pybind_api.h (ReadAligner), pepper/modules/src/local_reassembly/simple_aligner.cpp:60-106,
 *   called for every region by pepper/modules/python/AlignmentSummarizer.py:159-177,328-332 (realignment_flag
 *   defaults to True) before the summary is generated.
 * Every read is aligned against the reference suffix that starts at its mapped position with the SSW library's
 * striped Smith-Waterman (match 4, mismatch 6, gap open 8, gap extend 2; ssw.c / ssw_cpp.cpp next to
 * simple_aligner.cpp) and, when the score is > 1, gets the new position, end position and CIGAR.  Here the three
 * stages of that library (score + end cell, reversed pass for the begin cell, banded trace-back) run as HIP kernels,
 * one wavefront per read, and reproduce its results cell for cell (tests/test_gpu_realign.py; the CPU restatement
 * and its pinning against the reference's own SSW build are oracle/ssw_oracle.cpp, tests/test_realign_oracle.py).
 * Reads arrive as flat arrays (type_read.pos, concatenated type_read.sequence), as for pepper_amd_encoder.h.
 * Return codes and pa_last_error() are those of pepper_amd.h.
 */
#ifndef PEPPER_AMD_REALIGN_H
#define PEPPER_AMD_REALIGN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pa_realigner pa_realigner;

/* One re-aligner per (thread, GPU): owns a stream (or uses hip_stream) and reusable workspace. */
int pa_realigner_create(int32_t device, void* hip_stream, pa_realigner** out);
void pa_realigner_destroy(pa_realigner* r);

/* status values */
#define PA_REALIGN_DROPPED (-1)   /* read.pos < region_start: the reference skips the read (simple_aligner.cpp:72-76) */
#define PA_REALIGN_KEPT 0         /* score <= 1 or nothing to align: the read is passed through unchanged            */
#define PA_REALIGN_ALIGNED 1      /* new position / end position / CIGAR                                              */

/* Align n_reads reads of one region.
 *   reference      ref_seq of ReadAligner: text covering [region_start, region_start + reference_len)
 *   read_pos       type_read.pos of each read
 *   seq_offset     [n_reads + 1] offsets into seq (concatenated type_read.sequence, text)
 * Outputs (host arrays of n_reads): status, sw_score, new_pos (= pos + ref_begin), new_pos_end (= pos + ref_end),
 * query_begin / query_end (0-based inclusive ends of the aligned part of the read; may be NULL).
 * *n_cigar_ops = number of CIGAR operations of all PA_REALIGN_ALIGNED reads; results stay in the handle until the
 * next call. */
int pa_realigner_align(pa_realigner* r, const char* reference, int64_t reference_len, int64_t region_start,
                       int32_t n_reads, const int64_t* read_pos, const int64_t* seq_offset, const char* seq,
                       int32_t* status, int32_t* sw_score, int64_t* new_pos, int64_t* new_pos_end,
                       int32_t* query_begin, int32_t* query_end, int64_t* n_cigar_ops);

/* The same for the reads of several regions in one call (one job table, one pair of launches: a region at ordinary
 * coverage is a few dozen wavefronts, far from filling the chip): `reference` holds the n_windows window texts back to
 * back, window w = reference[window_offset[w] .. window_offset[w + 1]) starting at genome position window_start[w];
 * read k belongs to window read_window[k] (NULL: all reads in window 0).  Outputs as above, in read order. */
int pa_realigner_align_windows(pa_realigner* r, int32_t n_windows, const char* reference, const int64_t* window_offset,
                               const int64_t* window_start, int32_t n_reads, const int32_t* read_window,
                               const int64_t* read_pos, const int64_t* seq_offset, const char* seq, int32_t* status,
                               int32_t* sw_score, int64_t* new_pos, int64_t* new_pos_end, int32_t* query_begin,
                               int32_t* query_end, int64_t* n_cigar_ops);

/* CIGARs of the last call: cigar_offset [n_reads + 1] (empty range for reads that were not aligned), operations in
 * BAM codes: 7 '=', 8 'X', 1 'I', 2 'D', 4 'S' (the text of Alignment.cigar_string).  With collapse_eqx != 0 the
 * codes 7 and 8 are returned as 0 (MATCH) without merging neighbouring runs, which is what
 * ReadAligner::CigarStringToVector produces (simple_aligner.cpp:32-58). */
int pa_realigner_copy_cigars(pa_realigner* r, int32_t collapse_eqx, int64_t* cigar_offset, int32_t* cigar_op,
                             int32_t* cigar_len);

/* Device time of the last pa_realigner_align call (HIP events on the handle's stream): the score / end / begin kernel,
 * the band + trace-back launches, and the number of DP cells (reference suffix length x read length, summed over the
 * aligned reads) one score pass visits.  Any pointer may be NULL. */
int pa_realigner_last_timing(pa_realigner* r, double* score_kernel_ms, double* band_kernel_ms, int64_t* cells);

/* Per-read stage times of the last call in 10 ns ticks, ticks4[4 k .. 4 k + 3] = score passes, band DP (all widths),
 * trace-back, CIGAR emission (tools/realign_stages.py). */
int pa_realigner_stage_ticks(pa_realigner* r, int32_t* ticks4);

#ifdef __cplusplus
}
#endif

#endif
