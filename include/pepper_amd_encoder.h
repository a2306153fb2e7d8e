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
 * instead of per-read Python objects.  The whole per-read walk, the threshold/clamp pass and the
 * candidate window gather run as HIP kernels; the allele-string bookkeeping (ordered maps of
 * candidate strings) stays on the host, as in SURVEY.md section 7 step 7.
 * MANY REGIONS PER CALL (pa_encoder_generate_summary_batch) is the form that fills the chip: one
 * workgroup owns one 512-position tile of one region, a 100 kb region has ~200 of them.
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

/* Many regions per launch: pileups[n_regions], params[n_regions] (one window size and one feature size per batch),
 * n_candidates[n_regions] (may be NULL).  Results are those of region 0, then region 1, ... in pa_encoder_get_results.
 * The buffers behind `pileups` (reference, seq) must stay valid until the call returns: candidate allele strings are
 * cut from them on the host. */
int pa_encoder_generate_summary_batch(pa_encoder* e, int32_t n_regions, const pa_pileup* pileups,
                                      const pa_summary_params* params, int64_t* n_candidates);
/* The two halves of the call above, for a caller that keeps a batch resident in HBM: stage = validate + upload,
 * run = kernels + host candidate enumeration + window gather (may be repeated; the pileup buffers must outlive the
 * last run).  bench.py times pa_encoder_run_staged. */
int pa_encoder_stage_batch(pa_encoder* e, int32_t n_regions, const pa_pileup* pileups, const pa_summary_params* params);
int pa_encoder_run_staged(pa_encoder* e, int64_t* n_candidates);
/* ------------------------------------------------------------------------------------------
 * The packed form: reads as BAM stores them, clipped and decoded ON THE DEVICE.
 * replaces, for the image generator: BAM_handler.get_reads per region (bam_handler.cpp:176-303, the walk that clips a
 * read to the region and decodes its bases) + the hand-over of the clipped reads above.  pa_bam_pack_regions
 * (include/pepper_amd_io.h) fills the arena and the tables for a run of regions; here they are uploaded with one copy each and
 * unpack_clip_kernel (one wave per (read, region)) produces what pa_encoder_stage_batch would have been given.  Results are
 * those of the host-clipped form bit for bit.  Nothing in the call waits for the device; pa_encoder_run_staged follows.
 *   arena        what pa_bam_pack_regions wrote (pa_encoder_host_arena returns a page-locked block of the handle for it:
 *                the upload is then asynchronous; any host memory works); NULL: the span pa_encoder_inflate_bgzf left on
 *                the device (data_off then need not be aligned)
 *   reads        n_reads table entries; pair_read[region_pairs[r] .. region_pairs[r + 1]) = the reads of region r
 *   regions      per region the generator's ref_start / ref_end (= the fetch range given to the packer) and its reference
 * The `reference` buffers must stay valid until the run returns (deleted bases of candidate alleles are cut from them).
 * An operation of 2^24 bases or more fails the run with PA_ERR_UNSUPPORTED: take the host-clipped form for that batch.
 * ------------------------------------------------------------------------------------------ */
#ifndef PA_PACKED_READ_DEFINED
#define PA_PACKED_READ_DEFINED
typedef struct {
    int64_t data_off;      /* in the arena: n_cigar uint32 (len << 4 | op), (l_seq + 1) / 2 bytes of 4-bit bases, l_seq qualities */
    int32_t pos;           /* 0-based leftmost position of the record */
    int32_t n_cigar;
    int32_t l_seq;
    int32_t flags;         /* BAM flag | mapping quality << 16 */
} pa_packed_read;
#endif
typedef struct {
    int64_t region_start, region_end;
    const char* reference;
    int64_t reference_len;
} pa_packed_region;
void* pa_encoder_host_arena(pa_encoder* e, int64_t bytes);      /* page-locked, grows, valid until the next call with more bytes */
int pa_encoder_stage_packed(pa_encoder* e, int32_t n_regions, const pa_packed_region* regions, const pa_summary_params* params,
                            const uint8_t* arena, int64_t arena_bytes, const pa_packed_read* reads, int32_t n_reads,
                            const int32_t* pair_read, const int32_t* region_pairs);
/* The arena filled ON THE DEVICE from the file's own bytes: the BGZF members of a span (pa_bam_read_span, include/
 * pepper_amd_io.h, fills `comp` -- pa_encoder_host_span returns a second page-locked block for it -- and the four tables) are
 * inflated into the encoder's device arena, one wavefront per member (csrc/inflate.hip; what htslib's bgzf_read_block does
 * beneath sam_itr_next, bam_handler.cpp:341-372), and copied to host_out (NULL: not) for the host's record walk
 * (pa_bam_pack_inflated, whose data_off are offsets into exactly these bytes).  pa_encoder_stage_packed with arena = NULL then
 * takes the bytes where they are.  A malformed member fails the call (PA_ERR_INVALID, the member and the reason in
 * pa_last_error), and so does a member whose inflated bytes do not have the CRC-32 of its trailer (include/
 * pepper_amd_io_device.h: checked when the trailer lies inside `comp`).  Timings: [10] the inflate kernel (HIP events),
 * [11] the whole call on the host clock (upload, kernel, download). */
void* pa_encoder_host_span(pa_encoder* e, int64_t bytes);
int pa_encoder_inflate_bgzf(pa_encoder* e, const uint8_t* comp, int64_t comp_bytes, int32_t n_blocks, const int64_t* comp_off,
                            const int32_t* comp_len, const int64_t* out_off, const int32_t* out_len, int64_t out_bytes,
                            uint8_t* host_out);
/* The BAM records of the span pa_encoder_inflate_bgzf left on the device, read out THERE: 40 bytes per record
 * (pa_record_header, include/pepper_amd_io.h: where its CIGAR / bases / qualities lie, position, counts, flags, the reference
 * bases its operations cover) instead of the span itself back over PCIe -- pa_encoder_inflate_bgzf with host_out = NULL, then
 * this, then pa_bam_pack_headers.  entries: record starts inside the span, ascending (pa_bam_span_entries: the first record and
 * the linear index's entry of every later 16 kb window); one lane follows the records from each entry up to the next, at
 * most cap_per_entry of them.  flags[0] != 0: nothing was copied -- 1 a lane ran out of slots, 2 a record shorter than its
 * core fields, 4 more records than headers_cap; flags[1] = 1: the span ends inside a record (what the walk of
 * pa_bam_pack_inflated treats as a cut).  The caller then takes the span to the host after all. */
int pa_encoder_walk_records(pa_encoder* e, int64_t data_bytes, const int64_t* entries, int32_t n_entries, int32_t cap_per_entry,
                            void* headers, int64_t headers_cap, int64_t* n_headers, int32_t* flags);
/* Host threads of a run's candidate enumeration (one short task per region): 0 = the default (the CPUs the process may use),
 * 1 = the calling thread alone -- what image generation sets, whose workers each drive their own encoder while the other
 * CPUs inflate BGZF blocks. */
int pa_encoder_set_host_threads(pa_encoder* e, int32_t n);
/* Reads with at least one base inside each region of the last run -- the reference's len(all_reads) after get_reads (an
 * interval without any writes no summary group, AlignmentSummarizer.py:200-204); host-clipped form: the pileup's n_reads. */
int pa_encoder_region_reads(pa_encoder* e, int32_t* n_reads, int32_t n);

/* Times of the last run in milliseconds, HIP events on the encoder's stream: [0] record kernels (segment_reads x 2 +
 * tile_offsets), [1] tile_count_kernel, [2] compact_votes_kernel + pack_results_kernel, [3] gather_windows_kernel; host clock:
 * [4] candidate enumeration, [5] the whole run, [6], [7] parts of [4]; packed form: [8] upload of arena + tables,
 * [9] unpack_clip_kernel.  Sizes of the staged batch: [0] read bases, [1] matrix rows,
 * [2] reads, [3] CIGAR operations, [4] tiles, [5] regions. */
int pa_encoder_last_timing(pa_encoder* e, double* ms, int32_t n);
int pa_encoder_batch_stats(pa_encoder* e, int64_t* out, int32_t n);

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
/* Many regions per launch: pileups[n_regions], start_pos[n_regions], end_pos[n_regions] -> n_rows[n_regions] (may be NULL);
 * the results are the rows of region 0, then region 1, ...  The whole per-read walk runs on the device. */
int pa_polish_encoder_generate_summary_batch(pa_encoder* e, int32_t n_regions, const pa_pileup* pileups,
                                             const int64_t* start_pos, const int64_t* end_pos, int64_t* n_rows);
/* The two halves of the call above (stage = validate + upload, run = the kernels; may be repeated: what bench.py times with the
 * pileups resident in HBM), and the sizes of the staged batch / last run: [0] read bases, [1] output rows, [2] reads, [3] CIGAR
 * operations, [4] tiles, [5] regions. */
int pa_polish_encoder_stage_batch(pa_encoder* e, int32_t n_regions, const pa_pileup* pileups, const int64_t* start_pos,
                                  const int64_t* end_pos);
int pa_polish_encoder_run_staged(pa_encoder* e, int64_t* n_rows);
int pa_polish_encoder_batch_stats(pa_encoder* e, int64_t* out, int32_t n);
/* HOST pointers: image uint8 [n_rows, 10], positions int64 [n_rows, 2] (of all regions of the last call); either may be NULL. */
int pa_polish_encoder_get_results(pa_encoder* e, uint8_t* image, int64_t* positions);
/* HIP-event times of the last call in ms: [0] record / scan kernels, [1] polish_tile_kernel, [2] polish_insert_rows_kernel. */
int pa_polish_encoder_last_timing(pa_encoder* e, double* ms, int32_t n);

/* ------------------------------------------------------------------------------------------
 * The polish image chain: BAM records -> image chunks of N regions without a host hop in between.
 * replaces, per region: AlignmentSummarizer.create_summary's inference branch --
 *   bam_handler.get_reads(chr, start, end, False, 0, 0)            pepper/modules/python/AlignmentSummarizer.py:296-303
 *   reads_to_reference_realignment (ReadAligner over every read)   :159-177, 328-332; simple_aligner.cpp:66-106
 *   SummaryGenerator(...).generate_summary(reads, start, end)      :334-347; summary_generator.cpp:47-121, 274-306, 370-393
 *   chunk_images(summary, 1000, 50)                                :18-56
 * The reads arrive in the packed form of pa_encoder_stage_packed (or are the span pa_encoder_inflate_bgzf left on the device:
 * arena = NULL); unpack_clip_kernel clips and decodes them per (read, region), the re-aligner (include/pepper_amd_realign.h)
 * takes them from there and leaves positions and CIGARs on the device, the summary encoder reads those, and the rows are cut
 * into chunks of chunk_size rows (the next one starting chunk_overlap rows before the end of the previous one, the last one
 * padded with zero rows and (-1, -1) positions) by a kernel.  One download: the chunks.
 *   regions[r]   region_start / region_end = the region (reads fetched from it, summary over it); reference / reference_len =
 *                the draft from region_start to region_end + ALIGNMENT_SAFE_BASES (20), shorter at the contig's end: the
 *                re-aligner's window (ignored when realign == 0)
 *   realign      realignment_flag of create_summary
 * Outputs (any may be NULL): n_rows[r] summary rows, region_reads[r] = the reference's len(all_reads), n_chunks[r] (0 for a
 * region without reads: it writes nothing), *total_chunks.  The reservoir sample of a region with more than
 * MAX_READS_IN_REGION reads is the caller's business (such a region goes through the per-region entry points).
 * PA_ERR_UNSUPPORTED: a batch this form does not take (an operation of 2^24 bases, a read that keeps more than 2 L + 64 bases
 * of a region of L positions): nothing was produced, take the host-clipped form.
 * ------------------------------------------------------------------------------------------ */
int pa_polish_chain_run(pa_encoder* e, int32_t n_regions, const pa_packed_region* regions, const uint8_t* arena, int64_t arena_bytes,
                        const pa_packed_read* reads, int32_t n_reads, const int32_t* pair_read, const int32_t* region_pairs,
                        int32_t realign, int32_t chunk_size, int32_t chunk_overlap, int64_t* n_rows, int32_t* region_reads,
                        int32_t* n_chunks, int64_t* total_chunks);
/* The chunks of the last run, region after region, in page-locked memory of the handle (valid until its next run):
 * images uint8 [total_chunks, chunk_size, 10], position / index int64 [total_chunks, chunk_size]. */
int pa_polish_chain_chunks(pa_encoder* e, const uint8_t** images, const int64_t** position, const int64_t** index);
/* The same images where the chunk kernel left them ON THE DEVICE ([total_chunks, chunk_size, 10] uint8, complete when
 * pa_polish_chain_run has returned, valid until the handle's next run): what the polish model reads without the HDF5 round trip
 * (pa_polish_predict_device; pepper_amd/polish/fused.py). */
int pa_polish_chain_device_chunks(pa_encoder* e, const uint8_t** images);
/* Host-clock times of the last run in ms: [0] tables + upload + unpack launch, [1] re-aligner (its waits included) + apply,
 * [2] summary encoder (its wait included), [3] chunk kernel + download; HIP events: [5] score kernels, [6] band launches.
 * counts: [0] (read, region) pairs, [1] reads re-aligned, [2] CIGAR operations written, [3] summary rows, [4] reads whose 8-bit
 * score pass was proven to overflow from their BAM alignment and skipped (ssw.c:819-824 discards that pass's results). */
int pa_polish_chain_last_timing(pa_encoder* e, double* ms, int32_t n_ms, int64_t* counts, int32_t n_counts);

#ifdef __cplusplus
}
#endif
#endif /* PEPPER_AMD_ENCODER_H */
