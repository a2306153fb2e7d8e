/* pepper_amd_io C ABI -- the minimal HDF5 surface PEPPER's inference path touches.
 *
 * The reference reads and writes its step hand-off files through h5py:
 *   images       pepper_variant/modules/python/DataStore.py:54-71  (write_summary)
 *                pepper_variant/modules/python/models/dataloader_predict.py:45-79 (bulk reads)
 *   predictions  pepper_variant/modules/python/DataStorePredict.py:49-67 (write_prediction)
 *   polish       pepper/modules/python/DataStore.py:53-67, pepper/modules/python/DataStorePredict.py:49-76,
 *                pepper/modules/python/models/dataloader_predict.py:49-60
 * h5py is not installed for the torch-ROCm interpreter of this image, so the same libhdf5 C
 * calls are made directly (HDF5 1.10, /opt/conda/lib/libhdf5.so.103).  Datasets are written the
 * way `file[path] = ndarray` does it: contiguous, no chunking, no compression, intermediate
 * groups created on demand.
 */
#ifndef PEPPER_AMD_IO_H
#define PEPPER_AMD_IO_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pa_h5 pa_h5;

/* element type codes for numeric datasets */
#define PA_H5_I8 0
#define PA_H5_U8 1
#define PA_H5_I16 2
#define PA_H5_I32 3
#define PA_H5_I64 4
#define PA_H5_F32 5
#define PA_H5_F64 6
#define PA_H5_U16 7
#define PA_H5_U32 8
#define PA_H5_U64 9

/* dataset classes reported by pa_h5_info */
#define PA_H5_CLASS_INT 0
#define PA_H5_CLASS_FLOAT 1
#define PA_H5_CLASS_FIXED_STRING 2
#define PA_H5_CLASS_VLEN_STRING 3
#define PA_H5_CLASS_OTHER 4

const char* pa_h5_last_error(void);

/* mode: 0 = read-only, 1 = create/truncate ('w'), 2 = read-write existing ('r+'), 3 = create/truncate with the HDF5 1.10
 * object formats (written faster when a file is hundreds of thousands of small groups; the prediction stores use it) */
int pa_h5_open(const char* path, int32_t mode, pa_h5** out);
int pa_h5_close(pa_h5* f);
int pa_h5_flush(pa_h5* f);

/* 1 if `path` names an existing link (group or dataset), 0 if not, <0 on error */
int pa_h5_exists(pa_h5* f, const char* path);
/* child names of a group, NUL-separated, in HDF5's name order (= h5py's keys() order).
 * Returns 0 and sets *needed; call again with a buffer of that size if cap was too small. */
int pa_h5_list(pa_h5* f, const char* group, char* buf, int64_t cap, int64_t* needed, int64_t* count);

/* rank (<= 8), dims, class, element size in bytes (string width for fixed strings), signedness */
int pa_h5_info(pa_h5* f, const char* path, int32_t* rank, int64_t* dims, int32_t* cls, int32_t* elem_size,
               int32_t* is_signed);

/* whole-dataset numeric read with conversion to `type_code`; nbytes must equal the full size */
int pa_h5_read(pa_h5* f, const char* path, int32_t type_code, void* out, int64_t nbytes);
/* create + write a numeric dataset (rank 0 = scalar) */
int pa_h5_write(pa_h5* f, const char* path, int32_t type_code, int32_t rank, const int64_t* dims,
                const void* data);

/* all strings of a (fixed or variable length) string dataset, NUL-separated, row-major order */
int pa_h5_read_strings(pa_h5* f, const char* path, char* buf, int64_t cap, int64_t* needed);
/* numpy dtype 'S<width>' dataset: `data` holds prod(dims) fields of `width` bytes, NUL padded */
int pa_h5_write_fixed_strings(pa_h5* f, const char* path, int32_t rank, const int64_t* dims, int32_t width,
                              const char* data);
/* h5py special_dtype(vlen=str) dataset (variable-length UTF-8) */
int pa_h5_write_vlen_strings(pa_h5* f, const char* path, int32_t rank, const int64_t* dims,
                             const char* const* strings);

/* One `predictions/batch_<n>` group of the variant predictions file in a single call (DataStorePredict.py:49-67):
 * contigs (fixed-width strings, n rows of contig_stride bytes, null padded), positions int32 [n], depths uint8 [n],
 * candidates vlen utf-8 [n,1] (NUL-terminated strings at cand_blob + cand_offsets[i]), candidate_frequency uint8 [n,1],
 * base_prediction float64 [n, n_classes] (converted from the float32 device output). */
int pa_h5_write_prediction_batch(pa_h5* f, const char* group, int32_t n, const char* contigs, int32_t contig_stride,
                                 const int32_t* positions, const uint8_t* depths, const char* cand_blob,
                                 const int64_t* cand_offsets, const uint8_t* freqs, const float* probs, int32_t n_classes);

/* Polish stores, one block of chunks per call (the format keeps one group per 1000-row chunk):
 * read    summaries/<name>/{image u8 [seq,features], position, index [seq], region_start, region_end, chunk_id, contig}
 *         for the n NUL-separated group names into caller arrays (pepper/.../DataStore.py:53-67, dataloader_predict.py);
 * write   predictions/<contig>/<contig>-<start>-<end>/{contig_start, contig_end} where new_region[i] is set and
 *         .../<chunk_id>/{position, index int64 [seq], bases, phred_score uint8 [seq]} unless skip[i]
 *         (pepper/.../DataStorePredict.py:49-76). */
int pa_h5_read_polish_chunks(pa_h5* f, const char* names, int32_t n, int32_t seq_len, int32_t features, uint8_t* images,
                             int64_t* position, int64_t* index, int64_t* region_start, int64_t* region_end,
                             int64_t* chunk_id, char* contigs, int32_t contig_stride);
/* Candidate selection + VCF record text for the rows of one prediction batch whose candidate lists hold ONE allele each
 * (what pepper_amd's image generation writes): the rules of pepper_variant CandidateFinder.py:356-581 (small_chunk_stitch /
 * find_candidates) and VcfWriter.py:48-218 for a site with one allele record.  rules: thresholds per allele kind
 * (0 SNP "1...", 1 insert "2...", 2 delete "3...").  Per row: position, depth, support (candidate_frequency), prediction
 * float32 [n,3], the upper-cased reference base (0 = outside the contig) and low-complexity flag of the position, the
 * allele code ("1A", "2ACC", "3ACG") at alleles + allele_offsets[i] .. allele_offsets[i+1] - separator_bytes (1 for the
 * NUL-separated text pa_h5_read_strings returns).
 * -> the number m of rows kept, for each: its row (kept_row), len(REF) (ref_len), flags (bit 0 SNP, bit 1 selected for
 * re-genotyping, bit 2 REF/ALT swapped (a deletion called by probability), bits 4-5 genotype) and its VCF line
 * lines[line_offsets[k] .. line_offsets[k+1]) ('\n'-terminated).  -1: error; -2: a row only the reference-shaped Python path
 * reproduces (NaN probabilities, the reference's division by a zero depth): the caller takes that path for the batch. */
typedef struct {
    double p_value[3], p_value_in_lc[3], report_above_freq[3];
    double snp_q_cutoff, snp_q_cutoff_in_lc, indel_q_cutoff, indel_q_cutoff_in_lc;
} pa_candidate_rules;
/* The upper-cased reference base (0 outside the window) and the low-complexity flag of the reference's candidate finder
 * (CandidateFinder.py:397-418: a homopolymer run >= 5 touching [p - 5, p + 4) inside the context ref[p - 10, p + 10)) for n
 * positions, out of the text of ONE fetch ref[window_lo, window_lo + text_len) that covers them with 16 bases to spare. */
int pa_candidates_reference_flags(const char* text, int64_t text_len, int64_t window_lo, int64_t n, const int64_t* position,
                                  uint8_t* letters, uint8_t* in_repeat);
int64_t pa_candidates_select_format(const pa_candidate_rules* rules, const char* contig, int64_t n, const int64_t* position,
                                    const int64_t* depth, const int64_t* support, const float* prediction,
                                    const uint8_t* reference_base, const uint8_t* in_repeat, const char* alleles,
                                    const int64_t* allele_offsets, int32_t separator_bytes, int32_t* kept_row, int32_t* ref_len,
                                    uint8_t* flags, char* lines, int64_t lines_cap, int64_t* line_offsets);

/* One piece of a contig's consensus, as pepper Stitch.py:36-94 (small_chunk_stitch) builds it: the chunks of the given region
 * groups ("predictions/<contig>/<contig>-<start>-<end>", NUL-separated, each in files[file_of_region[r]]; chunk ids in string
 * order) are merged by (position, insert index) -- rows with a negative position or index are padding; in a region that does
 * not start at 0 the rows at positions <= region_start + buffer_positions are the overlap with the region before and are
 * dropped; the last write of a key wins -- and the labels of the keys in order are decoded (0 -> nothing, 1-4 -> ACGT).
 * -> first / last position of the piece (-1, -1 and length 0 when nothing is left), the length of its sequence, which
 * pa_h5_stitch_take then copies out (thread-local between the two calls).  A label above 4 is an error (*bad_label holds it:
 * the reference raises KeyError).  Chunks are read straight from the mapped file where the direct locator knows the format. */
int pa_h5_stitch_polish_regions(pa_h5* const* files, const int32_t* file_of_region, const char* region_paths,
                                const int64_t* region_start, int32_t n_regions, int64_t buffer_positions, int64_t* first_pos,
                                int64_t* last_pos, int64_t* sequence_len, int64_t* bad_label);
int pa_h5_stitch_take(char* out, int64_t cap);
/* The region groups of predictions/<contig>, in name order, with their contig_start / contig_end scalars (what
 * pepper perform_stitch.py:63-74 collects one h5py call at a time).  Call with buf == NULL for *needed (bytes of the
 * NUL-separated names) and *count, then with buffers of those sizes. */
int pa_h5_list_polish_regions(pa_h5* f, const char* contig, char* buf, int64_t cap, int64_t* needed, int64_t* count,
                              int64_t* starts, int64_t* ends, int64_t cap_regions);

/* Append-only builder of a polish prediction file (pepper_amd/csrc/h5build.cpp; no libhdf5 involved): the same groups and
 * datasets as pa_h5_write_polish_predictions -- predictions/<contig>/<contig>-<start>-<end>/{contig_start, contig_end int64
 * scalars} and .../<chunk_id>/{position, index int64 [seq], bases, phred_score uint8 [seq]} (pepper DataStorePredict.py:49-76)
 * -- in the classic HDF5 format h5py writes by default: raw rows are appended as they arrive, all metadata (object headers,
 * local heaps, symbol nodes, group B-trees, superblock) is written by pa_h5_builder_close.  ~3 us of CPU per chunk instead of
 * libhdf5's 70-150 us.  The file is not an HDF5 file until close has returned 0; names, shapes and dtypes are what the
 * reference's readers expect (tests/test_hdf5_layout.py reads it back with libhdf5 and h5py).  Same arguments, duplicate
 * handling (new_region / skip flags decided by the caller) and errors as the libhdf5 entry point. */
typedef struct pa_h5_builder pa_h5_builder;
int pa_h5_builder_open(const char* path, pa_h5_builder** out);
int pa_h5_builder_write_polish_predictions(pa_h5_builder* b, int32_t n, int32_t seq_len, const char* contigs, int32_t contig_stride,
                                           const int64_t* contig_start, const int64_t* contig_end, const int64_t* chunk_id,
                                           const uint8_t* new_region, const uint8_t* skip, const int64_t* position,
                                           const int64_t* index, const uint8_t* bases, const uint8_t* phred);
/* One integer dataset at `path` (intermediate groups are made as needed), as pa_h5_write; data of at most 64 bytes is kept
 * in the object header (compact layout). */
int pa_h5_builder_write(pa_h5_builder* b, const char* path, int32_t type_code, int32_t rank, const int64_t* dims, const void* data);
/* A variable-length UTF-8 string scalar, as h5py writes a Python str (global heap object; equal strings share one). */
int pa_h5_builder_write_string(pa_h5_builder* b, const char* path, const char* text);
/* The polish image chunks of one region, as pa_h5_write_polish_image_chunks (pepper DataStore.py:53-67): summaries/<name>/
 * {image u8 [seq, features], label u8 [seq], position, index int64 [seq], contig (string), region_start, region_end,
 * chunk_id int64 scalars}.  No libhdf5 and so no process-wide lock: the image-generation threads each write their own file. */
int pa_h5_builder_write_polish_image_chunks(pa_h5_builder* b, const char* names, int32_t n, int32_t seq_len, int32_t features,
                                            const char* contig, int64_t region_start, int64_t region_end, const int64_t* chunk_id,
                                            const uint8_t* images, const uint8_t* labels, const int64_t* position,
                                            const int64_t* index);
/* The chunks of MANY regions of one contig in one call (the polish image chain, include/pepper_amd_encoder.h): region r has
 * n_chunks[r] consecutive chunks in images / position / index (chunk ids 0 .. n_chunks[r] - 1, groups
 * summaries/<contig>_<start>_<end>_<chunk id> as pepper ImageGenerationUI.py:203-211 names them); labels NULL: zeros (inference
 * mode).  A group that exists already is skipped, as DataStore.write_summary does (DataStore.py:53-56). */
int pa_h5_builder_write_polish_image_regions(pa_h5_builder* b, int32_t n_regions, const char* contig, const int64_t* region_start,
                                             const int64_t* region_end, const int32_t* n_chunks, int32_t seq_len, int32_t features,
                                             const uint8_t* images, const uint8_t* labels, const int64_t* position, const int64_t* index);
/* One summaries/<name> group of a variant image file, as pepper_variant DataStore.py:54-71 (write_summary, inference mode):
 * contigs 'S<len>' [n] (the one contig name n times), positions int32 [n], depths uint8 [n], candidates variable-length utf-8
 * [n,1] (NUL-terminated at cand_blob + cand_offsets[i]), candidate_frequency uint8 [n,1], images int8 [n, window, features].
 * The image-generation workers each write their own file without libhdf5's process-wide lock. */
int pa_h5_builder_write_variant_summary(pa_h5_builder* b, const char* name, int32_t n, const char* contig, const int32_t* positions,
                                        const uint8_t* depths, const char* cand_blob, const int64_t* cand_offsets, const uint8_t* freqs,
                                        const int8_t* images, int32_t window, int32_t features);
/* One predictions/<name> group of a variant prediction file, as pepper_variant DataStorePredict.py:26-67 (write_prediction) --
 * the datasets of pa_h5_write_prediction_batch above without libhdf5: contigs 'S<longest>' [n], positions int32 [n], depths uint8
 * [n], candidates variable-length utf-8 [n,1], candidate_frequency uint8 [n,1], base_prediction float64 [n, n_classes] (probs
 * are float32 here; `np.float` in the reference is float64).  The fused call_variant's writer thread lays out ~9 000 such
 * groups per 256 Mb: 0.3 ms each through libhdf5, a few microseconds here. */
int pa_h5_builder_write_prediction_batch(pa_h5_builder* b, const char* name, int32_t n, const char* contigs, int32_t contig_stride,
                                         const int32_t* positions, const uint8_t* depths, const char* cand_blob,
                                         const int64_t* cand_offsets, const uint8_t* freqs, const float* probs, int32_t n_classes);
int pa_h5_builder_close(pa_h5_builder* b);

/* One predictions/batch_<n> group of a variant prediction file (pepper_variant DataStorePredict.py:26-67) read in one call through
 * the locator -- what the candidate finder does six h5py reads for (pepper_variant CandidateFinder.py:356-374).  load: 0 = the
 * batch is held for this thread (n candidates, contigs as n x contig_width bytes null padded, the candidate strings as
 * candidate_bytes bytes each followed by a NUL, n_classes probabilities per candidate), 1 = not a layout the locator reads
 * (read the datasets through pa_h5_read / pa_h5_read_strings), -1 = error.  take copies the held batch out:
 * contigs [n * contig_width], candidates [candidate_bytes], positions int32 [n], depths uint8 [n], freq uint8 [n],
 * probs float64 [n * n_classes]. */
int pa_h5_prediction_batch_load(pa_h5* f, const char* group, int64_t* n, int32_t* contig_width, int64_t* candidate_bytes,
                                int32_t* n_classes);
int pa_h5_prediction_batch_take(char* contigs, char* candidates, int32_t* positions, uint8_t* depths, uint8_t* freq, double* probs);

/* How the polish chunks of this handle were read so far: `direct_chunks` had their image / position / index bytes copied
 * straight out of the mapped file (classic-format files of h5py or pa_h5_open mode 1 opened read-only: the locator in
 * hdf5io.cpp walks object header -> symbol table -> B-tree -> symbol node -> layout itself), `library_chunks` went through
 * libhdf5 (anything the locator does not recognise; PEPPER_AMD_H5_DIRECT=0 forces it). */
int pa_h5_read_stats(pa_h5* f, int64_t* direct_chunks, int64_t* library_chunks);

int pa_h5_write_polish_predictions(pa_h5* f, int32_t n, int32_t seq_len, const char* contigs, int32_t contig_stride,
                                   const int64_t* contig_start, const int64_t* contig_end, const int64_t* chunk_id,
                                   const uint8_t* new_region, const uint8_t* skip, const int64_t* position,
                                   const int64_t* index, const uint8_t* bases, const uint8_t* phred);

/* All chunks of one predictions/<contig>/<contig>-<start>-<end> group for the stitcher: sub-groups other than
 * contig_start / contig_end in string order, their position / index (int64 [seq]) and bases (uint8 [seq]) into rows of the
 * caller's arrays (max_chunks rows); *n_chunks = number of chunks found (Stitch.py:36-62 reads them one dataset at a time). */
int pa_h5_read_polish_prediction_region(pa_h5* f, const char* region_path, int32_t seq_len, int32_t max_chunks,
                                        int64_t* position, int64_t* index, uint8_t* bases, int32_t* n_chunks);

/* The chunks of one polish region into the image file in one call: summaries/<name>/{image u8 [seq,features], label u8
 * [seq], position, index int64 [seq], contig (vlen string), region_start, region_end, chunk_id int64}
 * (pepper/.../DataStore.py:53-67); names = n NUL-separated group names. */
int pa_h5_write_polish_image_chunks(pa_h5* f, const char* names, int32_t n, int32_t seq_len, int32_t features,
                                    const char* contig, int64_t region_start, int64_t region_end, const int64_t* chunk_id,
                                    const uint8_t* images, const uint8_t* labels, const int64_t* position,
                                    const int64_t* index);

/* ------------------------------------------------------------------------------------------
 * BAM ingestion (pepper_amd/csrc/bamio.cpp; zlib, no htslib)
 * replaces the pybind surface of PEPPER_VARIANT.BAM_handler:
 *   BAM_handler(path)                                   bam_handler.cpp:6-28
 *   .get_chromosome_sequence_names()                    bam_handler.cpp:103-113
 *   .get_sample_names()                                 bam_handler.cpp:30-54   (from pa_bam_header_text)
 *   .get_reads(contig, start, stop, include_supplementary, min_mapq, min_baseq)
 *        -> vector<type_read>                           bam_handler.cpp:115-451 (read.h:52-64)
 * Reads come back clipped to [start, stop] exactly as the reference clips them, as the flat arrays
 * of pa_pileup (include/pepper_amd_encoder.h) instead of per-read objects.  A `<path>.bai` (or
 * `<stem>.bai`) index is used when present; without one the contig is scanned linearly.
 * ------------------------------------------------------------------------------------------ */
typedef struct pa_bam pa_bam;

const char* pa_bam_last_error(void);
int pa_bam_open(const char* path, pa_bam** out);
void pa_bam_close(pa_bam* b);
int pa_bam_has_index(pa_bam* b);
int pa_bam_n_targets(pa_bam* b);
/* name (NUL-terminated, truncated to cap) and length of target i; returns the name length */
int pa_bam_target(pa_bam* b, int32_t i, char* name, int32_t cap, int64_t* length);
/* SAM header text; *needed = bytes including the terminator */
int pa_bam_header_text(pa_bam* b, char* buf, int64_t cap, int64_t* needed);
/* Run the region query; results stay in the handle until the next call.  Sizes for pa_bam_copy_reads:
 * n_reads, total bases, total cigar operations, bytes of the NUL-separated query names. */
int pa_bam_get_reads(pa_bam* b, const char* contig, int64_t start, int64_t stop, int32_t include_supplementary,
                     int32_t min_mapq, int32_t min_baseq, int64_t* n_reads, int64_t* seq_bytes, int64_t* n_cigar,
                     int64_t* name_bytes);
/* Copy out (any pointer may be NULL): pos / pos_end int64 [n], reverse u8 [n], mapq / flags / hp int32 [n],
 * seq_offset int64 [n+1], seq char [bases], qual u8 [bases], cigar_offset int64 [n+1], cigar_op / cigar_len
 * int32 [ops] (BAM operation codes = CIGAR_OPERATIONS, cigar.h:17-27), names char [name_bytes]. */
int pa_bam_copy_reads(pa_bam* b, int64_t* pos, int64_t* pos_end, uint8_t* reverse, int32_t* mapq, int32_t* flags,
                      int32_t* hp, int64_t* seq_offset, char* seq, uint8_t* qual, int64_t* cigar_offset,
                      int32_t* cigar_op, int32_t* cigar_len, char* names);

/* The reads of a run of regions of ONE contig in the packed form the GPU encoder clips and decodes itself
 * (pa_encoder_stage_packed, include/pepper_amd_encoder.h): what get_reads(contig, start[r], stop[r], ...) would return for
 * every region r, without the per-read walk on the host.  start / stop ascend.  Per kept record (filters and region test of
 * bam_handler.cpp:115-151: pos < stop, end > start, not qc-fail / duplicate / secondary / unmapped, supplementary only on
 * request, mapq >= min_mapq) one table entry and ONE copy of `CIGAR words | 4-bit bases | qualities` as the record holds them
 * (4-byte aligned at data_off) in `arena`, shared by all the regions the read reaches; pair_read[region_pairs[r] ..
 * region_pairs[r + 1]) are the reads of region r in file order.  The reads the reference's clipping would drop (no base inside
 * the region) are still listed: the device drops them.  When the arena or a table fills up the call stops at a region
 * boundary: *n_done regions (>= 1, else the call fails) are complete and described by counts = {reads, pairs, arena bytes};
 * the caller continues with region n_done. */
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
int pa_bam_pack_regions(pa_bam* b, const char* contig, int32_t n_regions, const int64_t* start, const int64_t* stop,
                        int32_t include_supplementary, int32_t min_mapq, uint8_t* arena, int64_t arena_cap,
                        pa_packed_read* reads, int32_t reads_cap, int32_t* pair_read, int32_t pairs_cap,
                        int32_t* region_pairs, int32_t* n_done, int64_t* counts);

/* ---- the same packed form over a span inflated elsewhere (on the device: pa_encoder_inflate_bgzf, include/
 * pepper_amd_encoder.h; or any inflate) ------------------------------------------------------------------------------------
 * pa_bam_region_span  where the records that can reach [start, stop) of `contig` lie in the file (needs the .bai): from the
 *                     BGZF member of the linear index's offset for start's 16 kb window (begin_coffset; the first record
 *                     begins begin_uoffset bytes into that member's data) to end_coffset, the end of the member holding the
 *                     first indexed record of the window `lookahead_windows` beyond stop's -- or (to_contig_end = 1) where the
 *                     next contig's records begin / the end of the file.  Nothing of the contig: begin = end = 0.
 * pa_bam_read_span    the members of [begin, ...) up to the first one starting at or after end_min plus `extra_members` more,
 *                     read with one pread into buf, and their tables for the inflate (comp_off/comp_len: the raw DEFLATE bytes
 *                     inside buf; out_off/out_len: the ISIZE bytes laid back to back).  *complete = 0 when buf or the tables
 *                     were too small for that (the members listed are whole ones either way), 1 when covered, 3 when
 *                     the span also ran to the end of the file (the inflated data is then final for pack_inflated).
 * pa_bam_pack_inflated  pa_bam_pack_regions' walk over the inflated bytes, records left in place: data_off = the offset of
 *                     the record's `CIGAR words | 4-bit bases | qualities` in `data` (not aligned; the device form reads
 *                     unaligned words), counts[2] = the bytes of the kept slices.  The walk must see a record at or beyond
 *                     the last stop (or the end of a span that data_is_final says holds the contig's last record); when the
 *                     span ends earlier the regions closed by then are done (*n_done) and the caller takes a later span for the
 *                     rest -- none closed: -9.  A record whose CIGAR lives in the CG tag is no single slice: -8, take
 *                     pa_bam_pack_regions for that batch. */
/* One record of an inflated span as the device's walk reads it out (pa_encoder_walk_records, include/pepper_amd_encoder.h):
 * data_off = where the record's `CIGAR words | bases | qualities` start in the span, ref_len = the reference bases its
 * operations cover, state = 0, 1 (the placeholder of a CIGAR kept in the CG tag) or 2 (fields that overrun the record). */
#ifndef PA_RECORD_HEADER_DEFINED
#define PA_RECORD_HEADER_DEFINED
typedef struct {
    int64_t data_off;
    int32_t ref_id, pos, l_seq, n_cigar;
    int32_t flags;         /* BAM flag | mapping quality << 16 */
    int32_t ref_len, state, block_size;
} pa_record_header;
#endif
/* pa_bam_span_entries  record starts inside the span of the handle's last pa_bam_read_span, as offsets into the inflated bytes:
 *                     first_record, then the linear index's entry of every later 16 kb window whose record lies in the span
 *                     (ascending, distinct) -- where the device's walk starts its lanes.
 * pa_bam_pack_headers  pa_bam_pack_inflated's walk over the headers the device read out instead of over the bytes. */
int pa_bam_span_entries(pa_bam* b, const char* contig, int64_t first_record, const int64_t* out_off, int32_t n_blocks,
                        int64_t* entries, int32_t entries_cap, int32_t* n_entries);
int pa_bam_pack_headers(pa_bam* b, const pa_record_header* headers, int64_t n_headers, int32_t data_is_final, const char* contig,
                        int32_t n_regions, const int64_t* start, const int64_t* stop, int32_t include_supplementary, int32_t min_mapq,
                        pa_packed_read* reads, int32_t reads_cap, int32_t* pair_read, int32_t pairs_cap, int32_t* region_pairs,
                        int32_t* n_done, int64_t* counts);
/* The host's counterpart of the device inflate (include/pepper_amd_io_device.h) over the same member tables: libdeflate where
 * it is installed (htslib's choice), zlib otherwise, on n_threads threads -- the CPU baseline of the inflate bench. */
int pa_bgzf_inflate_host(const uint8_t* comp, int64_t comp_bytes, int32_t n_blocks, const int64_t* comp_off, const int32_t* comp_len,
                         const int64_t* out_off, const int32_t* out_len, uint8_t* out, int64_t out_bytes, int32_t n_threads);
int pa_bam_region_span(pa_bam* b, const char* contig, int64_t start, int64_t stop, int32_t lookahead_windows,
                       int64_t* begin_coffset, int32_t* begin_uoffset, int64_t* end_coffset, int32_t* to_contig_end);
int pa_bam_read_span(pa_bam* b, int64_t begin, int64_t end_min, int32_t extra_members, uint8_t* buf, int64_t buf_cap,
                     int64_t* comp_off, int32_t* comp_len, int64_t* out_off, int32_t* out_len, int32_t blocks_cap,
                     int32_t* n_blocks, int64_t* comp_bytes, int64_t* out_bytes, int32_t* complete);
int pa_bam_pack_inflated(pa_bam* b, const uint8_t* data, int64_t data_bytes, int64_t first_record, int32_t data_is_final,
                         const char* contig, int32_t n_regions, const int64_t* start, const int64_t* stop,
                         int32_t include_supplementary, int32_t min_mapq, pa_packed_read* reads, int32_t reads_cap,
                         int32_t* pair_read, int32_t pairs_cap, int32_t* region_pairs, int32_t* n_done, int64_t* counts);

#ifdef __cplusplus
}
#endif
#endif /* PEPPER_AMD_IO_H */
