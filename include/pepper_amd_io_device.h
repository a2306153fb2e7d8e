/* pepper_amd -- BGZF blocks inflated on the device (C ABI of libpepper_amd.so).
 *
 * What it replaces: the block inflate under every BAM read of image generation -- htslib's bgzf_read_block beneath
 * sam_itr_next in /root/reference/pepper_variant/modules/cpp/bam_handler.cpp:341-372 (get_reads) and
 * /root/reference/pepper_hp/modules/src/dataio/bam_handler.cpp (the polisher's copy).  A BGZF file (SAM/BAM specification,
 * section 4.1) is a sequence of independent gzip members of at most 64 KiB of data each; one wavefront inflates one member
 * (RFC 1951: stored, fixed and dynamic blocks) and, as htslib's inflate_block does, checks the CRC-32 of the inflated bytes
 * against the member's trailer.  Every structural error of the stream, and a CRC mismatch, fails the call (pa_last_error names
 * the block and the reason).
 *
 * The caller describes the blocks: comp_off/comp_len = the raw DEFLATE bytes of block b inside `comp` (after the member's
 * header, before its CRC32/ISIZE trailer), out_off/out_len = where its ISIZE bytes go inside `out`.
 * pa_bam_read_span (include/pepper_amd_io.h) reads a stretch of a BAM file and fills exactly these tables.
 * CRC contract: the 4 bytes at comp[comp_off[b] + comp_len[b] ..] are the member's CRC-32 (little endian) WHEN they lie inside
 * `comp` (comp_off[b] + comp_len[b] + 4 <= comp_bytes) -- `comp` then holds whole members, as pa_bam_read_span leaves them.  A
 * block whose DEFLATE bytes end less than 4 bytes before comp_bytes is inflated but not checked (the host reader does the same):
 * a caller that hands over a bare DEFLATE stream puts it last, or appends its CRC-32.
 */
#ifndef PEPPER_AMD_IO_DEVICE_H
#define PEPPER_AMD_IO_DEVICE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pa_inflater pa_inflater;

int pa_inflater_create(int32_t device, pa_inflater** out);
void pa_inflater_destroy(pa_inflater* h);
/* Host buffers in, host buffer out: upload, `repeats` runs of the kernel (>= 1; the runs write the same bytes -- for timing),
 * download.  Returns PA_ERR_INVALID with the first failing block's reason when a stream is malformed. */
int pa_inflater_inflate(pa_inflater* h, const uint8_t* comp, int64_t comp_bytes, int32_t n_blocks, const int64_t* comp_off,
                        const int32_t* comp_len, const int64_t* out_off, const int32_t* out_len, uint8_t* out, int64_t out_bytes,
                        int32_t repeats);
/* average duration of one kernel run of the last call, milliseconds (HIP events on the handle's stream) */
int pa_inflater_last_kernel_ms(pa_inflater* h, double* ms);

#ifdef __cplusplus
}
#endif
#endif
