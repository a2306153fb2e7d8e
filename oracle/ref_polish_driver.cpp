// ORACLE (test infrastructure): C-ABI driver around the REFERENCE's own polish SummaryGenerator, compiled
// from the sources where they lie under /root/reference/pepper/modules (see oracle/Makefile).  Nothing of
// the reference is copied into this repository.
//
// summary_generator.h reaches type_read / CigarOp through "../dataio/bam_handler.h", a header that first
// includes htslib (absent from this image) and ends with the htslib-backed BAM_handler class.  The Makefile
// lifts, verbatim and at build time, (1) the block of plain type definitions of that header (from `using
// namespace std;` to just before `class BAM_handler`), (2) summary_generator.h minus that one #include line
// and (3) summary_generator.cpp minus its #include of the header, all into a scratch directory made by mktemp (removed after the compile; oracle/_ref/ keeps only the .so).  No
// stand-in for any htslib header, type or function is written: the encoder never touches htslib.
#include <assert.h>
#include <math.h>

#include <algorithm>
#include <cstring>
#include <iomanip>
#include <iostream>
#include <map>
#include <set>
#include <sstream>
#include <string>
#include <vector>

#include "polish_read_types_extract.h"    // generated from pepper/modules/headers/dataio/bam_handler.h
#include "polish_summary_generator.h"     // generated from pepper/modules/headers/pileup_summary/summary_generator.h
#include "polish_summary_generator.cpp"   // generated from pepper/modules/src/pileup_summary/summary_generator.cpp

#include "pileup_abi.h"

extern "C" int64_t ref_polish_generate_summary(const oracle_pileup* p, int64_t start_pos, int64_t end_pos,
                                               uint8_t* out_image, int64_t* out_pos, int64_t cap_rows) {
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
    SummaryGenerator gen(std::string(p->reference, (size_t)p->reference_len), "contig", p->region_start, p->region_end);
    gen.generate_summary(reads, start_pos, end_pos);
    const int64_t rows = (int64_t)gen.image.size();
    if (out_image != nullptr && out_pos != nullptr) {
        for (int64_t i = 0; i < rows && i < cap_rows; ++i) {
            for (int f = 0; f < 10; ++f) out_image[i * 10 + f] = f < (int)gen.image[(size_t)i].size() ? gen.image[(size_t)i][(size_t)f] : 0;
            out_pos[2 * i] = gen.genomic_pos[(size_t)i].first;
            out_pos[2 * i + 1] = gen.genomic_pos[(size_t)i].second;
        }
    }
    return rows;
}
