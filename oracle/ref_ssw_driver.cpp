// ORACLE (test infrastructure): C-ABI driver around the REFERENCE's own striped Smith-Waterman (the SSW
// library vendored under /root/reference/pepper/modules/src/local_reassembly/{ssw.c,ssw_cpp.cpp}), compiled
// from the sources where they lie (see oracle/Makefile; output only in oracle/_ref/).  Nothing of the
// reference is copied into this repository.  The aligner object is configured exactly as the reference's
// LibSSWPairwiseAligner does it (simple_aligner.cpp:12-30, simple_aligner.h:19-25): match 4, mismatch 6,
// gap open 8, gap extend 2, default Filter, maskLen 0.
#include <cstdint>
#include <cstring>
#include <string>

#include "pepper/modules/src/local_reassembly/ssw_cpp.cpp"
#include "pepper/modules/src/local_reassembly/ssw.c"

extern "C" int ref_ssw_align(const char* ref, int32_t ref_len, const char* query, int32_t* score, int32_t* ref_begin,
                             int32_t* ref_end, int32_t* query_begin, int32_t* query_end, char* cigar, int32_t cigar_cap) {
    StripedSmithWaterman::Aligner aligner(4, 6, 8, 2);
    StripedSmithWaterman::Filter filter;
    StripedSmithWaterman::Alignment al;
    aligner.SetReferenceSequence(ref, ref_len);
    const bool ok = aligner.Align_cpp(query, filter, &al, 0);
    *score = al.sw_score;
    *ref_begin = al.ref_begin;
    *ref_end = al.ref_end;
    *query_begin = al.query_begin;
    *query_end = al.query_end;
    if ((int32_t)al.cigar_string.size() + 1 > cigar_cap) return -1;
    std::memcpy(cigar, al.cigar_string.c_str(), al.cigar_string.size() + 1);
    return ok ? 1 : 0;
}
