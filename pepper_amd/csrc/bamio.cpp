// BAM ingestion for the image generator: region query + the
// reference's read clipping, delivered as the flat arrays of pa_pileup (include/pepper_amd_encoder.h).
//
// replaces: /root/reference/pepper_variant/modules/cpp/bam_handler.cpp
//     BAM_handler::BAM_handler            :6-28    open + index + header
//     get_sample_names                    :30-54   @RG SM values (parsed in pepper_amd/variant/bam.py)
//     get_chromosome_sequence_names       :103-113
//     get_reads                           :115-451 region iterator, flag / mapq filters, clipping of every
//                                                  read to [start, stop], HP tag
// htslib is not part of this image; the containers are read from their published layouts (SAM/BAM
// specification sections 4.1 BGZF, 4.2 BAM, 5.2 BAI) with zlib.  PARITY UNPINNED: the reference's reader
// cannot be built here (htslib), so tests/test_bam_reader.py checks this file against a Python
// restatement of get_reads' rules on synthetic BAM files written by the test itself.
//
// Region semantics: like sam_itr_queryi(idx, tid, start, stop) the candidates are the records of the
// contig with pos < stop and end > start (end = pos + reference length, at least pos + 1), in file
// order.  With a .bai the scan starts at the linear-index offset of start's 16 kb window; without one
// the contig is scanned from its first record (correct, slower).
#include <zlib.h>
#ifdef PA_HAVE_LIBDEFLATE      // set by pepper_amd/build.py together with -ldeflate (one decision: header AND library)
#include <libdeflate.h>        // htslib's own choice for BGZF blocks where it is installed: 2-3x zlib's inflate
#endif

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <cstring>
#include <deque>
#include <memory>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/pepper_amd_io.h"

namespace {

// CRC-32 of an inflated member (libdeflate's where it is linked, else zlib's)
inline uint32_t member_crc32(const uint8_t* data, size_t n) {
#ifdef PA_HAVE_LIBDEFLATE
    return libdeflate_crc32(0, data, n);
#else
    return (uint32_t)crc32(crc32(0L, Z_NULL, 0), data, (uInt)n);
#endif
}

thread_local std::string g_bam_err;
int bam_fail(int code, const std::string& msg) {
    g_bam_err = msg;
    return code;
}

struct Bgzf {
    FILE* fp = nullptr;
    // Decompressed blocks are kept (FIFO, 256 x <= 64 KiB): image generation queries consecutive ~1 kb regions, and every
    // query re-reads the records of the 16 kb index window in front of it -- with long reads that is the same couple
    // of megabytes over and over, and inflating them was most of a query's time.
    typedef std::shared_ptr<std::vector<uint8_t>> Block;
    struct Cached { Block data; int64_t next; };
    std::unordered_map<int64_t, Cached> cache;
    std::deque<int64_t> order;
    static constexpr size_t kCacheBlocks = 256;
    z_stream zs{};
    bool zs_ready = false;
    std::vector<uint8_t> comp;       // compressed bytes of the block being loaded
#ifdef PA_HAVE_LIBDEFLATE
    libdeflate_decompressor* ld = nullptr;
#endif
    ~Bgzf() {
        if (zs_ready) inflateEnd(&zs);
#ifdef PA_HAVE_LIBDEFLATE
        if (ld) libdeflate_free_decompressor(ld);
#endif
    }
    Block cur = std::make_shared<std::vector<uint8_t>>();   // decompressed current block
    int64_t block_coffset = -1;      // file offset of the current block
    int64_t next_coffset = 0;        // file offset of the block after it
    size_t upos = 0;                 // read position inside the current block
    bool eof = false;
    bool failed = false;             // a block could not be decoded (bad magic, short block, inflate error): NOT end of file

    bool load_block(int64_t coffset) {
        const auto hit = cache.find(coffset);
        if (hit != cache.end()) {
            cur = hit->second.data;
            block_coffset = coffset;
            next_coffset = hit->second.next;
            upos = 0;
            eof = false;
            return true;
        }
        if (fseeko(fp, coffset, SEEK_SET) != 0) return false;
        uint8_t hdr[18];
        const size_t got = fread(hdr, 1, 18, fp);
        if (got == 0) {
            eof = true;
            cur = std::make_shared<std::vector<uint8_t>>();
            upos = 0;
            block_coffset = coffset;
            next_coffset = coffset;
            return true;
        }
        if (got != 18 || hdr[0] != 0x1f || hdr[1] != 0x8b || hdr[2] != 8 || !(hdr[3] & 4)) return false;
        const int xlen = hdr[10] | (hdr[11] << 8);
        // the BC subfield is the first extra field in every BGZF writer; scan anyway
        std::vector<uint8_t> extra(xlen);
        std::memcpy(extra.data(), hdr + 12, std::min(6, xlen));
        if (xlen > 6 && fread(extra.data() + 6, 1, xlen - 6, fp) != (size_t)(xlen - 6)) return false;
        int bsize = -1;
        for (int p = 0; p + 4 <= xlen;) {
            const int slen = extra[p + 2] | (extra[p + 3] << 8);
            if (extra[p] == 'B' && extra[p + 1] == 'C' && slen == 2 && p + 6 <= xlen) bsize = (extra[p + 4] | (extra[p + 5] << 8)) + 1;
            p += 4 + slen;
        }
        if (bsize < 0) return false;
        const int clen = bsize - 12 - xlen - 8;
        if (clen < 0) return false;
        comp.resize((size_t)clen + 8);
        if (fread(comp.data(), 1, clen + 8, fp) != (size_t)(clen + 8)) return false;
        const uint32_t isize = comp[clen + 4] | (comp[clen + 5] << 8) | (comp[clen + 6] << 16) | ((uint32_t)comp[clen + 7] << 24);
        Block fresh;
        if (order.size() >= kCacheBlocks) {               // evict the oldest block and take over its (already mapped) memory
            const auto oldest = cache.find(order.front());
            if (oldest != cache.end()) {
                if (oldest->second.data.use_count() == 1) fresh = oldest->second.data;
                cache.erase(oldest);
            }
            order.pop_front();
        }
        if (!fresh) fresh = std::make_shared<std::vector<uint8_t>>();
        fresh->resize(isize);
        if (isize) {
#ifdef PA_HAVE_LIBDEFLATE
            if (!ld) ld = libdeflate_alloc_decompressor();
            if (!ld || libdeflate_deflate_decompress(ld, comp.data(), (size_t)clen, fresh->data(), isize, nullptr) != LIBDEFLATE_SUCCESS)
                return false;
#else
            if (!zs_ready) {                              // one inflate state per handle, reset per block
                if (inflateInit2(&zs, -15) != Z_OK) return false;
                zs_ready = true;
            } else if (inflateReset(&zs) != Z_OK) {
                return false;
            }
            zs.next_in = comp.data();
            zs.avail_in = clen;
            zs.next_out = fresh->data();
            zs.avail_out = isize;
            if (inflate(&zs, Z_FINISH) != Z_STREAM_END) return false;
#endif
        }
        {   // the member's CRC-32 (htslib's inflate_block fails the read on a mismatch; so does this reader)
            const uint32_t want = comp[clen] | (comp[clen + 1] << 8) | (comp[clen + 2] << 16) | ((uint32_t)comp[clen + 3] << 24);
            if (member_crc32(fresh->data(), isize) != want) return false;
        }
        cur = fresh;
        block_coffset = coffset;
        next_coffset = coffset + bsize;
        upos = 0;
        eof = false;
        cache[coffset] = Cached{fresh, next_coffset};
        order.push_back(coffset);
        return true;
    }
    bool seek(uint64_t voffset) {
        const int64_t co = (int64_t)(voffset >> 16);
        if (co != block_coffset && !load_block(co)) return false;
        upos = voffset & 0xffff;
        return upos <= cur->size();
    }
    // returns bytes read (< n only at end of file)
    size_t read(void* dst, size_t n) {
        size_t done = 0;
        uint8_t* out = static_cast<uint8_t*>(dst);
        while (done < n) {
            if (upos >= cur->size()) {
                if (eof) break;
                if (!load_block(next_coffset)) { eof = true; failed = true; break; }
                if (eof) break;
                if (cur->empty()) continue;           // empty block (e.g. the EOF marker)
            }
            const size_t take = std::min(n - done, cur->size() - upos);
            if (out) std::memcpy(out + done, cur->data() + upos, take);      // (dst == nullptr: skip n bytes)
            upos += take;
            done += take;
        }
        return done;
    }
    // n bytes in place when they all lie in the current block (the cursor moves past them); nullptr otherwise
    const uint8_t* peek(size_t n) {
        if (upos + n > cur->size()) return nullptr;
        const uint8_t* p = cur->data() + upos;
        upos += n;
        return p;
    }
    uint64_t tell() const { return ((uint64_t)block_coffset << 16) | (uint64_t)upos; }
};

inline uint32_t le32(const uint8_t* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
inline uint64_t le64(const uint8_t* p) { return (uint64_t)le32(p) | ((uint64_t)le32(p + 4) << 32); }

struct ReadSet {
    std::vector<int64_t> pos, pos_end, seq_offset{0}, cigar_offset{0}, name_offset{0};
    std::vector<uint8_t> reverse, qual;
    std::vector<int32_t> mapq, flags, hp, cigar_op, cigar_len;
    std::string seq, names;
    void clear() {      // keeps the capacity: the next region's reads land in already-mapped memory
        pos.clear(); pos_end.clear(); reverse.clear(); qual.clear(); mapq.clear(); flags.clear(); hp.clear();
        cigar_op.clear(); cigar_len.clear(); seq.clear(); names.clear();
        seq_offset.assign(1, 0); cigar_offset.assign(1, 0); name_offset.assign(1, 0);
    }
};

}  // namespace

struct pa_bam {
    Bgzf bg;
    std::string text;
    std::vector<std::string> names;
    std::vector<int64_t> lengths;
    uint64_t first_record = 0;                                 // virtual offset after the header
    bool has_index = false;
    std::vector<std::vector<uint64_t>> ioff;                   // linear index per reference
    std::vector<uint64_t> ref_min;                             // smallest chunk begin per reference (0 = none)
    ReadSet reads;
    std::vector<char> scratch_seq;                             // one read's decoded bases / qualities while it is clipped
    std::vector<uint8_t> scratch_qual;
    std::vector<std::pair<int32_t, int32_t>> pack_pairs;       // pa_bam_pack_regions: (region, read) as the walk finds them
    std::vector<int64_t> pack_closed;
    std::string path;
    int span_fd = -1;                                          // pa_bam_read_span: its own descriptor (pread, no shared position)
    std::vector<int64_t> span_members;                         // file offsets of the members of the last pa_bam_read_span
    int64_t file_bytes = -1;
};

namespace {

bool load_bai(pa_bam* b, const std::string& path) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    std::vector<uint8_t> raw;
    uint8_t buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) raw.insert(raw.end(), buf, buf + n);
    fclose(f);
    if (raw.size() < 8 || std::memcmp(raw.data(), "BAI\1", 4) != 0) return false;
    size_t p = 4;
    const uint32_t n_ref = le32(&raw[p]);
    p += 4;
    b->ioff.assign(n_ref, {});
    b->ref_min.assign(n_ref, 0);
    for (uint32_t r = 0; r < n_ref; ++r) {
        if (p + 4 > raw.size()) return false;
        const uint32_t n_bin = le32(&raw[p]);
        p += 4;
        for (uint32_t i = 0; i < n_bin; ++i) {
            if (p + 8 > raw.size()) return false;
            const uint32_t bin = le32(&raw[p]);
            const uint32_t n_chunk = le32(&raw[p + 4]);
            p += 8;
            if (p + 16ull * n_chunk > raw.size()) return false;
            if (bin != 37450)                                   // pseudo-bin with metadata
                for (uint32_t c = 0; c < n_chunk; ++c) {
                    const uint64_t beg = le64(&raw[p + 16 * c]);
                    if (b->ref_min[r] == 0 || beg < b->ref_min[r]) b->ref_min[r] = beg;
                }
            p += 16ull * n_chunk;
        }
        if (p + 4 > raw.size()) return false;
        const uint32_t n_intv = le32(&raw[p]);
        p += 4;
        if (p + 8ull * n_intv > raw.size()) return false;
        b->ioff[r].resize(n_intv);
        for (uint32_t i = 0; i < n_intv; ++i) b->ioff[r][i] = le64(&raw[p + 8 * i]);
        p += 8ull * n_intv;
    }
    return true;
}

const char kSeqNt16[] = "=ACMGRSVTWYHKDBN";
struct SeqPairs {                         // byte of two 4-bit codes -> its two letters (first base in the high nibble)
    uint16_t t[256];
    SeqPairs() {
        for (int b = 0; b < 256; ++b) {
            const char two[2] = {kSeqNt16[b >> 4], kSeqNt16[b & 15]};
            std::memcpy(&t[b], two, 2);
        }
    }
    const uint16_t& operator[](uint8_t b) const { return t[b]; }
};
const SeqPairs kSeqPairs;

int aux_size(uint8_t t) {
    switch (t) {
        case 'A': case 'c': case 'C': return 1;
        case 's': case 'S': return 2;
        case 'f': case 'i': case 'I': return 4;
        case 'd': return 8;
        default: return -1;
    }
}

// HP:<integer> from the auxiliary block (0 when absent or malformed)
int parse_hp(const uint8_t* s, const uint8_t* end) {
    int hp = 0;
    while (end - s >= 4) {
        const bool is_hp = s[0] == 'H' && s[1] == 'P';
        const uint8_t type = s[2];
        s += 3;
        switch (type) {
            case 'A': s += 1; break;
            case 'c': case 'C': case 's': case 'S': case 'i': case 'I': {
                const int sz = aux_size(type);
                if (end - s < sz) return hp;
                if (is_hp) {
                    switch (type) {
                        case 'c': hp = (int8_t)s[0]; break;
                        case 'C': hp = s[0]; break;
                        case 's': hp = (int16_t)(s[0] | (s[1] << 8)); break;
                        case 'S': hp = s[0] | (s[1] << 8); break;
                        default: hp = (int32_t)le32(s); break;
                    }
                }
                s += sz;
                break;
            }
            case 'f': if (end - s < 4) return hp; s += 4; break;
            case 'Z': case 'H':
                while (s < end && *s) ++s;
                if (s >= end) return hp;
                ++s;
                break;
            case 'B': {
                if (end - s < 5) return hp;
                const int esz = aux_size(s[0]);
                if (esz < 0) return hp;
                const uint32_t cnt = le32(s + 1);
                s += 5 + (size_t)cnt * esz;
                break;
            }
            default: return hp;
        }
    }
    return hp;
}

// CG:B,I -- the real CIGAR of a record with more than 65535 operations (SAM spec 4.2.2: the core field then holds the
// placeholder <l_seq>S<ref_len>N).  Returns a pointer to the little-endian uint32 array and its length, or nullptr.
const uint8_t* find_cg(const uint8_t* s, const uint8_t* end, uint32_t* count) {
    while (end - s >= 4) {
        const bool is_cg = s[0] == 'C' && s[1] == 'G';
        const uint8_t type = s[2];
        s += 3;
        switch (type) {
            case 'A': s += 1; break;
            case 'c': case 'C': case 's': case 'S': case 'i': case 'I': case 'f': case 'd': {
                const int sz = aux_size(type);
                if (end - s < sz) return nullptr;
                s += sz;
                break;
            }
            case 'Z': case 'H':
                while (s < end && *s) ++s;
                if (s >= end) return nullptr;
                ++s;
                break;
            case 'B': {
                if (end - s < 5) return nullptr;
                const int esz = aux_size(s[0]);
                if (esz < 0) return nullptr;
                const uint32_t cnt = le32(s + 1);
                if ((uint64_t)(end - s - 5) < (uint64_t)cnt * esz) return nullptr;
                if (is_cg && (s[0] == 'I' || s[0] == 'i')) {
                    *count = cnt;
                    return s + 5;
                }
                s += 5 + (size_t)cnt * esz;
                break;
            }
            default: return nullptr;
        }
    }
    return nullptr;
}

}  // namespace

extern "C" {

const char* pa_bam_last_error(void) { return g_bam_err.c_str(); }

int pa_bam_open(const char* path, pa_bam** out) {
    if (!path || !out) return bam_fail(-1, "null argument");
    auto* b = new pa_bam();
    b->bg.fp = fopen(path, "rb");
    if (!b->bg.fp) { delete b; return bam_fail(-2, std::string("INVALID BAM FILE. PLEASE CHECK IF PATH IS CORRECT: ") + path); }
    uint8_t hdr[8];
    if (!b->bg.load_block(0) || b->bg.read(hdr, 8) != 8 || std::memcmp(hdr, "BAM\1", 4) != 0) {
        fclose(b->bg.fp); delete b;
        return bam_fail(-3, std::string("HEADER ERROR: INVALID BAM FILE: ") + path);
    }
    const uint32_t l_text = le32(hdr + 4);
    b->text.resize(l_text);
    uint8_t w[4];
    bool ok = b->bg.read(b->text.data(), l_text) == l_text && b->bg.read(w, 4) == 4;
    const uint32_t n_ref = ok ? le32(w) : 0;
    for (uint32_t i = 0; ok && i < n_ref; ++i) {
        ok = b->bg.read(w, 4) == 4;
        const uint32_t l_name = ok ? le32(w) : 0;
        std::string name(l_name, '\0');
        ok = ok && b->bg.read(name.data(), l_name) == l_name && b->bg.read(w, 4) == 4;
        if (!name.empty() && name.back() == '\0') name.pop_back();
        b->names.push_back(name);
        b->lengths.push_back(ok ? le32(w) : 0);
    }
    if (!ok) { fclose(b->bg.fp); delete b; return bam_fail(-3, std::string("HEADER ERROR: truncated BAM header: ") + path); }
    while (!b->text.empty() && b->text.back() == '\0') b->text.pop_back();
    b->first_record = b->bg.tell();
    b->path = path;
    const std::string p(path);
    b->has_index = load_bai(b, p + ".bai");
    if (!b->has_index && p.size() > 4 && p.substr(p.size() - 4) == ".bam") b->has_index = load_bai(b, p.substr(0, p.size() - 4) + ".bai");
    *out = b;
    return 0;
}

void pa_bam_close(pa_bam* b) {
    if (!b) return;
    if (b->bg.fp) fclose(b->bg.fp);
    if (b->span_fd >= 0) close(b->span_fd);
    delete b;
}

int pa_bam_has_index(pa_bam* b) { return b && b->has_index ? 1 : 0; }
int pa_bam_n_targets(pa_bam* b) { return b ? (int)b->names.size() : 0; }

int pa_bam_target(pa_bam* b, int32_t i, char* name, int32_t cap, int64_t* length) {
    if (!b || i < 0 || i >= (int)b->names.size()) return bam_fail(-1, "target index out of range");
    if (name && cap > 0) {
        std::strncpy(name, b->names[i].c_str(), cap - 1);
        name[cap - 1] = '\0';
    }
    if (length) *length = b->lengths[i];
    return (int)b->names[i].size();
}

int pa_bam_header_text(pa_bam* b, char* buf, int64_t cap, int64_t* needed) {
    if (!b) return bam_fail(-1, "null handle");
    if (needed) *needed = (int64_t)b->text.size() + 1;
    if (buf && cap > (int64_t)b->text.size()) std::memcpy(buf, b->text.c_str(), b->text.size() + 1);
    return 0;
}

int pa_bam_get_reads(pa_bam* b, const char* contig, int64_t start, int64_t stop, int32_t include_supplementary,
                     int32_t min_mapq, int32_t min_baseq, int64_t* n_reads, int64_t* seq_bytes, int64_t* n_cigar,
                     int64_t* name_bytes) {
    (void)min_baseq;   // only feeds type_read.bad_indicies in the reference, which the encoders never read
    if (!b || !contig) return bam_fail(-1, "null argument");
    ReadSet& rs = b->reads;
    rs.clear();
    int tid = -1;
    for (size_t i = 0; i < b->names.size(); ++i)
        if (b->names[i] == contig) tid = (int)i;
    if (tid < 0) return bam_fail(-4, std::string("contig not in the BAM header: ") + contig);

    uint64_t from = b->first_record;
    bool nothing = false;
    if (b->has_index && tid < (int)b->ioff.size()) {
        const auto& lin = b->ioff[tid];
        int64_t w = std::max<int64_t>(0, start) >> 14;
        uint64_t off = 0;
        if (!lin.empty()) {
            if (w >= (int64_t)lin.size()) w = (int64_t)lin.size() - 1;
            for (int64_t k = w; k >= 0 && off == 0; --k) off = lin[k];
        }
        if (off == 0) off = b->ref_min[tid];
        if (off == 0) nothing = true;     // no alignments on this reference
        from = off;
    }
    b->bg.failed = false;
    if (!nothing && !b->bg.seek(from)) return bam_fail(-5, "BGZF seek failed (corrupt file or index)");

    std::vector<uint8_t> rec;
    while (!nothing) {
        uint8_t w4[4];
        const size_t got4 = b->bg.read(w4, 4);
        if (b->bg.failed) return bam_fail(-5, "corrupt or truncated BGZF block");
        if (got4 != 4) {
            if (got4 != 0) return bam_fail(-6, "truncated BAM record");
            break;
        }
        const uint32_t block_size = le32(w4);
        if (block_size < 32) return bam_fail(-6, "corrupt BAM record");
        // A record that lies inside the current (inflated, cached) block is parsed where it is -- nine in ten with 64 KiB
        // blocks and reads of a few kb; one that crosses a block boundary is copied out first.
        const uint8_t* R = b->bg.peek(block_size);
        if (!R) {
            rec.resize(block_size);
            if (b->bg.read(rec.data(), block_size) != block_size)
                return bam_fail(b->bg.failed ? -5 : -6, b->bg.failed ? "corrupt or truncated BGZF block" : "truncated BAM record");
            R = rec.data();
        }
        const int32_t ref_id = (int32_t)le32(R);
        const int32_t pos = (int32_t)le32(R + 4);
        const uint32_t l_read_name = R[8];
        const int32_t mapq = R[9];
        const uint32_t n_cigar_op = R[12] | (R[13] << 8);
        const uint32_t flag = R[14] | (R[15] << 8);
        const uint32_t l_seq = le32(R + 16);
        if (ref_id != tid) {
            if (ref_id > tid || ref_id < 0) break;      // sorted file: past the contig (unmapped reads come last)
            continue;
        }
        if (pos >= stop) break;
        const size_t o_name = 32, o_cigar = o_name + l_read_name, o_seq = o_cigar + 4ull * n_cigar_op,
                     o_qual = o_seq + (l_seq + 1) / 2, o_aux = o_qual + l_seq;
        if (o_aux > block_size) return bam_fail(-6, "corrupt BAM record");
        // reads with more than 65535 CIGAR operations (ultra-long nanopore reads) keep their CIGAR in the CG:B,I tag and a
        // placeholder <l_seq>S<ref_len>N in the core field; htslib (the reference's reader) swaps it in transparently
        const uint8_t* cig = R + o_cigar;
        uint32_t n_cig = n_cigar_op;
        // same tests as htslib's bam_tag2cigar (sam.c): first operation <l_seq>S; a CG tag of type B,I (or B,i) with at least
        // as many operations as the core field; anything else keeps the core CIGAR as it is
        if (n_cigar_op >= 1 && (le32(cig) & 15) == 4 && (le32(cig) >> 4) == l_seq) {
            uint32_t cnt = 0;
            const uint8_t* real = find_cg(R + o_aux, R + block_size, &cnt);
            if (real && cnt >= n_cigar_op && cnt < (1u << 29)) {
                cig = real;
                n_cig = cnt;
            }
        }
        // The operations in front of the region only move the two cursors (a read of several kb that reaches a 1 kb region
        // has hundreds of them: this loop was a third of a query's time when it ran through the clipping switch below):
        // match runs that end at or before `start`, inserts / soft clips and deletions / skips that begin before it -- exactly
        // what the clipping loop does with them (:176-303: nothing is kept before the first base inside the region).  A read
        // whose operations run out here ends in front of the region.
        uint32_t k0 = 0;
        int64_t rpos = pos, ridx = 0;
        for (; k0 < n_cig; ++k0) {
            const uint32_t c = le32(cig + 4 * k0);
            const int op = c & 15;
            const int64_t len = c >> 4;
            if (op == 0 || op == 7 || op == 8) {
                if (rpos + len > start) break;
                rpos += len;
                ridx += len;
            } else if (op == 1 || op == 4) {
                if (rpos >= start) break;
                ridx += len;
            } else if (op == 2 || op == 3) {
                if (rpos >= start) break;
                rpos += len;
            }
        }
        if (k0 == n_cig) continue;

        // ---- filters of get_reads (:138-151) ----
        if (flag & (0x200 | 0x400 | 0x100 | 0x4)) continue;        // qc-fail, duplicate, secondary, unmapped
        if (!include_supplementary && (flag & 0x800)) continue;
        if (mapq < min_mapq) continue;

        // ---- clip to [start, stop] (:176-303) ----
        const uint8_t* seqi = R + o_seq;
        const uint8_t* qual = R + o_qual;
        const size_t cig0 = rs.cigar_op.size();
        if (rs.cigar_op.capacity() < cig0 + (n_cig - k0)) {
            rs.cigar_op.reserve(2 * (cig0 + (n_cig - k0)));
            rs.cigar_len.reserve(2 * (cig0 + (n_cig - k0)));
        }
        int64_t pos_start = -1, pos_end = -1;
        // the kept runs are decoded into a scratch row of the handle (4-bit codes -> upper-case letters two at a time,
        // qualities copied) and appended to the set when the read is done: what is kept is a region's worth of a read that
        // may be many kb long
        if (b->scratch_seq.size() < l_seq) {
            b->scratch_seq.resize(l_seq);
            b->scratch_qual.resize(l_seq);
        }
        char* seq_out = b->scratch_seq.data();
        uint8_t* qual_out = b->scratch_qual.data();
        size_t written = 0;
        // What is kept of a read is ONE stretch of its bases: from the first base inside the region on, match runs and the
        // inserts / soft clips behind them follow each other in the read until the region ends (deletions and skips hold no
        // bases).  The runs are therefore only counted here and the stretch is decoded in one go below.
        int64_t first_idx = -1;
        auto push_run = [&](int64_t idx, int64_t count) {
            if (first_idx < 0) first_idx = idx;
            written += (size_t)count;
        };
        for (uint32_t k = k0; k < n_cig; ++k) {
            const uint32_t c = le32(cig + 4 * k);
            const int op = c & 15;
            const int64_t len = c >> 4;
            if (rpos > stop) break;
            int64_t kept = 0;
            switch (op) {
                case 0: case 7: case 8: {                   // M, =, X
                    int64_t skip = 0;
                    if (rpos < start) {
                        skip = std::min<int64_t>(start - rpos, len);
                        ridx += skip;
                        rpos += skip;
                    }
                    kept = std::min<int64_t>(len - skip, stop - rpos + 1);   // bases at positions rpos .. stop
                    if (kept > 0) {
                        if (pos_start == -1) { pos_start = rpos; pos_end = rpos; }
                        push_run(ridx, kept);
                        pos_end += kept;
                        ridx += kept;
                        rpos += kept;
                    } else {
                        kept = 0;
                    }
                    break;
                }
                case 4: case 1:                               // S, I: kept only behind an anchored position
                    if (rpos >= start && rpos <= stop && pos_start != -1) {
                        push_run(ridx, len);
                        kept = len;
                    }
                    ridx += len;
                    break;
                case 3: case 2:                               // N, D
                    if (rpos >= start && rpos <= stop && pos_start != -1) {
                        for (int64_t i = 0; i < len; ++i) {
                            if (rpos > stop) break;
                            ++kept;
                            ++pos_end;
                            ++rpos;
                        }
                    } else {
                        rpos += len;
                    }
                    break;
                default:                                      // H, P: nothing
                    break;
            }
            if (kept > 0) {
                rs.cigar_op.push_back(op);
                rs.cigar_len.push_back((int32_t)kept);
            }
        }
        if (written == 0) {                                   // no base inside the region: read dropped (:432)
            rs.cigar_op.resize(cig0);
            rs.cigar_len.resize(cig0);
            continue;
        }
        if (first_idx < 0 || (uint64_t)first_idx + written > l_seq) {
            // the CIGAR walks over more bases than the record holds: a record without its bases (SEQ '*', l_seq 0 -- legal,
            // and nothing to pile up) is left out; anything else is a corrupt record, not something to read past
            rs.cigar_op.resize(cig0);
            rs.cigar_len.resize(cig0);
            if (l_seq == 0) continue;
            return bam_fail(-6, "corrupt BAM record (CIGAR longer than the read)");
        }
        {   // 4-bit codes -> upper-case letters, two per byte through a pair table; qualities as they are
            int64_t i = 0, idx = first_idx;
            const int64_t count = (int64_t)written;
            if ((idx & 1) && count > 0) { seq_out[0] = kSeqNt16[seqi[idx >> 1] & 15]; i = 1; }
            const uint8_t* src = seqi + ((idx + i) >> 1);
            for (; i + 1 < count; i += 2, ++src) std::memcpy(seq_out + i, &kSeqPairs[*src], 2);
            if (i < count) seq_out[i] = kSeqNt16[*src >> 4];
            std::memcpy(qual_out, qual + first_idx, written);
        }
        rs.seq.insert(rs.seq.end(), seq_out, seq_out + written);
        rs.qual.insert(rs.qual.end(), qual_out, qual_out + written);
        rs.pos.push_back(pos_start);
        rs.pos_end.push_back(pos_end);
        rs.reverse.push_back((flag & 0x10) ? 1 : 0);
        rs.mapq.push_back(mapq);
        rs.flags.push_back((int32_t)flag);
        rs.hp.push_back(parse_hp(R + o_aux, R + block_size));
        rs.seq_offset.push_back((int64_t)rs.seq.size());
        rs.cigar_offset.push_back((int64_t)rs.cigar_op.size());
        rs.names.append(reinterpret_cast<const char*>(R + o_name), l_read_name ? l_read_name - 1 : 0);
        rs.names.push_back('\0');
        rs.name_offset.push_back((int64_t)rs.names.size());
    }
    if (n_reads) *n_reads = (int64_t)rs.pos.size();
    if (seq_bytes) *seq_bytes = (int64_t)rs.seq.size();
    if (n_cigar) *n_cigar = (int64_t)rs.cigar_op.size();
    if (name_bytes) *name_bytes = (int64_t)rs.names.size();
    return 0;
}

// ---- packed form for the GPU encoder: no clipping, no decoding on the host --------------------------------------------
// What get_reads does per (read, region) on the host -- the walk that clips the read to the region, 4-bit codes -> letters --
// is left to the device (pepper_amd/csrc/encoder.hip, unpack_clip_kernel); the host inflates the BGZF blocks, walks the record
// headers, applies get_reads' filters (:138-151) and its region test, and copies each kept record's
//     CIGAR words | 4-bit bases | qualities
// -- one contiguous slice of the record as BAM stores it -- ONCE into the caller's arena (a page-locked buffer of the
// encoder), however many of the batch's regions the read reaches.  1.5 bytes per base + 4 per operation cross PCIe instead of
// 2 + 8, and the host's per-base work is one memcpy.
namespace {
// The walk shared by pa_bam_pack_regions (records from the BGZF reader, slices copied into the arena),
// pa_bam_pack_inflated (mem != NULL: records in place in an inflated span, data_off = the slice's offset in the span) and
// pa_bam_pack_headers (hdrs != NULL: the same span, its record headers already read out by the device's walk).
int pack_walk(pa_bam* b, int tid, bool nothing, const uint8_t* mem, int64_t mem_bytes, int64_t mem_first, bool mem_final,
              const pa_record_header* hdrs, int64_t n_hdrs, int32_t n_regions, const int64_t* start, const int64_t* stop, int32_t include_supplementary, int32_t min_mapq,
              uint8_t* arena, int64_t arena_cap, pa_packed_read* reads, int32_t reads_cap, int32_t* pair_read, int32_t pairs_cap,
              int32_t* region_pairs, int32_t* n_done, int64_t* counts) {
    // pairs arrive read by read; they leave grouped by region
    std::vector<std::pair<int32_t, int32_t>>& pairs = b->pack_pairs;     // (region, read)
    pairs.clear();
    int64_t used = 0;
    int32_t n_reads = 0;
    int r_lo = 0;                       // regions in front of r_lo end at or before the current record's position
    bool full = false, cut = false;
    int64_t mem_at = mem_first;
    const int64_t last_stop = stop[n_regions - 1];
    std::vector<uint8_t> rec;
    // what was complete when region k closed (every record with pos < stop[k] seen): reads, pairs, arena bytes
    std::vector<int64_t>& closed = b->pack_closed;
    closed.assign((size_t)n_regions * 3, 0);
    int n_closed = 0;
    auto close_up_to = [&](int64_t pos) {            // regions whose stop <= pos can get no further read
        while (n_closed < n_regions && stop[n_closed] <= pos) {
            closed[(size_t)n_closed * 3] = n_reads;
            closed[(size_t)n_closed * 3 + 1] = (int64_t)pairs.size();
            closed[(size_t)n_closed * 3 + 2] = used;
            ++n_closed;
        }
    };
    int64_t hk = 0;
    while (!nothing) {
        uint32_t block_size = 0;
        const uint8_t* R = nullptr;
        int32_t ref_id, pos, mapq;
        uint32_t n_cigar_op, flag, l_seq;
        size_t o_cigar = 0, o_seq = 0, o_aux = 0;
        const pa_record_header* H = nullptr;
        if (hdrs) {
            // the device's walk has read the headers: one entry per record of the span, in file order
            if (hk >= n_hdrs) { cut = !mem_final; break; }
            H = &hdrs[hk++];
            if (H->state == 2) return bam_fail(-6, "corrupt BAM record");
            ref_id = H->ref_id; pos = H->pos; mapq = (H->flags >> 16) & 0xff;
            n_cigar_op = (uint32_t)H->n_cigar; flag = (uint32_t)H->flags & 0xffffu; l_seq = (uint32_t)H->l_seq;
        } else {
        if (mem) {
            // records in place in an inflated span: one that the span cuts off ends the walk like a full table does
            // (a final span runs into the next contig's records or to the end of the file: what it cuts off is not this contig's)
            if (mem_at + 4 > mem_bytes) { cut = !mem_final; break; }
            block_size = le32(mem + mem_at);
            if (block_size < 32) return bam_fail(-6, "corrupt BAM record");
            if (mem_at + 4 + (int64_t)block_size > mem_bytes) { cut = !mem_final; break; }
            R = mem + mem_at + 4;
            mem_at += 4 + (int64_t)block_size;
        } else {
            uint8_t w4[4];
            const size_t got4 = b->bg.read(w4, 4);
            if (b->bg.failed) return bam_fail(-5, "corrupt or truncated BGZF block");
            if (got4 != 4) {
                if (got4 != 0) return bam_fail(-6, "truncated BAM record");
                break;
            }
            block_size = le32(w4);
            if (block_size < 32) return bam_fail(-6, "corrupt BAM record");
            R = b->bg.peek(block_size);
            if (!R) {
                rec.resize(block_size);
                if (b->bg.read(rec.data(), block_size) != block_size)
                    return bam_fail(b->bg.failed ? -5 : -6, b->bg.failed ? "corrupt or truncated BGZF block" : "truncated BAM record");
                R = rec.data();
            }
        }
        ref_id = (int32_t)le32(R);
        pos = (int32_t)le32(R + 4);
        mapq = R[9];
        n_cigar_op = R[12] | (R[13] << 8);
        flag = R[14] | (R[15] << 8);
        l_seq = le32(R + 16);
        }
        if (ref_id != tid) {
            if (ref_id > tid || ref_id < 0) break;
            continue;
        }
        if (pos >= last_stop) break;
        if (!H) {
            o_cigar = 32 + (size_t)R[8];
            o_seq = o_cigar + 4ull * n_cigar_op;
            o_aux = o_seq + (l_seq + 1) / 2 + l_seq;
            if (o_aux > block_size) return bam_fail(-6, "corrupt BAM record");
        }
        close_up_to(pos);
        while (r_lo < n_regions && stop[r_lo] <= pos) ++r_lo;
        // ---- filters of get_reads (:138-151); a record without bases has nothing to pile up ----
        if (flag & (0x200 | 0x400 | 0x100 | 0x4)) continue;
        if (!include_supplementary && (flag & 0x800)) continue;
        if (mapq < min_mapq) continue;
        if (l_seq == 0) continue;
        const uint8_t* cig = H ? nullptr : R + o_cigar;
        uint32_t n_cig = n_cigar_op;
        if (!H && n_cigar_op >= 1 && (le32(cig) & 15) == 4 && (le32(cig) >> 4) == l_seq) {      // long CIGAR in the CG tag (as above)
            uint32_t cnt = 0;
            const uint8_t* real = find_cg(R + o_aux, R + block_size, &cnt);
            if (real && cnt >= n_cigar_op && cnt < (1u << 29)) {
                cig = real;
                n_cig = cnt;
            }
        }
        if (n_cig == 0) continue;                            // no alignment to walk: get_reads keeps nothing of it
        // the region test of the iterator: pos < stop and end > start, end = pos + reference length (at least pos + 1); only a
        // read that starts in front of a region needs its end
        int64_t end = H ? (int64_t)pos + std::max<int64_t>(1, H->ref_len) : -1;     // (the device's walk summed the operations)
        int first_pair = -1;
        for (int r = r_lo; r < n_regions && start[r] < (end < 0 ? (int64_t)0x7fffffffffffll : end); ++r) {
            if (pos >= stop[r]) continue;
            if (pos < start[r]) {
                if (end < 0) {
                    int64_t ref_len = 0;
                    for (uint32_t k = 0; k < n_cig; ++k) {
                        const uint32_t c = le32(cig + 4 * k);
                        const int op = c & 15;
                        if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) ref_len += c >> 4;
                    }
                    end = pos + std::max<int64_t>(1, ref_len);
                }
                if (end <= start[r]) break;                  // (starts ascend: no later region either)
            }
            if ((int64_t)pairs.size() >= pairs_cap) { full = true; break; }
            if (first_pair < 0) first_pair = (int)pairs.size();
            pairs.emplace_back(r, n_reads);
        }
        if (full) break;
        if (first_pair < 0) continue;
        const int64_t bytes = 4ll * n_cig + (l_seq + 1) / 2 + l_seq;
        int64_t at;
        if (H) {
            if (H->state == 1) return bam_fail(-8, "pack_headers: a record keeps its CIGAR in the CG tag (take pack_regions)");
            if (n_reads >= reads_cap) {
                pairs.resize((size_t)first_pair);
                full = true;
                break;
            }
            at = H->data_off;
        } else if (mem) {
            // in place: the slice is the record's own bytes (any alignment); a CIGAR kept in the CG tag is not one slice
            if (cig != R + o_cigar) return bam_fail(-8, "pack_inflated: a record keeps its CIGAR in the CG tag (take pack_regions)");
            if (n_reads >= reads_cap) {
                pairs.resize((size_t)first_pair);
                full = true;
                break;
            }
            at = (int64_t)(cig - mem);
        } else {
            at = (used + 3) & ~(int64_t)3;
            if (n_reads >= reads_cap || at + bytes + 64 > arena_cap) {
                pairs.resize((size_t)first_pair);
                full = true;
                break;
            }
            if (cig == R + o_cigar) {
                std::memcpy(arena + at, cig, (size_t)bytes);      // the three fields follow each other in the record
            } else {
                std::memcpy(arena + at, cig, 4ull * n_cig);
                std::memcpy(arena + at + 4ll * n_cig, R + o_seq, (size_t)((l_seq + 1) / 2 + l_seq));
            }
        }
        pa_packed_read& pr = reads[n_reads++];
        pr.data_off = at;
        pr.pos = pos;
        pr.n_cigar = (int32_t)n_cig;
        pr.l_seq = (int32_t)l_seq;
        pr.flags = (int32_t)(flag | ((uint32_t)mapq << 16));
        used = (mem || H) ? used + bytes : at + bytes;
    }
    if (!full && !cut) close_up_to(0x7fffffffffffffffll);        // the walk ended: every region is complete
    if (n_closed == 0) {
        if (cut) return bam_fail(-9, "pack_inflated: the span ends before the first region's last read (take a longer span)");
        return bam_fail(-7, "pack_regions: the reads of one region do not fit the arena / tables (grow them or take get_reads)");
    }
    const int64_t reads_kept = closed[(size_t)(n_closed - 1) * 3], pairs_kept = closed[(size_t)(n_closed - 1) * 3 + 1];
    // pairs of the closed regions, grouped by region (a stable counting sort: the reads of a region stay in file order)
    for (int64_t k = 0; k < pairs_kept; ++k)
        if (pairs[(size_t)k].first < n_closed) region_pairs[pairs[(size_t)k].first + 1] += 1;
    for (int r = 0; r < n_regions; ++r) region_pairs[r + 1] += region_pairs[r];
    {
        std::vector<int32_t> fill(region_pairs, region_pairs + n_regions);
        for (int64_t k = 0; k < pairs_kept; ++k)
            if (pairs[(size_t)k].first < n_closed) pair_read[fill[(size_t)pairs[(size_t)k].first]++] = pairs[(size_t)k].second;
    }
    *n_done = n_closed;
    if (counts) {
        counts[0] = reads_kept;
        counts[1] = region_pairs[n_closed];
        counts[2] = closed[(size_t)(n_closed - 1) * 3 + 2];
    }
    return 0;
}
}  // namespace

int pa_bam_pack_regions(pa_bam* b, const char* contig, int32_t n_regions, const int64_t* start, const int64_t* stop,
                        int32_t include_supplementary, int32_t min_mapq, uint8_t* arena, int64_t arena_cap,
                        pa_packed_read* reads, int32_t reads_cap, int32_t* pair_read, int32_t pairs_cap,
                        int32_t* region_pairs, int32_t* n_done, int64_t* counts) {
    if (!b || !contig || n_regions < 0 || (n_regions > 0 && (!start || !stop)) || !arena || !reads || !pair_read || !region_pairs || !n_done)
        return bam_fail(-1, "null argument");
    *n_done = 0;
    for (int r = 0; r <= n_regions; ++r) region_pairs[r] = 0;
    if (counts) counts[0] = counts[1] = counts[2] = 0;
    if (n_regions == 0) return 0;
    for (int r = 0; r < n_regions; ++r)
        if (stop[r] < start[r] || (r > 0 && (start[r] < start[r - 1] || stop[r] < stop[r - 1])))
            return bam_fail(-1, "pack_regions: regions must be ascending in start and stop");
    int tid = -1;
    for (size_t i = 0; i < b->names.size(); ++i)
        if (b->names[i] == contig) tid = (int)i;
    if (tid < 0) return bam_fail(-4, std::string("contig not in the BAM header: ") + contig);

    uint64_t from = b->first_record;
    bool nothing = false;
    if (b->has_index && tid < (int)b->ioff.size()) {
        const auto& lin = b->ioff[tid];
        int64_t w = std::max<int64_t>(0, start[0]) >> 14;
        uint64_t off = 0;
        if (!lin.empty()) {
            if (w >= (int64_t)lin.size()) w = (int64_t)lin.size() - 1;
            for (int64_t k = w; k >= 0 && off == 0; --k) off = lin[k];
        }
        if (off == 0) off = b->ref_min[tid];
        if (off == 0) nothing = true;
        from = off;
    }
    b->bg.failed = false;
    if (!nothing && !b->bg.seek(from)) return bam_fail(-5, "BGZF seek failed (corrupt file or index)");

    return pack_walk(b, tid, nothing, nullptr, 0, 0, false, nullptr, 0, n_regions, start, stop, include_supplementary, min_mapq, arena, arena_cap,
                     reads, reads_cap, pair_read, pairs_cap, region_pairs, n_done, counts);
}

namespace {
int find_tid(pa_bam* b, const char* contig) {
    for (size_t i = 0; i < b->names.size(); ++i)
        if (b->names[i] == contig) return (int)i;
    return -1;
}
bool open_span_fd(pa_bam* b) {
    if (b->span_fd >= 0) return true;
    b->span_fd = open(b->path.c_str(), O_RDONLY);
    if (b->span_fd < 0) return false;
    struct stat st;
    if (fstat(b->span_fd, &st) != 0) return false;
    b->file_bytes = (int64_t)st.st_size;
    return true;
}
}  // namespace

// The host's counterpart of the device inflate, for baselines and cross-checks: the same member tables, libdeflate (zlib
// where it is not installed) on n_threads threads, members dealt round robin.
int pa_bgzf_inflate_host(const uint8_t* comp, int64_t comp_bytes, int32_t n_blocks, const int64_t* comp_off, const int32_t* comp_len,
                         const int64_t* out_off, const int32_t* out_len, uint8_t* out, int64_t out_bytes, int32_t n_threads) {
    if (n_blocks < 0 || comp_bytes < 0 || out_bytes < 0 || (n_blocks > 0 && (!comp || !comp_off || !comp_len || !out_off || !out_len)) ||
        (out_bytes > 0 && !out))
        return bam_fail(-1, "null or negative argument");
    for (int32_t k = 0; k < n_blocks; ++k)
        if (comp_off[k] < 0 || comp_len[k] < 0 || comp_off[k] + comp_len[k] > comp_bytes || out_off[k] < 0 || out_len[k] < 0 ||
            out_off[k] + out_len[k] > out_bytes)
            return bam_fail(-1, "block " + std::to_string(k) + " lies outside the buffers");
    const int nt = std::max(1, std::min<int>(n_threads, std::max(1, n_blocks)));
    std::vector<int> bad((size_t)nt, -1), crc_bad((size_t)nt, 0);
    auto work = [&](int t) {
#ifdef PA_HAVE_LIBDEFLATE
        libdeflate_decompressor* ld = libdeflate_alloc_decompressor();
#endif
        for (int32_t k = t; k < n_blocks; k += nt) {
            if (out_len[k] == 0) continue;
            bool ok;
#ifdef PA_HAVE_LIBDEFLATE
            ok = ld && libdeflate_deflate_decompress(ld, comp + comp_off[k], (size_t)comp_len[k], out + out_off[k], (size_t)out_len[k],
                                                     nullptr) == LIBDEFLATE_SUCCESS;
#else
            z_stream zs{};
            ok = inflateInit2(&zs, -15) == Z_OK;
            if (ok) {
                zs.next_in = const_cast<uint8_t*>(comp + comp_off[k]);
                zs.avail_in = (uInt)comp_len[k];
                zs.next_out = out + out_off[k];
                zs.avail_out = (uInt)out_len[k];
                ok = inflate(&zs, Z_FINISH) == Z_STREAM_END && zs.avail_out == 0;
                inflateEnd(&zs);
            }
#endif
            // (`comp` holds whole members: the CRC-32 is the 4 bytes behind the DEFLATE bytes)
            if (ok && comp_off[k] + comp_len[k] + 4 <= comp_bytes) {
                const uint8_t* t4 = comp + comp_off[k] + comp_len[k];
                const uint32_t want = t4[0] | (t4[1] << 8) | (t4[2] << 16) | ((uint32_t)t4[3] << 24);
                if (member_crc32(out + out_off[k], (size_t)out_len[k]) != want) { ok = false; crc_bad[(size_t)t] = 1; }
            }
            if (!ok && bad[(size_t)t] < 0) bad[(size_t)t] = k;
        }
#ifdef PA_HAVE_LIBDEFLATE
        if (ld) libdeflate_free_decompressor(ld);
#endif
    };
    std::vector<std::thread> threads;
    for (int t = 1; t < nt; ++t) threads.emplace_back(work, t);
    work(0);
    for (auto& th : threads) th.join();
    for (int t = 0; t < nt; ++t)
        if (bad[(size_t)t] >= 0)
            return bam_fail(-5, "BGZF block " + std::to_string(bad[(size_t)t]) +
                                    (crc_bad[(size_t)t] ? ": CRC32 of the inflated bytes differs from the member's trailer" : " does not inflate to its ISIZE"));
    return 0;
}

int pa_bam_region_span(pa_bam* b, const char* contig, int64_t start, int64_t stop, int32_t lookahead_windows,
                       int64_t* begin_coffset, int32_t* begin_uoffset, int64_t* end_coffset, int32_t* to_contig_end) {
    if (!b || !contig || !begin_coffset || !begin_uoffset || !end_coffset || !to_contig_end || stop < start || lookahead_windows < 0)
        return bam_fail(-1, "null or invalid argument");
    const int tid = find_tid(b, contig);
    if (tid < 0) return bam_fail(-4, std::string("contig not in the BAM header: ") + contig);
    if (!b->has_index || tid >= (int)b->ioff.size()) return bam_fail(-10, "region_span needs the .bai index");
    if (!open_span_fd(b)) return bam_fail(-2, "cannot open the BAM file for span reads");
    const auto& lin = b->ioff[tid];
    uint64_t off = 0;
    if (!lin.empty()) {
        int64_t w = std::max<int64_t>(0, start) >> 14;
        if (w >= (int64_t)lin.size()) w = (int64_t)lin.size() - 1;
        for (int64_t k = w; k >= 0 && off == 0; --k) off = lin[k];
    }
    if (off == 0) off = b->ref_min[tid];
    if (off == 0) {                                   // no record of the contig at all
        *begin_coffset = *end_coffset = 0;
        *begin_uoffset = 0;
        *to_contig_end = 1;
        return 0;
    }
    *begin_coffset = (int64_t)(off >> 16);
    *begin_uoffset = (int32_t)(off & 0xffff);
    // the end: the first indexed record of a window `lookahead_windows` beyond the one after stop -- a record there starts at
    // or after stop unless a read longer than the lookahead reaches it (the walk notices: it must SEE a record at or beyond
    // the last stop); past the contig's windows: where the next contig's records begin, or the end of the file
    uint64_t end = 0;
    const int64_t w2 = ((std::max<int64_t>(stop, 1) - 1) >> 14) + 1 + lookahead_windows;
    for (int64_t k = w2; k < (int64_t)lin.size() && end == 0; ++k)
        if (lin[k] > off) end = lin[k];
    *to_contig_end = 0;
    if (end == 0) {
        *to_contig_end = 1;
        for (size_t t = (size_t)tid + 1; t < b->ref_min.size() && end == 0; ++t) end = b->ref_min[t];
        *end_coffset = end ? (int64_t)(end >> 16) + 1 : b->file_bytes;     // (+1: the member that holds the next contig's first record)
        return 0;
    }
    *end_coffset = (int64_t)(end >> 16) + 1;
    return 0;
}

int pa_bam_read_span(pa_bam* b, int64_t begin, int64_t end_min, int32_t extra_members, uint8_t* buf, int64_t buf_cap,
                     int64_t* comp_off, int32_t* comp_len, int64_t* out_off, int32_t* out_len, int32_t blocks_cap,
                     int32_t* n_blocks, int64_t* comp_bytes, int64_t* out_bytes, int32_t* complete) {
    if (!b || !buf || !comp_off || !comp_len || !out_off || !out_len || !n_blocks || !comp_bytes || !out_bytes || !complete ||
        begin < 0 || buf_cap < 0 || blocks_cap < 0 || extra_members < 0)
        return bam_fail(-1, "null or invalid argument");
    if (!open_span_fd(b)) return bam_fail(-2, "cannot open the BAM file for span reads");
    *n_blocks = 0;
    *comp_bytes = *out_bytes = 0;
    *complete = 0;
    b->span_members.clear();
    end_min = std::min(end_min, b->file_bytes);
    const int64_t want = std::min<int64_t>(b->file_bytes - begin, (end_min - begin) + ((int64_t)extra_members + 1) * 65536);
    if (want <= 0) { *complete = 1; return 0; }
    const int64_t take = std::min(want, buf_cap);
    int64_t got = 0;
    while (got < take) {
        const ssize_t r = pread(b->span_fd, buf + got, (size_t)(take - got), (off_t)(begin + got));
        if (r < 0) return bam_fail(-5, "read error in the BAM file");
        if (r == 0) break;
        got += r;
    }
    int64_t p = 0, at = 0;
    int32_t n = 0, extra = 0;
    bool covered = false, at_eof = false;
    while (true) {
        if (begin + p >= b->file_bytes) { covered = at_eof = true; break; }
        if (begin + p >= end_min) {
            if (extra >= extra_members) { covered = true; break; }
            ++extra;
        }
        if (p + 18 > got) break;
        const uint8_t* h = buf + p;
        if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) return bam_fail(-5, "no BGZF member where the index points");
        const int xlen = h[10] | (h[11] << 8);
        if (p + 12 + xlen > got) break;
        int bsize = -1;
        for (int q = 0; q + 4 <= xlen;) {
            const int slen = h[12 + q + 2] | (h[12 + q + 3] << 8);
            if (h[12 + q] == 'B' && h[12 + q + 1] == 'C' && slen == 2 && q + 6 <= xlen) bsize = (h[12 + q + 4] | (h[12 + q + 5] << 8)) + 1;
            q += 4 + slen;
        }
        const int clen = bsize - 12 - xlen - 8;
        if (bsize < 0 || clen < 0) return bam_fail(-5, "malformed BGZF member");
        if (p + bsize > got) break;
        if (n >= blocks_cap) break;
        const uint32_t isize = le32(h + bsize - 4);
        if (isize > 65536) return bam_fail(-5, "BGZF member larger than 64 KiB");
        b->span_members.push_back(begin + p);
        comp_off[n] = p + 12 + xlen;
        comp_len[n] = clen;
        out_off[n] = at;
        out_len[n] = (int32_t)isize;
        at += isize;
        p += bsize;
        ++n;
    }
    *n_blocks = n;
    *comp_bytes = p;
    *out_bytes = at;
    *complete = covered ? (at_eof ? 3 : 1) : 0;
    return 0;
}

int pa_bam_span_entries(pa_bam* b, const char* contig, int64_t first_record, const int64_t* out_off, int32_t n_blocks,
                        int64_t* entries, int32_t entries_cap, int32_t* n_entries) {
    if (!b || !contig || !out_off || !entries || !n_entries || entries_cap < 1 || first_record < 0 || n_blocks < 0)
        return bam_fail(-1, "null or invalid argument");
    if ((size_t)n_blocks != b->span_members.size()) return bam_fail(-1, "span_entries: not the tables of the handle's last read_span");
    const int tid = find_tid(b, contig);
    if (tid < 0) return bam_fail(-4, std::string("contig not in the BAM header: ") + contig);
    int32_t n = 0;
    entries[n++] = first_record;
    if (b->has_index && tid < (int)b->ioff.size() && n_blocks > 0) {
        const auto& members = b->span_members;
        for (const uint64_t v : b->ioff[tid]) {
            if (v == 0) continue;
            const int64_t c = (int64_t)(v >> 16);
            if (c < members.front() || c > members.back()) continue;
            const auto it = std::lower_bound(members.begin(), members.end(), c);
            if (it == members.end() || *it != c) continue;                   // (not a member start: a foreign index)
            const int64_t at = out_off[it - members.begin()] + (int64_t)(v & 0xffff);
            if (at <= entries[n - 1]) continue;
            if (n >= entries_cap) break;
            entries[n++] = at;
        }
    }
    *n_entries = n;
    return 0;
}

int pa_bam_pack_headers(pa_bam* b, const pa_record_header* headers, int64_t n_headers, int32_t data_is_final, const char* contig,
                        int32_t n_regions, const int64_t* start, const int64_t* stop, int32_t include_supplementary, int32_t min_mapq,
                        pa_packed_read* reads, int32_t reads_cap, int32_t* pair_read, int32_t pairs_cap, int32_t* region_pairs,
                        int32_t* n_done, int64_t* counts) {
    if (!b || !contig || n_regions < 0 || (n_regions > 0 && (!start || !stop)) || (n_headers > 0 && !headers) || n_headers < 0 ||
        !reads || !pair_read || !region_pairs || !n_done)
        return bam_fail(-1, "null argument");
    *n_done = 0;
    for (int r = 0; r <= n_regions; ++r) region_pairs[r] = 0;
    if (counts) counts[0] = counts[1] = counts[2] = 0;
    if (n_regions == 0) return 0;
    for (int r = 0; r < n_regions; ++r)
        if (stop[r] < start[r] || (r > 0 && (start[r] < start[r - 1] || stop[r] < stop[r - 1])))
            return bam_fail(-1, "pack_headers: regions must be ascending in start and stop");
    const int tid = find_tid(b, contig);
    if (tid < 0) return bam_fail(-4, std::string("contig not in the BAM header: ") + contig);
    static const pa_record_header none{};
    return pack_walk(b, tid, false, nullptr, 0, 0, data_is_final != 0, headers ? headers : &none, n_headers, n_regions, start, stop,
                     include_supplementary, min_mapq, nullptr, 0, reads, reads_cap, pair_read, pairs_cap, region_pairs, n_done, counts);
}

int pa_bam_pack_inflated(pa_bam* b, const uint8_t* data, int64_t data_bytes, int64_t first_record, int32_t data_is_final,
                         const char* contig, int32_t n_regions, const int64_t* start, const int64_t* stop,
                         int32_t include_supplementary, int32_t min_mapq, pa_packed_read* reads, int32_t reads_cap,
                         int32_t* pair_read, int32_t pairs_cap, int32_t* region_pairs, int32_t* n_done, int64_t* counts) {
    if (!b || !contig || n_regions < 0 || (n_regions > 0 && (!start || !stop)) || (data_bytes > 0 && !data) || data_bytes < 0 ||
        first_record < 0 || !reads || !pair_read || !region_pairs || !n_done)
        return bam_fail(-1, "null argument");
    *n_done = 0;
    for (int r = 0; r <= n_regions; ++r) region_pairs[r] = 0;
    if (counts) counts[0] = counts[1] = counts[2] = 0;
    if (n_regions == 0) return 0;
    for (int r = 0; r < n_regions; ++r)
        if (stop[r] < start[r] || (r > 0 && (start[r] < start[r - 1] || stop[r] < stop[r - 1])))
            return bam_fail(-1, "pack_inflated: regions must be ascending in start and stop");
    const int tid = find_tid(b, contig);
    if (tid < 0) return bam_fail(-4, std::string("contig not in the BAM header: ") + contig);
    static const uint8_t none = 0;
    return pack_walk(b, tid, false, data ? data : &none, data_bytes, first_record, data_is_final != 0, nullptr, 0, n_regions, start, stop,
                     include_supplementary, min_mapq, nullptr, 0, reads, reads_cap, pair_read, pairs_cap, region_pairs, n_done, counts);
}

int pa_bam_copy_reads(pa_bam* b, int64_t* pos, int64_t* pos_end, uint8_t* reverse, int32_t* mapq, int32_t* flags,
                      int32_t* hp, int64_t* seq_offset, char* seq, uint8_t* qual, int64_t* cigar_offset,
                      int32_t* cigar_op, int32_t* cigar_len, char* names) {
    if (!b) return bam_fail(-1, "null handle");
    const ReadSet& rs = b->reads;
    const size_t n = rs.pos.size();
    auto cp = [](void* dst, const void* src, size_t bytes) { if (dst && bytes) std::memcpy(dst, src, bytes); };
    cp(pos, rs.pos.data(), n * 8);
    cp(pos_end, rs.pos_end.data(), n * 8);
    cp(reverse, rs.reverse.data(), n);
    cp(mapq, rs.mapq.data(), n * 4);
    cp(flags, rs.flags.data(), n * 4);
    cp(hp, rs.hp.data(), n * 4);
    cp(seq_offset, rs.seq_offset.data(), (n + 1) * 8);
    cp(seq, rs.seq.data(), rs.seq.size());
    cp(qual, rs.qual.data(), rs.qual.size());
    cp(cigar_offset, rs.cigar_offset.data(), (n + 1) * 8);
    cp(cigar_op, rs.cigar_op.data(), rs.cigar_op.size() * 4);
    cp(cigar_len, rs.cigar_len.data(), rs.cigar_len.size() * 4);
    cp(names, rs.names.data(), rs.names.size());
    return 0;
}

}  // extern "C"
