// Minimal HDF5 reader/writer behind include/pepper_amd_io.h (libhdf5 1.10 C API).
// Host-only C++ (g++), linked against /opt/conda/lib/libhdf5.so.103.
#include "../../include/pepper_amd_io.h"

#include <hdf5.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

thread_local std::string g_err;

int fail(const std::string& msg) {
    g_err = msg;
    return -1;
}

hid_t native_type(int code) {
    switch (code) {
        case PA_H5_I8: return H5T_NATIVE_INT8;
        case PA_H5_U8: return H5T_NATIVE_UINT8;
        case PA_H5_I16: return H5T_NATIVE_INT16;
        case PA_H5_I32: return H5T_NATIVE_INT32;
        case PA_H5_I64: return H5T_NATIVE_INT64;
        case PA_H5_F32: return H5T_NATIVE_FLOAT;
        case PA_H5_F64: return H5T_NATIVE_DOUBLE;
        case PA_H5_U16: return H5T_NATIVE_UINT16;
        case PA_H5_U32: return H5T_NATIVE_UINT32;
        case PA_H5_U64: return H5T_NATIVE_UINT64;
        default: return -1;
    }
}

int64_t type_bytes(int code) {
    switch (code) {
        case PA_H5_I8: case PA_H5_U8: return 1;
        case PA_H5_I16: case PA_H5_U16: return 2;
        case PA_H5_I32: case PA_H5_U32: case PA_H5_F32: return 4;
        default: return 8;
    }
}

struct Quiet {  // silence HDF5's automatic error stack printing for probing calls
    H5E_auto2_t fn;
    void* data;
    Quiet() {
        H5Eget_auto2(H5E_DEFAULT, &fn, &data);
        H5Eset_auto2(H5E_DEFAULT, nullptr, nullptr);
    }
    ~Quiet() { H5Eset_auto2(H5E_DEFAULT, fn, data); }
};

hid_t make_space(int rank, const int64_t* dims) {
    if (rank == 0) return H5Screate(H5S_SCALAR);
    hsize_t d[8];
    for (int i = 0; i < rank; ++i) d[i] = (hsize_t)dims[i];
    return H5Screate_simple(rank, d, nullptr);
}

// ---- direct locator: where the raw bytes of summaries/<chunk>/{image,position,index} lie in a classic-format file.
// A polish image file is a few hundred thousand tiny objects and libhdf5 spends ~30 us of CPU per dataset it opens, reads and
// closes (object header -> datatype/dataspace/layout objects, property lists, ids): ~95 us per chunk of three datasets, which on
// the 16 CPUs a GPU box grants is the wall of the whole call_consensus pipeline.  The file format itself answers
// "where are the bytes" in a few pointer hops: group object header -> symbol table message -> B-tree node -> symbol node ->
// dataset object header -> layout message.  This walks exactly that over a read-only mmap, for the subset of the format that
// h5py (libver earliest) and pa_h5_open mode 1 write: superblock v0/v1 with 8-byte offsets, version-1 object headers, version-1
// group B-trees, layout message v3 contiguous.  Anything else -- any signature, version, size or bound that is not what is
// expected -- makes it answer "don't know" and the caller reads that chunk through libhdf5 as before; it never guesses.
// (HDF5 File Format Specification 2.0, sections III.A.1 B-trees, III.B/C symbol nodes, III.D local heaps, IV.A.1.a version-1
// object headers, IV.A.2.i layout, IV.A.2.r symbol table message.)
class Direct {
public:
    static Direct* open(const char* path) {
        int fd = ::open(path, O_RDONLY | O_CLOEXEC);
        if (fd < 0) return nullptr;
        struct stat st;
        if (fstat(fd, &st) != 0 || st.st_size < 96) { ::close(fd); return nullptr; }
        void* p = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_SHARED, fd, 0);
        if (p == MAP_FAILED) { ::close(fd); return nullptr; }
        auto* d = new Direct();
        d->fd_ = fd;
        d->p_ = (const uint8_t*)p;
        d->n_ = (uint64_t)st.st_size;
        if (!d->superblock()) { delete d; return nullptr; }
        return d;
    }
    ~Direct() {
        if (p_) munmap((void*)p_, (size_t)n_);
        if (fd_ >= 0) ::close(fd_);
    }

    // the metadata is walked through the mapping (clustered in the file's metadata blocks: few pages); the raw data is read
    // with pread into the caller's array -- touching it through the mapping costs a page fault per 4 KB, 25 us per chunk here
    struct Span { const uint8_t* data = nullptr; uint64_t offset = 0, bytes = 0; };
    bool copy(const Span& s, void* out) const {
        if (s.data) { std::memcpy(out, s.data, (size_t)s.bytes); return true; }
        uint64_t done = 0;
        while (done < s.bytes) {
            const ssize_t r = pread(fd_, (char*)out + done, (size_t)(s.bytes - done), (off_t)(s.offset + done));
            if (r <= 0) return false;
            done += (uint64_t)r;
        }
        return true;
    }

    // object header address of <group_addr>/<name>; 0 when absent or not understood
    uint64_t child(uint64_t group_header, const char* name) const {
        uint64_t bt = 0, heap = 0;
        if (!symbol_table(group_header, bt, heap)) return 0;
        return lookup(bt, heap, name);
    }
    uint64_t root() const { return root_header_; }

    // <group>/a/b/c from `group_header`; 0 when absent or not understood
    uint64_t resolve(uint64_t group_header, const char* path) const {
        uint64_t at = group_header;
        std::string part;
        for (const char* c = path;; ++c) {
            if (*c == '/' || *c == 0) {
                if (!part.empty()) {
                    at = child(at, part.c_str());
                    if (!at) return 0;
                    part.clear();
                }
                if (*c == 0) return at;
            } else part.push_back(*c);
        }
    }

    // the links of a group in name order (the order of its B-tree); false = don't know
    bool children(uint64_t group_header, std::vector<std::pair<std::string, uint64_t>>& out) const {
        uint64_t bt = 0, heap = 0;
        if (!symbol_table(group_header, bt, heap)) return false;
        if (!in(heap, 32) || std::memcmp(p_ + heap, "HEAP", 4) != 0 || p_[heap + 4] != 0) return false;
        const uint64_t hsize = rd64(p_ + heap + 8), hdata = rd64(p_ + heap + 24);
        if (!in(hdata, hsize)) return false;
        return walk(bt, hdata, hsize, 0, out);
    }

    // raw bytes of a contiguous (or compact) dataset of `count` little-endian integers of `elem` bytes; false = don't know
    // count == ANY: whatever the dataspace says, reported in *count_out
    static constexpr uint64_t ANY = ~0ull;
    bool dataset(uint64_t header, uint32_t elem, uint64_t count, Span& out, uint64_t* count_out = nullptr) const {
        Msgs m;
        if (!messages(header, m) || !m.layout || !m.dtype || !m.dspace) return false;
        // datatype: class 0 (fixed point) version 1-3, little-endian, no padding games we care about; size == elem
        if (m.dtype_len < 8) return false;
        const uint8_t cv = m.dtype[0];
        if ((cv & 0x0f) != 0 || (cv >> 4) < 1 || (cv >> 4) > 3) return false;
        if (m.dtype[1] & 0x01) return false;                                  // big-endian
        if (rd32(m.dtype + 4) != elem) return false;
        if (m.dtype_len >= 12 && (rd16(m.dtype + 8) != 0 || rd16(m.dtype + 10) != 8 * elem)) return false;   // bit offset, precision
        // dataspace: simple, no permutation; product of the dimensions == count
        if (m.dspace_len < 8) return false;
        const uint8_t sv = m.dspace[0], rank = m.dspace[1];
        uint64_t at;
        if (sv == 1) at = 8;
        else if (sv == 2) { if (m.dspace[3] != 1 && !(m.dspace[3] == 0 && rank == 0)) return false; at = 4; }
        else return false;
        if (rank > 8 || m.dspace_len < at + 8ull * rank) return false;
        uint64_t npoints = 1;
        for (uint32_t k = 0; k < rank; ++k) {
            const uint64_t dim = rd64(m.dspace + at + 8 * k);
            if (dim != 0 && npoints > (1ull << 40) / dim) return false;
            npoints *= dim;
        }
        if (count == ANY) count = npoints;
        if (npoints != count) return false;
        if (count_out) *count_out = npoints;
        if (m.filters) return false;
        // layout v3: class 1 contiguous {address, size}; class 0 compact {size(2), bytes}
        if (m.layout_len < 2 || m.layout[0] != 3) return false;
        const uint64_t want = (uint64_t)elem * count;
        if (m.layout[1] == 1) {
            if (m.layout_len < 18) return false;
            const uint64_t addr = rd64(m.layout + 2), size = rd64(m.layout + 10);
            if (addr == ~0ull || size != want || !in(addr, size)) return false;
            out.data = nullptr;
            out.offset = addr;
            out.bytes = size;
            return true;
        }
        if (m.layout[1] == 0) {
            if (m.layout_len < 4) return false;
            const uint64_t size = rd16(m.layout + 2);
            if (size != want || m.layout_len < 4 + size) return false;
            out.data = m.layout + 4;
            out.bytes = size;
            return true;
        }
        return false;
    }

    // Any contiguous / compact dataset of the classes the prediction files hold: 0 fixed point (little-endian), 1 IEEE binary64
    // (little-endian), 3 fixed-width string, 9 variable-length string (16-byte references into global heap collections).
    // false = don't know (the caller reads through libhdf5).
    struct Raw { uint8_t cls = 0; uint32_t elem = 0, rank = 0; uint64_t dims[8] = {0}; uint64_t npoints = 0; Span span; };
    bool raw(uint64_t header, Raw& r) const {
        Msgs m;
        if (!header || !messages(header, m) || !m.layout || !m.dtype || !m.dspace || m.filters || m.dtype_len < 8) return false;
        const uint8_t cv = m.dtype[0];
        r.cls = cv & 0x0f;
        if ((cv >> 4) < 1 || (cv >> 4) > 3) return false;
        r.elem = rd32(m.dtype + 4);
        if (r.cls == 0) {
            if ((m.dtype[1] & 0x01) || (m.dtype_len >= 12 && (rd16(m.dtype + 8) != 0 || rd16(m.dtype + 10) != 8 * r.elem))) return false;
        } else if (r.cls == 1) {
            // little-endian, bit offset 0, precision 64, exponent at 52 (11 bits), mantissa at 0 (52 bits), bias 1023
            if (r.elem != 8 || (m.dtype[1] & 0x41) || m.dtype_len < 20 || rd16(m.dtype + 8) != 0 || rd16(m.dtype + 10) != 64 ||
                m.dtype[12] != 52 || m.dtype[13] != 11 || m.dtype[14] != 0 || m.dtype[15] != 52 || rd32(m.dtype + 16) != 1023) return false;
        } else if (r.cls == 3) {
            if (r.elem == 0) return false;
        } else if (r.cls == 9) {
            if ((m.dtype[1] & 0x0f) != 1 || r.elem != 16) return false;       // a variable-length STRING
        } else return false;
        if (m.dspace_len < 8) return false;
        const uint8_t sv = m.dspace[0];
        r.rank = m.dspace[1];
        uint64_t at;
        if (sv == 1) at = 8;
        else if (sv == 2) { if (m.dspace[3] != 1 && !(m.dspace[3] == 0 && r.rank == 0)) return false; at = 4; }
        else return false;
        if (r.rank > 8 || m.dspace_len < at + 8ull * r.rank) return false;
        r.npoints = 1;
        for (uint32_t k = 0; k < r.rank; ++k) {
            r.dims[k] = rd64(m.dspace + at + 8 * k);
            if (r.dims[k] != 0 && r.npoints > (1ull << 40) / r.dims[k]) return false;
            r.npoints *= r.dims[k];
        }
        if (m.layout_len < 2 || m.layout[0] != 3) return false;
        const uint64_t want = (uint64_t)r.elem * r.npoints;
        if (m.layout[1] == 1) {
            if (m.layout_len < 18) return false;
            const uint64_t addr = rd64(m.layout + 2), size = rd64(m.layout + 10);
            if (want == 0) { r.span = Span(); return true; }                 // (no storage allocated for an empty array)
            if (addr == ~0ull || size != want || !in(addr, size)) return false;
            r.span.data = p_ + addr;                                         // through the mapping: these datasets are a few KB
            r.span.offset = addr;
            r.span.bytes = size;
            return true;
        }
        if (m.layout[1] == 0) {
            if (m.layout_len < 4) return false;
            const uint64_t size = rd16(m.layout + 2);
            if (size != want || m.layout_len < 4 + size) return false;
            r.span.data = m.layout + 4;
            r.span.bytes = size;
            return true;
        }
        return false;
    }

    // object `index` of the global heap collection at `addr` (spec III.E: "GCOL", version 1, collection size; objects {index
    // u16, reference count u16, 4 reserved, size u64, data padded to 8}; index 0 is the free space).  A batch's strings are
    // consecutive objects, so the scan resumes behind the object found last.
    bool gcol_object(uint64_t addr, uint32_t index, const uint8_t*& data, uint64_t& len) const {
        if (!in(addr, 16) || std::memcmp(p_ + addr, "GCOL", 4) != 0 || p_[addr + 4] != 1 || index == 0) return false;
        const uint64_t size = rd64(p_ + addr + 8);
        if (size < 16 || !in(addr, size)) return false;
        auto scan = [&](uint64_t at) -> int {                               // 1 found, 0 not there, -1 a malformed collection
            while (at + 16 <= addr + size) {
                const uint32_t idx = rd16(p_ + at);
                const uint64_t osize = rd64(p_ + at + 8);
                if (idx == 0) return 0;                                      // the free space: nothing behind it
                if (osize > addr + size - (at + 16)) return -1;
                const uint64_t next = at + 16 + ((osize + 7) & ~7ull);
                if (idx == index) {
                    data = p_ + at + 16;
                    len = osize;
                    gcol_addr_ = addr;
                    gcol_next_index_ = index + 1;
                    gcol_at_ = next;
                    return 1;
                }
                at = next;
            }
            return 0;
        };
        const bool resume = gcol_addr_ == addr && gcol_next_index_ <= index && gcol_at_ >= addr + 16 && gcol_at_ <= addr + size;
        int rc = scan(resume ? gcol_at_ : addr + 16);
        if (rc == 0 && resume) rc = scan(addr + 16);
        return rc == 1;
    }

private:
    const uint8_t* p_ = nullptr;
    uint64_t n_ = 0, root_header_ = 0;
    int fd_ = -1;
    mutable uint64_t gcol_addr_ = 0, gcol_at_ = 0;
    mutable uint32_t gcol_next_index_ = 0;

    static uint16_t rd16(const uint8_t* q) { uint16_t v; std::memcpy(&v, q, 2); return v; }
    static uint32_t rd32(const uint8_t* q) { uint32_t v; std::memcpy(&v, q, 4); return v; }
    static uint64_t rd64(const uint8_t* q) { uint64_t v; std::memcpy(&v, q, 8); return v; }
    bool in(uint64_t off, uint64_t len) const { return off <= n_ && len <= n_ - off; }

    bool superblock() {
        static const uint8_t sig[8] = {0x89, 'H', 'D', 'F', '\r', '\n', 0x1a, '\n'};
        if (std::memcmp(p_, sig, 8) != 0) return false;                       // a user block moves it: not ours
        const uint8_t ver = p_[8];
        if (ver > 1 || p_[13] != 8 || p_[14] != 8) return false;            // 8-byte offsets and lengths
        uint64_t at = 24 + (ver == 1 ? 4 : 0);                               // v1 adds the indexed-storage K + 2 reserved bytes
        if (!in(at, 32 + 40)) return false;
        if (rd64(p_ + at) != 0) return false;                                 // base address
        at += 32;                                                             // base, free-space, end-of-file, driver info
        root_header_ = rd64(p_ + at + 8);                                     // root symbol table entry: name offset, header
        return root_header_ != 0 && in(root_header_, 16);
    }

    struct Msgs {
        const uint8_t *layout = nullptr, *dtype = nullptr, *dspace = nullptr, *symtab = nullptr;
        uint64_t layout_len = 0, dtype_len = 0, dspace_len = 0, symtab_len = 0;
        bool filters = false;
    };

    // version-1 object header: 16-byte prefix, then messages {type u16, size u16, flags u8, 3 reserved} in the first block
    // and in every continuation block (message 0x10 = {address, length})
    bool messages(uint64_t header, Msgs& m) const {
        if (!in(header, 16) || p_[header] != 1) return false;
        uint32_t left = rd16(p_ + header + 2);
        struct Block { uint64_t at, len; } blocks[16];
        int nb = 1, ib = 0;
        blocks[0] = {header + 16, rd32(p_ + header + 8)};
        while (left > 0 && ib < nb) {
            uint64_t at = blocks[ib].at;
            const uint64_t end = at + blocks[ib].len;
            ++ib;
            if (!in(at, end - at)) return false;
            while (left > 0 && at + 8 <= end) {
                const uint16_t type = rd16(p_ + at), size = rd16(p_ + at + 2);
                const uint8_t flags = p_[at + 4];
                const uint8_t* body = p_ + at + 8;
                if (at + 8 + size > end) return false;
                if (flags & 0x02) { if (type == 0x08 || type == 0x03 || type == 0x01) return false; }   // shared message
                else if (type == 0x08) { m.layout = body; m.layout_len = size; }
                else if (type == 0x03) { m.dtype = body; m.dtype_len = size; }
                else if (type == 0x01) { m.dspace = body; m.dspace_len = size; }
                else if (type == 0x11) { m.symtab = body; m.symtab_len = size; }
                else if (type == 0x0b) m.filters = true;
                else if (type == 0x10) {
                    if (size < 16 || nb == 16) return false;
                    blocks[nb++] = {rd64(body), rd64(body + 8)};
                }
                at += 8 + size;
                --left;
            }
        }
        return left == 0;
    }

    bool symbol_table(uint64_t header, uint64_t& btree, uint64_t& heap) const {
        Msgs m;
        if (!messages(header, m) || !m.symtab || m.symtab_len < 16) return false;
        btree = rd64(m.symtab);
        heap = rd64(m.symtab + 8);
        return true;
    }

    // name at `off` of a local heap's data segment; nullptr when out of bounds or unterminated
    const char* heap_name(uint64_t data, uint64_t size, uint64_t off) const {
        if (off >= size) return nullptr;
        const void* z = std::memchr(p_ + data + off, 0, (size_t)(size - off));
        return z ? (const char*)(p_ + data + off) : nullptr;
    }

    bool walk(uint64_t node, uint64_t hdata, uint64_t hsize, int depth, std::vector<std::pair<std::string, uint64_t>>& out) const {
        if (depth > 16 || !in(node, 24) || std::memcmp(p_ + node, "TREE", 4) != 0 || p_[node + 4] != 0) return false;
        const uint8_t level = p_[node + 5];
        const uint32_t used = rd16(p_ + node + 6);
        if (!in(node + 24, 16ull * used + 8)) return false;
        for (uint32_t c = 0; c < used; ++c) {
            const uint64_t kid = rd64(p_ + node + 24 + 16ull * c + 8);
            if (level > 0) {
                if (!walk(kid, hdata, hsize, depth + 1, out)) return false;
                continue;
            }
            if (!in(kid, 8) || std::memcmp(p_ + kid, "SNOD", 4) != 0 || p_[kid + 4] != 1) return false;
            const uint32_t count = rd16(p_ + kid + 6);
            if (!in(kid + 8, 40ull * count)) return false;
            for (uint32_t e = 0; e < count; ++e) {
                const uint8_t* entry = p_ + kid + 8 + 40ull * e;
                const char* name = heap_name(hdata, hsize, rd64(entry));
                const uint64_t header = rd64(entry + 8);
                if (!name || !in(header, 16) || out.size() >= (1u << 26)) return false;     // (a tree that loops: don't know)
                out.emplace_back(name, header);
            }
        }
        return true;
    }

    uint64_t lookup(uint64_t btree, uint64_t heap, const char* name) const {
        if (!in(heap, 32) || std::memcmp(p_ + heap, "HEAP", 4) != 0 || p_[heap + 4] != 0) return 0;
        const uint64_t hsize = rd64(p_ + heap + 8), hdata = rd64(p_ + heap + 24);
        if (!in(hdata, hsize)) return 0;
        uint64_t node = btree;
        for (int depth = 0; depth < 16; ++depth) {
            if (!in(node, 24) || std::memcmp(p_ + node, "TREE", 4) != 0 || p_[node + 4] != 0) return 0;
            const uint8_t level = p_[node + 5];
            const uint32_t used = rd16(p_ + node + 6);
            if (used == 0 || !in(node + 24, 16ull * used + 8)) return 0;
            const uint8_t* kc = p_ + node + 24;                               // key0 child0 key1 child1 ... key_used
            // child i holds the names in (key[i], key[i+1]]: the first i with name <= key[i+1]
            uint32_t lo = 0, hi = used;                                       // answer in [lo, hi)
            while (lo + 1 < hi) {
                const uint32_t mid = (lo + hi) / 2;                           // is name <= key[mid]?  then the answer is < mid
                const char* k = heap_name(hdata, hsize, rd64(kc + 16ull * mid));
                if (!k) return 0;
                if (std::strcmp(name, k) <= 0) hi = mid; else lo = mid;
            }
            const uint64_t next = rd64(kc + 16ull * lo + 8);
            if (level > 0) { node = next; continue; }
            // symbol node: "SNOD", version 1, count, entries of 40 bytes sorted by name
            if (!in(next, 8) || std::memcmp(p_ + next, "SNOD", 4) != 0 || p_[next + 4] != 1) return 0;
            const uint32_t count = rd16(p_ + next + 6);
            if (!in(next + 8, 40ull * count)) return 0;
            uint32_t a = 0, b = count;
            while (a < b) {
                const uint32_t mid = (a + b) / 2;
                const uint8_t* e = p_ + next + 8 + 40ull * mid;
                const char* k = heap_name(hdata, hsize, rd64(e));
                if (!k) return 0;
                const int c = std::strcmp(name, k);
                if (c == 0) { const uint64_t h = rd64(e + 8); return in(h, 16) ? h : 0; }
                if (c < 0) b = mid; else a = mid + 1;
            }
            return 0;
        }
        return 0;
    }
};

}  // namespace

struct pa_h5 {
    hid_t file = -1;
    hid_t lcpl = -1;  // create intermediate groups, as h5py's file[path] = data does
    std::string path;
    int32_t mode = 0;
    Direct* direct = nullptr;
    bool direct_tried = false;
    int64_t direct_chunks = 0, library_chunks = 0;     // polish chunks read either way (pa_h5_read_stats)
    ~pa_h5() { delete direct; }
};

void pa_h5_set_error(const std::string& msg) { g_err = msg; }     // h5build.cpp reports through the same thread-local text

extern "C" {

const char* pa_h5_last_error(void) { return g_err.c_str(); }

int pa_h5_open(const char* path, int32_t mode, pa_h5** out) {
    if (!path || !out) return fail("null argument");
    Quiet q;
    hid_t f = -1;
    // File access tuned for files made of hundreds of thousands of small objects (one group of 4-8 datasets per polish chunk /
    // per 512-window batch): metadata and small raw data are allocated in 1 MB blocks instead of 2 KB ones, the metadata cache
    // starts at 64 MB instead of 2 MB (the default cache evicts and re-reads symbol-table nodes all through such a file), and
    // files created with mode 3 (the prediction stores) use the 1.10 object formats (links of a small group live in its header: no
    // B-tree + heap per group; 17 % less time per chunk written, three times the time to open a group when read back -- so the
    // image stores, which are read group by group, keep the classic format).  Names, shapes and dtypes -- what the reference's
    // readers see -- do not change.  PEPPER_AMD_H5_PLAIN=1: library defaults.
    static const bool plain = getenv("PEPPER_AMD_H5_PLAIN") != nullptr;
    hid_t fapl = H5Pcreate(H5P_FILE_ACCESS);
    if (!plain) {
        H5Pset_meta_block_size(fapl, 1 << 20);
        H5Pset_small_data_block_size(fapl, 1 << 20);
        H5Pset_sieve_buf_size(fapl, 1 << 20);
        H5AC_cache_config_t mdc;
        mdc.version = H5AC__CURR_CACHE_CONFIG_VERSION;
        if (H5Pget_mdc_config(fapl, &mdc) >= 0) {
            mdc.set_initial_size = 1;
            mdc.initial_size = 64 << 20;
            mdc.min_size = 32 << 20;
            mdc.max_size = 512 << 20;
            mdc.decr_mode = H5C_decr__off;
            (void)H5Pset_mdc_config(fapl, &mdc);
        }
        if (mode == 3) (void)H5Pset_libver_bounds(fapl, H5F_LIBVER_V110, H5F_LIBVER_LATEST);
    }
    if (mode == 0) f = H5Fopen(path, H5F_ACC_RDONLY, fapl);
    else if (mode == 1 || mode == 3) f = H5Fcreate(path, H5F_ACC_TRUNC, H5P_DEFAULT, fapl);
    else if (mode == 2) f = H5Fopen(path, H5F_ACC_RDWR, fapl);
    else {
        H5Pclose(fapl);
        return fail("bad mode");
    }
    H5Pclose(fapl);
    if (f < 0) return fail(std::string("cannot open HDF5 file '") + path + "'");
    auto* h = new pa_h5();
    h->file = f;
    h->path = path;
    h->mode = mode;
    h->lcpl = H5Pcreate(H5P_LINK_CREATE);
    H5Pset_create_intermediate_group(h->lcpl, 1);
    *out = h;
    return 0;
}

int pa_h5_close(pa_h5* f) {
    if (!f) return 0;
    if (f->lcpl >= 0) H5Pclose(f->lcpl);
    int rc = 0;
    if (f->file >= 0 && H5Fclose(f->file) < 0) rc = fail("H5Fclose failed");
    delete f;
    return rc;
}

int pa_h5_flush(pa_h5* f) {
    if (!f) return fail("null file");
    return H5Fflush(f->file, H5F_SCOPE_GLOBAL) < 0 ? fail("H5Fflush failed") : 0;
}

int pa_h5_exists(pa_h5* f, const char* path) {
    if (!f || !path) return fail("null argument");
    Quiet q;
    // H5Lexists needs every intermediate link to exist: walk the components
    std::string p(path), cur;
    size_t i = 0;
    while (i < p.size()) {
        size_t j = p.find('/', i);
        if (j == std::string::npos) j = p.size();
        if (j > i) {
            cur += (cur.empty() ? "" : "/") + p.substr(i, j - i);
            htri_t e = H5Lexists(f->file, cur.c_str(), H5P_DEFAULT);
            if (e < 0) return fail("H5Lexists failed");
            if (e == 0) return 0;
        }
        i = j + 1;
    }
    return 1;
}

static herr_t list_cb(hid_t, const char* name, const H5L_info_t*, void* op) {
    static_cast<std::vector<std::string>*>(op)->push_back(name);
    return 0;
}

int pa_h5_list(pa_h5* f, const char* group, char* buf, int64_t cap, int64_t* needed, int64_t* count) {
    if (!f || !group || !needed) return fail("null argument");
    Quiet q;
    hid_t g = H5Gopen2(f->file, group, H5P_DEFAULT);
    if (g < 0) return fail(std::string("no such group '") + group + "'");
    std::vector<std::string> names;
    hsize_t idx = 0;
    herr_t rc = H5Literate(g, H5_INDEX_NAME, H5_ITER_INC, &idx, list_cb, &names);
    H5Gclose(g);
    if (rc < 0) return fail("H5Literate failed");
    int64_t total = 0;
    for (auto& n : names) total += (int64_t)n.size() + 1;
    *needed = total;
    if (count) *count = (int64_t)names.size();
    if (buf && cap >= total) {
        char* p = buf;
        for (auto& n : names) {
            std::memcpy(p, n.c_str(), n.size() + 1);
            p += n.size() + 1;
        }
    }
    return 0;
}

int pa_h5_info(pa_h5* f, const char* path, int32_t* rank, int64_t* dims, int32_t* cls, int32_t* elem_size,
               int32_t* is_signed) {
    if (!f || !path) return fail("null argument");
    Quiet q;
    hid_t d = H5Dopen2(f->file, path, H5P_DEFAULT);
    if (d < 0) return fail(std::string("no such dataset '") + path + "'");
    hid_t sp = H5Dget_space(d), ty = H5Dget_type(d);
    const int r = H5Sget_simple_extent_ndims(sp);
    int rc = 0;
    if (r < 0 || r > 8) rc = fail("unsupported rank");
    else {
        hsize_t hd[8] = {0};
        if (r > 0) H5Sget_simple_extent_dims(sp, hd, nullptr);
        if (rank) *rank = r;
        if (dims) for (int i = 0; i < r; ++i) dims[i] = (int64_t)hd[i];
        const H5T_class_t c = H5Tget_class(ty);
        int32_t k = PA_H5_CLASS_OTHER, sgn = 0;
        if (c == H5T_INTEGER) { k = PA_H5_CLASS_INT; sgn = H5Tget_sign(ty) == H5T_SGN_2; }
        else if (c == H5T_FLOAT) { k = PA_H5_CLASS_FLOAT; sgn = 1; }
        else if (c == H5T_STRING) k = H5Tis_variable_str(ty) > 0 ? PA_H5_CLASS_VLEN_STRING : PA_H5_CLASS_FIXED_STRING;
        if (cls) *cls = k;
        if (is_signed) *is_signed = sgn;
        if (elem_size) *elem_size = k == PA_H5_CLASS_VLEN_STRING ? 0 : (int32_t)H5Tget_size(ty);
    }
    H5Tclose(ty);
    H5Sclose(sp);
    H5Dclose(d);
    return rc;
}

int pa_h5_read(pa_h5* f, const char* path, int32_t type_code, void* out, int64_t nbytes) {
    if (!f || !path || (!out && nbytes > 0)) return fail("null argument");
    const hid_t mt = native_type(type_code);
    if (mt < 0) return fail("bad type code");
    Quiet q;
    hid_t d = H5Dopen2(f->file, path, H5P_DEFAULT);
    if (d < 0) return fail(std::string("no such dataset '") + path + "'");
    hid_t sp = H5Dget_space(d);
    const hssize_t n = H5Sget_simple_extent_npoints(sp);
    int rc = 0;
    if (n < 0 || (int64_t)n * type_bytes(type_code) != nbytes)
        rc = fail(std::string("size mismatch reading '") + path + "'");
    else if (n > 0 && H5Dread(d, mt, H5S_ALL, H5S_ALL, H5P_DEFAULT, out) < 0)
        rc = fail(std::string("H5Dread failed for '") + path + "'");
    H5Sclose(sp);
    H5Dclose(d);
    return rc;
}

static int write_dataset(pa_h5* f, const char* path, hid_t file_type, hid_t mem_type, int32_t rank,
                         const int64_t* dims, const void* data) {
    if (rank < 0 || rank > 8) return fail("unsupported rank");
    Quiet q;
    hid_t sp = make_space(rank, dims);
    if (sp < 0) return fail("cannot create dataspace");
    hid_t d = H5Dcreate2(f->file, path, file_type, sp, f->lcpl, H5P_DEFAULT, H5P_DEFAULT);
    int rc = 0;
    if (d < 0) rc = fail(std::string("cannot create dataset '") + path + "' (already exists?)");
    else {
        const hssize_t n = H5Sget_simple_extent_npoints(sp);
        if (n > 0 && data && H5Dwrite(d, mem_type, H5S_ALL, H5S_ALL, H5P_DEFAULT, data) < 0)
            rc = fail(std::string("H5Dwrite failed for '") + path + "'");
        H5Dclose(d);
    }
    H5Sclose(sp);
    return rc;
}

int pa_h5_write(pa_h5* f, const char* path, int32_t type_code, int32_t rank, const int64_t* dims,
                const void* data) {
    if (!f || !path) return fail("null argument");
    const hid_t t = native_type(type_code);
    if (t < 0) return fail("bad type code");
    return write_dataset(f, path, t, t, rank, dims, data);
}

int pa_h5_read_strings(pa_h5* f, const char* path, char* buf, int64_t cap, int64_t* needed) {
    if (!f || !path || !needed) return fail("null argument");
    Quiet q;
    hid_t d = H5Dopen2(f->file, path, H5P_DEFAULT);
    if (d < 0) return fail(std::string("no such dataset '") + path + "'");
    hid_t sp = H5Dget_space(d), ty = H5Dget_type(d);
    const hssize_t n = H5Sget_simple_extent_npoints(sp);
    int rc = 0;
    std::string out;
    if (H5Tget_class(ty) != H5T_STRING || n < 0) rc = fail(std::string("'") + path + "' is not a string dataset");
    else if (H5Tis_variable_str(ty) > 0) {
        std::vector<char*> ptrs((size_t)n, nullptr);
        hid_t mt = H5Tcopy(H5T_C_S1);
        H5Tset_size(mt, H5T_VARIABLE);
        H5Tset_cset(mt, H5Tget_cset(ty));
        if (n > 0 && H5Dread(d, mt, H5S_ALL, H5S_ALL, H5P_DEFAULT, ptrs.data()) < 0) rc = fail("H5Dread (vlen) failed");
        else {
            for (hssize_t i = 0; i < n; ++i) {
                if (ptrs[i]) out.append(ptrs[i]);
                out.push_back('\0');
            }
            if (n > 0) H5Dvlen_reclaim(mt, sp, H5P_DEFAULT, ptrs.data());
        }
        H5Tclose(mt);
    } else {
        const size_t w = H5Tget_size(ty);
        std::vector<char> raw((size_t)n * w + 1, 0);
        if (n > 0 && H5Dread(d, ty, H5S_ALL, H5S_ALL, H5P_DEFAULT, raw.data()) < 0) rc = fail("H5Dread (fixed) failed");
        else
            for (hssize_t i = 0; i < n; ++i) {
                const char* s = raw.data() + (size_t)i * w;
                out.append(s, strnlen(s, w));
                out.push_back('\0');
            }
    }
    if (rc == 0) {
        *needed = (int64_t)out.size();
        if (buf && cap >= (int64_t)out.size()) std::memcpy(buf, out.data(), out.size());
    }
    H5Tclose(ty);
    H5Sclose(sp);
    H5Dclose(d);
    return rc;
}

int pa_h5_write_fixed_strings(pa_h5* f, const char* path, int32_t rank, const int64_t* dims, int32_t width,
                              const char* data) {
    if (!f || !path || width <= 0) return fail("bad argument");
    hid_t t = H5Tcopy(H5T_C_S1);
    H5Tset_size(t, (size_t)width);
    H5Tset_strpad(t, H5T_STR_NULLPAD);   // numpy 'S' -> h5py: fixed width, null padded
    const int rc = write_dataset(f, path, t, t, rank, dims, data);
    H5Tclose(t);
    return rc;
}

int pa_h5_write_vlen_strings(pa_h5* f, const char* path, int32_t rank, const int64_t* dims,
                             const char* const* strings) {
    if (!f || !path) return fail("bad argument");
    hid_t t = H5Tcopy(H5T_C_S1);
    H5Tset_size(t, H5T_VARIABLE);
    H5Tset_cset(t, H5T_CSET_UTF8);       // h5py special_dtype(vlen=str)
    const int rc = write_dataset(f, path, t, t, rank, dims, strings);
    H5Tclose(t);
    return rc;
}

int pa_h5_write_prediction_batch(pa_h5* f, const char* group, int32_t n, const char* contigs, int32_t contig_stride,
                                 const int32_t* positions, const uint8_t* depths, const char* cand_blob,
                                 const int64_t* cand_offsets, const uint8_t* freqs, const float* probs, int32_t n_classes) {
    if (!f || !group || n < 0 || n_classes <= 0 || contig_stride <= 0 ||
        (n > 0 && (!contigs || !positions || !depths || !cand_blob || !cand_offsets || !freqs || !probs)))
        return fail("bad argument");
    Quiet q;
    hid_t g = H5Gcreate2(f->file, group, f->lcpl, H5P_DEFAULT, H5P_DEFAULT);
    if (g < 0) return fail(std::string("cannot create group '") + group + "' (already exists?)");
    int rc = 0;
    auto put = [&](const char* name, hid_t ft, hid_t mt, int rank, hsize_t d0, hsize_t d1, const void* data) {
        if (rc) return;
        hsize_t dims[2] = {d0, d1};
        hid_t sp = H5Screate_simple(rank, dims, nullptr);
        hid_t d = H5Dcreate2(g, name, ft, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
        if (d < 0) rc = fail(std::string("cannot create dataset '") + group + "/" + name + "'");
        else {
            if (n > 0 && H5Dwrite(d, mt, H5S_ALL, H5S_ALL, H5P_DEFAULT, data) < 0)
                rc = fail(std::string("H5Dwrite failed for '") + group + "/" + name + "'");
            H5Dclose(d);
        }
        H5Sclose(sp);
    };
    // contigs: fixed-width, null padded, as wide as the longest name of the batch (numpy 'S' array semantics)
    size_t w = 1;
    for (int32_t i = 0; i < n; ++i) w = std::max(w, strnlen(contigs + (size_t)i * contig_stride, (size_t)contig_stride));
    std::vector<char> packed((size_t)n * w + 1, 0);
    for (int32_t i = 0; i < n; ++i) {
        const char* src = contigs + (size_t)i * contig_stride;
        std::memcpy(packed.data() + (size_t)i * w, src, strnlen(src, std::min(w, (size_t)contig_stride)));
    }
    hid_t ts = H5Tcopy(H5T_C_S1);
    H5Tset_size(ts, w);
    H5Tset_strpad(ts, H5T_STR_NULLPAD);
    put("contigs", ts, ts, 1, (hsize_t)n, 0, packed.data());
    H5Tclose(ts);
    put("positions", H5T_STD_I32LE, H5T_NATIVE_INT32, 1, (hsize_t)n, 0, positions);
    put("depths", H5T_STD_U8LE, H5T_NATIVE_UINT8, 1, (hsize_t)n, 0, depths);
    std::vector<const char*> ptrs((size_t)std::max(n, 1), "");
    for (int32_t i = 0; i < n; ++i) ptrs[(size_t)i] = cand_blob + cand_offsets[i];
    hid_t tv = H5Tcopy(H5T_C_S1);
    H5Tset_size(tv, H5T_VARIABLE);
    H5Tset_cset(tv, H5T_CSET_UTF8);
    put("candidates", tv, tv, 2, (hsize_t)n, 1, ptrs.data());
    H5Tclose(tv);
    put("candidate_frequency", H5T_STD_U8LE, H5T_NATIVE_UINT8, 2, (hsize_t)n, 1, freqs);
    std::vector<double> p64((size_t)n * n_classes + 1);
    for (size_t i = 0; i < (size_t)n * n_classes; ++i) p64[i] = (double)probs[i];
    put("base_prediction", H5T_IEEE_F64LE, H5T_NATIVE_DOUBLE, 2, (hsize_t)n, (hsize_t)n_classes, p64.data());
    H5Gclose(g);
    return rc;
}

// the direct locator of a file opened read-only (made on first use); nullptr = through the library (PEPPER_AMD_H5_DIRECT=0, a
// file open for writing, or a file the locator could not map)
static const Direct* direct_of(pa_h5* f) {
    static const bool no_direct = getenv("PEPPER_AMD_H5_DIRECT") && getenv("PEPPER_AMD_H5_DIRECT")[0] == '0';
    if (no_direct) return nullptr;
    if (!f->direct_tried) {
        f->direct_tried = true;
        if (f->mode == 0) f->direct = Direct::open(f->path.c_str());
    }
    return f->direct;
}

// ---- polish stores in bulk: the format keeps one group per 1000-row chunk (8 small datasets in the image file, 4 in the
// prediction file), so a region-sized device pass touches thousands of datasets; these entry points do the whole block
// inside the library instead of one Python-level call per dataset.

static int read_numeric(hid_t loc, const char* name, hid_t mem_type, int64_t expect, void* out, const std::string& where) {
    hid_t d = H5Dopen2(loc, name, H5P_DEFAULT);
    if (d < 0) return fail("no dataset '" + where + name + "'");
    hid_t sp = H5Dget_space(d);
    const hssize_t n = H5Sget_simple_extent_npoints(sp);
    int rc = 0;
    if (n != expect) rc = fail("'" + where + name + "' has " + std::to_string((long long)n) + " elements, expected " + std::to_string((long long)expect));
    else if (H5Dread(d, mem_type, H5S_ALL, H5S_ALL, H5P_DEFAULT, out) < 0) rc = fail("H5Dread failed for '" + where + name + "'");
    H5Sclose(sp);
    H5Dclose(d);
    return rc;
}

static int read_string_scalar(hid_t loc, const char* name, char* out, int32_t cap, const std::string& where) {
    hid_t d = H5Dopen2(loc, name, H5P_DEFAULT);
    if (d < 0) return fail("no dataset '" + where + name + "'");
    hid_t ty = H5Dget_type(d), sp = H5Dget_space(d);
    int rc = 0;
    std::memset(out, 0, (size_t)cap);
    if (H5Tget_class(ty) != H5T_STRING || H5Sget_simple_extent_npoints(sp) != 1) rc = fail("'" + where + name + "' is not a string scalar");
    else if (H5Tis_variable_str(ty) > 0) {
        char* ptr = nullptr;
        hid_t mt = H5Tcopy(H5T_C_S1);
        H5Tset_size(mt, H5T_VARIABLE);
        H5Tset_cset(mt, H5Tget_cset(ty));
        if (H5Dread(d, mt, H5S_ALL, H5S_ALL, H5P_DEFAULT, &ptr) < 0) rc = fail("H5Dread failed for '" + where + name + "'");
        else {
            if (ptr) {
                if ((int32_t)strlen(ptr) >= cap) rc = fail("contig name longer than the buffer in '" + where + name + "'");
                else std::strcpy(out, ptr);
            }
            H5Dvlen_reclaim(mt, sp, H5P_DEFAULT, &ptr);
        }
        H5Tclose(mt);
    } else {
        const size_t w = H5Tget_size(ty);
        std::vector<char> raw(w + 1, 0);
        if (H5Dread(d, ty, H5S_ALL, H5S_ALL, H5P_DEFAULT, raw.data()) < 0) rc = fail("H5Dread failed for '" + where + name + "'");
        else if ((int32_t)strnlen(raw.data(), w) >= cap) rc = fail("contig name longer than the buffer in '" + where + name + "'");
        else std::memcpy(out, raw.data(), strnlen(raw.data(), w));
    }
    H5Sclose(sp);
    H5Tclose(ty);
    H5Dclose(d);
    return rc;
}

// <contig>_<region_start>_<region_end>_<chunk_id> (pepper/.../ImageGenerationUI.py:220): the four small datasets of a chunk
// group restate its name.  -> false when the name is not of that form
static bool parse_chunk_name(const char* name, std::string& contig, int64_t& start, int64_t& end, int64_t& chunk) {
    const std::string s(name);
    size_t p3 = s.rfind('_');
    if (p3 == std::string::npos || p3 == 0) return false;
    size_t p2 = s.rfind('_', p3 - 1);
    if (p2 == std::string::npos || p2 == 0) return false;
    size_t p1 = s.rfind('_', p2 - 1);
    if (p1 == std::string::npos || p1 == 0) return false;
    auto num = [&](size_t a, size_t b, int64_t& out) {
        if (b <= a) return false;
        char* endp = nullptr;
        const std::string t = s.substr(a, b - a);
        for (size_t k = (t[0] == '-' ? 1 : 0); k < t.size(); ++k)
            if (t[k] < '0' || t[k] > '9') return false;
        out = strtoll(t.c_str(), &endp, 10);
        return endp && *endp == 0;
    };
    contig = s.substr(0, p1);
    return num(p1 + 1, p2, start) && num(p2 + 1, p3, end) && num(p3 + 1, s.size(), chunk);
}

static int read_polish_chunks_impl(pa_h5* f, const char* names, int32_t n, int32_t seq_len, int32_t features, uint8_t* images,
                                   int64_t* position, int64_t* index, int64_t* region_start, int64_t* region_end,
                                   int64_t* chunk_id, char* contigs, int32_t contig_stride, bool from_names) {
    if (!f || n < 0 || seq_len <= 0 || features <= 0 || contig_stride <= 1 ||
        (n > 0 && (!names || !images || !position || !index || !region_start || !region_end || !chunk_id || !contigs)))
        return fail("bad argument");
    Quiet q;
    hid_t root = H5Gopen2(f->file, "summaries", H5P_DEFAULT);
    if (root < 0) return fail("no group 'summaries'");
    // the three large datasets of a chunk straight from the mapped file where the direct locator understands it (a file opened
    // read-only; PEPPER_AMD_H5_DIRECT=0: always through the library); per chunk, so one odd group costs one library read
    const Direct* dd = direct_of(f);
    const uint64_t dsum = dd ? dd->child(dd->root(), "summaries") : 0;
    const char* name = names;
    int rc = 0;
    for (int32_t i = 0; i < n && !rc; ++i, name += strlen(name) + 1) {
        const std::string where = std::string("summaries/") + name + "/";
        bool have = false;
        if (dsum) {
            const uint64_t g = dd->child(dsum, name);
            Direct::Span a, b, c;
            if (g && dd->dataset(dd->child(g, "image"), 1, (uint64_t)seq_len * features, a) &&
                dd->dataset(dd->child(g, "position"), 8, (uint64_t)seq_len, b) &&
                dd->dataset(dd->child(g, "index"), 8, (uint64_t)seq_len, c) &&
                dd->copy(a, images + (size_t)i * seq_len * features) && dd->copy(b, position + (size_t)i * seq_len) &&
                dd->copy(c, index + (size_t)i * seq_len)) {
                have = true;
                ++f->direct_chunks;
            }
        }
        std::string contig;
        int64_t s0 = 0, e0 = 0, c0 = 0;
        const bool named = from_names && parse_chunk_name(name, contig, s0, e0, c0) && (int)contig.size() < contig_stride;
        if (have && named && i != 0 && i != n - 1) {            // nothing of this chunk needs the library
            region_start[i] = s0;
            region_end[i] = e0;
            chunk_id[i] = c0;
            std::memset(contigs + (size_t)i * contig_stride, 0, (size_t)contig_stride);
            std::memcpy(contigs + (size_t)i * contig_stride, contig.data(), contig.size());
            continue;
        }
        hid_t g = H5Gopen2(root, name, H5P_DEFAULT);
        if (g < 0) { rc = fail("no group '" + where + "'"); break; }
        if (!have) {
            ++f->library_chunks;
            rc = read_numeric(g, "image", H5T_NATIVE_UINT8, (int64_t)seq_len * features, images + (size_t)i * seq_len * features, where);
            if (!rc) rc = read_numeric(g, "position", H5T_NATIVE_INT64, seq_len, position + (size_t)i * seq_len, where);
            if (!rc) rc = read_numeric(g, "index", H5T_NATIVE_INT64, seq_len, index + (size_t)i * seq_len, where);
        }
        // the small datasets: read for the first and the last chunk of the call (and checked against the name), taken from
        // the name for the others -- four of the seven objects of a chunk, at ~40 us of library time each
        if (!rc && (!named || i == 0 || i == n - 1)) {
            rc = read_numeric(g, "region_start", H5T_NATIVE_INT64, 1, region_start + i, where);
            if (!rc) rc = read_numeric(g, "region_end", H5T_NATIVE_INT64, 1, region_end + i, where);
            if (!rc) rc = read_numeric(g, "chunk_id", H5T_NATIVE_INT64, 1, chunk_id + i, where);
            if (!rc) rc = read_string_scalar(g, "contig", contigs + (size_t)i * contig_stride, contig_stride, where);
            if (!rc && named && (region_start[i] != s0 || region_end[i] != e0 || chunk_id[i] != c0 ||
                                 contig != std::string(contigs + (size_t)i * contig_stride)))
                rc = 2;                 // names do not restate the datasets in this file: the caller reads them all
        } else if (!rc) {
            region_start[i] = s0;
            region_end[i] = e0;
            chunk_id[i] = c0;
            std::memset(contigs + (size_t)i * contig_stride, 0, (size_t)contig_stride);
            std::memcpy(contigs + (size_t)i * contig_stride, contig.data(), contig.size());
        }
        H5Gclose(g);
    }
    H5Gclose(root);
    return rc;
}

int pa_h5_read_polish_chunks(pa_h5* f, const char* names, int32_t n, int32_t seq_len, int32_t features, uint8_t* images,
                             int64_t* position, int64_t* index, int64_t* region_start, int64_t* region_end,
                             int64_t* chunk_id, char* contigs, int32_t contig_stride) {
    // PEPPER_AMD_H5_ALL_DATASETS=1: every dataset of every chunk, as round 2 read them
    static const bool all = getenv("PEPPER_AMD_H5_ALL_DATASETS") != nullptr;
    int rc = read_polish_chunks_impl(f, names, n, seq_len, features, images, position, index, region_start, region_end, chunk_id,
                                     contigs, contig_stride, !all);
    if (rc == 2)
        rc = read_polish_chunks_impl(f, names, n, seq_len, features, images, position, index, region_start, region_end, chunk_id,
                                     contigs, contig_stride, false);
    return rc;
}

int pa_h5_read_stats(pa_h5* f, int64_t* direct_chunks, int64_t* library_chunks) {
    if (!f || !direct_chunks || !library_chunks) return fail("null argument");
    *direct_chunks = f->direct_chunks;
    *library_chunks = f->library_chunks;
    return 0;
}

int pa_h5_write_polish_predictions(pa_h5* f, int32_t n, int32_t seq_len, const char* contigs, int32_t contig_stride,
                                   const int64_t* contig_start, const int64_t* contig_end, const int64_t* chunk_id,
                                   const uint8_t* new_region, const uint8_t* skip, const int64_t* position,
                                   const int64_t* index, const uint8_t* bases, const uint8_t* phred) {
    if (!f || n < 0 || seq_len <= 0 || contig_stride <= 0 ||
        (n > 0 && (!contigs || !contig_start || !contig_end || !chunk_id || !new_region || !skip || !position || !index ||
                   !bases || !phred)))
        return fail("bad argument");
    Quiet q;
    hsize_t dims[1] = {(hsize_t)seq_len};
    hid_t sp_row = H5Screate_simple(1, dims, nullptr), sp_one = H5Screate(H5S_SCALAR);
    // rows of 1 - 8 KB: stored in the dataset's object header (compact layout) -- one metadata write instead of a header plus a
    // separately allocated raw block; readers see the same names, shapes and dtypes.  PEPPER_AMD_H5_PLAIN=1: contiguous
    static const bool plain = getenv("PEPPER_AMD_H5_PLAIN") != nullptr;
    hid_t dcpl = H5Pcreate(H5P_DATASET_CREATE);
    if (!plain && (size_t)seq_len * 8 < 60000) H5Pset_layout(dcpl, H5D_COMPACT);
    int rc = 0;
    auto put = [&](hid_t loc, const char* name, hid_t ft, hid_t mt, hid_t sp, const void* data, const std::string& where) {
        if (rc) return;
        hid_t d = H5Dcreate2(loc, name, ft, sp, f->lcpl, dcpl, H5P_DEFAULT);
        if (d < 0) { rc = fail("cannot create dataset '" + where + "/" + name + "' (already exists?)"); return; }
        if (H5Dwrite(d, mt, H5S_ALL, H5S_ALL, H5P_DEFAULT, data) < 0) rc = fail("H5Dwrite failed for '" + where + "/" + name + "'");
        H5Dclose(d);
    };
    for (int32_t i = 0; i < n && !rc; ++i) {
        const char* c = contigs + (size_t)i * contig_stride;
        const std::string contig(c, strnlen(c, (size_t)contig_stride));
        const std::string region = "predictions/" + contig + "/" + contig + "-" + std::to_string((long long)contig_start[i]) + "-" +
                                   std::to_string((long long)contig_end[i]);
        if (new_region[i]) {
            hid_t g = H5Gcreate2(f->file, region.c_str(), f->lcpl, H5P_DEFAULT, H5P_DEFAULT);
            if (g < 0) { rc = fail("cannot create group '" + region + "'"); break; }
            put(g, "contig_start", H5T_STD_I64LE, H5T_NATIVE_INT64, sp_one, contig_start + i, region);
            put(g, "contig_end", H5T_STD_I64LE, H5T_NATIVE_INT64, sp_one, contig_end + i, region);
            H5Gclose(g);
        }
        if (skip[i] || rc) continue;
        const std::string chunk = region + "/" + std::to_string((long long)chunk_id[i]);
        hid_t g = H5Gcreate2(f->file, chunk.c_str(), f->lcpl, H5P_DEFAULT, H5P_DEFAULT);
        if (g < 0) { rc = fail("cannot create group '" + chunk + "'"); break; }
        put(g, "position", H5T_STD_I64LE, H5T_NATIVE_INT64, sp_row, position + (size_t)i * seq_len, chunk);
        put(g, "index", H5T_STD_I64LE, H5T_NATIVE_INT64, sp_row, index + (size_t)i * seq_len, chunk);
        put(g, "bases", H5T_STD_U8LE, H5T_NATIVE_UINT8, sp_row, bases + (size_t)i * seq_len, chunk);
        put(g, "phred_score", H5T_STD_U8LE, H5T_NATIVE_UINT8, sp_row, phred + (size_t)i * seq_len, chunk);
        H5Gclose(g);
    }
    H5Pclose(dcpl);
    H5Sclose(sp_row);
    H5Sclose(sp_one);
    return rc;
}

static herr_t collect_names(hid_t, const char* name, const H5L_info_t*, void* p) {
    static_cast<std::vector<std::string>*>(p)->push_back(name);
    return 0;
}

int pa_h5_read_polish_prediction_region(pa_h5* f, const char* region_path, int32_t seq_len, int32_t max_chunks,
                                        int64_t* position, int64_t* index, uint8_t* bases, int32_t* n_chunks) {
    if (!f || !region_path || seq_len <= 0 || max_chunks <= 0 || !position || !index || !bases || !n_chunks)
        return fail("bad argument");
    Quiet q;
    hid_t g = H5Gopen2(f->file, region_path, H5P_DEFAULT);
    if (g < 0) return fail(std::string("no group '") + region_path + "'");
    std::vector<std::string> names;
    H5Literate(g, H5_INDEX_NAME, H5_ITER_NATIVE, nullptr, collect_names, &names);
    names.erase(std::remove_if(names.begin(), names.end(),
                               [](const std::string& s) { return s == "contig_start" || s == "contig_end"; }), names.end());
    std::sort(names.begin(), names.end());            // the order of Python's sorted() on the chunk ids as strings
    *n_chunks = (int32_t)names.size();
    int rc = 0;
    if ((int32_t)names.size() > max_chunks) rc = fail(std::string("more chunks than the buffer holds in '") + region_path + "'");
    for (size_t i = 0; i < names.size() && !rc; ++i) {
        hid_t c = H5Gopen2(g, names[i].c_str(), H5P_DEFAULT);
        const std::string where = std::string(region_path) + "/" + names[i] + "/";
        if (c < 0) { rc = fail("no group '" + where + "'"); break; }
        rc = read_numeric(c, "position", H5T_NATIVE_INT64, seq_len, position + i * (size_t)seq_len, where);
        if (!rc) rc = read_numeric(c, "index", H5T_NATIVE_INT64, seq_len, index + i * (size_t)seq_len, where);
        if (!rc) rc = read_numeric(c, "bases", H5T_NATIVE_UINT8, seq_len, bases + i * (size_t)seq_len, where);
        H5Gclose(c);
    }
    H5Gclose(g);
    return rc;
}

int pa_h5_write_polish_image_chunks(pa_h5* f, const char* names, int32_t n, int32_t seq_len, int32_t features,
                                    const char* contig, int64_t region_start, int64_t region_end, const int64_t* chunk_id,
                                    const uint8_t* images, const uint8_t* labels, const int64_t* position,
                                    const int64_t* index) {
    if (!f || n < 0 || seq_len <= 0 || features <= 0 || !contig ||
        (n > 0 && (!names || !chunk_id || !images || !labels || !position || !index)))
        return fail("bad argument");
    Quiet q;
    hsize_t d2[2] = {(hsize_t)seq_len, (hsize_t)features}, d1[1] = {(hsize_t)seq_len};
    hid_t sp_img = H5Screate_simple(2, d2, nullptr), sp_row = H5Screate_simple(1, d1, nullptr), sp_one = H5Screate(H5S_SCALAR);
    hid_t tv = H5Tcopy(H5T_C_S1);
    H5Tset_size(tv, H5T_VARIABLE);
    H5Tset_cset(tv, H5T_CSET_UTF8);
    int rc = 0;
    auto put = [&](hid_t loc, const char* name, hid_t ft, hid_t mt, hid_t sp, const void* data, const std::string& where) {
        if (rc) return;
        hid_t d = H5Dcreate2(loc, name, ft, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
        if (d < 0) { rc = fail("cannot create dataset '" + where + "/" + name + "' (already exists?)"); return; }
        if (H5Dwrite(d, mt, H5S_ALL, H5S_ALL, H5P_DEFAULT, data) < 0) rc = fail("H5Dwrite failed for '" + where + "/" + name + "'");
        H5Dclose(d);
    };
    const char* name = names;
    for (int32_t i = 0; i < n && !rc; ++i, name += strlen(name) + 1) {
        const std::string where = std::string("summaries/") + name;
        hid_t g = H5Gcreate2(f->file, where.c_str(), f->lcpl, H5P_DEFAULT, H5P_DEFAULT);
        if (g < 0) { rc = fail("cannot create group '" + where + "' (already exists?)"); break; }
        put(g, "image", H5T_STD_U8LE, H5T_NATIVE_UINT8, sp_img, images + (size_t)i * seq_len * features, where);
        put(g, "label", H5T_STD_U8LE, H5T_NATIVE_UINT8, sp_row, labels + (size_t)i * seq_len, where);
        put(g, "position", H5T_STD_I64LE, H5T_NATIVE_INT64, sp_row, position + (size_t)i * seq_len, where);
        put(g, "index", H5T_STD_I64LE, H5T_NATIVE_INT64, sp_row, index + (size_t)i * seq_len, where);
        put(g, "contig", tv, tv, sp_one, &contig, where);
        put(g, "region_start", H5T_STD_I64LE, H5T_NATIVE_INT64, sp_one, &region_start, where);
        put(g, "region_end", H5T_STD_I64LE, H5T_NATIVE_INT64, sp_one, &region_end, where);
        put(g, "chunk_id", H5T_STD_I64LE, H5T_NATIVE_INT64, sp_one, chunk_id + i, where);
        H5Gclose(g);
    }
    H5Tclose(tv);
    H5Sclose(sp_img);
    H5Sclose(sp_row);
    H5Sclose(sp_one);
    return rc;
}

}  // extern "C"

// ---- stitch (pepper Stitch.py:36-94): the chunks of prediction regions merged by (position, insert index) ----------------
namespace {

struct StitchRow { int64_t pos; int64_t idx; uint8_t label; };

thread_local std::string g_stitch;       // the last piece's sequence, until pa_h5_stitch_take copies it out

// rows of every chunk of one region group, chunk ids in string order, contig_start / contig_end skipped; chunks may have any
// length.  Straight from the mapped file where the locator knows the format, through the library otherwise.
// -> 0, or -1 (error text set)
int region_rows(pa_h5* f, const char* region_path, std::vector<int64_t>& pos, std::vector<int64_t>& idx, std::vector<uint8_t>& lab,
                std::vector<size_t>* chunk_ends = nullptr) {
    pos.clear();
    idx.clear();
    lab.clear();
    if (chunk_ends) chunk_ends->clear();
    if (const Direct* dd = direct_of(f)) {
        const uint64_t g = dd->resolve(dd->root(), region_path);
        std::vector<std::pair<std::string, uint64_t>> kids;
        bool ok = g != 0 && dd->children(g, kids);
        for (size_t k = 0; ok && k < kids.size(); ++k) {
            if (kids[k].first == "contig_start" || kids[k].first == "contig_end") continue;
            Direct::Span a, b, c;
            uint64_t na = 0, nb = 0, nc = 0;
            ok = dd->dataset(dd->child(kids[k].second, "position"), 8, Direct::ANY, a, &na) &&
                 dd->dataset(dd->child(kids[k].second, "index"), 8, Direct::ANY, b, &nb) &&
                 dd->dataset(dd->child(kids[k].second, "bases"), 1, Direct::ANY, c, &nc) && na == nb && na == nc;
            if (!ok) break;
            const size_t at = pos.size();
            pos.resize(at + na);
            idx.resize(at + na);
            lab.resize(at + na);
            ok = dd->copy(a, pos.data() + at) && dd->copy(b, idx.data() + at) && dd->copy(c, lab.data() + at);
            if (chunk_ends) chunk_ends->push_back(pos.size());
        }
        if (ok) return 0;
        pos.clear();
        idx.clear();
        lab.clear();
        if (chunk_ends) chunk_ends->clear();
    }
    Quiet q;
    hid_t g = H5Gopen2(f->file, region_path, H5P_DEFAULT);
    if (g < 0) return fail(std::string("no group '") + region_path + "'");
    std::vector<std::string> names;
    H5Literate(g, H5_INDEX_NAME, H5_ITER_NATIVE, nullptr, collect_names, &names);
    names.erase(std::remove_if(names.begin(), names.end(),
                               [](const std::string& s) { return s == "contig_start" || s == "contig_end"; }), names.end());
    std::sort(names.begin(), names.end());            // the order of Python's sorted() on the chunk ids as strings
    int rc = 0;
    for (size_t i = 0; i < names.size() && !rc; ++i) {
        const std::string where = std::string(region_path) + "/" + names[i] + "/";
        hid_t c = H5Gopen2(g, names[i].c_str(), H5P_DEFAULT);
        if (c < 0) { rc = fail("no group '" + where + "'"); break; }
        hid_t d = H5Dopen2(c, "position", H5P_DEFAULT);
        hssize_t n = -1;
        if (d >= 0) {
            hid_t sp = H5Dget_space(d);
            n = H5Sget_simple_extent_npoints(sp);
            H5Sclose(sp);
            H5Dclose(d);
        }
        if (n < 0) rc = fail("no dataset '" + where + "position'");
        else {
            const size_t at = pos.size();
            pos.resize(at + (size_t)n);
            idx.resize(at + (size_t)n);
            lab.resize(at + (size_t)n);
            rc = read_numeric(c, "position", H5T_NATIVE_INT64, n, pos.data() + at, where);
            if (!rc) rc = read_numeric(c, "index", H5T_NATIVE_INT64, n, idx.data() + at, where);
            if (!rc) rc = read_numeric(c, "bases", H5T_NATIVE_UINT8, n, lab.data() + at, where);
            if (chunk_ends) chunk_ends->push_back(pos.size());
        }
        H5Gclose(c);
    }
    H5Gclose(g);
    return rc;
}

inline bool key_less(const StitchRow& a, const StitchRow& b) { return a.pos < b.pos || (a.pos == b.pos && a.idx < b.idx); }

}  // namespace

extern "C" {

int pa_h5_stitch_polish_regions(pa_h5* const* files, const int32_t* file_of_region, const char* region_paths,
                                const int64_t* region_start, int32_t n_regions, int64_t buffer_positions, int64_t* first_pos,
                                int64_t* last_pos, int64_t* sequence_len, int64_t* bad_label) {
    if (n_regions < 0 || !first_pos || !last_pos || !sequence_len || !bad_label ||
        (n_regions > 0 && (!files || !file_of_region || !region_paths || !region_start)))
        return fail("bad argument");
    *first_pos = *last_pos = -1;
    *sequence_len = 0;
    *bad_label = -1;
    g_stitch.clear();
    // `merged` holds one row per (position, index) key, in key order, with the label of the LAST write of that key in the
    // order of the reference's loops (regions as given, chunk ids as strings, rows as stored): Stitch.py:60-80 fills a dict
    // that way and sorts its keys.  The rows of a chunk come sorted and a chunk mostly continues where the one before ended,
    // so each chunk is merged into the tail of `merged` (rows of a chunk that are out of order are sorted first, stably).
    std::vector<StitchRow> merged, rows, tail;
    std::vector<int64_t> pos, idx;
    std::vector<uint8_t> lab;
    std::vector<size_t> chunk_ends;
    const char* path = region_paths;
    for (int32_t r = 0; r < n_regions; ++r, path += strlen(path) + 1) {
        if (int rc = region_rows(files[file_of_region[r]], path, pos, idx, lab, &chunk_ends)) return rc;
        // (room for the whole piece at the first region's size per region: growing a quarter-gigabyte vector by doubling copied it
        // twice over and touched every page of every copy)
        if (r == 0 && n_regions > 1) merged.reserve((size_t)n_regions * (pos.size() + pos.size() / 8));
        const int64_t st = region_start[r];
        // chunk by chunk, in the order the reference's loops write them: a chunk's rows come sorted, and consecutive chunks of a
        // region overlap by their last 50 rows -- sorting a region's rows as one list (they are NOT sorted across chunks) was
        // most of this function's time (a stable sort of ~3 000 rows per region)
        size_t k0 = 0;
        for (size_t c = 0; c < chunk_ends.size(); ++c) {
            const size_t k1 = chunk_ends[c];
            rows.clear();
            bool sorted = true;
            for (size_t k = k0; k < k1; ++k) {
                if (idx[k] < 0 || pos[k] < 0) continue;
                if (st > 0 && !(pos[k] > st + buffer_positions)) continue;      // the overlap with the region before (:62-66)
                const StitchRow row{pos[k], idx[k], lab[k]};
                if (!rows.empty() && key_less(row, rows.back())) sorted = false;
                rows.push_back(row);
            }
            k0 = k1;
            if (rows.empty()) continue;
            if (!sorted) std::stable_sort(rows.begin(), rows.end(), key_less);
            // equal keys within the rows: the last one stays
            size_t w = 0;
            for (size_t k = 0; k < rows.size(); ++k) {
                if (w > 0 && !key_less(rows[w - 1], rows[k])) rows[w - 1] = rows[k];
                else rows[w++] = rows[k];
            }
            rows.resize(w);
            // the part of `merged` at or after the first new key
            const size_t from = std::lower_bound(merged.begin(), merged.end(), rows.front(), key_less) - merged.begin();
            if (from == merged.size()) {
                merged.insert(merged.end(), rows.begin(), rows.end());
                continue;
            }
            tail.assign(merged.begin() + from, merged.end());
            merged.resize(from);
            size_t a = 0, b = 0;
            while (a < tail.size() || b < rows.size()) {
                if (b == rows.size() || (a < tail.size() && key_less(tail[a], rows[b]))) merged.push_back(tail[a++]);
                else {
                    if (a < tail.size() && !key_less(rows[b], tail[a])) ++a;   // same key: the later write wins
                    merged.push_back(rows[b++]);
                }
            }
        }
    }
    if (merged.empty()) return 0;
    static const char letters[5] = {0, 'A', 'C', 'G', 'T'};
    g_stitch.reserve(merged.size());
    for (const StitchRow& row : merged) {
        if (row.label > 4) {
            *bad_label = row.label;          // label_decoder[...] raises KeyError in the reference
            g_stitch.clear();
            return fail("label " + std::to_string((int)row.label) + " is not a base");
        }
        if (row.label) g_stitch.push_back(letters[row.label]);
    }
    *first_pos = merged.front().pos;
    *last_pos = merged.back().pos;
    *sequence_len = (int64_t)g_stitch.size();
    return 0;
}

// ---- one predictions/batch_<n> group of a variant prediction file in one call, through the locator (no libhdf5) -------------
// The candidate finder reads six datasets per batch of 512 candidates; through libhdf5 that is ~1.7 ms of CPU per batch
// (0.5 ms for the 512 variable-length strings alone) against 0.25 ms for the batch's selection.  Layout as the writers of this
// package leave it (pa_h5_write_prediction_batch / pa_h5_builder_write_prediction_batch); anything else -> 1 ("read it the
// other way"), never a guess.
namespace {
struct BatchRows {
    int64_t n = 0;
    int32_t contig_width = 0, n_classes = 0;
    std::string contigs, candidates;              // n x width bytes; n strings each followed by a NUL
    std::vector<int32_t> positions;
    std::vector<uint8_t> depths, freq;
    std::vector<double> probs;
};
thread_local BatchRows g_batch;
}  // namespace

int pa_h5_prediction_batch_load(pa_h5* f, const char* group, int64_t* n, int32_t* contig_width, int64_t* candidate_bytes,
                                int32_t* n_classes) {
    if (!f || !group || !n || !contig_width || !candidate_bytes || !n_classes) return fail("null argument");
    const Direct* dd = direct_of(f);
    if (!dd) return 1;
    const uint64_t g = dd->resolve(dd->root(), group);
    if (!g) return 1;
    Direct::Raw contigs, positions, depths, cands, freq, probs;
    if (!dd->raw(dd->child(g, "contigs"), contigs) || !dd->raw(dd->child(g, "positions"), positions) ||
        !dd->raw(dd->child(g, "depths"), depths) || !dd->raw(dd->child(g, "candidates"), cands) ||
        !dd->raw(dd->child(g, "candidate_frequency"), freq) || !dd->raw(dd->child(g, "base_prediction"), probs))
        return 1;
    const uint64_t rows = contigs.npoints;
    if (contigs.cls != 3 || contigs.rank != 1 || contigs.elem > 4096 ||
        positions.cls != 0 || positions.elem != 4 || positions.rank != 1 || positions.npoints != rows ||
        depths.cls != 0 || depths.elem != 1 || depths.rank != 1 || depths.npoints != rows ||
        cands.cls != 9 || cands.rank != 2 || cands.dims[0] != rows || cands.dims[1] != 1 ||
        freq.cls != 0 || freq.elem != 1 || freq.rank != 2 || freq.dims[0] != rows || freq.dims[1] != 1 ||
        probs.cls != 1 || probs.rank != 2 || probs.dims[0] != rows || probs.dims[1] == 0 || probs.dims[1] > 64)
        return 1;
    BatchRows& b = g_batch;
    b.n = (int64_t)rows;
    b.contig_width = (int32_t)contigs.elem;
    b.n_classes = (int32_t)probs.dims[1];
    b.contigs.assign((size_t)contigs.span.bytes, '\0');
    b.positions.resize((size_t)rows);
    b.depths.resize((size_t)rows);
    b.freq.resize((size_t)rows);
    b.probs.resize((size_t)probs.npoints);
    if (rows) {
        if (!dd->copy(contigs.span, &b.contigs[0]) || !dd->copy(positions.span, b.positions.data()) ||
            !dd->copy(depths.span, b.depths.data()) || !dd->copy(freq.span, b.freq.data()) || !dd->copy(probs.span, b.probs.data()))
            return 1;
    }
    b.candidates.clear();
    b.candidates.reserve((size_t)rows * 8);
    for (uint64_t i = 0; i < rows; ++i) {
        const uint8_t* ref = cands.span.data + 16 * i;              // {length u32, collection address u64, object index u32}
        uint32_t len, idx;
        uint64_t addr;
        std::memcpy(&len, ref, 4);
        std::memcpy(&addr, ref + 4, 8);
        std::memcpy(&idx, ref + 12, 4);
        if (len != 0 || addr != 0) {                                 // (a null reference is the empty string)
            const uint8_t* data = nullptr;
            uint64_t size = 0;
            if (!dd->gcol_object(addr, idx, data, size) || size < len) return 1;
            b.candidates.append((const char*)data, strnlen((const char*)data, len));     // C-string semantics, as H5Dread gives
        }
        b.candidates.push_back('\0');
    }
    *n = b.n;
    *contig_width = b.contig_width;
    *candidate_bytes = (int64_t)b.candidates.size();
    *n_classes = b.n_classes;
    return 0;
}

int pa_h5_prediction_batch_take(char* contigs, char* candidates, int32_t* positions, uint8_t* depths, uint8_t* freq, double* probs) {
    const BatchRows& b = g_batch;
    if (b.n > 0 && (!contigs || !candidates || !positions || !depths || !freq || !probs)) return fail("null argument");
    if (!b.contigs.empty()) std::memcpy(contigs, b.contigs.data(), b.contigs.size());
    if (!b.candidates.empty()) std::memcpy(candidates, b.candidates.data(), b.candidates.size());
    if (b.n > 0) {
        std::memcpy(positions, b.positions.data(), b.positions.size() * 4);
        std::memcpy(depths, b.depths.data(), b.depths.size());
        std::memcpy(freq, b.freq.data(), b.freq.size());
        std::memcpy(probs, b.probs.data(), b.probs.size() * 8);
    }
    return 0;
}

int pa_h5_stitch_take(char* out, int64_t cap) {
    if ((int64_t)g_stitch.size() > cap || (!out && !g_stitch.empty())) return fail("buffer too small for the stitched sequence");
    if (!g_stitch.empty()) std::memcpy(out, g_stitch.data(), g_stitch.size());
    std::string().swap(g_stitch);
    return 0;
}

// the region groups of predictions/<contig> in name order with their contig_start / contig_end (perform_stitch.py:63-74 reads
// them one h5py call at a time).  names: NUL-separated into buf (needed bytes reported); -> 0, -1 error
int pa_h5_list_polish_regions(pa_h5* f, const char* contig, char* buf, int64_t cap, int64_t* needed, int64_t* count,
                              int64_t* starts, int64_t* ends, int64_t cap_regions) {
    if (!f || !contig || !needed || !count) return fail("null argument");
    std::vector<std::string> names;
    std::vector<int64_t> st, en;
    const std::string base = std::string("predictions/") + contig;
    bool have = false;
    if (const Direct* dd = direct_of(f)) {
        const uint64_t g = dd->resolve(dd->root(), base.c_str());
        std::vector<std::pair<std::string, uint64_t>> kids;
        have = g != 0 && dd->children(g, kids);
        for (size_t k = 0; have && k < kids.size(); ++k) {
            Direct::Span a, b;
            int64_t s = 0, e = 0;
            have = dd->dataset(dd->child(kids[k].second, "contig_start"), 8, 1, a) && dd->dataset(dd->child(kids[k].second, "contig_end"), 8, 1, b) &&
                   dd->copy(a, &s) && dd->copy(b, &e);
            names.push_back(kids[k].first);
            st.push_back(s);
            en.push_back(e);
        }
        if (!have) { names.clear(); st.clear(); en.clear(); }
    }
    if (!have) {
        Quiet q;
        hid_t g = H5Gopen2(f->file, base.c_str(), H5P_DEFAULT);
        if (g < 0) return fail("no group '" + base + "'");
        H5Literate(g, H5_INDEX_NAME, H5_ITER_NATIVE, nullptr, collect_names, &names);
        std::sort(names.begin(), names.end());
        int rc = 0;
        for (size_t k = 0; k < names.size() && !rc; ++k) {
            hid_t c = H5Gopen2(g, names[k].c_str(), H5P_DEFAULT);
            const std::string where = base + "/" + names[k] + "/";
            if (c < 0) { rc = fail("no group '" + where + "'"); break; }
            int64_t s = 0, e = 0;
            rc = read_numeric(c, "contig_start", H5T_NATIVE_INT64, 1, &s, where);
            if (!rc) rc = read_numeric(c, "contig_end", H5T_NATIVE_INT64, 1, &e, where);
            st.push_back(s);
            en.push_back(e);
            H5Gclose(c);
        }
        H5Gclose(g);
        if (rc) return rc;
    }
    int64_t bytes = 0;
    for (const auto& n : names) bytes += (int64_t)n.size() + 1;
    *needed = bytes;
    *count = (int64_t)names.size();
    if (!buf || cap < bytes || !starts || !ends || cap_regions < (int64_t)names.size()) return 0;   // sizes only
    char* w = buf;
    for (size_t k = 0; k < names.size(); ++k) {
        std::memcpy(w, names[k].c_str(), names[k].size() + 1);
        w += names[k].size() + 1;
        starts[k] = st[k];
        ends[k] = en[k];
    }
    return 0;
}

}  // extern "C"
