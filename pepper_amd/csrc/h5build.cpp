// Append-only builder of classic-format HDF5 files for the polish prediction store (include/pepper_amd_io.h,
// pa_h5_builder_*).  Host-only C++; no libhdf5 in here.
//
// Why: a polish prediction file is one group of four small datasets per 1000-row chunk (pepper DataStorePredict.py:49-76), and
// libhdf5 spends 70-150 us of CPU on each such group (object headers, B-tree and heap updates through the metadata cache,
// property lists, ids) -- with the reader side at ~10 us per chunk (hdf5io.cpp's direct locator) and the device at ~5 us per
// chunk, the writer processes were what bounded call_consensus on the 16 CPUs a GPU box grants.  The file FORMAT, though, is
// simple when nothing is ever modified: raw data can be appended as it arrives, and every piece of metadata (object headers,
// local heaps, symbol nodes, group B-trees) written once, bottom-up -- children before parents, so every address is known when
// it is needed.  A chunk group is complete the moment it is written, so its metadata follows its rows at once (`seal`) and
// only its name and header address (~50-100 bytes) wait in memory for the B-tree of its parent, which close() lays out
// together with everything still open.  ~3 us per chunk plus the write() of its 18 KB.
//
// What is written (HDF5 File Format Specification 2.0): superblock version 0 (8-byte offsets and lengths, group leaf K 4,
// internal K 16), version-1 object headers, "old style" groups (symbol table message -> version-1 B-tree of symbol nodes +
// local heap), datasets with dataspace v1 / fixed-point datatype v1 / fill value v2 (no fill value) / layout v3 (contiguous,
// or compact for scalars) -- the same structures h5py's default (libver earliest) writes, read back by libhdf5, h5py and this
// repository's own readers (tests/test_hdf5_layout.py).  Until pa_h5_builder_close() has run the file is not an HDF5 file.
#include "../../include/pepper_amd_io.h"

#include <fcntl.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

extern "C" const char* pa_h5_last_error(void);
void pa_h5_set_error(const std::string& msg);      // hdf5io.cpp

namespace {

int fail(const std::string& msg) {
    pa_h5_set_error(msg);
    return -1;
}

constexpr uint64_t UNDEF = ~0ull;
constexpr int LEAF_K = 4, INTERNAL_K = 16;                    // superblock defaults: 8 symbols per node, 32 children per B-tree node
constexpr uint64_t SNOD_BYTES = 8 + 2 * LEAF_K * 40;
constexpr uint64_t TREE_BYTES = 24 + (2 * INTERNAL_K + 1) * 8 + 2 * INTERNAL_K * 8;

// A child whose metadata is already in the file (a chunk group is complete the moment it is written: its object headers,
// heap, symbol node and B-tree node follow its rows at once): only its name and header address wait for the parent's B-tree.
struct Link {
    uint64_t header;
    uint64_t name_at;                                         // into pa_h5_builder::names
    uint32_t name_len;
};

struct Obj {
    std::string name;
    bool group = false;
    std::vector<uint32_t> kids;                               // group: indices into objs_
    std::vector<Link> sealed;                                 // group: children already written out
    uint8_t elem = 0, rank = 0;                               // dataset: bytes per integer element, signedness, shape
    bool is_signed = false;
    uint64_t dims[4] = {0, 0, 0, 0};
    uint64_t bytes = 0, addr = 0;                             // raw data: size; file address (contiguous layout)
    std::string small;                                        // the data itself when it is at most 64 bytes (compact layout)
    bool vlen = false;                                        // variable-length UTF-8 strings: a scalar ((collection, object) below,
    uint32_t heap_collection = 0, heap_object = 0, heap_length = 0;   // compact layout) or an array of references at `addr`
    bool fixed_string = false;                                // numpy 'S<elem>' fields: null-padded ASCII, `elem` bytes each
    bool is_float = false;                                    // IEEE little-endian float64 (elem == 8)
};

struct HeapCollection {                                       // a global heap collection being filled (GCOL, spec III.E)
    std::vector<std::string> objects;                         // object k + 1 of the collection
    uint64_t used = 16;                                       // header + objects (16-byte object headers, data padded to 8)
    uint64_t addr = 0, size = 4096;                           // its place in the file, reserved when it is opened; written at close
};


void put16(std::vector<uint8_t>& b, uint64_t at, uint16_t v) { std::memcpy(&b[at], &v, 2); }
void put32(std::vector<uint8_t>& b, uint64_t at, uint32_t v) { std::memcpy(&b[at], &v, 4); }
void put64(std::vector<uint8_t>& b, uint64_t at, uint64_t v) { std::memcpy(&b[at], &v, 8); }

}  // namespace

struct pa_h5_builder {
    int fd = -1;
    std::string path;
    std::vector<uint8_t> out;                                 // bytes not yet handed to write()
    uint64_t pos = 0;                                         // file offset of the end of `out`
    std::vector<Obj> objs;                                    // objs[0] = root group
    std::unordered_map<std::string, uint32_t> by_path;        // groups below the root, "predictions/<contig>[/<region>]"
    bool failed = false;
    std::vector<char> names;                                  // names of the sealed children, back to back
    std::vector<HeapCollection> heaps;                        // variable-length strings: one object per distinct string
    std::unordered_map<std::string, std::pair<uint32_t, uint32_t>> heap_of;

    // (collection, object index) of a string in the global heap; equal strings share one object (h5py writes one object per
    // dataset with reference count 0; readers only ever dereference)
    std::pair<uint32_t, uint32_t> heap_object(const std::string& text) {
        auto it = heap_of.find(text);
        if (it != heap_of.end()) return it->second;
        const uint64_t need = 16 + (text.size() + 7) / 8 * 8;
        if (heaps.empty() || heaps.back().used + need + 16 > heaps.back().size || heaps.back().objects.size() >= 65000) {
            heaps.emplace_back();
            HeapCollection& fresh = heaps.back();
            if (16 + need + 16 > fresh.size) fresh.size = 16 + need + 16;
            const uint64_t pad = (8 - pos % 8) % 8;           // its bytes are written by close(); sealed datasets refer to its
            out.insert(out.end(), (size_t)(pad + fresh.size), 0);   // address from now on
            fresh.addr = pos + pad;
            pos += pad + fresh.size;
        }
        HeapCollection& h = heaps.back();
        h.objects.push_back(text);
        h.used += need;
        const std::pair<uint32_t, uint32_t> at((uint32_t)heaps.size() - 1, (uint32_t)h.objects.size());
        heap_of.emplace(text, at);
        return at;
    }

    void vlen_string(uint32_t g, const std::string& name, const std::string& text) {
        Obj d;
        d.name = name;
        d.vlen = true;
        d.elem = 16;
        d.bytes = 16;
        const auto at = heap_object(text);
        d.heap_collection = at.first;
        d.heap_object = at.second;
        d.heap_length = (uint32_t)text.size();
        add(g, std::move(d));
    }

    // n strings as an array dataset of shape dims: the references {length u32, collection address u64, object u32} are raw
    // data like any other (the collections' places are reserved when they are opened)
    void vlen_string_array(uint32_t g, const std::string& name, int rank, const uint64_t* dims, const char* blob, const int64_t* offsets,
                           uint64_t n) {
        std::vector<uint8_t> refs((size_t)n * 16, 0);
        for (uint64_t i = 0; i < n; ++i) {
            const std::string text(blob + offsets[i]);
            const auto at = heap_object(text);
            const uint32_t len = (uint32_t)text.size(), obj = at.second;
            std::memcpy(&refs[(size_t)i * 16], &len, 4);
            std::memcpy(&refs[(size_t)i * 16 + 4], &heaps[at.first].addr, 8);
            std::memcpy(&refs[(size_t)i * 16 + 12], &obj, 4);
        }
        Obj d;
        d.name = name;
        d.vlen = true;
        d.elem = 16;
        d.rank = (uint8_t)rank;
        d.bytes = 16 * n;
        for (int k = 0; k < rank; ++k) d.dims[k] = dims[k];
        // (an empty array keeps the contiguous form with an undefined address, as the library writes it)
        d.addr = n ? append(refs.data(), d.bytes) : UNDEF;
        add(g, std::move(d));
    }

    void fixed_strings(uint32_t g, const std::string& name, uint64_t n, uint32_t width, const char* data) {
        Obj d;
        d.name = name;
        d.fixed_string = true;
        d.elem = (uint8_t)width;
        d.rank = 1;
        d.dims[0] = n;
        d.bytes = n * width;
        d.addr = n ? append(data, d.bytes) : UNDEF;
        add(g, std::move(d));
    }

    int flush() {
        size_t done = 0;
        while (done < out.size()) {
            const ssize_t w = ::write(fd, out.data() + done, out.size() - done);
            if (w < 0) {
                if (errno == EINTR) continue;
                failed = true;
                return fail("write to '" + path + "' failed: " + std::strerror(errno));
            }
            done += (size_t)w;
        }
        out.clear();
        return 0;
    }

    // bytes at a place reserved earlier (string collections, the superblock); short writes and EINTR carried on from
    bool write_at(const void* data, size_t bytes, uint64_t at) {
        size_t done = 0;
        while (done < bytes) {
            const ssize_t w = ::pwrite(fd, static_cast<const char*>(data) + done, bytes - done, (off_t)(at + done));
            if (w < 0 && errno == EINTR) continue;
            if (w <= 0) return false;
            done += (size_t)w;
        }
        return true;
    }

    // raw data goes out in 8-byte aligned runs; -> its file address
    uint64_t append(const void* data, uint64_t bytes) {
        const uint64_t pad = (8 - pos % 8) % 8;
        out.insert(out.end(), (size_t)pad, 0);
        pos += pad;
        const uint64_t at = pos;
        const auto* p = static_cast<const uint8_t*>(data);
        out.insert(out.end(), p, p + bytes);
        pos += bytes;
        return at;
    }

    uint32_t add(uint32_t parent, Obj&& o) {
        objs.push_back(std::move(o));
        const uint32_t id = (uint32_t)objs.size() - 1;
        objs[parent].kids.push_back(id);
        return id;
    }

    // the group at `path` below the root, made (with its parents) when absent
    uint32_t group(const std::string& full) {
        auto it = by_path.find(full);
        if (it != by_path.end()) return it->second;
        const size_t cut = full.rfind('/');
        const uint32_t parent = cut == std::string::npos ? 0 : group(full.substr(0, cut));
        Obj g;
        g.name = cut == std::string::npos ? full : full.substr(cut + 1);
        g.group = true;
        const uint32_t id = add(parent, std::move(g));
        by_path.emplace(full, id);
        return id;
    }

    // (linear: for the small groups only; a group of very many sealed children is checked when its B-tree is laid out)
    bool has_kid(uint32_t g, const std::string& name) const {
        for (uint32_t k : objs[g].kids)
            if (objs[k].name == name) return true;
        if (objs[g].sealed.size() <= 64)
            for (const Link& l : objs[g].sealed)
                if (l.name_len == name.size() && std::memcmp(names.data() + l.name_at, name.data(), name.size()) == 0) return true;
        return false;
    }

    void dataset(uint32_t g, const std::string& name, uint8_t elem, bool is_signed, int rank, const uint64_t* dims, const void* data) {
        Obj d;
        d.name = name;
        d.elem = elem;
        d.is_signed = is_signed;
        d.rank = (uint8_t)rank;
        d.bytes = elem;
        for (int k = 0; k < rank; ++k) {
            d.dims[k] = dims[k];
            d.bytes *= dims[k];
        }
        if (d.bytes <= 64) d.small.assign(static_cast<const char*>(data), (size_t)d.bytes);
        else d.addr = append(data, d.bytes);
        add(g, std::move(d));
    }

    // an array dataset in the contiguous layout whatever its size (what `file[path] = ndarray` gives; empty: no storage yet)
    void big_dataset(uint32_t g, const std::string& name, uint8_t elem, bool is_signed, int rank, const uint64_t* dims, const void* data) {
        Obj d;
        d.name = name;
        d.elem = elem;
        d.is_signed = is_signed;
        d.rank = (uint8_t)rank;
        d.bytes = elem;
        for (int k = 0; k < rank; ++k) {
            d.dims[k] = dims[k];
            d.bytes *= dims[k];
        }
        d.addr = d.bytes ? append(data, d.bytes) : UNDEF;
        add(g, std::move(d));
    }

    // float64 array, contiguous layout (what `file[path] = np.asarray(x, np.float64)` gives)
    void float64_dataset(uint32_t g, const std::string& name, int rank, const uint64_t* dims, const double* data) {
        Obj d;
        d.name = name;
        d.elem = 8;
        d.is_float = true;
        d.rank = (uint8_t)rank;
        d.bytes = 8;
        for (int k = 0; k < rank; ++k) {
            d.dims[k] = dims[k];
            d.bytes *= dims[k];
        }
        d.addr = d.bytes ? append(data, d.bytes) : UNDEF;
        add(g, std::move(d));
    }

    void row(uint32_t g, const char* name, uint8_t elem, bool is_signed, const void* data, uint64_t count) {
        dataset(g, name, elem, is_signed, 1, &count, data);
    }

    void scalar(uint32_t g, const char* name, int64_t value) { dataset(g, name, 8, true, 0, nullptr, &value); }

    // ---- close: the metadata, children before parents, into `meta` (file address = base + offset) ----------------------
    std::vector<uint8_t> meta;
    uint64_t base = 0;
    std::string duplicate;                                    // a name met twice in one group (seen when its B-tree is laid out)

    uint64_t reserve(uint64_t bytes) {                        // 8-byte aligned, zero-filled; -> offset into meta
        const uint64_t at = (meta.size() + 7) / 8 * 8;
        meta.resize(at + bytes, 0);
        return at;
    }

    void message(uint64_t& at, uint16_t type, uint16_t size) {      // message header; the data follows at `at`
        put16(meta, at, type);
        put16(meta, at + 2, size);
        at += 8;
    }

    uint64_t dataset_header(const Obj& d) {
        const bool compact = d.addr == 0;
        const uint16_t space = (uint16_t)(8 + 8 * d.rank), dtype = (d.vlen || d.is_float) ? 24 : (d.fixed_string ? 8 : 16), fill = 8;
        const uint16_t layout = compact ? (uint16_t)((4 + d.bytes + 7) / 8 * 8) : 24;
        const uint32_t body = 4 * 8 + space + dtype + fill + layout;
        const uint64_t h = reserve(16 + body);
        meta[h] = 1;
        put16(meta, h + 2, 4);
        put32(meta, h + 4, 1);
        put32(meta, h + 8, body);
        uint64_t at = h + 16;
        message(at, 0x0001, space);                           // dataspace v1: version, rank, flags, 5 reserved, dimensions
        meta[at] = 1;
        meta[at + 1] = d.rank;
        for (int k = 0; k < d.rank; ++k) put64(meta, at + 8 + 8 * k, d.dims[k]);
        at += space;
        message(at, 0x0003, dtype);
        if (d.vlen) {
            // class 9 (variable length) v1: a string, NUL-terminated, UTF-8; 16 bytes in the file (length, collection address,
            // object index); base type = one-byte integer -- the bytes h5py writes for a Python str
            static const uint8_t vl[20] = {0x19, 0x01, 0x01, 0x00, 16, 0, 0, 0, 0x10, 0, 0, 0, 1, 0, 0, 0, 0, 0, 8, 0};
            std::memcpy(&meta[at], vl, sizeof vl);
        } else if (d.is_float) {
            // class 1 (floating point) v1, IEEE binary64 little-endian: mantissa normalisation "msb implied", sign at bit 63;
            // properties: bit offset 0, precision 64, exponent at 52 (11 bits), mantissa at 0 (52 bits), bias 1023 (spec IV.A.2.d)
            static const uint8_t f64[20] = {0x11, 0x20, 0x3f, 0x00, 8, 0, 0, 0, 0, 0, 64, 0, 52, 11, 0, 52, 0xff, 0x03, 0, 0};
            std::memcpy(&meta[at], f64, sizeof f64);
        } else if (d.fixed_string) {
            meta[at] = 0x13;                                  // datatype v1, class 3 (string): null-padded, ASCII -- what h5py
            meta[at + 1] = 0x01;                              // writes for a numpy 'S' array
            put32(meta, at + 4, d.elem);
        } else {
            meta[at] = 0x10;                                  // datatype v1, class 0 (fixed point), little-endian
            meta[at + 1] = d.is_signed ? 0x08 : 0x00;
            put32(meta, at + 4, d.elem);
            put16(meta, at + 8, 0);                           // bit offset
            put16(meta, at + 10, (uint16_t)(8 * d.elem));     // precision
        }
        at += dtype;
        message(at, 0x0005, fill);                            // fill value v2: allocation early (compact) / late, written if set,
        meta[at] = 2;                                         //   defined with size 0 = the library default
        meta[at + 1] = compact ? 1 : 2;
        meta[at + 2] = 2;
        meta[at + 3] = 1;
        at += fill;
        message(at, 0x0008, layout);                          // layout v3
        meta[at] = 3;
        if (d.vlen && !compact) {
            meta[at + 1] = 1;                                 // contiguous: the array of 16-byte references
            put64(meta, at + 2, d.addr);
            put64(meta, at + 10, d.bytes);
        } else if (d.vlen) {
            meta[at + 1] = 0;                                 // compact: the 16-byte reference into the global heap
            put16(meta, at + 2, 16);
            put32(meta, at + 4, d.heap_length);
            put64(meta, at + 8, heaps[d.heap_collection].addr);
            put32(meta, at + 16, d.heap_object);
        } else if (compact) {
            meta[at + 1] = 0;                                 // compact: size, data
            put16(meta, at + 2, (uint16_t)d.bytes);
            if (d.bytes) std::memcpy(&meta[at + 4], d.small.data(), (size_t)d.bytes);
        } else {
            meta[at + 1] = 1;                                 // contiguous: address, size
            put64(meta, at + 2, d.addr);
            put64(meta, at + 10, d.bytes);
        }
        return base + h;
    }

    // the global heap collections: "GCOL", version 1, size; objects {index u16, references u16, 4 reserved, size u64, data
    // padded to 8}; what is left belongs to object 0, whose size counts its own header (at least 4096 bytes per collection)
    int heap_collections() {
        std::vector<uint8_t> block;
        for (HeapCollection& hc : heaps) {
            block.assign((size_t)hc.size, 0);
            std::memcpy(&block[0], "GCOL", 4);
            block[4] = 1;
            put64(block, 8, hc.size);
            uint64_t at = 16;
            for (size_t k = 0; k < hc.objects.size(); ++k) {
                put16(block, at, (uint16_t)(k + 1));
                put64(block, at + 8, hc.objects[k].size());
                std::memcpy(&block[at + 16], hc.objects[k].data(), hc.objects[k].size());
                at += 16 + (hc.objects[k].size() + 7) / 8 * 8;
            }
            put64(block, at + 8, hc.size - at);               // object 0: the free space
            if (!write_at(block.data(), block.size(), hc.addr)) return fail("cannot write a string collection of '" + path + "'");
        }
        return 0;
    }

    // The metadata of a group that is complete -- its datasets' headers, its heap, symbol node, B-tree node and header -- goes
    // into the file now, behind its rows, and the group shrinks to a Link in its parent: a file of a million chunk groups keeps
    // ~50 bytes per chunk in memory until close() instead of every header's fields.  The group and its children are the
    // last objects of `objs`.
    int seal(uint32_t parent, uint32_t id) {
        const uint64_t pad = (8 - pos % 8) % 8;
        out.insert(out.end(), (size_t)pad, 0);
        pos += pad;
        meta.clear();
        base = pos;
        const uint64_t header = group_header(id);
        out.insert(out.end(), meta.begin(), meta.end());
        pos += meta.size();
        meta.clear();
        const std::string& name = objs[id].name;
        objs[parent].sealed.push_back(Link{header, (uint64_t)names.size(), (uint32_t)name.size()});
        names.insert(names.end(), name.begin(), name.end());
        objs[parent].kids.pop_back();                         // (the group was the parent's newest child)
        objs.resize(id);
        return 0;
    }

    // -> address of the group's object header; *tree / *heap receive what the superblock's root entry caches
    uint64_t group_header(uint32_t id, uint64_t* tree_out = nullptr, uint64_t* heap_out = nullptr) {
        // the children in name order: open ones (their metadata is written here, children before parents) and sealed ones
        struct Kid { const char* name; size_t len; uint64_t header; uint32_t open; };
        std::vector<Kid> kids;
        kids.reserve(objs[id].kids.size() + objs[id].sealed.size());
        for (uint32_t k : objs[id].kids) kids.push_back(Kid{objs[k].name.data(), objs[k].name.size(), 0, k});
        for (const Link& l : objs[id].sealed) kids.push_back(Kid{names.data() + l.name_at, l.name_len, l.header, 0});
        auto less = [](const Kid& a, const Kid& b) {
            const int c = std::memcmp(a.name, b.name, std::min(a.len, b.len));
            return c < 0 || (c == 0 && a.len < b.len);
        };
        std::sort(kids.begin(), kids.end(), less);
        for (size_t k = 1; k < kids.size(); ++k)
            if (!less(kids[k - 1], kids[k])) duplicate = std::string(kids[k].name, kids[k].len);
        std::vector<uint64_t> header(kids.size());
        for (size_t k = 0; k < kids.size(); ++k)
            header[k] = kids[k].open == 0 ? kids[k].header
                                          : (objs[kids[k].open].group ? group_header(kids[k].open) : dataset_header(objs[kids[k].open]));
        // local heap: "" at offset 0, then the names, each NUL-terminated and padded to 8 bytes; no free block
        std::vector<uint64_t> name_at(kids.size());
        uint64_t heap_bytes = 8;
        for (size_t k = 0; k < kids.size(); ++k) {
            name_at[k] = heap_bytes;
            heap_bytes += (kids[k].len + 1 + 7) / 8 * 8;
        }
        const uint64_t heap = reserve(32 + heap_bytes);
        std::memcpy(&meta[heap], "HEAP", 4);
        put64(meta, heap + 8, heap_bytes);
        put64(meta, heap + 16, 1);                            // H5HL_FREE_NULL: the free list is empty
        put64(meta, heap + 24, base + heap + 32);
        for (size_t k = 0; k < kids.size(); ++k) std::memcpy(&meta[heap + 32 + name_at[k]], kids[k].name, kids[k].len);
        // symbol nodes of up to 2 * LEAF_K entries
        const size_t per = 2 * LEAF_K;
        const size_t n_snod = (kids.size() + per - 1) / per;
        std::vector<uint64_t> child(n_snod), last_name(n_snod);   // one B-tree level at a time: child addresses, their greatest names
        for (size_t s = 0; s < n_snod; ++s) {
            const uint64_t at = reserve(SNOD_BYTES);
            const size_t lo = s * per, hi = std::min(kids.size(), lo + per);
            std::memcpy(&meta[at], "SNOD", 4);
            meta[at + 4] = 1;
            put16(meta, at + 6, (uint16_t)(hi - lo));
            for (size_t k = lo; k < hi; ++k) {
                const uint64_t e = at + 8 + 40 * (k - lo);
                put64(meta, e, name_at[k]);
                put64(meta, e + 8, header[k]);                // cache type 0: nothing cached in the scratch pad
            }
            child[s] = base + at;
            last_name[s] = name_at[hi - 1];
        }
        // B-tree levels until one node is left (an empty group: one node without entries)
        uint64_t root = 0;
        for (uint8_t level = 0;; ++level) {
            const size_t fan = 2 * INTERNAL_K;
            const size_t n_node = std::max<size_t>(1, (child.size() + fan - 1) / fan);
            const uint64_t first = reserve(TREE_BYTES * n_node);
            std::vector<uint64_t> up_child(n_node), up_last(n_node);
            for (size_t t = 0; t < n_node; ++t) {
                const uint64_t at = first + TREE_BYTES * t;
                const size_t lo = t * fan, hi = std::min(child.size(), lo + fan);
                std::memcpy(&meta[at], "TREE", 4);
                meta[at + 4] = 0;                             // node type 0: group nodes
                meta[at + 5] = level;
                put16(meta, at + 6, (uint16_t)(hi - lo));
                put64(meta, at + 8, t == 0 ? UNDEF : base + at - TREE_BYTES);
                put64(meta, at + 16, t + 1 == n_node ? UNDEF : base + at + TREE_BYTES);
                // key[0] = the greatest name left of this node ("" for the leftmost), key[j + 1] = the greatest name in child j
                put64(meta, at + 24, lo == 0 ? 0 : last_name[lo - 1]);
                for (size_t c = lo; c < hi; ++c) {
                    put64(meta, at + 24 + 16 * (c - lo) + 8, child[c]);
                    put64(meta, at + 24 + 16 * (c - lo) + 16, last_name[c]);
                }
                up_child[t] = base + at;
                up_last[t] = hi > lo ? last_name[hi - 1] : 0;
            }
            if (n_node == 1) {
                root = base + first;
                break;
            }
            child.swap(up_child);
            last_name.swap(up_last);
        }
        const uint64_t h = reserve(16 + 24);                  // object header: one symbol table message
        meta[h] = 1;
        put16(meta, h + 2, 1);
        put32(meta, h + 4, 1);
        put32(meta, h + 8, 24);
        uint64_t at = h + 16;
        message(at, 0x0011, 16);
        put64(meta, at, root);
        put64(meta, at + 8, base + heap);
        if (tree_out) *tree_out = root;
        if (heap_out) *heap_out = base + heap;
        return base + h;
    }

    int finish() {
        if (failed) return -1;
        const uint64_t pad = (8 - pos % 8) % 8;
        out.insert(out.end(), (size_t)pad, 0);
        pos += pad;
        if (int rc = flush()) return rc;
        if (int rc = heap_collections()) return rc;
        meta.clear();
        base = pos;
        uint64_t tree = 0, heap = 0;
        const uint64_t root = group_header(0, &tree, &heap);
        if (!duplicate.empty()) return fail("two objects named '" + duplicate + "' in one group of '" + path + "'");
        out.swap(meta);
        pos += out.size();
        if (int rc = flush()) return rc;
        uint8_t sb[96] = {0x89, 'H', 'D', 'F', '\r', '\n', 0x1a, '\n'};
        sb[13] = 8;                                           // size of offsets, size of lengths
        sb[14] = 8;
        const uint16_t lk = LEAF_K, ik = INTERNAL_K;
        std::memcpy(sb + 16, &lk, 2);
        std::memcpy(sb + 18, &ik, 2);
        const uint64_t zero = 0, undef = UNDEF;
        std::memcpy(sb + 24, &zero, 8);                       // base address
        std::memcpy(sb + 32, &undef, 8);                      // free-space info
        std::memcpy(sb + 40, &pos, 8);                        // end of file
        std::memcpy(sb + 48, &undef, 8);                      // driver info
        std::memcpy(sb + 56, &zero, 8);                       // root entry: name offset, header, cache type 1, B-tree and heap
        std::memcpy(sb + 64, &root, 8);
        const uint32_t one = 1;
        std::memcpy(sb + 72, &one, 4);
        std::memcpy(sb + 80, &tree, 8);
        std::memcpy(sb + 88, &heap, 8);
        if (!write_at(sb, sizeof sb, 0)) return fail("cannot write the superblock of '" + path + "'");
        return 0;
    }
};

extern "C" {

int pa_h5_builder_open(const char* path, pa_h5_builder** out) {
    if (!path || !out) return fail("null argument");
    const int fd = ::open(path, O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, 0644);
    if (fd < 0) return fail(std::string("cannot create '") + path + "': " + std::strerror(errno));
    auto* b = new pa_h5_builder();
    b->fd = fd;
    b->path = path;
    b->out.reserve(8 << 20);
    b->out.assign(96, 0);                                     // the superblock's place; written by close
    b->pos = 96;
    Obj root;
    root.group = true;
    b->objs.push_back(std::move(root));
    *out = b;
    return 0;
}

int pa_h5_builder_write_polish_predictions(pa_h5_builder* b, int32_t n, int32_t seq_len, const char* contigs, int32_t contig_stride,
                                           const int64_t* contig_start, const int64_t* contig_end, const int64_t* chunk_id,
                                           const uint8_t* new_region, const uint8_t* skip, const int64_t* position,
                                           const int64_t* index, const uint8_t* bases, const uint8_t* phred) {
    if (!b || n < 0 || seq_len <= 0 || contig_stride <= 0 ||
        (n > 0 && (!contigs || !contig_start || !contig_end || !chunk_id || !new_region || !skip || !position || !index ||
                   !bases || !phred)))
        return fail("bad argument");
    if (b->failed) return fail("the file has a failed write behind it");
    for (int32_t i = 0; i < n; ++i) {
        const char* c = contigs + (size_t)i * contig_stride;
        const std::string contig(c, strnlen(c, (size_t)contig_stride));
        if (contig.empty() || contig.find('/') != std::string::npos) return fail("bad contig name '" + contig + "'");
        const std::string region = "predictions/" + contig + "/" + contig + "-" + std::to_string((long long)contig_start[i]) + "-" +
                                   std::to_string((long long)contig_end[i]);
        if (new_region[i]) {
            if (b->by_path.count(region)) return fail("cannot create group '" + region + "' (already exists)");
            const uint32_t g = b->group(region);
            b->scalar(g, "contig_start", contig_start[i]);
            b->scalar(g, "contig_end", contig_end[i]);
        }
        if (skip[i]) continue;
        auto it = b->by_path.find(region);
        if (it == b->by_path.end()) return fail("no group '" + region + "' (a chunk before its region)");
        const std::string chunk = std::to_string((long long)chunk_id[i]);
        if (b->has_kid(it->second, chunk)) return fail("cannot create group '" + region + "/" + chunk + "' (already exists)");
        Obj g;
        g.name = chunk;
        g.group = true;
        const uint32_t id = b->add(it->second, std::move(g));
        b->row(id, "position", 8, true, position + (size_t)i * seq_len, (uint64_t)seq_len);
        b->row(id, "index", 8, true, index + (size_t)i * seq_len, (uint64_t)seq_len);
        b->row(id, "bases", 1, false, bases + (size_t)i * seq_len, (uint64_t)seq_len);
        b->row(id, "phred_score", 1, false, phred + (size_t)i * seq_len, (uint64_t)seq_len);
        if (int rc = b->seal(it->second, id)) return rc;
        if (b->out.size() >= (4u << 20))
            if (int rc = b->flush()) return rc;
    }
    return 0;
}

int pa_h5_builder_write(pa_h5_builder* b, const char* path, int32_t type_code, int32_t rank, const int64_t* dims, const void* data) {
    if (!b || !path || rank < 0 || rank > 4 || (rank > 0 && !dims)) return fail("bad argument");
    if (b->failed) return fail("the file has a failed write behind it");
    uint8_t elem;
    bool is_signed;
    switch (type_code) {
        case PA_H5_I8: elem = 1; is_signed = true; break;
        case PA_H5_U8: elem = 1; is_signed = false; break;
        case PA_H5_I16: elem = 2; is_signed = true; break;
        case PA_H5_U16: elem = 2; is_signed = false; break;
        case PA_H5_I32: elem = 4; is_signed = true; break;
        case PA_H5_U32: elem = 4; is_signed = false; break;
        case PA_H5_I64: elem = 8; is_signed = true; break;
        case PA_H5_U64: elem = 8; is_signed = false; break;
        default: return fail("the builder writes integer datasets only");
    }
    std::string full(path);
    while (!full.empty() && full[0] == '/') full.erase(0, 1);
    const size_t cut = full.rfind('/');
    const std::string name = cut == std::string::npos ? full : full.substr(cut + 1);
    if (name.empty() || full.find("//") != std::string::npos) return fail(std::string("bad dataset path '") + path + "'");
    const uint32_t g = cut == std::string::npos ? 0 : b->group(full.substr(0, cut));
    if (b->has_kid(g, name)) return fail(std::string("cannot create dataset '") + path + "' (already exists)");
    uint64_t d[4] = {0, 0, 0, 0}, count = 1;
    for (int k = 0; k < rank; ++k) {
        if (dims[k] < 0) return fail("negative dimension");
        d[k] = (uint64_t)dims[k];
        count *= d[k];
    }
    if (count > 0 && !data) return fail("null data");
    b->dataset(g, name, elem, is_signed, rank, d, data);
    if (b->out.size() >= (4u << 20)) return b->flush();
    return 0;
}

int pa_h5_builder_write_string(pa_h5_builder* b, const char* path, const char* text) {
    if (!b || !path || !text) return fail("bad argument");
    if (b->failed) return fail("the file has a failed write behind it");
    std::string full(path);
    while (!full.empty() && full[0] == '/') full.erase(0, 1);
    const size_t cut = full.rfind('/');
    const std::string name = cut == std::string::npos ? full : full.substr(cut + 1);
    if (name.empty() || full.find("//") != std::string::npos) return fail(std::string("bad dataset path '") + path + "'");
    const uint32_t g = cut == std::string::npos ? 0 : b->group(full.substr(0, cut));
    if (b->has_kid(g, name)) return fail(std::string("cannot create dataset '") + path + "' (already exists)");
    b->vlen_string(g, name, text);
    return 0;
}

int pa_h5_builder_write_polish_image_chunks(pa_h5_builder* b, const char* names, int32_t n, int32_t seq_len, int32_t features,
                                            const char* contig, int64_t region_start, int64_t region_end, const int64_t* chunk_id,
                                            const uint8_t* images, const uint8_t* labels, const int64_t* position,
                                            const int64_t* index) {
    if (!b || n < 0 || seq_len <= 0 || features <= 0 || !contig ||
        (n > 0 && (!names || !chunk_id || !images || !labels || !position || !index)))
        return fail("bad argument");
    if (b->failed) return fail("the file has a failed write behind it");
    const uint32_t summaries = b->group("summaries");
    const uint64_t d2[2] = {(uint64_t)seq_len, (uint64_t)features}, d1[1] = {(uint64_t)seq_len};
    const char* name = names;
    for (int32_t i = 0; i < n; ++i, name += strlen(name) + 1) {
        if (!*name || std::strchr(name, '/')) return fail(std::string("bad chunk name '") + name + "'");
        if (b->has_kid(summaries, name)) return fail(std::string("cannot create group 'summaries/") + name + "' (already exists?)");
        Obj g;
        g.name = name;
        g.group = true;
        const uint32_t id = b->add(summaries, std::move(g));
        b->dataset(id, "image", 1, false, 2, d2, images + (size_t)i * seq_len * features);
        b->dataset(id, "label", 1, false, 1, d1, labels + (size_t)i * seq_len);
        b->dataset(id, "position", 8, true, 1, d1, position + (size_t)i * seq_len);
        b->dataset(id, "index", 8, true, 1, d1, index + (size_t)i * seq_len);
        b->vlen_string(id, "contig", contig);
        b->scalar(id, "region_start", region_start);
        b->scalar(id, "region_end", region_end);
        b->scalar(id, "chunk_id", chunk_id[i]);
        if (int rc = b->seal(summaries, id)) return rc;
        if (b->out.size() >= (4u << 20))
            if (int rc = b->flush()) return rc;
    }
    return 0;
}

int pa_h5_builder_write_polish_image_regions(pa_h5_builder* b, int32_t n_regions, const char* contig, const int64_t* region_start,
                                             const int64_t* region_end, const int32_t* n_chunks, int32_t seq_len, int32_t features,
                                             const uint8_t* images, const uint8_t* labels, const int64_t* position, const int64_t* index) {
    if (!b || n_regions < 0 || seq_len <= 0 || features <= 0 || !contig || (n_regions > 0 && (!region_start || !region_end || !n_chunks)))
        return fail("bad argument");
    if (b->failed) return fail("the file has a failed write behind it");
    if (!*contig || std::strchr(contig, '/')) return fail(std::string("bad contig name '") + contig + "'");
    const uint32_t summaries = b->group("summaries");
    const uint64_t d2[2] = {(uint64_t)seq_len, (uint64_t)features}, d1[1] = {(uint64_t)seq_len};
    const std::vector<uint8_t> zeros((size_t)seq_len, 0);
    size_t k = 0;
    for (int32_t r = 0; r < n_regions; ++r) {
        if (n_chunks[r] < 0) return fail("negative chunk count");
        if (n_chunks[r] > 0 && (!images || !position || !index)) return fail("bad argument");
        const std::string stem = std::string(contig) + "_" + std::to_string(region_start[r]) + "_" + std::to_string(region_end[r]) + "_";
        for (int32_t c = 0; c < n_chunks[r]; ++c, ++k) {
            const std::string name = stem + std::to_string(c);
            if (b->has_kid(summaries, name.c_str())) continue;          // (DataStore.write_summary: a group written before is skipped)
            Obj g;
            g.name = name;
            g.group = true;
            const uint32_t id = b->add(summaries, std::move(g));
            b->dataset(id, "image", 1, false, 2, d2, images + k * seq_len * features);
            b->dataset(id, "label", 1, false, 1, d1, labels ? labels + k * seq_len : zeros.data());
            b->dataset(id, "position", 8, true, 1, d1, position + k * seq_len);
            b->dataset(id, "index", 8, true, 1, d1, index + k * seq_len);
            b->vlen_string(id, "contig", contig);
            b->scalar(id, "region_start", region_start[r]);
            b->scalar(id, "region_end", region_end[r]);
            b->scalar(id, "chunk_id", (int64_t)c);
            if (int rc = b->seal(summaries, id)) return rc;
            if (b->out.size() >= (4u << 20))
                if (int rc = b->flush()) return rc;
        }
    }
    return 0;
}

int pa_h5_builder_write_variant_summary(pa_h5_builder* b, const char* name, int32_t n, const char* contig, const int32_t* positions,
                                        const uint8_t* depths, const char* cand_blob, const int64_t* cand_offsets, const uint8_t* freqs,
                                        const int8_t* images, int32_t window, int32_t features) {
    if (!b || !name || n < 0 || !contig || window <= 0 || features <= 0 ||
        (n > 0 && (!positions || !depths || !cand_blob || !cand_offsets || !freqs || !images)))
        return fail("bad argument");
    if (b->failed) return fail("the file has a failed write behind it");
    if (!*name || std::strchr(name, '/')) return fail(std::string("bad summary name '") + name + "'");
    const size_t width = n > 0 ? std::max<size_t>(1, strlen(contig)) : 1;      // (numpy: np.array([], dtype='S') is 'S1')
    if (width > 255) return fail("contig name longer than 255 bytes");
    const uint32_t summaries = b->group("summaries");
    if (b->has_kid(summaries, name)) return fail(std::string("cannot create group 'summaries/") + name + "' (already exists)");
    Obj g;
    g.name = name;
    g.group = true;
    const uint32_t id = b->add(summaries, std::move(g));
    std::string names((size_t)n * width, '\0');
    for (int32_t i = 0; i < n; ++i) std::memcpy(&names[(size_t)i * width], contig, strlen(contig));
    const uint64_t d1[1] = {(uint64_t)n}, d2[2] = {(uint64_t)n, 1}, d3[3] = {(uint64_t)n, (uint64_t)window, (uint64_t)features};
    b->fixed_strings(id, "contigs", (uint64_t)n, (uint32_t)width, names.data());
    b->big_dataset(id, "positions", 4, true, 1, d1, positions);
    b->big_dataset(id, "depths", 1, false, 1, d1, depths);
    b->vlen_string_array(id, "candidates", 2, d2, cand_blob, cand_offsets, (uint64_t)n);
    b->big_dataset(id, "candidate_frequency", 1, false, 2, d2, freqs);
    b->big_dataset(id, "images", 1, true, 3, d3, images);
    if (int rc = b->seal(summaries, id)) return rc;
    if (b->out.size() >= (4u << 20)) return b->flush();
    return 0;
}

int pa_h5_builder_write_prediction_batch(pa_h5_builder* b, const char* name, int32_t n, const char* contigs, int32_t contig_stride,
                                         const int32_t* positions, const uint8_t* depths, const char* cand_blob,
                                         const int64_t* cand_offsets, const uint8_t* freqs, const float* probs, int32_t n_classes) {
    if (!b || !name || n < 0 || n_classes <= 0 || contig_stride <= 0 ||
        (n > 0 && (!contigs || !positions || !depths || !cand_blob || !cand_offsets || !freqs || !probs)))
        return fail("bad argument");
    if (b->failed) return fail("the file has a failed write behind it");
    if (!*name || std::strchr(name, '/')) return fail(std::string("bad batch name '") + name + "'");
    const uint32_t predictions = b->group("predictions");
    if (b->has_kid(predictions, name)) return fail(std::string("cannot create group 'predictions/") + name + "' (already exists)");
    // contigs: fixed-width, null padded, as wide as the longest name of the batch (numpy 'S' array semantics)
    size_t width = 1;
    for (int32_t i = 0; i < n; ++i) width = std::max(width, strnlen(contigs + (size_t)i * contig_stride, (size_t)contig_stride));
    if (width > 255) return fail("contig name longer than 255 bytes");
    std::string names((size_t)n * width, '\0');
    for (int32_t i = 0; i < n; ++i) {
        const char* src = contigs + (size_t)i * contig_stride;
        std::memcpy(&names[(size_t)i * width], src, strnlen(src, std::min(width, (size_t)contig_stride)));
    }
    std::vector<double> p64((size_t)n * n_classes);
    for (size_t i = 0; i < p64.size(); ++i) p64[i] = (double)probs[i];
    Obj g;
    g.name = name;
    g.group = true;
    const uint32_t id = b->add(predictions, std::move(g));
    const uint64_t d1[1] = {(uint64_t)n}, d2[2] = {(uint64_t)n, 1}, dp[2] = {(uint64_t)n, (uint64_t)n_classes};
    b->fixed_strings(id, "contigs", (uint64_t)n, (uint32_t)width, names.data());
    b->big_dataset(id, "positions", 4, true, 1, d1, positions);
    b->big_dataset(id, "depths", 1, false, 1, d1, depths);
    b->vlen_string_array(id, "candidates", 2, d2, cand_blob, cand_offsets, (uint64_t)n);
    b->big_dataset(id, "candidate_frequency", 1, false, 2, d2, freqs);
    b->float64_dataset(id, "base_prediction", 2, dp, p64.data());
    if (int rc = b->seal(predictions, id)) return rc;
    if (b->out.size() >= (4u << 20)) return b->flush();
    return 0;
}

int pa_h5_builder_close(pa_h5_builder* b) {
    if (!b) return 0;
    int rc = b->finish();
    if (b->fd >= 0 && ::close(b->fd) != 0 && !rc) rc = fail("close of '" + b->path + "' failed");
    delete b;
    return rc;
}

}  // extern "C"
