// Minimal HDF5 reader/writer behind include/pepper_amd_io.h (libhdf5 1.10 C API).
// Host-only C++ (g++), linked against /opt/conda/lib/libhdf5.so.103.
#include "../../include/pepper_amd_io.h"

#include <hdf5.h>

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

}  // namespace

struct pa_h5 {
    hid_t file = -1;
    hid_t lcpl = -1;  // create intermediate groups, as h5py's file[path] = data does
};

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
    const char* name = names;
    int rc = 0;
    for (int32_t i = 0; i < n && !rc; ++i, name += strlen(name) + 1) {
        const std::string where = std::string("summaries/") + name + "/";
        hid_t g = H5Gopen2(root, name, H5P_DEFAULT);
        if (g < 0) { rc = fail("no group '" + where + "'"); break; }
        rc = read_numeric(g, "image", H5T_NATIVE_UINT8, (int64_t)seq_len * features, images + (size_t)i * seq_len * features, where);
        if (!rc) rc = read_numeric(g, "position", H5T_NATIVE_INT64, seq_len, position + (size_t)i * seq_len, where);
        if (!rc) rc = read_numeric(g, "index", H5T_NATIVE_INT64, seq_len, index + (size_t)i * seq_len, where);
        // the small datasets: read for the first and the last chunk of the call (and checked against the name), taken from
        // the name for the others -- four of the seven objects of a chunk, at ~40 us of library time each
        std::string contig;
        int64_t s0 = 0, e0 = 0, c0 = 0;
        const bool named = from_names && parse_chunk_name(name, contig, s0, e0, c0) && (int)contig.size() < contig_stride;
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
