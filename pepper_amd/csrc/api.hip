// C ABI of pepper_amd (see include/pepper_amd.h): model handles, weight packing, workspace,
// the two forward pipelines and the HIP-event profiler.  Host-side C++ only; kernels live in
// gemm.hip / rnn.hip / head.hip.
#include "../../include/pepper_amd.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "kernels.h"

namespace {
thread_local std::string g_err;
}

namespace pa {
int set_error(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
}  // namespace pa

namespace {

int fail(int code, const std::string& msg) { return pa::set_error(code, msg); }

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(PA_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));        \
    } while (0)

constexpr int MT = 64;   // batch padding of the LSTM kernel (rnn.hip)
constexpr int MTP = 128; // batch padding of the GRU kernel (two 64-row groups per workgroup at H = 128)

inline int64_t round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    int ensure(size_t need) {
        if (need <= bytes) return PA_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
        hipError_t e = hipMalloc(&p, need);
        if (e != hipSuccess)
            return fail(PA_ERR_HIP, "hipMalloc(" + std::to_string(need) + " B): " + hipGetErrorString(e));
        bytes = need;
        return PA_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    float* f() const { return static_cast<float*>(p); }
};

struct ProfSample {
    int label;
    hipEvent_t start, stop;
    double flops;
};

struct ModelBase {
    uint32_t magic = 0x50414d44;  // 'PAMD'
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    bool profiling = false;
    std::vector<std::string> labels;
    std::vector<double> total_ms, total_flops;
    std::vector<int64_t> launches;
    std::vector<ProfSample> pending;
    std::vector<DevBuf*> owned;  // weights + workspace, freed by the destructor

    int label_id(const char* name) {
        for (size_t i = 0; i < labels.size(); ++i)
            if (labels[i] == name) return (int)i;
        labels.push_back(name);
        total_ms.push_back(0.0);
        total_flops.push_back(0.0);
        launches.push_back(0);
        return (int)labels.size() - 1;
    }
    int drain() {
        if (pending.empty()) return PA_OK;
        HIP_TRY(hipStreamSynchronize(stream));
        for (auto& s : pending) {
            float ms = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms, s.start, s.stop));
            total_ms[s.label] += ms;
            total_flops[s.label] += s.flops;
            launches[s.label] += 1;
            (void)hipEventDestroy(s.start);
            (void)hipEventDestroy(s.stop);
        }
        pending.clear();
        return PA_OK;
    }
    virtual ~ModelBase() {
        for (auto& s : pending) {
            (void)hipEventDestroy(s.start);
            (void)hipEventDestroy(s.stop);
        }
        for (DevBuf* b : owned) {
            b->release();
            delete b;
        }
        pipe_destroy();
        if (own_stream && stream) (void)hipStreamDestroy(stream);
    }
    DevBuf* new_buf() {
        owned.push_back(new DevBuf());
        return owned.back();
    }

    // Host-buffer entry points (pa_*_host): device passes of max_chunk units with the H2D copy of pass i+1 and the D2H
    // copy of pass i-1 on their own streams beside the kernels of pass i (two staging slots each way).  With page-locked
    // host buffers nothing blocks the host until the final synchronise; pageable buffers work too (HIP stages them).
    struct HostPipe {
        hipStream_t h2d = nullptr, d2h = nullptr;
        hipEvent_t in_ready[2]{}, in_free[2]{}, out_ready[2]{}, out_free[2]{};
        bool ready = false;
    } pipe;
    int pipe_init() {
        if (pipe.ready) return PA_OK;
        HIP_TRY(hipStreamCreateWithFlags(&pipe.h2d, hipStreamNonBlocking));
        HIP_TRY(hipStreamCreateWithFlags(&pipe.d2h, hipStreamNonBlocking));
        for (int k = 0; k < 2; ++k) {
            HIP_TRY(hipEventCreateWithFlags(&pipe.in_ready[k], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&pipe.in_free[k], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&pipe.out_ready[k], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&pipe.out_free[k], hipEventDisableTiming));
        }
        pipe.ready = true;
        return PA_OK;
    }
    // after a host-buffer call has failed half way: what it queued must not still be reading or writing the caller's buffers once
    // the call has returned (the error text of the failure is kept: nothing here reports)
    void quiesce() {
        if (pipe.h2d) (void)hipStreamSynchronize(pipe.h2d);
        if (stream) (void)hipStreamSynchronize(stream);
        if (pipe.d2h) (void)hipStreamSynchronize(pipe.d2h);
    }
    void pipe_destroy() {
        if (!pipe.ready) return;
        for (int k = 0; k < 2; ++k) {
            (void)hipEventDestroy(pipe.in_ready[k]);
            (void)hipEventDestroy(pipe.in_free[k]);
            (void)hipEventDestroy(pipe.out_ready[k]);
            (void)hipEventDestroy(pipe.out_free[k]);
        }
        (void)hipStreamDestroy(pipe.h2d);
        (void)hipStreamDestroy(pipe.d2h);
        pipe.ready = false;
    }
};

// RAII bracket: records two events around a launch when profiling is on.
struct Timed {
    ModelBase* m;
    ProfSample s{};
    bool on;
    Timed(ModelBase* m_, const char* label, double flops) : m(m_), on(m_->profiling) {
        if (!on) return;
        s.label = m->label_id(label);
        s.flops = flops;
        if (hipEventCreate(&s.start) != hipSuccess || hipEventCreate(&s.stop) != hipSuccess) {
            on = false;
            return;
        }
        (void)hipEventRecord(s.start, m->stream);
    }
    ~Timed() {
        if (!on) return;
        (void)hipEventRecord(s.stop, m->stream);
        m->pending.push_back(s);
    }
};

// ---- state_dict lookup -----------------------------------------------------------------------
struct StateDict {
    std::map<std::string, std::pair<const float*, int64_t>> t;
    StateDict(const char* const* names, const float* const* data, const int64_t* numel, int n) {
        for (int i = 0; i < n; ++i) {
            std::string k = names[i];
            if (k.rfind("module.", 0) == 0) k = k.substr(7);  // ModelHander.py:35-39
            t[k] = {data[i], numel[i]};
        }
    }
    const float* get(const std::string& key, int64_t expect, std::string& err) const {
        auto it = t.find(key);
        if (it == t.end()) {
            err = "state_dict is missing key '" + key + "'";
            return nullptr;
        }
        if (it->second.second != expect) {
            err = "state_dict['" + key + "'] has " + std::to_string(it->second.second) +
                  " elements, expected " + std::to_string(expect);
            return nullptr;
        }
        if (it->second.first == nullptr) {
            err = "state_dict['" + key + "'] is a null pointer";
            return nullptr;
        }
        return it->second.first;
    }
};

// Largest |w| over the 2-D tensors of a state_dict (names containing "weight").
// The split-f16 operand format carries an ACTIVATION to 22 bits relative but with an absolute floor of 2^-25 (both halves
// are f16 numbers; below 6e-5 they are sub-normal): a product a * w therefore carries an absolute error of up to
// |w| * 3e-8 whatever the size of a.  With |w| of order 1 that is far below the f32 accumulation noise; a checkpoint
// with entries in the hundreds would turn it into 1e-5 .. 1e-4 of pre-activation error on unsaturated gates (numpy
// emulation in tests/weight_families.py: 4e-6 relative logit error at |w| = 100, 6e-5 at 500, 4e-3 at 2000).  Models
// whose largest weight reaches kSplitMaxWeight therefore run on the exact-f32 matrix instructions (the
// PA_SPLIT_GEMM=0 kernels): slower, same results as the f32 oracle for any weights.
constexpr float kSplitMaxWeight = 64.0f;
float state_dict_max_abs_weight(const StateDict& sd) {
    float m = 0.0f;
    for (const auto& kv : sd.t) {
        if (kv.first.find("weight") == std::string::npos || !kv.second.first) continue;
        const float* p = kv.second.first;
        for (int64_t i = 0; i < kv.second.second; ++i) {
            const float a = std::fabs(p[i]);
            if (!(a <= m)) m = a;      // NaN ends up in m as well
        }
    }
    return m;
}

int upload(DevBuf* b, const std::vector<float>& host) {
    if (int rc = b->ensure(host.size() * sizeof(float))) return rc;
    HIP_TRY(hipMemcpy(b->p, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));
    return PA_OK;
}

// f32 [rows, K] (dense) -> h2 split format on the device (K % 8 == 0); values must fit f16 range
int upload_h2(DevBuf* b, const float* host, int64_t rows, int K) {
    for (int64_t i = 0; i < rows * K; ++i)
        if (!(std::fabs(host[i]) < 65504.0f))
            return fail(PA_ERR_INVALID, "weight magnitude >= 65504 (or NaN): not representable in the split-f16 GEMM path");
    std::vector<uint32_t> h((size_t)rows * K);
    pa::split_h2_host(host, h.data(), rows, K, K, K);
    if (int rc = b->ensure(h.size() * sizeof(uint32_t))) return rc;
    HIP_TRY(hipMemcpy(b->p, h.data(), h.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    return PA_OK;
}

// W_hh [G*H, H] of both directions -> fragment order [dir][G*H/32][H/8][64][4] (rnn.hip):
// lane l of n-tile nt, k-block kb holds W[nt*32 + (l&31)][kb*8 + 4*(l>>5) + e], e = 0..3.
void pack_rec_weights(const float* const w[2], int G, int H, std::vector<float>& out) {
    const int NTt = G * H / 32, KB = H / 8;
    out.resize((size_t)2 * NTt * KB * 64 * 4);
    for (int d = 0; d < 2; ++d)
        for (int nt = 0; nt < NTt; ++nt)
            for (int kb = 0; kb < KB; ++kb)
                for (int l = 0; l < 64; ++l)
                    for (int e = 0; e < 4; ++e)
                        out[((((size_t)d * NTt + nt) * KB + kb) * 64 + l) * 4 + e] =
                            w[d][(size_t)(nt * 32 + (l & 31)) * H + kb * 8 + 4 * (l >> 5) + e];
}

// Fused first layer (rnn.hip lstm_rec_pp_kernel<H, 32>): per direction [W_hh | W_ih | 0] with
// K = H + 32 columns, in the same fragment order as pack_rec_weights.
void pack_fused_weights(const float* const whh[2], const float* const wih[2], int G, int H, int F, int KX,
                        std::vector<float>& out) {
    const int KT = H + KX, NTt = G * H / 32, KB = KT / 8;
    out.assign((size_t)2 * NTt * KB * 64 * 4, 0.0f);
    for (int d = 0; d < 2; ++d)
        for (int nt = 0; nt < NTt; ++nt)
            for (int kb = 0; kb < KB; ++kb)
                for (int l = 0; l < 64; ++l)
                    for (int e = 0; e < 4; ++e) {
                        const int n = nt * 32 + (l & 31), k = kb * 8 + 4 * (l >> 5) + e;
                        float v = 0.0f;
                        if (k < H) v = whh[d][(size_t)n * H + k];
                        else if (k - H < F) v = wih[d][(size_t)n * F + (k - H)];
                        out[((((size_t)d * NTt + nt) * KB + kb) * 64 + l) * 4 + e] = v;
                    }
}

// One bidirectional recurrent layer's device weights.
struct RecLayer {
    DevBuf *w_cat = nullptr;   // LSTM first layer only: fused [W_hh | W_ih] fragments
    int K = 0, Kp = 0;         // input width and its zero-padded row length
    DevBuf *w_ih = nullptr;    // [2*G*H, Kp]   rows: dir*G*H + gate*H + unit
    DevBuf *w_ih_h2 = nullptr; // the same matrix in the h2 split format (gemm_h2.hip), when K % 32 == 0
    bool prescaled = false;     // LSTM h2 path: w_ih_s / b_in_s / w_ih_h2 / w_hh_h2 / w_cat_h2 carry the exp2 gate scales
    DevBuf *w_ih_s = nullptr;   // [2*G*H, Kp] f32, gate-scaled (feeds the f32 GEMM of float / unfused first layers)
    DevBuf *b_in_s = nullptr;   // [2*G*H] gate-scaled bias
    DevBuf *w_hh_h2 = nullptr;  // LSTM H=256: W_hh as h2 fragments (rnn_h2.hip)
    DevBuf *w_cat_h2 = nullptr; // LSTM first layer: [W_hh | W_ih] as h2 fragments
    DevBuf *w_cat_dec_h2 = nullptr; // layer fed by an h2 layer output (K = 2H): [W_hh | W_ih] fragments
    DevBuf *w_hh_small_h2 = nullptr; // GRU H=128: W_hh as 16x16x32 fragments for the small-call step loop (gru_small_h2_kernel)
    DevBuf *b_in = nullptr;    // [2*G*H]       LSTM: b_ih + b_hh; GRU: b_ih + (b_hr, b_hz, 0)
    DevBuf *w_hh = nullptr;    // packed
    DevBuf *b_hn = nullptr;    // GRU only: [2*H]
};

// prescale (LSTM + split recurrences only): every copy of the weights that feeds rnn_h2.hip is multiplied
// per gate row by -log2(e) (i, f, o) or +2 log2(e) (g) so the kernel's accumulators are exp2 arguments.
int build_rec_layer(ModelBase* m, const StateDict& sd, const std::string& prefix, int layer, int G,
                    int H, int K, RecLayer& out, bool prescale = false, bool want_h2 = true) {
    const char* sfx[2] = {"", "_reverse"};
    const float *wih[2], *whh[2], *bih[2], *bhh[2];
    std::string err;
    for (int d = 0; d < 2; ++d) {
        const std::string l = "_l" + std::to_string(layer) + sfx[d];
        wih[d] = sd.get(prefix + ".weight_ih" + l, (int64_t)G * H * K, err);
        whh[d] = wih[d] ? sd.get(prefix + ".weight_hh" + l, (int64_t)G * H * H, err) : nullptr;
        bih[d] = whh[d] ? sd.get(prefix + ".bias_ih" + l, (int64_t)G * H, err) : nullptr;
        bhh[d] = bih[d] ? sd.get(prefix + ".bias_hh" + l, (int64_t)G * H, err) : nullptr;
        if (!bhh[d]) return fail(PA_ERR_INVALID, err);
    }
    out.K = K;
    out.Kp = (int)round_up(K, 4);
    std::vector<float> w((size_t)2 * G * H * out.Kp, 0.0f), b((size_t)2 * G * H), bn((size_t)2 * H, 0.0f);
    for (int d = 0; d < 2; ++d)
        for (int n = 0; n < G * H; ++n) {
            std::memcpy(&w[((size_t)d * G * H + n) * out.Kp], &wih[d][(size_t)n * K], K * sizeof(float));
            float bv = bih[d][n];
            if (G == 4 || n < 2 * H) bv += bhh[d][n];   // GRU keeps b_hn out of the r-gated term
            b[(size_t)d * G * H + n] = bv;
        }
    if (G == 3)
        for (int d = 0; d < 2; ++d)
            for (int j = 0; j < H; ++j) bn[(size_t)d * H + j] = bhh[d][2 * H + j];
    std::vector<float> packed;
    pack_rec_weights(whh, G, H, packed);
    prescale = prescale && G == 4;
    out.prescaled = prescale;
    // gate-scaled copies (identical to the originals when !prescale)
    std::vector<float> ws(w), bs(b), whs[2], wis[2];
    const float* whh_x[2] = {whh[0], whh[1]};
    const float* wih_x[2] = {wih[0], wih[1]};
    if (prescale) {
        auto gscale = [&](int n) { return (n / H) == 2 ? 2.8853900817779268f : -1.4426950408889634f; };
        for (int d = 0; d < 2; ++d) {
            whs[d].assign(whh[d], whh[d] + (size_t)G * H * H);
            wis[d].assign(wih[d], wih[d] + (size_t)G * H * K);
            for (int n = 0; n < G * H; ++n) {
                const float sc = gscale(n);
                for (int k = 0; k < H; ++k) whs[d][(size_t)n * H + k] *= sc;
                for (int k = 0; k < K; ++k) wis[d][(size_t)n * K + k] *= sc;
                for (int k = 0; k < out.Kp; ++k) ws[((size_t)d * G * H + n) * out.Kp + k] *= sc;
                bs[(size_t)d * G * H + n] *= sc;
            }
            whh_x[d] = whs[d].data();
            wih_x[d] = wis[d].data();
        }
        out.w_ih_s = m->new_buf();
        out.b_in_s = m->new_buf();
        if (int rc = upload(out.w_ih_s, ws)) return rc;
        if (int rc = upload(out.b_in_s, bs)) return rc;
    }
    out.w_ih = m->new_buf();
    out.b_in = m->new_buf();
    out.w_hh = m->new_buf();
    if (int rc = upload(out.w_ih, w)) return rc;
    if (int rc = upload(out.b_in, b)) return rc;
    if (int rc = upload(out.w_hh, packed)) return rc;
    if (want_h2 && K % 32 == 0) {
        out.w_ih_h2 = m->new_buf();
        if (int rc = upload_h2(out.w_ih_h2, ws.data(), (int64_t)2 * G * H, K)) return rc;
    }
    if (G == 3) {
        out.b_hn = m->new_buf();
        if (int rc = upload(out.b_hn, bn)) return rc;
    }
    if (want_h2 && ((G == 4 && H == 256) || (G == 3 && H == 128))) {
        const int KXh2 = G == 4 ? 32 : std::max(16, pa::gru_fused_input_kx(H, K));
        // bias column of the fused first layer (rnn_h2.hip BC): the (gate-scaled) input-side bias of every row, i.e. what
        // the step loop would otherwise seed its accumulators with
        const float* const bias_x[2] = {bs.data(), bs.data() + (size_t)G * H};
        auto pack_upload = [&](DevBuf*& dst, const float* const wx[2], int KX, bool with_bias = false) -> int {
            for (int d = 0; d < 2; ++d) {
                for (int64_t i = 0; i < (int64_t)G * H * H; ++i)
                    if (!(std::fabs(whh_x[d][i]) < 65504.0f)) return fail(PA_ERR_INVALID, "recurrent weight not representable in f16 range");
                if (wx[0])
                    for (int64_t i = 0; i < (int64_t)G * H * K; ++i)
                        if (!(std::fabs(wx[d][i]) < 65504.0f)) return fail(PA_ERR_INVALID, "input weight not representable in f16 range");
            }
            if (with_bias)
                for (int64_t i = 0; i < (int64_t)2 * G * H; ++i)
                    if (!(std::fabs(bs[(size_t)i]) < 65504.0f)) return fail(PA_ERR_INVALID, "bias not representable in f16 range");
            std::vector<uint32_t> hp(pa::rec_weights_h2_words(G, H, KX));
            pa::pack_rec_weights_h2(whh_x, wx, G, H, K, KX, hp.data(), with_bias ? bias_x : nullptr);
            dst = m->new_buf();
            if (int rc = dst->ensure(hp.size() * sizeof(uint32_t))) return rc;
            HIP_TRY(hipMemcpy(dst->p, hp.data(), hp.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
            return PA_OK;
        };
        const float* const none[2] = {nullptr, nullptr};
        if (int rc = pack_upload(out.w_hh_h2, none, 0)) return rc;
        if (K <= KXh2)
            if (int rc = pack_upload(out.w_cat_h2, wih_x, KXh2, true)) return rc;
        if (K == 2 * H)
            if (int rc = pack_upload(out.w_cat_dec_h2, wih_x, K)) return rc;
        if (G == 3 && H == 128) {
            std::vector<uint32_t> hp(pa::gru_small_weights_h2_words(H));
            pa::pack_gru_small_weights_h2(whh_x, H, hp.data());
            out.w_hh_small_h2 = m->new_buf();
            if (int rc = out.w_hh_small_h2->ensure(hp.size() * sizeof(uint32_t))) return rc;
            HIP_TRY(hipMemcpy(out.w_hh_small_h2->p, hp.data(), hp.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        }
    }
    if ((G == 4 && H == 256 && K <= 32) || (G == 3 && H == 128 && K <= 16)) {
        std::vector<float> cat;
        pack_fused_weights(whh, wih, G, H, K, G == 4 ? 32 : 16, cat);
        out.w_cat = m->new_buf();
        if (int rc = upload(out.w_cat, cat)) return rc;
    }
    return PA_OK;
}

struct Linear {
    int in = 0, out = 0;
    DevBuf *w = nullptr, *b = nullptr;
    DevBuf *w_h2 = nullptr;    // h2 split copy for the f16-pipe GEMM (built on request)
};

int build_linear(ModelBase* m, const StateDict& sd, const std::string& name, int in, int out, Linear& l,
                 bool with_h2 = false) {
    std::string err;
    const float* w = sd.get(name + ".weight", (int64_t)in * out, err);
    const float* b = w ? sd.get(name + ".bias", out, err) : nullptr;
    if (!b) return fail(PA_ERR_INVALID, err);
    l.in = in;
    l.out = out;
    l.w = m->new_buf();
    l.b = m->new_buf();
    if (int rc = upload(l.w, std::vector<float>(w, w + (size_t)in * out))) return rc;
    if (with_h2 && in % 32 == 0) {
        l.w_h2 = m->new_buf();
        if (int rc = upload_h2(l.w_h2, w, out, in)) return rc;
    }
    return upload(l.b, std::vector<float>(b, b + out));
}

int init_base(ModelBase* m, int device, void* hip_stream) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return fail(PA_ERR_NO_DEVICE, "no HIP device visible: the pepper_amd product path has no CPU fallback");
    if (device < 0 || device >= count)
        return fail(PA_ERR_INVALID, "device ordinal " + std::to_string(device) + " out of range");
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
        return fail(PA_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", kernels are built for gfx950 only");
    m->device = device;
    if (hip_stream) {
        m->stream = static_cast<hipStream_t>(hip_stream);
    } else {
        HIP_TRY(hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking));
        m->own_stream = true;
    }
    return PA_OK;
}

#define LAUNCH_TRY(m, label, flops, call)                                                     \
    do {                                                                                       \
        hipError_t e_;                                                                         \
        {                                                                                      \
            Timed t_(m, label, flops);                                                         \
            e_ = (call);                                                                       \
        }                                                                                      \
        if (e_ != hipSuccess)                                                                  \
            return fail(PA_ERR_HIP, std::string(label) + ": " + hipGetErrorString(e_));        \
    } while (0)

}  // namespace

// ================================================================================================
// Variant model
// ================================================================================================
struct pa_variant_model : ModelBase {
    pa_variant_config cfg{};
    int H = 256, L1 = 512;
    bool fuse_input = true;      // PA_FUSE_INPUT=0 falls back to GEMM + Xp for A/B measurements
    bool split_gemm = true;      // PA_SPLIT_GEMM=0 keeps the big GEMMs on the f32 matrix instructions
    bool split_rec = true;       // PA_SPLIT_REC=0 keeps the recurrences on the f32 matrix instructions
    bool fuse_dec = true;        // PA_FUSE_DEC=0: decoder projection as a GEMM + Xp instead of inside the step loop
    int64_t small_batch = 3072;  // calls of at most this many windows take the GEMM + Xp decoder (PA_SMALL_BATCH)
    int64_t small_rows = 3072;   // ... and 32-row workgroups in both step loops (PA_SMALL_ROWS); at 4096 windows both schedules take 2.25 ms
    std::vector<RecLayer> rec;   // encoder layers then decoder layers
    Linear lin[5], out;
    DevBuf *mlp_w = nullptr, *mlp_b = nullptr;   // linear_2..5 as h2 fragments + their biases (mlp_h2.hip)
    DevBuf *mlp_w32 = nullptr;   // device array of the four f32 weight pointers + the out-of-range row counter behind them
    DevBuf *xp, *ya, *yb, *l1, *l2, *stage_in[2], *stage_p[2], *stage_l[2];
    // Calls of at most 1024 windows (the reference's DataLoader batch is 512) run their step loops with the hidden units of a
    // 32-row tile split over eight (above 512 windows: four) workgroups that exchange h_t through memory every step (rnn_h2.hip lstm_rec_h2_split_kernel;
    // DESIGN.md 6).  The eight must be resident together, which nothing guarantees when other kernels hold the CUs (other
    // handles, other processes): a group that does not meet within ~25 ms gives up, the call is then run again with the
    // ordinary small-call schedule (same results, the caller sees nothing but the time) and the handle leaves the split
    // alone for its next `US_HOLDOFF` small calls.  PA_UNIT_SPLIT=0: never.
    bool unit_split = true;
    int64_t unit_split_max = 1024;    // PA_UNIT_SPLIT_MAX: 512 = only the eight-member form (513-1024 windows: four members of 64 units)
    int split_holdoff = 0;            // small calls still to run without the split after a group did not meet
    int64_t split_fallbacks = 0;      // calls that were run again (pa_variant_split_fallbacks)
    int split_sabotage = 0;           // PA_UNIT_SPLIT_SABOTAGE=n (tests): in the next n split launches one member never arrives
    DevBuf *us_exch = nullptr, *us_cnt = nullptr, *us_failed = nullptr;
    int* us_host = nullptr;           // page-locked: [0] the kernel's failure flag, [1] the out-of-range row counter before the call
    ~pa_variant_model() override {
        if (us_host) (void)hipHostFree(us_host);
    }
};

extern "C" {

const char* pa_last_error(void) { return g_err.c_str(); }
const char* pa_version(void) { return "pepper_amd 0.1.0 gfx950"; }
int pa_device_count(void) {
    int c = 0;
    if (hipGetDeviceCount(&c) != hipSuccess) return 0;
    return c;
}

int pa_variant_create(const pa_variant_config* cfg, const char* const* names, const float* const* data,
                      const int64_t* numel, int32_t n_tensors, void* hip_stream, pa_variant_model** out) {
    if (!cfg || !names || !data || !numel || !out) return fail(PA_ERR_INVALID, "null argument");
    if (cfg->image_features <= 0 || cfg->window <= 0 || cfg->gru_layers <= 0 ||
        cfg->num_classes_type <= 0 || cfg->num_classes_type > 8)
        return fail(PA_ERR_INVALID, "bad pa_variant_config");
    auto* m = new pa_variant_model();
    m->cfg = *cfg;
    if (m->cfg.max_chunk <= 0) m->cfg.max_chunk = 16384;
    // the h2 GEMM addresses its A operand through a 32-bit buffer descriptor: [n*T, 2H] h2 rows must stay < 4 GiB
    m->cfg.max_chunk = std::min<int32_t>(m->cfg.max_chunk, (int32_t)((int64_t)0xf0000000 / ((int64_t)cfg->window * 2 * 256 * 4)));
    if (const char* e = getenv("PA_FUSE_INPUT")) m->fuse_input = e[0] != '0';
    if (const char* e = getenv("PA_SPLIT_GEMM")) m->split_gemm = e[0] != '0';
    if (const char* e = getenv("PA_SPLIT_REC")) m->split_rec = e[0] != '0';
    StateDict sd(names, data, numel, n_tensors);
    if (!(state_dict_max_abs_weight(sd) < kSplitMaxWeight)) m->split_gemm = false;   // see kSplitMaxWeight
    m->split_rec = m->split_rec && m->split_gemm;   // the h2 layer output needs the h2 consumers
    if (const char* e = getenv("PA_FUSE_DEC")) m->fuse_dec = e[0] != '0';
    if (const char* e = getenv("PA_SMALL_BATCH")) m->small_batch = atoll(e);
    if (const char* e = getenv("PA_SMALL_ROWS")) m->small_rows = atoll(e);
    if (const char* e = getenv("PA_UNIT_SPLIT")) m->unit_split = e[0] != '0';
    if (const char* e = getenv("PA_UNIT_SPLIT_SABOTAGE")) m->split_sabotage = atoi(e);
    if (const char* e = getenv("PA_UNIT_SPLIT_MAX")) m->unit_split_max = std::min<int64_t>(1024, atoll(e));
    int rc = init_base(m, cfg->device, hip_stream);
    const int H = m->H;
    for (int mod = 0; mod < 2 && rc == PA_OK; ++mod)
        for (int layer = 0; layer < cfg->gru_layers && rc == PA_OK; ++layer) {
            const int K = (mod == 0 && layer == 0) ? cfg->image_features : 2 * H;
            m->rec.emplace_back();
            rc = build_rec_layer(m, sd, mod == 0 ? "encoder" : "decoder", layer, 4, H, K, m->rec.back(), m->split_rec,
                                 m->split_gemm);
        }
    const char* lin_names[5] = {"linear_1", "linear_2", "linear_3", "linear_4", "linear_5"};
    for (int i = 0; i < 5 && rc == PA_OK; ++i)
        rc = build_linear(m, sd, lin_names[i], i == 0 ? 2 * H * cfg->window : m->L1, m->L1, m->lin[i],
                          i == 0 && m->split_gemm);
    if (rc == PA_OK) rc = build_linear(m, sd, "output_layer_type", m->L1, cfg->num_classes_type, m->out);
    if (rc == PA_OK && m->split_gemm && m->L1 == 512) {
        std::string err;
        const float* w4[4];
        std::vector<float> b4((size_t)4 * m->L1);                 // raw biases; the kernel's table is built from them below
        bool ok = true;
        for (int i = 0; i < 4 && ok; ++i) {
            w4[i] = sd.get(std::string(lin_names[i + 1]) + ".weight", (int64_t)m->L1 * m->L1, err);
            const float* b = sd.get(std::string(lin_names[i + 1]) + ".bias", m->L1, err);
            for (int64_t k = 0; k < (int64_t)m->L1 * m->L1; ++k) ok = ok && std::fabs(w4[i][k]) < 65504.0f;
            std::memcpy(&b4[(size_t)i * m->L1], b, m->L1 * sizeof(float));
        }
        if (ok) {   // (weights outside the f16 range keep the f32 GEMM chain)
            std::vector<uint32_t> packed(pa::mlp_weights_h2_words(4));
            float scale[4];
            pa::pack_mlp_weights_h2(w4, 4, packed.data(), scale);
            std::vector<float> table((size_t)8 * m->L1 + 4);      // [4][512] bias x scale | [4] 1 / scale | [4][512] bias
            for (int i = 0; i < 4; ++i) {
                for (int k = 0; k < m->L1; ++k) {
                    table[(size_t)i * m->L1 + k] = b4[(size_t)i * m->L1 + k] * scale[i];
                    table[(size_t)4 * m->L1 + 4 + (size_t)i * m->L1 + k] = b4[(size_t)i * m->L1 + k];
                }
                table[(size_t)4 * m->L1 + i] = 1.0f / scale[i];
            }
            b4.swap(table);
            m->mlp_w = m->new_buf();
            m->mlp_b = m->new_buf();
            rc = m->mlp_w->ensure(packed.size() * sizeof(uint32_t));
            if (rc == PA_OK && hipMemcpy(m->mlp_w->p, packed.data(), packed.size() * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess)
                rc = fail(PA_ERR_HIP, "upload of the packed MLP weights failed");
            if (rc == PA_OK) rc = upload(m->mlp_b, b4);
            if (rc == PA_OK) {      // the f32 matrices the kernel re-runs a tile on when an activation leaves the f16 range
                const float* ptrs[5] = {m->lin[1].w->f(), m->lin[2].w->f(), m->lin[3].w->f(), m->lin[4].w->f(), nullptr};
                m->mlp_w32 = m->new_buf();
                rc = m->mlp_w32->ensure(sizeof(ptrs));
                if (rc == PA_OK && hipMemcpy(m->mlp_w32->p, ptrs, sizeof(ptrs), hipMemcpyHostToDevice) != hipSuccess)
                    rc = fail(PA_ERR_HIP, "upload of the MLP weight table failed");
            }
        }
    }
    if (rc != PA_OK) {
        delete m;
        return rc;
    }
    m->xp = m->new_buf(); m->ya = m->new_buf(); m->yb = m->new_buf();
    m->l1 = m->new_buf(); m->l2 = m->new_buf();
    m->us_exch = m->new_buf(); m->us_cnt = m->new_buf(); m->us_failed = m->new_buf();
    for (int k = 0; k < 2; ++k) {
        m->stage_in[k] = m->new_buf(); m->stage_p[k] = m->new_buf(); m->stage_l[k] = m->new_buf();
    }
    *out = m;
    return PA_OK;
}

void pa_variant_destroy(pa_variant_model* m) {
    if (!m) return;
    (void)hipSetDevice(m->device);
    if (m->stream) (void)hipStreamSynchronize(m->stream);
    delete m;
}

constexpr int US_HOLDOFF = 256;

static int variant_forward_chunk(pa_variant_model* m, int a_kind, const void* images, int64_t n,
                                 float* probs, float* logits, bool allow_split = true) {
    const int T = m->cfg.window, F = m->cfg.image_features, H = m->H, C = m->cfg.num_classes_type;
    const int64_t np = round_up(n, MT);
    const int NX = 2 * 4 * H;  // both directions' gate pre-activations
    // Xp (gate pre-activations, 4.4 GB at 16384 windows) is only materialised when some layer's projection is
    // NOT contracted inside its step loop; otherwise the workspace just holds linear_1's split-K partials
    // Small calls (the reference's DataLoader batch is 512 windows, predict_distributed_gpu.py:58-67): a step of the fused
    // decoder loop costs 41 us whatever the number of 64-row tiles -- 8 waves on one CU issue the K = 768 contraction of their
    // tile, 3 MB of weight fragments per step -- so a call of 128 ... 2048 windows took 2.15 ms.  Below `small_batch` windows the
    // decoder's input projection runs as one GEMM over all T steps on the whole chip instead (0.13 ms at 512 windows) and the
    // step loop contracts K = 256 only (14.7 us per step): 512 windows 2.16 -> 1.42 ms, 1024: 2.17 -> 1.59, 2048: 2.18 -> 1.83;
    // at 4096 the fused loop wins again (profiles/r03_small_batch_kernels.json).  PA_SMALL_BATCH=0: always fused.
    const bool fuse_dec = m->fuse_dec && n > m->small_batch;
    // ... and, up to `small_rows` windows, both step loops run with 32-row workgroups (rnn_h2.hip MTILES = 1): a step is one
    // CU's affair, half the rows are half the MFMAs and half the gate phase per step (PA_SMALL_ROWS, 0 = never)
    const bool small_rows = n <= m->small_rows;
    // up to 1024 windows: every layer as projection GEMM + the unit-split step loop (see pa_variant_model::unit_split)
    bool unit_split = allow_split && m->unit_split && n <= m->unit_split_max && H == 256 && m->split_rec && !fuse_dec && m->mlp_w32 != nullptr &&
                      m->mlp_w != nullptr && C <= 8;
    for (const RecLayer& r : m->rec) unit_split = unit_split && r.w_hh_h2 != nullptr && r.prescaled;
    // the members of a group wait for each other: every workgroup of the launch must be on the device at once (a CPX
    // partition of 32 CUs, or a device that reports fewer CUs, never holds 256 of them -- no split there, no spin, no re-run)
    if (unit_split && pa::lstm_split_grid((int)n) > pa::lstm_split_resident_workgroups(n <= 512 ? 1 : 2)) unit_split = false;
    if (unit_split && m->split_holdoff > 0) {
        --m->split_holdoff;
        unit_split = false;
    }
    int* const ovf_counter = m->mlp_w32 ? reinterpret_cast<int*>(static_cast<char*>(m->mlp_w32->p) + 4 * sizeof(float*)) : nullptr;
    if (unit_split) {
        if (int rc = m->us_exch->ensure(pa::lstm_split_exchange_bytes(1024))) return rc;
        if (int rc = m->us_cnt->ensure(pa::lstm_split_counter_bytes(1024))) return rc;
        if (int rc = m->us_failed->ensure(sizeof(int))) return rc;
        if (!m->us_host) HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&m->us_host), 2 * sizeof(int), hipHostMallocDefault));
        HIP_TRY(hipMemsetAsync(m->us_failed->p, 0, sizeof(int), m->stream));
        // (the out-of-range counter as it is before this call: a pass that gave up feeds garbage to the MLP kernel)
        HIP_TRY(hipMemcpyAsync(&m->us_host[1], ovf_counter, sizeof(int), hipMemcpyDeviceToHost, m->stream));
    }
    const bool fuse_in = m->fuse_input && !unit_split;
    bool need_xp = !(a_kind == pa::A_I8 && fuse_in && !m->rec.empty() && m->rec[0].w_cat != nullptr);
    for (size_t li = 1; li < m->rec.size(); ++li)
        need_xp = need_xp || !(m->split_rec && fuse_dec && m->rec[li].w_cat_dec_h2 != nullptr);
    const size_t xp_bytes = std::max(need_xp ? (size_t)np * T * NX * sizeof(float) : (size_t)0,
                                     (size_t)8 * n * m->L1 * sizeof(float));
    if (int rc = m->xp->ensure(xp_bytes)) return rc;
    if (int rc = m->ya->ensure((size_t)np * T * 2 * H * sizeof(float))) return rc;
    if (m->rec.size() > 1)
        if (int rc = m->yb->ensure((size_t)np * T * 2 * H * sizeof(float))) return rc;
    if (int rc = m->l1->ensure((size_t)n * m->L1 * sizeof(float))) return rc;
    if (int rc = m->l2->ensure((size_t)n * m->L1 * sizeof(float))) return rc;

    const void* cur = images;
    int cur_kind = a_kind, cur_ld = F;
    bool cur_h2 = false;         // cur is a layer output already in the h2 split format
    float* ybuf[2] = {m->ya->f(), m->yb->f()};
    int which = 0;
    const int M = (int)(n * T);
    for (size_t li = 0; li < m->rec.size(); ++li) {
        const RecLayer& r = m->rec[li];
        float* y = ybuf[which];
        const bool rec_h2 = m->split_rec && r.w_hh_h2 != nullptr;
        // bias / f32 projection weights matching what the recurrent kernel of this layer expects
        const float* bias_l = (rec_h2 && r.prescaled) ? r.b_in_s->f() : r.b_in->f();
        const float* wih_l = (rec_h2 && r.prescaled) ? r.w_ih_s->f() : r.w_ih->f();
        if (li == 0 && cur_kind == pa::A_I8 && r.w_cat != nullptr && fuse_in) {
            // int8 summaries straight into the recurrent kernel: no Xp round trip
            if (rec_h2 && r.w_cat_h2 != nullptr)
                LAUNCH_TRY(m, "lstm_rec_h2_fused_in", 2.0 * n * T * (4.0 * H) * (H + r.K) * 2,
                           pa::launch_lstm_rec_h2(H, nullptr, 0, static_cast<const int8_t*>(cur), r.K, bias_l,
                                                  r.w_cat_h2->p, y, 2 * H, (int)n, T, m->stream, r.prescaled, small_rows));
            else
                LAUNCH_TRY(m, "lstm_rec_fused_in", 2.0 * n * T * (4.0 * H) * (H + r.K) * 2,
                           pa::launch_lstm_rec_fused(H, static_cast<const int8_t*>(cur), r.K, r.b_in->f(),
                                                     r.w_cat->f(), y, 2 * H, (int)n, T, m->stream));
        } else if (li > 0 && rec_h2 && cur_h2 && r.w_cat_dec_h2 != nullptr && fuse_dec) {
            // h2 layer output -> this layer: projection contracted inside the step loop (no GEMM, no Xp)
            LAUNCH_TRY(m, "lstm_dec_h2_fused", 2.0 * n * T * (4.0 * H) * (H + r.K) * 2,
                       pa::launch_lstm_dec_h2(H, cur, cur_ld, bias_l, r.w_cat_dec_h2->p, y, 2 * H, (int)n, T, m->stream,
                                              r.prescaled));
        } else {
            if (li > 0 && m->split_gemm && r.w_ih_h2 != nullptr) {
                // the previous layer's y is only read by this projection: if it is still f32, split it
                // in place; then the three-MFMA f16 product (gemm_h2.hip)
                const size_t a_bytes = (size_t)M * cur_ld * sizeof(float);
                if (!cur_h2)
                    LAUNCH_TRY(m, "cvt_h2", 0.0,
                               pa::launch_f32_to_h2(static_cast<const float*>(cur), const_cast<void*>(cur), M, r.K,
                                                    cur_ld, m->stream));
                LAUNCH_TRY(m, "gemm_h2_inproj", 2.0 * M * NX * r.K,
                           pa::launch_gemm_h2(cur, cur_ld, a_bytes, r.w_ih_h2->p, r.K, (size_t)NX * r.K * 4, bias_l,
                                              m->xp->f(), NX, (int)(np * T), NX, r.K, 0, 0, 0, T, (int)n, m->stream));
            } else {
                LAUNCH_TRY(m, li == 0 ? "gemm_inproj_in" : "gemm_inproj", 2.0 * M * NX * r.K,
                           pa::launch_gemm_nt(cur_kind, cur, cur_ld, wih_l, r.Kp, bias_l, m->xp->f(),
                                              NX, (int)(np * T), NX, r.K, 0, 0, 0, T, (int)n, m->stream));
            }
            if (rec_h2 && unit_split) {
                const int sabotage = m->split_sabotage > 0 ? 1 : 0;
                m->split_sabotage -= sabotage;
                LAUNCH_TRY(m, "lstm_rec_h2_split", 2.0 * n * T * (4.0 * H) * H * 2,
                           pa::launch_lstm_rec_h2_split(H, m->xp->f(), NX, r.w_hh_h2->p, y, 2 * H, (int)n, T, m->us_exch->p,
                                                        m->us_cnt->p, static_cast<int*>(m->us_failed->p), m->stream, sabotage));
            } else if (rec_h2)
                LAUNCH_TRY(m, "lstm_rec_h2", 2.0 * n * T * (4.0 * H) * H * 2,
                           pa::launch_lstm_rec_h2(H, m->xp->f(), NX, nullptr, 0, nullptr, r.w_hh_h2->p, y, 2 * H, (int)n,
                                                  T, m->stream, r.prescaled, small_rows));
            else
                LAUNCH_TRY(m, "lstm_rec", 2.0 * n * T * (4.0 * H) * H * 2,
                           pa::launch_lstm_rec(H, m->xp->f(), NX, r.w_hh->f(), y, 2 * H, (int)n, T, m->stream));
        }
        cur = y;
        cur_kind = pa::A_F32;
        cur_ld = 2 * H;
        cur_h2 = rec_h2;
        which ^= 1;
    }
    // flatten(start_dim=1, end_dim=2): [n, T, 2H] rows are already contiguous -> [n, T*2H]
    const int K1 = T * 2 * H;
    // linear_1 has few output tiles (n/256 x 2) and a long K: slice K so that every CU gets a tile; the
    // partial sums reuse the Xp workspace (dead after the last recurrent layer)
    const int tiles1 = (int)((n + 255) / 256) * ((m->L1 + 255) / 256);
    int splits = 1;
    // (up to 32 slices: a call of 512 windows has 4 output tiles, 0.137 ms with 8 slices, 0.078 with 32; big calls have tiles
    // enough and stay at 2 slices.  PA_L1_SPLITS overrides the cap)
    static const int max_splits = getenv("PA_L1_SPLITS") ? atoi(getenv("PA_L1_SPLITS")) : 32;
    while (splits < max_splits && tiles1 * splits * 2 <= 256 && (size_t)(splits * 2) * n * m->L1 * sizeof(float) <= m->xp->bytes)
        splits *= 2;
    if (m->split_gemm && m->lin[0].w_h2 != nullptr && cur_kind == pa::A_F32) {
        if (!cur_h2)
            LAUNCH_TRY(m, "cvt_h2", 0.0,
                       pa::launch_f32_to_h2(static_cast<const float*>(cur), const_cast<void*>(cur), n, K1, K1, m->stream));
        LAUNCH_TRY(m, "gemm_h2_linear_1", 2.0 * n * m->L1 * K1,
                   pa::launch_gemm_h2(cur, K1, (size_t)n * K1 * 4, m->lin[0].w_h2->p, K1, (size_t)m->L1 * K1 * 4,
                                      m->lin[0].b->f(), m->l1->f(), m->L1, (int)n, m->L1, K1, 1, 0, 0, 0, 0, m->stream,
                                      splits > 1 ? m->xp->f() : nullptr, splits));
    } else {
        LAUNCH_TRY(m, "gemm_linear_1", 2.0 * n * m->L1 * K1,
                   pa::launch_gemm_nt(pa::A_F32, cur, K1, m->lin[0].w->f(), K1, m->lin[0].b->f(), m->l1->f(),
                                      m->L1, (int)n, m->L1, K1, 1, 0, 0, 0, 0, m->stream));
    }
    if (m->mlp_w != nullptr && C <= 8) {
        LAUNCH_TRY(m, "mlp_tail_h2", 2.0 * n * m->L1 * (4.0 * m->L1 + C),
                   pa::launch_mlp_tail_h2(m->l1->f(), m->L1, m->mlp_w->p, m->mlp_b->f(), 4, m->out.w->f(), m->out.b->f(), C,
                                          probs, logits, (int)n, m->stream, static_cast<const float* const*>(m->mlp_w32->p),
                                          reinterpret_cast<int*>(static_cast<char*>(m->mlp_w32->p) + 4 * sizeof(float*))));
        if (unit_split) {
            // did every group of the split step loops meet?  (one synchronise per small call: 10-20 us beside its 0.65 ms)
            HIP_TRY(hipMemcpyAsync(&m->us_host[0], m->us_failed->p, sizeof(int), hipMemcpyDeviceToHost, m->stream));
            HIP_TRY(hipStreamSynchronize(m->stream));
            if (m->us_host[0] != 0) {
                ++m->split_fallbacks;
                m->split_holdoff = US_HOLDOFF;
                HIP_TRY(hipMemcpyAsync(ovf_counter, &m->us_host[1], sizeof(int), hipMemcpyHostToDevice, m->stream));
                return variant_forward_chunk(m, a_kind, images, n, probs, logits, false);
            }
        }
        return PA_OK;
    }
    float* a = m->l1->f();
    float* b = m->l2->f();
    for (int i = 1; i < 5; ++i) {
        LAUNCH_TRY(m, "gemm_linear_2to5", 2.0 * n * m->L1 * m->L1,
                   pa::launch_gemm_nt(pa::A_F32, a, m->L1, m->lin[i].w->f(), m->L1, m->lin[i].b->f(), b,
                                      m->L1, (int)n, m->L1, m->L1, 1, 0, 0, 0, 0, m->stream));
        std::swap(a, b);
    }
    LAUNCH_TRY(m, "head_softmax", 2.0 * n * m->L1 * C,
               pa::launch_dense_small(0, a, m->L1, m->out.w->f(), m->out.b->f(), probs, logits, (int)n,
                                      m->L1, C, 1, 1, 0, m->stream));
    return PA_OK;
}

static int variant_forward(pa_variant_model* m, int a_kind, const void* images, size_t elem, int64_t n,
                           float* probs, float* logits) {
    if (!m || m->magic != 0x50414d44) return fail(PA_ERR_INVALID, "bad model handle");
    if (n < 0 || (n > 0 && (!images || !probs))) return fail(PA_ERR_INVALID, "null buffer");
    HIP_TRY(hipSetDevice(m->device));
    const int64_t per = (int64_t)m->cfg.window * m->cfg.image_features;
    const int C = m->cfg.num_classes_type;
    for (int64_t off = 0; off < n; off += m->cfg.max_chunk) {
        const int64_t c = std::min<int64_t>(m->cfg.max_chunk, n - off);
        const char* img = static_cast<const char*>(images) + (size_t)off * per * elem;
        if (int rc = variant_forward_chunk(m, a_kind, img, c, probs + off * C,
                                           logits ? logits + off * C : nullptr))
            return rc;
    }
    return PA_OK;
}

int pa_variant_split_fallbacks(pa_variant_model* m, int64_t* calls) {
    if (!m || m->magic != 0x50414d44 || !calls) return fail(PA_ERR_INVALID, "bad argument");
    *calls = m->split_fallbacks;
    return PA_OK;
}

int pa_variant_overflow_rows(pa_variant_model* m, int64_t* rows) {
    if (!m || m->magic != 0x50414d44 || !rows) return fail(PA_ERR_INVALID, "bad model handle");
    HIP_TRY(hipSetDevice(m->device));
    int v = 0;
    if (m->mlp_w32 != nullptr) {
        HIP_TRY(hipStreamSynchronize(m->stream));
        HIP_TRY(hipMemcpy(&v, static_cast<char*>(m->mlp_w32->p) + 4 * sizeof(float*), sizeof(int), hipMemcpyDeviceToHost));
    }
    *rows = v;
    return PA_OK;
}

int pa_variant_forward_device(pa_variant_model* m, const int8_t* images, int64_t n, float* probs,
                              float* logits) {
    return variant_forward(m, pa::A_I8, images, 1, n, probs, logits);
}

int pa_variant_forward_device_f32(pa_variant_model* m, const float* images, int64_t n, float* probs,
                                  float* logits) {
    return variant_forward(m, pa::A_F32_SCALAR, images, 4, n, probs, logits);
}

static int variant_forward_host_run(pa_variant_model* m, const int8_t* images, int64_t n, float* probs, float* logits) {
    HIP_TRY(hipSetDevice(m->device));
    if (int rc = m->pipe_init()) return rc;
    const size_t per = (size_t)m->cfg.window * m->cfg.image_features;
    const int C = m->cfg.num_classes_type;
    const int64_t chunk = m->cfg.max_chunk;
    auto& pp = m->pipe;
    int64_t i = 0;
    for (int64_t off = 0; off < n; off += chunk, ++i) {
        const int64_t c = std::min<int64_t>(chunk, n - off);
        const int k = (int)(i & 1);
        if (int rc = m->stage_in[k]->ensure((size_t)std::min<int64_t>(chunk, n) * per)) return rc;
        if (int rc = m->stage_p[k]->ensure((size_t)std::min<int64_t>(chunk, n) * C * sizeof(float))) return rc;
        if (logits)
            if (int rc = m->stage_l[k]->ensure((size_t)std::min<int64_t>(chunk, n) * C * sizeof(float))) return rc;
        if (i >= 2) HIP_TRY(hipStreamWaitEvent(pp.h2d, pp.in_free[k], 0));      // pass i-2 has consumed this slot
        HIP_TRY(hipMemcpyAsync(m->stage_in[k]->p, images + (size_t)off * per, (size_t)c * per, hipMemcpyHostToDevice, pp.h2d));
        HIP_TRY(hipEventRecord(pp.in_ready[k], pp.h2d));
        HIP_TRY(hipStreamWaitEvent(m->stream, pp.in_ready[k], 0));
        if (i >= 2) HIP_TRY(hipStreamWaitEvent(m->stream, pp.out_free[k], 0));  // results of pass i-2 have left the slot
        if (int rc = variant_forward_chunk(m, pa::A_I8, m->stage_in[k]->p, c, m->stage_p[k]->f(),
                                           logits ? m->stage_l[k]->f() : nullptr))
            return rc;
        HIP_TRY(hipEventRecord(pp.in_free[k], m->stream));
        HIP_TRY(hipEventRecord(pp.out_ready[k], m->stream));
        HIP_TRY(hipStreamWaitEvent(pp.d2h, pp.out_ready[k], 0));
        HIP_TRY(hipMemcpyAsync(probs + off * C, m->stage_p[k]->p, (size_t)c * C * sizeof(float), hipMemcpyDeviceToHost, pp.d2h));
        if (logits)
            HIP_TRY(hipMemcpyAsync(logits + off * C, m->stage_l[k]->p, (size_t)c * C * sizeof(float), hipMemcpyDeviceToHost, pp.d2h));
        HIP_TRY(hipEventRecord(pp.out_free[k], pp.d2h));
    }
    HIP_TRY(hipStreamSynchronize(pp.d2h));
    HIP_TRY(hipStreamSynchronize(m->stream));
    return PA_OK;
}

int pa_variant_forward_host(pa_variant_model* m, const int8_t* images, int64_t n, float* probs,
                            float* logits) {
    if (!m || m->magic != 0x50414d44) return fail(PA_ERR_INVALID, "bad model handle");
    if (n < 0 || (n > 0 && (!images || !probs))) return fail(PA_ERR_INVALID, "null buffer");
    if (n == 0) return PA_OK;
    const int rc = variant_forward_host_run(m, images, n, probs, logits);
    if (rc != PA_OK) m->quiesce();
    return rc;
}

}  // extern "C"

// ================================================================================================
// Polish model
// ================================================================================================
struct pa_polish_model : ModelBase {
    pa_polish_config cfg{};
    bool fuse_input = true;      // PA_FUSE_INPUT=0 falls back to GEMM + Xp
    bool split_gemm = true;      // PA_SPLIT_GEMM=0 keeps the projections on the f32 matrix instructions
    bool split_rec = true;       // PA_SPLIT_REC=0 keeps the recurrences on the f32 matrix instructions
    bool fuse_dec = true;        // PA_FUSE_DEC=0: decoder projection as a GEMM + Xp instead of inside the step loop
    bool y_h2 = false;           // format of the last polish_window output
    bool fuse_head = true;       // PA_FUSE_HEAD=0: last decoder layer writes y, dense1 + softmax + overlap-add as their own kernel
    int small_max = 4096;        // calls of at most this many chunks take the small-call schedule (PA_POLISH_SMALL_MAX; 0: never)
    std::vector<RecLayer> enc, dec;
    Linear dense;
    DevBuf *dense_h2 = nullptr;  // dense1 as h2 fragments of the 16x16x32 tile (rnn_h2.hip pack_dense_head_h2)
    DevBuf *part = nullptr;      // per-direction partial logits of the fused head
    DevBuf *xp, *y1, *y2, *hid_a, *hid_b, *acc, *stage_in[2], *stage_lab[2], *stage_ph[2], *stage_acc[2];
};

// One module forward on n (<= padded workspace) sequences of T steps.
//   x_kind/x/x_ld/x_rpb/x_bstride describe the first layer's input rows (see launch_gemm_nt);
//   hidden_in may be null (zeros); hidden_out receives the decoder's final states.
// head_fused (in/out): the caller wants softmax-ready logits only; set to true when the last decoder layer ran with
// dense1 contracted inside its step loop (m->part then holds the partial logits and *y_last is not written).
static int polish_window(pa_polish_model* m, int x_kind, const void* x, int x_ld, int x_rpb,
                         int64_t x_bstride, const float* hidden_in, float* hidden_out, int64_t n, int T,
                         float** y_last, bool* head_fused = nullptr) {
    const bool want_head = head_fused && *head_fused;
    if (head_fused) *head_fused = false;
    const int H = m->cfg.hidden_size, L = m->cfg.gru_layers, NX = 2 * 3 * H, ldh = 2 * L * H;
    const int M = (int)(n * T);
    const void* cur = x;
    int cur_kind = x_kind, cur_ld = x_ld, cur_rpb = x_rpb;
    int64_t cur_bs = x_bstride;
    bool cur_h2 = false;         // cur is a layer output in the h2 split format
    float* ybuf[2] = {m->y1->f(), m->y2->f()};
    int which = 0;
    // encoder: h0 = hidden_in, h_n -> hid_a ; decoder: h0 = hid_a, h_n -> hidden_out
    for (int stage = 0; stage < 2; ++stage) {
        std::vector<RecLayer>& layers = stage == 0 ? m->enc : m->dec;
        const float* h0 = stage == 0 ? hidden_in : m->hid_a->f();
        float* hn = stage == 0 ? m->hid_a->f() : hidden_out;
        for (int l = 0; l < L; ++l) {
            const RecLayer& r = layers[l];
            const bool rec_h2 = m->split_rec && r.w_hh_h2 != nullptr;
            const float* h0l = h0 ? h0 + (size_t)l * 2 * H : nullptr;
            float* hnl = hn ? hn + (size_t)l * 2 * H : nullptr;
            float* y = ybuf[which];
            const int64_t xbs = cur_bs > 0 ? cur_bs : (int64_t)T * r.K;
            // wide uint8 inputs (16 < F <= 128) are read as dwords by the step loop: rows must be 4-byte aligned
            const bool fused_h2_ok = rec_h2 && r.w_cat_h2 != nullptr &&
                                     (r.K <= 16 || ((xbs & 3) == 0 && (reinterpret_cast<uintptr_t>(cur) & 3) == 0));
            // Small calls: a step's latency is what counts (19 windows x 200 dependent steps whatever n is), so the projections
            // run as GEMMs over all T steps and the step loops as 16-row workgroups with their weights in registers
            const bool small = m->small_max > 0 && n <= m->small_max && rec_h2 && r.w_hh_small_h2 != nullptr && H == 128 &&
                               (cur_h2 ? (r.w_ih_h2 != nullptr && m->split_gemm) : (stage == 0 && l == 0));
            if (small) {
                if (cur_h2)
                    LAUNCH_TRY(m, "gemm_h2_inproj_small", 2.0 * M * NX * r.K,
                               pa::launch_gemm_h2(cur, cur_ld, (size_t)M * cur_ld * 4, r.w_ih_h2->p, r.K,
                                                  (size_t)NX * r.K * 4, r.b_in->f(), m->xp->f(), NX,
                                                  (int)(round_up(n, MTP) * T), NX, r.K, 0, 0, 0, T, (int)n, m->stream));
                else
                    LAUNCH_TRY(m, "gemm_inproj_in_small", 2.0 * M * NX * r.K,
                               pa::launch_gemm_nt(cur_kind, cur, cur_ld, r.w_ih->f(), r.Kp, r.b_in->f(), m->xp->f(), NX,
                                                  (int)(round_up(n, MTP) * T), NX, r.K, 0, 0, cur_bs, T, (int)n, m->stream));
                LAUNCH_TRY(m, "gru_small_h2", 2.0 * n * T * (3.0 * H) * H * 2,
                           pa::launch_gru_small_h2(H, m->xp->f(), NX, r.w_hh_small_h2->p, r.b_hn->f(), h0l, ldh, hnl, ldh, y, 2 * H,
                                                   (int)n, T, m->stream));
                cur_h2 = true;
            } else if (stage == 0 && l == 0 && cur_kind == pa::A_U8 && (r.w_cat != nullptr || fused_h2_ok) && m->fuse_input) {
                if (fused_h2_ok)
                    LAUNCH_TRY(m, "gru_rec_h2_fused_in", 2.0 * n * T * (3.0 * H) * (H + r.K) * 2,
                               pa::launch_gru_rec_h2(H, nullptr, 0, static_cast<const uint8_t*>(cur), r.K, xbs, r.b_in->f(),
                                                     r.w_cat_h2->p, r.b_hn->f(), h0l, ldh, hnl, ldh, y, 2 * H, (int)n, T,
                                                     m->stream));
                else
                    LAUNCH_TRY(m, "gru_rec_fused_in", 2.0 * n * T * (3.0 * H) * (H + r.K) * 2,
                               pa::launch_gru_rec_fused(H, static_cast<const uint8_t*>(cur), r.K, xbs, r.b_in->f(),
                                                        r.w_cat->f(), r.b_hn->f(), h0l, ldh, hnl, ldh, y, 2 * H, (int)n, T,
                                                        m->stream));
                cur_h2 = fused_h2_ok;
            } else if (want_head && stage == 1 && l == L - 1 && rec_h2 && cur_h2 && r.w_cat_dec_h2 != nullptr && cur_rpb == 0 &&
                       cur_bs == 0 && m->fuse_dec && m->fuse_head && m->dense_h2 != nullptr && H == 128) {
                // last layer: projection AND dense1 contracted inside the step loop; no layer output at all
                if (int rc = m->part->ensure(pa::dense_partials_floats((int)n, T) * sizeof(float))) return rc;
                LAUNCH_TRY(m, "gru_dec_h2_fused_dense", 2.0 * n * T * ((3.0 * H) * (H + r.K) + m->cfg.num_classes * H) * 2,
                           pa::launch_gru_dec_h2_dense(H, cur, cur_ld, r.b_in->f(), r.w_cat_dec_h2->p, r.b_hn->f(), h0l, ldh, hnl,
                                                       ldh, m->dense_h2->p, m->part->f(), (int)n, T, m->stream));
                *head_fused = true;
                cur_h2 = true;
            } else if (rec_h2 && cur_h2 && r.w_cat_dec_h2 != nullptr && cur_rpb == 0 && cur_bs == 0 && m->fuse_dec) {
                // h2 layer output -> this layer: projection contracted inside the step loop (no GEMM, no Xp)
                LAUNCH_TRY(m, "gru_dec_h2_fused", 2.0 * n * T * (3.0 * H) * (H + r.K) * 2,
                           pa::launch_gru_dec_h2(H, cur, cur_ld, r.b_in->f(), r.w_cat_dec_h2->p, r.b_hn->f(), h0l, ldh, hnl,
                                                 ldh, y, 2 * H, (int)n, T, m->stream));
                cur_h2 = true;
            } else {
                if (!(stage == 0 && l == 0) && m->split_gemm && r.w_ih_h2 != nullptr) {
                    // the previous layer's y (workspace, read only here): split in place if needed, f16-pipe GEMM
                    if (!cur_h2)
                        LAUNCH_TRY(m, "cvt_h2", 0.0,
                                   pa::launch_f32_to_h2(static_cast<const float*>(cur), const_cast<void*>(cur), M, r.K,
                                                        cur_ld, m->stream));
                    LAUNCH_TRY(m, "gemm_h2_inproj", 2.0 * M * NX * r.K,
                               pa::launch_gemm_h2(cur, cur_ld, (size_t)M * cur_ld * 4, r.w_ih_h2->p, r.K,
                                                  (size_t)NX * r.K * 4, r.b_in->f(), m->xp->f(), NX,
                                                  (int)(round_up(n, MTP) * T), NX, r.K, 0, 0, 0, T, (int)n, m->stream));
                } else {
                    LAUNCH_TRY(m, (stage == 0 && l == 0) ? "gemm_inproj_in" : "gemm_inproj", 2.0 * M * NX * r.K,
                               pa::launch_gemm_nt(cur_kind, cur, cur_ld, r.w_ih->f(), r.Kp, r.b_in->f(), m->xp->f(), NX,
                                                  (int)(round_up(n, MTP) * T), NX, r.K, 0, 0, cur_bs, T, (int)n, m->stream));
                }
                if (rec_h2)
                    LAUNCH_TRY(m, "gru_rec_h2", 2.0 * n * T * (3.0 * H) * H * 2,
                               pa::launch_gru_rec_h2(H, m->xp->f(), NX, nullptr, 0, 0, nullptr, r.w_hh_h2->p, r.b_hn->f(),
                                                     h0l, ldh, hnl, ldh, y, 2 * H, (int)n, T, m->stream));
                else
                    LAUNCH_TRY(m, "gru_rec", 2.0 * n * T * (3.0 * H) * H * 2,
                               pa::launch_gru_rec(H, m->xp->f(), NX, r.w_hh->f(), r.b_hn->f(), h0l, ldh, hnl, ldh, y,
                                                  2 * H, (int)n, T, m->stream));
                cur_h2 = rec_h2;
            }
            cur = y;
            cur_kind = pa::A_F32;
            cur_ld = 2 * H;
            cur_rpb = 0;
            cur_bs = 0;
            which ^= 1;
        }
    }
    m->y_h2 = cur_h2;
    (void)cur_rpb;
    *y_last = const_cast<float*>(static_cast<const float*>(cur));
    return PA_OK;
}

static int polish_ensure(pa_polish_model* m, int64_t n, int T) {
    const int H = m->cfg.hidden_size, L = m->cfg.gru_layers;
    const int64_t np = round_up(n, MTP);
    if (int rc = m->xp->ensure((size_t)np * T * 6 * H * sizeof(float))) return rc;
    if (int rc = m->y1->ensure((size_t)np * T * 2 * H * sizeof(float))) return rc;
    if (int rc = m->y2->ensure((size_t)np * T * 2 * H * sizeof(float))) return rc;
    if (int rc = m->hid_a->ensure((size_t)np * 2 * L * H * sizeof(float))) return rc;
    if (int rc = m->hid_b->ensure((size_t)np * 2 * L * H * sizeof(float))) return rc;
    return PA_OK;
}

extern "C" {

int pa_polish_create(const pa_polish_config* cfg, const char* const* names, const float* const* data,
                     const int64_t* numel, int32_t n_tensors, void* hip_stream, pa_polish_model** out) {
    if (!cfg || !names || !data || !numel || !out) return fail(PA_ERR_INVALID, "null argument");
    if (cfg->image_features <= 0 || cfg->gru_layers <= 0 || cfg->num_classes <= 0 || cfg->num_classes > 8 ||
        (cfg->hidden_size != 128 && cfg->hidden_size != 256) || cfg->seq_length <= 0 || cfg->window <= 0 ||
        cfg->jump <= 0 || cfg->window > cfg->seq_length || cfg->overlap < 0 || 2 * cfg->overlap > cfg->seq_length)
        return fail(PA_ERR_INVALID, "bad pa_polish_config (hidden_size must be 128 or 256)");
    auto* m = new pa_polish_model();
    m->cfg = *cfg;
    if (m->cfg.max_chunk <= 0) m->cfg.max_chunk = 16384;   // 128 chunks per workgroup and direction: 256 workgroups
    m->cfg.max_chunk = std::min<int32_t>(m->cfg.max_chunk, (int32_t)((int64_t)0xf0000000 / ((int64_t)cfg->window * 2 * cfg->hidden_size * 4)));
    if (const char* e = getenv("PA_FUSE_INPUT")) m->fuse_input = e[0] != '0';
    if (const char* e = getenv("PA_SPLIT_GEMM")) m->split_gemm = e[0] != '0';
    if (const char* e = getenv("PA_SPLIT_REC")) m->split_rec = e[0] != '0';
    StateDict sd(names, data, numel, n_tensors);
    if (!(state_dict_max_abs_weight(sd) < kSplitMaxWeight)) m->split_gemm = false;   // see kSplitMaxWeight
    m->split_rec = m->split_rec && m->split_gemm && cfg->hidden_size == 128;
    if (const char* e = getenv("PA_FUSE_DEC")) m->fuse_dec = e[0] != '0';
    int rc = init_base(m, cfg->device, hip_stream);
    const int H = cfg->hidden_size;
    for (int stage = 0; stage < 2 && rc == PA_OK; ++stage)
        for (int l = 0; l < cfg->gru_layers && rc == PA_OK; ++l) {
            auto& vec = stage == 0 ? m->enc : m->dec;
            vec.emplace_back();
            const int K = (stage == 0 && l == 0) ? cfg->image_features : 2 * H;
            rc = build_rec_layer(m, sd, stage == 0 ? "gru_encoder" : "gru_decoder", l, 3, H, K, vec.back(), false,
                                 m->split_gemm);
        }
    if (rc == PA_OK) rc = build_linear(m, sd, "dense1", 2 * H, cfg->num_classes, m->dense);
    if (const char* e = getenv("PA_FUSE_HEAD")) m->fuse_head = e[0] != '0';
    if (const char* e = getenv("PA_POLISH_SMALL_MAX")) m->small_max = std::max(0, atoi(e));
    if (rc == PA_OK && m->split_rec && H == 128 && cfg->num_classes <= 5) {
        std::string err;
        const float* w = sd.get("dense1.weight", (int64_t)cfg->num_classes * 2 * H, err);
        std::vector<uint32_t> packed(pa::dense_head_h2_words(H));
        pa::pack_dense_head_h2(w, cfg->num_classes, H, packed.data());
        m->dense_h2 = m->new_buf();
        rc = m->dense_h2->ensure(packed.size() * sizeof(uint32_t));
        if (rc == PA_OK && hipMemcpy(m->dense_h2->p, packed.data(), packed.size() * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess)
            rc = fail(PA_ERR_HIP, "upload of the packed dense1 fragments failed");
    }
    if (rc != PA_OK) {
        delete m;
        return rc;
    }
    m->part = m->new_buf();
    m->xp = m->new_buf(); m->y1 = m->new_buf(); m->y2 = m->new_buf();
    m->hid_a = m->new_buf(); m->hid_b = m->new_buf(); m->acc = m->new_buf();
    for (int k = 0; k < 2; ++k) {
        m->stage_in[k] = m->new_buf(); m->stage_lab[k] = m->new_buf(); m->stage_ph[k] = m->new_buf();
        m->stage_acc[k] = m->new_buf();
    }
    *out = m;
    return PA_OK;
}

void pa_polish_destroy(pa_polish_model* m) {
    if (!m) return;
    (void)hipSetDevice(m->device);
    if (m->stream) (void)hipStreamSynchronize(m->stream);
    delete m;
}

int pa_polish_forward_device(pa_polish_model* m, const float* x, const float* hidden, int64_t n, int32_t T,
                             float* logits, float* hidden_out) {
    if (!m || m->magic != 0x50414d44) return fail(PA_ERR_INVALID, "bad model handle");
    if (n < 0 || T <= 0 || (n > 0 && (!x || !logits))) return fail(PA_ERR_INVALID, "bad argument");
    if (n == 0) return PA_OK;
    HIP_TRY(hipSetDevice(m->device));
    const int H = m->cfg.hidden_size, L = m->cfg.gru_layers, F = m->cfg.image_features, C = m->cfg.num_classes;
    const size_t hbytes = (size_t)2 * L * H * sizeof(float);
    for (int64_t off = 0; off < n; off += m->cfg.max_chunk) {
        const int64_t c = std::min<int64_t>(m->cfg.max_chunk, n - off);
        if (int rc = polish_ensure(m, c, T)) return rc;
        // user hidden buffers are exactly [c, 2L, H]; the kernels want MT-padded rows -> stage
        const float* hin = nullptr;
        if (hidden) {
            HIP_TRY(hipMemcpyAsync(m->hid_b->p, hidden + (size_t)off * 2 * L * H, c * hbytes,
                                   hipMemcpyDeviceToDevice, m->stream));
            hin = m->hid_b->f();
        }
        float* y = nullptr;
        if (int rc = polish_window(m, pa::A_F32_SCALAR, x + (size_t)off * T * F, F, 0, 0, hin, m->hid_b->f(), c,
                                   T, &y))
            return rc;
        if (m->y_h2)
            LAUNCH_TRY(m, "cvt_f32", 0.0, pa::launch_h2_to_f32(y, c * T, 2 * H, 2 * H, m->stream));
        LAUNCH_TRY(m, "dense_logits", 2.0 * c * T * 2 * H * C,
                   pa::launch_dense_small(2, y, 2 * H, m->dense.w->f(), m->dense.b->f(),
                                          logits + (size_t)off * T * C, nullptr, (int)(c * T), 2 * H, C, 1, 1, 0,
                                          m->stream));
        if (hidden_out)
            HIP_TRY(hipMemcpyAsync(hidden_out + (size_t)off * 2 * L * H, m->hid_b->p, c * hbytes,
                                   hipMemcpyDeviceToDevice, m->stream));
    }
    return PA_OK;
}

int pa_polish_predict_device(pa_polish_model* m, const uint8_t* images, int64_t n, uint8_t* labels,
                             uint8_t* phred, float* acc_out) {
    if (!m || m->magic != 0x50414d44) return fail(PA_ERR_INVALID, "bad model handle");
    if (n < 0 || (n > 0 && (!images || !labels || !phred))) return fail(PA_ERR_INVALID, "null buffer");
    if (n == 0) return PA_OK;
    HIP_TRY(hipSetDevice(m->device));
    const int H = m->cfg.hidden_size, F = m->cfg.image_features, C = m->cfg.num_classes;
    const int S = m->cfg.seq_length, T = m->cfg.window;
    for (int64_t off = 0; off < n; off += m->cfg.max_chunk) {
        const int64_t c = std::min<int64_t>(m->cfg.max_chunk, n - off);
        if (int rc = polish_ensure(m, c, T)) return rc;
        float* acc = acc_out ? acc_out + (size_t)off * S * C : nullptr;
        if (!acc) {
            if (int rc = m->acc->ensure((size_t)c * S * C * sizeof(float))) return rc;
            acc = m->acc->f();
        }
        HIP_TRY(hipMemsetAsync(acc, 0, (size_t)c * S * C * sizeof(float), m->stream));
        const uint8_t* img = images + (size_t)off * S * F;
        bool first = true;
        // predict_distributed_cpu.py:50-53: i = 0, jump, ... while i + window <= seq_length
        for (int i = 0; i + T <= S; i += m->cfg.jump) {
            float* y = nullptr;
            bool head = true;
            if (int rc = polish_window(m, pa::A_U8, img + (size_t)i * F, F, T, (int64_t)S * F,
                                       first ? nullptr : m->hid_b->f(), m->hid_b->f(), c, T, &y, &head))
                return rc;
            first = false;
            if (head)
                LAUNCH_TRY(m, "head_combine_acc", 0.0,
                           pa::launch_polish_combine(m->part->f(), m->dense.b->f(), acc, (int)c, T, C, S, i, m->stream));
            else if (m->y_h2 && 2 * H == 256 && C <= 5)
                LAUNCH_TRY(m, "dense_softmax_acc", 2.0 * c * T * 2 * H * C,
                           pa::launch_polish_dense_acc_h2(y, 2 * H, m->dense.w->f(), m->dense.b->f(), acc, (int)(c * T),
                                                          2 * H, C, T, S, i, m->stream));
            else {
                if (m->y_h2)
                    LAUNCH_TRY(m, "cvt_f32", 0.0, pa::launch_h2_to_f32(y, c * T, 2 * H, 2 * H, m->stream));
                LAUNCH_TRY(m, "dense_softmax_acc", 2.0 * c * T * 2 * H * C,
                           pa::launch_dense_small(1, y, 2 * H, m->dense.w->f(), m->dense.b->f(), acc, nullptr,
                                                  (int)(c * T), 2 * H, C, T, S, i, m->stream));
            }
        }
        LAUNCH_TRY(m, "polish_finalize", 0.0,
                   pa::launch_polish_finalize(acc, labels + (size_t)off * S, phred + (size_t)off * S, c, S, C,
                                              m->cfg.overlap, m->stream));
    }
    return PA_OK;
}

// One or several host blocks as ONE sequence of device passes: the polish kernels give a workgroup 128 chunks of one
// direction and walk their time steps in sequence, so a pass costs about the same for 2 048 chunks as for 16 384 -- callers
// that hold their chunks in several buffers (the reader lanes' slots, pepper_amd/hostpipe.py) hand them over together.
static int polish_predict_host_run(pa_polish_model* m, int32_t n_parts, const uint8_t* const* images, const int64_t* counts,
                                   uint8_t* const* labels, uint8_t* const* phred, float* acc) {
    int64_t n = 0;
    for (int32_t p = 0; p < n_parts; ++p) {
        if (counts[p] < 0 || (counts[p] > 0 && (!images[p] || !labels[p] || !phred[p]))) return fail(PA_ERR_INVALID, "null buffer");
        n += counts[p];
    }
    if (n == 0) return PA_OK;
    HIP_TRY(hipSetDevice(m->device));
    if (int rc = m->pipe_init()) return rc;
    const size_t S = m->cfg.seq_length, F = m->cfg.image_features, C = m->cfg.num_classes;
    const int64_t chunk = m->cfg.max_chunk;
    const size_t cap = (size_t)std::min<int64_t>(chunk, n);
    auto& pp = m->pipe;
    // the pieces of the parts that make up units [off, off + c): f(part, first unit within the part, first unit within the pass, units)
    auto pieces = [&](int64_t off, int64_t c, auto&& f) -> int {
        int64_t start = 0;
        for (int32_t p = 0; p < n_parts; ++p) {
            const int64_t lo = std::max(off, start), hi = std::min(off + c, start + counts[p]);
            if (lo < hi)
                if (int rc = f(p, lo - start, lo - off, hi - lo)) return rc;
            start += counts[p];
        }
        return PA_OK;
    };
    int64_t i = 0;
    for (int64_t off = 0; off < n; off += chunk, ++i) {
        const int64_t c = std::min<int64_t>(chunk, n - off);
        const int k = (int)(i & 1);
        if (int rc = m->stage_in[k]->ensure(cap * S * F)) return rc;
        if (int rc = m->stage_lab[k]->ensure(cap * S)) return rc;
        if (int rc = m->stage_ph[k]->ensure(cap * S)) return rc;
        if (acc)
            if (int rc = m->stage_acc[k]->ensure(cap * S * C * sizeof(float))) return rc;
        if (i >= 2) HIP_TRY(hipStreamWaitEvent(pp.h2d, pp.in_free[k], 0));
        if (int rc = pieces(off, c, [&](int32_t p, int64_t in_part, int64_t in_pass, int64_t units) -> int {
                HIP_TRY(hipMemcpyAsync(static_cast<uint8_t*>(m->stage_in[k]->p) + (size_t)in_pass * S * F,
                                       images[p] + (size_t)in_part * S * F, (size_t)units * S * F, hipMemcpyHostToDevice, pp.h2d));
                return PA_OK;
            }))
            return rc;
        HIP_TRY(hipEventRecord(pp.in_ready[k], pp.h2d));
        HIP_TRY(hipStreamWaitEvent(m->stream, pp.in_ready[k], 0));
        if (i >= 2) HIP_TRY(hipStreamWaitEvent(m->stream, pp.out_free[k], 0));
        if (int rc = pa_polish_predict_device(m, static_cast<const uint8_t*>(m->stage_in[k]->p), c,
                                              static_cast<uint8_t*>(m->stage_lab[k]->p),
                                              static_cast<uint8_t*>(m->stage_ph[k]->p), acc ? m->stage_acc[k]->f() : nullptr))
            return rc;
        HIP_TRY(hipEventRecord(pp.in_free[k], m->stream));
        HIP_TRY(hipEventRecord(pp.out_ready[k], m->stream));
        HIP_TRY(hipStreamWaitEvent(pp.d2h, pp.out_ready[k], 0));
        if (int rc = pieces(off, c, [&](int32_t p, int64_t in_part, int64_t in_pass, int64_t units) -> int {
                HIP_TRY(hipMemcpyAsync(labels[p] + (size_t)in_part * S, static_cast<uint8_t*>(m->stage_lab[k]->p) + (size_t)in_pass * S,
                                       (size_t)units * S, hipMemcpyDeviceToHost, pp.d2h));
                HIP_TRY(hipMemcpyAsync(phred[p] + (size_t)in_part * S, static_cast<uint8_t*>(m->stage_ph[k]->p) + (size_t)in_pass * S,
                                       (size_t)units * S, hipMemcpyDeviceToHost, pp.d2h));
                return PA_OK;
            }))
            return rc;
        if (acc)     // single-part callers only (pa_polish_predict_host)
            HIP_TRY(hipMemcpyAsync(acc + (size_t)off * S * C, m->stage_acc[k]->p, (size_t)c * S * C * sizeof(float),
                                   hipMemcpyDeviceToHost, pp.d2h));
        HIP_TRY(hipEventRecord(pp.out_free[k], pp.d2h));
    }
    HIP_TRY(hipStreamSynchronize(pp.d2h));
    HIP_TRY(hipStreamSynchronize(m->stream));
    return PA_OK;
}

static int polish_predict_host_impl(pa_polish_model* m, int32_t n_parts, const uint8_t* const* images, const int64_t* counts,
                                    uint8_t* const* labels, uint8_t* const* phred, float* acc) {
    const int rc = polish_predict_host_run(m, n_parts, images, counts, labels, phred, acc);
    if (rc != PA_OK) m->quiesce();
    return rc;
}

int pa_polish_predict_host(pa_polish_model* m, const uint8_t* images, int64_t n, uint8_t* labels,
                           uint8_t* phred, float* acc) {
    if (!m || m->magic != 0x50414d44) return fail(PA_ERR_INVALID, "bad model handle");
    if (n < 0 || (n > 0 && (!images || !labels || !phred))) return fail(PA_ERR_INVALID, "null buffer");
    return polish_predict_host_impl(m, 1, &images, &n, &labels, &phred, acc);
}

int pa_polish_predict_host_parts(pa_polish_model* m, int32_t n_parts, const uint8_t* const* images, const int64_t* counts,
                                 uint8_t* const* labels, uint8_t* const* phred) {
    if (!m || m->magic != 0x50414d44) return fail(PA_ERR_INVALID, "bad model handle");
    if (n_parts < 0 || (n_parts > 0 && (!images || !counts || !labels || !phred))) return fail(PA_ERR_INVALID, "bad argument");
    return polish_predict_host_impl(m, n_parts, images, counts, labels, phred, nullptr);
}

// ---- profiler / sync -----------------------------------------------------------------------------
static ModelBase* as_base(void* model) {
    // Both handle structs derive from ModelBase as their only (polymorphic) base, so a handle's
    // address is its ModelBase address; the magic word guards against foreign pointers.
    auto* b = reinterpret_cast<ModelBase*>(model);
    if (!model || b->magic != 0x50414d44) return nullptr;
    return b;
}

int pa_profile_enable(void* model, int32_t on) {
    ModelBase* b = as_base(model);
    if (!b) return fail(PA_ERR_INVALID, "bad model handle");
    if (int rc = b->drain()) return rc;
    b->profiling = on != 0;
    b->labels.clear();
    b->total_ms.clear();
    b->total_flops.clear();
    b->launches.clear();
    return PA_OK;
}

int pa_profile_count(void* model) {
    ModelBase* b = as_base(model);
    if (!b) return -1;
    if (b->drain() != PA_OK) return -1;
    return (int)b->labels.size();
}

int pa_profile_get(void* model, int32_t idx, char* label, int32_t label_cap, double* total_ms,
                   int64_t* launches, double* flops) {
    ModelBase* b = as_base(model);
    if (!b) return fail(PA_ERR_INVALID, "bad model handle");
    if (int rc = b->drain()) return rc;
    if (idx < 0 || idx >= (int)b->labels.size()) return fail(PA_ERR_INVALID, "profile index out of range");
    if (label && label_cap > 0) {
        std::strncpy(label, b->labels[idx].c_str(), label_cap - 1);
        label[label_cap - 1] = 0;
    }
    if (total_ms) *total_ms = b->total_ms[idx];
    if (launches) *launches = b->launches[idx];
    if (flops) *flops = b->total_flops[idx];
    return PA_OK;
}

int pa_host_register(void* ptr, int64_t bytes) {
    if (!ptr || bytes <= 0) return fail(PA_ERR_INVALID, "bad buffer");
    // portable: the caller may be a helper thread whose current device is not the model's (HIP's current device is per thread)
    HIP_TRY(hipHostRegister(ptr, (size_t)bytes, hipHostRegisterPortable));
    return PA_OK;
}

int pa_host_unregister(void* ptr) {
    if (!ptr) return fail(PA_ERR_INVALID, "bad buffer");
    HIP_TRY(hipHostUnregister(ptr));
    return PA_OK;
}

int pa_synchronize(void* model) {
    ModelBase* b = as_base(model);
    if (!b) return fail(PA_ERR_INVALID, "bad model handle");
    HIP_TRY(hipSetDevice(b->device));
    HIP_TRY(hipStreamSynchronize(b->stream));
    return PA_OK;
}

}  // extern "C"
