// t3mi_hift_*: host side of the MI355X-native HiFT vocoder (include/t3mi.h; SURVEY §8 f4, third piece; R/hifigan.R:337-627 as
// create_s3gen_vocoder configures it, :788-817).  Weights come from the reference's unchanged s3gen safetensors file under
// "mel2wav." through load_hifigan_weights' key map (R/hifigan.R:640-777): weight-normalised convolutions are folded
// (w = g v / ||v||) at load and re-laid [out][tap][in] (K padded to the GEMM's 64-element step) for the window GEMMs.
#include "t3mi.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "cfm_kernels.hpp"
#include "hift_kernels.hpp"
#include "kernels.hpp"
#include "safetensors.hpp"

int t3mi_set_error(int code, const std::string& msg);

namespace {

int fail(int code, const std::string& msg) { return t3mi_set_error(code, msg); }

#define HIPCHK(expr)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(T3MI_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));     \
    } while (0)

struct Buf {
    void* p = nullptr;
    size_t bytes = 0;
    ~Buf() { if (p) (void)hipFree(p); }
    hipError_t need(size_t n) {
        if (n <= bytes) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
        hipError_t e = hipMalloc(&p, n);
        if (e == hipSuccess) bytes = n;
        return e;
    }
    float* f() const { return static_cast<float*>(p); }
};

// create_s3gen_vocoder's configuration (R/hifigan.R:796-814)
constexpr int kUps[3][2] = {{8, 16}, {5, 11}, {3, 7}};   // (rate, kernel)
constexpr int kResK[3] = {3, 7, 11};
constexpr int kSrcK[3] = {7, 7, 11};
constexpr int kDil[3] = {1, 3, 5};
constexpr int kFft = 16, kHop = 4, kSpecCh = kFft + 2, kSpecLd = 20;   // STFT rows padded to 20 floats (16-byte pieces)
constexpr int kTotalUp = 8 * 5 * 3 * kHop;                             // 480 samples per mel frame
constexpr float kSr = 24000.f, kVoiced = 10.f, kLimit = 0.99f;

struct Conv {          // conv as a window GEMM: W [N][Kp] on device, taps k, input channels C (row stride of the input may be larger)
    const float* w = nullptr;
    const float* b = nullptr;
    int N = 0, C = 0, k = 0, Kp = 0;
};
struct ResUnit {
    Conv c1, c2;
    const float *a1 = nullptr, *a2 = nullptr;   // Snake alphas
};
struct ResBlock {
    ResUnit u[3];
    int k = 0;
};

}  // namespace

struct t3mi_hift {
    int device = 0, mel = 0, base = 0, f0_ch = 0, n_harm = 0;
    hipStream_t stream = nullptr;
    Buf wbuf;
    int64_t weight_bytes = 0;
    Conv f0c[5], f0_cls, conv_pre, ups[3], src_down[3], conv_post;
    const float *src_lw = nullptr, *src_lb = nullptr;
    ResBlock res[9], src_res[3];
    // workspace
    Buf melr, f0a, f0b, f0v, phase, noise, base_d, src, stft, col, x, xr[3], t1, upb, si, post, spec, audio;
    int ws_T = 0;
};

namespace {

struct Loader {
    const t3mi::SafeTensorsFile& st;
    std::string prefix;
    std::vector<float> host;
    std::string err;

    const t3mi::StTensor* find(const std::string& key, size_t nd) {
        const t3mi::StTensor* t = st.find(prefix + key);
        if (!t) { err = "checkpoint has no tensor '" + prefix + key + "'"; return nullptr; }
        if (t->shape.size() != nd) { err = "tensor '" + prefix + key + "' has an unexpected rank"; return nullptr; }
        for (int64_t s : t->shape)
            if (s <= 0 || s > (1 << 24)) { err = "tensor '" + prefix + key + "': implausible extent"; return nullptr; }
        if (t->dtype != t3mi::StDtype::F32 && t->dtype != t3mi::StDtype::F16 && t->dtype != t3mi::StDtype::BF16) {
            err = "tensor '" + prefix + key + "' is not a floating-point tensor";
            return nullptr;
        }
        return t;
    }
    bool fetch(const t3mi::StTensor* t, std::vector<float>& out) {
        out.resize((size_t)t->numel());
        if (t->dtype == t3mi::StDtype::F32) { std::memcpy(out.data(), t->data, out.size() * 4); return true; }
        return t3mi::SafeTensorsFile::to_f32(*t, out.data(), err);
    }
    long long append(const std::vector<float>& v) {
        const size_t off = host.size();
        host.resize(off + ((v.size() + 3) & ~size_t(3)));
        std::memcpy(host.data() + off, v.data(), v.size() * 4);
        return (long long)off;
    }
    // a plain vector tensor of n elements
    long long vec(const std::string& key, int n) {
        const t3mi::StTensor* t = find(key, 1);
        if (!t) return -1;
        if (t->shape[0] != n) { err = "tensor '" + prefix + key + "' has an unexpected shape"; return -1; }
        std::vector<float> v;
        if (!fetch(t, v)) return -1;
        return append(v);
    }
    // conv weight [A][B][k] from `key`.weight, or folded from the weight-norm pair (R/hifigan.R:649-667): w = g v / ||v||, the
    // norm over all but the first axis.  transposed: the tensor is [in][out][k] (conv_transpose1d, :715-722).
    // Stored [out][k][c_ld] (c_ld >= in, zero columns after `in`), rows padded with zeros to Kp.
    long long conv(const std::string& base, bool transposed, int c_ld, Conv& c) {
        const t3mi::StTensor* tw = st.find(prefix + base + ".weight");
        std::vector<float> w;
        std::vector<int64_t> shape;
        if (tw) {
            if (!(tw = find(base + ".weight", 3)) || !fetch(tw, w)) return -1;
            shape = tw->shape;
        } else {
            const t3mi::StTensor* tg = find(base + ".parametrizations.weight.original0", 3);
            const t3mi::StTensor* tv = tg ? find(base + ".parametrizations.weight.original1", 3) : nullptr;
            std::vector<float> g;
            if (!tv || !fetch(tg, g) || !fetch(tv, w)) return -1;
            shape = tv->shape;
            if (tg->shape[0] != shape[0] || tg->numel() != shape[0]) { err = "weight-norm pair of '" + prefix + base + "' does not match"; return -1; }
            const size_t per = (size_t)(shape[1] * shape[2]);
            for (int64_t a = 0; a < shape[0]; ++a) {
                float ss = 0.f;
                for (size_t i = 0; i < per; ++i) ss += w[(size_t)a * per + i] * w[(size_t)a * per + i];
                const float nrm = sqrtf(ss);
                for (size_t i = 0; i < per; ++i) w[(size_t)a * per + i] = g[(size_t)a] * w[(size_t)a * per + i] / nrm;
            }
        }
        const int A = (int)shape[0], B = (int)shape[1], k = (int)shape[2];
        const int O = transposed ? B : A, I = transposed ? A : B;
        if (c_ld < I) c_ld = I;
        c.N = O; c.C = c_ld; c.k = k;
        c.Kp = (k * c_ld + 63) & ~63;
        std::vector<float> out((size_t)O * c.Kp, 0.f);
        for (int o = 0; o < O; ++o)
            for (int i = 0; i < I; ++i)
                for (int j = 0; j < k; ++j) {
                    const float v = transposed ? w[((size_t)i * O + o) * k + j] : w[((size_t)o * I + i) * k + j];
                    out[(size_t)o * c.Kp + (size_t)j * c_ld + i] = v;
                }
        return append(out);
    }
};

struct Pending {
    const float** slot;
    long long off;
};

int load_hift(t3mi_hift* m, const t3mi::SafeTensorsFile& st, const std::string& prefix) {
    Loader L{st, prefix, {}, ""};
    std::vector<Pending> pend;
    bool ok = true;
    auto reg = [&](const float** slot, long long off) {
        if (off < 0) ok = false;
        pend.push_back({slot, off});
    };
    auto conv = [&](Conv& c, const std::string& base, bool transposed = false, int c_ld = 0) {
        if (!ok) return;
        reg(&c.w, L.conv(base, transposed, c_ld, c));
        if (ok) reg(&c.b, L.vec(base + ".bias", c.N));
    };
    auto resblock = [&](ResBlock& rb, const std::string& p, int k) {
        rb.k = k;
        for (int j = 0; j < 3 && ok; ++j) {
            conv(rb.u[j].c1, p + ".convs1." + std::to_string(j));
            conv(rb.u[j].c2, p + ".convs2." + std::to_string(j));
            if (ok) reg(&rb.u[j].a1, L.vec(p + ".activations1." + std::to_string(j) + ".alpha", rb.u[j].c1.N));
            if (ok) reg(&rb.u[j].a2, L.vec(p + ".activations2." + std::to_string(j) + ".alpha", rb.u[j].c1.N));
            if (ok && (rb.u[j].c1.k != k || rb.u[j].c2.k != k || rb.u[j].c1.N != rb.u[j].c1.C || rb.u[j].c2.N != rb.u[j].c1.N)) {
                L.err = "residual block '" + prefix + p + "' does not have the shapes of create_s3gen_vocoder (R/hifigan.R:808-811)";
                ok = false;
            }
        }
    };
    for (int i = 0; i < 5; ++i) conv(m->f0c[i], "f0_predictor.condnet." + std::to_string(2 * i));
    if (ok) {
        // classifier: Linear(c, 1) as a 1-tap conv
        const t3mi::StTensor* t = L.find("f0_predictor.classifier.weight", 2);
        std::vector<float> w;
        if (!t || !L.fetch(t, w) || t->shape[0] != 1) ok = false;
        else {
            m->f0_cls.N = 1; m->f0_cls.C = (int)t->shape[1]; m->f0_cls.k = 1; m->f0_cls.Kp = (m->f0_cls.C + 63) & ~63;
            w.resize((size_t)m->f0_cls.Kp, 0.f);
            reg(&m->f0_cls.w, L.append(w));
            reg(&m->f0_cls.b, L.vec("f0_predictor.classifier.bias", 1));
        }
    }
    if (ok) {
        const t3mi::StTensor* t = L.find("m_source.l_linear.weight", 2);
        std::vector<float> w;
        if (!t || !L.fetch(t, w) || t->shape[0] != 1) ok = false;
        else {
            m->n_harm = (int)t->shape[1];
            reg(&m->src_lw, L.append(w));
            reg(&m->src_lb, L.vec("m_source.l_linear.bias", 1));
        }
    }
    conv(m->conv_pre, "conv_pre");
    for (int i = 0; i < 3 && ok; ++i) {
        conv(m->ups[i], "ups." + std::to_string(i), true);
        conv(m->src_down[i], "source_downs." + std::to_string(i), false, kSpecLd);
        if (ok) resblock(m->src_res[i], "source_resblocks." + std::to_string(i), kSrcK[i]);
        for (int j = 0; j < 3 && ok; ++j) resblock(m->res[i * 3 + j], "resblocks." + std::to_string(i * 3 + j), kResK[j]);
    }
    conv(m->conv_post, "conv_post");
    if (!ok) return fail(T3MI_ERR_IO, L.err.empty() ? "HiFT checkpoint: malformed tensor" : L.err);
    // the configuration of create_s3gen_vocoder, checked against the file
    m->mel = m->conv_pre.C;
    m->base = m->conv_pre.N;
    m->f0_ch = m->f0c[0].N;
    bool shape_ok = m->conv_pre.k == 7 && m->conv_post.k == 7 && m->conv_post.N == kSpecCh && m->f0c[0].C == m->mel && m->mel % 4 == 0 &&
                    m->base % 32 == 0 && m->n_harm >= 1 && m->n_harm <= 64 && m->f0_cls.C == m->f0_ch;
    for (int i = 0; i < 5 && shape_ok; ++i) shape_ok = m->f0c[i].k == 3 && m->f0c[i].N == m->f0_ch && (i == 0 || m->f0c[i].C == m->f0_ch);
    int rate_prod = 1;
    for (int i = 0; i < 3 && shape_ok; ++i) {
        rate_prod *= kUps[i][0];
        const int du = 120 / rate_prod;
        shape_ok = m->ups[i].k == kUps[i][1] && m->ups[i].C == (m->base >> i) && m->ups[i].N == (m->base >> (i + 1)) &&
                   m->src_down[i].N == m->ups[i].N && m->src_down[i].k == (du == 1 ? 1 : 2 * du) && m->src_res[i].u[0].c1.N == m->ups[i].N;
        for (int j = 0; j < 3 && shape_ok; ++j) shape_ok = m->res[i * 3 + j].u[0].c1.N == m->ups[i].N;
    }
    if (shape_ok) shape_ok = m->conv_post.C == (m->base >> 3);
    if (!shape_ok) return fail(T3MI_ERR_IO, "HiFT checkpoint does not have the shapes of create_s3gen_vocoder (R/hifigan.R:796-814)");
    m->weight_bytes = (int64_t)L.host.size() * 4;
    HIPCHK(m->wbuf.need(L.host.size() * 4));
    HIPCHK(hipMemcpy(m->wbuf.p, L.host.data(), L.host.size() * 4, hipMemcpyHostToDevice));
    for (const Pending& p : pend) *p.slot = m->wbuf.f() + p.off;
    return T3MI_OK;
}

int ensure_ws(t3mi_hift* m, int T) {
    if (T <= m->ws_T) return T3MI_OK;
    HIPCHK(hipStreamSynchronize(m->stream));
    const size_t Tw = (size_t)T * kTotalUp, frames = Tw / kHop + 1;     // 120 T + 1 rows at the last stage
    size_t col = 0, xmax = 0;
    // window buffer: the largest L_out * Kp over every convolution of the pass
    col = std::max(col, (size_t)T * std::max(m->conv_pre.Kp, m->f0c[1].Kp));
    size_t rows = (size_t)T;
    for (int i = 0; i < 3; ++i) {
        rows = i == 2 ? frames : rows * kUps[i][0];
        int kp = std::max(m->ups[i].Kp, m->src_down[i].Kp);
        kp = std::max(kp, m->src_res[i].u[0].c1.Kp);
        for (int j = 0; j < 3; ++j) kp = std::max(kp, m->res[i * 3 + j].u[0].c1.Kp);
        col = std::max(col, rows * (size_t)kp);
        xmax = std::max(xmax, rows * (size_t)m->ups[i].N);
    }
    col = std::max(col, frames * (size_t)m->conv_post.Kp);
    xmax = std::max(xmax, (size_t)T * std::max(m->base, m->f0_ch));
    HIPCHK(m->melr.need((size_t)T * m->mel * 4));
    HIPCHK(m->f0a.need((size_t)T * m->f0_ch * 4));
    HIPCHK(m->f0b.need((size_t)T * m->f0_ch * 4));
    HIPCHK(m->f0v.need((size_t)T * 4));
    HIPCHK(m->phase.need((size_t)m->n_harm * 4));
    HIPCHK(m->noise.need((size_t)m->n_harm * Tw * 4));
    HIPCHK(m->base_d.need((size_t)m->n_harm * T * 8));
    HIPCHK(m->src.need(Tw * 4));
    HIPCHK(m->stft.need(frames * kSpecLd * 4));
    HIPCHK(m->col.need(col * 4));
    HIPCHK(m->x.need((xmax + 1024) * 4));
    for (int r = 0; r < 3; ++r) HIPCHK(m->xr[r].need(xmax * 4));
    HIPCHK(m->t1.need(xmax * 4));
    HIPCHK(m->upb.need((xmax + 1024) * 4));
    HIPCHK(m->si.need(xmax * 4));
    HIPCHK(m->post.need(frames * kSpecCh * 4));
    HIPCHK(m->spec.need(frames * kSpecCh * 4));
    HIPCHK(m->audio.need(Tw * 4));
    m->ws_T = T;
    return T3MI_OK;
}

struct WinSpec {
    int stride = 1, dil = 1, pad = 0, up = 0, act = 0;
    float slope = 0.f;
    const float* alpha = nullptr;
};

// y[L_out][N] = conv(act(x)) (+ resid): window rows -> exact-fp32 MFMA GEMM
int conv_rows(t3mi_hift* m, const Conv& c, const float* x, int L_in, int ldx, int L_out, const WinSpec& ws, float* y, const float* resid) {
    t3k::HiftWindowArgs a;
    a.x = x; a.L_in = L_in; a.C = c.C; a.ldx = ldx; a.col = m->col.f(); a.L_out = L_out; a.k = c.k; a.Kp = c.Kp;
    a.stride = ws.stride; a.dil = ws.dil; a.pad = ws.pad; a.up = ws.up; a.act = ws.act; a.slope = ws.slope; a.alpha = ws.alpha;
    HIPCHK(t3k::launch_hift_window(a, m->stream));
    const int epi = resid ? t3k::EPI_RESID : t3k::EPI_STORE;
    if (L_out >= 16 && t3k::dense_gemm_supported(L_out, c.N, c.Kp, epi)) {   // everything but conv_post (N = 18) and the F0 classifier
        t3k::DenseGemmArgs d;
        d.A = m->col.f(); d.lda = c.Kp; d.W = c.w; d.bias = c.b; d.M = L_out; d.N = c.N; d.K = c.Kp; d.epi = epi;
        d.y = y; d.ldy = c.N; d.resid = resid;
        HIPCHK(t3k::launch_dense_gemm(d, m->stream));
        return T3MI_OK;
    }
    t3k::GemmArgs g;
    g.W = c.w; g.A = m->col.f(); g.M = L_out; g.N = c.N; g.K = c.Kp; g.bias = c.b;
    g.y = y; g.resid = resid; g.ldy = c.N;
    HIPCHK(t3k::launch_gemm(g, epi, 0, m->stream));
    return T3MI_OK;
}

// hifigan_resblock$forward (R/hifigan.R:129-139): out = the block applied to x [L][C]; x itself stays (three blocks share it)
int resblock(t3mi_hift* m, const ResBlock& rb, const float* x, int L, float* out) {
    const int C = rb.u[0].c1.N;
    const float* cur = x;
    for (int j = 0; j < 3; ++j) {
        WinSpec w1;
        w1.dil = kDil[j]; w1.pad = (rb.k * kDil[j] - kDil[j]) / 2; w1.act = 2; w1.alpha = rb.u[j].a1;
        int rc = conv_rows(m, rb.u[j].c1, cur, L, C, L, w1, m->t1.f(), nullptr);
        if (rc) return rc;
        WinSpec w2;
        w2.pad = (rb.k - 1) / 2; w2.act = 2; w2.alpha = rb.u[j].a2;
        if ((rc = conv_rows(m, rb.u[j].c2, m->t1.f(), L, C, L, w2, out, cur))) return rc;   // x <- xt + x
        cur = out;
    }
    return T3MI_OK;
}

}  // namespace

extern "C" {

int t3mi_hift_open(const char* path, const char* prefix, int device, t3mi_hift** out) {
    if (!path || !out) return fail(T3MI_ERR_ARG, "t3mi_hift_open: null argument");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(T3MI_ERR_HIP, "t3mi_hift_open: no HIP device visible (the MI355X path has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(T3MI_ERR_ARG, "t3mi_hift_open: device index out of range");
    HIPCHK(hipSetDevice(device));
    t3mi::SafeTensorsFile st;
    std::string err;
    if (!st.open(path, err)) return fail(T3MI_ERR_IO, err);
    std::unique_ptr<t3mi_hift> m(new t3mi_hift());
    m->device = device;
    int rc = load_hift(m.get(), st, prefix && prefix[0] ? prefix : "mel2wav.");
    if (rc) return rc;
    HIPCHK(hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking));
    *out = m.release();
    return T3MI_OK;
}

int t3mi_hift_close(t3mi_hift* m) {
    if (!m) return T3MI_OK;
    (void)hipSetDevice(m->device);
    if (m->stream) {
        (void)hipStreamSynchronize(m->stream);
        (void)hipStreamDestroy(m->stream);
    }
    delete m;
    return T3MI_OK;
}

int t3mi_hift_get_info(const t3mi_hift* m, t3mi_hift_info* info) {
    if (!m || !info) return fail(T3MI_ERR_ARG, "t3mi_hift_get_info: null argument");
    info->mel = m->mel; info->base_channels = m->base; info->f0_channels = m->f0_ch; info->n_harmonics = m->n_harm;
    info->samples_per_frame = kTotalUp; info->sampling_rate = (int32_t)kSr; info->device = m->device; info->reserved_ = 0;
    info->weight_bytes = m->weight_bytes;
    return T3MI_OK;
}

int t3mi_hift_inference(t3mi_hift* m, int T, const float* mel, const float* phase, const float* noise, const float* cache_source,
                        int cache_len, float* audio, float* source_out, float* f0_out, float* device_ms) {
    if (!m) return fail(T3MI_ERR_ARG, "null HiFT handle");
    if (!mel || !audio) return fail(T3MI_ERR_ARG, "t3mi_hift_inference: null argument");
    if (T < 2 || T > (1 << 15)) return fail(T3MI_ERR_ARG, "t3mi_hift_inference: the mel must hold 2 .. 32768 frames");
    if (cache_len < 0 || (cache_len > 0 && !cache_source) || (long long)cache_len > (long long)T * kTotalUp)
        return fail(T3MI_ERR_ARG, "t3mi_hift_inference: cache_source must hold cache_len <= 480 T samples");
    HIPCHK(hipSetDevice(m->device));
    int rc = ensure_ws(m, T);
    if (rc) return rc;
    hipStream_t st = m->stream;
    const int Tw = T * kTotalUp, frames = Tw / kHop + 1;
    // mel (mel, T) -> rows [T][mel]
    std::vector<float> mr((size_t)T * m->mel), ph((size_t)m->n_harm, 0.f);
    for (int c = 0; c < m->mel; ++c)
        for (int t = 0; t < T; ++t) mr[(size_t)t * m->mel + c] = mel[(size_t)c * T + t];
    if (phase) std::memcpy(ph.data(), phase, ph.size() * 4);
    HIPCHK(hipMemcpyAsync(m->melr.p, mr.data(), mr.size() * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(m->phase.p, ph.data(), ph.size() * 4, hipMemcpyHostToDevice, st));
    if (noise) HIPCHK(hipMemcpyAsync(m->noise.p, noise, (size_t)m->n_harm * Tw * 4, hipMemcpyHostToDevice, st));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (device_ms) {
        HIPCHK(hipEventCreate(&e0));
        HIPCHK(hipEventCreate(&e1));
        HIPCHK(hipEventRecord(e0, st));
    }
    auto done = [&](int code) {
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        if (code) (void)hipStreamSynchronize(st);
        return code;
    };
#define STEP(expr) do { if ((rc = (expr))) return done(rc); } while (0)
#define HSTEP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return done(fail(T3MI_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_))); } while (0)
    // ---- conv_rnn_f0_predictor (R/hifigan.R:174-180): 5 x (conv k3 pad 1, ELU), Linear, abs ----
    {
        const float* cur = m->melr.f();
        int ld = m->mel;
        for (int i = 0; i < 5; ++i) {
            WinSpec w;
            w.pad = 1;
            float* dst = (i & 1) ? m->f0b.f() : m->f0a.f();
            STEP(conv_rows(m, m->f0c[i], cur, T, ld, T, w, dst, nullptr));
            HSTEP(t3k::launch_hift_elem(dst, nullptr, nullptr, dst, (size_t)T * m->f0_ch, 0, st));
            cur = dst;
            ld = m->f0_ch;
        }
        WinSpec w;
        STEP(conv_rows(m, m->f0_cls, cur, T, m->f0_ch, T, w, m->f0v.f(), nullptr));
        HSTEP(t3k::launch_hift_elem(m->f0v.f(), nullptr, nullptr, m->f0v.f(), (size_t)T, 1, st));
    }
    // ---- f0_upsamp + m_source (R/hifigan.R:609-614) ----
    HSTEP(t3k::launch_hift_source(m->f0v.f(), static_cast<double*>(m->base_d.p), m->phase.f(), noise ? m->noise.f() : nullptr, m->src_lw,
                                  m->src_lb, T, m->n_harm, kTotalUp, kSr, kVoiced, m->src.f(), st));
    if (cache_len > 0)   // s[,, 1:cache_len] <- cache_source (R/hifigan.R:617-620)
        HSTEP(hipMemcpyAsync(m->src.p, cache_source, (size_t)cache_len * 4, hipMemcpyHostToDevice, st));
    // ---- decode (R/hifigan.R:503-578) ----
    HSTEP(t3k::launch_hift_stft(m->src.f(), Tw, m->stft.f(), kSpecLd, st));
    {
        WinSpec w;
        w.pad = 3;
        STEP(conv_rows(m, m->conv_pre, m->melr.f(), T, m->mel, T, w, m->x.f(), nullptr));
    }
    int L = T;
    float* x = m->x.f();
    for (int i = 0; i < 3; ++i) {
        const int u = kUps[i][0], k = kUps[i][1], C = m->ups[i].N;
        const int Lup = L * u;
        WinSpec wu;
        wu.up = u; wu.pad = (k - u) / 2; wu.act = 1; wu.slope = 0.1f;                       // leaky_relu(x, 0.1) then the transposed conv
        float* up_out = m->upb.f();
        int Lx = Lup;
        if (i == 2) {
            // reflection_pad1d(c(1, 0)) (R/hifigan.R:455, 533): one row in front, a copy of row 1 of the unpadded sequence
            STEP(conv_rows(m, m->ups[i], x, L, m->ups[i].C, Lup, wu, up_out + C, nullptr));
            HSTEP(hipMemcpyAsync(up_out, up_out + (size_t)2 * C, (size_t)C * 4, hipMemcpyDeviceToDevice, st));
            Lx = Lup + 1;
        } else {
            STEP(conv_rows(m, m->ups[i], x, L, m->ups[i].C, Lup, wu, up_out, nullptr));
        }
        // source fusion: source_downs[i](s_stft) -> source_resblocks[i] (R/hifigan.R:537-549)
        int du = 120;
        for (int q = 0; q <= i; ++q) du /= kUps[q][0];
        WinSpec wd;
        wd.stride = du; wd.pad = du == 1 ? 0 : du / 2;
        const int Ls = (frames + 2 * wd.pad - m->src_down[i].k) / du + 1;
        const int Lm = std::min(Lx, Ls);
        STEP(conv_rows(m, m->src_down[i], m->stft.f(), frames, kSpecLd, Lm, wd, m->si.f(), nullptr));
        STEP(resblock(m, m->src_res[i], m->si.f(), Lm, m->xr[0].f()));
        HSTEP(t3k::launch_hift_elem(up_out, m->xr[0].f(), nullptr, x, (size_t)Lm * C, 3, st));   // x <- x[:L] + si[:L]
        // three residual blocks on the same x, averaged (R/hifigan.R:552-561)
        for (int j = 0; j < 3; ++j) STEP(resblock(m, m->res[i * 3 + j], x, Lm, m->xr[j].f()));
        HSTEP(t3k::launch_hift_elem(m->xr[0].f(), m->xr[1].f(), m->xr[2].f(), x, (size_t)Lm * C, 2, st));
        L = Lm;
    }
    {
        WinSpec w;
        w.pad = 3; w.act = 1; w.slope = 0.01f;                                                  // nnf_leaky_relu's default slope (R/hifigan.R:565)
        STEP(conv_rows(m, m->conv_post, x, L, m->conv_post.C, L, w, m->post.f(), nullptr));
    }
    HSTEP(t3k::launch_hift_istft(m->post.f(), L, m->spec.f(), m->audio.f(), kLimit, st));
    const int n_audio = kHop * (L - 1);
    if (device_ms) {
        HSTEP(hipEventRecord(e1, st));
        HSTEP(hipEventSynchronize(e1));
        HSTEP(hipEventElapsedTime(device_ms, e0, e1));
    }
    if (n_audio != Tw) return done(fail(T3MI_ERR_STATE, "t3mi_hift_inference: internal length mismatch"));
    HSTEP(hipMemcpyAsync(audio, m->audio.p, (size_t)Tw * 4, hipMemcpyDeviceToHost, st));
    if (source_out) HSTEP(hipMemcpyAsync(source_out, m->src.p, (size_t)Tw * 4, hipMemcpyDeviceToHost, st));
    if (f0_out) HSTEP(hipMemcpyAsync(f0_out, m->f0v.p, (size_t)T * 4, hipMemcpyDeviceToHost, st));
    HSTEP(hipStreamSynchronize(st));
#undef STEP
#undef HSTEP
    return done(T3MI_OK);
}

}  // extern "C"
