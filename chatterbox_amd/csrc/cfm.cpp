// t3mi_cfm_*: host side of the MI355X-native CFM estimator and its Euler solvers (include/t3mi.h; SURVEY §8 f4,
// R/s3gen.R:389-751).  Owns the estimator's weights (read from the reference's unchanged s3gen safetensors layout,
// keys of load_cfm_estimator_weights R/s3gen.R:1070-1178) and a workspace sized by the rows of a call.  Activations are
// rows [B * T, C] fp32; the causal convolutions run as GEMMs over 3-tap row windows with their weights re-laid
// [out][tap][in] at load; every linear is an exact-fp32 MFMA GEMM (launch_gemm, wdtype 0).
#include "t3mi.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "cfm_kernels.hpp"
#include "kernels.hpp"
#include "safetensors.hpp"

int t3mi_set_error(int code, const std::string& msg);   // t3mi.cpp: stores the text t3mi_last_error() returns

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

struct Lin {          // y = A W^T + b ; W [N][K] row-major on device
    const float* w = nullptr;
    const float* b = nullptr;
    const float* colsum = nullptr;   // != nullptr: the LayerNorm in front of the layer is folded into w / b (cfm_kernels.hpp, DenseGemmArgs)
    int N = 0, K = 0;
};
struct Norm {
    const float *w = nullptr, *b = nullptr;
};
struct Block1d {      // causal_block1d: conv (k = 3, as [out][3 * in]) + LayerNorm
    Lin conv;
    Norm norm;
};
struct Resnet {
    Block1d b1, b2;
    Lin res;          // 1 x 1 res_conv
    int mlp_row = 0;  // row offset of this block's time-embedding projection in the stacked mlp GEMM
    int cin = 0;
};
struct Tfm {
    // (norm1 and norm3 are folded into qkv and ff1 by the loader)
    Lin qkv, out, ff1, ff2;
};

struct ConfLayer {    // conformer_encoder_layer (R/conformer.R:356-404)
    Norm n_mha, n_ff;
    Lin qkv, pos, out, w1, w2;          // qkv: linear_q | linear_k | linear_v stacked (weights and biases)
    const float *bias_u = nullptr, *bias_v = nullptr;   // pos_bias_u / pos_bias_v [heads][64]
};
struct Flow {         // causal_masked_diff_xvec (R/s3gen.R:771-803) without its decoder
    int vocab = 0, d = 0, heads = 0, ffn = 0, spk = 0, n_blocks = 0, n_up = 0;
    const float* embedding = nullptr;   // input_embedding [vocab][d]
    Lin spk_affine, enc_proj, embed, up_embed, pre1, pre2, up_conv;
    Norm embed_n, up_embed_n, after_norm;
    int pre1_k = 4, up_k = 5;
    std::vector<ConfLayer> layers;      // encoders then up_encoders
    Buf wbuf;
    Buf ids, x0, xs, xn, col, qkv, ao, ffb, pe, pp, bd, apo, apml, part, mu, spk_in, spk_out;
    size_t ws_T = 0;
};

}  // namespace

struct t3mi_cfm {
    int device = 0;
    int mel = 0, C = 0, in_ch = 0, inner = 0, heads = 0, ff = 0, time_in = 0, time_dim = 0, n_mid = 0, n_tfm = 0;
    bool meanflow = false;
    hipStream_t stream = nullptr;
    Buf wbuf;
    int64_t weight_bytes = 0;
    Lin time1, time2, mixer, mlp_all, down_conv, up_conv, final_proj;
    Block1d final_block;
    std::vector<Resnet> resnets;       // down, mid 0..n_mid-1, up
    std::vector<Tfm> tfms;             // down (n_tfm), mid (n_mid * n_tfm), up (n_tfm)
    // workspace of the current (B, T)
    Buf h0, col, xa, xb, xn, xm, qkv, ao, ffb, skip, cat, dout, tsin, temb, temb2, tact, tadd, tcur, mask, xrows, stat, ttab;
    Buf apo, apml;      // attention key-split partials (launch_cfm_attn)
    Buf gpart;          // K-split partial slabs of the GEMMs with few feature tiles (launch_gemm's `part`)
    size_t ws_rows = 0;
    int ws_B = 0;
    std::unique_ptr<Flow> flow;        // t3mi_flow_open: the modules in front of the estimator
};

namespace {

// ---- checkpoint -> HBM -------------------------------------------------------------------------------------------
struct Loader {
    t3mi_cfm* m;
    const t3mi::SafeTensorsFile& st;
    std::string prefix;
    std::vector<float> host;   // everything, in upload order
    std::string err;

    const t3mi::StTensor* find(const std::string& key, std::vector<int64_t> shape) {
        const t3mi::StTensor* t = st.find(prefix + key);
        if (!t) { err = "checkpoint has no tensor '" + prefix + key + "'"; return nullptr; }
        if (t->shape != shape) { err = "tensor '" + prefix + key + "' has an unexpected shape"; return nullptr; }
        if (t->dtype != t3mi::StDtype::F32 && t->dtype != t3mi::StDtype::F16 && t->dtype != t3mi::StDtype::BF16) {
            err = "tensor '" + prefix + key + "' is not a floating-point tensor";
            return nullptr;
        }
        return t;
    }
    // appends the tensor (converted to fp32) and returns its offset in `host`, or -1
    long long put(const std::string& key, std::vector<int64_t> shape) {
        const t3mi::StTensor* t = find(key, shape);
        if (!t) return -1;
        const size_t n = (size_t)t->numel(), off = host.size();
        host.resize(off + ((n + 3) & ~size_t(3)));   // 16-byte aligned starts
        if (t->dtype == t3mi::StDtype::F32) std::memcpy(host.data() + off, t->data, n * 4);
        else if (!t3mi::SafeTensorsFile::to_f32(*t, host.data() + off, err)) return -1;
        return (long long)off;
    }
    // conv weight [O][I][k] -> [O][k][I] (the row order of cfm_window3_kernel); k = 1 is a plain [O][I]
    long long put_conv(const std::string& key, int O, int I, int k) {
        const long long off = put(key, {O, I, k});
        if (off < 0 || k == 1) return off;
        std::vector<float> tmp(host.begin() + off, host.begin() + off + (size_t)O * I * k);
        for (int o = 0; o < O; ++o)
            for (int i = 0; i < I; ++i)
                for (int j = 0; j < k; ++j) host[(size_t)off + ((size_t)o * k + j) * I + i] = tmp[((size_t)o * I + i) * k + j];
        return off;
    }
};

struct Pending {   // device pointers are fixed up after the single upload
    const float** slot;
    long long off;
};

int load_estimator(t3mi_cfm* m, const t3mi::SafeTensorsFile& st, const std::string& prefix, int n_heads) {
    auto dim_of = [&](const std::string& key, size_t nd) -> const t3mi::StTensor* {
        const t3mi::StTensor* t = st.find(prefix + key);
        return (t && t->shape.size() == nd) ? t : nullptr;
    };
    const t3mi::StTensor* fp = dim_of("final_proj.weight", 3);
    const t3mi::StTensor* c0 = dim_of("down_blocks.0.0.block1.block.0.weight", 3);
    const t3mi::StTensor* t1 = dim_of("time_mlp.linear_1.weight", 2);
    const t3mi::StTensor* q0 = dim_of("down_blocks.0.1.0.attn1.to_q.weight", 2);
    const t3mi::StTensor* f0 = dim_of("down_blocks.0.1.0.ff.net.0.proj.weight", 2);
    if (!fp || !c0 || !t1 || !q0 || !f0)
        return fail(T3MI_ERR_IO, "checkpoint lacks the CFM estimator tensors under '" + prefix + "' (final_proj, down_blocks.0.*, time_mlp)");
    for (const t3mi::StTensor* t : {fp, c0, t1, q0, f0})
        for (int64_t s : t->shape)
            if (s <= 0 || s > (1 << 20)) return fail(T3MI_ERR_IO, "CFM checkpoint: implausible tensor extents");
    m->mel = (int)fp->shape[0];
    m->C = (int)fp->shape[1];
    m->in_ch = (int)c0->shape[1];
    m->time_dim = (int)t1->shape[0];
    m->time_in = (int)t1->shape[1];
    m->inner = (int)q0->shape[0];
    m->ff = (int)f0->shape[0];
    if (m->in_ch != 4 * m->mel) return fail(T3MI_ERR_IO, "CFM estimator: in_channels must be 4 x mel bins (x + mu + spks + cond, R/s3gen.R:470-478)");
    if (m->inner % 64) return fail(T3MI_ERR_IO, "CFM estimator: attention inner dimension must be heads x 64 (R/s3gen.R:241)");
    m->heads = m->inner / 64;
    if (n_heads > 0 && n_heads != m->heads) return fail(T3MI_ERR_ARG, "t3mi_cfm_open: n_heads does not match the checkpoint (inner / 64)");
    // every GEMM's K: a multiple of 64 (launch_gemm's fp32 fragment step), rows of 16-byte pieces
    for (int k : {m->C, 3 * m->C, 6 * m->C, 3 * m->in_ch, m->in_ch, m->inner, m->ff, m->time_in, m->time_dim})
        if (k % 64) return fail(T3MI_ERR_IO, "CFM estimator: every channel count must be a multiple of 64 elements per GEMM row (got " + std::to_string(k) + ")");
    if (m->C > 1024 || m->mel % 4) return fail(T3MI_ERR_IO, "CFM estimator: hidden size above 1024 or mel bins not a multiple of 4");
    m->n_mid = 0;
    while (st.find(prefix + "mid_blocks." + std::to_string(m->n_mid) + ".0.block1.block.0.weight")) ++m->n_mid;
    m->n_tfm = 0;
    while (st.find(prefix + "down_blocks.0.1." + std::to_string(m->n_tfm) + ".norm1.weight")) ++m->n_tfm;
    m->meanflow = st.find(prefix + "time_embed_mixer.weight") != nullptr;
    if (m->n_tfm <= 0) return fail(T3MI_ERR_IO, "CFM estimator: no transformer blocks under down_blocks.0.1");

    Loader L{m, st, prefix, {}, ""};
    std::vector<Pending> pend;
    const int C = m->C, Tm = m->time_dim;
    bool ok = true;
    auto reg = [&](const float** slot, long long off) {
        if (off < 0) ok = false;
        pend.push_back({slot, off});
    };
    auto lin = [&](Lin& l, const std::string& key, int N, int K, bool bias) {
        l.N = N; l.K = K;
        reg(&l.w, L.put(key + ".weight", {N, K}));
        if (bias && ok) reg(&l.b, L.put(key + ".bias", {N}));
    };
    auto conv = [&](Lin& l, const std::string& key, int O, int I, int k) {
        l.N = O; l.K = I * k;
        reg(&l.w, L.put_conv(key + ".weight", O, I, k));
        if (ok) reg(&l.b, L.put(key + ".bias", {O}));
    };
    auto norm = [&](Norm& n, const std::string& key, int d) {
        reg(&n.w, L.put(key + ".weight", {d}));
        if (ok) reg(&n.b, L.put(key + ".bias", {d}));
    };
    const int n_res = m->n_mid + 2;
    m->resnets.resize(n_res);
    m->tfms.resize((size_t)(m->n_mid + 2) * m->n_tfm);
    lin(m->time1, "time_mlp.linear_1", Tm, m->time_in, true);
    if (ok) lin(m->time2, "time_mlp.linear_2", Tm, Tm, true);
    if (ok && m->meanflow) lin(m->mixer, "time_embed_mixer", Tm, 2 * Tm, false);
    // the time-embedding projections of all resnets, stacked [n_res * C][Tm] so that one GEMM serves an estimator call
    m->mlp_all.N = n_res * C;
    m->mlp_all.K = Tm;
    long long mlp_w0 = -1, mlp_b0 = -1;
    std::vector<std::string> res_prefix;
    res_prefix.push_back("down_blocks.0.0.");
    for (int i = 0; i < m->n_mid; ++i) res_prefix.push_back("mid_blocks." + std::to_string(i) + ".0.");
    res_prefix.push_back("up_blocks.0.0.");
    for (int r = 0; r < n_res && ok; ++r) {
        const long long o = L.put(res_prefix[r] + "mlp.1.weight", {C, Tm});
        if (o < 0) ok = false;
        if (r == 0) mlp_w0 = o;
    }
    for (int r = 0; r < n_res && ok; ++r) {
        const long long o = L.put(res_prefix[r] + "mlp.1.bias", {C});
        if (o < 0) ok = false;
        if (r == 0) mlp_b0 = o;
    }
    if (ok) { reg(&m->mlp_all.w, mlp_w0); reg(&m->mlp_all.b, mlp_b0); }
    auto resnet = [&](Resnet& r, const std::string& p, int cin, int idx) {
        r.cin = cin;
        r.mlp_row = idx * C;
        conv(r.b1.conv, p + "block1.block.0", C, cin, 3);
        if (ok) norm(r.b1.norm, p + "block1.block.2", C);
        if (ok) conv(r.b2.conv, p + "block2.block.0", C, C, 3);
        if (ok) norm(r.b2.norm, p + "block2.block.2", C);
        if (ok) conv(r.res, p + "res_conv", C, cin, 1);
    };
    // the LayerNorm `nkey` in front of the linear layer at host offset ow (bias at ob, or -1) is folded into it
    // (dense_fold_layernorm): the layer's bias and column sums are new host arrays
    auto fold = [&](Lin& l, long long ow, long long ob, const std::string& nkey) {
        const long long lw = L.put(nkey + ".weight", {l.K});
        const long long lb = lw >= 0 ? L.put(nkey + ".bias", {l.K}) : -1;
        if (lb < 0) { ok = false; return; }
        const size_t npad = ((size_t)l.N + 3) & ~size_t(3), ob2 = L.host.size();
        L.host.resize(ob2 + 2 * npad);
        float* H = L.host.data();
        t3k::dense_fold_layernorm(H + ow, ob >= 0 ? H + ob : nullptr, H + lw, H + lb, l.N, l.K, H + ob2, H + ob2 + npad);
        reg(&l.w, ow);
        reg(&l.b, (long long)ob2);
        reg(&l.colsum, (long long)(ob2 + npad));
    };
    auto tfm = [&](Tfm& t, const std::string& p) {
        // to_q | to_k | to_v stacked [3 * inner][C]: one GEMM, the layout cfm_attn_kernel reads; norm1 folded in
        t.qkv.N = 3 * m->inner; t.qkv.K = C;
        const long long oq = ok ? L.put(p + "attn1.to_q.weight", {m->inner, C}) : -1;
        const long long ok_ = oq >= 0 ? L.put(p + "attn1.to_k.weight", {m->inner, C}) : -1;
        const long long ov = ok_ >= 0 ? L.put(p + "attn1.to_v.weight", {m->inner, C}) : -1;
        if (ov < 0) { ok = false; return; }
        fold(t.qkv, oq, -1, p + "norm1");
        if (ok) lin(t.out, p + "attn1.to_out.0", C, m->inner, true);
        t.ff1.N = m->ff; t.ff1.K = C;
        const long long o1 = ok ? L.put(p + "ff.net.0.proj.weight", {m->ff, C}) : -1;
        const long long b1 = o1 >= 0 ? L.put(p + "ff.net.0.proj.bias", {m->ff}) : -1;
        if (b1 < 0) { ok = false; return; }
        fold(t.ff1, o1, b1, p + "norm3");
        if (ok) lin(t.ff2, p + "ff.net.2", C, m->ff, true);
    };
    if (ok) resnet(m->resnets[0], "down_blocks.0.0.", m->in_ch, 0);
    for (int i = 0; i < m->n_tfm && ok; ++i) tfm(m->tfms[i], "down_blocks.0.1." + std::to_string(i) + ".");
    if (ok) conv(m->down_conv, "down_blocks.0.2", C, C, 3);
    for (int b = 0; b < m->n_mid && ok; ++b) {
        resnet(m->resnets[1 + b], "mid_blocks." + std::to_string(b) + ".0.", C, 1 + b);
        for (int j = 0; j < m->n_tfm && ok; ++j)
            tfm(m->tfms[(size_t)(1 + b) * m->n_tfm + j], "mid_blocks." + std::to_string(b) + ".1." + std::to_string(j) + ".");
    }
    if (ok) resnet(m->resnets[n_res - 1], "up_blocks.0.0.", 2 * C, n_res - 1);
    for (int i = 0; i < m->n_tfm && ok; ++i)
        tfm(m->tfms[(size_t)(n_res - 1) * m->n_tfm + i], "up_blocks.0.1." + std::to_string(i) + ".");
    if (ok) conv(m->up_conv, "up_blocks.0.2", C, C, 3);
    if (ok) conv(m->final_block.conv, "final_block.block.0", C, C, 3);
    if (ok) norm(m->final_block.norm, "final_block.block.2", C);
    if (ok) conv(m->final_proj, "final_proj", m->mel, C, 1);
    if (!ok) return fail(T3MI_ERR_IO, L.err.empty() ? "CFM checkpoint: malformed tensor" : L.err);

    m->weight_bytes = (int64_t)L.host.size() * 4;
    HIPCHK(m->wbuf.need(L.host.size() * 4));
    HIPCHK(hipMemcpy(m->wbuf.p, L.host.data(), L.host.size() * 4, hipMemcpyHostToDevice));
    for (const Pending& p : pend) *p.slot = m->wbuf.f() + p.off;
    return T3MI_OK;
}

// ---- one estimator call over the rows in the workspace -----------------------------------------------------------
// y = epi(LN?(A) W^T + b).  The dense fp32 kernel (dense_gemm.hip) takes every shape it accepts (a layer with a folded LayerNorm
// runs nowhere else); the rest (N = 80: final_proj; the time path's few rows) goes through launch_gemm.
int gemm(t3mi_cfm* m, const float* A, const Lin& l, int M, int epi, float* y, const float* resid, float* yact, float ln_eps = 1e-5f) {
    if ((M >= 16 || l.colsum) && t3k::dense_gemm_supported(M, l.N, l.K, epi)) {
        t3k::DenseGemmArgs d;
        d.A = A; d.lda = l.K; d.W = l.w; d.bias = l.b; d.M = M; d.N = l.N; d.K = l.K; d.epi = epi;
        d.y = epi == t3k::EPI_GELU ? yact : y; d.ldy = l.N; d.resid = resid;
        d.ln_colsum = l.colsum; d.ln_eps = ln_eps;
        HIPCHK(t3k::launch_dense_gemm(d, m->stream));
        return T3MI_OK;
    }
    if (l.colsum) return fail(T3MI_ERR_STATE, "a layer with a folded LayerNorm has a shape the dense GEMM does not take (N % 32, K % 64)");
    t3k::GemmArgs a;
    a.W = l.w; a.A = A; a.M = M; a.N = l.N; a.K = l.K; a.bias = l.b;
    a.y = y; a.resid = resid; a.ldy = l.N; a.yact = yact;
    if (l.N <= m->C && M >= 64) {   // few feature tiles (N = hidden or mel): K split over blocks, folded by launch_gemm in a fixed order
        a.part = m->gpart.f();
        a.part_floats = m->gpart.bytes / 4;
    }
    HIPCHK(t3k::launch_gemm(a, epi, 0, m->stream));
    return T3MI_OK;
}

int ensure_ws(t3mi_cfm* m, int B, int T) {
    const size_t M = (size_t)B * T;
    if (M <= m->ws_rows && B <= m->ws_B) return T3MI_OK;
    const size_t C = m->C, rows = std::max(M, m->ws_rows);
    const int Bn = std::max(B, m->ws_B);
    HIPCHK(hipStreamSynchronize(m->stream));
    const size_t cmax = std::max<size_t>(2 * C, m->in_ch);
    HIPCHK(m->h0.need(rows * m->in_ch * 4));
    HIPCHK(m->col.need(rows * 3 * cmax * 4));
    HIPCHK(m->xa.need(rows * C * 4));
    HIPCHK(m->xb.need(rows * C * 4));
    HIPCHK(m->xn.need(rows * C * 4));
    HIPCHK(m->xm.need(rows * cmax * 4));
    HIPCHK(m->qkv.need(rows * 3 * m->inner * 4));
    HIPCHK(m->ao.need(rows * m->inner * 4));
    HIPCHK(m->apo.need((size_t)t3k::kCfmAttnMaxSplit * rows * m->inner * 4));
    HIPCHK(m->apml.need((size_t)t3k::kCfmAttnMaxSplit * rows * m->heads * 2 * 4));
    HIPCHK(m->gpart.need((size_t)16 * rows * C * 4));
    HIPCHK(m->ffb.need(rows * m->ff * 4));
    HIPCHK(m->skip.need(rows * C * 4));
    HIPCHK(m->cat.need(rows * 2 * C * 4));
    HIPCHK(m->dout.need(rows * m->mel * 4));
    HIPCHK(m->mask.need(rows * 4));
    HIPCHK(m->xrows.need(rows * m->mel * 4));
    HIPCHK(m->stat.need(rows * 3 * m->mel * 4));
    HIPCHK(m->tsin.need((size_t)2 * Bn * m->time_in * 4));
    HIPCHK(m->temb.need((size_t)2 * Bn * m->time_dim * 4));
    HIPCHK(m->temb2.need((size_t)2 * Bn * m->time_dim * 4));
    HIPCHK(m->tact.need((size_t)2 * Bn * m->time_dim * 4));
    HIPCHK(m->tadd.need((size_t)Bn * m->mlp_all.N * 4));
    HIPCHK(m->tcur.need((size_t)2 * Bn * 4));
    m->ws_rows = rows;
    m->ws_B = Bn;
    return T3MI_OK;
}

// causal_block1d (R/s3gen.R:186-196): out = mish(LayerNorm(conv3(x * mask))) * mask (+ add rows); `masked` (optional)
// receives x * mask
int block1d(t3mi_cfm* m, const Block1d& b, const float* x, int cin, int M, int T, const float* add, float* out, float* masked) {
    HIPCHK(t3k::launch_cfm_window3(x, m->mask.f(), m->col.f(), masked, M, cin, T, m->stream));
    int rc = gemm(m, m->col.f(), b.conv, M, t3k::EPI_STORE, m->xn.f(), nullptr, nullptr);
    if (rc) return rc;
    HIPCHK(t3k::launch_cfm_norm_mish(m->xn.f(), b.norm.w, b.norm.b, m->mask.f(), add, m->mlp_all.N, out, M, m->C, T, 1e-5f, m->stream));
    return T3MI_OK;
}

// causal_resnet_block1d (R/s3gen.R:223-230): x [M][cin] -> out [M][C]
int resnet(t3mi_cfm* m, const Resnet& r, const float* x, int M, int T, float* out) {
    int rc = block1d(m, r.b1, x, r.cin, M, T, m->tadd.f() + r.mlp_row, m->xb.f(), m->xm.f());
    if (rc) return rc;
    if ((rc = block1d(m, r.b2, m->xb.f(), m->C, M, T, nullptr, out, nullptr))) return rc;
    return gemm(m, m->xm.f(), r.res, M, t3k::EPI_RESID, out, out, nullptr);   // h + res_conv(x * mask)
}

// basic_transformer_block (R/s3gen.R:359-372), in place on x [M][C]
int transformer(t3mi_cfm* m, const Tfm& t, float* x, int B, int T) {
    const int M = B * T;
    int rc = gemm(m, x, t.qkv, M, t3k::EPI_STORE, m->qkv.f(), nullptr, nullptr);   // norm1 folded into the layer
    if (rc) return rc;
    HIPCHK(t3k::launch_cfm_attn(m->qkv.f(), m->ao.f(), B, T, m->heads, 64, m->apo.f(), m->apml.f(), m->stream));
    if ((rc = gemm(m, m->ao.f(), t.out, M, t3k::EPI_RESID, x, x, nullptr))) return rc;
    if ((rc = gemm(m, x, t.ff1, M, t3k::EPI_GELU, nullptr, nullptr, m->ffb.f()))) return rc;   // norm3 folded
    return gemm(m, m->ffb.f(), t.ff2, M, t3k::EPI_RESID, x, x, nullptr);
}

// time_embeddings + time_mlp (+ meanflow mixer) (R/s3gen.R:459-468) for t (and r) at tcur[0..B) ([B..2B)), then every
// resnet's mlp(mish(t_emb)) in one stacked GEMM -> tadd [B][n_res * C]
int time_path(t3mi_cfm* m, int B, bool with_r) {
    const int nb = with_r ? 2 * B : B;
    HIPCHK(t3k::launch_cfm_time_sin(m->tcur.f(), m->tsin.f(), nb, m->time_in, 1000.0f, m->stream));
    int rc = gemm(m, m->tsin.f(), m->time1, nb, t3k::EPI_STORE, m->temb.f(), nullptr, nullptr);
    if (rc) return rc;
    HIPCHK(t3k::launch_cfm_act(m->temb.f(), m->tact.f(), (size_t)nb * m->time_dim, 0, m->stream));
    if ((rc = gemm(m, m->tact.f(), m->time2, nb, t3k::EPI_STORE, m->temb.f(), nullptr, nullptr))) return rc;
    const float* te = m->temb.f();
    if (with_r) {   // cat(t_emb, r_emb) along the features, then the mixer (no bias)
        HIPCHK(t3k::launch_cfm_concat(m->temb.f(), m->time_dim, m->temb.f() + (size_t)B * m->time_dim, m->time_dim, m->temb2.f(), B, m->stream));
        if ((rc = gemm(m, m->temb2.f(), m->mixer, B, t3k::EPI_STORE, m->tact.f(), nullptr, nullptr))) return rc;
        HIPCHK(hipMemcpyAsync(m->temb.p, m->tact.p, (size_t)B * m->time_dim * 4, hipMemcpyDeviceToDevice, m->stream));
    }
    HIPCHK(t3k::launch_cfm_act(te, m->tact.f(), (size_t)B * m->time_dim, 1, m->stream));
    return gemm(m, m->tact.f(), m->mlp_all, B, t3k::EPI_STORE, m->tadd.f(), nullptr, nullptr);
}

// cfm_estimator$forward (R/s3gen.R:446-520) on the workspace: h0 [M][4 mel] packed, mask, tcur set -> dout [M][mel]
int estimator_rows(t3mi_cfm* m, int B, int T, bool with_r) {
    const int M = B * T, C = m->C, nt = m->n_tfm;
    int rc = time_path(m, B, with_r);
    if (rc) return rc;
    float* h = m->xa.f();
    if ((rc = resnet(m, m->resnets[0], m->h0.f(), M, T, h))) return rc;
    for (int i = 0; i < nt; ++i)
        if ((rc = transformer(m, m->tfms[i], h, B, T))) return rc;
    HIPCHK(hipMemcpyAsync(m->skip.p, h, (size_t)M * C * 4, hipMemcpyDeviceToDevice, m->stream));   // hidden_skip, :488
    HIPCHK(t3k::launch_cfm_window3(h, m->mask.f(), m->col.f(), nullptr, M, C, T, m->stream));
    if ((rc = gemm(m, m->col.f(), m->down_conv, M, t3k::EPI_STORE, m->cat.f(), nullptr, nullptr))) return rc;   // cat reused as a [M][C] buffer
    const float* cur = m->cat.f();
    for (int b = 0; b < m->n_mid; ++b) {
        if ((rc = resnet(m, m->resnets[1 + b], cur, M, T, h))) return rc;
        for (int j = 0; j < nt; ++j)
            if ((rc = transformer(m, m->tfms[(size_t)(1 + b) * nt + j], h, B, T))) return rc;
        cur = h;
    }
    if (m->n_mid == 0) {   // no mid block: the down conv's output is what the up block sees
        HIPCHK(hipMemcpyAsync(h, m->cat.p, (size_t)M * C * 4, hipMemcpyDeviceToDevice, m->stream));
    }
    HIPCHK(t3k::launch_cfm_concat(h, C, m->skip.f(), C, m->cat.f(), M, m->stream));                 // :503
    if ((rc = resnet(m, m->resnets.back(), m->cat.f(), M, T, h))) return rc;
    for (int i = 0; i < nt; ++i)
        if ((rc = transformer(m, m->tfms[(size_t)(m->n_mid + 1) * nt + i], h, B, T))) return rc;
    HIPCHK(t3k::launch_cfm_window3(h, m->mask.f(), m->col.f(), nullptr, M, C, T, m->stream));
    if ((rc = gemm(m, m->col.f(), m->up_conv, M, t3k::EPI_STORE, m->xb.f(), nullptr, nullptr))) return rc;       // :513
    if ((rc = block1d(m, m->final_block, m->xb.f(), C, M, T, nullptr, h, nullptr))) return rc;                  // :516
    HIPCHK(t3k::launch_cfm_mask_rows(h, m->mask.f(), m->xm.f(), M, C, m->stream));
    return gemm(m, m->xm.f(), m->final_proj, M, t3k::EPI_STORE, m->dout.f(), nullptr, nullptr);                // :517
}

// (B, C, T) host -> rows [B * T][C] host
void to_rows(const float* src, int B, int C, int T, float* dst, int ld, int col0) {
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c) {
            const float* s = src + ((size_t)b * C + c) * T;
            for (int t = 0; t < T; ++t) dst[((size_t)b * T + t) * ld + col0 + c] = s[t];
        }
}

// ---- flow module: checkpoint -> HBM (keys of load_s3gen_weights R/s3gen.R:1215-1228, load_conformer_encoder_weights
// R/conformer.R:627-713) ------------------------------------------------------------------------------------------
int load_flow(t3mi_cfm* m, const t3mi::SafeTensorsFile& st) {
    std::unique_ptr<Flow> f(new Flow());
    const t3mi::StTensor* emb = st.find("flow.input_embedding.weight");
    const t3mi::StTensor* sa = st.find("flow.spk_embed_affine_layer.weight");
    const t3mi::StTensor* w1 = st.find("flow.encoder.encoders.0.feed_forward.w_1.weight");
    const t3mi::StTensor* c1 = st.find("flow.encoder.pre_lookahead_layer.conv1.weight");
    const t3mi::StTensor* uc = st.find("flow.encoder.up_layer.conv.weight");
    const t3mi::StTensor* bu = st.find("flow.encoder.encoders.0.self_attn.pos_bias_u");
    if (!emb || !sa || !w1 || !c1 || !uc || !bu || emb->shape.size() != 2 || sa->shape.size() != 2 || w1->shape.size() != 2 ||
        c1->shape.size() != 3 || uc->shape.size() != 3 || bu->shape.size() != 2)
        return fail(T3MI_ERR_IO, "checkpoint lacks the S3Gen flow tensors (flow.input_embedding, flow.spk_embed_affine_layer, flow.encoder.*)");
    for (const t3mi::StTensor* t : {emb, sa, w1, c1, uc, bu})
        for (int64_t sdim : t->shape)
            if (sdim <= 0 || sdim > (1 << 24)) return fail(T3MI_ERR_IO, "flow checkpoint: implausible tensor extents");
    f->vocab = (int)emb->shape[0];
    f->d = (int)emb->shape[1];
    f->spk = (int)sa->shape[1];
    f->ffn = (int)w1->shape[0];
    f->heads = (int)bu->shape[0];
    f->pre1_k = (int)c1->shape[2];
    f->up_k = (int)uc->shape[2];
    const int d = f->d;
    if ((int)sa->shape[0] != m->mel) return fail(T3MI_ERR_IO, "flow.spk_embed_affine_layer does not project to the estimator's mel bins");
    if (f->heads <= 0 || d != f->heads * 64 || (int)bu->shape[1] != 64)
        return fail(T3MI_ERR_IO, "conformer encoder: attention needs heads x 64 = model dimension (R/conformer.R:203-208)");
    if (d % 64 || f->ffn % 64 || f->spk % 64 || f->pre1_k < 1 || f->pre1_k > 8 || f->up_k < 1 || f->up_k > 8)
        return fail(T3MI_ERR_IO, "conformer encoder: dimensions must be multiples of 64 and kernels at most 8 taps");
    while (st.find("flow.encoder.encoders." + std::to_string(f->n_blocks) + ".norm_mha.weight")) ++f->n_blocks;
    while (st.find("flow.encoder.up_encoders." + std::to_string(f->n_up) + ".norm_mha.weight")) ++f->n_up;

    Loader L{m, st, "flow.", {}, ""};
    std::vector<Pending> pend;
    bool ok = true;
    auto reg = [&](const float** slot, long long off) {
        if (off < 0) ok = false;
        pend.push_back({slot, off});
    };
    auto lin = [&](Lin& l, const std::string& key, int N, int K, bool bias) {
        l.N = N; l.K = K;
        if (ok) reg(&l.w, L.put(key + ".weight", {N, K}));
        if (bias && ok) reg(&l.b, L.put(key + ".bias", {N}));
    };
    auto conv = [&](Lin& l, const std::string& key, int O, int I, int k) {
        l.N = O; l.K = I * k;
        if (ok) reg(&l.w, L.put_conv(key + ".weight", O, I, k));
        if (ok) reg(&l.b, L.put(key + ".bias", {O}));
    };
    auto norm = [&](Norm& n, const std::string& key, int dd) {
        if (ok) reg(&n.w, L.put(key + ".weight", {dd}));
        if (ok) reg(&n.b, L.put(key + ".bias", {dd}));
    };
    reg(&f->embedding, L.put("input_embedding.weight", {f->vocab, d}));
    lin(f->spk_affine, "spk_embed_affine_layer", m->mel, f->spk, true);
    lin(f->enc_proj, "encoder_proj", m->mel, d, true);
    lin(f->embed, "encoder.embed.out.0", d, d, true);
    norm(f->embed_n, "encoder.embed.out.1", d);
    conv(f->pre1, "encoder.pre_lookahead_layer.conv1", d, d, f->pre1_k);
    conv(f->pre2, "encoder.pre_lookahead_layer.conv2", d, d, 3);
    conv(f->up_conv, "encoder.up_layer.conv", d, d, f->up_k);
    lin(f->up_embed, "encoder.up_embed.out.0", d, d, true);
    norm(f->up_embed_n, "encoder.up_embed.out.1", d);
    norm(f->after_norm, "encoder.after_norm", d);
    f->layers.resize((size_t)f->n_blocks + f->n_up);
    for (int i = 0; i < f->n_blocks + f->n_up && ok; ++i) {
        ConfLayer& l = f->layers[i];
        const std::string p = i < f->n_blocks ? "encoder.encoders." + std::to_string(i) + "."
                                              : "encoder.up_encoders." + std::to_string(i - f->n_blocks) + ".";
        norm(l.n_mha, p + "norm_mha", d);
        norm(l.n_ff, p + "norm_ff", d);
        l.qkv.N = 3 * d; l.qkv.K = d;
        const long long wq = ok ? L.put(p + "self_attn.linear_q.weight", {d, d}) : -1;
        const long long wk = wq >= 0 ? L.put(p + "self_attn.linear_k.weight", {d, d}) : -1;
        const long long wv = wk >= 0 ? L.put(p + "self_attn.linear_v.weight", {d, d}) : -1;
        const long long bq = wv >= 0 ? L.put(p + "self_attn.linear_q.bias", {d}) : -1;
        const long long bk = bq >= 0 ? L.put(p + "self_attn.linear_k.bias", {d}) : -1;
        const long long bvv = bk >= 0 ? L.put(p + "self_attn.linear_v.bias", {d}) : -1;
        if (bvv < 0) { ok = false; break; }
        reg(&l.qkv.w, wq);
        reg(&l.qkv.b, bq);
        lin(l.pos, p + "self_attn.linear_pos", d, d, false);
        lin(l.out, p + "self_attn.linear_out", d, d, true);
        if (ok) reg(&l.bias_u, L.put(p + "self_attn.pos_bias_u", {f->heads, 64}));
        if (ok) reg(&l.bias_v, L.put(p + "self_attn.pos_bias_v", {f->heads, 64}));
        lin(l.w1, p + "feed_forward.w_1", f->ffn, d, true);
        lin(l.w2, p + "feed_forward.w_2", d, f->ffn, true);
    }
    if (!ok) return fail(T3MI_ERR_IO, L.err.empty() ? "flow checkpoint: malformed tensor" : L.err);
    HIPCHK(f->wbuf.need(L.host.size() * 4));
    HIPCHK(hipMemcpy(f->wbuf.p, L.host.data(), L.host.size() * 4, hipMemcpyHostToDevice));
    for (const Pending& p : pend) *p.slot = f->wbuf.f() + p.off;
    m->weight_bytes += (int64_t)L.host.size() * 4;
    m->flow = std::move(f);
    return T3MI_OK;
}

int flow_ws(t3mi_cfm* m, int T) {   // T = tokens; the up stage runs at 2T
    Flow* f = m->flow.get();
    if ((size_t)T <= f->ws_T) return T3MI_OK;
    HIPCHK(hipStreamSynchronize(m->stream));
    const size_t T2 = 2 * (size_t)T, d = f->d, kmax = std::max(std::max(f->pre1_k, f->up_k), 3);
    HIPCHK(f->ids.need(T2 * 4));
    HIPCHK(f->x0.need(T2 * d * 4));
    HIPCHK(f->xs.need(T2 * d * 4));
    HIPCHK(f->xn.need(T2 * d * 4));
    HIPCHK(f->col.need(T2 * kmax * d * 4));
    HIPCHK(f->qkv.need(T2 * 3 * d * 4));
    HIPCHK(f->ao.need(T2 * d * 4));
    HIPCHK(f->ffb.need(T2 * f->ffn * 4));
    HIPCHK(f->pe.need(2 * T2 * d * 4));
    HIPCHK(f->pp.need(2 * T2 * d * 4));
    HIPCHK(f->bd.need((size_t)f->heads * T2 * 2 * T2 * 4));
    HIPCHK(f->apo.need((size_t)t3k::kCfmAttnMaxSplit * T2 * d * 4));
    HIPCHK(f->apml.need((size_t)t3k::kCfmAttnMaxSplit * T2 * f->heads * 2 * 4));
    HIPCHK(f->part.need((size_t)16 * T2 * d * 4));
    HIPCHK(f->mu.need(T2 * m->mel * 4));
    HIPCHK(f->spk_in.need((size_t)f->spk * 4));
    HIPCHK(f->spk_out.need((size_t)m->mel * 4));
    f->ws_T = (size_t)T;
    return T3MI_OK;
}

int fgemm(t3mi_cfm* m, const float* A, const Lin& l, int M, int epi, float* y, const float* resid) {
    Flow* f = m->flow.get();
    if (M >= 16 && t3k::dense_gemm_supported(M, l.N, l.K, epi)) {
        t3k::DenseGemmArgs d;
        d.A = A; d.lda = l.K; d.W = l.w; d.bias = l.b; d.M = M; d.N = l.N; d.K = l.K; d.epi = epi;
        d.y = y; d.ldy = l.N; d.resid = resid;
        HIPCHK(t3k::launch_dense_gemm(d, m->stream));
        return T3MI_OK;
    }
    t3k::GemmArgs a;
    a.W = l.w; a.A = A; a.M = M; a.N = l.N; a.K = l.K; a.bias = l.b;
    a.y = y; a.resid = resid; a.ldy = l.N;
    if (l.N <= f->d && M >= 64) {   // few feature tiles: K split over blocks, folded in a fixed order
        a.part = f->part.f();
        a.part_floats = f->part.bytes / 4;
    }
    HIPCHK(t3k::launch_gemm(a, epi, 0, m->stream));
    return T3MI_OK;
}

// linear_no_subsampling (R/conformer.R:124-134): Linear -> LayerNorm -> * sqrt(d) into xs; pos_emb for T rows into pe
int conf_embed(t3mi_cfm* m, const float* x, const Lin& l, const Norm& n, int T) {
    Flow* f = m->flow.get();
    int rc = fgemm(m, x, l, T, t3k::EPI_STORE, f->xn.f(), nullptr);
    if (rc) return rc;
    HIPCHK(t3k::launch_layernorm_rows(f->xn.f(), n.w, n.b, 1e-5f, f->xs.p, T, f->d, 0, m->stream));
    HIPCHK(t3k::launch_cfm_act(f->xs.f(), f->xs.f(), (size_t)T * f->d, 3, m->stream, sqrtf((float)f->d)));
    HIPCHK(t3k::launch_rel_pos_emb(f->pe.f(), T, f->d, m->stream));
    return T3MI_OK;
}

// conformer_encoder_layer$forward (R/conformer.R:376-403) in place on xs [T][d]; pe holds the 2T - 1 position rows
int conf_layer(t3mi_cfm* m, const ConfLayer& l, int T) {
    Flow* f = m->flow.get();
    const int d = f->d;
    float* x = f->xs.f();
    HIPCHK(t3k::launch_layernorm_rows(x, l.n_mha.w, l.n_mha.b, 1e-12f, f->xn.p, T, d, 0, m->stream));
    int rc = fgemm(m, f->xn.f(), l.qkv, T, t3k::EPI_STORE, f->qkv.f(), nullptr);
    if (rc) return rc;
    if ((rc = fgemm(m, f->pe.f(), l.pos, 2 * T - 1, t3k::EPI_STORE, f->pp.f(), nullptr))) return rc;     // linear_pos(pos_emb), :241
    HIPCHK(t3k::launch_conf_bd(f->qkv.f(), 3 * d, l.bias_v, f->pp.f(), d, f->bd.f(), T, f->heads, m->stream));
    HIPCHK(t3k::launch_cfm_attn(f->qkv.f(), f->ao.f(), 1, T, f->heads, 64, f->apo.f(), f->apml.f(), m->stream, l.bias_u, f->bd.f()));
    if ((rc = fgemm(m, f->ao.f(), l.out, T, t3k::EPI_RESID, x, x))) return rc;
    HIPCHK(t3k::launch_layernorm_rows(x, l.n_ff.w, l.n_ff.b, 1e-12f, f->xn.p, T, d, 0, m->stream));
    if ((rc = fgemm(m, f->xn.f(), l.w1, T, t3k::EPI_STORE, f->ffb.f(), nullptr))) return rc;
    HIPCHK(t3k::launch_cfm_act(f->ffb.f(), f->ffb.f(), (size_t)T * f->ffn, 0, m->stream));
    return fgemm(m, f->ffb.f(), l.w2, T, t3k::EPI_RESID, x, x);
}

// upsample_conformer_encoder_full$forward (R/conformer.R:517-578), one sequence, every position valid:
// x0 [T][d] (embedded tokens) -> xs [2T][d]
int encoder_rows(t3mi_cfm* m, int T) {
    Flow* f = m->flow.get();
    const int d = f->d;
    int rc = conf_embed(m, f->x0.f(), f->embed, f->embed_n, T);
    if (rc) return rc;
    // pre_lookahead_layer (:160-185): look-ahead conv -> leaky_relu -> causal conv -> + residual
    HIPCHK(t3k::launch_rows_window(f->xs.f(), f->col.f(), T, T, d, f->pre1_k, 0, 1, m->stream));
    if ((rc = fgemm(m, f->col.f(), f->pre1, T, t3k::EPI_STORE, f->xn.f(), nullptr))) return rc;
    HIPCHK(t3k::launch_cfm_act(f->xn.f(), f->xn.f(), (size_t)T * d, 2, m->stream));
    HIPCHK(t3k::launch_rows_window(f->xn.f(), f->col.f(), T, T, d, 3, -2, 1, m->stream));
    if ((rc = fgemm(m, f->col.f(), f->pre2, T, t3k::EPI_RESID, f->xs.f(), f->xs.f()))) return rc;
    for (int i = 0; i < f->n_blocks; ++i)
        if ((rc = conf_layer(m, f->layers[i], T))) return rc;
    // upsample_1d (:426-444): nearest x 2, left pad, conv
    const int T2 = 2 * T;
    HIPCHK(t3k::launch_rows_window(f->xs.f(), f->col.f(), T2, T, d, f->up_k, -(f->up_k - 1), 2, m->stream));
    if ((rc = fgemm(m, f->col.f(), f->up_conv, T2, t3k::EPI_STORE, f->x0.f(), nullptr))) return rc;
    if ((rc = conf_embed(m, f->x0.f(), f->up_embed, f->up_embed_n, T2))) return rc;
    for (int i = 0; i < f->n_up; ++i)
        if ((rc = conf_layer(m, f->layers[(size_t)f->n_blocks + i], T2))) return rc;
    HIPCHK(t3k::launch_layernorm_rows(f->xs.f(), f->after_norm.w, f->after_norm.b, 1e-5f, f->xn.p, T2, d, 0, m->stream));
    return T3MI_OK;   // result in xn [2T][d]
}

// a pair of timing events that cannot leak on an early return
struct EventPair {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    EventPair() = default;
    EventPair(const EventPair&) = delete;
    EventPair& operator=(const EventPair&) = delete;
    ~EventPair() {
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
    }
    hipError_t start(hipStream_t st) {
        hipError_t e = hipEventCreate(&e0);
        if (e == hipSuccess) e = hipEventCreate(&e1);
        if (e == hipSuccess) e = hipEventRecord(e0, st);
        return e;
    }
    hipError_t stop(hipStream_t st, float* ms) {
        hipError_t e = hipEventRecord(e1, st);
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        if (e == hipSuccess) e = hipEventElapsedTime(ms, e0, e1);
        return e;
    }
};

int check_handle(t3mi_cfm* m) {
    if (!m) return fail(T3MI_ERR_ARG, "null CFM handle");
    return T3MI_OK;
}

}  // namespace

extern "C" {

int t3mi_cfm_open(const char* path, const char* prefix, int device, int n_heads, t3mi_cfm** out) {
    if (!path || !out) return fail(T3MI_ERR_ARG, "t3mi_cfm_open: null argument");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(T3MI_ERR_HIP, "t3mi_cfm_open: no HIP device visible (the MI355X path has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(T3MI_ERR_ARG, "t3mi_cfm_open: device index out of range");
    HIPCHK(hipSetDevice(device));
    t3mi::SafeTensorsFile st;
    std::string err;
    if (!st.open(path, err)) return fail(T3MI_ERR_IO, err);
    std::unique_ptr<t3mi_cfm> m(new t3mi_cfm());
    m->device = device;
    int rc = load_estimator(m.get(), st, prefix && prefix[0] ? prefix : "flow.decoder.estimator.", n_heads);
    if (rc) return rc;
    HIPCHK(hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking));
    *out = m.release();
    return T3MI_OK;
}

int t3mi_cfm_close(t3mi_cfm* m) {
    if (!m) return T3MI_OK;
    (void)hipSetDevice(m->device);
    if (m->stream) {
        (void)hipStreamSynchronize(m->stream);
        (void)hipStreamDestroy(m->stream);
    }
    delete m;
    return T3MI_OK;
}

int t3mi_cfm_get_info(const t3mi_cfm* m, t3mi_cfm_info* info) {
    if (!m || !info) return fail(T3MI_ERR_ARG, "t3mi_cfm_get_info: null argument");
    info->mel = m->mel; info->hidden = m->C; info->n_mid = m->n_mid; info->n_tfm = m->n_tfm; info->heads = m->heads;
    info->head_dim = 64; info->d_ff = m->ff; info->time_in = m->time_in; info->time_dim = m->time_dim;
    info->meanflow = m->meanflow ? 1 : 0; info->device = m->device; info->reserved_ = 0;
    info->weight_bytes = m->weight_bytes;
    return T3MI_OK;
}

int t3mi_cfm_estimator(t3mi_cfm* m, int B, int T, const float* x, const float* mask, const float* mu, const float* t,
                       const float* spks, const float* cond, const float* r, float* out) {
    int rc = check_handle(m);
    if (rc) return rc;
    if (!x || !mask || !mu || !t || !spks || !cond || !out) return fail(T3MI_ERR_ARG, "t3mi_cfm_estimator: null argument (spks and cond are part of the 4 x mel input)");
    if (B <= 0 || T <= 0 || (long long)B * T > (1 << 22)) return fail(T3MI_ERR_ARG, "t3mi_cfm_estimator: bad batch or sequence length");
    if (r && !m->meanflow) return fail(T3MI_ERR_ARG, "t3mi_cfm_estimator: r given but the checkpoint has no time_embed_mixer (not a meanflow model)");
    HIPCHK(hipSetDevice(m->device));
    if ((rc = ensure_ws(m, B, T))) return rc;
    const int M = B * T, mel = m->mel;
    std::vector<float> h0((size_t)M * 4 * mel), mk((size_t)M), tt((size_t)2 * B);
    to_rows(x, B, mel, T, h0.data(), 4 * mel, 0);
    to_rows(mu, B, mel, T, h0.data(), 4 * mel, mel);
    for (int b = 0; b < B; ++b)
        for (int tq = 0; tq < T; ++tq)
            for (int c = 0; c < mel; ++c) h0[((size_t)b * T + tq) * 4 * mel + 2 * mel + c] = spks[(size_t)b * mel + c];   // :473
    to_rows(cond, B, mel, T, h0.data(), 4 * mel, 3 * mel);
    std::memcpy(mk.data(), mask, (size_t)M * 4);     // (B, 1, T) is already row order
    for (int b = 0; b < B; ++b) { tt[b] = t[b]; tt[B + b] = r ? r[b] : 0.f; }
    HIPCHK(hipMemcpyAsync(m->h0.p, h0.data(), h0.size() * 4, hipMemcpyHostToDevice, m->stream));
    HIPCHK(hipMemcpyAsync(m->mask.p, mk.data(), mk.size() * 4, hipMemcpyHostToDevice, m->stream));
    HIPCHK(hipMemcpyAsync(m->tcur.p, tt.data(), tt.size() * 4, hipMemcpyHostToDevice, m->stream));
    if ((rc = estimator_rows(m, B, T, r != nullptr))) return rc;
    std::vector<float> d((size_t)M * mel);
    HIPCHK(hipMemcpyAsync(d.data(), m->dout.p, d.size() * 4, hipMemcpyDeviceToHost, m->stream));
    HIPCHK(hipStreamSynchronize(m->stream));
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < mel; ++c)
            for (int tq = 0; tq < T; ++tq) out[((size_t)b * mel + c) * T + tq] = d[((size_t)b * T + tq) * mel + c];
    return T3MI_OK;
}

int t3mi_cfm_solve(t3mi_cfm* m, int T, const float* z, const float* mu, const float* mask, const float* spks, const float* cond,
                   int n_timesteps, double cfg_rate, int meanflow, float* out, float* device_ms) {
    int rc = check_handle(m);
    if (rc) return rc;
    if (!z || !mu || !mask || !spks || !cond || !out) return fail(T3MI_ERR_ARG, "t3mi_cfm_solve: null argument");
    if (T <= 0 || T > (1 << 16)) return fail(T3MI_ERR_ARG, "t3mi_cfm_solve: the sequence must hold 1 .. 65536 mel frames");
    if (n_timesteps <= 0 || n_timesteps > 1000) return fail(T3MI_ERR_ARG, "t3mi_cfm_solve: n_timesteps must be in 1..1000");
    if (meanflow && !m->meanflow) return fail(T3MI_ERR_ARG, "t3mi_cfm_solve: meanflow asked of a checkpoint without time_embed_mixer");
    HIPCHK(hipSetDevice(m->device));
    const int B = meanflow ? 1 : 2;    // solve_euler: the cond / uncond pair (R/s3gen.R:700-726); basic_euler: one row
    if ((rc = ensure_ws(m, B, T))) return rc;
    const int mel = m->mel, M = B * T;
    // rows layout; row block 2 (unconditional) keeps x and the mask, mu / spks / cond stay zero (R/s3gen.R:712-726)
    std::vector<float> xr((size_t)T * mel), stat((size_t)M * 3 * mel, 0.f), mk((size_t)M);
    to_rows(z, 1, mel, T, xr.data(), mel, 0);
    to_rows(mu, 1, mel, T, stat.data(), 3 * mel, 0);
    for (int tq = 0; tq < T; ++tq)
        for (int c = 0; c < mel; ++c) stat[(size_t)tq * 3 * mel + mel + c] = spks[c];
    to_rows(cond, 1, mel, T, stat.data(), 3 * mel, 2 * mel);
    for (int b = 0; b < B; ++b) std::memcpy(mk.data() + (size_t)b * T, mask, (size_t)T * 4);
    // time schedule (R/s3gen.R:622-625): float32 linspace, cosine for the standard solver; t accumulates in float32 (:738)
    const int n = n_timesteps;
    std::vector<float> ts(n + 1);
    for (int i = 0; i <= n; ++i) {
        // torch_linspace(0, 1, n + 1) in float32: start + i * step for the first half, end - (n - i) * step for the second
        const float step = 1.0f / (float)n;
        ts[i] = i < (n + 1) / 2 ? (float)i * step : 1.0f - (float)(n - i) * step;
        if (!meanflow) ts[i] = 1.0f - cosf((ts[i] * 0.5f) * (float)M_PI);
    }
    std::vector<float> ttab((size_t)n * 2 * B), dts(n);
    {
        float t = ts[0], dt = ts[1] - ts[0];
        for (int s = 0; s < n; ++s) {
            for (int b = 0; b < B; ++b) {
                ttab[(size_t)s * 2 * B + b] = meanflow ? ts[s] : t;          // basic_euler passes t_span[step] itself (:641)
                ttab[(size_t)s * 2 * B + B + b] = meanflow ? ts[s + 1] : 0.f;   // r (:642)
            }
            dts[s] = meanflow ? ts[s + 1] - ts[s] : dt;
            t = t + dt;
            if (s + 1 < n) dt = ts[s + 2] - ts[s + 1];
        }
    }
    HIPCHK(m->ttab.need(ttab.size() * 4));
    hipStream_t st = m->stream;
    HIPCHK(hipMemcpyAsync(m->xrows.p, xr.data(), xr.size() * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(m->stat.p, stat.data(), stat.size() * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(m->mask.p, mk.data(), mk.size() * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(m->ttab.p, ttab.data(), ttab.size() * 4, hipMemcpyHostToDevice, st));
    EventPair ev;
    if (device_ms) HIPCHK(ev.start(st));
    const float rate = (float)cfg_rate, one_plus = (float)(1.0 + cfg_rate);
    for (int s = 0; s < n && !rc; ++s) {
        hipError_t e = hipMemcpyAsync(m->tcur.p, m->ttab.f() + (size_t)s * 2 * B, (size_t)2 * B * 4, hipMemcpyDeviceToDevice, st);
        // h0 rows = [x | mu spks cond]: both row blocks see the same x (x_in[1:2] <- x, :717)
        for (int b = 0; b < B && e == hipSuccess; ++b)
            e = t3k::launch_cfm_concat(m->xrows.f(), mel, m->stat.f() + (size_t)b * T * 3 * mel, 3 * mel, m->h0.f() + (size_t)b * T * 4 * mel, T, st);
        if (e != hipSuccess) { rc = fail(T3MI_ERR_HIP, std::string("t3mi_cfm_solve: ") + hipGetErrorString(e)); break; }
        if ((rc = estimator_rows(m, B, T, meanflow != 0))) break;
        e = t3k::launch_cfm_euler(m->xrows.f(), m->dout.f(), meanflow ? nullptr : m->dout.f() + (size_t)T * mel, (size_t)T * mel, dts[s],
                                  one_plus, rate, st);
        if (e != hipSuccess) rc = fail(T3MI_ERR_HIP, std::string("t3mi_cfm_solve: ") + hipGetErrorString(e));
    }
    if (device_ms && !rc) {
        hipError_t e = ev.stop(st, device_ms);
        if (e != hipSuccess) rc = fail(T3MI_ERR_HIP, std::string("t3mi_cfm_solve: event timing: ") + hipGetErrorString(e));
    }
    if (rc) {
        (void)hipStreamSynchronize(st);
        return rc;
    }
    HIPCHK(hipMemcpyAsync(xr.data(), m->xrows.p, xr.size() * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    for (int c = 0; c < mel; ++c)
        for (int tq = 0; tq < T; ++tq) out[(size_t)c * T + tq] = xr[(size_t)tq * mel + c];
    return T3MI_OK;
}

// The relative-position term of the conformer attention is a [heads][2T][4T - 1] fp32 matrix: 17 GB at 8192 tokens (5.5 minutes of
// speech); longer inputs are refused rather than left to fail in hipMalloc.
static constexpr int kFlowMaxTokens = 8192;

int t3mi_flow_open(const char* path, int device, t3mi_cfm** out) {
    if (!path || !out) return fail(T3MI_ERR_ARG, "t3mi_flow_open: null argument");
    *out = nullptr;
    t3mi_cfm* m = nullptr;
    int rc = t3mi_cfm_open(path, "flow.decoder.estimator.", device, 0, &m);
    if (rc) return rc;
    t3mi::SafeTensorsFile st;
    std::string err;
    if (!st.open(path, err)) { t3mi_cfm_close(m); return fail(T3MI_ERR_IO, err); }
    if ((rc = load_flow(m, st))) {
        const std::string keep = t3mi_last_error();
        t3mi_cfm_close(m);
        return fail(rc, keep);
    }
    *out = m;
    return T3MI_OK;
}

int t3mi_flow_get_info(const t3mi_cfm* m, t3mi_flow_info* info) {
    if (!m || !info) return fail(T3MI_ERR_ARG, "t3mi_flow_get_info: null argument");
    if (!m->flow) return fail(T3MI_ERR_STATE, "not a flow handle (open it with t3mi_flow_open)");
    const Flow* f = m->flow.get();
    info->vocab = f->vocab; info->d_model = f->d; info->heads = f->heads; info->d_ff = f->ffn; info->n_blocks = f->n_blocks;
    info->n_up_blocks = f->n_up; info->spk_embed_dim = f->spk; info->mel = m->mel;
    info->weight_bytes = m->weight_bytes;
    return T3MI_OK;
}

int t3mi_flow_encoder(t3mi_cfm* m, int T, const float* x, float* out) {
    int rc = check_handle(m);
    if (rc) return rc;
    if (!m->flow) return fail(T3MI_ERR_STATE, "not a flow handle (open it with t3mi_flow_open)");
    if (!x || !out || T <= 0 || T > kFlowMaxTokens) return fail(T3MI_ERR_ARG, "t3mi_flow_encoder: bad argument (1 .. 8192 tokens)");
    HIPCHK(hipSetDevice(m->device));
    if ((rc = flow_ws(m, T))) return rc;
    Flow* f = m->flow.get();
    HIPCHK(hipMemcpyAsync(f->x0.p, x, (size_t)T * f->d * 4, hipMemcpyHostToDevice, m->stream));
    if ((rc = encoder_rows(m, T))) return rc;
    HIPCHK(hipMemcpyAsync(out, f->xn.p, (size_t)2 * T * f->d * 4, hipMemcpyDeviceToHost, m->stream));
    HIPCHK(hipStreamSynchronize(m->stream));
    return T3MI_OK;
}

int t3mi_flow_inference(t3mi_cfm* m, const int32_t* token, int n_token, const int32_t* prompt_token, int n_prompt,
                        const float* prompt_feat, int n_feat, const float* embedding, const float* noise, int noise_T,
                        int n_timesteps, double cfg_rate, int meanflow, int finalize, float* out, int cap_frames, int* n_frames,
                        float* device_ms) {
    int rc = check_handle(m);
    if (rc) return rc;
    if (!m->flow) return fail(T3MI_ERR_STATE, "not a flow handle (open it with t3mi_flow_open)");
    if (!token || !prompt_token || !prompt_feat || !embedding || !noise || !out || !n_frames)
        return fail(T3MI_ERR_ARG, "t3mi_flow_inference: null argument");
    if (n_token <= 0 || n_prompt < 0 || n_feat < 0 || n_token > kFlowMaxTokens || n_prompt > kFlowMaxTokens || n_token + n_prompt > kFlowMaxTokens)
        return fail(T3MI_ERR_ARG, "t3mi_flow_inference: bad token counts (prompt + tokens must be 1 .. 8192)");
    HIPCHK(hipSetDevice(m->device));
    Flow* f = m->flow.get();
    const int T = n_prompt + n_token, mel = m->mel;
    int T2 = 2 * T;
    if (!finalize) T2 -= 3 * 2;                                                                    // pre_lookahead_len * token_mel_ratio, :844-846
    const int mel_len1 = 2 * n_prompt, mel_len2 = T2 - mel_len1;
    if (mel_len2 <= 0) return fail(T3MI_ERR_ARG, "t3mi_flow_inference: nothing left to generate after the prompt");
    if (cap_frames < mel_len2) return fail(T3MI_ERR_ARG, "t3mi_flow_inference: output holds fewer than " + std::to_string(mel_len2) + " frames");
    if (noise_T < T2) return fail(T3MI_ERR_ARG, "t3mi_flow_inference: the noise must cover " + std::to_string(T2) + " frames");
    if (n_timesteps <= 0 || n_timesteps > 1000) return fail(T3MI_ERR_ARG, "t3mi_flow_inference: n_timesteps must be in 1..1000");
    if (meanflow && !m->meanflow) return fail(T3MI_ERR_ARG, "t3mi_flow_inference: meanflow asked of a checkpoint without time_embed_mixer");
    if ((rc = flow_ws(m, T))) return rc;
    hipStream_t st = m->stream;
    EventPair ev;
    if (device_ms) HIPCHK(ev.start(st));
    // token <- cat(prompt_token, token), clamped to the table (R/s3gen.R:826-836; the mask is all ones for one utterance)
    std::vector<int32_t> ids((size_t)T);
    for (int i = 0; i < T; ++i) {
        const int32_t v = i < n_prompt ? prompt_token[i] : token[i - n_prompt];
        ids[i] = std::min(std::max(v, 0), f->vocab - 1);
    }
    // embedding <- spk_embed_affine_layer(nnf_normalize(embedding)) (R/s3gen.R:822-823): the 192-float norm on the host
    std::vector<float> en((size_t)f->spk);
    {
        float ss = 0.f;
        for (int i = 0; i < f->spk; ++i) ss += embedding[i] * embedding[i];
        const float nrm = std::max(sqrtf(ss), 1e-12f);
        for (int i = 0; i < f->spk; ++i) en[i] = embedding[i] / nrm;
    }
    HIPCHK(hipMemcpyAsync(f->ids.p, ids.data(), (size_t)T * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(f->spk_in.p, en.data(), en.size() * 4, hipMemcpyHostToDevice, st));
    HIPCHK(t3k::launch_embed_rows(f->embedding, static_cast<const int*>(f->ids.p), nullptr, f->x0.f(), T, f->d, st));
    if ((rc = encoder_rows(m, T))) return rc;                                                      // :839
    if ((rc = fgemm(m, f->xn.f(), f->enc_proj, T2, t3k::EPI_STORE, f->mu.f(), nullptr))) return rc;   // encoder_proj, :854
    if ((rc = fgemm(m, f->spk_in.f(), f->spk_affine, 1, t3k::EPI_STORE, f->spk_out.f(), nullptr))) return rc;
    std::vector<float> mu_rows((size_t)T2 * mel), spk((size_t)mel);
    HIPCHK(hipMemcpyAsync(mu_rows.data(), f->mu.p, mu_rows.size() * 4, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(spk.data(), f->spk_out.p, spk.size() * 4, hipMemcpyDeviceToHost, st));
    float enc_ms = 0.f;
    if (device_ms) HIPCHK(ev.stop(st, &enc_ms));
    HIPCHK(hipStreamSynchronize(st));
    // decoder inputs in the reference's layouts (R/s3gen.R:857-883): mu (1, mel, T2); conds = the prompt mel over its frames
    std::vector<float> mu((size_t)mel * T2), cond((size_t)mel * T2, 0.f), mask((size_t)T2, 1.0f), z((size_t)mel * T2), res((size_t)mel * T2);
    const int nc = std::min(n_feat, mel_len1);
    for (int c = 0; c < mel; ++c)
        for (int t = 0; t < T2; ++t) {
            mu[(size_t)c * T2 + t] = mu_rows[(size_t)t * mel + c];
            if (t < nc) cond[(size_t)c * T2 + t] = prompt_feat[(size_t)t * mel + c];
            z[(size_t)c * T2 + t] = noise[(size_t)c * noise_T + t];
        }
    float solve_ms = 0.f;
    rc = t3mi_cfm_solve(m, T2, z.data(), mu.data(), mask.data(), spk.data(), cond.data(), n_timesteps, cfg_rate, meanflow, res.data(),
                        device_ms ? &solve_ms : nullptr);
    if (rc) return rc;
    for (int c = 0; c < mel; ++c)                                                                    // feat[,, (mel_len1 + 1):...], :887
        std::memcpy(out + (size_t)c * mel_len2, res.data() + (size_t)c * T2 + mel_len1, (size_t)mel_len2 * 4);
    *n_frames = mel_len2;
    if (device_ms) *device_ms = enc_ms + solve_ms;
    return T3MI_OK;
}

}  // extern "C"
