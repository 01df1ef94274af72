// t3mi C ABI: host side of the MI355X-native T3 decoder (see include/t3mi.h).
//
// Owns the weights (read from the reference's unchanged safetensors layout), the RoPE tables,
// the KV slabs and all per-session state in HBM; drives the decode step as a replayed HIP graph
// so that a step needs no host synchronisation (the reference syncs 2-3x per token:
// R/t3.R:530,574 ; src/t3_decode.cpp:339,361-362).
#include "t3mi.h"

#include <hip/hip_runtime.h>
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <functional>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <string>
#include <vector>

#include "kernels.hpp"
#include "prompt.hpp"
#include "safetensors.hpp"

// ---- fixed working memory of t3mi_prepare_input_embeds_static (the entry point that allocates nothing) -------------
// One array per GPU, part of the code object: every GPU the library runs on holds it until the process ends.
// Layout: [0, kStaticRowsFloats * 4) perceiver rows | ids | speaker vector.  In host code the NAME `scratch` is only a
// handle for hipGetSymbolAddress: kernels and copies take the per-device address that call returns (static_scratch()).
constexpr size_t kStaticRowsFloats = 2u << 20;   // (4 * max(n_cond, n_query) + 4 * n_query) * d floats: 8 MiB
constexpr size_t kStaticIds = 4096;              // conditioning + text ids (+ start / stop / BOS)
constexpr size_t kStaticSpk = 1024;              // speaker-embedding floats
constexpr size_t kStaticScratchBytes = kStaticRowsFloats * 4 + kStaticIds * 4 + kStaticSpk * 4;
static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];

namespace {

thread_local std::string g_err = "";

constexpr int kMaxDevices = 64;
// device address of `scratch` on each GPU, filled by t3mi_open.  Not locked: the library's contract is one handle per thread
// (include/t3mi.h), and two threads opening handles on the same device at once would both store the same address here.
void* g_static_scratch[kMaxDevices] = {};

// with `device` current: the address of this GPU's copy of `scratch` (looked up once per device)
hipError_t static_scratch(int device, void** out) {
    *out = nullptr;
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    if (g_static_scratch[device] == nullptr) {
        void* p = nullptr;
        hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        g_static_scratch[device] = p;
    }
    *out = g_static_scratch[device];
    return hipSuccess;
}

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

#define HIPCHK(expr)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(T3MI_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));     \
    } while (0)

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    hipError_t alloc(size_t n) {
        release();
        if (n == 0) n = 16;
        hipError_t e = hipMalloc(&p, n);
        if (e == hipSuccess) bytes = n;
        else p = nullptr;
        return e;
    }
    template <typename T>
    T* as() const { return static_cast<T*>(p); }
};

struct LayerW {
    const float* ln1 = nullptr;
    const void* wqkv = nullptr;  // [3d, d]
    const void* wo = nullptr;    // [d, d]
    const float* ln2 = nullptr;
    const void* wgu = nullptr;   // [2F, d]  gate rows then up rows
    const void* wdown = nullptr; // [d, F]
    // Turbo (GPT-2, R/t3.R:1190-1310): wqkv = c_attn, wo = attn.c_proj, wgu = c_fc [F, d], wdown = mlp.c_proj
    const float* ln1_b = nullptr;   // LayerNorm biases
    const float* ln2_b = nullptr;
    const float* b_qkv = nullptr;   // raw linear biases (many-row path: exact LayerNorm rows + bias)
    const float* b_fc = nullptr;
    const float* b_qkv_f = nullptr; // bias + W . b_ln  (decode GEMV path: the LayerNorm bias folded in)
    const float* b_fc_f = nullptr;
    const float* o_colmean = nullptr;   // Turbo: [d] column means of attn.c_proj's weight (AttnRowsArgs::dot_vec)
    float o_bias_mean = 0.f;            // and the mean of its bias (GemmArgs::mean_shift)
    const float* b_o = nullptr;
    const float* b_down = nullptr;
};

struct Session {
    int n_utt = 0, rpu = 1, rows = 0, max_len = 0, cap = 0, max_s0 = 0;
    int kps = 64, n_split = 1;  // attention split: keys per block, blocks per (row, head)
    bool gemm_mode = false;     // > 4 activation rows: linears run as MFMA GEMMs (batch_kernels.hip)
    bool kv_bf16 = false;       // KV slabs hold bf16 (many-row regime with bf16 weights), else fp32
    DevBuf gpart;               // GEMM-path K-split partial sums: [<= 8][rows][d] fp32
    DevBuf rowss;               // split norm: [d / 32][<= kAsMaxRows] partial sums of squares of the residual rows, a spare slab, then
    float* rowmu = nullptr;     // (inside rowss; Turbo's LayerNorm form) [kAsMaxRows] row means, [<= kAsMaxRows][H <= 32] head products
    int pend_ks = 0;            // the decode step's last down_proj left its fold to the head's norm launch
    const float* pend_bias = nullptr;
    DevBuf xn, attn_out, hact;  // GEMM-path activation rows (float with fp32 weights, bf16 with bf16 weights)
    t3mi_params prm{};
    std::vector<int32_t> s0;
    DevBuf kc, vc, x, qbuf, pm, pl, pacc, hbuf, logits, rope_cur, rowpos, start, emb_rows, emb_data, seen, tokens, n_tokens,
        done, s0_dev, forced, dist_ids, dist_probs, dist_n, drawn;
    DevBuf pfs[9];              // prefill scratch (prefill_rows): grown on demand, freed after t3mi_session_begin's prefill when it is large
    DevBuf utt_ids, max_new_u;  // [n_utt] global utterance index (RNG stream) and token limit of each slot (t3mi_session_refill changes them)
    DevBuf filt;                // [n_utt][V] logits after CFG / penalty / temperature (t3mi_session_keep_filtered), else empty
    int n_forced = 0;
    bool gemm_prefill = true, pf_act_f32 = false;   // how session_begin prefilled (t3mi_session_refill does the same)
    hipGraph_t graph = nullptr;
    hipGraphExec_t gexec = nullptr;
    bool graph_tried = false;
    ~Session() {
        if (gexec) (void)hipGraphExecDestroy(gexec);
        if (graph) (void)hipGraphDestroy(graph);
    }
};

}  // namespace

// the error text of this thread, for the other translation units of the library (cfm.cpp)
int t3mi_set_error(int code, const std::string& msg) { return fail(code, msg); }

struct t3mi_model {
    int device = 0, variant = 0, wdtype = 0;
    int n_layers = 0, d = 0, H = 0, F = 0, V = 0, max_pos = 0;
    float eps = 1e-5f;
    hipStream_t stream = nullptr;
    DevBuf wbuf, small, emb, pos_emb, rope_cos, rope_sin;
    DevBuf wfrag;                      // bf16 models: the streamed matrices once more, fragment-major, at wbuf's offsets (build_fragments)
    int as_mask = 15;                  // which GEMMs may take it: 1 qkv, 2 o_proj, 4 gate_up / c_fc, 8 down, 16 head
    int norm_split = 1;                // T3MI_NORM_SPLIT: bit 1 the post-attention norm is split over o_proj / gate_up; bit 2 (experiment, measured
                                       // slower: profiles/r03_ln1_split_ab.txt) Llama's pre-attention norm over down_proj / the next layer's qkv;
                                       // 0: every norm is a launch of its own
    int ln1_lo = 17, ln1_hi = 128;     // row range of the pre-attention split (T3MI_LN1_ROWS=lo,hi)
    bool as_forced = false;            // T3MI_AS=<mask> set: exactly the named GEMMs take it at every row count the kernel accepts (tests, A/B runs)
    // The fragment-major twin of a matrix in wbuf for a launch of M rows, or nullptr (the LDS-staged kernels then run).
    // Row ranges per GEMM, measured over 400-step sessions (profiles/r02_gemm_as_ab.txt).  bf16: qkv wins up to 1024
    // rows, gate_up up to 512, down up to 384 (at 512 it loses 0.1 ms per step), o_proj from 96 rows (below, its K-split
    // LDS-staged form is 1 % ahead); the head only when T3MI_AS names it (no faster at any size).  fp32 (exact-fp32
    // MFMA, 16x the matrix-core time per product): qkv up to 1024 rows, gate_up up to 96, down up to 128, o_proj never.
    const void* frag(const void* w, int which, int M, bool any_rows = false) const {
        if (!wfrag.p || !(as_mask & which)) return nullptr;
        int lo = 1, hi = 0;
        if (as_forced || any_rows) which = 0;   // no tuned range: the kernel's own limit
        if (wdtype == T3MI_W_BF16) {
            switch (which) {
                case 1: hi = 1024; break;
                case 2: lo = 96; hi = 1024; break;
                case 4: hi = 512; break;
                case 8: hi = 384; break;
                case 16: hi = 0; break;
                default: hi = t3k::kAsMaxRows; break;
            }
        } else {
            switch (which) {
                case 1: hi = 1024; break;
                case 4: hi = 96; break;
                case 8: hi = 128; break;
                case 2: case 16: hi = 0; break;
                default: hi = t3k::kAsMaxRows; break;
            }
        }
        if (M < lo || M > hi) return nullptr;
        return static_cast<const char*>(wfrag.p) + (static_cast<const char*>(w) - static_cast<const char*>(wbuf.p));
    }
    // Weight prefetch from idle CUs (t3k::PfArgs): up to pf_max_rows activation rows, a launch that leaves CUs
    // idle carries extra blocks that touch the fragment-major weights the NEXT launches read, dealt so that a line lands in the L2 of
    // the XCD whose block will read it.  pf_blocks: prefetch blocks on the attention launch (it fetches o_proj + gate_up when it has at
    // most pf_max_blocks (item, head) blocks; with more, the qkv launch does); pf_gblocks: on a GEMM launch (o_proj fetches down,
    // gate_up the next layer's qkv); pf_mask: 1 o_proj, 2 gate_up, 4 down, 8 next qkv, 16 the qkv launch may prefetch, 32 always the
    // qkv launch instead of attention.  T3MI_PF=0 turns it off; A/B in profiles/r03_weight_prefetch_ab.txt.
    int pf_blocks = 128, pf_gblocks = 128, pf_max_blocks = 128, pf_mask = 31, pf_xcd = 1, pf_max_rows = 16;   // blocks on the attention launch / on a GEMM launch
    DevBuf pf_sink;   // [max(pf_blocks, pf_gblocks) * 512] words the prefetch threads leave their xor in
    int rope_len = 0;
    std::vector<LayerW> layers;
    const float* norm_w = nullptr;
    const float* norm_b = nullptr;     // Turbo ln_f bias
    const void* head_w = nullptr;
    const float* head_b = nullptr;     // Turbo speech_head bias, raw and with ln_f's bias folded in
    const float* head_b_f = nullptr;
    int64_t weight_bytes = 0;
    std::unique_ptr<Session> sess;
    int cache_state = 0;                   // T3MI_CACHE_*: what t3mi_open_cached did with the packed bf16 file
    void* static_scratch = nullptr;        // this GPU's copy of the file-scope `scratch` array (hipGetSymbolAddress at open)
    t3mi::PromptModel* prompt = nullptr;   // text_emb / cond_enc.* tensors (nullptr: decode-only checkpoint)
    DevBuf prompt_buf;                     // prompts built on device, alive until the next build
    bool use_graph = true;
    int tune[5] = {8, 4, 8, 8, 8};  // waves per block: qkv, o_proj, gate_up, down, head (T3MI_WAVES=a,b,c,d,e)
    int tune_rows[5] = {2, 1, 2, 2, 2};  // output rows per wave-chunk, same order (T3MI_ROWS=a,b,c,d,e)
};

namespace {

// ---- RoPE tables with Llama-3 band scaling (R/llama.R:87-152), float arithmetic as R torch ------
void build_rope(int head_dim, int rows, std::vector<float>& cosv, std::vector<float>& sinv) {
    const float theta = 500000.0f, factor = 8.0f;
    const double low_freq_factor = 1.0, high_freq_factor = 4.0, old_ctx = 8192.0;
    const int half = head_dim / 2;
    std::vector<float> inv(half);
    const double low_wl = old_ctx / low_freq_factor, high_wl = old_ctx / high_freq_factor;
    for (int i = 0; i < half; ++i) {
        const float expo = (float)(2 * i) / (float)head_dim;
        const float f = 1.0f / powf(theta, expo);
        const float wlf = (float)(2.0 * M_PI) / f;
        const double wl = (double)wlf;
        if (wl < high_wl) inv[i] = f;
        else if (wl > low_wl) inv[i] = f / factor;
        else {
            const double smooth = (old_ctx / wl - low_freq_factor) / (high_freq_factor - low_freq_factor);
            const float a = ((float)(1.0 - smooth) * f) / factor;
            const float b = (float)smooth * f;
            inv[i] = a + b;
        }
    }
    cosv.resize((size_t)rows * head_dim);
    sinv.resize((size_t)rows * head_dim);
    for (int t = 0; t < rows; ++t)
        for (int i = 0; i < half; ++i) {
            const float fr = (float)t * inv[i];
            const float c = cosf(fr), s = sinf(fr);
            cosv[(size_t)t * head_dim + i] = c;
            cosv[(size_t)t * head_dim + half + i] = c;
            sinv[(size_t)t * head_dim + i] = s;
            sinv[(size_t)t * head_dim + half + i] = s;
        }
}

int ensure_rope(t3mi_model* m, int rows) {
    if (rows <= m->rope_len) return T3MI_OK;
    std::vector<float> c, s;
    build_rope(64, rows, c, s);
    HIPCHK(m->rope_cos.alloc(c.size() * 4));
    HIPCHK(m->rope_sin.alloc(s.size() * 4));
    HIPCHK(hipMemcpy(m->rope_cos.p, c.data(), c.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(m->rope_sin.p, s.data(), s.size() * 4, hipMemcpyHostToDevice));
    m->rope_len = rows;
    return T3MI_OK;
}

// ---- checkpoint validation (no device): every tensor the loaders below fetch, by name and shape ------------------
struct CkptDims {
    int L = 0, d = 0, H = 0, F = 0, V = 0, max_pos = 0;
};
struct TensorSpec {
    std::string name;
    int64_t d0, d1;   // d1 < 0: one-dimensional
};

int check_specs(const t3mi::SafeTensorsFile& st, const std::vector<TensorSpec>& specs) {
    for (const TensorSpec& sp : specs) {
        const t3mi::StTensor* t = st.find(sp.name);
        if (!t) return fail(T3MI_ERR_IO, "checkpoint has no tensor '" + sp.name + "'");
        const bool ok = (sp.d1 < 0) ? (t->shape.size() == 1 && t->shape[0] == sp.d0)
                                    : (t->shape.size() == 2 && t->shape[0] == sp.d0 && t->shape[1] == sp.d1);
        if (!ok) return fail(T3MI_ERR_IO, "tensor '" + sp.name + "' has an unexpected shape");
        if (t->dtype != t3mi::StDtype::F32 && t->dtype != t3mi::StDtype::F16 && t->dtype != t3mi::StDtype::BF16)
            return fail(T3MI_ERR_IO, "tensor '" + sp.name + "' is not a floating-point tensor (F32 / F16 / BF16)");
    }
    return T3MI_OK;
}

// shapes of the Llama-backbone checkpoint from the file itself (R/llama.R:20-45 for the 520M model), then every tensor
// of R/llama.R:566-607 / R/t3.R:1065-1097 that the decode path reads
int llama_dims(const t3mi::SafeTensorsFile& st, CkptDims& c) {
    const t3mi::StTensor* q0 = st.find("tfmr.layers.0.self_attn.q_proj.weight");
    const t3mi::StTensor* g0 = st.find("tfmr.layers.0.mlp.gate_proj.weight");
    const t3mi::StTensor* hd = st.find("speech_head.weight");
    const t3mi::StTensor* pe = st.find("speech_pos_emb.emb.weight");
    if (!q0 || !g0 || !hd || !pe || q0->shape.size() != 2 || g0->shape.size() != 2 || hd->shape.size() != 2 ||
        pe->shape.size() != 2)
        return fail(T3MI_ERR_IO, "checkpoint lacks the T3 Llama tensors (tfmr.layers.0.*, speech_head, speech_pos_emb)");
    if (q0->shape[1] <= 0 || q0->shape[1] > (1 << 20) || g0->shape[0] <= 0 || g0->shape[0] > (1 << 24) || hd->shape[0] <= 0 ||
        hd->shape[0] > (1 << 24) || pe->shape[0] <= 0 || pe->shape[0] > (1 << 24))
        return fail(T3MI_ERR_IO, "checkpoint: implausible tensor extents");
    c.d = (int)q0->shape[1];
    c.F = (int)g0->shape[0];
    c.V = (int)hd->shape[0];
    c.max_pos = (int)pe->shape[0];
    if (c.d % 64 || q0->shape[0] != c.d) return fail(T3MI_ERR_IO, "hidden size must be heads*64");
    c.H = c.d / 64;
    c.L = 0;
    while (st.find("tfmr.layers." + std::to_string(c.L) + ".self_attn.q_proj.weight")) ++c.L;
    const int64_t d = c.d, F = c.F, V = c.V;
    std::vector<TensorSpec> sp;
    for (int l = 0; l < c.L; ++l) {
        const std::string p = "tfmr.layers." + std::to_string(l) + ".";
        sp.push_back({p + "input_layernorm.weight", d, -1});
        sp.push_back({p + "self_attn.q_proj.weight", d, d});
        sp.push_back({p + "self_attn.k_proj.weight", d, d});
        sp.push_back({p + "self_attn.v_proj.weight", d, d});
        sp.push_back({p + "self_attn.o_proj.weight", d, d});
        sp.push_back({p + "post_attention_layernorm.weight", d, -1});
        sp.push_back({p + "mlp.gate_proj.weight", F, d});
        sp.push_back({p + "mlp.up_proj.weight", F, d});
        sp.push_back({p + "mlp.down_proj.weight", d, F});
    }
    sp.push_back({"tfmr.norm.weight", d, -1});
    sp.push_back({"speech_head.weight", V, d});
    sp.push_back({"speech_emb.weight", V, d});
    sp.push_back({"speech_pos_emb.emb.weight", c.max_pos, d});
    return check_specs(st, sp);
}

// Turbo (GPT-2 backbone): keys of R/t3.R:1508-1549, linear weights [out, in] as nn_linear holds them
int turbo_dims(const t3mi::SafeTensorsFile& st, CkptDims& c) {
    const t3mi::StTensor* qkv0 = st.find("tfmr.h.0.attn.c_attn.weight");
    const t3mi::StTensor* fc0 = st.find("tfmr.h.0.mlp.c_fc.weight");
    const t3mi::StTensor* hd = st.find("speech_head.weight");
    if (!qkv0 || !fc0 || !hd || qkv0->shape.size() != 2 || fc0->shape.size() != 2 || hd->shape.size() != 2)
        return fail(T3MI_ERR_IO, "checkpoint lacks the T3 Turbo tensors (tfmr.h.0.*, speech_head)");
    if (qkv0->shape[1] <= 0 || qkv0->shape[1] > (1 << 20) || fc0->shape[0] <= 0 || fc0->shape[0] > (1 << 24) ||
        hd->shape[0] <= 0 || hd->shape[0] > (1 << 24))
        return fail(T3MI_ERR_IO, "checkpoint: implausible tensor extents");
    c.d = (int)qkv0->shape[1];
    if (qkv0->shape[0] != 3 * (int64_t)c.d)
        return fail(T3MI_ERR_IO, "tfmr.h.0.attn.c_attn.weight must be [3*d, d] (nn_linear layout, R/t3.R:1220)");
    c.F = (int)fc0->shape[0];
    c.V = (int)hd->shape[0];
    c.max_pos = 1 << 30;   // no learned speech position table on this path (R/t3.R:1628)
    if (c.d % 64) return fail(T3MI_ERR_IO, "hidden size must be heads*64");
    c.H = c.d / 64;
    c.L = 0;
    while (st.find("tfmr.h." + std::to_string(c.L) + ".attn.c_attn.weight")) ++c.L;
    const int64_t d = c.d, F = c.F, V = c.V;
    std::vector<TensorSpec> sp;
    for (int l = 0; l < c.L; ++l) {
        const std::string p = "tfmr.h." + std::to_string(l) + ".";
        sp.push_back({p + "ln_1.weight", d, -1});
        sp.push_back({p + "ln_1.bias", d, -1});
        sp.push_back({p + "ln_2.weight", d, -1});
        sp.push_back({p + "ln_2.bias", d, -1});
        sp.push_back({p + "attn.c_attn.weight", 3 * d, d});
        sp.push_back({p + "attn.c_attn.bias", 3 * d, -1});
        sp.push_back({p + "attn.c_proj.weight", d, d});
        sp.push_back({p + "attn.c_proj.bias", d, -1});
        sp.push_back({p + "mlp.c_fc.weight", F, d});
        sp.push_back({p + "mlp.c_fc.bias", F, -1});
        sp.push_back({p + "mlp.c_proj.weight", d, F});
        sp.push_back({p + "mlp.c_proj.bias", d, -1});
    }
    sp.push_back({"tfmr.ln_f.weight", d, -1});
    sp.push_back({"tfmr.ln_f.bias", d, -1});
    sp.push_back({"speech_head.weight", V, d});
    sp.push_back({"speech_head.bias", V, -1});
    sp.push_back({"speech_emb.weight", V, d});
    return check_specs(st, sp);
}

// ---- packed bf16 weight cache ---------------------------------------------------------------------
// The bf16 image of the streamed matrices (wbuf: 1.02 GB for the 520M model) exactly as the kernels read it, kept in a
// side file so the next t3mi_open reads half the bytes and converts nothing.  The header ties it to its source file
// (size + mtime in ns) and to the layout (variant, dims, version); anything that does not match, a short file or a
// checksum mismatch means "no cache": the weights are converted from the checkpoint again and the file is rewritten.
struct CacheHeader {
    char magic[8];           // "T3MIW16\0"
    uint32_t version, variant, n_layers, d, F, V;
    uint64_t src_size;
    int64_t src_mtime_ns;
    uint64_t src_fingerprint;   // checksum of the checkpoint's JSON header and of its first and last MiB of tensor data
    uint64_t payload_bytes, checksum;
};
constexpr uint32_t kCacheVersion = 2;

// four independent multiply-xor lanes over 32-byte groups (a single chain is latency-bound at ~1 GB/s; this runs at
// memory speed), folded at the end; every byte and its position enter the result
uint64_t payload_checksum(const uint8_t* p, size_t n) {
    uint64_t h[4] = {0x9E3779B97F4A7C15ULL, 0xBF58476D1CE4E5B9ULL, 0x94D049BB133111EBULL, 0x2545F4914F6CDD1DULL};
    size_t i = 0;
    for (; i + 32 <= n; i += 32) {
        uint64_t w[4];
        std::memcpy(w, p + i, 32);
        for (int k = 0; k < 4; ++k) {
            h[k] = (h[k] ^ w[k]) * 0x100000001B3ULL;
            h[k] = (h[k] << 23) | (h[k] >> 41);
        }
    }
    uint64_t r = n;
    for (; i < n; ++i) r = (r ^ p[i]) * 0x100000001B3ULL;
    for (int k = 0; k < 4; ++k) {
        r = (r ^ h[k]) * 0xFF51AFD7ED558CCDULL;
        r ^= r >> 33;
    }
    return r;
}

// ---- checkpoint -> HBM -----------------------------------------------------------------------
struct Uploader {
    t3mi_model* m;
    const t3mi::SafeTensorsFile& st;
    std::vector<float> host;
    DevBuf stage;  // fp32 staging on device for the bf16 conversion
    bool skip_w = false;   // the streamed matrices came from the packed cache: put_w only advances

    int get(const std::string& name, int64_t d0, int64_t d1, const t3mi::StTensor** out) {
        const t3mi::StTensor* t = st.find(name);
        if (!t) return fail(T3MI_ERR_IO, "checkpoint has no tensor '" + name + "'");
        const bool ok = (d1 < 0) ? (t->shape.size() == 1 && t->shape[0] == d0)
                                 : (t->shape.size() == 2 && t->shape[0] == d0 && t->shape[1] == d1);
        if (!ok) return fail(T3MI_ERR_IO, "tensor '" + name + "' has an unexpected shape");
        *out = t;
        return T3MI_OK;
    }
    // fp32 copy to dst (device)
    int put_f32(const std::string& name, int64_t d0, int64_t d1, float* dst) {
        const t3mi::StTensor* t = nullptr;
        int rc = get(name, d0, d1, &t);
        if (rc) return rc;
        const size_t n = (size_t)t->numel();
        std::string err;
        if (t->dtype == t3mi::StDtype::F32) {
            HIPCHK(hipMemcpy(dst, t->data, n * 4, hipMemcpyHostToDevice));
        } else {
            host.resize(n);
            if (!t3mi::SafeTensorsFile::to_f32(*t, host.data(), err)) return fail(T3MI_ERR_IO, name + ": " + err);
            HIPCHK(hipMemcpy(dst, host.data(), n * 4, hipMemcpyHostToDevice));
        }
        return T3MI_OK;
    }
    // host fp32 copy of a tensor
    int fetch(const std::string& name, int64_t d0, int64_t d1, std::vector<float>& out) {
        const t3mi::StTensor* t = nullptr;
        int rc = get(name, d0, d1, &t);
        if (rc) return rc;
        out.resize((size_t)t->numel());
        std::string err;
        if (t->dtype == t3mi::StDtype::F32) std::memcpy(out.data(), t->data, out.size() * 4);
        else if (!t3mi::SafeTensorsFile::to_f32(*t, out.data(), err)) return fail(T3MI_ERR_IO, name + ": " + err);
        return T3MI_OK;
    }
    // dst_raw[n] = bias[n]; dst_fold[n] = bias[n] + sum_k W'[n][k] * b_ln[k], W' = W as the kernels stream it
    // (rounded to bf16 when the model holds bf16 weights).  LayerNorm(x) W^T + b = rstd * ((x - mu) * g) W^T + fold.
    int put_folded_bias(const std::string& wname, const std::string& bname, int64_t N, int64_t K,
                        const std::vector<float>& b_ln, float* dst_raw, float* dst_fold) {
        std::vector<float> W, b;
        int rc;
        if ((rc = fetch(wname, N, K, W))) return rc;
        if ((rc = fetch(bname, N, -1, b))) return rc;
        std::vector<float> fold((size_t)N);
        const bool bf = m->wdtype == T3MI_W_BF16;
        for (int64_t n = 0; n < N; ++n) {
            double acc = 0.0;
            const float* wr = W.data() + (size_t)n * K;
            for (int64_t k = 0; k < K; ++k) {
                float w = wr[k];
                if (bf) {
                    uint32_t u;
                    std::memcpy(&u, &w, 4);
                    if ((u & 0x7FFFFFFFu) <= 0x7F800000u) u = (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u;
                    std::memcpy(&w, &u, 4);
                }
                acc += (double)w * (double)b_ln[k];
            }
            fold[n] = (float)((double)b[n] + acc);
        }
        HIPCHK(hipMemcpy(dst_raw, b.data(), (size_t)N * 4, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(dst_fold, fold.data(), (size_t)N * 4, hipMemcpyHostToDevice));
        return T3MI_OK;
    }
    // For the LayerNorm split of a residual GEMM y = resid + W a + b: dst[k] = sum_n W[n][k] / N (W as stored: bf16-rounded for a
    // bf16 model), bias_mean = mean(b) -- the mean of an output row is then resid's mean + a . dst + bias_mean
    int put_colmean(const std::string& wname, const std::string& bname, int64_t N, int64_t K, float* dst, float& bias_mean) {
        std::vector<float> W, b;
        int rc;
        if ((rc = fetch(wname, N, K, W))) return rc;
        if ((rc = fetch(bname, N, -1, b))) return rc;
        std::vector<double> acc((size_t)K + 1, 0.0);
        const bool bf = m->wdtype == T3MI_W_BF16;
        for (int64_t n = 0; n < N; ++n) {
            const float* wr = W.data() + (size_t)n * K;
            for (int64_t k = 0; k < K; ++k) {
                float w = wr[k];
                if (bf) {
                    uint32_t u;
                    std::memcpy(&u, &w, 4);
                    if ((u & 0x7FFFFFFFu) <= 0x7F800000u) u = (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u;
                    std::memcpy(&w, &u, 4);
                }
                acc[(size_t)k] += (double)w;
            }
            acc[(size_t)K] += (double)b[(size_t)n];
        }
        std::vector<float> out((size_t)K);
        for (int64_t k = 0; k < K; ++k) out[(size_t)k] = (float)(acc[(size_t)k] / (double)N);
        bias_mean = (float)(acc[(size_t)K] / (double)N);
        HIPCHK(hipMemcpy(dst, out.data(), (size_t)K * 4, hipMemcpyHostToDevice));
        return T3MI_OK;
    }
    // streamed matrix in the model's weight dtype; dst points into wbuf
    int put_w(const std::string& name, int64_t d0, int64_t d1, void* dst) {
        if (skip_w) return T3MI_OK;
        if (m->wdtype == T3MI_W_F32) return put_f32(name, d0, d1, static_cast<float*>(dst));
        const size_t n = (size_t)d0 * (size_t)d1;
        if (stage.bytes < n * 4) HIPCHK(stage.alloc(n * 4));
        int rc = put_f32(name, d0, d1, stage.as<float>());
        if (rc) return rc;
        HIPCHK(t3k::launch_f32_to_bf16(stage.as<float>(), dst, n, nullptr));
        HIPCHK(hipDeviceSynchronize());
        return T3MI_OK;
    }
};

// A fine-tune of the same architecture copied over the checkpoint with its mtime preserved has the size, mtime and dims of
// the original: the header also carries a content fingerprint -- the safetensors JSON header (names, shapes, offsets) plus
// the first and the last MiB of the file (tensor data), a few milliseconds to compute.
uint64_t source_fingerprint(const t3mi::SafeTensorsFile& st) {
    const uint8_t* base = st.base();
    const size_t n = st.file_size();
    if (!base || n == 0) return 0;
    const size_t mib = (size_t)1 << 20;
    uint64_t hl = 0;
    if (n >= 8) std::memcpy(&hl, base, 8);
    const size_t head = (size_t)std::min<uint64_t>((uint64_t)n, 8 + hl + mib);   // length word + JSON header + first MiB of data
    uint64_t f = payload_checksum(base, head);
    if (n > head) {
        const size_t tail = std::min(mib, n - head);
        f ^= payload_checksum(base + (n - tail), tail) * 0x9E3779B97F4A7C15ULL;
    }
    return f;
}

CacheHeader cache_header(const t3mi_model* m, const t3mi::SafeTensorsFile& st, size_t payload) {
    CacheHeader h;
    std::memset(&h, 0, sizeof(h));
    std::memcpy(h.magic, "T3MIW16", 8);
    h.version = kCacheVersion;
    h.variant = (uint32_t)m->variant;
    h.n_layers = (uint32_t)m->n_layers;
    h.d = (uint32_t)m->d;
    h.F = (uint32_t)m->F;
    h.V = (uint32_t)m->V;
    h.src_size = (uint64_t)st.file_size();
    h.src_mtime_ns = st.mtime_ns();
    h.src_fingerprint = source_fingerprint(st);
    h.payload_bytes = (uint64_t)payload;
    return h;
}

// true: wbuf now holds the cached bf16 matrices.  false: no usable cache (m->cache_state says why); never an error.
bool cache_read(t3mi_model* m, const t3mi::SafeTensorsFile& st, const char* cache_path, size_t payload) {
    m->cache_state = T3MI_CACHE_NONE;
    if (!cache_path || !cache_path[0] || m->wdtype != T3MI_W_BF16) return false;
    m->cache_state = T3MI_CACHE_WRITTEN;   // provisional: no file yet
    const int fd = ::open(cache_path, O_RDONLY);
    if (fd < 0) return false;
    m->cache_state = T3MI_CACHE_REPLACED;  // a file exists; anything wrong with it below means it is rewritten
    struct stat sb;
    bool ok = fstat(fd, &sb) == 0 && (uint64_t)sb.st_size == sizeof(CacheHeader) + (uint64_t)payload;   // not truncated, nothing appended
    void* map = ok ? mmap(nullptr, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, fd, 0) : MAP_FAILED;
    ::close(fd);
    if (map == MAP_FAILED) return false;
    const CacheHeader want = cache_header(m, st, payload);
    CacheHeader got;
    std::memcpy(&got, map, sizeof(got));
    const uint64_t sum = got.checksum;
    got.checksum = 0;
    const uint8_t* body = static_cast<const uint8_t*>(map) + sizeof(CacheHeader);
    ok = std::memcmp(&got, &want, sizeof(want)) == 0 && payload_checksum(body, payload) == sum;
    if (ok) ok = hipMemcpy(m->wbuf.p, body, payload, hipMemcpyHostToDevice) == hipSuccess;   // straight from the page cache
    munmap(map, (size_t)sb.st_size);
    if (ok) m->cache_state = T3MI_CACHE_USED;
    return ok;
}

// best effort: a cache that cannot be written (read-only directory, disk full) costs the next open the conversion
void cache_write(t3mi_model* m, const t3mi::SafeTensorsFile& st, const char* cache_path, size_t payload) {
    if (!cache_path || !cache_path[0] || m->wdtype != T3MI_W_BF16) return;
    std::vector<uint8_t> buf(payload);
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(buf.data(), m->wbuf.p, payload, hipMemcpyDeviceToHost) != hipSuccess) {
        m->cache_state = T3MI_CACHE_NONE;
        return;
    }
    CacheHeader h = cache_header(m, st, payload);
    h.checksum = payload_checksum(buf.data(), payload);
    const std::string tmp = std::string(cache_path) + ".tmp" + std::to_string((long long)getpid());
    FILE* f = std::fopen(tmp.c_str(), "wb");
    bool ok = f != nullptr;
    if (ok) {
        ok = std::fwrite(&h, sizeof(h), 1, f) == 1 && std::fwrite(buf.data(), 1, payload, f) == payload;
        ok = (std::fclose(f) == 0) && ok;
    }
    if (ok) ok = std::rename(tmp.c_str(), cache_path) == 0;   // atomic: a reader never sees a half-written cache
    if (!ok) {
        std::remove(tmp.c_str());
        m->cache_state = T3MI_CACHE_NONE;
    }
}

int load_llama(t3mi_model* m, const t3mi::SafeTensorsFile& st, const char* cache_path) {
    CkptDims cd;
    int rc = llama_dims(st, cd);   // every name, shape and dtype is checked before the first allocation
    if (rc) return rc;
    m->d = cd.d; m->F = cd.F; m->V = cd.V; m->max_pos = cd.max_pos; m->H = cd.H; m->n_layers = cd.L;
    const int L = cd.L;
    const size_t d = m->d, F = m->F, V = m->V;
    const size_t es = m->wdtype == T3MI_W_BF16 ? 2 : 4;
    const size_t per_layer = (3 * d * d + d * d + 2 * F * d + d * F) * es;
    const size_t total = per_layer * L + V * d * es;
    HIPCHK(m->wbuf.alloc(total));
    HIPCHK(m->small.alloc((2 * d * L + d) * 4));
    HIPCHK(m->emb.alloc(V * d * 4));
    HIPCHK(m->pos_emb.alloc((size_t)m->max_pos * d * 4));
    m->weight_bytes = (int64_t)total + (int64_t)(2 * d * L + d) * 4;

    Uploader up{m, st, {}, {}};
    up.skip_w = cache_read(m, st, cache_path, total);
    char* w = m->wbuf.as<char>();
    float* sm = m->small.as<float>();
    m->layers.resize(L);
    for (int l = 0; l < L; ++l) {
        const std::string p = "tfmr.layers." + std::to_string(l) + ".";
        LayerW& lw = m->layers[l];
        lw.ln1 = sm;
        if ((rc = up.put_f32(p + "input_layernorm.weight", d, -1, sm))) return rc;
        sm += d;
        lw.wqkv = w;
        if ((rc = up.put_w(p + "self_attn.q_proj.weight", d, d, w))) return rc;
        w += d * d * es;
        if ((rc = up.put_w(p + "self_attn.k_proj.weight", d, d, w))) return rc;
        w += d * d * es;
        if ((rc = up.put_w(p + "self_attn.v_proj.weight", d, d, w))) return rc;
        w += d * d * es;
        lw.wo = w;
        if ((rc = up.put_w(p + "self_attn.o_proj.weight", d, d, w))) return rc;
        w += d * d * es;
        lw.ln2 = sm;
        if ((rc = up.put_f32(p + "post_attention_layernorm.weight", d, -1, sm))) return rc;
        sm += d;
        lw.wgu = w;
        if ((rc = up.put_w(p + "mlp.gate_proj.weight", F, d, w))) return rc;
        w += F * d * es;
        if ((rc = up.put_w(p + "mlp.up_proj.weight", F, d, w))) return rc;
        w += F * d * es;
        lw.wdown = w;
        if ((rc = up.put_w(p + "mlp.down_proj.weight", d, F, w))) return rc;
        w += d * F * es;
    }
    m->norm_w = sm;
    if ((rc = up.put_f32("tfmr.norm.weight", d, -1, sm))) return rc;
    m->head_w = w;
    if ((rc = up.put_w("speech_head.weight", V, d, w))) return rc;
    if ((rc = up.put_f32("speech_emb.weight", V, d, m->emb.as<float>()))) return rc;
    if ((rc = up.put_f32("speech_pos_emb.emb.weight", m->max_pos, d, m->pos_emb.as<float>()))) return rc;
    if (!up.skip_w) cache_write(m, st, cache_path, total);
    return T3MI_OK;
}

// Turbo T3: GPT-2 medium backbone.  Keys and [out, in] linear layout as load_t3_turbo_weights copies them
// into nn_linear parameters (R/t3.R:1489-1551); tfmr.wpe / text_emb / cond_enc are not on the decode path.
int load_turbo(t3mi_model* m, const t3mi::SafeTensorsFile& st, const char* cache_path) {
    CkptDims cd;
    int rc = turbo_dims(st, cd);
    if (rc) return rc;
    m->d = cd.d; m->F = cd.F; m->V = cd.V; m->max_pos = cd.max_pos; m->H = cd.H; m->n_layers = cd.L;
    const int L = cd.L;
    const size_t d = m->d, F = m->F, V = m->V;
    const size_t es = m->wdtype == T3MI_W_BF16 ? 2 : 4;
    const size_t per_layer = (3 * d * d + d * d + F * d + d * F) * es;
    const size_t total = per_layer * L + V * d * es;
    // small: per layer ln1 w,b | ln2 w,b | b_qkv raw,fold | b_o | c_proj column means | b_fc raw,fold | b_down ;
    // ln_f w,b ; head bias raw,fold
    const size_t small_layer = 4 * d + 2 * 3 * d + d + d + 2 * F + d;
    const size_t small_total = small_layer * L + 2 * d + 2 * V;
    HIPCHK(m->wbuf.alloc(total));
    HIPCHK(m->small.alloc(small_total * 4));
    HIPCHK(m->emb.alloc(V * d * 4));
    m->weight_bytes = (int64_t)total + (int64_t)small_total * 4;

    Uploader up{m, st, {}, {}};
    up.skip_w = cache_read(m, st, cache_path, total);
    char* w = m->wbuf.as<char>();
    float* sm = m->small.as<float>();
    m->layers.resize(L);
    std::vector<float> bln;
    for (int l = 0; l < L; ++l) {
        const std::string p = "tfmr.h." + std::to_string(l) + ".";
        LayerW& lw = m->layers[l];
        lw.ln1 = sm;
        if ((rc = up.put_f32(p + "ln_1.weight", d, -1, sm))) return rc;
        sm += d;
        lw.ln1_b = sm;
        if ((rc = up.put_f32(p + "ln_1.bias", d, -1, sm))) return rc;
        sm += d;
        lw.ln2 = sm;
        if ((rc = up.put_f32(p + "ln_2.weight", d, -1, sm))) return rc;
        sm += d;
        lw.ln2_b = sm;
        if ((rc = up.put_f32(p + "ln_2.bias", d, -1, sm))) return rc;
        sm += d;
        lw.wqkv = w;
        if ((rc = up.put_w(p + "attn.c_attn.weight", 3 * d, d, w))) return rc;
        w += 3 * d * d * es;
        if ((rc = up.fetch(p + "ln_1.bias", d, -1, bln))) return rc;
        lw.b_qkv = sm;
        lw.b_qkv_f = sm + 3 * d;
        if ((rc = up.put_folded_bias(p + "attn.c_attn.weight", p + "attn.c_attn.bias", 3 * d, d, bln, sm, sm + 3 * d)))
            return rc;
        sm += 6 * d;
        lw.wo = w;
        if ((rc = up.put_w(p + "attn.c_proj.weight", d, d, w))) return rc;
        w += d * d * es;
        lw.b_o = sm;
        if ((rc = up.put_f32(p + "attn.c_proj.bias", d, -1, sm))) return rc;
        sm += d;
        lw.o_colmean = sm;
        if ((rc = up.put_colmean(p + "attn.c_proj.weight", p + "attn.c_proj.bias", d, d, sm, lw.o_bias_mean))) return rc;
        sm += d;
        lw.wgu = w;
        if ((rc = up.put_w(p + "mlp.c_fc.weight", F, d, w))) return rc;
        w += F * d * es;
        if ((rc = up.fetch(p + "ln_2.bias", d, -1, bln))) return rc;
        lw.b_fc = sm;
        lw.b_fc_f = sm + F;
        if ((rc = up.put_folded_bias(p + "mlp.c_fc.weight", p + "mlp.c_fc.bias", F, d, bln, sm, sm + F))) return rc;
        sm += 2 * F;
        lw.wdown = w;
        if ((rc = up.put_w(p + "mlp.c_proj.weight", d, F, w))) return rc;
        w += d * F * es;
        lw.b_down = sm;
        if ((rc = up.put_f32(p + "mlp.c_proj.bias", d, -1, sm))) return rc;
        sm += d;
    }
    m->norm_w = sm;
    if ((rc = up.put_f32("tfmr.ln_f.weight", d, -1, sm))) return rc;
    sm += d;
    m->norm_b = sm;
    if ((rc = up.put_f32("tfmr.ln_f.bias", d, -1, sm))) return rc;
    sm += d;
    m->head_w = w;
    if ((rc = up.put_w("speech_head.weight", V, d, w))) return rc;
    if ((rc = up.fetch("tfmr.ln_f.bias", d, -1, bln))) return rc;
    m->head_b = sm;
    m->head_b_f = sm + V;
    if ((rc = up.put_folded_bias("speech_head.weight", "speech_head.bias", V, d, bln, sm, sm + V))) return rc;
    if ((rc = up.put_f32("speech_emb.weight", V, d, m->emb.as<float>()))) return rc;
    if (!up.skip_w) cache_write(m, st, cache_path, total);
    return T3MI_OK;
}

// bf16 models keep a second, fragment-major image of every streamed matrix for the activation-stationary GEMM of
// batches of up to t3k::kAsMaxRows rows (batch_kernels.hip, gemm_as_kernel): +1.02 GB of HBM for the 520M model, built
// on the device in about a millisecond.  T3MI_AS=<mask> picks the GEMMs that use it (t3mi_model::as_mask; 0 = none, the LDS-staged kernels then serve those
// batches; the head is not in the default: measured no faster there).
int build_fragments(t3mi_model* m) {
    if (const char* e = std::getenv("T3MI_AS")) {
        m->as_mask = std::atoi(e) & 31;
        m->as_forced = true;
    }
    if (const char* e = std::getenv("T3MI_NORM_SPLIT")) m->norm_split = std::atoi(e);   // bit 1: post-attention norm, bit 2: pre-attention (Llama)
    if (const char* e = std::getenv("T3MI_LN1_ROWS")) sscanf(e, "%d,%d", &m->ln1_lo, &m->ln1_hi);
    if (const char* e = std::getenv("T3MI_PF")) {   // attention blocks[,max (item, head) blocks[,mask[,xcd-aware[,GEMM blocks[,max rows]]]]]
        int b = 0, mx = 128, mk = 31, xc = 1, gb = 128, mr = 16;
        if (sscanf(e, "%d,%d,%d,%d,%d,%d", &b, &mx, &mk, &xc, &gb, &mr) >= 1) {
            auto clampb = [](int v) { return v < 0 ? 0 : (v > 1024 ? 1024 : v & ~7); };
            m->pf_blocks = clampb(b); m->pf_gblocks = clampb(gb); m->pf_max_blocks = mx; m->pf_mask = mk; m->pf_xcd = xc; m->pf_max_rows = mr;
        }
    }
    if (m->wdtype != T3MI_W_BF16) m->pf_blocks = 0;   // the prefetch rides on the bf16 instantiations only
    if (m->pf_blocks > 0) HIPCHK(m->pf_sink.alloc((size_t)t3k::kPfMaxBlocks * 512 * 4));
    if (m->as_mask == 0) return T3MI_OK;
    const int d = m->d, F = m->F, V = m->V;
    const bool turbo = m->variant == T3MI_VARIANT_TURBO;
    if (d % 32 || F % 32) return T3MI_OK;
    const size_t es = m->wdtype == T3MI_W_BF16 ? 2 : 4;
    // every per-layer matrix has a whole number of 32-feature tiles, so its fragment image has its size; the head (last
    // in wbuf) is rounded up to one
    const size_t head_off = static_cast<const char*>(m->head_w) - m->wbuf.as<char>();
    const size_t need = head_off + t3k::weight_fragment_elems(V, d, t3k::EPI_STORE) * es;
    HIPCHK(hipDeviceSynchronize());   // the uploads and conversions into wbuf
    HIPCHK(m->wfrag.alloc(need));
    const int wdt = m->wdtype == T3MI_W_BF16 ? 1 : 0;
    auto put = [&](const void* w, int N, int K, int epi) -> hipError_t {
        return t3k::launch_weight_fragments(w, m->wfrag.as<char>() + (static_cast<const char*>(w) - m->wbuf.as<char>()), N, K, epi, wdt, m->stream);
    };
    for (const LayerW& lw : m->layers) {
        HIPCHK(put(lw.wqkv, 3 * d, d, t3k::EPI_ROPE_KV));
        HIPCHK(put(lw.wo, d, d, t3k::EPI_RESID));
        HIPCHK(put(lw.wgu, turbo ? F : 2 * F, d, turbo ? t3k::EPI_GELU : t3k::EPI_SWIGLU));
        HIPCHK(put(lw.wdown, d, F, t3k::EPI_RESID));
    }
    HIPCHK(put(m->head_w, V, d, t3k::EPI_STORE));
    HIPCHK(hipStreamSynchronize(m->stream));
    return T3MI_OK;
}

// ---- one transformer pass over the rows currently in sess->x ------------------------------------
#ifdef T3K_STAMPS
unsigned long long* g_stamp_buf = nullptr;   // [launch][T3K_STAMP_WAVES][T3K_STAMP_SLOTS]
int g_stamp_launch = 0;
unsigned long long* next_stamps() {
    if (!g_stamp_buf) return nullptr;
    return g_stamp_buf + (size_t)(g_stamp_launch++) * T3K_STAMP_WAVES * T3K_STAMP_SLOTS;
}
#define T3MI_SET_STAMPS(a) (a).stamps = next_stamps()
#else
#define T3MI_SET_STAMPS(a) do { } while (0)
#endif

struct StepTimer {  // optional per-launch events for t3mi_session_profile_step
    std::vector<hipEvent_t> ev;
    std::vector<int> kind;
    hipStream_t s;
    bool on = false;
    void mark(int k) {
        if (!on) return;
        hipEvent_t e;
        (void)hipEventCreate(&e);
        (void)hipEventRecord(e, s);
        ev.push_back(e);
        kind.push_back(k);
    }
};
enum { K_SAMPLER = 0, K_QKV, K_ATTN, K_OPROJ, K_GATEUP, K_DOWN, K_HEAD, K_COUNT };
const char* const kKindName[K_COUNT] = {"sampler", "qkv_rope_kv", "attn_decode", "o_proj_resid",
                                        "gate_up_swiglu", "down_resid", "head"};

int run_layers(t3mi_model* m, Session* s, StepTimer* tm) {
    const int d = m->d, F = m->F, H = m->H;
    const bool turbo = m->variant == T3MI_VARIANT_TURBO;
    const size_t layer_kv = (size_t)s->rows * H * s->max_len * 64;
    for (int l = 0; l < m->n_layers; ++l) {
        const LayerW& lw = m->layers[l];
        float* kc = s->kc.as<float>() + layer_kv * l;
        float* vc = s->vc.as<float>() + layer_kv * l;
        // K1: RMSNorm + QKV + RoPE + KV store   (Turbo: LayerNorm + c_attn + bias, no rotation)
        for (int r0 = 0; r0 < s->rows;) {
            const int rem = s->rows - r0, rg = rem >= 4 ? 4 : (rem >= 2 ? 2 : 1);
            t3k::GemvArgs a;
            a.W = lw.wqkv; a.x = s->x.as<float>() + (size_t)r0 * d; a.ln_w = lw.ln1; a.eps = m->eps;
            a.N = 3 * d; a.K = d; a.rows = rg;
            a.y = s->qbuf.as<float>() + (size_t)r0 * d; a.ldy = d;
            a.rowpos = s->rowpos.as<int>() + r0;
            a.rope_cur = s->rope_cur.as<float>() + (size_t)r0 * 128;
            a.kcache = kc + (size_t)r0 * H * s->max_len * 64;
            a.vcache = vc + (size_t)r0 * H * s->max_len * 64;
            a.n_heads = H; a.max_len = s->max_len;
            a.waves = m->tune[0]; a.rows_per_chunk = m->tune_rows[0];
            if (turbo) { a.layernorm = 1; a.bias = lw.b_qkv_f; a.rope = 0; }
            T3MI_SET_STAMPS(a);
            HIPCHK(t3k::launch_gemv(a, t3k::EPI_ROPE_KV, m->wdtype, m->stream));
            r0 += rg;
        }
        if (tm) tm->mark(K_QKV);
        // K2: attention over the cache, split along the keys; partials merged in K3's prologue
        {
            t3k::AttnArgs a;
            a.q = s->qbuf.as<float>(); a.kcache = kc; a.vcache = vc; a.rowpos = s->rowpos.as<int>();
            a.pm = s->pm.as<float>(); a.pl = s->pl.as<float>(); a.pacc = s->pacc.as<float>();
            a.rows = s->rows; a.n_heads = H; a.max_len = s->max_len;
            a.n_split_max = s->n_split; a.keys_per_split = s->kps;
            T3MI_SET_STAMPS(a);
            HIPCHK(t3k::launch_attn_decode(a, m->stream));
        }
        if (tm) tm->mark(K_ATTN);
        // K3: softmax-merge + o_proj + residual
        for (int r0 = 0; r0 < s->rows;) {
            const int rem = s->rows - r0, rg = rem >= 4 ? 4 : (rem >= 2 ? 2 : 1);
            t3k::GemvArgs a;
            a.W = lw.wo; a.N = d; a.K = d; a.rows = rg;
#ifdef T3K_DIAG_PLAIN_OPROJ   // measurement aid, separate diagnostic library only (make diag): o_proj as a plain GEMV
            a.x = s->qbuf.as<float>() + (size_t)r0 * d;
#else
            a.pm = s->pm.as<float>() + (size_t)r0 * H * s->n_split;
            a.pl = s->pl.as<float>() + (size_t)r0 * H * s->n_split;
            a.pacc = s->pacc.as<float>() + (size_t)r0 * H * s->n_split * 64;
#endif
            a.n_split_max = s->n_split; a.keys_per_split = s->kps;
            a.rowpos = s->rowpos.as<int>() + r0;
            a.y = s->x.as<float>() + (size_t)r0 * d; a.resid = a.y; a.ldy = d;
            a.waves = m->tune[1]; a.rows_per_chunk = m->tune_rows[1];
            a.bias = lw.b_o;
            T3MI_SET_STAMPS(a);
            HIPCHK(t3k::launch_gemv(a, t3k::EPI_RESID, m->wdtype, m->stream));
            r0 += rg;
        }
        if (tm) tm->mark(K_OPROJ);
        // K4: RMSNorm + gate/up + SwiGLU
        for (int r0 = 0; r0 < s->rows;) {
            const int rem = s->rows - r0, rg = rem >= 4 ? 4 : (rem >= 2 ? 2 : 1);
            t3k::GemvArgs a;
            a.W = lw.wgu; a.x = s->x.as<float>() + (size_t)r0 * d; a.ln_w = lw.ln2; a.eps = m->eps;
            a.N = turbo ? F : 2 * F; a.K = d; a.rows = rg;
            a.y = s->hbuf.as<float>() + (size_t)r0 * F; a.ldy = F;
            a.waves = m->tune[2]; a.rows_per_chunk = m->tune_rows[2];
            if (turbo) { a.layernorm = 1; a.bias = lw.b_fc_f; }   // LayerNorm + c_fc + gelu_tanh (R/t3.R:1303-1304)
            T3MI_SET_STAMPS(a);
            HIPCHK(t3k::launch_gemv(a, turbo ? t3k::EPI_GELU : t3k::EPI_SWIGLU, m->wdtype, m->stream));
            r0 += rg;
        }
        if (tm) tm->mark(K_GATEUP);
        // K5: down_proj + residual
        for (int r0 = 0; r0 < s->rows;) {
            const int rem = s->rows - r0, rg = rem >= 4 ? 4 : (rem >= 2 ? 2 : 1);
            t3k::GemvArgs a;
            a.W = lw.wdown; a.x = s->hbuf.as<float>() + (size_t)r0 * F; a.N = d; a.K = F; a.rows = rg;
            a.y = s->x.as<float>() + (size_t)r0 * d; a.resid = a.y; a.ldy = d;
            a.waves = m->tune[3]; a.rows_per_chunk = m->tune_rows[3];
            a.bias = lw.b_down;
            T3MI_SET_STAMPS(a);
            HIPCHK(t3k::launch_gemv(a, t3k::EPI_RESID, m->wdtype, m->stream));
            r0 += rg;
        }
        if (tm) tm->mark(K_DOWN);
    }
    return T3MI_OK;
}

// ---- many-row path: the same layer as MFMA GEMMs over M items (batched decode rows, or the flat list of
// (row, position) prompt items of a prefill).  posmap[m] = cache slot, rowmap[m] = cache row (nullptr = m).
struct GemmCtx {
    int M = 0;
    float* x = nullptr;        // [M][d] residual stream
    float* q = nullptr;        // [M][d]
    void* xn = nullptr;        // [M][d] activation type
    void* attn_out = nullptr;  // [M][d]
    void* hact = nullptr;      // [M][F]
    const int* posmap = nullptr;
    const int* rowmap = nullptr;
    float *pm = nullptr, *pl = nullptr, *pacc = nullptr;
    int kps = 64, n_split = 1;
    float* row_ss = nullptr;   // split norm scratch (Session::rowss), or nullptr: every norm is a launch
    float* row_mu = nullptr;   // [M] row means of x as ln_1 saw it (Turbo's LayerNorm form of the split)
    // K-split residual GEMM whose partial slabs are still to be folded into x (by the next norm launch)
    int pend_ks = 0;
    const float* pend_bias = nullptr;
    // bf16 weight storage with fp32 activation rows (exact fp32 products, the arithmetic of the GEMV path): the prefill
    // of sessions that decode with <= 4 rows
    bool act_f32 = false;
};

int run_layers_gemm(t3mi_model* m, Session* s, GemmCtx& c, StepTimer* tm) {
    const int d = m->d, F = m->F, H = m->H;
    const int abf = (m->wdtype == T3MI_W_BF16 && !c.act_f32) ? 1 : 0;
    const int wdt = (m->wdtype == T3MI_W_BF16 && c.act_f32) ? 2 : m->wdtype;   // launch_gemm's operand-type selector
    const bool turbo = m->variant == T3MI_VARIANT_TURBO;
    const bool as_ok = wdt == m->wdtype;   // not the mixed form (bf16 weights x fp32 activation rows): the fragment image serves one operand type
    const size_t layer_kv = (size_t)s->rows * H * s->max_len * 64 * (s->kv_bf16 ? 2 : 4);   // bytes
    hipStream_t st = m->stream;
    // Weight prefetch (t3k::PfArgs) while the (item, head) blocks of the attention launch leave CUs idle: that launch fetches o_proj's and
    // gate_up's matrices, o_proj's launch down_proj's, gate_up's launch the next layer's qkv -- each one or two launches ahead,
    // about an XCD's L2 in flight at a time.
    // pf_attn: the attention launch has idle CUs (at most pf_max_blocks (item, head) blocks) and fetches o_proj's and gate_up's
    // matrices; otherwise, up to pf_max_rows rows, the qkv launch does.
    const bool pf_any = m->pf_blocks > 0 && as_ok && c.M <= m->pf_max_rows && ((c.M * H) & 7) == 0;
    const bool pf_attn = pf_any && (long long)c.M * H <= m->pf_max_blocks && !(m->pf_mask & 32);
    const bool pf_q = pf_any && !pf_attn && (m->pf_mask & 16);
    const size_t pf_eb = 2;   // bf16 (as the instantiations that carry it)
    auto pf_region = [&](t3k::PfArgs& f, int slot, int bit, const void* w_frag, int N, int K, int epi, bool splittable, bool remap) {
        if (!(m->pf_mask & bit) || !w_frag) return;   // only what an activation-stationary launch will read (its fragment-major image)
        int gz = 1;
        const int nsub = t3k::gemm_as_plan(c.M, N, K, epi, splittable, &gz);
        if (nsub == 0) return;
        f.ptr[slot] = w_frag;
        f.bytes[slot] = (unsigned)((size_t)N * K * pf_eb);
        // the fragment-major image is laid out by 16-feature sub-tile over all of K: a block column of the consumer owns nsub of them
        const size_t unit = (size_t)nsub * 16 * K * pf_eb;
        if (m->pf_xcd && f.bytes[slot] % (8 * unit) == 0) {
            f.unit[slot] = (unsigned)unit;
            f.remap[slot] = remap ? 1 : 0;
        }
    };
    auto pf_arm = [&](t3k::PfArgs& f, int blocks) {
        if (f.ptr[0] == nullptr && f.ptr[1] == nullptr) return;
        f.n_blocks = blocks;
        f.sink = m->pf_sink.as<unsigned>();
    };
    bool ln1_done = false;
    for (int l = 0; l < m->n_layers; ++l) {
        const LayerW& lw = m->layers[l];
        char* kc = s->kc.as<char>() + layer_kv * l;
        char* vc = s->vc.as<char>() + layer_kv * l;
        const float* gp = s->gpart.as<float>();
        // The post-attention norm is split over its neighbours when both run on the activation-stationary kernel with K unsplit:
        // o_proj's epilogue writes w_ln * x (Turbo: w_ln * (x - mean)) in the activation type and per-tile sums of squares, gate_up /
        // c_fc scales its output rows by 1 / sqrt(mean + eps) -- which removes the norm launch (4-5 us per layer in the graph).
        // Turbo's LayerNorm needs the row mean before the product exists: mean(x + W a + b) = mean(x) + a . colmean(W) + mean(b),
        // with mean(x) from ln_1's launch and the dot product from the attention launch (per head, from the values it stores); its
        // bias b_ln rides in c_fc's bias (b_fc_f, the copy the <= 4-row path uses).
        const void* gu_f = as_ok ? m->frag(lw.wgu, 4, c.M) : nullptr;
        const void* wo_f = as_ok ? m->frag(lw.wo, 2, c.M) : nullptr;
        bool split_ln2 = false;
        // (measured, Llama bf16, ms per step without -> with: 4 utterances 0.929 -> 0.875, 16: 1.056 -> 1.009, 32: 1.303 -> 1.293, 64: 1.804 ->
        // 1.773, 96: 2.349 -> 2.340, 128: 2.764 -> 2.762, 192: 3.741 -> 3.753: up to 128 rows)
        if ((m->norm_split & 1) && c.row_ss != nullptr && gu_f != nullptr && d % 32 == 0 && d / 32 <= 32 && c.M <= 128 && t3k::gemm_as_takes(c.M, d) &&
            (!turbo || (c.row_mu != nullptr && lw.o_colmean != nullptr && H <= 32 && (abf || c.M <= 32)))) {
            // (Turbo, S0 407, ms per step without -> with; bf16: 8 sentences 0.826 -> 0.787, 16: 0.870 -> 0.830, 32: 0.995 -> 0.953, 64: 1.321 ->
            // 1.300, 128: 1.991 -> 1.957; fp32: 8: 1.216 -> 1.216, 16: 1.272 -> 1.254, 32: 1.573 -> 1.557, 64: 2.345 -> 2.414: fp32 up to 32 rows)
            const void* any = m->frag(lw.wo, 2, c.M, true);   // o_proj below its tuned row range still pays for itself here
            if (any != nullptr) { wo_f = any; split_ln2 = true; }
        }
        // ln1_done: the previous layer's down_proj left w_ln1 * x in xn and the per-tile sums of squares in row_ss (below)
        if (turbo)
            HIPCHK(t3k::launch_layernorm_rows(c.x, lw.ln1, lw.ln1_b, m->eps, c.xn, c.M, d, abf, st, gp, c.pend_ks, c.pend_bias,
                                              split_ln2 ? c.row_mu : nullptr));
        else if (!ln1_done) HIPCHK(t3k::launch_rmsnorm_rows(c.x, lw.ln1, m->eps, c.xn, c.M, d, abf, st, gp, c.pend_ks, c.pend_bias));
        c.pend_ks = 0;
        {
            t3k::GemmArgs a;
            a.W = lw.wqkv; a.Wf = as_ok ? m->frag(lw.wqkv, 1, c.M) : nullptr; a.A = c.xn; a.M = c.M; a.N = 3 * d; a.K = d;
            a.posmap = c.posmap; a.rowmap = c.rowmap;
            a.rope_cos = m->rope_cos.as<float>(); a.rope_sin = m->rope_sin.as<float>();
            a.q = c.q; a.kcache = kc; a.vcache = vc; a.kv_bf16 = s->kv_bf16 ? 1 : 0; a.n_heads = H; a.max_len = s->max_len;
            if (turbo) { a.bias = lw.b_qkv; a.rope = 0; }
            if (ln1_done) { a.row_ss = c.row_ss; a.row_ss_tiles = d / 32; a.norm_eps = m->eps; }
            if (pf_q && a.Wf) {
                pf_region(a.pf, 0, 1, wo_f, d, d, t3k::EPI_RESID, true, split_ln2);
                pf_region(a.pf, 1, 2, gu_f, turbo ? F : 2 * F, d, turbo ? t3k::EPI_GELU : t3k::EPI_SWIGLU, false, false);
                pf_arm(a.pf, m->pf_gblocks);
            }
            HIPCHK(t3k::launch_gemm(a, t3k::EPI_ROPE_KV, wdt, st));
        }
        if (tm) tm->mark(K_QKV);
        {
            t3k::AttnRowsArgs a;
            a.q = c.q; a.kcache = kc; a.vcache = vc; a.posmap = c.posmap; a.rowmap = c.rowmap;
            a.out = c.attn_out; a.M = c.M; a.n_heads = H; a.max_len = s->max_len;
            a.kv_bf16 = s->kv_bf16 ? 1 : 0; a.out_bf16 = abf;
            if (split_ln2 && turbo) { a.dot_vec = lw.o_colmean; a.dot_out = c.row_mu + t3k::kAsMaxRows; }
            if (pf_attn) {   // the idle CUs fetch o_proj's and gate_up's weights
                pf_region(a.pf, 0, 1, wo_f, d, d, t3k::EPI_RESID, true, split_ln2);
                pf_region(a.pf, 1, 2, gu_f, turbo ? F : 2 * F, d, turbo ? t3k::EPI_GELU : t3k::EPI_SWIGLU, false, false);
                pf_arm(a.pf, m->pf_blocks);
            }
            HIPCHK(t3k::launch_attn_rows(a, st));
        }
        if (tm) tm->mark(K_ATTN);
        {
            t3k::GemmArgs a;
            a.W = lw.wo; a.Wf = wo_f; a.A = c.attn_out; a.M = c.M; a.N = d; a.K = d;
            a.y = c.x; a.resid = c.x; a.ldy = d; a.bias = lw.b_o;
            a.part = s->gpart.as<float>(); a.part_floats = s->gpart.bytes / 4;
            if (split_ln2) { a.next_act = c.xn; a.next_norm_w = lw.ln2; a.next_ss = c.row_ss; }
            if (split_ln2 && turbo) { a.resid_mean = c.row_mu; a.row_dot = c.row_mu + t3k::kAsMaxRows; a.row_dot_n = H; a.mean_shift = lw.o_bias_mean; }
            if (pf_any && m->pf_gblocks > 0 && wo_f) {
                pf_region(a.pf, 0, 4, m->frag(lw.wdown, 8, c.M), d, F, t3k::EPI_RESID, true, false);
                pf_arm(a.pf, m->pf_gblocks);
            }
            int ks = 1;
            a.ks_out = &ks;
            HIPCHK(t3k::launch_gemm(a, t3k::EPI_RESID, wdt, st));
            c.pend_ks = ks > 1 ? ks : 0;
            c.pend_bias = lw.b_o;
        }
        if (tm) tm->mark(K_OPROJ);
        if (!split_ln2) {
            if (turbo) HIPCHK(t3k::launch_layernorm_rows(c.x, lw.ln2, lw.ln2_b, m->eps, c.xn, c.M, d, abf, st, gp, c.pend_ks, c.pend_bias));
            else HIPCHK(t3k::launch_rmsnorm_rows(c.x, lw.ln2, m->eps, c.xn, c.M, d, abf, st, gp, c.pend_ks, c.pend_bias));
            c.pend_ks = 0;
        }
        {
            t3k::GemmArgs a;
            a.W = lw.wgu; a.Wf = gu_f; a.A = c.xn; a.M = c.M; a.N = turbo ? F : 2 * F; a.K = d; a.yact = c.hact;
            a.bias = split_ln2 ? lw.b_fc_f : lw.b_fc;   // Turbo: with the split, b_ln arrives as W . b_ln in the bias (Llama: both null)
            if (split_ln2) { a.row_ss = c.row_ss; a.row_ss_tiles = d / 32; a.norm_eps = m->eps; }
            if (pf_any && m->pf_gblocks > 0 && gu_f && l + 1 < m->n_layers) {
                pf_region(a.pf, 0, 8, m->frag(m->layers[l + 1].wqkv, 1, c.M), 3 * d, d, t3k::EPI_ROPE_KV, false, false);
                pf_arm(a.pf, m->pf_gblocks);
            }
            HIPCHK(t3k::launch_gemm(a, turbo ? t3k::EPI_GELU : t3k::EPI_SWIGLU, wdt, st));
        }
        if (tm) tm->mark(K_GATEUP);
        {
            t3k::GemmArgs a;
            a.W = lw.wdown; a.Wf = as_ok ? m->frag(lw.wdown, 8, c.M) : nullptr; a.A = c.hact; a.M = c.M; a.N = d; a.K = F;
            a.y = c.x; a.resid = c.x; a.ldy = d; a.bias = lw.b_down;
            a.part = s->gpart.as<float>(); a.part_floats = s->gpart.bytes / 4;
            // Llama's pre-attention norm of the NEXT layer split like the post-attention one: down_proj with K unsplit (16-row chunks, all
            // of K in LDS) writes x, w_ln1 * x in the activation type and per-tile sums of squares; the next qkv scales its rows before
            // RoPE.  Removes the norm launch and down_proj's partial slabs -- and measured SLOWER (64 utterances 1.758 -> 1.770 ms per step,
            // 16 utterances 1.007 -> 1.074: down_proj 6.7 -> 10.1 us, qkv 6.5 -> 7.5, against 4.9 for the norm launch), so it is off by default.
            ln1_done = (m->norm_split & 2) && !turbo && abf && c.row_ss != nullptr && l + 1 < m->n_layers && F == 4096 && d % 32 == 0 &&
                       d / 32 <= 32 && c.M >= m->ln1_lo && c.M <= m->ln1_hi && c.M <= 128 && a.Wf != nullptr &&
                       m->frag(m->layers[l + 1].wqkv, 1, c.M) != nullptr;
            if (ln1_done) { a.part = nullptr; a.part_floats = 0; a.next_act = c.xn; a.next_norm_w = m->layers[l + 1].ln1; a.next_ss = c.row_ss; }
            int ks = 1;
            a.ks_out = &ks;
            HIPCHK(t3k::launch_gemm(a, t3k::EPI_RESID, wdt, st));
            c.pend_ks = ks > 1 ? ks : 0;
            c.pend_bias = lw.b_down;
        }
        if (tm) tm->mark(K_DOWN);
    }
    return T3MI_OK;   // c.pend_ks > 0: the last down_proj is still to be folded (run_head's norm, or fold_pending)
}

int fold_pending(t3mi_model* m, Session* s, GemmCtx& c) {
    if (c.pend_ks > 0) HIPCHK(t3k::launch_fold_partials(c.x, s->gpart.as<float>(), c.pend_ks, c.pend_bias, c.M, m->d, m->stream));
    c.pend_ks = 0;
    return T3MI_OK;
}

GemmCtx decode_ctx(Session* s) {
    GemmCtx c;
    c.M = s->rows; c.x = s->x.as<float>(); c.q = s->qbuf.as<float>();
    c.xn = s->xn.p; c.attn_out = s->attn_out.p; c.hact = s->hact.p;
    c.posmap = s->rowpos.as<int>(); c.rowmap = nullptr;
    c.pm = s->pm.as<float>(); c.pl = s->pl.as<float>(); c.pacc = s->pacc.as<float>();
    c.kps = s->kps; c.n_split = s->n_split;
    c.row_ss = s->rowss.as<float>();
    c.row_mu = s->rowmu;
    return c;
}

int run_head(t3mi_model* m, Session* s, StepTimer* tm) {
    const int d = m->d;
    const bool turbo = m->variant == T3MI_VARIANT_TURBO;
    if (s->gemm_mode) {
        const int abf = m->wdtype == T3MI_W_BF16 ? 1 : 0;
        if (turbo)
            HIPCHK(t3k::launch_layernorm_rows(s->x.as<float>(), m->norm_w, m->norm_b, m->eps, s->xn.p, s->rows, d, abf,
                                              m->stream, s->gpart.as<float>(), s->pend_ks, s->pend_bias));
        else
            HIPCHK(t3k::launch_rmsnorm_rows(s->x.as<float>(), m->norm_w, m->eps, s->xn.p, s->rows, d, abf, m->stream,
                                            s->gpart.as<float>(), s->pend_ks, s->pend_bias));
        s->pend_ks = 0;
        t3k::GemmArgs a;
        a.W = m->head_w; a.Wf = m->frag(m->head_w, 16, s->rows); a.A = s->xn.p; a.M = s->rows; a.N = m->V; a.K = d;
        a.y = s->logits.as<float>(); a.ldy = m->V; a.bias = m->head_b;
        HIPCHK(t3k::launch_gemm(a, t3k::EPI_STORE, m->wdtype, m->stream));
        if (tm) tm->mark(K_HEAD);
        return T3MI_OK;
    }
    for (int r0 = 0; r0 < s->rows;) {
        const int rem = s->rows - r0, rg = rem >= 4 ? 4 : (rem >= 2 ? 2 : 1);
        t3k::GemvArgs a;
        a.W = m->head_w; a.x = s->x.as<float>() + (size_t)r0 * d; a.ln_w = m->norm_w; a.eps = m->eps;
        a.N = m->V; a.K = d; a.rows = rg;
        a.y = s->logits.as<float>() + (size_t)r0 * m->V; a.ldy = m->V;
        a.waves = m->tune[4]; a.rows_per_chunk = m->tune_rows[4];
        if (turbo) { a.layernorm = 1; a.bias = m->head_b_f; }   // ln_f + speech_head with bias (R/t3.R:1444)
        T3MI_SET_STAMPS(a);
            HIPCHK(t3k::launch_gemv(a, t3k::EPI_STORE, m->wdtype, m->stream));
        r0 += rg;
    }
    if (tm) tm->mark(K_HEAD);
    return T3MI_OK;
}

int run_sampler(t3mi_model* m, Session* s, StepTimer* tm) {
    t3k::SamplerArgs a;
    const t3mi_params& p = s->prm;
    a.logits = s->logits.as<float>(); a.V = m->V; a.rpu = s->rpu; a.n_utt = s->n_utt;
    a.cfg_w = p.cfg_weight; a.temperature = p.temperature; a.top_p = p.top_p; a.min_p = p.min_p;
    a.penalty = p.repetition_penalty; a.variant = p.sampler; a.top_k = p.top_k; a.stop_token = p.stop_token;
    a.mask_stop = p.mask_stop; a.max_new_tokens = p.max_new_tokens; a.seed = p.seed;
    a.utt_ids = s->utt_ids.as<unsigned long long>(); a.max_new_u = s->max_new_u.as<int>();
    a.cache_cap = p.max_cache_len > 0 ? p.max_cache_len : 0;
    a.cap_extra = m->variant == T3MI_VARIANT_TURBO ? 1 : 0;
    a.start_token = p.start_token;
    a.seen = s->seen.as<uint8_t>(); a.tokens = s->tokens.as<int>(); a.cap = s->cap;
    a.n_tokens = s->n_tokens.as<int>(); a.done = s->done.as<int>(); a.rowpos = s->rowpos.as<int>();
    a.s0 = s->s0_dev.as<int>();
    a.forced = s->n_forced > 0 ? s->forced.as<int>() : nullptr; a.n_forced = s->n_forced;
    a.emb = m->emb.as<float>(); a.d = m->d; a.max_pos = m->max_pos;
    a.pos_emb = m->variant == T3MI_VARIANT_TURBO ? nullptr : m->pos_emb.as<float>();
    a.max_len = s->max_len; a.x = s->x.as<float>();
    a.rope_cos = m->rope_cos.as<float>(); a.rope_sin = m->rope_sin.as<float>(); a.rope_cur = s->rope_cur.as<float>();
    a.dist_ids = s->dist_ids.as<int>(); a.dist_probs = s->dist_probs.as<float>(); a.dist_n = s->dist_n.as<int>();
    a.drawn = s->drawn.as<int>();
    a.dist_cap = m->V;
    a.filt = s->filt.p ? s->filt.as<float>() : nullptr;
    HIPCHK(t3k::launch_sampler(a, m->stream));
    if (tm) tm->mark(K_SAMPLER);
    return T3MI_OK;
}

int run_decode_step(t3mi_model* m, Session* s, StepTimer* tm) {
    int rc;
    if (!(rc = run_sampler(m, s, tm))) {
        if (s->gemm_mode) {
            GemmCtx c = decode_ctx(s);
            rc = run_layers_gemm(m, s, c, tm);
            s->pend_ks = c.pend_ks;
            s->pend_bias = c.pend_bias;
        }
        else rc = run_layers(m, s, tm);
    }
    if (!rc) rc = run_head(m, s, tm);
    return rc;
}

int build_graph(t3mi_model* m, Session* s) {
    s->graph_tried = true;
    HIPCHK(hipStreamBeginCapture(m->stream, hipStreamCaptureModeThreadLocal));
    int rc = run_decode_step(m, s, nullptr);
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture(m->stream, &g);
    if (rc) {
        if (g) (void)hipGraphDestroy(g);
        return rc;
    }
    if (e != hipSuccess) return fail(T3MI_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
    s->graph = g;
    HIPCHK(hipGraphInstantiate(&s->gexec, g, nullptr, nullptr, 0));
    return T3MI_OK;
}

// ---- prefill of a list of session rows (R/t3.R:505): all prompt positions of the listed rows as ONE list of (row, position) items
// through the MFMA GEMM path (run_layers_gemm with posmap / rowmap), the last position's hidden state of each row left in s->x.
// rows_list: ascending session rows (row = utterance * rpu + r); row_ptr[i]: device address of the [s0, d] prompt of rows_list[i];
// the prompt length of a row is s->s0[row / rpu].  Rows are prefilled in groups of at most kPrefillItems items (a row only attends
// to its own cache row, so groups are independent): the scratch stays bounded however many prompts a batch holds.
int prefill_rows(t3mi_model* m, Session* s, const std::vector<int>& rows_list, const std::vector<const float*>& row_ptr, bool pf_act_f32) {
    constexpr size_t kPrefillItems = 32768;
    const size_t d = m->d, F = m->F;
    const int rpu = s->rpu;
    hipStream_t st = m->stream;
    const int nr = (int)rows_list.size();
    std::vector<std::pair<int, int>> groups;   // [begin, end) into rows_list
    size_t max_items = 0;
    for (int i0 = 0; i0 < nr;) {
        size_t n = 0;
        int i1 = i0;
        while (i1 < nr && (i1 == i0 || n + (size_t)s->s0[rows_list[i1] / rpu] <= kPrefillItems)) n += (size_t)s->s0[rows_list[i1++] / rpu];
        groups.emplace_back(i0, i1);
        max_items = std::max(max_items, n);
        i0 = i1;
    }
    // scratch of the prefill: kept in the session and only ever grown (t3mi_session_refill prefills again and again; hipFree
    // synchronises the device), released by t3mi_session_prefill_done / the session's end
    DevBuf &xpf = s->pfs[0], &qpf = s->pfs[1], &xnpf = s->pfs[2], &aopf = s->pfs[3], &hpf = s->pfs[4], &dsrc = s->pfs[5], &dpos = s->pfs[6],
           &drow = s->pfs[7], &dlast = s->pfs[8];
    const size_t as = (m->wdtype == T3MI_W_BF16 && !pf_act_f32) ? 2 : 4;
    auto grow = [](DevBuf& b, size_t bytes) { return b.bytes >= bytes && b.p ? hipSuccess : b.alloc(bytes); };
    HIPCHK(grow(xpf, max_items * d * 4));
    HIPCHK(grow(qpf, max_items * d * 4));
    HIPCHK(grow(xnpf, max_items * d * as));
    HIPCHK(grow(aopf, max_items * d * as));
    HIPCHK(grow(hpf, max_items * F * as));
    HIPCHK(grow(dsrc, max_items * sizeof(void*)));
    HIPCHK(grow(dpos, max_items * 4));
    HIPCHK(grow(drow, max_items * 4));
    HIPCHK(grow(dlast, (size_t)nr * sizeof(void*)));
    std::vector<const float*> isrc, last;
    std::vector<int32_t> ipos, irow;
    int rc;
    for (const auto& grp : groups) {
        isrc.clear(); ipos.clear(); irow.clear(); last.clear();
        for (int i = grp.first; i < grp.second; ++i) {
            const int r = rows_list[i], u = r / rpu;
            for (int t = 0; t < s->s0[u]; ++t) {
                isrc.push_back(row_ptr[i] + (size_t)t * d);
                ipos.push_back(t);
                irow.push_back(r);
            }
            last.push_back(xpf.as<float>() + (isrc.size() - 1) * d);
        }
        const size_t items = isrc.size();
        const int grows = grp.second - grp.first;
        HIPCHK(hipMemcpyAsync(dsrc.p, isrc.data(), items * sizeof(void*), hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(dpos.p, ipos.data(), items * 4, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(drow.p, irow.data(), items * 4, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(dlast.p, last.data(), (size_t)grows * sizeof(void*), hipMemcpyHostToDevice, st));
        HIPCHK(t3k::launch_gather_rows(dsrc.as<const float*>(), xpf.as<float>(), (int)items, m->d, st));
        GemmCtx c;
        c.M = (int)items; c.x = xpf.as<float>(); c.q = qpf.as<float>();
        c.xn = xnpf.p; c.attn_out = aopf.p; c.hact = hpf.p;
        c.posmap = dpos.as<int>(); c.rowmap = drow.as<int>();
        c.act_f32 = pf_act_f32;
        if ((rc = run_layers_gemm(m, s, c, nullptr))) return rc;
        if ((rc = fold_pending(m, s, c))) return rc;
        // the last prompt position of every row feeds the first speech-head evaluation: runs of consecutive session rows at a time
        for (int i = grp.first; i < grp.second;) {
            int j = i + 1;
            while (j < grp.second && rows_list[j] == rows_list[j - 1] + 1) ++j;
            HIPCHK(t3k::launch_gather_rows(dlast.as<const float*>() + (i - grp.first), s->x.as<float>() + (size_t)rows_list[i] * d, j - i, m->d, st));
            i = j;
        }
        HIPCHK(hipStreamSynchronize(st));  // the host vectors are refilled by the next group; the scratch dies after the last
    }
    return T3MI_OK;
}

int need_session(t3mi_model* m) {
    if (!m) return fail(T3MI_ERR_ARG, "null model handle");
    if (!m->sess) return fail(T3MI_ERR_STATE, "no active session (call t3mi_session_begin first)");
    return T3MI_OK;
}

}  // namespace

extern "C" {

void t3mi_default_params(t3mi_params* p) {
    if (!p) return;
    std::memset(p, 0, sizeof(*p));
    p->max_new_tokens = 1000;
    p->temperature = 0.8f;
    p->cfg_weight = 0.5f;
    p->top_p = 0.95f;
    p->min_p = 0.05f;
    p->repetition_penalty = 1.2f;
    p->start_token = 6561;
    p->stop_token = 6562;
    p->sampler = T3MI_SAMPLER_R;
}

const char* t3mi_last_error(void) { return g_err.c_str(); }

int t3mi_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int t3mi_inspect(const char* path, int variant, t3mi_info* info) {
    if (!path || !info) return fail(T3MI_ERR_ARG, "t3mi_inspect: null argument");
    if (variant != T3MI_VARIANT_LLAMA && variant != T3MI_VARIANT_TURBO)
        return fail(T3MI_ERR_ARG, "t3mi_inspect: variant must be T3MI_VARIANT_LLAMA or T3MI_VARIANT_TURBO");
    t3mi::SafeTensorsFile st;
    std::string err;
    if (!st.open(path, err)) return fail(T3MI_ERR_IO, err);
    CkptDims c;
    int rc = variant == T3MI_VARIANT_TURBO ? turbo_dims(st, c) : llama_dims(st, c);
    if (rc) return rc;
    t3mi::PromptModel* pm = nullptr;
    const int prc = t3mi::prompt_load(st, variant, c.d, &pm, err, /*validate_only=*/true);
    if (prc < 0) return fail(prc, err);
    std::memset(info, 0, sizeof(*info));
    info->variant = variant;
    info->n_layers = c.L;
    info->d_model = c.d;
    info->n_heads = c.H;
    info->head_dim = 64;
    info->d_ff = c.F;
    info->vocab = c.V;
    info->max_pos = c.max_pos;
    info->device = -1;
    info->has_prompt_tensors = prc == 0 ? 1 : 0;
    return T3MI_OK;
}

int t3mi_open(const char* path, int variant, int device, int wdtype, t3mi_model** out) {
    return t3mi_open_cached(path, variant, device, wdtype, nullptr, out);
}

int t3mi_open_cached(const char* path, int variant, int device, int wdtype, const char* cache_path, t3mi_model** out) {
    if (!path || !out) return fail(T3MI_ERR_ARG, "t3mi_open: null argument");
    *out = nullptr;
    if (variant != T3MI_VARIANT_LLAMA && variant != T3MI_VARIANT_TURBO)
        return fail(T3MI_ERR_ARG, "t3mi_open: variant must be T3MI_VARIANT_LLAMA or T3MI_VARIANT_TURBO");
    if (wdtype != T3MI_W_F32 && wdtype != T3MI_W_BF16) return fail(T3MI_ERR_ARG, "t3mi_open: bad wdtype");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(T3MI_ERR_HIP, "t3mi_open: no HIP device visible (the MI355X path has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(T3MI_ERR_ARG, "t3mi_open: device index out of range");
    HIPCHK(hipSetDevice(device));
    t3mi::SafeTensorsFile st;
    std::string err;
    if (!st.open(path, err)) return fail(T3MI_ERR_IO, err);
    std::unique_ptr<t3mi_model> m(new t3mi_model());
    m->device = device;
    m->variant = variant;
    m->wdtype = wdtype;
    const char* g = std::getenv("T3MI_GRAPH");
    m->use_graph = !(g && g[0] == '0');
    if (const char* tw = std::getenv("T3MI_WAVES")) {  // tuning knob, see DESIGN.md
        int v[5];
        if (std::sscanf(tw, "%d,%d,%d,%d,%d", &v[0], &v[1], &v[2], &v[3], &v[4]) == 5)
            for (int i = 0; i < 5; ++i)
                if (v[i] >= 1 && v[i] <= 8) m->tune[i] = v[i];
    }
    if (const char* tr = std::getenv("T3MI_ROWS")) {
        int v[5];
        if (std::sscanf(tr, "%d,%d,%d,%d,%d", &v[0], &v[1], &v[2], &v[3], &v[4]) == 5)
            for (int i = 0; i < 5; ++i)
                if (v[i] == 1 || v[i] == 2 || v[i] == 4) m->tune_rows[i] = v[i];
    }
    HIPCHK(t3k::init_device());
    HIPCHK(static_scratch(device, &m->static_scratch));   // once per device, with that device current
    HIPCHK(hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking));
    int rc = variant == T3MI_VARIANT_TURBO ? load_turbo(m.get(), st, cache_path) : load_llama(m.get(), st, cache_path);
    if (!rc) rc = build_fragments(m.get());
    if (rc) {
        (void)hipStreamDestroy(m->stream);
        return rc;
    }
    {
        std::string perr;
        const int prc = t3mi::prompt_load(st, variant, m->d, &m->prompt, perr);
        if (prc < 0) {
            (void)hipStreamDestroy(m->stream);
                return fail(prc, perr);
        }
    }
    if (t3k::sampler_lds_bytes(m->V) > 160 * 1024 || m->V > 16384) {
        (void)hipStreamDestroy(m->stream);
        return fail(T3MI_ERR_IO, "vocabulary too large for the single-workgroup sampler");
    }
    HIPCHK(hipDeviceSynchronize());
    *out = m.release();
    return T3MI_OK;
}

int t3mi_close(t3mi_model* m) {
    if (!m) return T3MI_OK;
    (void)hipSetDevice(m->device);
    if (m->stream) (void)hipStreamSynchronize(m->stream);
    m->sess.reset();
    t3mi::prompt_free(m->prompt);
    m->prompt = nullptr;
    if (m->stream) (void)hipStreamDestroy(m->stream);
    delete m;
    return T3MI_OK;
}

int t3mi_get_info(const t3mi_model* m, t3mi_info* info) {
    if (!m || !info) return fail(T3MI_ERR_ARG, "t3mi_get_info: null argument");
    info->variant = m->variant;
    info->n_layers = m->n_layers;
    info->d_model = m->d;
    info->n_heads = m->H;
    info->head_dim = 64;
    info->d_ff = m->F;
    info->vocab = m->V;
    info->max_pos = m->max_pos;
    info->wdtype = m->wdtype;
    info->device = m->device;
    info->weight_bytes = m->weight_bytes;
    info->cache_state = m->cache_state;
    info->has_prompt_tensors = m->prompt != nullptr ? 1 : 0;
    return T3MI_OK;
}

void* t3mi_stream(t3mi_model* m) { return m ? (void*)m->stream : nullptr; }

int t3mi_session_end(t3mi_model* m) {
    if (!m) return fail(T3MI_ERR_ARG, "null model handle");
    (void)hipSetDevice(m->device);
    if (m->stream) (void)hipStreamSynchronize(m->stream);
    m->sess.reset();
    return T3MI_OK;
}

int t3mi_session_begin(t3mi_model* m, int n_utt, int rpu, const float* const* embeds, const int32_t* s0,
                       const t3mi_params* p, int embeds_on_device) {
    if (!m || !embeds || !s0 || !p) return fail(T3MI_ERR_ARG, "t3mi_session_begin: null argument");
    if (n_utt <= 0) return fail(T3MI_ERR_ARG, "t3mi_session_begin: n_utt must be positive");
    if (rpu != 1 && rpu != 2) return fail(T3MI_ERR_ARG, "t3mi_session_begin: rows per utterance must be 1 or 2");
    if (p->cfg_weight > 0.f && rpu != 2)
        return fail(T3MI_ERR_ARG, "cfg_weight > 0 needs the cond/uncond row pair (rows = 2, R/t3.R:480-491)");
    if (p->max_new_tokens <= 0) return fail(T3MI_ERR_ARG, "max_new_tokens must be positive");
    const bool turbo = m->variant == T3MI_VARIANT_TURBO;
    if (turbo) {
        // t3_inference_turbo: one row per utterance, no CFG, its own logit processors (R/t3.R:1578-1742)
        if (p->sampler != T3MI_SAMPLER_TURBO) return fail(T3MI_ERR_ARG, "the Turbo variant needs sampler = T3MI_SAMPLER_TURBO");
        if (rpu != 1 || p->cfg_weight > 0.f) return fail(T3MI_ERR_ARG, "the Turbo variant runs without CFG (rows = 1, cfg_weight = 0)");
        if (p->start_token < 0 || p->start_token >= m->V) return fail(T3MI_ERR_ARG, "start_token outside the vocabulary");
    } else {
        if (p->sampler != T3MI_SAMPLER_R && p->sampler != T3MI_SAMPLER_CPP)
            return fail(T3MI_ERR_ARG, "sampler must be T3MI_SAMPLER_R or T3MI_SAMPLER_CPP for the Llama variant");
        if (!(p->temperature > 0.f)) return fail(T3MI_ERR_ARG, "temperature must be > 0");
    }
    if (!(p->repetition_penalty > 0.f)) return fail(T3MI_ERR_ARG, "repetition_penalty must be > 0");
    if (p->stop_token < 0 || p->stop_token >= m->V) return fail(T3MI_ERR_ARG, "stop_token outside the vocabulary");
    if (p->drop_invalid_from < 0) return fail(T3MI_ERR_ARG, "drop_invalid_from must be >= 0");
    HIPCHK(hipSetDevice(m->device));
    if (m->sess) t3mi_session_end(m);

    std::unique_ptr<Session> s(new Session());
    s->n_utt = n_utt;
    s->rpu = rpu;
    s->rows = n_utt * rpu;
    s->prm = *p;
    s->s0.assign(s0, s0 + n_utt);
    int max_s0 = 0, min_s0 = 1 << 30;
    for (int u = 0; u < n_utt; ++u) {
        if (s0[u] <= 0) return fail(T3MI_ERR_ARG, "prompt length must be positive");
        if (!embeds[u]) return fail(T3MI_ERR_ARG, "null prompt pointer");
        max_s0 = std::max(max_s0, s0[u]);
        min_s0 = std::min(min_s0, s0[u]);
    }
    s->max_s0 = max_s0;
    int max_new = p->max_new_tokens;
    if (p->max_cache_len > 0) {
        // t3_inference_cpp: max_gen_tokens = min(max_new_tokens, max_cache_len - cond_len - 1)  (R/t3.R:996), per
        // utterance: the sampler applies each utterance's own clamp, the host sizes the token buffers for the largest
        if (p->max_cache_len - max_s0 - 1 <= 0) return fail(T3MI_ERR_ARG, "prompt does not fit max_cache_len (R/llama_traced.R:343-345)");
        max_new = std::min(max_new, p->max_cache_len - min_s0 - 1);
    }
    if (max_new + 1 > m->max_pos) max_new = m->max_pos - 1;  // learned speech pos-emb rows (R/t3.R:349)
    // Turbo samples one token from the prefill logits and then up to max_new_tokens more (R/t3.R:1614-1658)
    if (turbo) max_new += 1;
    s->prm.max_new_tokens = max_new;
    s->cap = max_new;
    s->max_len = ((p->max_cache_len > 0 ? p->max_cache_len : max_s0 + max_new) + 7) & ~7;
    int rc = ensure_rope(m, s->max_len);
    if (rc) return rc;

    const size_t d = m->d, F = m->F, V = m->V, rows = s->rows, U = n_utt;
    {
        const char* gm = std::getenv("T3MI_GEMM_ROWS");  // rows above which the MFMA path is used (default 4)
        const int thr = gm ? std::atoi(gm) : 4;
        s->gemm_mode = (int)rows > thr;
        const char* kve = std::getenv("T3MI_KV");        // "f32" keeps fp32 KV slabs in the many-row bf16 regime
        s->kv_bf16 = s->gemm_mode && m->wdtype == T3MI_W_BF16 && d % 256 == 0 && !(kve && kve[0] == 'f');
    }
    const size_t kv = (size_t)m->n_layers * rows * m->H * s->max_len * 64 * (s->kv_bf16 ? 2 : 4);
    HIPCHK(s->kc.alloc(kv));
    HIPCHK(s->vc.alloc(kv));
    HIPCHK(s->x.alloc(rows * d * 4));
    HIPCHK(s->qbuf.alloc(rows * d * 4));
    // attention split geometry: more blocks per (row, head) when few rows are live
    s->kps = (rows * m->H <= 512) ? 64 : ((rows * m->H <= 2048) ? 128 : 256);
    s->n_split = (s->max_len + s->kps - 1) / s->kps;
    HIPCHK(s->pm.alloc(rows * m->H * (size_t)s->n_split * 4));
    HIPCHK(s->pl.alloc(rows * m->H * (size_t)s->n_split * 4));
    HIPCHK(s->pacc.alloc(rows * m->H * (size_t)s->n_split * 64 * 4));
    HIPCHK(s->hbuf.alloc(rows * F * 4));
    {
        const size_t as = m->wdtype == T3MI_W_BF16 ? 2 : 4;
        HIPCHK(s->xn.alloc(rows * d * as));
        HIPCHK(s->attn_out.alloc(rows * d * as));
        HIPCHK(s->hact.alloc(rows * F * as));
        // K-split partial slabs: the decode rows of a many-row session, and the prompt items of a short prefill
        // (few row chunks: the residual GEMMs need the K split to fill the chip)
        size_t items = 0;
        for (int u = 0; u < n_utt; ++u) items += (size_t)rpu * s0[u];
        const size_t part_rows = std::max<size_t>(s->gemm_mode ? rows : 0, items <= 1024 ? items : 0);
        HIPCHK(s->gpart.alloc(part_rows ? 8 * part_rows * d * 4 : 16));
        HIPCHK(s->rowss.alloc((size_t)(d / 32 + 2 + 32) * t3k::kAsMaxRows * 4));
        s->rowmu = s->rowss.as<float>() + (size_t)(d / 32 + 1) * t3k::kAsMaxRows;
    }
    HIPCHK(s->logits.alloc(rows * V * 4));
    HIPCHK(s->rowpos.alloc(rows * 4));
    HIPCHK(s->rope_cur.alloc(rows * 128 * 4));
    HIPCHK(hipMemsetAsync(s->rope_cur.p, 0, rows * 128 * 4, m->stream));
    HIPCHK(s->start.alloc(rows * 4));
    HIPCHK(s->emb_rows.alloc(rows * sizeof(void*)));
    HIPCHK(s->seen.alloc(U * V));
    HIPCHK(s->tokens.alloc(U * (size_t)s->cap * 4));
    HIPCHK(s->n_tokens.alloc(U * 4));
    HIPCHK(s->done.alloc(U * 4));
    HIPCHK(s->s0_dev.alloc(U * 4));
    HIPCHK(s->dist_ids.alloc(U * V * 4));
    HIPCHK(s->dist_probs.alloc(U * V * 4));
    HIPCHK(s->dist_n.alloc(U * 4));
    HIPCHK(s->drawn.alloc(U * 4));
    HIPCHK(s->utt_ids.alloc(U * 8));
    HIPCHK(s->max_new_u.alloc(U * 4));
    hipStream_t st = m->stream;
    {
        std::vector<unsigned long long> ids(U);
        std::vector<int32_t> mn(U, max_new);
        for (size_t u = 0; u < U; ++u) ids[u] = (unsigned long long)p->first_utterance_id + u;
        HIPCHK(hipMemcpy(s->utt_ids.p, ids.data(), U * 8, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(s->max_new_u.p, mn.data(), U * 4, hipMemcpyHostToDevice));
    }
    // create_kv_cache zero-fills the slabs (R/llama_traced.R:276-283)
    HIPCHK(hipMemsetAsync(s->kc.p, 0, kv, st));
    HIPCHK(hipMemsetAsync(s->vc.p, 0, kv, st));
    HIPCHK(hipMemsetAsync(s->x.p, 0, rows * d * 4, st));
    HIPCHK(hipMemsetAsync(s->qbuf.p, 0, rows * d * 4, st));
    HIPCHK(hipMemsetAsync(s->pm.p, 0, s->pm.bytes, st));
    HIPCHK(hipMemsetAsync(s->pl.p, 0, s->pl.bytes, st));
    HIPCHK(hipMemsetAsync(s->pacc.p, 0, s->pacc.bytes, st));
    HIPCHK(hipMemsetAsync(s->hbuf.p, 0, rows * F * 4, st));
    HIPCHK(hipMemsetAsync(s->logits.p, 0, rows * V * 4, st));
    HIPCHK(hipMemsetAsync(s->seen.p, 0, U * V, st));
    HIPCHK(hipMemsetAsync(s->tokens.p, 0, U * (size_t)s->cap * 4, st));
    HIPCHK(hipMemsetAsync(s->n_tokens.p, 0, U * 4, st));
    HIPCHK(hipMemsetAsync(s->done.p, 0, U * 4, st));
    HIPCHK(hipMemsetAsync(s->dist_n.p, 0, U * 4, st));

    // prompts: right-align so every utterance finishes its prefill at the same step
    std::vector<const float*> row_ptr(rows);
    std::vector<int32_t> start(rows);
    if (embeds_on_device) {
        for (int u = 0; u < n_utt; ++u)
            for (int r = 0; r < rpu; ++r) {
                row_ptr[u * rpu + r] = embeds[u] + (size_t)r * s0[u] * d;
                start[u * rpu + r] = max_s0 - s0[u];
            }
    } else {
        size_t tot = 0;
        for (int u = 0; u < n_utt; ++u) tot += (size_t)rpu * s0[u] * d;
        HIPCHK(s->emb_data.alloc(tot * 4));
        size_t off = 0;
        for (int u = 0; u < n_utt; ++u) {
            const size_t n = (size_t)rpu * s0[u] * d;
            HIPCHK(hipMemcpyAsync(s->emb_data.as<float>() + off, embeds[u], n * 4, hipMemcpyHostToDevice, st));
            for (int r = 0; r < rpu; ++r) {
                row_ptr[u * rpu + r] = s->emb_data.as<float>() + off + (size_t)r * s0[u] * d;
                start[u * rpu + r] = max_s0 - s0[u];
            }
            off += n;
        }
    }
    HIPCHK(hipMemcpyAsync(s->emb_rows.p, row_ptr.data(), rows * sizeof(void*), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(s->start.p, start.data(), rows * 4, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(s->s0_dev.p, s->s0.data(), U * 4, hipMemcpyHostToDevice, st));
    if (p->sampler == T3MI_SAMPLER_R && p->repetition_penalty != 1.0f) {
        // Q3: generated_ids starts as the raw BOS id used as a 1-based column => 0-based column BOS-1
        // is penalised from the first step on (R/t3.R:509, 530-531)
        const int col = p->start_token - 1;
        if (col >= 0 && col < m->V)
            for (int u = 0; u < n_utt; ++u)
                HIPCHK(hipMemsetAsync(s->seen.as<uint8_t>() + (size_t)u * V + col, 1, 1, st));
    }
    HIPCHK(hipStreamSynchronize(st));  // row_ptr/start are stack-local

    // ---- prefill (R/t3.R:505): all prompt positions of all rows as ONE list of (row, position) items through
    // the MFMA GEMM path (causality = each item attends to cache slots <= its own position; Q12's -65504
    // mask contributes exactly 0 in fp32).  Sessions that decode on the GEMV path with bf16 weights keep fp32
    // activation rows here too (GemmBF16W), so prefill and decode share one arithmetic.  T3MI_PREFILL=token runs
    // the decode kernels one position at a time instead.
    const char* pfenv = std::getenv("T3MI_PREFILL");
    bool gemm_prefill = true;
    const bool pf_act_f32 = m->wdtype == T3MI_W_BF16 && !s->gemm_mode;
    if (pfenv && std::strcmp(pfenv, "token") == 0) gemm_prefill = false;
    if (pfenv && std::strcmp(pfenv, "gemm") == 0) gemm_prefill = true;
    if (s->kv_bf16) gemm_prefill = true;   // the one-position decode kernels write fp32 cache slots
    s->gemm_prefill = gemm_prefill; s->pf_act_f32 = pf_act_f32;
    if (gemm_prefill) {
        std::vector<int> all_rows(rows);
        for (size_t r = 0; r < rows; ++r) all_rows[r] = (int)r;
        if ((rc = prefill_rows(m, s.get(), all_rows, row_ptr, pf_act_f32))) return rc;
        {   // a big batch's prefill scratch (32768 items: 1.3 GB) goes back; a small one stays for t3mi_session_refill
            size_t held = 0;
            for (auto& b : s->pfs) held += b.bytes;
            if (held > (size_t)256 << 20)
                for (auto& b : s->pfs) b.release();
        }
    } else {
        for (int t = 0; t < max_s0; ++t) {
            t3k::EmbedArgs e;
            e.emb_rows = s->emb_rows.as<const float*>();
            e.start = s->start.as<int>();
            e.t = t; e.d = m->d; e.rows = s->rows;
            e.x = s->x.as<float>(); e.rowpos = s->rowpos.as<int>();
            e.rope_cos = m->rope_cos.as<float>(); e.rope_sin = m->rope_sin.as<float>(); e.rope_cur = s->rope_cur.as<float>();
            HIPCHK(t3k::launch_prefill_embed(e, st));
            // (the one-position path always uses the decode GEMV kernels, whatever the row count)
            if ((rc = run_layers(m, s.get(), nullptr))) return rc;
        }
    }
    if ((rc = run_head(m, s.get(), nullptr))) return rc;
    HIPCHK(hipStreamSynchronize(st));
    m->sess = std::move(s);
    return T3MI_OK;
}

int t3mi_session_force(t3mi_model* m, const int32_t* forced, int n_forced) {
    int rc = need_session(m);
    if (rc) return rc;
    Session* s = m->sess.get();
    if (!forced || n_forced <= 0) return fail(T3MI_ERR_ARG, "t3mi_session_force: empty token list");
    if (s->gexec) return fail(T3MI_ERR_STATE, "t3mi_session_force must precede the first decode call");
    HIPCHK(hipSetDevice(m->device));
    HIPCHK(s->forced.alloc((size_t)s->n_utt * n_forced * 4));
    HIPCHK(hipMemcpy(s->forced.p, forced, (size_t)s->n_utt * n_forced * 4, hipMemcpyHostToDevice));
    s->n_forced = n_forced;
    return T3MI_OK;
}

int t3mi_session_decode(t3mi_model* m, int n_steps) {
    int rc = need_session(m);
    if (rc) return rc;
    if (n_steps <= 0) return fail(T3MI_ERR_ARG, "n_steps must be positive");
    Session* s = m->sess.get();
    HIPCHK(hipSetDevice(m->device));
    if (m->use_graph && !s->graph_tried) {
        if ((rc = build_graph(m, s))) return rc;
    }
    for (int i = 0; i < n_steps; ++i) {
        if (s->gexec) HIPCHK(hipGraphLaunch(s->gexec, m->stream));
        else if ((rc = run_decode_step(m, s, nullptr))) return rc;
    }
    return T3MI_OK;
}

int t3mi_session_decode_timed(t3mi_model* m, int n_steps, float* ms) {
    int rc = need_session(m);
    if (rc) return rc;
    if (!ms) return fail(T3MI_ERR_ARG, "null ms");
    HIPCHK(hipSetDevice(m->device));
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    HIPCHK(hipEventRecord(e0, m->stream));
    rc = t3mi_session_decode(m, n_steps);
    if (!rc) {
        hipError_t e = hipEventRecord(e1, m->stream);
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        if (e == hipSuccess) e = hipEventElapsedTime(ms, e0, e1);
        if (e != hipSuccess) rc = fail(T3MI_ERR_HIP, std::string("event timing: ") + hipGetErrorString(e));
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return rc;
}

int t3mi_session_sync(t3mi_model* m) {
    int rc = need_session(m);
    if (rc) return rc;
    HIPCHK(hipSetDevice(m->device));
    HIPCHK(hipStreamSynchronize(m->stream));
    return T3MI_OK;
}

int t3mi_session_status(t3mi_model* m, int32_t* n_tokens, int32_t* done) {
    int rc = need_session(m);
    if (rc) return rc;
    Session* s = m->sess.get();
    HIPCHK(hipSetDevice(m->device));
    HIPCHK(hipStreamSynchronize(m->stream));
    if (n_tokens) HIPCHK(hipMemcpy(n_tokens, s->n_tokens.p, (size_t)s->n_utt * 4, hipMemcpyDeviceToHost));
    if (done) HIPCHK(hipMemcpy(done, s->done.p, (size_t)s->n_utt * 4, hipMemcpyDeviceToHost));
    return T3MI_OK;
}

int t3mi_session_tokens(t3mi_model* m, int utt, int32_t* out, int32_t cap, int32_t* n_out) {
    int rc = need_session(m);
    if (rc) return rc;
    Session* s = m->sess.get();
    if (utt < 0 || utt >= s->n_utt || !out || !n_out) return fail(T3MI_ERR_ARG, "t3mi_session_tokens: bad argument");
    HIPCHK(hipSetDevice(m->device));
    HIPCHK(hipStreamSynchronize(m->stream));
    int32_t n = 0;
    HIPCHK(hipMemcpy(&n, s->n_tokens.as<int>() + utt, 4, hipMemcpyDeviceToHost));
    n = std::min(n, (int32_t)s->cap);
    const int32_t k = std::min(n, cap);
    if (k > 0) HIPCHK(hipMemcpy(out, s->tokens.as<int>() + (size_t)utt * s->cap, (size_t)k * 4, hipMemcpyDeviceToHost));
    *n_out = k;
    return T3MI_OK;
}

int t3mi_session_logits(t3mi_model* m, float* out) {
    int rc = need_session(m);
    if (rc) return rc;
    if (!out) return fail(T3MI_ERR_ARG, "null out");
    Session* s = m->sess.get();
    HIPCHK(hipSetDevice(m->device));
    HIPCHK(hipStreamSynchronize(m->stream));
    HIPCHK(hipMemcpy(out, s->logits.p, (size_t)s->rows * m->V * 4, hipMemcpyDeviceToHost));
    return T3MI_OK;
}

int t3mi_session_set_logits(t3mi_model* m, const float* logits) {
    int rc = need_session(m);
    if (rc) return rc;
    if (!logits) return fail(T3MI_ERR_ARG, "t3mi_session_set_logits: null logits");
    Session* s = m->sess.get();
    HIPCHK(hipSetDevice(m->device));
    HIPCHK(hipStreamSynchronize(m->stream));   // the head launch that wrote the pending logits
    HIPCHK(hipMemcpy(s->logits.p, logits, (size_t)s->rows * m->V * 4, hipMemcpyHostToDevice));
    return T3MI_OK;
}

int t3mi_session_keep_filtered(t3mi_model* m) {
    int rc = need_session(m);
    if (rc) return rc;
    Session* s = m->sess.get();
    if (s->gexec) return fail(T3MI_ERR_STATE, "t3mi_session_keep_filtered must precede the first decode call");
    HIPCHK(hipSetDevice(m->device));
    if (!s->filt.p) {
        HIPCHK(s->filt.alloc((size_t)s->n_utt * m->V * 4));
        HIPCHK(hipMemset(s->filt.p, 0, s->filt.bytes));
    }
    return T3MI_OK;
}

int t3mi_session_filtered_logits(t3mi_model* m, int utt, float* out) {
    int rc = need_session(m);
    if (rc) return rc;
    Session* s = m->sess.get();
    if (utt < 0 || utt >= s->n_utt || !out) return fail(T3MI_ERR_ARG, "t3mi_session_filtered_logits: bad argument");
    if (!s->filt.p) return fail(T3MI_ERR_STATE, "t3mi_session_filtered_logits: call t3mi_session_keep_filtered before the first decode");
    HIPCHK(hipSetDevice(m->device));
    HIPCHK(hipStreamSynchronize(m->stream));
    HIPCHK(hipMemcpy(out, s->filt.as<float>() + (size_t)utt * m->V, (size_t)m->V * 4, hipMemcpyDeviceToHost));
    return T3MI_OK;
}

int t3mi_session_dist(t3mi_model* m, int utt, int32_t* ids, float* probs, int32_t cap, int32_t* n) {
    int rc = need_session(m);
    if (rc) return rc;
    Session* s = m->sess.get();
    if (utt < 0 || utt >= s->n_utt || !ids || !probs || !n) return fail(T3MI_ERR_ARG, "t3mi_session_dist: bad argument");
    HIPCHK(hipSetDevice(m->device));
    HIPCHK(hipStreamSynchronize(m->stream));
    int32_t k = 0;
    HIPCHK(hipMemcpy(&k, s->dist_n.as<int>() + utt, 4, hipMemcpyDeviceToHost));
    *n = k;
    k = std::min(std::min(k, cap), (int32_t)m->V);
    if (k > 0) {
        HIPCHK(hipMemcpy(ids, s->dist_ids.as<int>() + (size_t)utt * m->V, (size_t)k * 4, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(probs, s->dist_probs.as<float>() + (size_t)utt * m->V, (size_t)k * 4, hipMemcpyDeviceToHost));
    }
    return T3MI_OK;
}

int t3mi_session_drawn(t3mi_model* m, int32_t* drawn) {
    int rc = need_session(m);
    if (rc) return rc;
    if (!drawn) return fail(T3MI_ERR_ARG, "null out");
    Session* s = m->sess.get();
    HIPCHK(hipSetDevice(m->device));
    HIPCHK(hipStreamSynchronize(m->stream));
    HIPCHK(hipMemcpy(drawn, s->drawn.p, (size_t)s->n_utt * 4, hipMemcpyDeviceToHost));
    return T3MI_OK;
}

int t3mi_session_profile_step(t3mi_model* m, t3mi_kstat* stats, int cap, int* n_stats) {
    int rc = need_session(m);
    if (rc) return rc;
    if (!stats || !n_stats || cap < K_COUNT) return fail(T3MI_ERR_ARG, "t3mi_session_profile_step: need room for 7 entries");
    Session* s = m->sess.get();
    HIPCHK(hipSetDevice(m->device));
    HIPCHK(hipStreamSynchronize(m->stream));
    // cache lengths before the step, for the attention byte count
    std::vector<int32_t> ntok(s->n_utt), done(s->n_utt);
    HIPCHK(hipMemcpy(ntok.data(), s->n_tokens.p, (size_t)s->n_utt * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(done.data(), s->done.p, (size_t)s->n_utt * 4, hipMemcpyDeviceToHost));
    StepTimer tm;
    tm.s = m->stream;
    tm.on = true;
    tm.mark(-1);
    rc = run_decode_step(m, s, &tm);
    if (!rc) {
        hipError_t e = hipStreamSynchronize(m->stream);
        if (e != hipSuccess) rc = fail(T3MI_ERR_HIP, hipGetErrorString(e));
    }
    if (!rc) {
        const double es = m->wdtype == T3MI_W_BF16 ? 2.0 : 4.0;
        const double d = m->d, F = m->F, V = m->V;
        double kvb = 0;
        for (int u = 0; u < s->n_utt; ++u)
            if (!done[u]) kvb += (double)s->rpu * 2.0 * d * (s->s0[u] + ntok[u] + 1) * (s->kv_bf16 ? 2.0 : 4.0);
        for (int k = 0; k < K_COUNT; ++k) {
            std::memset(&stats[k], 0, sizeof(t3mi_kstat));
            std::snprintf(stats[k].name, sizeof(stats[k].name), "%s", kKindName[k]);
        }
        for (size_t i = 1; i < tm.ev.size(); ++i) {
            float us = 0.f;
            (void)hipEventElapsedTime(&us, tm.ev[i - 1], tm.ev[i]);
            t3mi_kstat& ks = stats[tm.kind[i]];
            ks.launches += 1;
            ks.total_us += us * 1000.f;
        }
        const double L = m->n_layers;
        stats[K_QKV].bytes = L * 3 * d * d * es;
        stats[K_ATTN].bytes = L * kvb;
        stats[K_OPROJ].bytes = L * d * d * es;
        stats[K_GATEUP].bytes = L * 2 * F * d * es;
        stats[K_DOWN].bytes = L * d * F * es;
        stats[K_HEAD].bytes = V * d * es + (double)s->rows * V * 4;
        stats[K_SAMPLER].bytes = (double)s->rows * V * 4;
        *n_stats = K_COUNT;
    }
    for (hipEvent_t e : tm.ev) (void)hipEventDestroy(e);
    return rc;
}

#ifdef T3K_STAMPS
// diagnostic build only: run ONE eager decode step with per-wave timestamps; out receives
// [n_launches][T3K_STAMP_WAVES][T3K_STAMP_SLOTS] u64 (100 MHz ticks); returns n_launches or < 0
int t3mi_debug_stamp_step(t3mi_model* m, unsigned long long* out, int max_launches) {
    int rc = need_session(m);
    if (rc) return rc;
    Session* s = m->sess.get();
    HIPCHK(hipSetDevice(m->device));
    HIPCHK(hipStreamSynchronize(m->stream));
    const size_t per = (size_t)T3K_STAMP_WAVES * T3K_STAMP_SLOTS;
    DevBuf buf;
    HIPCHK(buf.alloc((size_t)max_launches * per * 8));
    HIPCHK(hipMemset(buf.p, 0, (size_t)max_launches * per * 8));
    g_stamp_buf = buf.as<unsigned long long>();
    g_stamp_launch = 0;
    rc = run_decode_step(m, s, nullptr);
    g_stamp_buf = nullptr;
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(m->stream));
    const int n = g_stamp_launch;
    if (n > max_launches) return fail(T3MI_ERR_ARG, "stamp buffer too small");
    HIPCHK(hipMemcpy(out, buf.p, (size_t)n * per * 8, hipMemcpyDeviceToHost));
    return n;
}
#endif

// decode loop shared by the generate entry points: the session is open, its prefill enqueued
static int run_generate(t3mi_model* m, int n_utt, int32_t* out, int32_t cap, int32_t* n_out) {
    Session* s = m->sess.get();
    const int max_new = s->prm.max_new_tokens;
    if (cap < max_new) {
        t3mi_session_end(m);
        return fail(T3MI_ERR_ARG, "output capacity is smaller than max_new_tokens");
    }
    std::vector<int32_t> ntok(n_utt), done(n_utt);
    int issued = 0, rc = T3MI_OK;
    bool all_done = false;
    while (!all_done && issued < max_new) {
        const int chunk = std::min(32, max_new - issued);
        if ((rc = t3mi_session_decode(m, chunk))) break;
        issued += chunk;
        if ((rc = t3mi_session_status(m, ntok.data(), done.data()))) break;
        all_done = true;
        for (int u = 0; u < n_utt; ++u) all_done = all_done && done[u] != 0;
    }
    int warn = T3MI_OK;
    if (!rc && s->prm.drop_invalid_from > 0) {
        // generate()'s next step on the T3 output (R/tts.R:366): drop_invalid_tokens, here on the device before the copy
        DevBuf kept, nk;
        hipError_t e = kept.alloc((size_t)n_utt * s->cap * 4);
        if (e == hipSuccess) e = nk.alloc((size_t)n_utt * 4);
        if (e == hipSuccess)
            e = t3k::launch_drop_invalid(s->tokens.as<int>(), s->n_tokens.as<int>(), s->cap, s->prm.drop_invalid_from, kept.as<int>(),
                                         nk.as<int>(), n_utt, m->stream);
        std::vector<int32_t> nkh(n_utt);
        if (e == hipSuccess) e = hipMemcpyAsync(nkh.data(), nk.p, (size_t)n_utt * 4, hipMemcpyDeviceToHost, m->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(m->stream);
        for (int u = 0; u < n_utt && e == hipSuccess; ++u) {
            n_out[u] = std::min(nkh[u], cap);
            if (n_out[u] > 0)
                e = hipMemcpy(out + (size_t)u * cap, kept.as<int>() + (size_t)u * s->cap, (size_t)n_out[u] * 4, hipMemcpyDeviceToHost);
            if (done[u] == 3) warn = T3MI_WARN_CACHE_FULL;
        }
        if (e != hipSuccess) rc = fail(T3MI_ERR_HIP, std::string("drop_invalid_tokens: ") + hipGetErrorString(e));
    } else if (!rc) {
        for (int u = 0; u < n_utt && !rc; ++u) {
            rc = t3mi_session_tokens(m, u, out + (size_t)u * cap, cap, &n_out[u]);
            if (done[u] == 3) warn = T3MI_WARN_CACHE_FULL;
        }
    }
    std::string keep = g_err;
    t3mi_session_end(m);
    if (rc) {
        g_err = keep;
        return rc;
    }
    if (warn) g_err = "KV cache full, generation stopped (src/t3_decode.cpp:394-397)";
    return warn;
}

int t3mi_generate_batch(t3mi_model* m, int n_utt, int rpu, const float* const* embeds, const int32_t* s0,
                        const t3mi_params* p, int32_t* out, int32_t cap, int32_t* n_out) {
    if (!out || !n_out) return fail(T3MI_ERR_ARG, "t3mi_generate_batch: null output");
    int rc = t3mi_session_begin(m, n_utt, rpu, embeds, s0, p, 0);
    if (rc) return rc;
    return run_generate(m, n_utt, out, cap, n_out);
}

// ---- prompt builder entry points -----------------------------------------------------------
static t3mi::PromptEnv prompt_env(const t3mi_model* m, int start_token) {
    t3mi::PromptEnv e;
    e.variant = m->variant;
    e.d = m->d;
    e.vocab = m->V;
    e.max_pos = m->variant == T3MI_VARIANT_TURBO ? 0 : m->max_pos;
    e.speech_emb = m->emb.as<float>();
    e.speech_pos = m->variant == T3MI_VARIANT_TURBO ? nullptr : m->pos_emb.as<float>();
    e.start_token = start_token;
    e.stream = m->stream;
    return e;
}
static t3mi::PromptCond prompt_cond(const t3mi_cond* c) {
    t3mi::PromptCond pc;
    pc.speaker_emb = c->speaker_emb;
    pc.n_speaker = c->n_speaker;
    pc.cond_tokens = c->cond_prompt_speech_tokens;
    pc.n_cond = c->n_cond_tokens;
    pc.emotion = c->emotion_adv;
    pc.has_emotion = c->has_emotion_adv;
    return pc;
}

int t3mi_prompt_shape(t3mi_model* m, const t3mi_cond* cond, int n_text, float cfg_weight, int32_t* rows, int32_t* s0,
                      int32_t* len_cond) {
    if (!m || !cond || !rows || !s0 || !len_cond) return fail(T3MI_ERR_ARG, "t3mi_prompt_shape: null argument");
    std::string err;
    int r = 0, s = 0, lc = 0;
    const int rc = t3mi::prompt_shape(m->prompt, prompt_env(m, 0), prompt_cond(cond), n_text, cfg_weight, &r, &s, &lc, err);
    if (rc) return fail(rc, err);
    *rows = r;
    *s0 = s;
    *len_cond = lc;
    return T3MI_OK;
}

// builds the prompts of n_utt utterances into m->prompt_buf; ptrs[u] / s0[u] describe them
static int build_prompts(t3mi_model* m, int n_utt, const t3mi_cond* conds, const int32_t* const* text, const int32_t* n_text,
                         float cfg_weight, int start_token, std::vector<const float*>& ptrs, std::vector<int32_t>& s0, int* rpu) {
    if (!m || !conds || !text || !n_text || n_utt <= 0) return fail(T3MI_ERR_ARG, "prompt: null argument");
    HIPCHK(hipSetDevice(m->device));
    const t3mi::PromptEnv env = prompt_env(m, start_token);
    std::string err;
    ptrs.assign(n_utt, nullptr);
    s0.assign(n_utt, 0);
    std::vector<size_t> off(n_utt);
    size_t tot = 0, total_ids = 0;
    int rows = 1;
    for (int u = 0; u < n_utt; ++u) {
        int r = 0, lc = 0;
        total_ids += (size_t)std::max(conds[u].n_cond_tokens, 0) + (size_t)std::max(n_text[u], 0) + 3;
        const int rc = t3mi::prompt_shape(m->prompt, env, prompt_cond(&conds[u]), n_text[u], cfg_weight, &r, &s0[u], &lc, err);
        if (rc) return fail(rc, "utterance " + std::to_string(u) + ": " + err);
        rows = r;
        off[u] = tot;
        tot += (size_t)r * s0[u] * m->d;
    }
    // the previous call's prefill may still read the old buffer: finish it before the buffer is replaced
    HIPCHK(hipStreamSynchronize(m->stream));
    {
        const int rc = t3mi::prompt_batch_begin(m->prompt, total_ids, n_utt, err);
        if (rc) return fail(rc, err);
    }
    if (m->prompt_buf.bytes < tot * 4) HIPCHK(m->prompt_buf.alloc(tot * 4));
    for (int u = 0; u < n_utt; ++u) {
        float* dst = m->prompt_buf.as<float>() + off[u];
        const int rc = t3mi::prompt_build(m->prompt, env, prompt_cond(&conds[u]), text[u], n_text[u], cfg_weight, dst, err);
        if (rc) return fail(rc, "utterance " + std::to_string(u) + ": " + err);
        ptrs[u] = dst;
    }
    *rpu = rows;
    return T3MI_OK;
}

int t3mi_prepare_input_embeds(t3mi_model* m, const t3mi_cond* cond, const int32_t* text_tokens, int n_text, float cfg_weight,
                              int32_t start_token, float* out, size_t cap_floats) {
    if (!out || !cond) return fail(T3MI_ERR_ARG, "t3mi_prepare_input_embeds: null argument");
    std::vector<const float*> ptrs;
    std::vector<int32_t> s0;
    int rpu = 1;
    const int32_t* tt[1] = {text_tokens};
    const int32_t nt[1] = {n_text};
    int rc = build_prompts(m, 1, cond, tt, nt, cfg_weight, start_token, ptrs, s0, &rpu);
    if (rc) return rc;
    const size_t n = (size_t)rpu * s0[0] * m->d;
    if (cap_floats < n) return fail(T3MI_ERR_ARG, "t3mi_prepare_input_embeds: output buffer too small");
    HIPCHK(hipStreamSynchronize(m->stream));
    HIPCHK(hipMemcpy(out, ptrs[0], n * 4, hipMemcpyDeviceToHost));
    return T3MI_OK;
}

int t3mi_prepare_input_embeds_static(t3mi_model* m, const t3mi_cond* cond, const int32_t* text_tokens, int n_text,
                                     float cfg_weight, int32_t start_token, float* out_device, size_t cap_floats) {
    if (!m || !cond || !out_device) return fail(T3MI_ERR_ARG, "t3mi_prepare_input_embeds_static: null argument");
    HIPCHK(hipSetDevice(m->device));
    if (!m->static_scratch) return fail(T3MI_ERR_STATE, "t3mi_prepare_input_embeds_static: no working memory on this device");
    const t3mi::PromptEnv env = prompt_env(m, start_token);
    const t3mi::PromptCond pc = prompt_cond(cond);
    std::string err;
    int rows = 0, s0 = 0, lc = 0;
    int rc = t3mi::prompt_shape(m->prompt, env, pc, n_text, cfg_weight, &rows, &s0, &lc, err);
    if (rc) return fail(rc, err);
    if (cap_floats < (size_t)rows * s0 * m->d) return fail(T3MI_ERR_ARG, "t3mi_prepare_input_embeds_static: output buffer too small");
    {
        // the sibling entry point takes a HOST pointer in this slot: a host address here would fault the GPU inside a kernel, so
        // it is refused up front (a query, no allocation)
        hipPointerAttribute_t at;
        std::memset(&at, 0, sizeof(at));
        const hipError_t pe = hipPointerGetAttributes(&at, out_device);
        if (pe != hipSuccess || at.type != hipMemoryTypeDevice || at.device != m->device) {
            (void)hipGetLastError();
            return fail(T3MI_ERR_ARG, "t3mi_prepare_input_embeds_static: out_device must be device memory on the model's GPU");
        }
    }
    size_t need_rows = 0, need_ids = 0;
    t3mi::prompt_scratch_need(m->prompt, env, pc, n_text, &need_rows, &need_ids);
    if (need_rows > kStaticRowsFloats || need_ids > kStaticIds || (size_t)pc.n_speaker > kStaticSpk)
        return fail(T3MI_ERR_ARG, "t3mi_prepare_input_embeds_static: inputs exceed the fixed working memory (" +
                                      std::to_string(need_ids) + " ids, limit " + std::to_string(kStaticIds) + "; " +
                                      std::to_string(need_rows) + " scratch floats, limit " + std::to_string(kStaticRowsFloats) +
                                      "; " + std::to_string(pc.n_speaker) + " speaker floats, limit " + std::to_string(kStaticSpk) +
                                      "): use t3mi_prepare_input_embeds");
    // m->static_scratch is the DEVICE address of this GPU's `scratch`; the array's name is never used below
    unsigned char* base = static_cast<unsigned char*>(m->static_scratch);
    t3mi::PromptScratch ws;
    ws.rows = reinterpret_cast<float*>(base);
    ws.rows_floats = kStaticRowsFloats;
    ws.ids = reinterpret_cast<int*>(base + kStaticRowsFloats * 4);
    ws.ids_cap = kStaticIds;
    ws.spk = reinterpret_cast<float*>(base + kStaticRowsFloats * 4 + kStaticIds * 4);
    ws.spk_cap = kStaticSpk;
    rc = t3mi::prompt_build(m->prompt, env, pc, text_tokens, n_text, cfg_weight, out_device, err, &ws);
    // the copies above read the caller's host arrays and the next call reuses the same working memory: drain the stream
    hipError_t e = hipStreamSynchronize(m->stream);
    if (rc) return fail(rc, err);
    if (e != hipSuccess) return fail(T3MI_ERR_HIP, std::string("hipStreamSynchronize: ") + hipGetErrorString(e));
    return T3MI_OK;
}

int t3mi_session_begin_tokens(t3mi_model* m, int n_utt, const t3mi_cond* conds, const int32_t* const* text_tokens,
                              const int32_t* n_text, const t3mi_params* p) {
    if (!p) return fail(T3MI_ERR_ARG, "t3mi_session_begin_tokens: null params");
    std::vector<const float*> ptrs;
    std::vector<int32_t> s0;
    int rpu = 1;
    int rc = build_prompts(m, n_utt, conds, text_tokens, n_text, p->cfg_weight, p->start_token, ptrs, s0, &rpu);
    if (rc) return rc;
    return t3mi_session_begin(m, n_utt, rpu, ptrs.data(), s0.data(), p, 1);
}

// Continuous batching: slots of the running session whose utterance has ended (or any slot) take new utterances while the others keep
// their state.  Their prompts are prefilled together (the same kernels as t3mi_session_begin), their cache rows are simply
// overwritten from position 0 (attention never reads a slot beyond the row's position), the per-slot state restarts, and the first
// logits of ALL rows are re-evaluated from the resident hidden states (bit-identical for the untouched rows).
int t3mi_session_refill(t3mi_model* m, int n, const int32_t* utts, const float* const* embeds, const int32_t* s0,
                        const int32_t* max_new_tokens, const int64_t* utterance_ids, int embeds_on_device) {
    int rc = need_session(m);
    if (rc) return rc;
    Session* s = m->sess.get();
    if (n <= 0 || !utts || !embeds || !s0) return fail(T3MI_ERR_ARG, "t3mi_session_refill: empty or null argument");
    if (!s->gemm_prefill) return fail(T3MI_ERR_STATE, "t3mi_session_refill needs the GEMM prefill (T3MI_PREFILL=token is set)");
    if (s->n_forced > 0) return fail(T3MI_ERR_STATE, "t3mi_session_refill: the session runs teacher-forced (t3mi_session_force)");
    const bool turbo = m->variant == T3MI_VARIANT_TURBO;
    const int rpu = s->rpu;
    const size_t d = m->d, V = m->V;
    std::vector<char> taken(s->n_utt, 0);
    std::vector<int32_t> mn(n);
    for (int i = 0; i < n; ++i) {
        const int u = utts[i];
        if (u < 0 || u >= s->n_utt) return fail(T3MI_ERR_ARG, "t3mi_session_refill: slot " + std::to_string(u) + " outside the session");
        if (taken[u]) return fail(T3MI_ERR_ARG, "t3mi_session_refill: slot " + std::to_string(u) + " listed twice");
        taken[u] = 1;
        if (!embeds[i] || s0[i] <= 0) return fail(T3MI_ERR_ARG, "t3mi_session_refill: null or empty prompt");
        // the cache rows were sized by t3mi_session_begin: prompt + one token at least must fit (the sampler stops a row at the end
        // of its slab, status 3), and under max_cache_len the rule of R/llama_traced.R:343-345 applies
        const int limit = s->prm.max_cache_len > 0 ? s->prm.max_cache_len : s->max_len;
        if (s0[i] + 1 >= limit)
            return fail(T3MI_ERR_ARG, "t3mi_session_refill: prompt of " + std::to_string(s0[i]) + " positions does not fit the session's cache rows (" +
                                          std::to_string(limit) + " slots)");
        int want = max_new_tokens ? max_new_tokens[i] : s->prm.max_new_tokens - (turbo ? 1 : 0);
        if (want <= 0) return fail(T3MI_ERR_ARG, "t3mi_session_refill: max_new_tokens must be positive");
        if (turbo) want += 1;   // the first token comes from the prefill logits (R/t3.R:1614-1658)
        mn[i] = std::min(want, s->cap);
    }
    HIPCHK(hipSetDevice(m->device));
    hipStream_t st = m->stream;
    HIPCHK(hipStreamSynchronize(st));   // the steps enqueued so far are over: the slots' tokens were the caller's to read before this call
    // prompts to the device
    DevBuf stage;
    std::vector<int> order(n);
    for (int i = 0; i < n; ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](int a, int b) { return utts[a] < utts[b]; });
    std::vector<const float*> base(n);
    if (embeds_on_device) {
        for (int i = 0; i < n; ++i) base[i] = embeds[i];
    } else {
        size_t tot = 0;
        for (int i = 0; i < n; ++i) tot += (size_t)rpu * s0[i] * d;
        HIPCHK(stage.alloc(tot * 4));
        size_t off = 0;
        for (int i = 0; i < n; ++i) {
            const size_t cnt = (size_t)rpu * s0[i] * d;
            HIPCHK(hipMemcpyAsync(stage.as<float>() + off, embeds[i], cnt * 4, hipMemcpyHostToDevice, st));
            base[i] = stage.as<float>() + off;
            off += cnt;
        }
    }
    std::vector<int> rows_list;
    std::vector<const float*> row_ptr;
    for (int oi = 0; oi < n; ++oi) {
        const int i = order[oi], u = utts[i];
        s->s0[u] = s0[i];
        for (int r = 0; r < rpu; ++r) {
            rows_list.push_back(u * rpu + r);
            row_ptr.push_back(base[i] + (size_t)r * s0[i] * d);
        }
    }
    s->max_s0 = *std::max_element(s->s0.begin(), s->s0.end());
    if ((rc = prefill_rows(m, s, rows_list, row_ptr, s->pf_act_f32))) return rc;
    // per-slot state: token count, end flag, seen set (Q3: column BOS - 1 from the first step on), prompt length, limit, RNG stream
    std::vector<unsigned long long> idv(n);
    const int32_t zero = 0;
    for (int i = 0; i < n; ++i) {
        const int u = utts[i];
        idv[i] = utterance_ids ? (unsigned long long)utterance_ids[i] : (unsigned long long)s->prm.first_utterance_id + (unsigned long long)u;
        HIPCHK(hipMemsetAsync(s->seen.as<uint8_t>() + (size_t)u * V, 0, V, st));
        if (s->prm.sampler == T3MI_SAMPLER_R && s->prm.repetition_penalty != 1.0f) {
            const int col = s->prm.start_token - 1;
            if (col >= 0 && col < m->V) HIPCHK(hipMemsetAsync(s->seen.as<uint8_t>() + (size_t)u * V + col, 1, 1, st));
        }
        HIPCHK(hipMemsetAsync(s->tokens.as<int>() + (size_t)u * s->cap, 0, (size_t)s->cap * 4, st));
        HIPCHK(hipMemcpyAsync(s->n_tokens.as<int>() + u, &zero, 4, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(s->done.as<int>() + u, &zero, 4, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(s->dist_n.as<int>() + u, &zero, 4, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(s->s0_dev.as<int>() + u, &s0[i], 4, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(s->max_new_u.as<int>() + u, &mn[i], 4, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(s->utt_ids.as<unsigned long long>() + u, &idv[i], 8, hipMemcpyHostToDevice, st));
    }
    // (the sources -- s0, mn, idv, zero -- live until the synchronisation below)
    if ((rc = run_head(m, s, nullptr))) return rc;
    HIPCHK(hipStreamSynchronize(st));
    return T3MI_OK;
}

int t3mi_session_refill_tokens(t3mi_model* m, int n, const int32_t* utts, const t3mi_cond* conds, const int32_t* const* text_tokens,
                               const int32_t* n_text, const int32_t* max_new_tokens, const int64_t* utterance_ids) {
    int rc = need_session(m);
    if (rc) return rc;
    Session* s = m->sess.get();
    std::vector<const float*> ptrs;
    std::vector<int32_t> s0;
    int rpu = 1;
    rc = build_prompts(m, n, conds, text_tokens, n_text, s->prm.cfg_weight, s->prm.start_token, ptrs, s0, &rpu);
    if (rc) return rc;
    if (rpu != s->rpu) return fail(T3MI_ERR_ARG, "t3mi_session_refill_tokens: the prompt has another number of rows than the session's utterances");
    return t3mi_session_refill(m, n, utts, ptrs.data(), s0.data(), max_new_tokens, utterance_ids, 1);
}

// ---- continuous batching driver: n_utt utterances through `slots` session slots (t3mi_session_refill between polls) ---------------
namespace {
struct ContinuousJob {
    int n_utt = 0, slots = 0, poll = 16;
    std::vector<int32_t> mx;                 // token limit of each utterance (without Turbo's extra first token)
    std::vector<int32_t> order;              // the queue: utterance indices in the order they are started
    std::function<int(const std::vector<int32_t>& items)> begin;                                          // session whose slot u holds utterance items[u]
    std::function<int(const std::vector<int32_t>& slots, const std::vector<int32_t>& items)> refill;      // slots[i] takes utterance items[i]
};

int run_continuous(t3mi_model* m, const ContinuousJob& job, int32_t* out, int32_t cap, int32_t* n_out, int32_t* stats) {
    const int extra = m->variant == T3MI_VARIANT_TURBO ? 1 : 0;
    std::vector<int32_t> owner(job.order.begin(), job.order.begin() + job.slots);
    int rc = job.begin(owner);
    if (rc) return rc;
    Session* s = m->sess.get();
    {   // t3mi_session_begin numbered the slots first_utterance_id + slot and gave all of them the session's limit: each slot gets
        // its utterance's global index (random stream) and own limit, as t3mi_session_refill sets them
        std::vector<unsigned long long> ids(job.slots);
        std::vector<int32_t> lim(job.slots);
        for (int u = 0; u < job.slots; ++u) {
            ids[u] = (unsigned long long)s->prm.first_utterance_id + (unsigned long long)owner[u];
            lim[u] = std::min(job.mx[owner[u]] + extra, s->cap);
        }
        hipError_t e = hipMemcpy(s->utt_ids.p, ids.data(), (size_t)job.slots * 8, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(s->max_new_u.p, lim.data(), (size_t)job.slots * 4, hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            t3mi_session_end(m);
            return fail(T3MI_ERR_HIP, std::string("t3mi_generate_continuous: ") + hipGetErrorString(e));
        }
    }
    const int drop_from = s->prm.drop_invalid_from;
    std::vector<int32_t> ntok(job.slots), done(job.slots), buf((size_t)s->cap);
    int next = job.slots, live = job.slots, warn = T3MI_OK;
    long long steps = 0, refills = 0, busy = 0;
    while (live > 0 && !rc) {
        if ((rc = t3mi_session_decode(m, job.poll))) break;
        steps += job.poll;
        busy += (long long)job.poll * live;
        if ((rc = t3mi_session_status(m, ntok.data(), done.data()))) break;
        std::vector<int32_t> freed;
        for (int u = 0; u < job.slots && !rc; ++u) {
            const int i = owner[u];
            if (i < 0 || done[u] == 0) continue;
            int32_t n = 0;
            if ((rc = t3mi_session_tokens(m, u, buf.data(), (int32_t)buf.size(), &n))) break;
            n = std::min(n, job.mx[i] + extra);
            int32_t k = 0;
            int32_t* dst = out + (size_t)i * cap;
            for (int32_t j = 0; j < n && k < cap; ++j)
                if (drop_from <= 0 || buf[j] < drop_from) dst[k++] = buf[j];   // drop_invalid_tokens (R/s3tokenizer.R:678-687)
            n_out[i] = k;
            if (done[u] == 3) warn = T3MI_WARN_CACHE_FULL;
            owner[u] = -1;
            --live;
            freed.push_back(u);
        }
        if (rc) break;
        const int take = std::min<int>((int)freed.size(), job.n_utt - next);
        if (take > 0) {
            std::vector<int32_t> sl(freed.begin(), freed.begin() + take), it(take);
            for (int k = 0; k < take; ++k) { it[k] = job.order[next + k]; owner[sl[k]] = it[k]; }
            if ((rc = job.refill(sl, it))) break;
            next += take;
            live += take;
            ++refills;
        }
    }
    if (stats) {
        stats[0] = (int32_t)std::min<long long>(steps, 0x7FFFFFFF);
        stats[1] = (int32_t)refills;
        stats[2] = (int32_t)std::min<long long>(busy, 0x7FFFFFFF);
        stats[3] = job.slots;
    }
    std::string keep = g_err;
    t3mi_session_end(m);
    if (rc) {
        g_err = keep;
        return rc;
    }
    if (warn) g_err = "KV cache full, generation stopped (src/t3_decode.cpp:394-397)";
    return warn;
}

// common argument checks; fills job.mx, the queue order and the session parameters (cache rows long enough for every utterance that
// may come).  length_hint[i]: what T3MI_CONTINUOUS_LONGEST_FIRST sorts by when no per-utterance limits are given.
int continuous_setup(t3mi_model* m, int n_utt, const int32_t* s0, const int32_t* max_new_each, const int32_t* length_hint, int slots,
                     int poll_steps, int flags, const t3mi_params* p, int32_t cap, ContinuousJob& job, t3mi_params& sp) {
    if (!m || !p || n_utt <= 0) return fail(T3MI_ERR_ARG, "t3mi_generate_continuous: null or empty argument");
    if (slots <= 0 || poll_steps <= 0) return fail(T3MI_ERR_ARG, "t3mi_generate_continuous: slots and poll_steps must be positive");
    if (flags & ~T3MI_CONTINUOUS_LONGEST_FIRST) return fail(T3MI_ERR_ARG, "t3mi_generate_continuous: unknown flag");
    const int extra = m->variant == T3MI_VARIANT_TURBO ? 1 : 0;
    job.n_utt = n_utt;
    job.slots = std::min(slots, n_utt);
    job.poll = poll_steps;
    job.mx.resize(n_utt);
    int top = 0, need = 0;
    for (int i = 0; i < n_utt; ++i) {
        const int v = max_new_each ? max_new_each[i] : p->max_new_tokens;
        if (v <= 0 || v > p->max_new_tokens) return fail(T3MI_ERR_ARG, "t3mi_generate_continuous: every token limit must be in 1..max_new_tokens");
        job.mx[i] = v;
        top = std::max(top, v);
        need = std::max(need, s0[i] + v);
    }
    job.order.resize(n_utt);
    for (int i = 0; i < n_utt; ++i) job.order[i] = i;
    if (flags & T3MI_CONTINUOUS_LONGEST_FIRST) {
        // the utterances expected to run longest start first, so that the job does not end on one long straggler with every other slot
        // idle: by their own limits when given, else by the hint (text length / prompt length: longer text, longer speech)
        const int32_t* key = max_new_each ? max_new_each : length_hint;
        if (key) std::stable_sort(job.order.begin(), job.order.end(), [&](int32_t a, int32_t b) { return key[a] > key[b]; });
    }
    if (cap < top + extra) return fail(T3MI_ERR_ARG, "output capacity is smaller than the largest token limit");
    sp = *p;
    sp.max_new_tokens = top;
    if (sp.max_cache_len <= 0) sp.max_cache_len = need + 2 + extra;   // long enough that the per-row clamp of R/t3.R:996 never binds
    return T3MI_OK;
}
}  // namespace

int t3mi_generate_continuous(t3mi_model* m, int n_utt, int rows_per_utt, const float* const* embeds, const int32_t* s0,
                             const int32_t* max_new_each, int slots, int poll_steps, int flags, const t3mi_params* p, int32_t* out,
                             int32_t cap, int32_t* n_out, int32_t* stats) {
    if (!embeds || !s0 || !out || !n_out) return fail(T3MI_ERR_ARG, "t3mi_generate_continuous: null argument");
    ContinuousJob job;
    t3mi_params sp;
    int rc = continuous_setup(m, n_utt, s0, max_new_each, s0, slots, poll_steps, flags, p, cap, job, sp);
    if (rc) return rc;
    job.begin = [&](const std::vector<int32_t>& it) {
        std::vector<const float*> e(it.size());
        std::vector<int32_t> l(it.size());
        for (size_t k = 0; k < it.size(); ++k) { e[k] = embeds[it[k]]; l[k] = s0[it[k]]; }
        return t3mi_session_begin(m, (int)it.size(), rows_per_utt, e.data(), l.data(), &sp, 0);
    };
    job.refill = [&](const std::vector<int32_t>& sl, const std::vector<int32_t>& it) {
        std::vector<const float*> e(it.size());
        std::vector<int32_t> l(it.size()), mn(it.size());
        std::vector<int64_t> id(it.size());
        for (size_t k = 0; k < it.size(); ++k) {
            e[k] = embeds[it[k]]; l[k] = s0[it[k]]; mn[k] = job.mx[it[k]];
            id[k] = (int64_t)sp.first_utterance_id + it[k];
        }
        return t3mi_session_refill(m, (int)it.size(), sl.data(), e.data(), l.data(), mn.data(), id.data(), 0);
    };
    return run_continuous(m, job, out, cap, n_out, stats);
}

int t3mi_generate_continuous_tokens(t3mi_model* m, int n_utt, const t3mi_cond* conds, const int32_t* const* text_tokens,
                                    const int32_t* n_text, const int32_t* max_new_each, int slots, int poll_steps, int flags,
                                    const t3mi_params* p, int32_t* out, int32_t cap, int32_t* n_out, int32_t* stats) {
    if (!m || !p || !conds || !text_tokens || !n_text || !out || !n_out || n_utt <= 0)
        return fail(T3MI_ERR_ARG, "t3mi_generate_continuous_tokens: null or empty argument");
    std::vector<int32_t> s0(n_utt);
    {
        const t3mi::PromptEnv env = prompt_env(m, p->start_token);
        std::string err;
        for (int i = 0; i < n_utt; ++i) {
            int r = 0, lc = 0, l = 0;
            const int rc = t3mi::prompt_shape(m->prompt, env, prompt_cond(&conds[i]), n_text[i], p->cfg_weight, &r, &l, &lc, err);
            if (rc) return fail(rc, "utterance " + std::to_string(i) + ": " + err);
            s0[i] = l;
        }
    }
    ContinuousJob job;
    t3mi_params sp;
    int rc = continuous_setup(m, n_utt, s0.data(), max_new_each, n_text, slots, poll_steps, flags, p, cap, job, sp);
    if (rc) return rc;
    auto pick = [&](const std::vector<int32_t>& it, std::vector<t3mi_cond>& c, std::vector<const int32_t*>& t, std::vector<int32_t>& nt) {
        c.resize(it.size()); t.resize(it.size()); nt.resize(it.size());
        for (size_t k = 0; k < it.size(); ++k) { c[k] = conds[it[k]]; t[k] = text_tokens[it[k]]; nt[k] = n_text[it[k]]; }
    };
    job.begin = [&](const std::vector<int32_t>& it) {
        std::vector<t3mi_cond> c; std::vector<const int32_t*> t; std::vector<int32_t> nt;
        pick(it, c, t, nt);
        return t3mi_session_begin_tokens(m, (int)it.size(), c.data(), t.data(), nt.data(), &sp);
    };
    job.refill = [&](const std::vector<int32_t>& sl, const std::vector<int32_t>& it) {
        std::vector<t3mi_cond> c; std::vector<const int32_t*> t; std::vector<int32_t> nt, mn(it.size());
        std::vector<int64_t> id(it.size());
        pick(it, c, t, nt);
        for (size_t k = 0; k < it.size(); ++k) { mn[k] = job.mx[it[k]]; id[k] = (int64_t)sp.first_utterance_id + it[k]; }
        return t3mi_session_refill_tokens(m, (int)it.size(), sl.data(), c.data(), t.data(), nt.data(), mn.data(), id.data());
    };
    return run_continuous(m, job, out, cap, n_out, stats);
}

int t3mi_generate_batch_tokens(t3mi_model* m, int n_utt, const t3mi_cond* conds, const int32_t* const* text_tokens,
                               const int32_t* n_text, const t3mi_params* p, int32_t* out, int32_t cap, int32_t* n_out) {
    if (!out || !n_out) return fail(T3MI_ERR_ARG, "t3mi_generate_batch_tokens: null output");
    int rc = t3mi_session_begin_tokens(m, n_utt, conds, text_tokens, n_text, p);
    if (rc) return rc;
    return run_generate(m, n_utt, out, cap, n_out);
}

int t3mi_generate_tokens(t3mi_model* m, const t3mi_cond* cond, const int32_t* text_tokens, int n_text, const t3mi_params* p,
                         int32_t* out, int32_t cap, int32_t* n_out) {
    const int32_t* tt[1] = {text_tokens};
    const int32_t nt[1] = {n_text};
    return t3mi_generate_batch_tokens(m, 1, cond, tt, nt, p, out, cap, n_out);
}

int t3mi_generate(t3mi_model* m, const float* embeds, int rows, int s0, const t3mi_params* p, int32_t* out,
                  int32_t cap, int32_t* n_out) {
    const float* ptrs[1] = {embeds};
    const int32_t s0v[1] = {s0};
    return t3mi_generate_batch(m, 1, rows, ptrs, s0v, p, out, cap, n_out);
}

}  // extern "C"
