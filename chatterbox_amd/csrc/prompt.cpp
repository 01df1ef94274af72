// Prompt builder: conditioning encoder + perceiver resampler + text / speech embeddings on device
// (see prompt.hpp; reference R/t3.R:48-90, 140-316, 361-403, Turbo 1447-1477).
#include "prompt.hpp"

#include <algorithm>
#include <cstring>
#include <memory>
#include <vector>

#include "t3mi.h"
#include "kernels.hpp"

namespace t3mi {

namespace {

struct Buf {
    void* p = nullptr;
    size_t bytes = 0;
    ~Buf() { if (p) (void)hipFree(p); }
    hipError_t reserve(size_t n) {
        if (n <= bytes) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
        hipError_t e = hipMalloc(&p, n);
        if (e == hipSuccess) bytes = n;
        else p = nullptr;
        return e;
    }
    template <typename T> T* as() const { return static_cast<T*>(p); }
};

constexpr int kStartText = 255, kStopText = 0;   // R/t3.R:14-15
constexpr int kPerceiverHeads = 4;               // R/t3.R:274

}  // namespace

struct PromptModel {
    int variant = 0, d = 0, text_vocab = 0, max_text_pos = 0, spk_dim = 0, n_query = 0;
    bool perceiver = false, emotion = false;
    Buf w;
    const float *text_emb = nullptr, *text_pos = nullptr, *spk_w = nullptr, *spk_b = nullptr, *emo_w = nullptr,
                *query = nullptr, *ln_w = nullptr, *ln_b = nullptr, *wq = nullptr, *bq = nullptr, *wk = nullptr,
                *bk = nullptr, *wv = nullptr, *bv = nullptr, *wo = nullptr, *bo = nullptr;
    Buf scratch;
    // per-batch staging of the host inputs: every utterance gets its own slice of the device arenas and its own
    // host copy, so an asynchronous upload never races the next utterance's (the copies may still be in flight when
    // prompt_build returns); both are recycled by prompt_batch_begin, which the caller runs on a drained stream
    Buf ids_arena, spk_arena;
    size_t ids_off = 0, spk_off = 0;
    std::vector<std::vector<int32_t>> hold_ids;
    std::vector<std::vector<float>> hold_spk;
};

namespace {

struct Item {
    const char* name;
    std::vector<int64_t> shape;
    const float** slot;
};

#define PHIP(expr)                                                                  \
    do {                                                                            \
        hipError_t e_ = (expr);                                                     \
        if (e_ != hipSuccess) {                                                     \
            err = std::string(#expr) + ": " + hipGetErrorString(e_);                \
            return T3MI_ERR_HIP;                                                    \
        }                                                                           \
    } while (0)

}  // namespace

void prompt_free(PromptModel* pm) { delete pm; }

int prompt_load(const SafeTensorsFile& st, int variant, int d, PromptModel** out, std::string& err, bool validate_only) {
    *out = nullptr;
    const StTensor* te = st.find("text_emb.weight");
    const StTensor* sw = st.find("cond_enc.spkr_enc.weight");
    if (!te && !sw) return 1;   // decode-only checkpoint: the prompt entry points report it when called
    if (!te || !sw || te->shape.size() != 2 || sw->shape.size() != 2 || te->shape[1] != d || sw->shape[0] != d) {
        err = "prompt tensors: text_emb.weight must be [text_vocab, d] and cond_enc.spkr_enc.weight [d, speaker_dim]";
        return T3MI_ERR_IO;
    }
    std::unique_ptr<PromptModel> pm(new PromptModel());
    pm->variant = variant;
    pm->d = d;
    pm->text_vocab = (int)te->shape[0];
    pm->spk_dim = (int)sw->shape[1];
    if (pm->spk_dim % 64 || d % 64) {
        err = "prompt tensors: speaker_dim and d must be multiples of 64";
        return T3MI_ERR_IO;
    }
    std::vector<Item> items;
    items.push_back({"text_emb.weight", {pm->text_vocab, d}, &pm->text_emb});
    items.push_back({"cond_enc.spkr_enc.weight", {d, pm->spk_dim}, &pm->spk_w});
    items.push_back({"cond_enc.spkr_enc.bias", {d}, &pm->spk_b});
    if (variant == T3MI_VARIANT_LLAMA) {
        const StTensor* tp = st.find("text_pos_emb.emb.weight");
        if (!tp || tp->shape.size() != 2 || tp->shape[1] != d) {
            err = "prompt tensors: text_pos_emb.emb.weight [max_text_pos, d] is missing (R/t3.R:1084)";
            return T3MI_ERR_IO;
        }
        pm->max_text_pos = (int)tp->shape[0];
        items.push_back({"text_pos_emb.emb.weight", {pm->max_text_pos, d}, &pm->text_pos});
        if (st.find("cond_enc.emotion_adv_fc.weight")) {
            pm->emotion = true;
            items.push_back({"cond_enc.emotion_adv_fc.weight", {d, 1}, &pm->emo_w});
        }
        if (const StTensor* pq = st.find("cond_enc.perceiver.pre_attention_query")) {
            if (pq->shape.size() != 3 || pq->shape[0] != 1 || pq->shape[2] != d || d % kPerceiverHeads) {
                err = "prompt tensors: cond_enc.perceiver.pre_attention_query must be [1, n_query, d]";
                return T3MI_ERR_IO;
            }
            pm->perceiver = true;
            pm->n_query = (int)pq->shape[1];
            items.push_back({"cond_enc.perceiver.pre_attention_query", {1, pm->n_query, d}, &pm->query});
            items.push_back({"cond_enc.perceiver.attn.norm.weight", {d}, &pm->ln_w});
            items.push_back({"cond_enc.perceiver.attn.norm.bias", {d}, &pm->ln_b});
            items.push_back({"cond_enc.perceiver.attn.to_q.weight", {d, d}, &pm->wq});
            items.push_back({"cond_enc.perceiver.attn.to_q.bias", {d}, &pm->bq});
            items.push_back({"cond_enc.perceiver.attn.to_k.weight", {d, d}, &pm->wk});
            items.push_back({"cond_enc.perceiver.attn.to_k.bias", {d}, &pm->bk});
            items.push_back({"cond_enc.perceiver.attn.to_v.weight", {d, d}, &pm->wv});
            items.push_back({"cond_enc.perceiver.attn.to_v.bias", {d}, &pm->bv});
            items.push_back({"cond_enc.perceiver.attn.proj_out.weight", {d, d}, &pm->wo});
            items.push_back({"cond_enc.perceiver.attn.proj_out.bias", {d}, &pm->bo});
        }
    }
    size_t total = 0;
    for (const Item& it : items) {
        const StTensor* t = st.find(it.name);
        if (!t || t->shape != it.shape) {
            err = std::string("prompt tensors: '") + it.name + "' is missing or has an unexpected shape";
            return T3MI_ERR_IO;
        }
        if (t->dtype != StDtype::F32 && t->dtype != StDtype::F16 && t->dtype != StDtype::BF16) {
            err = std::string("prompt tensors: '") + it.name + "' is not a floating-point tensor (F32 / F16 / BF16)";
            return T3MI_ERR_IO;
        }
        total += ((size_t)t->numel() + 63) / 64 * 64;   // 256-byte aligned slots
    }
    if (validate_only) return 0;
    PHIP(pm->w.reserve(total * 4));
    float* dst = pm->w.as<float>();
    std::vector<float> host;
    for (const Item& it : items) {
        const StTensor* t = st.find(it.name);
        const size_t n = (size_t)t->numel();
        const void* src = t->data;
        if (t->dtype != StDtype::F32) {
            host.resize(n);
            if (!SafeTensorsFile::to_f32(*t, host.data(), err)) return T3MI_ERR_IO;
            src = host.data();
        }
        PHIP(hipMemcpy(dst, src, n * 4, hipMemcpyHostToDevice));
        *it.slot = dst;
        dst += (n + 63) / 64 * 64;
    }
    *out = pm.release();
    return 0;
}

int prompt_batch_begin(PromptModel* pm, size_t total_ids, int n_utt, std::string& err) {
    if (!pm) return T3MI_OK;   // prompt_shape reports the missing tensors
    PHIP(pm->ids_arena.reserve((total_ids + 16) * 4));
    PHIP(pm->spk_arena.reserve(((size_t)n_utt * pm->spk_dim + 16) * 4));
    pm->ids_off = pm->spk_off = 0;
    pm->hold_ids.clear();
    pm->hold_spk.clear();
    return T3MI_OK;
}

int prompt_shape(const PromptModel* pm, const PromptEnv& env, const PromptCond& c, int n_text, float cfg_weight, int* rows,
                 int* s0, int* len_cond, std::string& err) {
    if (!pm) {
        err = "this checkpoint has no prompt tensors (text_emb.weight, cond_enc.*): pass embeddings instead";
        return T3MI_ERR_STATE;
    }
    if (n_text < 0 || c.n_cond < 0 || (c.n_cond > 0 && !c.cond_tokens) || !c.speaker_emb) {
        err = "prompt: null or negative-length input";
        return T3MI_ERR_ARG;
    }
    if (c.n_speaker != pm->spk_dim) {
        err = "prompt: speaker_emb has " + std::to_string(c.n_speaker) + " values, cond_enc.spkr_enc expects " +
              std::to_string(pm->spk_dim);
        return T3MI_ERR_ARG;
    }
    const bool llama = env.variant == T3MI_VARIANT_LLAMA;
    int lc = 1;
    if (c.n_cond > 0) lc += (llama && pm->perceiver) ? pm->n_query : c.n_cond;
    if (llama && pm->emotion && c.has_emotion) lc += 1;
    const int n_tt = llama ? n_text + 2 : n_text;   // start / stop text tokens (R/t3.R:473-477); none on Turbo (:1590)
    if (llama && n_tt > pm->max_text_pos) {
        err = "prompt: text is longer than the learned text position table";
        return T3MI_ERR_ARG;
    }
    if (llama && c.n_cond > env.max_pos) {
        err = "prompt: conditioning prompt is longer than the learned speech position table";
        return T3MI_ERR_ARG;
    }
    *rows = (llama && cfg_weight > 0.0f) ? 2 : 1;
    *len_cond = lc;
    *s0 = lc + n_tt + 1;
    return T3MI_OK;
}

void prompt_scratch_need(const PromptModel* pm, const PromptEnv& env, const PromptCond& c, int n_text, size_t* rows_floats,
                         size_t* n_ids) {
    const bool llama = env.variant == T3MI_VARIANT_LLAMA;
    const int nq = (pm && pm->perceiver) ? pm->n_query : 0;
    const int mk = std::max(std::max(c.n_cond, 0), nq);
    // scratch rows: E, N2, K, V [mk] | N1, Q, A, P [nq]
    *rows_floats = (size_t)(4 * mk + 4 * nq) * (size_t)(pm ? pm->d : 0);
    *n_ids = (size_t)std::max(c.n_cond, 0) + (size_t)std::max(n_text, 0) + (llama ? 2 : 0) + 1;
}

namespace {

// y[M, d] = A[M, K] W[d, K]^T + b (+ resid)
hipError_t linear(const float* A, const float* W, const float* b, const float* resid, float* y, int M, int N, int K,
                  hipStream_t s) {
    t3k::GemmArgs g;
    g.W = W;
    g.A = A;
    g.M = M;
    g.N = N;
    g.K = K;
    g.y = y;
    g.ldy = N;
    g.bias = b;
    g.resid = resid;
    return t3k::launch_gemm(g, resid ? t3k::EPI_RESID : t3k::EPI_STORE, 0, s);
}

}  // namespace

int prompt_build(PromptModel* pm, const PromptEnv& env, const PromptCond& c, const int32_t* text_tokens, int n_text,
                 float cfg_weight, float* out_dev, std::string& err, const PromptScratch* ext) {
    int rows = 0, s0 = 0, lc = 0;
    int rc = prompt_shape(pm, env, c, n_text, cfg_weight, &rows, &s0, &lc, err);
    if (rc) return rc;
    if (!out_dev || (n_text > 0 && !text_tokens)) {
        err = "prompt: null buffer";
        return T3MI_ERR_ARG;
    }
    const bool llama = env.variant == T3MI_VARIANT_LLAMA;
    const int d = pm->d, n_cond = c.n_cond;
    const int n_tt = llama ? n_text + 2 : n_text;
    hipStream_t st = env.stream;
    // every id is checked against its table on the host: the gather kernels index with it unchecked
    if (env.start_token < 0 || env.start_token >= env.vocab) {
        err = "prompt: start_speech_token outside the speech table";
        return T3MI_ERR_ARG;
    }
    for (int i = 0; i < n_cond; ++i)
        if (c.cond_tokens[i] < 0 || c.cond_tokens[i] >= env.vocab) {
            err = "prompt: cond_prompt_speech_tokens[" + std::to_string(i) + "] outside the speech table";
            return T3MI_ERR_ARG;
        }
    for (int i = 0; i < n_text; ++i)
        if (text_tokens[i] < 0 || text_tokens[i] >= pm->text_vocab) {
            err = "prompt: text_tokens[" + std::to_string(i) + "] outside the text table";
            return T3MI_ERR_ARG;
        }
    if (llama && (kStartText >= pm->text_vocab)) {
        err = "prompt: text table smaller than start_text_token";
        return T3MI_ERR_ARG;
    }
    // ids on device: [cond tokens | text tokens (with start / stop) | start speech token]
    const int nq = pm->perceiver ? pm->n_query : 0;
    const int mk = std::max(n_cond, nq);
    const size_t n_ids = (size_t)n_cond + (size_t)n_tt + 1;
    const size_t rows_floats = (size_t)(4 * mk + 4 * nq) * d;
    int* ids_dev = nullptr;
    float* spk = nullptr;
    float* E = nullptr;
    if (ext != nullptr) {
        // caller-owned working memory: nothing is allocated; the three small host -> device copies below read the
        // caller's arrays directly, which stay valid until the caller has synchronised the stream
        if (!ext->rows || !ext->ids || !ext->spk || n_ids > ext->ids_cap || (size_t)pm->spk_dim > ext->spk_cap ||
            rows_floats > ext->rows_floats) {
            err = "prompt: inputs exceed the fixed working memory (" + std::to_string(n_ids) + " ids of " +
                  std::to_string(ext->ids_cap) + ", " + std::to_string(rows_floats) + " scratch floats of " +
                  std::to_string(ext->rows_floats) + ")";
            return T3MI_ERR_ARG;
        }
        ids_dev = ext->ids;
        spk = ext->spk;
        E = ext->rows;
        int* q = ids_dev;
        if (n_cond > 0) PHIP(hipMemcpyAsync(q, c.cond_tokens, (size_t)n_cond * 4, hipMemcpyHostToDevice, st));
        q += n_cond;
        // start / stop text tokens and the start speech token are constants: written by 4-byte fills, no host buffer
        if (llama) PHIP(hipMemsetD32Async((hipDeviceptr_t)(q++), kStartText, 1, st));
        if (n_text > 0) PHIP(hipMemcpyAsync(q, text_tokens, (size_t)n_text * 4, hipMemcpyHostToDevice, st));
        q += n_text;
        if (llama) PHIP(hipMemsetD32Async((hipDeviceptr_t)(q++), kStopText, 1, st));
        PHIP(hipMemsetD32Async((hipDeviceptr_t)q, env.start_token, 1, st));
        PHIP(hipMemcpyAsync(spk, c.speaker_emb, (size_t)pm->spk_dim * 4, hipMemcpyHostToDevice, st));
    } else {
        pm->hold_ids.emplace_back();
        std::vector<int32_t>& ids = pm->hold_ids.back();
        ids.insert(ids.end(), c.cond_tokens, c.cond_tokens + n_cond);
        if (llama) ids.push_back(kStartText);
        ids.insert(ids.end(), text_tokens, text_tokens + n_text);
        if (llama) ids.push_back(kStopText);
        ids.push_back(env.start_token);
        if ((pm->ids_off + ids.size()) * 4 > pm->ids_arena.bytes || (pm->spk_off + pm->spk_dim) * 4 > pm->spk_arena.bytes) {
            err = "prompt: staging arenas too small (prompt_batch_begin not called for this batch)";
            return T3MI_ERR_STATE;
        }
        ids_dev = pm->ids_arena.as<int>() + pm->ids_off;
        pm->ids_off += ids.size();
        PHIP(hipMemcpyAsync(ids_dev, ids.data(), ids.size() * 4, hipMemcpyHostToDevice, st));
        // scratch rows (reused by every utterance, in stream order): E, N2, K, V [mk] | N1, Q, A, P [nq]
        PHIP(pm->scratch.reserve((rows_floats + 16) * 4));
        spk = pm->spk_arena.as<float>() + pm->spk_off;
        pm->spk_off += pm->spk_dim;
        E = pm->scratch.as<float>();
        pm->hold_spk.emplace_back(c.speaker_emb, c.speaker_emb + pm->spk_dim);
        PHIP(hipMemcpyAsync(spk, pm->hold_spk.back().data(), (size_t)pm->spk_dim * 4, hipMemcpyHostToDevice, st));
    }
    const int* id_cond = ids_dev;
    const int* id_text = id_cond + n_cond;
    const int* id_bos = id_text + n_tt;

    float* N2 = E + (size_t)mk * d;
    float* K = N2 + (size_t)mk * d;
    float* V = K + (size_t)mk * d;
    float* N1 = V + (size_t)mk * d;
    float* Q = N1 + (size_t)nq * d;
    float* A = Q + (size_t)nq * d;
    float* P = A + (size_t)nq * d;

    float* row0 = out_dev;
    // speaker projection (R/t3.R:285-286)
    PHIP(linear(spk, pm->spk_w, pm->spk_b, nullptr, row0, 1, d, pm->spk_dim, st));
    int at = 1;
    if (n_cond > 0) {
        if (llama && pm->perceiver) {
            // prompt-speech embeddings with their position term (R/t3.R:366), then the perceiver (R/t3.R:227-241):
            // the same attention block twice -- queries attend to the prompt, the result attends to itself
            PHIP(t3k::launch_embed_rows(env.speech_emb, id_cond, env.speech_pos, E, n_cond, d, st));
            const float* x1 = pm->query;
            const float* x2 = E;
            int m2 = n_cond;
            for (int pass = 0; pass < 2; ++pass) {
                float* y = pass == 0 ? P : row0 + (size_t)at * d;
                PHIP(t3k::launch_layernorm_rows(const_cast<float*>(x1), pm->ln_w, pm->ln_b, 1e-5f, N1, nq, d, 0, st));
                PHIP(t3k::launch_layernorm_rows(const_cast<float*>(x2), pm->ln_w, pm->ln_b, 1e-5f, N2, m2, d, 0, st));
                PHIP(linear(N1, pm->wq, pm->bq, nullptr, Q, nq, d, d, st));
                PHIP(linear(N2, pm->wk, pm->bk, nullptr, K, m2, d, d, st));
                PHIP(linear(N2, pm->wv, pm->bv, nullptr, V, m2, d, d, st));
                PHIP(t3k::launch_mha_rows(Q, K, V, A, nq, m2, kPerceiverHeads, d / kPerceiverHeads, st));
                PHIP(linear(A, pm->wo, pm->bo, x1, y, nq, d, d, st));   // x1 + proj_out(attn)  (R/t3.R:193-194)
                x1 = P;
                x2 = P;
                m2 = nq;
            }
            at += nq;
        } else {
            // no perceiver: the embeddings themselves (Turbo: without a position term, R/t3.R:1452)
            PHIP(t3k::launch_embed_rows(env.speech_emb, id_cond, llama ? env.speech_pos : nullptr, row0 + (size_t)at * d,
                                        n_cond, d, st));
            at += n_cond;
        }
    }
    if (llama && pm->emotion && c.has_emotion) {
        PHIP(t3k::launch_scale_row(pm->emo_w, c.emotion, row0 + (size_t)at * d, d, st));   // R/t3.R:305-310
        at += 1;
    }
    if (at != lc) {
        err = "prompt: internal length mismatch";
        return T3MI_ERR_STATE;
    }
    if (n_tt > 0)   // text embeddings (+ learned position 0..T-1 on the Llama variant, R/t3.R:379-380)
        PHIP(t3k::launch_embed_rows(pm->text_emb, id_text, llama ? pm->text_pos : nullptr, row0 + (size_t)at * d, n_tt, d, st));
    at += n_tt;
    // start-of-speech embedding (+ position 0, R/t3.R:389-390)
    PHIP(t3k::launch_embed_rows(env.speech_emb, id_bos, llama ? env.speech_pos : nullptr, row0 + (size_t)at * d, 1, d, st));
    if (rows == 2) {
        // unconditional row of the CFG pair: same conditioning and BOS, text span zeroed (R/t3.R:383-386)
        float* row1 = out_dev + (size_t)s0 * d;
        PHIP(hipMemcpyAsync(row1, row0, (size_t)s0 * d * 4, hipMemcpyDeviceToDevice, st));
        if (n_tt > 0) PHIP(hipMemsetAsync(row1 + (size_t)lc * d, 0, (size_t)n_tt * d * 4, st));
    }
    return T3MI_OK;
}

}  // namespace t3mi
