/* R glue for libt3mi: the .Call entry points that replace
 *     SEXP cpp_t3_decode(SEXP x 11)            /root/reference src/t3_decode.cpp:207-219
 *     registered in R_init_chatterbox           /root/reference src/init.c:5-15
 * The reference passes R-torch external pointers (src/t3_decode.cpp:26-32); here the model is
 * an external pointer to a t3mi handle that owns its weights in HBM, the prompt is either the
 * conditioning + text token ids t3_inference() receives (built on the GPU, t3mi_generate_tokens) or
 * a plain numeric [rows, S0, d] array, and the decode parameters are the SAME named list the reference builds at
 * R/t3.R:1022-1034 (read by name, like src/t3_decode.cpp:43-64, 221-231).
 *
 * Compiled by the package's `configure` + R CMD INSTALL on a machine with R and ROCm; in the build container (no R) it is
 * compiled against a test double of the R C API and driven like .Call (tests/r_mock/, tests/test_gpu_r_glue.py).
 * Plain C, R API only.
 */
#include <R.h>
#include <Rinternals.h>
#include <R_ext/Rdynload.h>
#include <stdlib.h>
#include <string.h>

#ifndef HAVE_T3MI
/* configure found no hipcc: same behaviour as the reference's build without libtorch
 * (src/t3_decode.cpp:432-446) -- the package loads, the backend raises when called. */
static SEXP no_backend(void) {
    Rf_error("chatterbox was built without the MI355X T3 backend (hipcc not found at configure time)");
    return R_NilValue;
}
SEXP t3mi_r_open(SEXP a, SEXP b, SEXP c, SEXP d, SEXP e) { return no_backend(); }
SEXP t3mi_r_close(SEXP a) { return R_NilValue; }
SEXP t3mi_r_generate(SEXP a, SEXP b, SEXP c, SEXP d) { return no_backend(); }
SEXP t3mi_r_generate_batch(SEXP a, SEXP b, SEXP c, SEXP d) { return no_backend(); }
SEXP t3mi_r_generate_tokens(SEXP a, SEXP b, SEXP c, SEXP d) { return no_backend(); }
#else
#include "t3mi.h"

static double cfg_real(SEXP list, const char* name, double dflt, int required) {
    SEXP names = Rf_getAttrib(list, R_NamesSymbol);
    if (names != R_NilValue)
        for (R_xlen_t i = 0; i < Rf_xlength(list); i++)
            if (strcmp(CHAR(STRING_ELT(names, i)), name) == 0) return Rf_asReal(VECTOR_ELT(list, i));
    if (required) Rf_error("Missing config parameter: %s", name); /* same text as src/t3_decode.cpp:50 */
    return dflt;
}

/* optional integer-vector entry of the config list (max_tokens_each); NULL when absent.  *n receives its length. */
static const int* cfg_ints(SEXP list, const char* name, int* n) {
    SEXP names = Rf_getAttrib(list, R_NamesSymbol);
    *n = 0;
    if (TYPEOF(list) != VECSXP || names == R_NilValue) return NULL;
    for (R_xlen_t i = 0; i < Rf_xlength(list); i++)
        if (strcmp(CHAR(STRING_ELT(names, i)), name) == 0) {
            SEXP v = VECTOR_ELT(list, i);
            if (!Rf_isInteger(v)) return NULL;
            *n = (int)Rf_xlength(v);
            return INTEGER(v);
        }
    return NULL;
}

static void model_finalizer(SEXP ptr) {
    t3mi_model* m = (t3mi_model*)R_ExternalPtrAddr(ptr);
    if (m) {
        t3mi_close(m);
        R_ClearExternalPtr(ptr);
    }
}

static t3mi_model* get_model(SEXP ptr) {
    if (TYPEOF(ptr) != EXTPTRSXP) Rf_error("t3mi: not a model handle");
    t3mi_model* m = (t3mi_model*)R_ExternalPtrAddr(ptr);
    if (!m) Rf_error("t3mi: model handle was closed");
    return m;
}

/* .Call("t3mi_r_open", path, variant, device, dtype, cache_path or NULL) -> external pointer */
SEXP t3mi_r_open(SEXP path, SEXP variant, SEXP device, SEXP dtype, SEXP cache) {
    t3mi_model* m = NULL;
    if (TYPEOF(path) != STRSXP || Rf_length(path) != 1) Rf_error("t3mi_open: path must be one string");
    const char* cp = (TYPEOF(cache) == STRSXP && Rf_length(cache) == 1) ? CHAR(STRING_ELT(cache, 0)) : NULL;
    int rc = t3mi_open_cached(CHAR(STRING_ELT(path, 0)), Rf_asInteger(variant), Rf_asInteger(device),
                              Rf_asInteger(dtype), cp, &m);
    if (rc != T3MI_OK) Rf_error("t3mi_open: %s", t3mi_last_error());
    SEXP ptr = PROTECT(R_MakeExternalPtr(m, R_NilValue, R_NilValue));
    R_RegisterCFinalizerEx(ptr, model_finalizer, TRUE);
    UNPROTECT(1);
    return ptr;
}

SEXP t3mi_r_close(SEXP ptr) {
    model_finalizer(ptr);
    return R_NilValue;
}

static void fill_params(SEXP cfg, t3mi_params* p) {
    t3mi_default_params(p);
    /* names of the reference's list (R/t3.R:1022-1034); cond_len / n_heads / head_dim / rms_eps are
     * properties of the prompt and the checkpoint here and are accepted but not needed */
    p->max_new_tokens = (int32_t)cfg_real(cfg, "max_tokens", 1000, 1);
    p->temperature = (float)cfg_real(cfg, "temperature", 0.8, 1);
    p->cfg_weight = (float)cfg_real(cfg, "cfg_weight", 0.5, 1);
    p->top_p = (float)cfg_real(cfg, "top_p", 0.95, 1);
    p->min_p = (float)cfg_real(cfg, "min_p", 0.05, 1);
    p->repetition_penalty = (float)cfg_real(cfg, "rep_penalty", 1.2, 1);
    p->stop_token = (int32_t)cfg_real(cfg, "stop_token", 6562, 1);
    p->start_token = (int32_t)cfg_real(cfg, "start_token", 6561, 0);
    p->sampler = (int32_t)cfg_real(cfg, "sampler", T3MI_SAMPLER_R, 0);
    p->top_k = (int32_t)cfg_real(cfg, "top_k", 0, 0);                 /* Turbo (.turbo_sample_token, R/t3.R:1688-1742) */
    p->max_cache_len = (int32_t)cfg_real(cfg, "max_cache_len", 0, 0);
    p->seed = (uint64_t)cfg_real(cfg, "seed", 0, 0);
    p->first_utterance_id = (int64_t)cfg_real(cfg, "first_utterance_id", 0, 0);
    p->drop_invalid_from = (int32_t)cfg_real(cfg, "drop_invalid_from", 0, 0);   /* drop_invalid_tokens on device (R/tts.R:366) */
}

/* R arrays are column-major doubles: embeds[r, t, k] sits at r + rows*(t + s0*k).  Returns NULL for a wrong shape or
 * when out of memory -- never longjmps, so a caller that already holds C allocations can free them first. */
static float* to_row_major(SEXP embeds, int rows, int s0, int d) {
    const R_xlen_t n = (R_xlen_t)rows * s0 * d;
    if (rows <= 0 || s0 <= 0 || d <= 0 || !Rf_isReal(embeds) || Rf_xlength(embeds) != n) return NULL;
    float* out = (float*)malloc((size_t)n * sizeof(float));
    if (!out) return NULL;
    const double* src = REAL(embeds);
    for (int r = 0; r < rows; r++)
        for (int t = 0; t < s0; t++)
            for (int k = 0; k < d; k++)
                out[((size_t)r * s0 + t) * d + k] = (float)src[r + (R_xlen_t)rows * (t + (R_xlen_t)s0 * k)];
    return out;
}

/* .Call("t3mi_r_generate", model, embeds, c(rows, s0, d), config) -> integer vector of 0-indexed
 * token ids including the stop token if hit -- the return contract of cpp_t3_decode
 * (src/t3_decode.cpp:420-427). */
SEXP t3mi_r_generate(SEXP model, SEXP embeds, SEXP dims, SEXP cfg) {
    t3mi_model* m = get_model(model);
    if (Rf_length(dims) != 3) Rf_error("t3mi: dims must be c(rows, S0, d)");
    const int rows = INTEGER(dims)[0], s0 = INTEGER(dims)[1], d = INTEGER(dims)[2];
    t3mi_params p;
    fill_params(cfg, &p);
    float* e = to_row_major(embeds, rows, s0, d);
    const int cap = (p.max_new_tokens > 0 ? p.max_new_tokens : 1) + 1;   /* Turbo returns up to max_tokens + 1 ids */
    int32_t* out = (int32_t*)malloc((size_t)cap * sizeof(int32_t));
    int32_t n = 0;
    const int bad_args = !e || !out;
    int rc = bad_args ? T3MI_ERR_ARG : t3mi_generate(m, e, rows, s0, &p, out, cap, &n);
    free(e);
    if (rc < 0) { /* free C state first, then longjmp (SURVEY §8b "Errors") */
        free(out);
        Rf_error("t3mi_generate: %s", bad_args ? "embeds must be a numeric [rows, S0, d] array" : t3mi_last_error());
    }
    SEXP res = PROTECT(Rf_allocVector(INTSXP, n));
    for (int i = 0; i < n; i++) INTEGER(res)[i] = (int)out[i];
    free(out);
    if (rc == T3MI_WARN_CACHE_FULL) Rf_warning("KV cache full, stopping generation"); /* src/t3_decode.cpp:395 */
    UNPROTECT(1);
    return res;
}

/* .Call("t3mi_r_generate_batch", model, list_of_embeds, matrix dims [n_utt, 3], config) -> list of
 * integer vectors (the batched front door for tts_chunked, R/tts.R:462-474). */
SEXP t3mi_r_generate_batch(SEXP model, SEXP embeds_list, SEXP dims, SEXP cfg) {
    t3mi_model* m = get_model(model);
    const int n_utt = Rf_length(embeds_list);
    if (n_utt <= 0 || Rf_length(dims) != 3 * n_utt) Rf_error("t3mi: dims must be an [n_utt, 3] integer matrix");
    t3mi_params p;
    fill_params(cfg, &p);
    const int* dm = INTEGER(dims); /* column-major [n_utt, 3] */
    const int rows = dm[0], d = dm[2 * n_utt];
    float** e = (float**)calloc((size_t)n_utt, sizeof(float*));
    int32_t* s0 = (int32_t*)malloc((size_t)n_utt * sizeof(int32_t));
    const int cap = (p.max_new_tokens > 0 ? p.max_new_tokens : 1) + 1;   /* Turbo returns up to max_tokens + 1 ids */
    int32_t* out = (int32_t*)malloc((size_t)n_utt * cap * sizeof(int32_t));
    int32_t* n_out = (int32_t*)calloc((size_t)n_utt, sizeof(int32_t));
    int rc = T3MI_OK;
    if (!e || !s0 || !out || !n_out) rc = T3MI_ERR_ARG;
    for (int u = 0; u < n_utt && rc == T3MI_OK; u++) {
        if (dm[u] != rows || dm[2 * n_utt + u] != d) { rc = T3MI_ERR_ARG; break; }
        s0[u] = dm[n_utt + u];
        e[u] = to_row_major(VECTOR_ELT(embeds_list, u), rows, s0[u], d);
        if (!e[u]) rc = T3MI_ERR_ARG;   /* wrong shape / out of memory: everything is freed below before Rf_error */
    }
    /* slots > 0 in the config: continuous batching (t3mi_generate_continuous) -- the utterances go through that many batch slots,
     * a slot takes the next one as soon as its own has ended; max_tokens_each (integer, one per utterance) are their own limits */
    const int slots = (int)cfg_real(cfg, "slots", 0, 0), poll = (int)cfg_real(cfg, "poll_steps", 16, 0);
    const int lflag = cfg_real(cfg, "longest_first", 0, 0) != 0 ? T3MI_CONTINUOUS_LONGEST_FIRST : 0;
    int n_each = 0;
    const int* each = cfg_ints(cfg, "max_tokens_each", &n_each);
    if (rc == T3MI_OK && each && n_each != n_utt) rc = T3MI_ERR_ARG;
    const int bad_args = rc != T3MI_OK;   /* refused here (shapes, memory), as opposed to refused by the library */
    if (rc == T3MI_OK) {
        if (slots > 0) rc = t3mi_generate_continuous(m, n_utt, rows, (const float* const*)e, s0, (const int32_t*)each, slots, poll, lflag, &p, out, cap, n_out, NULL);
        else rc = t3mi_generate_batch(m, n_utt, rows, (const float* const*)e, s0, &p, out, cap, n_out);
    }
    if (e) for (int u = 0; u < n_utt; u++) free(e[u]);
    free(e);
    free(s0);
    if (rc < 0) {
        free(out);
        free(n_out);
        Rf_error("t3mi_generate_batch: %s", bad_args ? "bad arguments: every element must be a numeric [rows, S0, d] array "
                 "matching its row of dims" : t3mi_last_error());
    }
    SEXP res = PROTECT(Rf_allocVector(VECSXP, n_utt));
    for (int u = 0; u < n_utt; u++) {
        SEXP v = PROTECT(Rf_allocVector(INTSXP, n_out[u]));
        for (int i = 0; i < n_out[u]; i++) INTEGER(v)[i] = (int)out[(size_t)u * cap + i];
        SET_VECTOR_ELT(res, u, v);
        UNPROTECT(1);
    }
    free(out);
    free(n_out);
    UNPROTECT(1);
    return res;
}

/* one utterance's conditioning from plain R vectors: list(speaker_emb = numeric, tokens = integer or NULL,
 * emotion_adv = numeric(1) or NULL) -- the fields of t3_cond (R/t3.R:103-113) after as.numeric / as.integer.
 * The float / int32 copies live in `hold` (caller frees). */
typedef struct { float* spk; int32_t* tok; int32_t* text; } cond_hold;

static int fill_cond(SEXP cond, SEXP text, t3mi_cond* c, cond_hold* h, int32_t* n_text) {
    memset(c, 0, sizeof(*c));
    memset(h, 0, sizeof(*h));
    SEXP names = Rf_getAttrib(cond, R_NamesSymbol);
    SEXP spk = R_NilValue, tok = R_NilValue, emo = R_NilValue;
    if (TYPEOF(cond) != VECSXP || names == R_NilValue) return 0;
    for (R_xlen_t i = 0; i < Rf_xlength(cond); i++) {
        const char* nm = CHAR(STRING_ELT(names, i));
        if (strcmp(nm, "speaker_emb") == 0) spk = VECTOR_ELT(cond, i);
        else if (strcmp(nm, "tokens") == 0) tok = VECTOR_ELT(cond, i);
        else if (strcmp(nm, "emotion_adv") == 0) emo = VECTOR_ELT(cond, i);
    }
    if (!Rf_isReal(spk) || !Rf_isInteger(text) || (tok != R_NilValue && !Rf_isInteger(tok))) return 0;
    const R_xlen_t ns = Rf_xlength(spk), nt = tok == R_NilValue ? 0 : Rf_xlength(tok), nx = Rf_xlength(text);
    h->spk = (float*)malloc((size_t)(ns > 0 ? ns : 1) * sizeof(float));
    h->tok = (int32_t*)malloc((size_t)(nt > 0 ? nt : 1) * sizeof(int32_t));
    h->text = (int32_t*)malloc((size_t)(nx > 0 ? nx : 1) * sizeof(int32_t));
    if (!h->spk || !h->tok || !h->text) return 0;
    for (R_xlen_t i = 0; i < ns; i++) h->spk[i] = (float)REAL(spk)[i];
    for (R_xlen_t i = 0; i < nt; i++) h->tok[i] = (int32_t)INTEGER(tok)[i];
    for (R_xlen_t i = 0; i < nx; i++) h->text[i] = (int32_t)INTEGER(text)[i];
    c->speaker_emb = h->spk;
    c->n_speaker = (int32_t)ns;
    c->cond_prompt_speech_tokens = nt > 0 ? h->tok : NULL;
    c->n_cond_tokens = (int32_t)nt;
    c->has_emotion_adv = emo != R_NilValue && Rf_length(emo) > 0;
    c->emotion_adv = c->has_emotion_adv ? (float)Rf_asReal(emo) : 0.0f;
    *n_text = (int32_t)nx;
    return 1;
}

static void free_hold(cond_hold* h, int n) {
    if (!h) return;
    for (int u = 0; u < n; u++) { free(h[u].spk); free(h[u].tok); free(h[u].text); }
    free(h);
}

/* .Call("t3mi_r_generate_tokens", model, list_of_cond, list_of_text_tokens, config) -> list of integer vectors.
 * The whole of t3_inference(model, cond, text_tokens, ...) (R/t3.R:461-605) for one or many utterances: prompt
 * built on the GPU (prepare_input_embeds, R/t3.R:372-403), prefill, decode loop.  Text tokens WITHOUT the
 * start / stop text tokens (added by the library as R/t3.R:473-477 does), 0-indexed as the tokenizer gives them. */
SEXP t3mi_r_generate_tokens(SEXP model, SEXP conds, SEXP texts, SEXP cfg) {
    t3mi_model* m = get_model(model);
    const int n_utt = Rf_length(conds);
    if (n_utt <= 0 || TYPEOF(conds) != VECSXP || TYPEOF(texts) != VECSXP || Rf_length(texts) != n_utt)
        Rf_error("t3mi: conds and text_tokens must be lists of the same positive length");
    t3mi_params p;
    fill_params(cfg, &p);
    const int cap = (p.max_new_tokens > 0 ? p.max_new_tokens : 1) + 1;   /* Turbo returns up to max_tokens + 1 ids */
    t3mi_cond* c = (t3mi_cond*)calloc((size_t)n_utt, sizeof(t3mi_cond));
    cond_hold* h = (cond_hold*)calloc((size_t)n_utt, sizeof(cond_hold));
    const int32_t** tp = (const int32_t**)calloc((size_t)n_utt, sizeof(int32_t*));
    int32_t* nt = (int32_t*)calloc((size_t)n_utt, sizeof(int32_t));
    int32_t* out = (int32_t*)malloc((size_t)n_utt * cap * sizeof(int32_t));
    int32_t* n_out = (int32_t*)calloc((size_t)n_utt, sizeof(int32_t));
    int rc = (c && h && tp && nt && out && n_out) ? T3MI_OK : T3MI_ERR_ARG;
    for (int u = 0; u < n_utt && rc == T3MI_OK; u++) {
        if (!fill_cond(VECTOR_ELT(conds, u), VECTOR_ELT(texts, u), &c[u], &h[u], &nt[u])) rc = T3MI_ERR_ARG;
        tp[u] = h[u].text;
    }
    const int slots = (int)cfg_real(cfg, "slots", 0, 0), poll = (int)cfg_real(cfg, "poll_steps", 16, 0);   /* as in t3mi_r_generate_batch */
    const int lflag = cfg_real(cfg, "longest_first", 0, 0) != 0 ? T3MI_CONTINUOUS_LONGEST_FIRST : 0;
    int n_each = 0;
    const int* each = cfg_ints(cfg, "max_tokens_each", &n_each);
    if (rc == T3MI_OK && each && n_each != n_utt) rc = T3MI_ERR_ARG;
    const int bad_args = rc != T3MI_OK;
    if (rc == T3MI_OK) {
        if (slots > 0) rc = t3mi_generate_continuous_tokens(m, n_utt, c, tp, nt, (const int32_t*)each, slots, poll, lflag, &p, out, cap, n_out, NULL);
        else rc = t3mi_generate_batch_tokens(m, n_utt, c, tp, nt, &p, out, cap, n_out);
    }
    free(c);
    free_hold(h, n_utt);
    free((void*)tp);
    free(nt);
    if (rc < 0) { /* free C state first, then longjmp (SURVEY 8b "Errors") */
        free(out);
        free(n_out);
        Rf_error("t3mi_generate_tokens: %s", bad_args ? "cond must be list(speaker_emb = numeric, tokens = integer, "
                 "emotion_adv = numeric), text_tokens integer" : t3mi_last_error());
    }
    SEXP res = PROTECT(Rf_allocVector(VECSXP, n_utt));
    for (int u = 0; u < n_utt; u++) {
        SEXP v = PROTECT(Rf_allocVector(INTSXP, n_out[u]));
        for (int i = 0; i < n_out[u]; i++) INTEGER(v)[i] = (int)out[(size_t)u * cap + i];
        SET_VECTOR_ELT(res, u, v);
        UNPROTECT(1);
    }
    free(out);
    free(n_out);
    if (rc == T3MI_WARN_CACHE_FULL) Rf_warning("KV cache full, stopping generation"); /* src/t3_decode.cpp:395 */
    UNPROTECT(1);
    return res;
}

#endif /* HAVE_T3MI */

/* S3Gen entry points (r_glue_s3gen.c) */
SEXP t3mi_r_flow_open(SEXP, SEXP);
SEXP t3mi_r_hift_open(SEXP, SEXP);
SEXP t3mi_r_flow_inference(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP);
SEXP t3mi_r_hift_inference(SEXP, SEXP, SEXP, SEXP, SEXP);

static const R_CallMethodDef call_methods[] = {
    {"t3mi_r_open", (DL_FUNC)&t3mi_r_open, 5},
    {"t3mi_r_close", (DL_FUNC)&t3mi_r_close, 1},
    {"t3mi_r_generate", (DL_FUNC)&t3mi_r_generate, 4},
    {"t3mi_r_generate_batch", (DL_FUNC)&t3mi_r_generate_batch, 4},
    {"t3mi_r_generate_tokens", (DL_FUNC)&t3mi_r_generate_tokens, 4},
    {"t3mi_r_flow_open", (DL_FUNC)&t3mi_r_flow_open, 2},
    {"t3mi_r_hift_open", (DL_FUNC)&t3mi_r_hift_open, 2},
    {"t3mi_r_flow_inference", (DL_FUNC)&t3mi_r_flow_inference, 7},
    {"t3mi_r_hift_inference", (DL_FUNC)&t3mi_r_hift_inference, 5},
    {NULL, NULL, 0}};

/* same registration pattern as the reference's src/init.c:12-15 */
void R_init_chatterbox(DllInfo* dll) {
    R_registerRoutines(dll, NULL, call_methods, NULL, NULL);
    R_useDynamicSymbols(dll, FALSE);
}
