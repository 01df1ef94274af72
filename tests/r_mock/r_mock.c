/* TEST DOUBLE of the R C API (see include/Rinternals.h) plus a harness that drives r-package/src/r_glue.c the way R's
 * .Call would: it builds the argument SEXPs (named lists, integer / double vectors, column-major arrays), calls the
 * registered routines through the table R_init_chatterbox registers, and hands the results back as plain C arrays.
 * Rf_error is a longjmp back into the harness (as R's is a longjmp to top level): the glue's "free C state first,
 * then raise" paths really run. */
#include <R.h>
#include <Rinternals.h>
#include <R_ext/Rdynload.h>
#include <setjmp.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

struct r_mock_sexp {
    int type;
    R_xlen_t n;
    void* data;          /* int* / double* / SEXP* / char* */
    SEXP names;          /* STRSXP or NULL */
    void* ext;
    R_CFinalizer_t fin;
};
struct r_mock_dll { int dummy; };

static struct r_mock_sexp nil_obj = {NILSXP, 0, NULL, NULL, NULL, NULL}, names_sym = {NILSXP, 0, NULL, NULL, NULL, NULL};
SEXP R_NilValue = &nil_obj, R_NamesSymbol = &names_sym;

/* every allocation of a harness call is tracked and released at its end (the mock has no garbage collector) */
#define MAX_OBJ 65536
static SEXP g_obj[MAX_OBJ];
static int g_nobj = 0, g_protect = 0;
static jmp_buf g_top;
static int g_top_set = 0;
static char g_error[1024], g_warning[1024];
static const R_CallMethodDef* g_calls = NULL;

static SEXP new_obj(int type, R_xlen_t n, size_t elem) {
    SEXP s = (SEXP)calloc(1, sizeof(*s));
    s->type = type;
    s->n = n;
    s->data = calloc((size_t)(n > 0 ? n : 1), elem);
    if (g_nobj < MAX_OBJ) g_obj[g_nobj++] = s;
    return s;
}
static void free_all(void) {
    for (int i = 0; i < g_nobj; i++) {
        if (g_obj[i]->type == EXTPTRSXP && g_obj[i]->fin && g_obj[i]->ext) g_obj[i]->fin(g_obj[i]);   /* "gc" runs the finalizer */
        free(g_obj[i]->data);
        free(g_obj[i]);
    }
    g_nobj = 0;
}

int TYPEOF(SEXP x) { return x->type; }
int Rf_length(SEXP x) { return (int)x->n; }
R_xlen_t Rf_xlength(SEXP x) { return x->n; }
int* INTEGER(SEXP x) { return (int*)x->data; }
double* REAL(SEXP x) { return (double*)x->data; }
const char* CHAR(SEXP x) { return (const char*)x->data; }
SEXP STRING_ELT(SEXP x, R_xlen_t i) { return ((SEXP*)x->data)[i]; }
SEXP VECTOR_ELT(SEXP x, R_xlen_t i) { return ((SEXP*)x->data)[i]; }
SEXP SET_VECTOR_ELT(SEXP x, R_xlen_t i, SEXP v) { ((SEXP*)x->data)[i] = v; return v; }
SEXP Rf_getAttrib(SEXP x, SEXP name) { return (name == R_NamesSymbol && x->names) ? x->names : R_NilValue; }
SEXP Rf_allocVector(unsigned type, R_xlen_t n) {
    SEXP s = new_obj((int)type, n, type == INTSXP ? sizeof(int) : (type == REALSXP ? sizeof(double) : sizeof(SEXP)));
    if (type == VECSXP || type == STRSXP) for (R_xlen_t i = 0; i < n; i++) ((SEXP*)s->data)[i] = R_NilValue;
    return s;
}
int Rf_asInteger(SEXP x) { return x->type == INTSXP ? INTEGER(x)[0] : (x->type == REALSXP ? (int)REAL(x)[0] : 0); }
double Rf_asReal(SEXP x) { return x->type == REALSXP ? REAL(x)[0] : (x->type == INTSXP ? (double)INTEGER(x)[0] : 0.0); }
Rboolean Rf_isReal(SEXP x) { return x->type == REALSXP ? TRUE : FALSE; }
Rboolean Rf_isInteger(SEXP x) { return x->type == INTSXP ? TRUE : FALSE; }
SEXP Rf_protect(SEXP x) { g_protect++; return x; }
void Rf_unprotect(int n) { g_protect -= n; }
void Rf_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
    if (!g_top_set) { fprintf(stderr, "Rf_error outside a harness call: %s\n", g_error); abort(); }
    longjmp(g_top, 1);
}
void Rf_warning(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_warning, sizeof(g_warning), fmt, ap);
    va_end(ap);
}
SEXP R_MakeExternalPtr(void* p, SEXP tag, SEXP prot) { (void)tag; (void)prot; SEXP s = new_obj(EXTPTRSXP, 0, 1); s->ext = p; return s; }
void* R_ExternalPtrAddr(SEXP s) { return s->ext; }
void R_ClearExternalPtr(SEXP s) { s->ext = NULL; }
void R_RegisterCFinalizerEx(SEXP s, R_CFinalizer_t fun, Rboolean onexit) { (void)onexit; s->fin = fun; }
int R_registerRoutines(DllInfo* info, const void* c, const R_CallMethodDef* call, const void* f, const void* e) {
    (void)info; (void)c; (void)f; (void)e; g_calls = call; return 1;
}
Rboolean R_useDynamicSymbols(DllInfo* info, Rboolean value) { (void)info; return value; }

/* ---- constructors the harness uses (what R code would pass) -------------------------------------------------------- */
static SEXP mk_str1(const char* v) {
    SEXP c = new_obj(CHARSXP, (R_xlen_t)strlen(v), 1);
    free(c->data);
    c->data = strdup(v);
    SEXP s = Rf_allocVector(STRSXP, 1);
    ((SEXP*)s->data)[0] = c;
    return s;
}
static SEXP mk_int(const int32_t* v, R_xlen_t n) { SEXP s = Rf_allocVector(INTSXP, n); for (R_xlen_t i = 0; i < n; i++) INTEGER(s)[i] = v[i]; return s; }
static SEXP mk_int1(int v) { int32_t x = v; return mk_int(&x, 1); }
static SEXP mk_real1(double v) { SEXP s = Rf_allocVector(REALSXP, 1); REAL(s)[0] = v; return s; }
static SEXP mk_names(SEXP list, const char* const* names) {
    SEXP nm = Rf_allocVector(STRSXP, list->n);
    for (R_xlen_t i = 0; i < list->n; i++) ((SEXP*)nm->data)[i] = STRING_ELT(mk_str1(names[i]), 0);
    list->names = nm;
    return list;
}

extern void R_init_chatterbox(DllInfo* dll);
typedef SEXP (*call4)(SEXP, SEXP, SEXP, SEXP);
typedef SEXP (*call5)(SEXP, SEXP, SEXP, SEXP, SEXP);
typedef SEXP (*call1)(SEXP);
static DL_FUNC find_call(const char* name, int nargs) {
    if (!g_calls) { struct r_mock_dll d; R_init_chatterbox(&d); }
    for (const R_CallMethodDef* c = g_calls; c && c->name; c++)
        if (strcmp(c->name, name) == 0 && c->numArgs == nargs) return c->fun;
    return NULL;
}

/* decode parameters as the named list the R wrappers build (R/t3.R:1022-1034 names) */
typedef struct {
    int max_tokens, sampler, top_k, max_cache_len, start_token, stop_token, omit_required;
    double temperature, cfg_weight, top_p, min_p, rep_penalty, seed;
    int slots, poll_steps, n_each;          /* slots > 0: continuous batching; n_each > 0: max_tokens_each follows */
    const int32_t* each;
} mock_cfg;
static SEXP mk_cfg(const mock_cfg* c) {
    static const char* const nm[] = {"max_tokens", "temperature", "cfg_weight", "top_p", "min_p", "rep_penalty", "stop_token",
                                     "start_token", "sampler", "top_k", "max_cache_len", "seed"};
    const int n = c->omit_required ? 5 : 12;        /* omit_required: the list lacks rep_penalty .. (error path) */
    SEXP l = Rf_allocVector(VECSXP, n);
    SEXP v[12] = {mk_int1(c->max_tokens), mk_real1(c->temperature), mk_real1(c->cfg_weight), mk_real1(c->top_p), mk_real1(c->min_p),
                  mk_real1(c->rep_penalty), mk_int1(c->stop_token), mk_int1(c->start_token), mk_int1(c->sampler), mk_int1(c->top_k),
                  mk_int1(c->max_cache_len), mk_real1(c->seed)};
    if (!c->omit_required && c->slots > 0) {
        static const char* const nx[] = {"max_tokens", "temperature", "cfg_weight", "top_p", "min_p", "rep_penalty", "stop_token",
                                         "start_token", "sampler", "top_k", "max_cache_len", "seed", "slots", "poll_steps", "max_tokens_each"};
        const int nn = c->n_each > 0 ? 15 : 14;
        SEXP lx = Rf_allocVector(VECSXP, nn);
        for (int i = 0; i < 12; i++) SET_VECTOR_ELT(lx, i, v[i]);
        SET_VECTOR_ELT(lx, 12, mk_int1(c->slots));
        SET_VECTOR_ELT(lx, 13, mk_int1(c->poll_steps));
        if (c->n_each > 0) SET_VECTOR_ELT(lx, 14, mk_int(c->each, c->n_each));
        return mk_names(lx, nx);
    }
    for (int i = 0; i < n; i++) SET_VECTOR_ELT(l, i, v[i]);
    return mk_names(l, nm);
}

static void* g_model_sexp = NULL;   /* the external pointer lives across harness calls, like an R object */
static struct r_mock_sexp g_model_obj;

const char* mock_last_error(void) { return g_error; }
const char* mock_last_warning(void) { return g_warning; }
int mock_protect_balance(void) { return g_protect; }

#define ENTER()                                   \
    g_error[0] = 0;                               \
    g_warning[0] = 0;                             \
    g_top_set = 1;                                \
    if (setjmp(g_top)) { g_top_set = 0; free_all(); return -1; }
#define LEAVE(rc) do { g_top_set = 0; free_all(); return (rc); } while (0)

/* .Call("t3mi_r_open", path, variant, device, dtype, cache) */
int mock_open(const char* path, int variant, int device, int dtype, const char* cache) {
    ENTER();
    call5 fn = (call5)find_call("t3mi_r_open", 5);
    if (!fn) { snprintf(g_error, sizeof(g_error), "t3mi_r_open not registered"); LEAVE(-2); }
    SEXP ptr = fn(mk_str1(path), mk_int1(variant), mk_int1(device), mk_int1(dtype), cache ? mk_str1(cache) : R_NilValue);
    /* keep the handle beyond this call: copy the object out of the per-call arena (its finalizer must not run now) */
    g_model_obj = *ptr;
    ptr->fin = NULL;
    ptr->ext = NULL;
    g_model_sexp = &g_model_obj;
    LEAVE(0);
}
int mock_close(void) {
    ENTER();
    call1 fn = (call1)find_call("t3mi_r_close", 1);
    if (fn && g_model_sexp) fn((SEXP)g_model_sexp);
    const int cleared = g_model_sexp && R_ExternalPtrAddr((SEXP)g_model_sexp) == NULL;
    g_model_sexp = NULL;
    LEAVE(cleared ? 0 : -3);
}

static int copy_list_out(SEXP res, int n_utt, int32_t* out, int cap, int32_t* n_out) {
    if (TYPEOF(res) != VECSXP || Rf_length(res) != n_utt) return -4;
    for (int u = 0; u < n_utt; u++) {
        SEXP v = VECTOR_ELT(res, u);
        if (TYPEOF(v) != INTSXP || Rf_length(v) > cap) return -5;
        n_out[u] = Rf_length(v);
        for (int i = 0; i < n_out[u]; i++) out[(size_t)u * cap + i] = INTEGER(v)[i];
    }
    return 0;
}

/* .Call("t3mi_r_generate_tokens", model, list(cond...), list(text...), cfg): cond = list(speaker_emb, tokens, emotion_adv) */
int mock_generate_tokens(int n_utt, const double* spk, int n_spk, const int32_t* const* ctok, const int32_t* n_ctok,
                         const double* emotion /* NaN = NULL */, const int32_t* const* text, const int32_t* n_text,
                         const mock_cfg* cfg, int wrong_types, int32_t* out, int cap, int32_t* n_out) {
    ENTER();
    call4 fn = (call4)find_call("t3mi_r_generate_tokens", 4);
    if (!fn || !g_model_sexp) { snprintf(g_error, sizeof(g_error), "no model / routine"); LEAVE(-2); }
    static const char* const cn[] = {"speaker_emb", "tokens", "emotion_adv"};
    SEXP conds = Rf_allocVector(VECSXP, n_utt), texts = Rf_allocVector(VECSXP, n_utt);
    for (int u = 0; u < n_utt; u++) {
        SEXP c = Rf_allocVector(VECSXP, 3);
        SEXP s = Rf_allocVector(REALSXP, n_spk);
        for (int i = 0; i < n_spk; i++) REAL(s)[i] = spk[(size_t)u * n_spk + i];
        SET_VECTOR_ELT(c, 0, s);
        SET_VECTOR_ELT(c, 1, n_ctok[u] > 0 ? mk_int(ctok[u], n_ctok[u]) : R_NilValue);
        SET_VECTOR_ELT(c, 2, emotion[u] != emotion[u] ? R_NilValue : mk_real1(emotion[u]));
        SET_VECTOR_ELT(conds, u, mk_names(c, cn));
        /* wrong_types: the text ids arrive as doubles (a caller that forgot as.integer): the glue must refuse, not misread */
        if (wrong_types) { SEXP t = Rf_allocVector(REALSXP, n_text[u]); SET_VECTOR_ELT(texts, u, t); }
        else SET_VECTOR_ELT(texts, u, mk_int(text[u], n_text[u]));
    }
    SEXP res = fn((SEXP)g_model_sexp, conds, texts, mk_cfg(cfg));
    const int rc = copy_list_out(res, n_utt, out, cap, n_out);
    LEAVE(rc);
}

/* .Call("t3mi_r_generate", model, embeds, dims, cfg) with embeds as R holds a [rows, s0, d] array: column-major doubles */
int mock_generate(const float* embeds_row_major, int rows, int s0, int d, const mock_cfg* cfg, int bad_dims, int32_t* out, int cap,
                  int32_t* n_out) {
    ENTER();
    call4 fn = (call4)find_call("t3mi_r_generate", 4);
    if (!fn || !g_model_sexp) { snprintf(g_error, sizeof(g_error), "no model / routine"); LEAVE(-2); }
    SEXP e = Rf_allocVector(REALSXP, (R_xlen_t)rows * s0 * d);
    for (int r = 0; r < rows; r++)
        for (int t = 0; t < s0; t++)
            for (int k = 0; k < d; k++)
                REAL(e)[r + (R_xlen_t)rows * (t + (R_xlen_t)s0 * k)] = embeds_row_major[((size_t)r * s0 + t) * d + k];
    int32_t dm[3] = {rows, bad_dims ? s0 + 1 : s0, d};
    SEXP res = fn((SEXP)g_model_sexp, e, mk_int(dm, 3), mk_cfg(cfg));
    if (TYPEOF(res) != INTSXP || Rf_length(res) > cap) LEAVE(-5);
    *n_out = Rf_length(res);
    for (int i = 0; i < *n_out; i++) out[i] = INTEGER(res)[i];
    LEAVE(0);
}

/* .Call("t3mi_r_generate_batch", model, list(embeds...), dims [n_utt, 3] column-major, cfg) */
int mock_generate_batch(int n_utt, const float* const* embeds_row_major, int rows, const int32_t* s0, int d, const mock_cfg* cfg,
                        int bad_utt /* index whose array gets the wrong length, or -1 */, int32_t* out, int cap, int32_t* n_out) {
    ENTER();
    call4 fn = (call4)find_call("t3mi_r_generate_batch", 4);
    if (!fn || !g_model_sexp) { snprintf(g_error, sizeof(g_error), "no model / routine"); LEAVE(-2); }
    SEXP list = Rf_allocVector(VECSXP, n_utt);
    SEXP dims = Rf_allocVector(INTSXP, 3 * (R_xlen_t)n_utt);
    for (int u = 0; u < n_utt; u++) {
        const R_xlen_t n = (R_xlen_t)rows * s0[u] * d;
        SEXP e = Rf_allocVector(REALSXP, u == bad_utt ? n - 1 : n);
        if (u != bad_utt)
            for (int r = 0; r < rows; r++)
                for (int t = 0; t < s0[u]; t++)
                    for (int k = 0; k < d; k++)
                        REAL(e)[r + (R_xlen_t)rows * (t + (R_xlen_t)s0[u] * k)] = embeds_row_major[u][((size_t)r * s0[u] + t) * d + k];
        SET_VECTOR_ELT(list, u, e);
        INTEGER(dims)[u] = rows;
        INTEGER(dims)[n_utt + u] = s0[u];
        INTEGER(dims)[2 * n_utt + u] = d;
    }
    SEXP res = fn((SEXP)g_model_sexp, list, dims, mk_cfg(cfg));
    const int rc = copy_list_out(res, n_utt, out, cap, n_out);
    LEAVE(rc);
}

/* ---- S3Gen (r_glue_s3gen.c): open both handles, tokens -> mel -> waveform through the four .Call entry points; the handles
 * live in the per-call arena, so their finalizers (run by free_all, as R's GC would) close them on every path. ---- */
typedef SEXP (*call2)(SEXP, SEXP);
typedef SEXP (*call7)(SEXP, SEXP, SEXP, SEXP, SEXP, SEXP, SEXP);
static SEXP mk_real_f(const float* v, R_xlen_t n) { SEXP s = Rf_allocVector(REALSXP, n); for (R_xlen_t i = 0; i < n; i++) REAL(s)[i] = v[i]; return s; }

/* bad: 0 none, 1 token ids as doubles, 2 prompt_feat one value short, 3 token id outside the vocabulary, 4 mel one value short */
int mock_s3gen(const char* flow_path, const char* hift_path, const int32_t* tok, int n_tok, const int32_t* ptok, int n_ptok,
               const float* pfeat, int n_feat, int mel_bins, const float* emb, int n_emb, const float* noise, int noise_T, int steps,
               double cfg_rate, int meanflow, int finalize, const float* phase, int n_harm, const float* src_noise, int bad,
               float* mel_out, int cap_frames, int32_t* n_frames, float* audio_out) {
    ENTER();
    call2 fopen_ = (call2)find_call("t3mi_r_flow_open", 2), hopen = (call2)find_call("t3mi_r_hift_open", 2);
    call7 finf = (call7)find_call("t3mi_r_flow_inference", 7);
    call5 hinf = (call5)find_call("t3mi_r_hift_inference", 5);
    if (!fopen_ || !hopen || !finf || !hinf) { snprintf(g_error, sizeof(g_error), "S3Gen routines not registered"); LEAVE(-2); }
    SEXP flow = fopen_(mk_str1(flow_path), mk_int1(0));
    SEXP voc = hopen(mk_str1(hift_path), mk_int1(0));
    SEXP cfg = Rf_allocVector(REALSXP, 6);
    const double c[6] = {(double)steps, cfg_rate, (double)meanflow, (double)finalize, (double)n_feat, (double)noise_T};
    for (int i = 0; i < 6; i++) REAL(cfg)[i] = c[i];
    SEXP tk;
    if (bad == 1) tk = Rf_allocVector(REALSXP, n_tok);
    else {
        tk = mk_int(tok, n_tok);
        if (bad == 3) INTEGER(tk)[0] = 1 << 20;
    }
    SEXP mel = finf(flow, tk, mk_int(ptok, n_ptok), mk_real_f(pfeat, (R_xlen_t)n_feat * mel_bins - (bad == 2)), mk_real_f(emb, n_emb),
                    mk_real_f(noise, (R_xlen_t)noise_T * mel_bins), cfg);
    if (TYPEOF(mel) != REALSXP || Rf_xlength(mel) % mel_bins || Rf_xlength(mel) / mel_bins > cap_frames) LEAVE(-5);
    const int T = (int)(Rf_xlength(mel) / mel_bins);
    *n_frames = T;
    for (R_xlen_t i = 0; i < Rf_xlength(mel); i++) mel_out[i] = (float)REAL(mel)[i];
    SEXP mel_in = mel;
    if (bad == 4) mel_in = Rf_allocVector(REALSXP, Rf_xlength(mel) - 1);
    SEXP wav = hinf(voc, mel_in, mk_int1(T), phase ? mk_real_f(phase, n_harm) : R_NilValue,
                    src_noise ? mk_real_f(src_noise, (R_xlen_t)n_harm * 480 * T) : R_NilValue);
    if (TYPEOF(wav) != REALSXP || Rf_xlength(wav) != (R_xlen_t)480 * T) LEAVE(-6);
    for (R_xlen_t i = 0; i < Rf_xlength(wav); i++) audio_out[i] = (float)REAL(wav)[i];
    LEAVE(0);
}
