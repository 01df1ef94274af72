/* R glue for the S3Gen half of libt3mi: the .Call entry points behind `s3gen$inference`
 *     R/s3gen.R:987-1056  (flow: R/s3gen.R:805-890, R/conformer.R:517-578; vocoder: R/hifigan.R:601-626)
 * The reference runs these modules through R torch; here the flow module and the vocoder are handles that own their
 * weights in HBM (read from the unchanged s3gen.safetensors), and what crosses the boundary are plain R vectors:
 * integer token ids, the prompt mel as a numeric matrix, the x-vector, and the random draws the reference takes from
 * torch's generator (CFM start noise, the source's start phases and noise), which the R wrapper draws with rnorm /
 * runif so that set.seed() governs.  Plain C, R API only; registered from R_init_chatterbox in r_glue.c.
 */
#include <R.h>
#include <Rinternals.h>
#include <stdlib.h>
#include <string.h>

#ifdef HAVE_T3MI
#include "t3mi.h"

static void flow_finalizer(SEXP ptr) {
    t3mi_cfm* m = (t3mi_cfm*)R_ExternalPtrAddr(ptr);
    if (m) { t3mi_cfm_close(m); R_ClearExternalPtr(ptr); }
}
static void hift_finalizer(SEXP ptr) {
    t3mi_hift* m = (t3mi_hift*)R_ExternalPtrAddr(ptr);
    if (m) { t3mi_hift_close(m); R_ClearExternalPtr(ptr); }
}

/* .Call("t3mi_r_flow_open", path, device) -> external pointer (flow module + CFM estimator) */
SEXP t3mi_r_flow_open(SEXP path, SEXP device) {
    t3mi_cfm* m = NULL;
    if (TYPEOF(path) != STRSXP || Rf_length(path) != 1) Rf_error("t3mi_flow_open: path must be one string");
    if (t3mi_flow_open(CHAR(STRING_ELT(path, 0)), Rf_asInteger(device), &m) != T3MI_OK) Rf_error("t3mi_flow_open: %s", t3mi_last_error());
    SEXP ptr = PROTECT(R_MakeExternalPtr(m, R_NilValue, R_NilValue));
    R_RegisterCFinalizerEx(ptr, flow_finalizer, TRUE);
    UNPROTECT(1);
    return ptr;
}

/* .Call("t3mi_r_hift_open", path, device) -> external pointer (HiFT vocoder, keys under "mel2wav.") */
SEXP t3mi_r_hift_open(SEXP path, SEXP device) {
    t3mi_hift* m = NULL;
    if (TYPEOF(path) != STRSXP || Rf_length(path) != 1) Rf_error("t3mi_hift_open: path must be one string");
    if (t3mi_hift_open(CHAR(STRING_ELT(path, 0)), NULL, Rf_asInteger(device), &m) != T3MI_OK) Rf_error("t3mi_hift_open: %s", t3mi_last_error());
    SEXP ptr = PROTECT(R_MakeExternalPtr(m, R_NilValue, R_NilValue));
    R_RegisterCFinalizerEx(ptr, hift_finalizer, TRUE);
    UNPROTECT(1);
    return ptr;
}

static float* to_float(SEXP v, R_xlen_t n) {   /* numeric vector -> malloc'd float copy, NULL on a wrong type / length / no memory */
    if (!Rf_isReal(v) || Rf_xlength(v) != n || n <= 0) return NULL;
    float* out = (float*)malloc((size_t)n * sizeof(float));
    if (out) for (R_xlen_t i = 0; i < n; i++) out[i] = (float)REAL(v)[i];
    return out;
}

/* .Call("t3mi_r_flow_inference", flow, token, prompt_token, prompt_feat, embedding, noise, cfg) -> numeric vector of
 * 80 * frames values, mel-major (frames fastest): causal_masked_diff_xvec$forward (R/s3gen.R:805-890).
 *   token, prompt_token : integer, 0-indexed ids
 *   prompt_feat         : numeric, t(prompt_feat) flattened, i.e. frame-major [frames][80] in memory
 *   embedding           : numeric [192]
 *   noise               : numeric [80 * noise_T], mel-major (the reference's rand_noise[1, , 1:noise_T] flattened by row)
 *   cfg                 : c(n_timesteps, cfg_rate, meanflow, finalize, n_feat_frames, noise_T) */
SEXP t3mi_r_flow_inference(SEXP flow, SEXP token, SEXP prompt_token, SEXP prompt_feat, SEXP embedding, SEXP noise, SEXP cfg) {
    if (TYPEOF(flow) != EXTPTRSXP) Rf_error("t3mi: not a flow handle");
    t3mi_cfm* m = (t3mi_cfm*)R_ExternalPtrAddr(flow);
    if (!m) Rf_error("t3mi: flow handle was closed");
    if (!Rf_isInteger(token) || !Rf_isInteger(prompt_token) || !Rf_isReal(cfg) || Rf_length(cfg) != 6)
        Rf_error("t3mi_flow_inference: token ids must be integer vectors, cfg a numeric vector of 6");
    t3mi_flow_info fi;
    if (t3mi_flow_get_info(m, &fi) != T3MI_OK) Rf_error("t3mi_flow_inference: %s", t3mi_last_error());
    const double* c = REAL(cfg);
    const int n_tok = Rf_length(token), n_prompt = Rf_length(prompt_token), n_feat = (int)c[4], noise_T = (int)c[5];
    float* pf = to_float(prompt_feat, (R_xlen_t)n_feat * fi.mel);
    float* emb = to_float(embedding, fi.spk_embed_dim);
    float* nz = to_float(noise, (R_xlen_t)noise_T * fi.mel);
    const int cap = 2 * n_tok;
    float* out = (float*)malloc((size_t)(cap > 0 ? cap : 1) * fi.mel * sizeof(float));
    int frames = 0, rc = T3MI_ERR_ARG;
    const int bad_args = !pf || !emb || !nz || !out;
    if (!bad_args)
        rc = t3mi_flow_inference(m, (const int32_t*)INTEGER(token), n_tok, (const int32_t*)INTEGER(prompt_token), n_prompt, pf, n_feat, emb, nz,
                                 noise_T, (int)c[0], c[1], (int)c[2], (int)c[3], out, cap, &frames, NULL);
    free(pf); free(emb); free(nz);
    if (rc < 0) {   /* C state is freed before the longjmp */
        free(out);
        Rf_error("t3mi_flow_inference: %s", bad_args ? "prompt_feat / embedding / noise must be numeric vectors of the stated sizes" : t3mi_last_error());
    }
    SEXP res = PROTECT(Rf_allocVector(REALSXP, (R_xlen_t)frames * fi.mel));
    for (R_xlen_t i = 0; i < (R_xlen_t)frames * fi.mel; i++) REAL(res)[i] = (double)out[i];
    free(out);
    UNPROTECT(1);
    return res;
}

/* .Call("t3mi_r_hift_inference", vocoder, mel, n_frames, phase, noise) -> numeric vector of 480 * n_frames samples:
 * hift_generator$inference (R/hifigan.R:601-626).  mel: numeric [80 * n_frames] mel-major; phase: numeric [9] or NULL;
 * noise: numeric [9 * 480 * n_frames] harmonic-major, or NULL. */
SEXP t3mi_r_hift_inference(SEXP voc, SEXP mel, SEXP n_frames, SEXP phase, SEXP noise) {
    if (TYPEOF(voc) != EXTPTRSXP) Rf_error("t3mi: not a vocoder handle");
    t3mi_hift* m = (t3mi_hift*)R_ExternalPtrAddr(voc);
    if (!m) Rf_error("t3mi: vocoder handle was closed");
    t3mi_hift_info hi;
    if (t3mi_hift_get_info(m, &hi) != T3MI_OK) Rf_error("t3mi_hift_inference: %s", t3mi_last_error());
    const int T = Rf_asInteger(n_frames);
    const R_xlen_t n = (R_xlen_t)T * hi.samples_per_frame;
    float* ml = to_float(mel, (R_xlen_t)T * hi.mel);
    float* ph = phase == R_NilValue ? NULL : to_float(phase, hi.n_harmonics);
    float* nz = noise == R_NilValue ? NULL : to_float(noise, n * hi.n_harmonics);
    float* audio = (float*)malloc((size_t)(n > 0 ? n : 1) * sizeof(float));
    const int bad_args = !ml || !audio || (phase != R_NilValue && !ph) || (noise != R_NilValue && !nz);
    int rc = bad_args ? T3MI_ERR_ARG : t3mi_hift_inference(m, T, ml, ph, nz, NULL, 0, audio, NULL, NULL, NULL);
    free(ml); free(ph); free(nz);
    if (rc < 0) {
        free(audio);
        Rf_error("t3mi_hift_inference: %s", bad_args ? "mel / phase / noise must be numeric vectors of the stated sizes" : t3mi_last_error());
    }
    SEXP res = PROTECT(Rf_allocVector(REALSXP, n));
    for (R_xlen_t i = 0; i < n; i++) REAL(res)[i] = (double)audio[i];
    free(audio);
    UNPROTECT(1);
    return res;
}

#else  /* configure found no hipcc: the package loads, the backend raises when called (like src/t3_decode.cpp:432-446) */
static SEXP no_backend(void) {
    Rf_error("chatterbox was built without the MI355X backend (hipcc not found at configure time)");
    return R_NilValue;
}
SEXP t3mi_r_flow_open(SEXP a, SEXP b) { return no_backend(); }
SEXP t3mi_r_hift_open(SEXP a, SEXP b) { return no_backend(); }
SEXP t3mi_r_flow_inference(SEXP a, SEXP b, SEXP c, SEXP d, SEXP e, SEXP f, SEXP g) { return no_backend(); }
SEXP t3mi_r_hift_inference(SEXP a, SEXP b, SEXP c, SEXP d, SEXP e) { return no_backend(); }
#endif
