# MI355X-native T3 decode behind the reference's R API.
# Drop this file into the package's R/ directory (it replaces the bodies of t3_inference /
# t3_inference_cpp, R/t3.R:461-605, 932-1050; signatures and return values are unchanged) and add
#     useDynLib(chatterbox, .registration = TRUE)
# to NAMESPACE (missing in the reference, SURVEY Q18).

.t3mi <- new.env(parent = emptyenv())

#' Path of the T3 safetensors file the reference's loaders read: the same hfhub cache lookup as load_chatterbox
#' (R/tts.R:51-52, 72-79 -> get_model_paths()$t3_cfg, R/download.R:129-150) and load_chatterbox_turbo
#' (R/tts.R:104-106, 127 -> get_turbo_model_paths()$t3_turbo_v1, R/download.R:241-263).  options(chatterbox.t3_path = ,
#' chatterbox.t3_turbo_path = ) override it (a checkpoint outside the hub cache).
t3mi_default_path <- function (variant = c("llama", "turbo"))
{
    variant <- match.arg(variant)
    opt <- getOption(if (variant == "turbo") "chatterbox.t3_turbo_path" else "chatterbox.t3_path")
    if (!is.null(opt)) return(opt)
    path <- if (variant == "turbo") get_turbo_model_paths()$t3_turbo_v1 else get_model_paths()$t3_cfg
    if (is.null(path) || !file.exists(path))
        stop("T3 checkpoint not found: run download_chatterbox_models() (or download_chatterbox_turbo_models()), ",
             "or set options(chatterbox.t3_path = '<file>.safetensors')", call. = FALSE)
    path
}

#' Open (once per file, device and dtype) the native model.  `path = NULL` resolves the file the R model itself was
#' loaded from (t3mi_default_path); the handle is cached, so every t3_inference() call after the first reuses the
#' weights already resident in HBM.
t3mi_model <- function (path = NULL, device = 0L, dtype = c("f32", "bf16"), variant = c("llama", "turbo"),
                        cache = FALSE)
{
    dtype <- match.arg(dtype)
    variant <- match.arg(variant)
    if (is.null(path)) path <- t3mi_default_path(variant)
    if (!is.character(path) || length(path) != 1L || !file.exists(path))
        stop("t3mi_model: '", format(path), "' is not a readable safetensors file", call. = FALSE)
    key <- paste(normalizePath(path), device, dtype, variant)
    if (is.null(.t3mi[[key]])) {
        # cache = TRUE keeps the packed bf16 weights next to the checkpoint in the HF cache ("<file>.t3mi-bf16"): the next
        # session reads half the bytes and converts nothing; a stale or damaged cache is detected and rewritten
        cache_path <- if (isTRUE(cache)) paste0(path, ".t3mi-bf16") else if (is.character(cache)) cache else NULL
        .t3mi[[key]] <- .Call("t3mi_r_open", path, if (variant == "turbo") 1L else 0L, as.integer(device),
                              if (dtype == "bf16") 1L else 0L, cache_path, PACKAGE = "chatterbox")
    }
    .t3mi[[key]]
}

# t3_cond (R/t3.R:103-113) as the plain vectors the native prompt builder takes
.t3mi_cond <- function (cond)
{
    emo <- cond$emotion_adv
    if (inherits(emo, "torch_tensor")) emo <- as.numeric(emo$cpu())
    list(speaker_emb = as.numeric(cond$speaker_emb$cpu()),
         tokens = if (is.null(cond$cond_prompt_speech_tokens)) NULL
                  else as.integer(cond$cond_prompt_speech_tokens$cpu()),
         emotion_adv = if (is.null(emo)) NULL else as.numeric(emo)[1])
}

.t3mi_decode <- function (model, cond, text_tokens, cfg, native, sampler, max_cache_len = 0L)
{
    config <- model$config
    device <- model$text_emb$weight$device
    cfg$stop_token <- as.integer(config$stop_speech_token)
    cfg$start_token <- as.integer(config$start_speech_token)
    cfg$sampler <- as.integer(sampler)
    cfg$max_cache_len <- as.integer(max_cache_len)
    if (is.null(cond$cond_prompt_speech_emb)) {
        # the whole of t3_inference natively: conditioning encoder, perceiver, embeddings, prefill, decode
        # (prepare_input_embeds R/t3.R:372-403 runs on the GPU; the ids stay 0-indexed as tokenized)
        ids <- .Call("t3mi_r_generate_tokens", native, list(.t3mi_cond(cond)),
                     list(as.integer(text_tokens$cpu())), cfg, PACKAGE = "chatterbox")[[1]]
        return(list(ids = ids, device = device, stop = config$stop_speech_token))
    }
    # pre-computed prompt-speech embeddings (not produced by generate()): prompt built in R torch
    if (text_tokens$dim() == 1) text_tokens <- text_tokens$unsqueeze(1)
    text_tokens <- torch::nnf_pad(text_tokens, c(1, 0), value = config$start_text_token)
    text_tokens <- torch::nnf_pad(text_tokens, c(0, 1), value = config$stop_text_token)
    bos <- torch::torch_tensor(matrix(config$start_speech_token, nrow = 1), device = device,
                               dtype = torch::torch_long())
    if (cfg$cfg_weight > 0) {
        text_tokens <- torch::torch_cat(list(text_tokens, text_tokens), dim = 1)
        bos <- torch::torch_cat(list(bos, bos), dim = 1)
    }
    embeds <- torch::with_no_grad(model$prepare_input_embeds(cond, text_tokens, bos, cfg$cfg_weight)$embeds)
    e <- as.array(embeds$to(device = "cpu", dtype = torch::torch_float32()))   # [rows, S0, d], column-major
    ids <- .Call("t3mi_r_generate", native, e, as.integer(dim(e)), cfg, PACKAGE = "chatterbox")
    list(ids = ids, device = device, stop = config$stop_speech_token)
}

#' Same signature and return value as the reference's t3_inference (R/t3.R:461-463, 598-604).
t3_inference <- function (model, cond, text_tokens, max_new_tokens = 1000,
                          temperature = 0.8, cfg_weight = 0.5, top_p = 0.95,
                          min_p = 0.05, repetition_penalty = 1.2,
                          native = t3mi_model())
{
    cfg <- list(max_tokens = as.integer(max_new_tokens), temperature = as.double(temperature),
                cfg_weight = as.double(cfg_weight), top_p = as.double(top_p), min_p = as.double(min_p),
                rep_penalty = as.double(repetition_penalty))
    r <- .t3mi_decode(model, cond, text_tokens, cfg, native, sampler = 0L)
    if (length(r$ids) && r$ids[length(r$ids)] == r$stop) message("EOS detected at step ", length(r$ids))
    if (length(r$ids) > 0) torch::torch_tensor(r$ids, dtype = torch::torch_long(), device = r$device)
    else torch::torch_tensor(integer(0), device = r$device)
}

#' Same signature and return value as t3_inference_cpp (R/t3.R:932-935, 1039-1049).
t3_inference_cpp <- function (model, cond, text_tokens, max_new_tokens = 1000,
                              temperature = 0.8, cfg_weight = 0.5, top_p = 0.95,
                              min_p = 0.05, repetition_penalty = 1.2, max_cache_len = 350L,
                              native = t3mi_model())
{
    cfg <- list(max_tokens = as.integer(max_new_tokens), temperature = as.double(temperature),
                cfg_weight = as.double(cfg_weight), top_p = as.double(top_p), min_p = as.double(min_p),
                rep_penalty = as.double(repetition_penalty))
    r <- .t3mi_decode(model, cond, text_tokens, cfg, native, sampler = 1L, max_cache_len = max_cache_len)
    ids <- r$ids
    eos <- match(r$stop, ids)
    if (!is.na(eos)) {
        message("EOS detected at step ", eos)
        ids <- ids[seq_len(eos - 1L)]
    }
    if (length(ids) > 0L) torch::torch_tensor(ids, dtype = torch::torch_long(), device = r$device)
    else torch::torch_tensor(integer(0), device = r$device)
}

#' Same signature and return value as t3_inference_traced (R/t3.R:685-688, 863-870): the R sampling rule on a
#' cache of max_cache_len slots, at most max_cache_len - cond_len - 1 tokens (R/t3.R:753), stop token kept.
#' Nothing is traced: the decode step is a replayed HIP graph on either path.
t3_inference_traced <- function (model, cond, text_tokens, max_new_tokens = 1000,
                                 temperature = 0.8, cfg_weight = 0.5, top_p = 0.95,
                                 min_p = 0.05, repetition_penalty = 1.2, max_cache_len = 350L,
                                 native = t3mi_model())
{
    cfg <- list(max_tokens = as.integer(max_new_tokens), temperature = as.double(temperature),
                cfg_weight = as.double(cfg_weight), top_p = as.double(top_p), min_p = as.double(min_p),
                rep_penalty = as.double(repetition_penalty))
    r <- suppressWarnings(.t3mi_decode(model, cond, text_tokens, cfg, native, sampler = 0L,
                                       max_cache_len = max_cache_len))
    if (length(r$ids) && r$ids[length(r$ids)] == r$stop) message("EOS detected at step ", length(r$ids))
    if (length(r$ids) > 0L) torch::torch_tensor(r$ids, dtype = torch::torch_long(), device = r$device)
    else torch::torch_tensor(integer(0), device = r$device)
}

#' Same signature and return value as t3_inference_turbo (R/t3.R:1578-1580, 1662-1676): GPT-2 backbone, no CFG,
#' top-k / top-p logit processors; the trailing stop token is removed.
t3_inference_turbo <- function (model, cond, text_tokens, max_new_tokens = 1000,
                                temperature = 0.8, top_k = 1000L, top_p = 0.95,
                                repetition_penalty = 1.2,
                                native = t3mi_model(variant = "turbo"))
{
    config <- model$config
    device <- model$speech_emb$weight$device
    cfg <- list(max_tokens = as.integer(max_new_tokens), temperature = as.double(temperature),
                cfg_weight = 0, top_k = as.integer(top_k), top_p = as.double(top_p), min_p = 0,
                rep_penalty = as.double(repetition_penalty), sampler = 2L,
                start_token = as.integer(config$start_speech_token),
                stop_token = as.integer(config$stop_speech_token))
    # prompt (speaker row + prompt-token embeddings + text embeddings + BOS, R/t3.R:1447-1477) built on the GPU
    ids <- .Call("t3mi_r_generate_tokens", native, list(.t3mi_cond(cond)), list(as.integer(text_tokens$cpu())),
                 cfg, PACKAGE = "chatterbox")[[1]]
    n <- length(ids)
    if (n > 0L && ids[n] == config$stop_speech_token) {
        message("EOS at step ", n)
        ids <- ids[seq_len(n - 1L)]
    }
    if (length(ids) > 0L) torch::torch_tensor(ids, dtype = torch::torch_long(), device = device)
    else torch::torch_tensor(integer(0), device = device)
}

#' Many utterances at once (the batch axis tts_chunked leaves serial, R/tts.R:462-474): `conds` and
#' `text_tokens_list` are lists of the per-sentence arguments of t3_inference; returns a list of token tensors.
t3_inference_batch <- function (model, conds, text_tokens_list, max_new_tokens = 1000,
                                temperature = 0.8, cfg_weight = 0.5, top_p = 0.95,
                                min_p = 0.05, repetition_penalty = 1.2,
                                native = t3mi_model())
{
    config <- model$config
    device <- model$text_emb$weight$device
    cfg <- list(max_tokens = as.integer(max_new_tokens), temperature = as.double(temperature),
                cfg_weight = as.double(cfg_weight), top_p = as.double(top_p), min_p = as.double(min_p),
                rep_penalty = as.double(repetition_penalty), sampler = 0L,
                start_token = as.integer(config$start_speech_token),
                stop_token = as.integer(config$stop_speech_token))
    ids <- .Call("t3mi_r_generate_tokens", native, lapply(conds, .t3mi_cond),
                 lapply(text_tokens_list, function (t) as.integer(t$cpu())), cfg, PACKAGE = "chatterbox")
    lapply(ids, function (v) torch::torch_tensor(v, dtype = torch::torch_long(), device = device))
}

#' Turbo counterpart of t3_inference_batch: per-sentence semantics of t3_inference_turbo (trailing stop token removed).
t3_inference_turbo_batch <- function (model, conds, text_tokens_list, max_new_tokens = 1000,
                                      temperature = 0.8, top_k = 1000L, top_p = 0.95, repetition_penalty = 1.2,
                                      native = t3mi_model(variant = "turbo"))
{
    config <- model$config
    device <- model$speech_emb$weight$device
    cfg <- list(max_tokens = as.integer(max_new_tokens), temperature = as.double(temperature),
                cfg_weight = 0, top_k = as.integer(top_k), top_p = as.double(top_p), min_p = 0,
                rep_penalty = as.double(repetition_penalty), sampler = 2L,
                start_token = as.integer(config$start_speech_token),
                stop_token = as.integer(config$stop_speech_token))
    ids <- .Call("t3mi_r_generate_tokens", native, lapply(conds, .t3mi_cond),
                 lapply(text_tokens_list, function (t) as.integer(t$cpu())), cfg, PACKAGE = "chatterbox")
    lapply(ids, function (v) {
        n <- length(v)
        if (n > 0L && v[n] == config$stop_speech_token) v <- v[seq_len(n - 1L)]
        torch::torch_tensor(v, dtype = torch::torch_long(), device = device)
    })
}

#' tts_chunked (R/tts.R:455-480) with the sentence loop turned into ONE batched T3 call: same arguments, same
#' sentence split (strsplit(text, "(?<=[.!?])\\s+", perl = TRUE), R/tts.R:462), same per-sentence tokenisation,
#' conditioning and defaults as generate() (R/tts.R:278-293, 341-348), audio concatenated in sentence order.
#' Everything after the speech tokens (drop_invalid_tokens, S3Gen, vocoder: R/tts.R:366-412) is the reference's.
tts_chunked <- function (model, text, voice, chunk_size = 200, exaggeration = 0.5, cfg_weight = 0.5,
                         temperature = 0.8, top_p = 0.9, autocast = NULL, traced = FALSE, top_k = 1000L,
                         repetition_penalty = 1.2, ...)
{
    if (!is_loaded(model)) stop("Model not loaded. Call load_chatterbox() first.")
    sentences <- strsplit(text, "(?<=[.!?])\\s+", perl = TRUE)[[1]]
    if (length(sentences) == 0L) return(list(audio = numeric(0), sample_rate = S3GEN_SR))
    device <- model$device
    is_turbo <- isTRUE(model$turbo)
    use_autocast <- if (is.null(autocast)) grepl("^cuda", device) else autocast
    if (is.character(voice)) voice <- create_voice_embedding(model, voice, autocast = use_autocast)
    else if (!inherits(voice, "voice_embedding")) stop("voice must be a voice_embedding object or path to reference audio")
    toks <- lapply(sentences, function (s) {
        ids <- if (is_turbo) tokenize_text_gpt2(model$tokenizer, punc_norm(s)) else tokenize_text(model$tokenizer, s)
        torch::torch_tensor(ids, dtype = torch::torch_long())
    })
    cond <- t3_cond(speaker_emb = voice$ve_embedding, cond_prompt_speech_tokens = voice$cond_prompt_speech_tokens,
                    emotion_adv = if (is_turbo) NULL else exaggeration)
    conds <- rep(list(cond), length(sentences))
    message(sprintf("Generating speech tokens for %d sentences in one batch...", length(sentences)))
    speech <- if (is_turbo)
        t3_inference_turbo_batch(model$t3, conds, toks, temperature = temperature, top_k = top_k, top_p = top_p,
                                 repetition_penalty = repetition_penalty)
    else
        t3_inference_batch(model$t3, conds, toks, cfg_weight = cfg_weight, temperature = temperature, top_p = top_p)
    all_audio <- numeric(0)
    for (i in seq_along(sentences)) {
        tk <- drop_invalid_tokens(speech[[i]])
        if (length(tk) == 0L) next
        vals <- as.integer(tk)
        if (is_turbo) vals <- c(vals, rep(S3GEN_SIL, 3L))
        st <- torch::torch_tensor(vals, dtype = torch::torch_long())$unsqueeze(1L)$to(device = device)
        audio <- torch::with_no_grad(model$s3gen$inference(speech_tokens = st, ref_dict = voice$ref_dict, finalize = TRUE,
                                                           traced = traced,
                                                           n_cfm_timesteps = if (is_turbo) 2L else NULL)[[1]])
        all_audio <- c(all_audio, as.numeric(audio$squeeze()$cpu()))
    }
    list(audio = all_audio, sample_rate = S3GEN_SR)
}
