# MI355X-native T3 decode behind the reference's R API.
# Drop this file into the package's R/ directory (it replaces the bodies of t3_inference /
# t3_inference_cpp, R/t3.R:461-605, 932-1050; signatures and return values are unchanged) and add
#     useDynLib(chatterbox, .registration = TRUE)
# to NAMESPACE (missing in the reference, SURVEY Q18).

.t3mi <- new.env(parent = emptyenv())

#' Open (once) the native model for the safetensors file the R model was loaded from.
t3mi_model <- function (path, device = 0L, dtype = c("f32", "bf16"), variant = c("llama", "turbo"))
{
    dtype <- match.arg(dtype)
    variant <- match.arg(variant)
    key <- paste(normalizePath(path), device, dtype, variant)
    if (is.null(.t3mi[[key]])) {
        .t3mi[[key]] <- .Call("t3mi_r_open", path, if (variant == "turbo") 1L else 0L, as.integer(device),
                              if (dtype == "bf16") 1L else 0L, PACKAGE = "chatterbox")
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
                          native = t3mi_model(attr(model, "t3_path")))
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
                              native = t3mi_model(attr(model, "t3_path")))
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
                                 native = t3mi_model(attr(model, "t3_path")))
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
                                native = t3mi_model(attr(model, "t3_path"), variant = "turbo"))
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
                                native = t3mi_model(attr(model, "t3_path")))
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
