# MI355X-native S3Gen behind the reference's R API (SURVEY §8 f4): speech tokens -> mel -> waveform without R torch.
# Drop this file into the package's R/ directory next to t3_native.R.  It replaces the body of s3gen$inference
# (R/s3gen.R:987-1056) when a native handle is given; embed_ref (R/s3gen.R:931-984: reference mel, x-vector, prompt tokens --
# once per voice) stays with the reference and produces the ref_dict used here.

#' Open (once per file and device) the native flow module + CFM estimator and the HiFT vocoder from the unchanged
#' s3gen.safetensors that load_s3gen reads (R/s3gen.R:1249-1274; get_model_paths()$s3gen, R/download.R:129-150).
t3mi_s3gen <- function (path = NULL, device = 0L)
{
    if (is.null(path)) path <- getOption("chatterbox.s3gen_path", get_model_paths()$s3gen)
    if (!is.character(path) || length(path) != 1L || !file.exists(path))
        stop("t3mi_s3gen: '", format(path), "' is not a readable safetensors file", call. = FALSE)
    key <- paste("s3gen", normalizePath(path), device)
    if (is.null(.t3mi[[key]]))
        .t3mi[[key]] <- list(flow = .Call("t3mi_r_flow_open", path, as.integer(device), PACKAGE = "chatterbox"),
                             vocoder = .Call("t3mi_r_hift_open", path, as.integer(device), PACKAGE = "chatterbox"))
    .t3mi[[key]]
}

#' Same arguments and return value as s3gen$inference (R/s3gen.R:987-1056) for a ref_dict: list(output_wavs, NULL) with
#' output_wavs a (1, samples) float tensor on the reference's device.  The random draws the reference takes from torch's
#' generator (the CFM start noise, R/s3gen.R:560-565 / :1019-1022; the source's start phases and noise, R/hifigan.R:237-249)
#' are drawn here with rnorm / runif: set.seed() governs them.  meanflow (Turbo): 2 steps, no guidance (R/s3gen.R:1016-1023).
s3gen_inference_native <- function (speech_tokens, ref_dict, finalize = TRUE, n_cfm_timesteps = NULL, meanflow = FALSE,
                                    native = t3mi_s3gen())
{
    device <- ref_dict$embedding$device
    tok <- as.integer(speech_tokens$cpu())
    ptok <- as.integer(ref_dict$prompt_token$cpu())
    pfeat <- as.array(ref_dict$prompt_feat$to(device = "cpu", dtype = torch::torch_float32()))     # (1, frames, 80)
    n_feat <- dim(pfeat)[2]
    pfeat <- as.numeric(t(pfeat[1, , ]))                       # frame-major [frames][80] in memory (R is column-major)
    emb <- as.numeric(ref_dict$embedding$cpu())
    frames_all <- 2L * (length(ptok) + length(tok))
    n_steps <- if (!is.null(n_cfm_timesteps)) as.integer(n_cfm_timesteps) else if (meanflow) 2L else 10L
    noise <- rnorm(80L * frames_all)                           # mel-major: rand_noise[1, m, t] at (m - 1) * frames_all + t
    mel <- .Call("t3mi_r_flow_inference", native$flow, tok, ptok, pfeat, emb, noise,
                 c(n_steps, 0.7, as.numeric(meanflow), as.numeric(finalize), n_feat, frames_all), PACKAGE = "chatterbox")
    n_frames <- length(mel) %/% 80L
    phase <- runif(9L, -pi, pi)
    wav <- .Call("t3mi_r_hift_inference", native$vocoder, mel, n_frames, phase, rnorm(9L * 480L * n_frames), PACKAGE = "chatterbox")
    # fade-in (R/s3gen.R:922-927, 1047-1049): 20 ms of silence, 20 ms of raised cosine
    n_trim <- S3GEN_SR %/% 50L
    fade <- c(rep(0, n_trim), (cos(seq(pi, 0, length.out = n_trim)) + 1) / 2)
    k <- min(length(fade), length(wav))
    wav[seq_len(k)] <- wav[seq_len(k)] * fade[seq_len(k)]
    list(torch::torch_tensor(wav, dtype = torch::torch_float32(), device = device)$unsqueeze(1L), NULL)
}
