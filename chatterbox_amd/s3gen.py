"""Host-side mirror of the reference's S3Gen flow decoder interface over the t3mi_cfm_* C ABI (first piece).

`CfmEstimator` stands for `cfm_estimator` with `load_cfm_estimator_weights` applied (R/s3gen.R:389-528, 1070-1178):
`forward(x, mask, mu, t, spks, cond, r = None)` keeps the reference's argument order and (B, 80, T) layouts.
`causal_cfm_forward` is `causal_cfm$forward` (R/s3gen.R:594-636): cosine schedule + solve_euler with classifier-free
guidance, or the linear schedule + basic_euler of a meanflow model.  There is no CPU fallback: the arithmetic runs in
libt3mi.so's gfx950 kernels or not at all."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np

from . import _lib

__all__ = ["CfmEstimator", "load_cfm_estimator", "causal_cfm_forward", "S3GenFlow", "load_s3gen_flow", "HiftVocoder", "load_hift_vocoder", "s3gen_inference", "trim_fade"]

_FP = C.POINTER(C.c_float)


def _f32(a, shape=None) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.float32)
    if shape is not None and a.shape != tuple(shape):
        raise ValueError(f"expected shape {tuple(shape)}, got {a.shape}")
    return a


def _p(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(_FP)


class CfmEstimator:
    """Weights of the CFM estimator resident on one MI355X (read from the unchanged s3gen safetensors file)."""

    def __init__(self, path: str, device: int = 0, prefix: str = "flow.decoder.estimator.", n_heads: int = 0):
        self._lib = _lib.load()
        h = C.c_void_p()
        _lib.check(self._lib.t3mi_cfm_open(path.encode(), prefix.encode(), int(device), int(n_heads), C.byref(h)))
        self._h = h
        info = _lib.CfmInfo()
        _lib.check(self._lib.t3mi_cfm_get_info(self._h, C.byref(info)))
        self.info = info
        self.mel, self.meanflow = int(info.mel), bool(info.meanflow)

    def forward(self, x, mask, mu, t, spks, cond, r=None) -> np.ndarray:
        """cfm_estimator$forward (R/s3gen.R:446-520).  x, mu, cond (B, mel, T); mask (B, 1, T); t, r [B]; spks (B, mel)."""
        x = _f32(x)
        B, mel, T = x.shape
        mask, mu, cond = _f32(mask, (B, 1, T)), _f32(mu, (B, mel, T)), _f32(cond, (B, mel, T))
        t, spks = _f32(np.reshape(t, -1), (B,)), _f32(spks, (B, mel))
        rr = None if r is None else _f32(np.reshape(r, -1), (B,))
        out = np.empty((B, mel, T), dtype=np.float32)
        _lib.check(self._lib.t3mi_cfm_estimator(self._h, B, T, _p(x), _p(mask), _p(mu), _p(t), _p(spks), _p(cond), _p(rr), _p(out)))
        return out

    def close(self):
        if getattr(self, "_h", None):
            self._lib.t3mi_cfm_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def load_cfm_estimator(path: str, device: int = 0, **kw) -> CfmEstimator:
    return CfmEstimator(path, device, **kw)


def causal_cfm_forward(estimator: CfmEstimator, mu, mask, spks, cond, noise, n_timesteps: int = 10, temperature: float = 1.0,
                       inference_cfg_rate: float = 0.7, meanflow: Optional[bool] = None, timing: Optional[dict] = None) -> np.ndarray:
    """causal_cfm$forward(mu, mask, spks, cond, n_timesteps, temperature, ...) (R/s3gen.R:594-636), one utterance, untraced.
    `noise` (1, mel, >= T) plays the part of the reference's own `rand_noise` buffer (standard model: `noise[:, :, :T] *
    temperature`, :616-619) or of `torch_randn_like(mu)` with `noised_mels` already laid over it (meanflow, :610-615).
    Returns the mel (1, mel, T).  timing: a dict that receives {"device_ms": ...}."""
    mu = _f32(mu)
    _, mel, T = mu.shape
    use_mf = estimator.meanflow if meanflow is None else bool(meanflow)
    z = _f32(noise)[:, :, :T]
    if not use_mf:
        z = z * np.float32(temperature)
    z = _f32(z, (1, mel, T))
    mask, spks, cond = _f32(mask, (1, 1, T)), _f32(spks, (1, mel)), _f32(cond, (1, mel, T))
    out = np.empty((1, mel, T), dtype=np.float32)
    ms = C.c_float(0)
    _lib.check(estimator._lib.t3mi_cfm_solve(estimator._h, T, _p(z), _p(mu), _p(mask), _p(spks), _p(cond), int(n_timesteps),
                                             float(inference_cfg_rate), 1 if use_mf else 0, _p(out), C.byref(ms)))
    if timing is not None:
        timing["device_ms"] = float(ms.value)
    return out


class S3GenFlow(CfmEstimator):
    """`causal_masked_diff_xvec` with load_s3gen_weights' flow part applied (R/s3gen.R:768-891, 1215-1231): token embedding,
    speaker projection, the upsampling conformer encoder, encoder projection and the CFM decoder, resident on one MI355X.
    Also a CfmEstimator (its `forward` and causal_cfm_forward work on it)."""

    def __init__(self, path: str, device: int = 0):
        self._lib = _lib.load()
        h = C.c_void_p()
        _lib.check(self._lib.t3mi_flow_open(path.encode(), int(device), C.byref(h)))
        self._h = h
        info = _lib.CfmInfo()
        _lib.check(self._lib.t3mi_cfm_get_info(self._h, C.byref(info)))
        self.info = info
        self.mel, self.meanflow = int(info.mel), bool(info.meanflow)
        fi = _lib.FlowInfo()
        _lib.check(self._lib.t3mi_flow_get_info(self._h, C.byref(fi)))
        self.flow_info = fi

    def encoder(self, x) -> np.ndarray:
        """upsample_conformer_encoder_full$forward (R/conformer.R:517-578) for one sequence: x [T, d] -> [2T, d]."""
        x = _f32(x)
        T, d = x.shape
        if d != self.flow_info.d_model:
            raise ValueError(f"x must be [T, {self.flow_info.d_model}]")
        out = np.empty((2 * T, d), dtype=np.float32)
        _lib.check(self._lib.t3mi_flow_encoder(self._h, T, _p(x), _p(out)))
        return out

    def inference(self, token, prompt_token, prompt_feat, embedding, noise, finalize: bool = True, n_timesteps: Optional[int] = None,
                  inference_cfg_rate: float = 0.7, meanflow: Optional[bool] = None, timing: Optional[dict] = None) -> np.ndarray:
        """causal_masked_diff_xvec$forward(token, ..., prompt_token, ..., prompt_feat, ..., embedding, finalize, n_timesteps)
        (R/s3gen.R:805-890) for one utterance: 0-indexed speech tokens -> mel (1, mel, 2 * len(token)) of the generated part.
        `noise` (1, mel, >= 2 * (len(prompt_token) + len(token))): see causal_cfm_forward."""
        tok = np.ascontiguousarray(np.reshape(token, -1), dtype=np.int32)
        ptok = np.ascontiguousarray(np.reshape(prompt_token, -1), dtype=np.int32)
        pf = _f32(prompt_feat).reshape(-1, self.mel)
        emb = _f32(np.reshape(embedding, -1), (self.flow_info.spk_embed_dim,))
        z = _f32(noise)
        if z.ndim != 3 or z.shape[:2] != (1, self.mel):
            raise ValueError(f"noise must be (1, {self.mel}, frames)")
        use_mf = self.meanflow if meanflow is None else bool(meanflow)
        steps = int(n_timesteps) if n_timesteps is not None else (2 if use_mf else 10)        # R/s3gen.R:873, :1017
        cap = 2 * tok.size
        out = np.empty((1, self.mel, cap), dtype=np.float32)
        n = C.c_int(0)
        ms = C.c_float(0)
        IP = C.POINTER(C.c_int32)
        _lib.check(self._lib.t3mi_flow_inference(self._h, tok.ctypes.data_as(IP), tok.size, ptok.ctypes.data_as(IP), ptok.size, _p(pf),
                                                 pf.shape[0], _p(emb), _p(z), z.shape[2], steps, float(inference_cfg_rate),
                                                 1 if use_mf else 0, 1 if finalize else 0, _p(out), cap, C.byref(n), C.byref(ms)))
        if timing is not None:
            timing["device_ms"] = float(ms.value)
        # the C side packs rows of n frames
        return out.reshape(-1)[: self.mel * n.value].reshape(1, self.mel, n.value).copy()


def load_s3gen_flow(path: str, device: int = 0) -> S3GenFlow:
    return S3GenFlow(path, device)


class HiftVocoder:
    """`hift_generator` as create_s3gen_vocoder builds it, with load_hifigan_weights applied (R/hifigan.R:337-817), resident
    on one MI355X."""

    def __init__(self, path: str, device: int = 0, prefix: str = "mel2wav."):
        self._lib = _lib.load()
        h = C.c_void_p()
        _lib.check(self._lib.t3mi_hift_open(path.encode(), prefix.encode(), int(device), C.byref(h)))
        self._h = h
        info = _lib.HiftInfo()
        _lib.check(self._lib.t3mi_hift_get_info(self._h, C.byref(info)))
        self.info = info
        self.mel, self.n_harmonics, self.samples_per_frame = int(info.mel), int(info.n_harmonics), int(info.samples_per_frame)

    def inference(self, speech_feat, phase=None, noise=None, timing: Optional[dict] = None, cache_source=None) -> dict:
        """hift_generator$inference(speech_feat) (R/hifigan.R:601-626): speech_feat (1, mel, T) or (mel, T) -> {"audio": [480 T],
        "source": [480 T], "f0": [T]}.  phase [n_harmonics] / noise [n_harmonics, 480 T]: the draws the reference takes from
        torch's generator (R/hifigan.R:237-249); None = zero phases / no noise term.  cache_source [n <= 480 T]: the previous
        chunk's source tail, written over the start of this call's source (R/hifigan.R:617-620)."""
        m = _f32(speech_feat)
        if m.ndim == 3:
            m = m[0]
        mel, T = m.shape
        if mel != self.mel:
            raise ValueError(f"speech_feat must have {self.mel} mel bins")
        m = np.ascontiguousarray(m)
        n = T * self.samples_per_frame
        ph = None if phase is None else _f32(np.reshape(phase, -1), (self.n_harmonics,))
        nz = None if noise is None else _f32(noise, (self.n_harmonics, n))
        audio, src, f0 = np.empty(n, np.float32), np.empty(n, np.float32), np.empty(T, np.float32)
        ms = C.c_float(0)
        cs = None if cache_source is None else _f32(np.reshape(cache_source, -1))
        _lib.check(self._lib.t3mi_hift_inference(self._h, T, _p(m), _p(ph), _p(nz), _p(cs), 0 if cs is None else int(cs.size), _p(audio),
                                                 _p(src), _p(f0), C.byref(ms)))
        if timing is not None:
            timing["device_ms"] = float(ms.value)
        return {"audio": audio, "source": src, "f0": f0}

    def close(self):
        if getattr(self, "_h", None):
            self._lib.t3mi_hift_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def load_hift_vocoder(path: str, device: int = 0, **kw) -> HiftVocoder:
    return HiftVocoder(path, device, **kw)


S3GEN_SR = 24000      # R/s3gen.R:5


def trim_fade(sr: int = S3GEN_SR) -> np.ndarray:
    """s3gen's fade-in buffer (R/s3gen.R:922-927): 20 ms of zeros, then 20 ms of a raised-cosine ramp."""
    n_trim = sr // 50
    fade = np.zeros(2 * n_trim, dtype=np.float32)
    fade[n_trim:] = ((np.cos(np.linspace(np.pi, 0.0, n_trim, dtype=np.float32)) + np.float32(1)) / np.float32(2)).astype(np.float32)
    return fade


def s3gen_inference(flow: S3GenFlow, vocoder: HiftVocoder, speech_tokens, ref_dict: dict, noise, phase=None, source_noise=None,
                    finalize: bool = True, n_cfm_timesteps: Optional[int] = None, timing: Optional[dict] = None):
    """s3gen$inference(speech_tokens, ref_dict = ..., finalize, n_cfm_timesteps) (R/s3gen.R:987-1056): speech tokens -> mel (flow)
    -> waveform (HiFT) -> fade-in.  ref_dict carries what embed_ref returns (R/s3gen.R:977-983): prompt_token, prompt_feat
    [frames, 80], embedding [192].  `noise` feeds the CFM solve, `phase` / `source_noise` the vocoder's source (the draws the
    reference takes from torch's generator).  Returns (audio [T_wav], mel (1, 80, T))."""
    t_flow, t_voc = {}, {}
    mel = flow.inference(speech_tokens, ref_dict["prompt_token"], ref_dict["prompt_feat"], ref_dict["embedding"], noise,
                         finalize=finalize, n_timesteps=n_cfm_timesteps, timing=t_flow)
    out = vocoder.inference(mel, phase, source_noise, timing=t_voc)
    audio = out["audio"]
    fade = trim_fade()
    n = min(fade.size, audio.size)
    audio[:n] = audio[:n] * fade[:n]                               # R/s3gen.R:1047-1049
    if timing is not None:
        timing.update(flow_ms=t_flow.get("device_ms"), vocoder_ms=t_voc.get("device_ms"))
    return audio, mel


S3GEN_SIL = 4299      # silence token appended three times on Turbo (R/s3gen.R:6, R/tts.R:374-378)


def synthesize_one(flow, voc, tokens, index: int, ref_dict: dict, seed: int, turbo: bool = False,
                   n_cfm_timesteps: Optional[int] = None, timing: Optional[dict] = None) -> np.ndarray:
    """One utterance of synthesize(): its draws (CFM start noise, harmonic start phases, source noise) come from the numpy generator
    keyed by (seed, index), so the audio is the same whichever handle, thread or call produces it."""
    t = np.asarray(tokens, dtype=np.int64).reshape(-1)
    if t.size == 0:
        return np.zeros(0, np.float32)
    if turbo:
        t = np.concatenate([t, np.full(3, S3GEN_SIL, t.dtype)])
    n_prompt = int(np.reshape(ref_dict["prompt_token"], -1).size)
    rng = np.random.Generator(np.random.PCG64([int(seed), int(index)]))
    z = rng.standard_normal((1, flow.mel, 2 * (n_prompt + t.size)), dtype=np.float32)
    phase = rng.uniform(-np.pi, np.pi, voc.n_harmonics).astype(np.float32)
    snoise = rng.standard_normal((voc.n_harmonics, 2 * t.size * voc.samples_per_frame), dtype=np.float32)
    audio, _ = s3gen_inference(flow, voc, t, ref_dict, z, phase, snoise, finalize=True, n_cfm_timesteps=n_cfm_timesteps, timing=timing)
    return audio


def synthesize(handles, token_lists: Sequence, ref_dict: dict, seed: Optional[int] = None, turbo: bool = False,
               n_cfm_timesteps: Optional[int] = None, timing: Optional[dict] = None) -> list:
    """The waveform half of generate() (R/tts.R:365-412) for several utterances of ONE voice: every token vector (already through
    drop_invalid_tokens) -> s3gen$inference(finalize = TRUE) -> audio samples; an empty vector gives an empty array (the
    reference warns "No valid speech tokens generated" and returns numeric(0)).  On Turbo three silence tokens are appended and
    the meanflow solve takes 2 steps (R/tts.R:374-389).

    `handles`: one (flow, vocoder) pair, or a list of pairs -- the utterances are then dealt over the pairs, one host thread per
    pair (independent launch chains fill each other's gaps: DESIGN.md 3.5; four pairs per GPU measured best; pairs may sit
    on different GPUs).  The draws the reference takes from torch's generator (CFM start noise, harmonic start phases, source
    noise) come from numpy generators keyed by (seed, utterance index): the audio of an utterance does not depend on the number
    of handles.  seed = None draws one from numpy's global RNG, like t3.py's wrappers."""
    import threading

    pairs = [handles] if isinstance(handles[0], S3GenFlow) else list(handles)
    if seed is None:
        seed = int(np.random.randint(0, 2**31 - 1))
    toks = [np.asarray(t, dtype=np.int64).reshape(-1) for t in token_lists]
    out: list = [None] * len(toks)
    times: list = [None] * len(toks)
    steps = n_cfm_timesteps if n_cfm_timesteps is not None else (2 if turbo else None)
    errors: list = []

    def work(w: int):
        flow, voc = pairs[w]
        try:
            for u in range(w, len(toks), len(pairs)):
                tm: dict = {}
                out[u] = synthesize_one(flow, voc, toks[u], u, ref_dict, seed, turbo=turbo, n_cfm_timesteps=steps, timing=tm)
                times[u] = tm
        except Exception as e:  # noqa: BLE001 -- re-raised in the caller's thread
            errors.append(e)

    if len(pairs) == 1:
        work(0)
    else:
        ts = [threading.Thread(target=work, args=(w,)) for w in range(len(pairs))]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
    if errors:
        raise errors[0]
    if timing is not None:
        timing["per_utterance"] = times
    return out
