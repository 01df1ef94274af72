"""CPU restatement (numpy, fp32) of chatterbox's T3 speech-token decoder.

TEST INFRASTRUCTURE ONLY.  Nothing under ``chatterbox_amd/`` imports this
module; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may.  It is the checker, never the thing measured/shipped.

What it restates (all citations are into /root/reference, read as text):

* RoPE tables with Llama-3 band scaling ........ R/llama.R:87-152
* rotate_half / apply_rotary_pos_emb ........... R/llama.R:158-188
* RMSNorm (fp32, weight applied last) .......... R/llama.R:65-72, src/t3_decode.cpp:71-76
* attention (q/k/v/o, KV append, SDPA) ......... R/llama.R:220-278, src/t3_decode.cpp:127-166
* SwiGLU MLP .................................... R/llama.R:305-308, src/t3_decode.cpp:169-179
* decoder layer / model forward ................. R/llama.R:333-367, 422-546
* t3_inference loop + sampling tail ............. R/t3.R:461-605
* cpp_t3_decode loop (differs in Q3/Q4) ......... src/t3_decode.cpp:301-417
* Turbo (GPT-2) blocks + sampler ................ R/t3.R:1190-1410, 1578-1742
* prompt construction (cond encoder, perceiver) . R/t3.R:48-90, 140-316, 361-403, 1447-1477

Parity pin: the arithmetic of the reference lives in libtorch (R package
`torch` >= 0.12, unpinned; DESCRIPTION:19).  This restatement is pinned by
(a) the reference's own fixtures outputs/t3_debug.safetensors (KAT-1, KAT-2,
KAT-4, KAT-5) and the RoPE band table (KAT-3) -- tests/test_oracle_kat.py;
(b) an op-for-op run of the same sequence on CPU libtorch via python `torch`
in the build container -- tests/test_oracle_vs_torch.py.  Full-model parity
against real weights is UNPINNED (weights are not in the container).

Index convention: everything here is 0-indexed.  The R code's 1-indexed
quirks are restated explicitly (Q1/Q3 in SURVEY §3.6).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

F32 = np.float32

# --------------------------------------------------------------------------
# Configurations (R/llama.R:20-45, R/t3.R:11-36, R/t3.R:1154-1179, 1347-1357)
# --------------------------------------------------------------------------


@dataclass
class LlamaConfig:
    hidden_size: int = 1024
    intermediate_size: int = 4096
    num_hidden_layers: int = 30
    num_attention_heads: int = 16
    head_dim: int = 64
    rms_norm_eps: float = 1e-5
    rope_theta: float = 500000.0
    rope_factor: float = 8.0
    rope_high_freq_factor: float = 4.0
    rope_low_freq_factor: float = 1.0
    rope_original_max_pos: int = 8192
    speech_vocab: int = 8194          # speech_tokens_dict_size
    max_speech_pos: int = 4100        # max_speech_tokens + 4  (R/t3.R:349)
    start_speech_token: int = 6561
    stop_speech_token: int = 6562


@dataclass
class SampleParams:
    """Arguments of t3_inference (R/t3.R:461-463)."""
    max_new_tokens: int = 1000
    temperature: float = 0.8
    cfg_weight: float = 0.5
    top_p: float = 0.95
    min_p: float = 0.05
    repetition_penalty: float = 1.2
    # 'r'   : R/t3.R:529-563  (Q3 BOS quirk, drop the crossing token)
    # 'cpp' : src/t3_decode.cpp:318-357 (empty seen set, keep crossing token)
    variant: str = "r"
    seed: int = 0
    max_cache_len: int = 0            # 0 = no cap (R path); 350 on the cpp path


# --------------------------------------------------------------------------
# RoPE  (R/llama.R:87-152)
# --------------------------------------------------------------------------

def llama3_inv_freq(cfg: LlamaConfig) -> np.ndarray:
    """inv_freq after Llama-3 band scaling; float32 [head_dim/2].

    R/llama.R:92 builds 1/(theta^(arange(0,dim,2)/dim)) in float32; :120-152
    classifies each wavelength (as an R double) into high / medium / low bands.
    """
    dim = cfg.head_dim
    expo = (np.arange(0, dim, 2, dtype=F32) / F32(dim)).astype(F32)
    inv = (F32(1.0) / np.power(F32(cfg.rope_theta), expo, dtype=F32)).astype(F32)
    low_wl = cfg.rope_original_max_pos / cfg.rope_low_freq_factor
    high_wl = cfg.rope_original_max_pos / cfg.rope_high_freq_factor
    out = np.zeros_like(inv)
    wavelen = (F32(2.0 * math.pi) / inv).astype(F32)       # R/llama.R:130 (float32 tensor)
    for i in range(inv.shape[0]):
        wl = float(wavelen[i])                              # as.numeric() -> double
        if wl < high_wl:
            out[i] = inv[i]
        elif wl > low_wl:
            out[i] = inv[i] / F32(cfg.rope_factor)
        else:
            smooth = (cfg.rope_original_max_pos / wl - cfg.rope_low_freq_factor) / (
                cfg.rope_high_freq_factor - cfg.rope_low_freq_factor)
            a = F32(F32(1.0 - smooth) * inv[i]) / F32(cfg.rope_factor)
            b = F32(F32(smooth) * inv[i])
            out[i] = F32(a + b)
    return out.astype(F32)


def rope_band_table(cfg: LlamaConfig) -> str:
    """'H' unscaled / 'M' interpolated / 'L' divided -- KAT-3 helper."""
    dim = cfg.head_dim
    expo = (np.arange(0, dim, 2, dtype=F32) / F32(dim)).astype(F32)
    inv = (F32(1.0) / np.power(F32(cfg.rope_theta), expo, dtype=F32)).astype(F32)
    low_wl = cfg.rope_original_max_pos / cfg.rope_low_freq_factor
    high_wl = cfg.rope_original_max_pos / cfg.rope_high_freq_factor
    s = ""
    for v in inv:
        wl = float(F32(2.0 * math.pi) / v)
        s += "H" if wl < high_wl else ("L" if wl > low_wl else "M")
    return s


def rope_tables(cfg: LlamaConfig, max_len: int) -> Tuple[np.ndarray, np.ndarray]:
    """cos/sin [max_len, head_dim] float32 (R/llama.R:100-111: emb = cat(freqs, freqs))."""
    inv = llama3_inv_freq(cfg)
    t = np.arange(max_len, dtype=F32)
    freqs = np.outer(t, inv).astype(F32)
    emb = np.concatenate([freqs, freqs], axis=-1)
    return np.cos(emb).astype(F32), np.sin(emb).astype(F32)


def rotate_half(x: np.ndarray) -> np.ndarray:
    """R/llama.R:158-163."""
    h = x.shape[-1] // 2
    return np.concatenate([-x[..., h:], x[..., :h]], axis=-1)


def apply_rope(q: np.ndarray, k: np.ndarray, cos: np.ndarray, sin: np.ndarray):
    """q,k [R,H,S,Dh]; cos,sin [S,Dh] already gathered (R/llama.R:173-188)."""
    c = cos[None, None, :, :]
    s = sin[None, None, :, :]
    qe = (q * c).astype(F32) + (rotate_half(q) * s).astype(F32)
    ke = (k * c).astype(F32) + (rotate_half(k) * s).astype(F32)
    return qe.astype(F32), ke.astype(F32)


# --------------------------------------------------------------------------
# Layers
# --------------------------------------------------------------------------

def rms_norm(x: np.ndarray, w: np.ndarray, eps: float) -> np.ndarray:
    """R/llama.R:65-72 -- w * (x * rsqrt(mean(x^2) + eps)), fp32."""
    x = x.astype(F32)
    var = np.mean(np.square(x), axis=-1, keepdims=True, dtype=F32)
    rs = (F32(1.0) / np.sqrt(var + F32(eps), dtype=F32)).astype(F32)
    return (w * (x * rs).astype(F32)).astype(F32)


def linear(x: np.ndarray, w: np.ndarray, b: Optional[np.ndarray] = None) -> np.ndarray:
    """torch linear: x @ w.T (+ b); w is [out, in] row-major."""
    y = np.matmul(x.astype(F32), w.T).astype(F32)
    if b is not None:
        y = (y + b).astype(F32)
    return y


def sdpa(q: np.ndarray, k: np.ndarray, v: np.ndarray, mask: Optional[np.ndarray]) -> np.ndarray:
    """softmax(q k^T / sqrt(Dh) + mask) v ; q [R,H,Sq,Dh], k/v [R,H,L,Dh].

    R/llama.R:257-264 (mask = additive -65504 upper triangle in prefill, none in decode).
    """
    scale = F32(1.0 / math.sqrt(q.shape[-1]))
    s = (np.matmul(q, np.swapaxes(k, -1, -2)) * scale).astype(F32)
    if mask is not None:
        s = (s + mask).astype(F32)
    m = s.max(axis=-1, keepdims=True)
    e = np.exp(s - m, dtype=F32)
    p = (e / e.sum(axis=-1, keepdims=True, dtype=F32)).astype(F32)
    return np.matmul(p, v).astype(F32)


def silu(x: np.ndarray) -> np.ndarray:
    """x * sigmoid(x) (R/llama.R:302)."""
    return (x * (F32(1.0) / (F32(1.0) + np.exp(-x, dtype=F32)))).astype(F32)


def causal_mask(s: int) -> np.ndarray:
    """Additive -65504 strictly-upper-triangular mask (R/llama.R:467-469, Q12)."""
    m = np.full((s, s), -65504.0, dtype=F32)
    return np.triu(m, k=1)[None, None, :, :]


class LlamaOracle:
    """Llama-520M-shaped backbone + T3 speech head/embeddings over a dict of
    fp32 arrays keyed by the reference's safetensors names (R/t3.R:1065-1097,
    R/llama.R:566-607)."""

    def __init__(self, weights: Dict[str, np.ndarray], cfg: LlamaConfig):
        self.w = weights
        self.cfg = cfg
        self._rope_len = 0
        self._cos = self._sin = None

    def rope(self, need: int):
        if need > self._rope_len:
            n = max(need, 4096)                     # R/llama.R:413
            self._cos, self._sin = rope_tables(self.cfg, n)
            self._rope_len = n
        return self._cos, self._sin

    def layer(self, i: int, x: np.ndarray, pos0: int, mask, past):
        """llama_decoder_layer$forward (R/llama.R:333-367). x [R,S,d]."""
        cfg, w = self.cfg, self.w
        p = f"tfmr.layers.{i}."
        R, S, d = x.shape
        H, Dh = cfg.num_attention_heads, cfg.head_dim
        h = rms_norm(x, w[p + "input_layernorm.weight"], cfg.rms_norm_eps)
        q = linear(h, w[p + "self_attn.q_proj.weight"]).reshape(R, S, H, Dh).transpose(0, 2, 1, 3)
        k = linear(h, w[p + "self_attn.k_proj.weight"]).reshape(R, S, H, Dh).transpose(0, 2, 1, 3)
        v = linear(h, w[p + "self_attn.v_proj.weight"]).reshape(R, S, H, Dh).transpose(0, 2, 1, 3)
        cos, sin = self.rope(pos0 + S)
        q, k = apply_rope(q, k, cos[pos0:pos0 + S], sin[pos0:pos0 + S])
        if past is not None:
            k = np.concatenate([past[0], k], axis=2)
            v = np.concatenate([past[1], v], axis=2)
        a = sdpa(q, k, v, mask)
        a = a.transpose(0, 2, 1, 3).reshape(R, S, d)
        x = (x + linear(a, w[p + "self_attn.o_proj.weight"])).astype(F32)
        h = rms_norm(x, w[p + "post_attention_layernorm.weight"], cfg.rms_norm_eps)
        g = linear(h, w[p + "mlp.gate_proj.weight"])
        u = linear(h, w[p + "mlp.up_proj.weight"])
        x = (x + linear((silu(g) * u).astype(F32), w[p + "mlp.down_proj.weight"])).astype(F32)
        return x, (k, v)

    def forward(self, embeds: np.ndarray, past: Optional[list] = None):
        """llama_model$forward (R/llama.R:422-546): returns (last_hidden_state
        already final-normed [R,S,d], per-layer [(k,v)])."""
        cfg = self.cfg
        x = embeds.astype(F32)
        S = x.shape[1]
        pos0 = 0 if past is None else past[0][0].shape[2]
        mask = None
        if S > 1:
            mask = causal_mask(S)
            if pos0 > 0:
                mask = np.concatenate([np.zeros((1, 1, S, pos0), F32), mask], axis=-1)
        new_past = []
        for i in range(cfg.num_hidden_layers):
            x, kv = self.layer(i, x, pos0, mask, None if past is None else past[i])
            new_past.append(kv)
        x = rms_norm(x, self.w["tfmr.norm.weight"], cfg.rms_norm_eps)
        return x, new_past

    def speech_logits(self, hidden_last: np.ndarray) -> np.ndarray:
        """speech_head, no bias (R/t3.R:355, 515). hidden_last [R,d] -> [R,V]."""
        return linear(hidden_last, self.w["speech_head.weight"])

    def next_embed(self, token0: int, step: int) -> np.ndarray:
        """speech_emb[tok] + speech_pos_emb[step] (R/t3.R:582; Q11)."""
        return (self.w["speech_emb.weight"][token0] + self.w["speech_pos_emb.emb.weight"][step]).astype(F32)


# --------------------------------------------------------------------------
# Sampling tail (R/t3.R:518-567 ; src/t3_decode.cpp:307-366)
# --------------------------------------------------------------------------

def cfg_combine(logits: np.ndarray, cfg_weight: float) -> np.ndarray:
    """cond + w*(cond-uncond) (R/t3.R:518-524). logits [2,V] -> [V]."""
    if cfg_weight > 0.0 and logits.shape[0] > 1:
        c, u = logits[0].astype(F32), logits[1].astype(F32)
        return (c + (F32(cfg_weight) * (c - u).astype(F32)).astype(F32)).astype(F32)
    return logits[0].astype(F32)


def rep_penalty(logits: np.ndarray, seen0: Sequence[int], penalty: float) -> np.ndarray:
    """Divide regardless of sign (R/t3.R:529-532, Q2). seen0: 0-indexed column ids."""
    out = logits.astype(F32).copy()
    if penalty != 1.0 and len(seen0) > 0:
        ids = np.unique(np.asarray(list(seen0), dtype=np.int64))
        ids = ids[(ids >= 0) & (ids < out.shape[0])]
        out[ids] = (out[ids] / F32(penalty)).astype(F32)
    return out


def softmax1d(x: np.ndarray) -> np.ndarray:
    m = x.max()
    e = np.exp((x - m).astype(F32), dtype=F32)
    return (e / e.sum(dtype=F32)).astype(F32)


def counter_uniform(seed: int, utt: int, step: int) -> float:
    """The build's own RNG (libtorch's multinomial stream cannot be matched,
    SURVEY §7.3): splitmix64 of (seed, utterance, step) -> 24-bit uniform in [0,1).
    The HIP sampler implements the identical function (csrc/sampler.hip)."""
    M = (1 << 64) - 1
    z = (seed * 0x9E3779B97F4A7C15 + utt * 0xBF58476D1CE4E5B9 + step * 0x94D049BB133111EB + 0x2545F4914F6CDD1D) & M
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
    z = z ^ (z >> 31)
    return float(z >> 40) * (1.0 / 16777216.0)


@dataclass
class TailResult:
    token: int                       # 0-indexed sampled token
    sorted_idx: np.ndarray           # descending-prob order, 0-indexed ids (support only)
    sorted_probs: np.ndarray         # final renormalised probs of the support (float32)
    logits_after_temp: np.ndarray    # [V]
    cum: Optional[np.ndarray] = None  # float32 cumsum over the kept prefix
    target: float = 0.0              # u * cum[-1] (float32) of the draw


def sampling_tail(logits_cfg: np.ndarray, seen0: Sequence[int], p: SampleParams,
                  u: float) -> TailResult:
    """penalty -> /T -> softmax -> min-p (prob space) -> renorm -> sort desc ->
    cumsum -> top-p mask -> renorm -> draw.   R/t3.R:529-567 ('r') and
    src/t3_decode.cpp:318-362 ('cpp').  cumsum accumulates in double and rounds
    each prefix to float32, as libtorch's CPU cumsum does for float tensors."""
    lg = rep_penalty(logits_cfg, seen0, p.repetition_penalty)
    if p.temperature != 1.0:
        lg = (lg / F32(p.temperature)).astype(F32)
    probs = softmax1d(lg)
    thr = F32(F32(p.min_p) * probs.max())
    if p.variant == "r":
        probs = np.where(probs < thr, F32(0), probs)           # R/t3.R:545
    else:
        probs = np.where(probs >= thr, probs, F32(0))          # src/t3_decode.cpp:341
    probs = (probs / probs.sum(dtype=F32)).astype(F32)
    order = np.argsort(-probs, kind="stable")                   # ties: lowest id first
    sp = probs[order]
    cum = np.cumsum(sp.astype(np.float64)).astype(F32)
    mask = cum > F32(p.top_p)
    if p.variant == "r":
        mask[0] = False                                         # R/t3.R:559 (drops crossing token, Q4)
    else:
        mask = np.concatenate([[False], mask[:-1]])            # src/t3_decode.cpp:352-355
    kept = np.where(mask, F32(0), sp).astype(F32)
    nz = int(np.count_nonzero(kept))
    # the kept set is a prefix of the sorted order in both variants
    sup_idx = order[:nz].copy()
    sup_p = (kept[:nz] / kept.sum(dtype=F32)).astype(F32)       # R/t3.R:563
    # Draw.  libtorch's multinomial stream cannot be reproduced, so the build
    # defines its own: inverse CDF over the kept prefix of `cum` (the float32 cumsum above)
    # at target = u * cum[nz-1]; first index whose cumulative mass exceeds the target.
    target = F32(F32(u) * cum[nz - 1])
    j = int(np.count_nonzero(cum[:nz] <= target))
    j = min(j, nz - 1)
    return TailResult(int(sup_idx[j]), sup_idx, sup_p, lg, cum[:nz].copy(), float(target))


# --------------------------------------------------------------------------
# t3_inference  (R/t3.R:461-605)  /  cpp_t3_decode (src/t3_decode.cpp:301-417)
# --------------------------------------------------------------------------

@dataclass
class InferenceTrace:
    tokens: List[int] = field(default_factory=list)          # 0-indexed, includes EOS if hit (Q14)
    raw_logits: List[np.ndarray] = field(default_factory=list)   # per step [R,V]
    tails: List[TailResult] = field(default_factory=list)
    hit_eos: bool = False
    cache_full: bool = False


def t3_inference(model: LlamaOracle, embeds: np.ndarray, p: SampleParams, utt: int = 0,
                 keep_logits: bool = True, forced_tokens: Optional[Sequence[int]] = None,
                 mask_stop: bool = False) -> InferenceTrace:
    """The decode loop.  `embeds` is the output of prepare_input_embeds
    ([R,S0,d]; R = 2 with CFG, row 1 = uncond, R/t3.R:372-403).

    forced_tokens: teacher-force these ids instead of sampling (parity hook).
    mask_stop: set the stop-token logit to -inf before the tail (bench workloads
    that must run a fixed number of steps, SURVEY §8d cfg 3)."""
    cfg = model.cfg
    tr = InferenceTrace()
    hidden, past = model.forward(embeds)                       # prefill R/t3.R:505
    cond_len = embeds.shape[1]
    R = embeds.shape[0]
    if p.variant == "r":
        # generated_ids seeded with raw BOS id used as a 1-based column => column 6560 (Q3)
        seen: List[int] = [cfg.start_speech_token - 1]
        max_tokens = p.max_new_tokens
    else:
        seen = []
        cap = p.max_cache_len if p.max_cache_len > 0 else 350
        max_tokens = min(p.max_new_tokens, cap - cond_len - 1)  # R/t3.R:996
    last = hidden[:, -1, :]
    for step in range(max_tokens):
        logits = model.speech_logits(last)                     # [R,V]
        if keep_logits:
            tr.raw_logits.append(logits.copy())
        lc = cfg_combine(logits, p.cfg_weight)
        if mask_stop:
            lc = lc.copy()
            lc[cfg.stop_speech_token] = -np.inf
        tail = sampling_tail(lc, seen, p, counter_uniform(p.seed, utt, step))
        tok = tail.token if forced_tokens is None else int(forced_tokens[step])
        if keep_logits:
            tr.tails.append(tail)
        tr.tokens.append(tok)
        seen.append(tok)
        if tok == cfg.stop_speech_token:
            tr.hit_eos = True
            break
        if step + 1 >= max_tokens:
            break                                              # no forward needed after the last sample
        if p.variant != "r":
            cap = p.max_cache_len if p.max_cache_len > 0 else 350
            if cond_len + step + 1 > cap:                      # src/t3_decode.cpp:391-397
                tr.cache_full = True
                break
        e = model.next_embed(tok, step + 1)                    # R/t3.R:582 (pos row i; BOS row 0)
        nxt = np.broadcast_to(e[None, None, :], (R, 1, e.shape[0])).astype(F32)
        hidden, past = model.forward(nxt, past)
        last = hidden[:, -1, :]
    return tr


SPEECH_VOCAB_SIZE = 6561    # R/s3tokenizer.R:7


def drop_invalid_tokens(tokens: Sequence[int], speech_vocab_size: int = SPEECH_VOCAB_SIZE) -> np.ndarray:
    """R/s3tokenizer.R:678-687: `tokens[tokens < SPEECH_VOCAB_SIZE]` -- the ids of the start / stop / padding range are
    removed, order kept.  generate() applies it to the T3 output before S3Gen (R/tts.R:366)."""
    t = np.asarray(list(tokens), dtype=np.int64)
    return t[t < speech_vocab_size]


# --------------------------------------------------------------------------
# Turbo T3: GPT-2 medium backbone (R/t3.R:1190-1410) + sampler (R/t3.R:1688-1742)
# --------------------------------------------------------------------------

@dataclass
class TurboConfig:
    hidden_size: int = 1024
    n_head: int = 16
    n_layer: int = 24
    layer_norm_epsilon: float = 1e-5
    speech_vocab: int = 6563
    start_speech_token: int = 6561
    stop_speech_token: int = 6562


def layer_norm(x: np.ndarray, w: np.ndarray, b: np.ndarray, eps: float) -> np.ndarray:
    """nnf_layer_norm (R/t3.R:1200-1203): biased variance, fp32."""
    x = x.astype(F32)
    mu = x.mean(axis=-1, keepdims=True, dtype=F32)
    xc = (x - mu).astype(F32)
    var = np.mean(np.square(xc), axis=-1, keepdims=True, dtype=F32)
    return ((xc / np.sqrt(var + F32(eps), dtype=F32)).astype(F32) * w + b).astype(F32)


def gelu_tanh(x: np.ndarray) -> np.ndarray:
    """nnf_gelu(approximate = "tanh") (R/t3.R:1304)."""
    x = x.astype(F32)
    c = F32(math.sqrt(2.0 / math.pi))
    inner = (c * (x + F32(0.044715) * x * x * x)).astype(F32)
    return (F32(0.5) * x * (F32(1.0) + np.tanh(inner, dtype=F32))).astype(F32)


class GPT2Oracle:
    """gpt2_model + t3_model_turbo heads over the reference's turbo key names (R/t3.R:1508-1549).
    No positional term of any kind is added (wpe is loaded but unused, R/t3.R:1387-1409)."""

    def __init__(self, weights: Dict[str, np.ndarray], cfg: TurboConfig):
        self.w = weights
        self.cfg = cfg

    def forward(self, embeds: np.ndarray, past: Optional[list] = None):
        cfg, w = self.cfg, self.w
        x = embeds.astype(F32)
        R, S, d = x.shape
        H = cfg.n_head
        Dh = d // H
        new_past = []
        for i in range(cfg.n_layer):
            p = f"tfmr.h.{i}."
            h = layer_norm(x, w[p + "ln_1.weight"], w[p + "ln_1.bias"], cfg.layer_norm_epsilon)
            qkv = linear(h, w[p + "attn.c_attn.weight"], w[p + "attn.c_attn.bias"])
            q, k, v = [t.reshape(R, S, H, Dh).transpose(0, 2, 1, 3) for t in np.split(qkv, 3, axis=-1)]
            if past is not None:
                k = np.concatenate([past[i][0], k], axis=2)
                v = np.concatenate([past[i][1], v], axis=2)
            new_past.append((k, v))
            mask = None
            if S > 1 and past is None:                      # is_causal = TRUE in prefill (R/t3.R:1257)
                mask = np.triu(np.full((S, S), -np.inf, dtype=F32), k=1)[None, None]
            a = sdpa(q, k, v, mask).transpose(0, 2, 1, 3).reshape(R, S, d)
            x = (x + linear(a, w[p + "attn.c_proj.weight"], w[p + "attn.c_proj.bias"])).astype(F32)
            h = layer_norm(x, w[p + "ln_2.weight"], w[p + "ln_2.bias"], cfg.layer_norm_epsilon)
            h = gelu_tanh(linear(h, w[p + "mlp.c_fc.weight"], w[p + "mlp.c_fc.bias"]))
            x = (x + linear(h, w[p + "mlp.c_proj.weight"], w[p + "mlp.c_proj.bias"])).astype(F32)
        x = layer_norm(x, w["tfmr.ln_f.weight"], w["tfmr.ln_f.bias"], cfg.layer_norm_epsilon)
        return x, new_past

    def speech_logits(self, hidden_last: np.ndarray) -> np.ndarray:
        return linear(hidden_last, self.w["speech_head.weight"], self.w["speech_head.bias"])   # R/t3.R:1444


@dataclass
class TurboParams:
    """Arguments of t3_inference_turbo (R/t3.R:1578-1580)."""
    max_new_tokens: int = 1000
    temperature: float = 0.8
    top_k: int = 1000
    top_p: float = 0.95
    repetition_penalty: float = 1.2
    seed: int = 0


def turbo_sampling_tail(logits: np.ndarray, pen_ids0: Sequence[int], p: TurboParams, u: float) -> TailResult:
    """.turbo_sample_token (R/t3.R:1688-1742), 0-indexed: sign-aware penalty, /T, top-k by the
    k-th largest value, top-p on the softmax of the sorted logits (first always kept), softmax over
    the survivors.  The draw is the build's inverse-CDF rule (see sampling_tail)."""
    lg = logits.astype(F32).copy()
    V = lg.shape[0]
    if p.repetition_penalty != 1.0 and len(pen_ids0) > 0:
        ids = np.unique(np.asarray(list(pen_ids0), dtype=np.int64))
        ids = ids[(ids >= 0) & (ids < V)]
        pos = lg[ids] > 0
        lg[ids] = np.where(pos, lg[ids] / F32(p.repetition_penalty), lg[ids] * F32(p.repetition_penalty)).astype(F32)
    if p.temperature > 0 and p.temperature != 1.0:
        lg = (lg / F32(p.temperature)).astype(F32)
    if 0 < p.top_k < V:
        kth = np.sort(lg)[::-1][p.top_k - 1]
        lg = np.where(lg < kth, F32(-np.inf), lg).astype(F32)
    order = np.argsort(-lg, kind="stable")
    sl = lg[order]
    n_fin = int(np.count_nonzero(np.isfinite(sl)))
    e = np.exp((sl - sl[0]).astype(F32), dtype=F32)
    sp = (e / e.sum(dtype=F32)).astype(F32)
    cum = np.cumsum(sp.astype(np.float64)).astype(F32)
    if p.top_p < 1.0:
        mask = cum > F32(p.top_p)
        mask[0] = False
        nz = int(np.count_nonzero(~mask[:n_fin]))
    else:
        nz = n_fin
    sup_idx = order[:nz].copy()
    sup_p = (sp[:nz] / sp[:nz].sum(dtype=F32)).astype(F32)
    target = F32(F32(u) * cum[nz - 1])
    j = min(int(np.count_nonzero(cum[:nz] <= target)), nz - 1)
    return TailResult(int(sup_idx[j]), sup_idx, sup_p, lg, cum[:nz].copy(), float(target))


def t3_inference_turbo(model: GPT2Oracle, embeds: np.ndarray, p: TurboParams, utt: int = 0,
                       keep_logits: bool = True, forced_tokens: Optional[Sequence[int]] = None,
                       mask_stop: bool = False) -> InferenceTrace:
    """R/t3.R:1578-1677 after prepare_input_embeds: batch 1, no CFG, no positional embedding for the
    generated tokens; first token from the prefill logits, then up to max_new_tokens more; the
    returned trace keeps the stop token (the R wrapper strips a trailing one, :1667-1671).
    Penalty ids (Q17): step 0 uses the raw BOS id as a 1-based column (= column BOS-1); later
    steps use the generated tokens only."""
    cfg = model.cfg
    tr = InferenceTrace()
    hidden, past = model.forward(embeds)
    last = hidden[:, -1, :]
    gen: List[int] = []
    for step in range(p.max_new_tokens + 1):
        logits = model.speech_logits(last)                    # [1,V]
        if keep_logits:
            tr.raw_logits.append(logits.copy())
        lc = logits[0].astype(F32).copy()
        if mask_stop:
            lc[cfg.stop_speech_token] = -np.inf
        pen = [cfg.start_speech_token - 1] if step == 0 else gen
        tail = turbo_sampling_tail(lc, pen, p, counter_uniform(p.seed, utt, step))
        tok = tail.token if forced_tokens is None else int(forced_tokens[step])
        if keep_logits:
            tr.tails.append(tail)
        tr.tokens.append(tok)
        gen.append(tok)
        if tok == cfg.stop_speech_token:
            tr.hit_eos = True
            break
        if step >= p.max_new_tokens:
            break
        e = model.w["speech_emb.weight"][tok].astype(F32)     # no position embedding (R/t3.R:1628)
        hidden, past = model.forward(e[None, None, :], past)
        last = hidden[:, -1, :]
    return tr


# --------------------------------------------------------------------------
# Prompt construction: conditioning encoder, perceiver resampler,
# text / speech embeddings.  R/t3.R:48-90, 140-316, 361-403; Turbo R/t3.R:1447-1477.
# --------------------------------------------------------------------------

PERCEIVER_QUERIES = 32      # R/t3.R:274
PERCEIVER_HEADS = 4         # R/t3.R:274
START_TEXT_TOKEN = 255      # R/t3.R:14
STOP_TEXT_TOKEN = 0         # R/t3.R:15


def attention_block(w: Dict[str, np.ndarray], pfx: str, x1: np.ndarray, x2: np.ndarray,
                    heads: int = PERCEIVER_HEADS) -> np.ndarray:
    """AttentionBlock.forward (R/t3.R:161-195): ONE LayerNorm applied to both inputs, q from x1, k/v from x2,
    plain softmax(q k^T / sqrt(Dh)) v, output projection, residual on the UN-normalised x1.  x1 [q, d], x2 [kv, d]."""
    d = x1.shape[-1]
    dh = d // heads
    nw, nb = w[pfx + "norm.weight"], w[pfx + "norm.bias"]
    n1 = layer_norm(x1, nw, nb, 1e-5)
    n2 = layer_norm(x2, nw, nb, 1e-5)
    q = linear(n1, w[pfx + "to_q.weight"], w[pfx + "to_q.bias"])
    k = linear(n2, w[pfx + "to_k.weight"], w[pfx + "to_k.bias"])
    v = linear(n2, w[pfx + "to_v.weight"], w[pfx + "to_v.bias"])
    qh = q.reshape(-1, heads, dh).transpose(1, 0, 2)[None]      # [1, H, q, Dh]
    kh = k.reshape(-1, heads, dh).transpose(1, 0, 2)[None]
    vh = v.reshape(-1, heads, dh).transpose(1, 0, 2)[None]
    o = sdpa(qh, kh, vh, None)[0].transpose(1, 0, 2).reshape(-1, d)
    h = linear(o, w[pfx + "proj_out.weight"], w[pfx + "proj_out.bias"])
    return (x1 + h).astype(F32)


def perceiver(w: Dict[str, np.ndarray], x: np.ndarray) -> np.ndarray:
    """Perceiver.forward (R/t3.R:227-241): the SAME attention block twice -- queries attend to the input, then
    the result attends to itself.  x [n, d] -> [32, d]."""
    q0 = w["cond_enc.perceiver.pre_attention_query"].reshape(-1, x.shape[-1]).astype(F32)
    pre = attention_block(w, "cond_enc.perceiver.attn.", q0, x)
    return attention_block(w, "cond_enc.perceiver.attn.", pre, pre)


def cond_enc(w: Dict[str, np.ndarray], speaker_emb: np.ndarray, cond_prompt_emb: Optional[np.ndarray],
             emotion_adv: Optional[float], use_perceiver: bool) -> np.ndarray:
    """T3CondEnc.forward (R/t3.R:278-315): [speaker projection | (perceiver of) prompt-speech embeddings | emotion row]."""
    spk = linear(speaker_emb.reshape(1, -1).astype(F32), w["cond_enc.spkr_enc.weight"], w["cond_enc.spkr_enc.bias"])
    parts = [spk]
    if cond_prompt_emb is not None:
        parts.append(perceiver(w, cond_prompt_emb) if use_perceiver else cond_prompt_emb.astype(F32))
    if emotion_adv is not None and "cond_enc.emotion_adv_fc.weight" in w:
        ew = w["cond_enc.emotion_adv_fc.weight"].reshape(-1)            # Linear(1, d, bias = FALSE)
        parts.append((ew * F32(emotion_adv)).astype(F32)[None, :])
    return np.concatenate(parts, axis=0).astype(F32)


def prepare_input_embeds(w: Dict[str, np.ndarray], speaker_emb: np.ndarray, cond_prompt_tokens: Optional[Sequence[int]],
                         emotion_adv: Optional[float], text_tokens: Sequence[int], cfg_weight: float,
                         start_speech_token: int) -> Tuple[np.ndarray, int]:
    """t3_inference's prompt (R/t3.R:473-494) through prepare_input_embeds (R/t3.R:372-403), 0-indexed ids.
    text_tokens: WITHOUT start/stop text tokens (they are added here, R/t3.R:473-477).
    Returns (embeds [rows, S0, d], len_cond); rows = 2 with CFG, row 1 = text span zeroed (R/t3.R:383-386)."""
    emb, pos = w["speech_emb.weight"], w["speech_pos_emb.emb.weight"]
    prompt = None
    if cond_prompt_tokens is not None:
        ids = np.asarray(cond_prompt_tokens, dtype=np.int64)
        prompt = (emb[ids] + pos[: ids.size]).astype(F32)                 # R/t3.R:366
    cond = cond_enc(w, speaker_emb, prompt, emotion_adv, use_perceiver=True)
    tt = np.concatenate([[START_TEXT_TOKEN], np.asarray(text_tokens, dtype=np.int64), [STOP_TEXT_TOKEN]])
    text = (w["text_emb.weight"][tt] + w["text_pos_emb.emb.weight"][: tt.size]).astype(F32)   # R/t3.R:379-380
    bos = (emb[start_speech_token] + pos[0]).astype(F32)[None, :]        # R/t3.R:389-390
    row0 = np.concatenate([cond, text, bos], axis=0)
    if cfg_weight > 0.0:
        row1 = np.concatenate([cond, np.zeros_like(text), bos], axis=0)
        return np.stack([row0, row1]).astype(F32), cond.shape[0]
    return row0[None].astype(F32), cond.shape[0]


def prepare_input_embeds_turbo(w: Dict[str, np.ndarray], speaker_emb: np.ndarray,
                               cond_prompt_tokens: Optional[Sequence[int]], text_tokens: Sequence[int],
                               start_speech_token: int) -> Tuple[np.ndarray, int]:
    """t3_model_turbo$prepare_input_embeds (R/t3.R:1447-1477) as t3_inference_turbo calls it (R/t3.R:1593-1600):
    no position terms, no perceiver, no emotion row, no text start/stop tokens, one row."""
    emb = w["speech_emb.weight"]
    prompt = None
    if cond_prompt_tokens is not None:
        prompt = emb[np.asarray(cond_prompt_tokens, dtype=np.int64)].astype(F32)     # R/t3.R:1452
    cond = cond_enc(w, speaker_emb, prompt, None, use_perceiver=False)
    text = w["text_emb.weight"][np.asarray(text_tokens, dtype=np.int64)].astype(F32)
    bos = emb[start_speech_token].astype(F32)[None, :]
    return np.concatenate([cond, text, bos], axis=0)[None].astype(F32), cond.shape[0]
