"""CPU restatement (numpy, fp32) of chatterbox's S3Gen flow-matching decoder: the CFM estimator and its Euler solvers.

TEST INFRASTRUCTURE ONLY.  Nothing under ``chatterbox_amd/`` imports this module; only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s CPU-baseline leg may.  It is the checker, never the thing measured or
shipped.

What it restates (citations into /root/reference, read as text) -- SURVEY §8 row f4, first closed piece:

* sinusoidal_pos_emb / timestep_embedding ........ R/s3gen.R:64-115
* causal_conv1d / causal_block1d (conv+LN+Mish) .. R/s3gen.R:124-197
* causal_resnet_block1d ......................... R/s3gen.R:204-231
* cfm_attention (8 x 64, no mask) ............... R/s3gen.R:238-291
* gelu_with_proj / feed_forward ................. R/s3gen.R:297-342
* basic_transformer_block ....................... R/s3gen.R:348-373
* cfm_estimator$forward ......................... R/s3gen.R:446-520
* causal_cfm$forward / solve_euler / basic_euler  R/s3gen.R:594-751

PARITY UNPINNED against the reference: it ships no S3Gen fixture (outputs/ holds T3 dumps only) and the
weights (s3gen.safetensors) are not in the container.  What pins this restatement is an op-for-op run of the same
sequence on CPU libtorch through python `torch` (tests/test_s3gen_oracle_vs_torch.py): nn.Conv1d, F.layer_norm,
F.softplus, F.gelu(approximate="tanh"), softmax, matmul -- the ATen kernels R torch dispatches to.

Layouts follow the reference: sequences are (B, C, T); weights are the tensors of `load_cfm_estimator_weights`
(R/s3gen.R:1070-1178) under their checkpoint keys (conv weights [out, in, k], linear weights [out, in]).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np

F32 = np.float32


def _lin(x: np.ndarray, w: np.ndarray, b: Optional[np.ndarray] = None) -> np.ndarray:
    y = x.astype(F32) @ w.astype(F32).T
    if b is not None:
        y = y + b.astype(F32)
    return y.astype(F32)


def silu(x):
    return (x / (F32(1) + np.exp(-x, dtype=F32))).astype(F32)


def softplus(x):
    # torch.nn.functional.softplus: beta 1, threshold 20 (x above it is returned unchanged)
    x = x.astype(F32)
    return np.where(x > F32(20), x, np.log1p(np.exp(np.minimum(x, F32(20)), dtype=F32), dtype=F32)).astype(F32)


def mish(x):
    """R/s3gen.R:165-168: x * tanh(softplus(x))."""
    return (x * np.tanh(softplus(x), dtype=F32)).astype(F32)


def gelu_tanh(x):
    x = x.astype(F32)
    inner = F32(0.7978845608028654) * (x + F32(0.044715) * x * x * x)
    return (F32(0.5) * x * (F32(1) + np.tanh(inner, dtype=F32))).astype(F32)


def layer_norm(x, w, b, eps=1e-5):
    """torch::nn_layer_norm over the last axis (biased variance, eps 1e-5)."""
    x = x.astype(F32)
    mu = x.mean(axis=-1, keepdims=True, dtype=F32)
    xc = (x - mu).astype(F32)
    var = (xc * xc).mean(axis=-1, keepdims=True, dtype=F32)
    return ((xc / np.sqrt(var + F32(eps), dtype=F32)) * w.astype(F32) + b.astype(F32)).astype(F32)


def sinusoidal_pos_emb(t: np.ndarray, dim: int = 320, scale: float = 1000.0) -> np.ndarray:
    """R/s3gen.R:72-91.  t [B] -> [B, dim] = cat(sin, cos) of (scale * t) * exp(arange(half) * -log(1e4) / (half - 1))."""
    t = np.asarray(t, dtype=F32).reshape(-1)
    half = dim // 2
    emb = np.exp(np.arange(half, dtype=F32) * F32(-math.log(10000.0) / (half - 1)), dtype=F32)
    e = ((F32(scale) * t)[:, None] * emb[None, :]).astype(F32)
    return np.concatenate([np.sin(e, dtype=F32), np.cos(e, dtype=F32)], axis=-1).astype(F32)


def causal_conv1d(x: np.ndarray, w: np.ndarray, b: np.ndarray) -> np.ndarray:
    """R/s3gen.R:138-142: left-pad (k - 1) zeros, conv1d.  x (B, Cin, T), w [Cout, Cin, k] -> (B, Cout, T)."""
    B, Cin, T = x.shape
    k = w.shape[2]
    xp = np.concatenate([np.zeros((B, Cin, k - 1), F32), x.astype(F32)], axis=2)
    y = np.zeros((B, w.shape[0], T), F32)
    for j in range(k):
        # out[:, o, t] += sum_c w[o, c, j] * xp[:, c, t + j]
        y += np.einsum("oc,bct->bot", w[:, :, j].astype(F32), xp[:, :, j:j + T]).astype(F32)
    return (y + b.astype(F32)[None, :, None]).astype(F32)


class CfmEstimatorOracle:
    """cfm_estimator (R/s3gen.R:389-528) over a dict of checkpoint tensors with `prefix` stripped."""

    def __init__(self, weights: Dict[str, np.ndarray], prefix: str = "flow.decoder.estimator.", heads: int = 8,
                 head_dim: int = 64, time_in: int = 320):
        self.w = {k[len(prefix):]: np.asarray(v, dtype=F32) for k, v in weights.items() if k.startswith(prefix)}
        self.heads, self.head_dim = heads, head_dim
        self.time_in = int(self.w["time_mlp.linear_1.weight"].shape[1]) if "time_mlp.linear_1.weight" in self.w else time_in
        self.n_mid = 0
        while f"mid_blocks.{self.n_mid}.0.block1.block.0.weight" in self.w:
            self.n_mid += 1
        self.n_tfm = 0
        while f"down_blocks.0.1.{self.n_tfm}.norm1.weight" in self.w:
            self.n_tfm += 1
        self.meanflow = "time_embed_mixer.weight" in self.w

    # -- pieces ---------------------------------------------------------------------------------------------
    def time_embed(self, t: np.ndarray) -> np.ndarray:
        w = self.w
        e = sinusoidal_pos_emb(t, self.time_in)                                      # R/s3gen.R:459
        e = _lin(e, w["time_mlp.linear_1.weight"], w["time_mlp.linear_1.bias"])
        e = silu(e)
        return _lin(e, w["time_mlp.linear_2.weight"], w["time_mlp.linear_2.bias"])   # :460

    def block1d(self, x, mask, p):
        """causal_block1d (R/s3gen.R:186-196): conv(x * mask) -> LayerNorm over channels -> Mish -> * mask."""
        w = self.w
        h = causal_conv1d(x * mask, w[p + "block.0.weight"], w[p + "block.0.bias"])
        h = layer_norm(h.transpose(0, 2, 1), w[p + "block.2.weight"], w[p + "block.2.bias"]).transpose(0, 2, 1)
        return (mish(h) * mask).astype(F32)

    def resnet(self, x, mask, temb, p):
        """causal_resnet_block1d (R/s3gen.R:223-230)."""
        w = self.w
        h = self.block1d(x, mask, p + "block1.")
        h = (h + _lin(mish(temb), w[p + "mlp.1.weight"], w[p + "mlp.1.bias"])[:, :, None]).astype(F32)
        h = self.block1d(h, mask, p + "block2.")
        res = causal_conv1d(x * mask, w[p + "res_conv.weight"], w[p + "res_conv.bias"])   # 1 x 1 conv
        return (h + res).astype(F32)

    def attention(self, x, p):
        """cfm_attention (R/s3gen.R:260-290): x (B, T, C); no mask (compute_attn_mask returns NULL, :522-527)."""
        w = self.w
        B, T, _ = x.shape
        H, Dh = self.heads, self.head_dim
        q = _lin(x, w[p + "to_q.weight"]).reshape(B, T, H, Dh).transpose(0, 2, 1, 3)
        k = _lin(x, w[p + "to_k.weight"]).reshape(B, T, H, Dh).transpose(0, 2, 1, 3)
        v = _lin(x, w[p + "to_v.weight"]).reshape(B, T, H, Dh).transpose(0, 2, 1, 3)
        s = (np.matmul(q, k.transpose(0, 1, 3, 2)).astype(F32) * F32(Dh ** -0.5)).astype(F32)
        s = s - s.max(axis=-1, keepdims=True)
        e = np.exp(s, dtype=F32)
        a = (e / e.sum(axis=-1, keepdims=True, dtype=F32)).astype(F32)
        o = np.matmul(a, v).astype(F32).transpose(0, 2, 1, 3).reshape(B, T, H * Dh)
        return _lin(o, w[p + "to_out.0.weight"], w[p + "to_out.0.bias"])

    def transformer(self, x, p):
        """basic_transformer_block (R/s3gen.R:359-372)."""
        w = self.w
        n = layer_norm(x, w[p + "norm1.weight"], w[p + "norm1.bias"])
        x = (x + self.attention(n, p + "attn1.")).astype(F32)
        n = layer_norm(x, w[p + "norm3.weight"], w[p + "norm3.bias"])
        f = gelu_tanh(_lin(n, w[p + "ff.net.0.proj.weight"], w[p + "ff.net.0.proj.bias"]))
        f = _lin(f, w[p + "ff.net.2.weight"], w[p + "ff.net.2.bias"])
        return (x + f).astype(F32)

    # -- forward (R/s3gen.R:446-520) ------------------------------------------------------------------------
    def forward(self, x, mask, mu, t, spks=None, cond=None, r=None) -> np.ndarray:
        """x, mu, cond (B, 80, T); mask (B, 1, T); t [B]; spks (B, 80); r [B] (meanflow).  Returns (B, 80, T)."""
        w = self.w
        x, mask, mu = x.astype(F32), mask.astype(F32), mu.astype(F32)
        T = x.shape[2]
        temb = self.time_embed(t)
        if self.meanflow and r is not None:                                          # :463-468
            remb = self.time_embed(r)
            temb = _lin(np.concatenate([temb, remb], axis=1), w["time_embed_mixer.weight"])
        h = np.concatenate([x, mu], axis=1)                                          # :471-478
        if spks is not None:
            h = np.concatenate([h, np.broadcast_to(spks.astype(F32)[:, :, None], spks.shape + (T,))], axis=1)
        if cond is not None:
            h = np.concatenate([h, cond.astype(F32)], axis=1)
        h = self.resnet(h, mask, temb, "down_blocks.0.0.")                           # :481
        h = h.transpose(0, 2, 1)
        for i in range(self.n_tfm):
            h = self.transformer(h, f"down_blocks.0.1.{i}.")
        h = h.transpose(0, 2, 1)
        skip = h                                                                     # :488
        h = causal_conv1d(h * mask, w["down_blocks.0.2.weight"], w["down_blocks.0.2.bias"])
        for m in range(self.n_mid):                                                  # :492-499
            h = self.resnet(h, mask, temb, f"mid_blocks.{m}.0.")
            h = h.transpose(0, 2, 1)
            for j in range(self.n_tfm):
                h = self.transformer(h, f"mid_blocks.{m}.1.{j}.")
            h = h.transpose(0, 2, 1)
        h = np.concatenate([h, skip], axis=1)                                        # :503
        h = self.resnet(h, mask, temb, "up_blocks.0.0.")
        h = h.transpose(0, 2, 1)
        for i in range(self.n_tfm):
            h = self.transformer(h, f"up_blocks.0.1.{i}.")
        h = h.transpose(0, 2, 1)
        h = causal_conv1d(h * mask, w["up_blocks.0.2.weight"], w["up_blocks.0.2.bias"])   # :513
        h = self.block1d(h, mask, "final_block.")                                    # :516
        return causal_conv1d(h * mask, w["final_proj.weight"], w["final_proj.bias"])      # :517 (1 x 1)


def linspace_f32(n_timesteps: int) -> np.ndarray:
    """torch_linspace(0, 1, n + 1, dtype = float32) as ATen computes it on the CPU: step = 1 / n in float32, the first
    half counted up from the start, the second half counted down from the end."""
    n = n_timesteps
    step = F32(1) / F32(n)
    return np.array([F32(i) * step if i < (n + 1) // 2 else F32(1) - F32(n - i) * step for i in range(n + 1)], dtype=F32)


def t_span_cosine(n_timesteps: int) -> np.ndarray:
    """R/s3gen.R:622-625: linspace(0, 1, n + 1) in float32, then 1 - cos(t * 0.5 * pi)."""
    ts = linspace_f32(n_timesteps)
    return (F32(1) - np.cos((ts * F32(0.5)).astype(F32) * F32(math.pi), dtype=F32)).astype(F32)


def solve_euler(est: CfmEstimatorOracle, z, mu, mask, spks, cond, n_timesteps: int = 10, cfg_rate: float = 0.7,
                keep: Optional[list] = None) -> np.ndarray:
    """causal_cfm$forward + solve_euler, untraced (R/s3gen.R:594-636, 651-751): cosine schedule, classifier-free
    guidance as a batch of two (row 2: mu, spks, cond zeroed), t accumulated in float32.  z is the start noise
    (rand_noise[:, :, 1:T] * temperature, :616-619 -- a torch_randn buffer the caller owns).  All (1, 80, T)."""
    ts = t_span_cosine(n_timesteps)
    x = z.astype(F32)
    t = F32(ts[0])
    dt = F32(ts[1] - ts[0])
    for step in range(1, len(ts)):                                   # R: step in 2:length(t_span)
        x_in = np.concatenate([x, x], axis=0)
        mask_in = np.concatenate([mask, mask], axis=0).astype(F32)
        mu_in = np.concatenate([mu, np.zeros_like(mu)], axis=0).astype(F32)
        t_in = np.array([t, t], dtype=F32)
        spks_in = np.concatenate([spks, np.zeros_like(spks)], axis=0).astype(F32)
        cond_in = np.concatenate([cond, np.zeros_like(cond)], axis=0).astype(F32)
        d = est.forward(x_in, mask_in, mu_in, t_in, spks_in, cond_in)
        if keep is not None:
            keep.append(d.copy())
        dphi = (F32(1.0 + cfg_rate) * d[0:1] - F32(cfg_rate) * d[1:2]).astype(F32)      # :734
        x = (x + dt * dphi).astype(F32)                              # :737
        t = F32(t + dt)
        if step < len(ts) - 1:
            dt = F32(ts[step + 1] - ts[step])                        # :745-747
    return x


def basic_euler(est: CfmEstimatorOracle, z, mu, mask, spks, cond, n_timesteps: int = 2) -> np.ndarray:
    """MeanFlow solver (R/s3gen.R:639-649): linear schedule, no CFG, the estimator sees (t, r) of every step."""
    ts = linspace_f32(n_timesteps)
    x = z.astype(F32)
    for step in range(len(ts) - 1):
        t, r = ts[step:step + 1], ts[step + 1:step + 2]
        d = est.forward(x, mask, mu, t, spks, cond, r=r)
        x = (x + F32(r[0] - t[0]) * d).astype(F32)
    return x


# =====================================================================================================================
# Conformer encoder (R/conformer.R:18-593) and the flow module around it (R/s3gen.R:768-891).  Same status as above:
# pinned op-for-op against CPU libtorch (tests/test_s3gen_oracle_vs_torch.py), PARITY UNPINNED against the reference itself.
# One utterance (B = 1), every position valid -- what s3gen$inference passes (R/s3gen.R:1006-1011): the padding masks of
# R/conformer.R:259-267, 397-400 are then all-true and drop out.
# =====================================================================================================================

def conv1d_valid(x: np.ndarray, w: np.ndarray, b: np.ndarray) -> np.ndarray:
    """nn_conv1d without padding on rows: x [T, Cin], w [Cout, Cin, k] -> [T - k + 1, Cout]."""
    k = w.shape[2]
    To = x.shape[0] - k + 1
    y = np.zeros((To, w.shape[0]), F32)
    for j in range(k):
        y += (x[j:j + To].astype(F32) @ w[:, :, j].astype(F32).T).astype(F32)
    return (y + b.astype(F32)).astype(F32)


def rel_pos_emb(T: int, d_model: int) -> np.ndarray:
    """espnet_rel_positional_encoding (R/conformer.R:34-95): the slice pe[center - T + 1 : center + T - 1] = rows for the
    relative positions +(T - 1) ... 0 ... -(T - 1); even columns sin(pos * div), odd columns cos(pos * div), float32."""
    div = np.exp(np.arange(0, d_model, 2, dtype=F32) * F32(-(math.log(10000.0) / d_model)), dtype=F32)
    rel = np.arange(T - 1, -T, -1).astype(F32)                       # +(T-1) .. -(T-1)
    ang = (rel[:, None] * div[None, :]).astype(F32)
    pe = np.zeros((2 * T - 1, d_model), F32)
    pe[:, 0::2] = np.sin(ang, dtype=F32)
    pe[:, 1::2] = np.cos(ang, dtype=F32)
    return pe


def rel_shift(bd: np.ndarray) -> np.ndarray:
    """rel_position_attention$rel_shift (R/conformer.R:281-307), literally: pad, view, drop the first row, view, crop.
    bd [H, T, 2T - 1] -> [H, T, T]."""
    H, T, P = bd.shape
    x = np.concatenate([np.zeros((H, T, 1), F32), bd], axis=2)        # (H, T, P + 1)
    x = x.reshape(H, P + 1, T)[:, 1:, :]                              # (H, P, T)
    return x.reshape(H, T, P)[:, :, :T].copy()


class ConformerOracle:
    """upsample_conformer_encoder_full (R/conformer.R:464-593) over checkpoint tensors with `prefix` stripped."""

    def __init__(self, weights: Dict[str, np.ndarray], prefix: str = "flow.encoder.", heads: int = 8):
        self.w = {k[len(prefix):]: np.asarray(v, dtype=F32) for k, v in weights.items() if k.startswith(prefix)}
        self.heads = heads
        self.d = self.w["embed.out.0.weight"].shape[0]
        self.n_blocks = 0
        while f"encoders.{self.n_blocks}.norm_mha.weight" in self.w:
            self.n_blocks += 1
        self.n_up = 0
        while f"up_encoders.{self.n_up}.norm_mha.weight" in self.w:
            self.n_up += 1

    def embed(self, x, p):
        """linear_no_subsampling (R/conformer.R:124-134): Linear -> LayerNorm (eps 1e-5) -> * sqrt(d); pos_emb beside it."""
        w = self.w
        h = layer_norm(_lin(x, w[p + "out.0.weight"], w[p + "out.0.bias"]), w[p + "out.1.weight"], w[p + "out.1.bias"])
        return (h * F32(math.sqrt(self.d))).astype(F32), rel_pos_emb(x.shape[0], self.d)

    def pre_lookahead(self, x):
        """pre_lookahead_layer (R/conformer.R:160-185): right-pad 3, conv k = 4, leaky_relu, left-pad 2, conv k = 3, + x."""
        w = self.w
        k1 = w["pre_lookahead_layer.conv1.weight"].shape[2]
        h = np.concatenate([x, np.zeros((k1 - 1, x.shape[1]), F32)], axis=0)
        h = conv1d_valid(h, w["pre_lookahead_layer.conv1.weight"], w["pre_lookahead_layer.conv1.bias"])
        h = np.where(h >= 0, h, h * F32(0.01)).astype(F32)
        h = np.concatenate([np.zeros((2, x.shape[1]), F32), h], axis=0)
        h = conv1d_valid(h, w["pre_lookahead_layer.conv2.weight"], w["pre_lookahead_layer.conv2.bias"])
        return (h + x).astype(F32)

    def attention(self, x, pos_emb, p):
        """rel_position_attention$forward (R/conformer.R:226-279), B = 1, no padding."""
        w = self.w
        T, H = x.shape[0], self.heads
        dk = self.d // H
        q = _lin(x, w[p + "linear_q.weight"], w[p + "linear_q.bias"]).reshape(T, H, dk).transpose(1, 0, 2)
        k = _lin(x, w[p + "linear_k.weight"], w[p + "linear_k.bias"]).reshape(T, H, dk).transpose(1, 0, 2)
        v = _lin(x, w[p + "linear_v.weight"], w[p + "linear_v.bias"]).reshape(T, H, dk).transpose(1, 0, 2)
        pp = _lin(pos_emb, w[p + "linear_pos.weight"]).reshape(-1, H, dk).transpose(1, 0, 2)      # (H, 2T-1, dk)
        qu = (q + w[p + "pos_bias_u"][:, None, :]).astype(F32)
        qv = (q + w[p + "pos_bias_v"][:, None, :]).astype(F32)
        ac = np.matmul(qu, k.transpose(0, 2, 1)).astype(F32)
        bd = rel_shift(np.matmul(qv, pp.transpose(0, 2, 1)).astype(F32))
        s = ((ac + bd) / F32(math.sqrt(dk))).astype(F32)
        s = s - s.max(axis=-1, keepdims=True)
        e = np.exp(s, dtype=F32)
        a = (e / e.sum(axis=-1, keepdims=True, dtype=F32)).astype(F32)
        o = np.matmul(a, v).astype(F32).transpose(1, 0, 2).reshape(T, H * dk)
        return _lin(o, w[p + "linear_out.weight"], w[p + "linear_out.bias"])

    def layer(self, x, pos_emb, p):
        """conformer_encoder_layer$forward (R/conformer.R:376-403): pre-norm attention and feed-forward, LayerNorm eps 1e-12."""
        w = self.w
        n = layer_norm(x, w[p + "norm_mha.weight"], w[p + "norm_mha.bias"], eps=1e-12)
        x = (x + self.attention(n, pos_emb, p + "self_attn.")).astype(F32)
        n = layer_norm(x, w[p + "norm_ff.weight"], w[p + "norm_ff.bias"], eps=1e-12)
        f = _lin(silu(_lin(n, w[p + "feed_forward.w_1.weight"], w[p + "feed_forward.w_1.bias"])),
                 w[p + "feed_forward.w_2.weight"], w[p + "feed_forward.w_2.bias"])
        return (x + f).astype(F32)

    def upsample(self, x):
        """upsample_1d (R/conformer.R:426-444): nearest x 2, left-pad 4, conv k = 5."""
        w = self.w
        up = np.repeat(x, 2, axis=0)
        k = w["up_layer.conv.weight"].shape[2]
        up = np.concatenate([np.zeros((k - 1, x.shape[1]), F32), up], axis=0)
        return conv1d_valid(up, w["up_layer.conv.weight"], w["up_layer.conv.bias"])

    def forward(self, x: np.ndarray) -> np.ndarray:
        """x [T, d] (embedded tokens) -> [2T, d]  (R/conformer.R:517-578)."""
        w = self.w
        xs, pe = self.embed(x.astype(F32), "embed.")
        xs = self.pre_lookahead(xs)
        for i in range(self.n_blocks):
            xs = self.layer(xs, pe, f"encoders.{i}.")
        xs = self.upsample(xs)
        xs, pe = self.embed(xs, "up_embed.")
        for i in range(self.n_up):
            xs = self.layer(xs, pe, f"up_encoders.{i}.")
        return layer_norm(xs, w["after_norm.weight"], w["after_norm.bias"])


def flow_inputs(weights: Dict[str, np.ndarray], token, prompt_token, prompt_feat, embedding, heads: int = 8,
                finalize: bool = True, vocab_size: int = 6561):
    """causal_masked_diff_xvec$forward up to the decoder call (R/s3gen.R:805-872), B = 1: returns (mu (1, 80, T), spks (1, 80),
    cond (1, 80, T), mel_len1).  token / prompt_token: 0-indexed ids; prompt_feat [frames, 80]; embedding [192]."""
    w = weights
    emb = np.asarray(embedding, dtype=F32).reshape(-1)
    emb = (emb / np.maximum(np.sqrt((emb * emb).sum(dtype=F32), dtype=F32), F32(1e-12))).astype(F32)      # nnf_normalize, :822
    spks = _lin(emb[None, :], w["flow.spk_embed_affine_layer.weight"], w["flow.spk_embed_affine_layer.bias"])   # :823
    tok = np.concatenate([np.asarray(prompt_token, np.int64).reshape(-1), np.asarray(token, np.int64).reshape(-1)])   # :826
    tok = np.clip(tok, 0, vocab_size - 1)                                                                   # :833
    x = w["flow.input_embedding.weight"][tok].astype(F32)                                                    # :836 (mask all ones)
    h = ConformerOracle(w, "flow.encoder.", heads).forward(x)                                               # :839
    if not finalize:
        h = h[: h.shape[0] - 3 * 2]                                                                         # :844-846
    mel_len1 = 2 * int(np.asarray(prompt_token).size)
    mel_len2 = h.shape[0] - mel_len1
    h = _lin(h, w["flow.encoder_proj.weight"], w["flow.encoder_proj.bias"])                                  # :854
    pf = np.asarray(prompt_feat, dtype=F32)
    conds = np.zeros((mel_len1 + mel_len2, h.shape[1]), F32)                                                # :857-865
    n = min(pf.shape[0], mel_len1)
    conds[:n] = pf[:n]
    return h.T[None].copy(), spks, conds.T[None].copy(), mel_len1


def flow_inference(weights, token, prompt_token, prompt_feat, embedding, noise, n_timesteps: int = 10, heads: int = 8,
                   cfm_heads: int = 8, finalize: bool = True, meanflow: bool = False) -> np.ndarray:
    """causal_masked_diff_xvec$forward (R/s3gen.R:805-890): tokens -> mel of the generated part (1, 80, mel_len2)."""
    mu, spks, cond, mel_len1 = flow_inputs(weights, token, prompt_token, prompt_feat, embedding, heads, finalize)
    T = mu.shape[2]
    mask = np.ones((1, 1, T), F32)
    est = CfmEstimatorOracle(weights, heads=cfm_heads)
    z = np.asarray(noise, dtype=F32)[:, :, :T]
    feat = basic_euler(est, z, mu, mask, spks, cond, n_timesteps) if meanflow else solve_euler(est, z, mu, mask, spks, cond, n_timesteps)
    return feat[:, :, mel_len1:].copy()                                                                      # :887


# =====================================================================================================================
# HiFT vocoder (R/hifigan.R:13-627): F0 predictor -> harmonic-plus-noise source -> ISTFTNet decoder.  Same status as above:
# pinned op-for-op against CPU libtorch (conv1d, conv_transpose1d, stft, istft, cumsum ...), PARITY UNPINNED against the
# reference itself.  One utterance (B = 1).  The reference draws the harmonics' start phases and the source noise from
# torch's generator (R/hifigan.R:237-249); here the caller hands them in.
# =====================================================================================================================

def conv1d_full(x: np.ndarray, w: np.ndarray, b: Optional[np.ndarray], stride: int = 1, padding: int = 0, dilation: int = 1) -> np.ndarray:
    """nn_conv1d on one sequence: x [Cin, L], w [Cout, Cin, k] -> [Cout, Lout]."""
    Cin, L = x.shape
    k = w.shape[2]
    xp = np.concatenate([np.zeros((Cin, padding), F32), x.astype(F32), np.zeros((Cin, padding), F32)], axis=1)
    Lout = (L + 2 * padding - dilation * (k - 1) - 1) // stride + 1
    y = np.zeros((w.shape[0], Lout), F32)
    for j in range(k):
        seg = xp[:, j * dilation: j * dilation + (Lout - 1) * stride + 1: stride]
        y += (w[:, :, j].astype(F32) @ seg).astype(F32)
    if b is not None:
        y = y + b.astype(F32)[:, None]
    return y.astype(F32)


def conv_transpose1d(x: np.ndarray, w: np.ndarray, b: np.ndarray, stride: int, padding: int, output_padding: int = 0) -> np.ndarray:
    """nn_conv_transpose1d on one sequence: x [Cin, L], w [Cin, Cout, k] -> [Cout, (L - 1) stride - 2 padding + k + output_padding]."""
    Cin, L = x.shape
    k = w.shape[2]
    full = np.zeros((w.shape[1], (L - 1) * stride + k + output_padding), F32)
    for j in range(k):
        full[:, j: j + (L - 1) * stride + 1: stride] += (w[:, :, j].astype(F32).T @ x.astype(F32)).astype(F32)
    out = full[:, padding: full.shape[1] - padding]
    return (out + b.astype(F32)[:, None]).astype(F32)


def weight_norm_weight(g: np.ndarray, v: np.ndarray) -> np.ndarray:
    """get_parametrized_weight (R/hifigan.R:649-667): w = g * v / ||v||, the norm over all but the first axis."""
    nrm = np.sqrt((v.reshape(v.shape[0], -1).astype(F32) ** 2).sum(axis=1, dtype=F32), dtype=F32).reshape(-1, 1, 1)
    return (g.astype(F32) * v.astype(F32) / nrm).astype(F32)


def snake(x: np.ndarray, alpha: np.ndarray) -> np.ndarray:
    """snake_activation (R/hifigan.R:75-86), alpha_logscale = FALSE: x + 1 / (a + 1e-9) * sin(x a)^2, per channel."""
    a = alpha.astype(F32)[:, None]
    return (x + (F32(1.0) / (a + F32(1e-9))) * np.sin(x * a, dtype=F32) ** 2).astype(F32)


def elu(x):
    return np.where(x > 0, x, np.expm1(np.minimum(x, F32(0)), dtype=F32)).astype(F32)


def leaky_relu(x, slope):
    return np.where(x >= 0, x, x * F32(slope)).astype(F32)


HIFT_UPS = ((8, 16), (5, 11), (3, 7))          # create_s3gen_vocoder (R/hifigan.R:804-805): (rate, kernel)
HIFT_RES_K = (3, 7, 11)
HIFT_SRC_K = (7, 7, 11)
HIFT_DIL = (1, 3, 5)


class HiftOracle:
    """hift_generator (R/hifigan.R:337-627) as create_s3gen_vocoder configures it (:788-817), weights through
    load_hifigan_weights' key map (:640-777) with the weight-norm pairs already folded."""

    def __init__(self, weights: Dict[str, np.ndarray], prefix: str = "mel2wav.", sampling_rate: int = 24000, n_fft: int = 16, hop: int = 4,
                 ups=HIFT_UPS):
        raw = {k[len(prefix):]: np.asarray(v, dtype=F32) for k, v in weights.items() if k.startswith(prefix)}
        w: Dict[str, np.ndarray] = {}
        for k, v in raw.items():
            if k.endswith(".parametrizations.weight.original0"):
                base = k[: -len(".parametrizations.weight.original0")]
                w[base + ".weight"] = weight_norm_weight(v, raw[base + ".parametrizations.weight.original1"])
            elif not k.endswith(".parametrizations.weight.original1"):
                w[k] = v
        self.w, self.sr, self.n_fft, self.hop, self.ups = w, sampling_rate, n_fft, hop, tuple(ups)
        self.total_up = int(np.prod([u for u, _ in ups])) * hop
        self.n_harm = w["m_source.l_linear.weight"].shape[1]          # harmonic_num + 1
        n = np.arange(n_fft, dtype=np.float64)
        self.window = (0.5 * (1 - np.cos(2 * np.pi * n / n_fft))).astype(F32)      # R/hifigan.R:458-461

    def f0_predictor(self, mel: np.ndarray) -> np.ndarray:
        """conv_rnn_f0_predictor (R/hifigan.R:174-180): mel [80, T] -> f0 [T]."""
        x = mel.astype(F32)
        for i in range(5):
            x = elu(conv1d_full(x, self.w[f"f0_predictor.condnet.{2 * i}.weight"], self.w[f"f0_predictor.condnet.{2 * i}.bias"], padding=1))
        y = _lin(x.T, self.w["f0_predictor.classifier.weight"], self.w["f0_predictor.classifier.bias"])[:, 0]
        return np.abs(y).astype(F32)

    def source(self, f0: np.ndarray, phase: np.ndarray, noise: np.ndarray) -> np.ndarray:
        """f0_upsamp + source_module_hn_nsf (R/hifigan.R:216-256, 291-307, 609-614): f0 [T] -> sine_merge [T * total_up].
        phase [n_harm] (entry 0 forced to 0), noise [n_harm, T_wav] standard normal."""
        up = np.repeat(f0.astype(F32), self.total_up)                                         # nn_upsample, nearest
        h = np.arange(1, self.n_harm + 1, dtype=F32)
        fmat = ((up[None, :] * h[:, None]).astype(F32) / F32(self.sr)).astype(F32)            # :230
        cs = np.cumsum(fmat.astype(np.float64), axis=1).astype(F32)                           # libtorch CPU cumsum: double accumulate
        theta = (F32(2 * math.pi) * np.mod(cs, F32(1))).astype(F32)                           # :234
        ph = phase.astype(F32).copy()
        ph[0] = 0
        sine = (F32(0.1) * np.sin(theta + ph[:, None], dtype=F32)).astype(F32)                # :242
        uv = (up > F32(10)).astype(F32)[None, :]                                              # voiced_threshold 10 (:803)
        namp = (uv * F32(0.003) + (F32(1) - uv) * F32(0.1) / F32(3)).astype(F32)              # :248
        sine = (sine * uv + namp * noise.astype(F32)).astype(F32)                             # :249-252
        lin = _lin(sine.T, self.w["m_source.l_linear.weight"], self.w["m_source.l_linear.bias"])[:, 0]
        return np.tanh(lin, dtype=F32)                                                        # :301

    def stft(self, s: np.ndarray) -> np.ndarray:
        """torch_stft(center = TRUE, reflect, hann, onesided) then cat(real, imag) (R/hifigan.R:464-481, 516-517): [n_fft + 2, frames]."""
        n, hop = self.n_fft, self.hop
        sp = np.pad(s.astype(F32), (n // 2, n // 2), mode="reflect")
        frames = 1 + (sp.size - n) // hop
        idx = np.arange(n)[None, :] + hop * np.arange(frames)[:, None]
        seg = (sp[idx] * self.window[None, :]).astype(F32)
        spec = np.fft.rfft(seg.astype(np.float64), axis=1)
        return np.concatenate([spec.real.T, spec.imag.T], axis=0).astype(F32)

    def istft(self, mag: np.ndarray, phase: np.ndarray) -> np.ndarray:
        """.istft (R/hifigan.R:483-501): clamp, polar -> complex, torch_istft(center = TRUE): windowed overlap-add over the
        window envelope, n_fft / 2 trimmed from both ends."""
        n, hop = self.n_fft, self.hop
        mag = np.minimum(mag.astype(F32), F32(100))
        re, im = (mag * np.cos(phase, dtype=F32)).astype(F32), (mag * np.sin(phase, dtype=F32)).astype(F32)
        frames = re.shape[1]
        seg = np.fft.irfft((re.astype(np.float64) + 1j * im.astype(np.float64)).T, n=n, axis=1).astype(F32)
        seg = (seg * self.window[None, :]).astype(F32)
        L = n + hop * (frames - 1)
        y = np.zeros(L, np.float64)
        env = np.zeros(L, np.float64)
        w2 = (self.window.astype(np.float64)) ** 2
        for f in range(frames):
            y[f * hop: f * hop + n] += seg[f]
            env[f * hop: f * hop + n] += w2
        y, env = y[n // 2: L - n // 2], env[n // 2: L - n // 2]
        return (y / env).astype(F32)

    def resblock(self, x, p, k):
        """hifigan_resblock$forward (R/hifigan.R:129-139)."""
        w = self.w
        for j, d in enumerate(HIFT_DIL):
            xt = snake(x, w[f"{p}.activations1.{j}.alpha"])
            xt = conv1d_full(xt, w[f"{p}.convs1.{j}.weight"], w[f"{p}.convs1.{j}.bias"], padding=(k * d - d) // 2, dilation=d)
            xt = snake(xt, w[f"{p}.activations2.{j}.alpha"])
            xt = conv1d_full(xt, w[f"{p}.convs2.{j}.weight"], w[f"{p}.convs2.{j}.bias"], padding=(k - 1) // 2)
            x = (xt + x).astype(F32)
        return x

    def decode(self, mel: np.ndarray, s: np.ndarray) -> np.ndarray:
        """hift_generator$decode (R/hifigan.R:503-578): mel [80, T], source s [T_wav] -> audio [T_wav]."""
        w = self.w
        s_stft = self.stft(s)
        x = conv1d_full(mel.astype(F32), w["conv_pre.weight"], w["conv_pre.bias"], padding=3)
        rates = [u for u, _ in self.ups]
        total = int(np.prod(rates))
        cum = np.cumprod(rates)
        for i, (u, k) in enumerate(self.ups):
            x = leaky_relu(x, 0.1)
            x = conv_transpose1d(x, w[f"ups.{i}.weight"], w[f"ups.{i}.bias"], stride=u, padding=(k - u) // 2)
            if i == len(self.ups) - 1:
                x = np.concatenate([x[:, 1:2], x], axis=1)                                   # reflection_pad1d(c(1, 0)), :455, :533
            du = total // int(cum[i])
            if du == 1:
                si = conv1d_full(s_stft, w[f"source_downs.{i}.weight"], w[f"source_downs.{i}.bias"])
            else:
                si = conv1d_full(s_stft, w[f"source_downs.{i}.weight"], w[f"source_downs.{i}.bias"], stride=du, padding=du // 2)
            si = self.resblock(si, f"source_resblocks.{i}", HIFT_SRC_K[i])
            L = min(x.shape[1], si.shape[1])
            x = (x[:, :L] + si[:, :L]).astype(F32)
            xs = None
            for j, rk in enumerate(HIFT_RES_K):
                r = self.resblock(x, f"resblocks.{i * len(HIFT_RES_K) + j}", rk)
                xs = r if xs is None else (xs + r).astype(F32)
            x = (xs / F32(len(HIFT_RES_K))).astype(F32)
        x = leaky_relu(x, 0.01)                                                               # nnf_leaky_relu default slope, :565
        x = conv1d_full(x, w["conv_post.weight"], w["conv_post.bias"], padding=3)
        nf = self.n_fft // 2 + 1
        mag = np.exp(x[:nf], dtype=F32)
        ph = np.sin(x[nf:], dtype=F32)                                                        # :571
        audio = self.istft(mag, ph)
        return np.clip(audio, F32(-0.99), F32(0.99)).astype(F32)

    def inference(self, mel: np.ndarray, phase: np.ndarray, noise: np.ndarray):
        """hift_generator$inference (R/hifigan.R:601-626) without a source cache: mel [80, T] -> (audio [T * 480], source)."""
        f0 = self.f0_predictor(mel)
        s = self.source(f0, phase, noise)
        return self.decode(mel, s), s
