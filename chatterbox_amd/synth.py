"""Synthetic T3 checkpoints with the reference's safetensors key names/shapes.

No real weights exist offline, so tests and bench.py build
checkpoints from a fixed numpy PCG64 seed: linear weights
N(0, .02^2), RMSNorm weights 1 + N(0, .1^2), embeddings / head N(0, .02^2).
Key names follow R/t3.R:1065-1097 and R/llama.R:566-607 of the reference so
the native loader is exercised on the layout of the real `t3_cfg.safetensors`.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Dict

import numpy as np

F32 = np.float32


@dataclass(frozen=True)
class T3Dims:
    """Shapes of the Llama-backbone T3 (R/llama.R:20-45, R/t3.R:11-36)."""
    hidden_size: int = 1024
    intermediate_size: int = 4096
    num_hidden_layers: int = 30
    num_attention_heads: int = 16
    head_dim: int = 64
    speech_vocab: int = 8194
    max_speech_pos: int = 4100
    start_speech_token: int = 6561
    stop_speech_token: int = 6562
    # prompt construction (R/t3.R:11-36, 263-275)
    text_vocab: int = 704
    max_text_pos: int = 2050
    speaker_embed: int = 256
    n_query: int = 32


FULL = T3Dims()
# reduced shape for committed goldens / fast CPU parity (head_dim stays 64)
TINY = T3Dims(hidden_size=512, intermediate_size=1024, num_hidden_layers=2,
              num_attention_heads=8, head_dim=64, speech_vocab=1000, max_speech_pos=600,
              start_speech_token=990, stop_speech_token=991, text_vocab=300, max_text_pos=80)


def synth_llama_weights(dims: T3Dims, seed: int = 1234, head_scale: float = 1.0) -> Dict[str, np.ndarray]:
    """head_scale > 1 widens the speech-head weights so the next-token distribution is peaked
    like a trained model's (tens of tokens above min-p) instead of nearly flat."""
    rng = np.random.Generator(np.random.PCG64(seed))
    d, F, V = dims.hidden_size, dims.intermediate_size, dims.speech_vocab

    def lin(o, i):
        return (rng.standard_normal((o, i), dtype=F32) * F32(0.02)).astype(F32)

    def nrm():
        return (F32(1.0) + rng.standard_normal(d, dtype=F32) * F32(0.1)).astype(F32)

    w: Dict[str, np.ndarray] = {}
    for i in range(dims.num_hidden_layers):
        p = f"tfmr.layers.{i}."
        w[p + "input_layernorm.weight"] = nrm()
        w[p + "self_attn.q_proj.weight"] = lin(d, d)
        w[p + "self_attn.k_proj.weight"] = lin(d, d)
        w[p + "self_attn.v_proj.weight"] = lin(d, d)
        w[p + "self_attn.o_proj.weight"] = lin(d, d)
        w[p + "post_attention_layernorm.weight"] = nrm()
        w[p + "mlp.gate_proj.weight"] = lin(F, d)
        w[p + "mlp.up_proj.weight"] = lin(F, d)
        w[p + "mlp.down_proj.weight"] = lin(d, F)
    w["tfmr.norm.weight"] = nrm()
    w["speech_emb.weight"] = lin(V, d)
    w["speech_pos_emb.emb.weight"] = lin(dims.max_speech_pos, d)
    w["speech_head.weight"] = lin(V, d)
    if head_scale != 1.0:
        w["speech_head.weight"] = (w["speech_head.weight"] * F32(head_scale)).astype(F32)
    # prompt-construction tensors (keys of R/t3.R:1076-1141), drawn AFTER everything above so the decode-path
    # tensors of a given seed keep their values.  Scales chosen so the prompt rows look like the real ones
    # (std ~0.5, outputs/cond_enc_debug.safetensors) rather than like raw N(0, .02^2) embeddings.
    def vec(n, mean, std):
        return (F32(mean) + rng.standard_normal(n, dtype=F32) * F32(std)).astype(F32)

    w["text_emb.weight"] = (lin(dims.text_vocab, d) * F32(20.0)).astype(F32)
    w["text_pos_emb.emb.weight"] = (lin(dims.max_text_pos, d) * F32(5.0)).astype(F32)
    w["cond_enc.spkr_enc.weight"] = (lin(d, dims.speaker_embed) * F32(8.0)).astype(F32)
    w["cond_enc.spkr_enc.bias"] = vec(d, 0.0, 0.05)
    w["cond_enc.emotion_adv_fc.weight"] = vec(d, 0.0, 0.5).reshape(d, 1)
    qv = float(np.sqrt(3.0) * np.sqrt(2.0 / (2 * dims.n_query)))      # R/t3.R:218-221
    w["cond_enc.perceiver.pre_attention_query"] = rng.uniform(-qv, qv, size=(1, dims.n_query, d)).astype(F32)
    w["cond_enc.perceiver.attn.norm.weight"] = vec(d, 1.0, 0.1)
    w["cond_enc.perceiver.attn.norm.bias"] = vec(d, 0.0, 0.05)
    for nm in ("to_q", "to_k", "to_v", "proj_out"):
        w[f"cond_enc.perceiver.attn.{nm}.weight"] = (lin(d, d) * F32(2.0)).astype(F32)
        w[f"cond_enc.perceiver.attn.{nm}.bias"] = vec(d, 0.0, 0.05)
    return w


@dataclass(frozen=True)
class TurboDims:
    """Shapes of the GPT-2 backbone Turbo T3 (R/t3.R:1154-1179, 1347-1357)."""
    hidden_size: int = 1024
    n_head: int = 16
    n_layer: int = 24
    speech_vocab: int = 6563
    start_speech_token: int = 6561
    stop_speech_token: int = 6562
    text_vocab: int = 50276          # R/t3.R:1159
    speaker_embed: int = 256


TURBO_FULL = TurboDims()
TURBO_TINY = TurboDims(hidden_size=512, n_head=8, n_layer=2, speech_vocab=900, start_speech_token=890,
                       stop_speech_token=891, text_vocab=400)


def synth_turbo_weights(dims: TurboDims, seed: int = 4321, head_scale: float = 1.0) -> Dict[str, np.ndarray]:
    """Turbo checkpoint keys of R/t3.R:1508-1549; linear weights stored [out, in] as the
    reference's loader copies them (R/t3.R:1540-1549)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    d, V = dims.hidden_size, dims.speech_vocab
    F = 4 * d

    def lin(o, i):
        return (rng.standard_normal((o, i), dtype=F32) * F32(0.02)).astype(F32)

    def vec(n, mean, std):
        return (F32(mean) + rng.standard_normal(n, dtype=F32) * F32(std)).astype(F32)

    w: Dict[str, np.ndarray] = {}
    for i in range(dims.n_layer):
        p = f"tfmr.h.{i}."
        w[p + "ln_1.weight"] = vec(d, 1.0, 0.1)
        w[p + "ln_1.bias"] = vec(d, 0.0, 0.05)
        w[p + "attn.c_attn.weight"] = lin(3 * d, d)
        w[p + "attn.c_attn.bias"] = vec(3 * d, 0.0, 0.02)
        w[p + "attn.c_proj.weight"] = lin(d, d)
        w[p + "attn.c_proj.bias"] = vec(d, 0.0, 0.02)
        w[p + "ln_2.weight"] = vec(d, 1.0, 0.1)
        w[p + "ln_2.bias"] = vec(d, 0.0, 0.05)
        w[p + "mlp.c_fc.weight"] = lin(F, d)
        w[p + "mlp.c_fc.bias"] = vec(F, 0.0, 0.02)
        w[p + "mlp.c_proj.weight"] = lin(d, F)
        w[p + "mlp.c_proj.bias"] = vec(d, 0.0, 0.02)
    w["tfmr.ln_f.weight"] = vec(d, 1.0, 0.1)
    w["tfmr.ln_f.bias"] = vec(d, 0.0, 0.05)
    w["speech_emb.weight"] = lin(V, d)
    w["speech_head.weight"] = (lin(V, d) * F32(head_scale)).astype(F32)
    w["speech_head.bias"] = vec(V, 0.0, 0.02)
    # prompt-construction tensors (R/t3.R:1508-1517), drawn after the decode-path tensors
    w["text_emb.weight"] = (lin(dims.text_vocab, d) * F32(20.0)).astype(F32)
    w["cond_enc.spkr_enc.weight"] = (lin(d, dims.speaker_embed) * F32(8.0)).astype(F32)
    w["cond_enc.spkr_enc.bias"] = vec(d, 0.0, 0.05)
    return w


def ensure_turbo_checkpoint(dims: TurboDims, seed: int = 4321, root: str | None = None, head_scale: float = 1.0) -> str:
    root = root or os.environ.get("T3MI_SYNTH_DIR", "/tmp/t3mi_synth")
    os.makedirs(root, exist_ok=True)
    tag = f"t3turbo_L{dims.n_layer}_d{dims.hidden_size}_V{dims.speech_vocab}_s{seed}_p1" + (
        "" if head_scale == 1.0 else f"_h{head_scale:g}")
    path = os.path.join(root, tag + ".safetensors")
    if not os.path.exists(path):
        write_safetensors(synth_turbo_weights(dims, seed, head_scale), path)
    return path


def synth_embeds(rows: int, s0: int, d: int, seed: int) -> np.ndarray:
    """Prompt embeddings ~ N(0, .53^2) (the std of the real activations in the
    reference's outputs/cond_enc_debug.safetensors), shape [rows, s0, d]."""
    rng = np.random.Generator(np.random.PCG64(seed))
    return (rng.standard_normal((rows, s0, d), dtype=F32) * F32(0.53)).astype(F32)


def to_bf16_bits(a: np.ndarray) -> np.ndarray:
    """float32 -> bf16 bit patterns (uint16), round to nearest even, NaN kept quiet."""
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
    r = ((u.astype(np.uint64) + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)
    nan = (u & 0x7FFFFFFF) > 0x7F800000
    r[nan] = ((u[nan] >> 16) | 0x40).astype(np.uint16)
    return r


def bf16_round(a: np.ndarray) -> np.ndarray:
    """float32 values rounded to bf16 precision (as float32)."""
    return (to_bf16_bits(a).astype(np.uint32) << 16).view(np.float32).reshape(a.shape)


def encode_tensor(a: np.ndarray, dtype: str) -> bytes:
    a = np.ascontiguousarray(a, dtype=np.float32)
    if dtype == "F32":
        return a.tobytes()
    if dtype == "F16":
        return a.astype(np.float16).tobytes()
    if dtype == "BF16":
        return to_bf16_bits(a).tobytes()
    raise ValueError(dtype)


def write_safetensors(weights: Dict[str, np.ndarray], path: str, dtype: str = "F32") -> str:
    """Plain safetensors writer (8-byte LE header length, JSON header, raw data)
    so no third-party package is needed on the GPU box.  dtype: storage type of every tensor
    ("F32", "F16" or "BF16": the three the native reader converts, R/safetensors.R:7-10)."""
    import json
    header = {}
    off = 0
    names = sorted(weights.keys())
    es = 4 if dtype == "F32" else 2
    for k in names:
        a = weights[k]
        assert a.dtype == np.float32
        n = a.size * es
        header[k] = {"dtype": dtype, "shape": list(a.shape), "data_offsets": [off, off + n]}
        off += n
    hj = json.dumps(header, separators=(",", ":")).encode()
    hj += b" " * ((8 - len(hj) % 8) % 8)
    tmp = path + ".tmp"
    with open(tmp, "wb") as f:
        f.write(len(hj).to_bytes(8, "little"))
        f.write(hj)
        for k in names:
            f.write(encode_tensor(weights[k], dtype))
    os.replace(tmp, path)
    return path


def checkpoint_path(dims: T3Dims, seed: int = 1234, root: str | None = None, head_scale: float = 1.0) -> str:
    root = root or os.environ.get("T3MI_SYNTH_DIR", "/tmp/t3mi_synth")
    os.makedirs(root, exist_ok=True)
    tag = f"t3_L{dims.num_hidden_layers}_d{dims.hidden_size}_F{dims.intermediate_size}_V{dims.speech_vocab}_s{seed}_p1" + (
        "" if head_scale == 1.0 else f"_h{head_scale:g}")
    return os.path.join(root, tag + ".safetensors")


def ensure_checkpoint(dims: T3Dims, seed: int = 1234, root: str | None = None, head_scale: float = 1.0) -> str:
    """Create the synthetic checkpoint file once and return its path."""
    path = checkpoint_path(dims, seed, root, head_scale)
    if not os.path.exists(path):
        write_safetensors(synth_llama_weights(dims, seed, head_scale), path)
    return path


# ---- S3Gen flow decoder: the CFM estimator ------------------------------------------------------------

@dataclass(frozen=True)
class CfmDims:
    """Shapes of cfm_estimator (R/s3gen.R:389-444): in = x + mu + spks + cond = 4 x mel bins."""
    mel: int = 80
    hidden: int = 256
    n_mid: int = 12
    n_tfm: int = 4
    heads: int = 8
    head_dim: int = 64
    time_in: int = 320          # sinusoidal_pos_emb(320), R/s3gen.R:400
    time_dim: int = 1024        # timestep_embedding(320, 1024), R/s3gen.R:401
    meanflow: bool = False


CFM_FULL = CfmDims()
CFM_TINY = CfmDims(mel=16, hidden=64, n_mid=2, n_tfm=1, heads=2, head_dim=64, time_in=64, time_dim=128)
CFM_PREFIX = "flow.decoder.estimator."      # load_s3gen_weights, R/s3gen.R:1231


def synth_cfm_weights(dims: CfmDims = CFM_FULL, seed: int = 777, prefix: str = CFM_PREFIX) -> Dict[str, np.ndarray]:
    """A synthetic checkpoint in the key layout load_cfm_estimator_weights reads (R/s3gen.R:1070-1178): conv weights
    [out, in, k], linear weights [out, in].  Scales keep every activation O(1) through 56 residual blocks (weights
    ~ 0.7 / sqrt(fan_in), LayerNorm weights 1 + N(0, .1^2), biases N(0, .05^2))."""
    rng = np.random.Generator(np.random.PCG64(seed))
    C, inner, Tm = dims.hidden, dims.heads * dims.head_dim, dims.time_dim
    w: Dict[str, np.ndarray] = {}

    def lin(name, o, i, bias=True, gain=0.7):
        w[prefix + name + ".weight"] = (rng.standard_normal((o, i), dtype=F32) * F32(gain / np.sqrt(i))).astype(F32)
        if bias:
            w[prefix + name + ".bias"] = (rng.standard_normal(o, dtype=F32) * F32(0.05)).astype(F32)

    def conv(name, o, i, k, gain=0.7):
        w[prefix + name + ".weight"] = (rng.standard_normal((o, i, k), dtype=F32) * F32(gain / np.sqrt(i * k))).astype(F32)
        w[prefix + name + ".bias"] = (rng.standard_normal(o, dtype=F32) * F32(0.05)).astype(F32)

    def norm(name, n):
        w[prefix + name + ".weight"] = (F32(1) + rng.standard_normal(n, dtype=F32) * F32(0.1)).astype(F32)
        w[prefix + name + ".bias"] = (rng.standard_normal(n, dtype=F32) * F32(0.05)).astype(F32)

    def resnet(p, cin):
        conv(p + "block1.block.0", C, cin, 3)
        norm(p + "block1.block.2", C)
        conv(p + "block2.block.0", C, C, 3)
        norm(p + "block2.block.2", C)
        lin(p + "mlp.1", C, Tm)
        conv(p + "res_conv", C, cin, 1)

    def tfm(p):
        norm(p + "norm1", C)
        lin(p + "attn1.to_q", inner, C, bias=False, gain=1.0)
        lin(p + "attn1.to_k", inner, C, bias=False, gain=1.0)
        lin(p + "attn1.to_v", inner, C, bias=False)
        lin(p + "attn1.to_out.0", C, inner, gain=0.4)
        norm(p + "norm3", C)
        lin(p + "ff.net.0.proj", 4 * C, C)
        lin(p + "ff.net.2", C, 4 * C, gain=0.4)

    lin("time_mlp.linear_1", Tm, dims.time_in)
    lin("time_mlp.linear_2", Tm, Tm)
    if dims.meanflow:
        lin("time_embed_mixer", Tm, 2 * Tm, bias=False)
    resnet("down_blocks.0.0.", 4 * dims.mel)
    for i in range(dims.n_tfm):
        tfm(f"down_blocks.0.1.{i}.")
    conv("down_blocks.0.2", C, C, 3)
    for m in range(dims.n_mid):
        resnet(f"mid_blocks.{m}.0.", C)
        for j in range(dims.n_tfm):
            tfm(f"mid_blocks.{m}.1.{j}.")
    resnet("up_blocks.0.0.", 2 * C)
    for i in range(dims.n_tfm):
        tfm(f"up_blocks.0.1.{i}.")
    conv("up_blocks.0.2", C, C, 3)
    conv("final_block.block.0", C, C, 3)
    norm("final_block.block.2", C)
    conv("final_proj", dims.mel, C, 1)
    return w


def ensure_cfm_checkpoint(dims: CfmDims = CFM_FULL, seed: int = 777, root: str | None = None) -> str:
    root = root or os.environ.get("T3MI_SYNTH_DIR", "/tmp/t3mi_synth")
    os.makedirs(root, exist_ok=True)
    tag = f"s3gen_cfm_C{dims.hidden}_M{dims.n_mid}x{dims.n_tfm}_mel{dims.mel}_t{dims.time_dim}_s{seed}" + ("_mf" if dims.meanflow else "")
    path = os.path.join(root, tag + ".safetensors")
    if not os.path.exists(path):
        write_safetensors(synth_cfm_weights(dims, seed), path)
    return path


def synth_cfm_inputs(dims: CfmDims, T: int, seed: int, B: int = 1):
    """mu, spks, cond, z for one solve, in the reference's layouts: (B, mel, T), (B, mel), (B, mel, T), (B, mel, T).
    cond is a prompt mel over the first third of the frames and zero after it (R/s3gen.R:857-866)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    mu = rng.standard_normal((B, dims.mel, T), dtype=F32)
    spks = (rng.standard_normal((B, dims.mel), dtype=F32) * F32(0.5)).astype(F32)
    cond = np.zeros((B, dims.mel, T), F32)
    cond[:, :, : T // 3] = rng.standard_normal((B, dims.mel, T // 3), dtype=F32)
    z = rng.standard_normal((B, dims.mel, T), dtype=F32)
    return mu, spks, cond, z


# ---- S3Gen flow module: token embedding, speaker projection, conformer encoder, encoder projection --------

@dataclass(frozen=True)
class FlowDims:
    """causal_masked_diff_xvec (R/s3gen.R:771-803) + upsample_conformer_encoder_full (R/conformer.R:467-510)."""
    vocab: int = 6561
    d: int = 512
    heads: int = 8
    ffn: int = 2048
    n_blocks: int = 6
    n_up: int = 4
    spk: int = 192
    cfm: CfmDims = CFM_FULL


FLOW_FULL = FlowDims()
FLOW_TINY = FlowDims(vocab=300, d=128, heads=2, ffn=256, n_blocks=2, n_up=1, spk=64, cfm=CFM_TINY)


def synth_flow_weights(dims: FlowDims = FLOW_FULL, seed: int = 555) -> Dict[str, np.ndarray]:
    """A synthetic s3gen checkpoint restricted to the flow module, keys as load_s3gen_weights / load_conformer_encoder_weights
    read them (R/s3gen.R:1215-1231, R/conformer.R:627-713), the CFM estimator's tensors included."""
    rng = np.random.Generator(np.random.PCG64(seed))
    d, H, mel = dims.d, dims.heads, dims.cfm.mel
    w: Dict[str, np.ndarray] = {}

    def lin(name, o, i, bias=True, gain=0.7):
        w[name + ".weight"] = (rng.standard_normal((o, i), dtype=F32) * F32(gain / np.sqrt(i))).astype(F32)
        if bias:
            w[name + ".bias"] = (rng.standard_normal(o, dtype=F32) * F32(0.05)).astype(F32)

    def conv(name, o, i, k, gain=0.7):
        w[name + ".weight"] = (rng.standard_normal((o, i, k), dtype=F32) * F32(gain / np.sqrt(i * k))).astype(F32)
        w[name + ".bias"] = (rng.standard_normal(o, dtype=F32) * F32(0.05)).astype(F32)

    def norm(name, n):
        w[name + ".weight"] = (F32(1) + rng.standard_normal(n, dtype=F32) * F32(0.1)).astype(F32)
        w[name + ".bias"] = (rng.standard_normal(n, dtype=F32) * F32(0.05)).astype(F32)

    def layer(p):
        for nm in ("linear_q", "linear_k", "linear_v"):
            lin(p + "self_attn." + nm, d, d, gain=1.0)
        lin(p + "self_attn.linear_out", d, d, gain=0.4)
        lin(p + "self_attn.linear_pos", d, d, bias=False, gain=1.0)
        w[p + "self_attn.pos_bias_u"] = (rng.standard_normal((H, d // H), dtype=F32) * F32(0.3)).astype(F32)
        w[p + "self_attn.pos_bias_v"] = (rng.standard_normal((H, d // H), dtype=F32) * F32(0.3)).astype(F32)
        lin(p + "feed_forward.w_1", dims.ffn, d)
        lin(p + "feed_forward.w_2", d, dims.ffn, gain=0.4)
        norm(p + "norm_mha", d)
        norm(p + "norm_ff", d)

    w["flow.input_embedding.weight"] = rng.standard_normal((dims.vocab, d), dtype=F32)
    lin("flow.spk_embed_affine_layer", mel, dims.spk, gain=1.0)
    lin("flow.encoder_proj", mel, d, gain=1.0)
    e = "flow.encoder."
    lin(e + "embed.out.0", d, d)
    norm(e + "embed.out.1", d)
    conv(e + "pre_lookahead_layer.conv1", d, d, 4)
    conv(e + "pre_lookahead_layer.conv2", d, d, 3)
    for i in range(dims.n_blocks):
        layer(e + f"encoders.{i}.")
    conv(e + "up_layer.conv", d, d, 5)
    lin(e + "up_embed.out.0", d, d)
    norm(e + "up_embed.out.1", d)
    for i in range(dims.n_up):
        layer(e + f"up_encoders.{i}.")
    norm(e + "after_norm", d)
    w.update(synth_cfm_weights(dims.cfm, seed + 222))
    return w


def ensure_flow_checkpoint(dims: FlowDims = FLOW_FULL, seed: int = 555, root: str | None = None) -> str:
    root = root or os.environ.get("T3MI_SYNTH_DIR", "/tmp/t3mi_synth")
    os.makedirs(root, exist_ok=True)
    path = os.path.join(root, f"s3gen_flow_d{dims.d}_L{dims.n_blocks}+{dims.n_up}_V{dims.vocab}_C{dims.cfm.hidden}_s{seed}.safetensors")
    if not os.path.exists(path):
        write_safetensors(synth_flow_weights(dims, seed), path)
    return path


def synth_flow_inputs(dims: FlowDims, n_tokens: int, n_prompt: int, seed: int):
    """speech tokens, prompt tokens, prompt mel [2 * n_prompt, mel], x-vector [spk], start noise (1, mel, 2 * (n_prompt + n_tokens))."""
    rng = np.random.Generator(np.random.PCG64(seed))
    mel = dims.cfm.mel
    tok = rng.integers(0, dims.vocab, n_tokens)
    ptok = rng.integers(0, dims.vocab, n_prompt)
    pfeat = rng.standard_normal((2 * n_prompt, mel), dtype=F32)
    xvec = rng.standard_normal(dims.spk, dtype=F32)
    z = rng.standard_normal((1, mel, 2 * (n_prompt + n_tokens)), dtype=F32)
    return tok, ptok, pfeat, xvec, z


# ---- HiFT vocoder (third piece) -----------------------------------------------------------------------

@dataclass(frozen=True)
class HiftDims:
    """hift_generator as create_s3gen_vocoder configures it (R/hifigan.R:788-817); base_channels reducible for tests."""
    mel: int = 80
    base: int = 512
    f0_ch: int = 512
    n_harm: int = 9              # nb_harmonics 8 + the fundamental
    n_fft: int = 16


HIFT_FULL = HiftDims()
HIFT_TINY = HiftDims(mel=16, base=64, f0_ch=64)
HIFT_UPS = ((8, 16), (5, 11), (3, 7))
HIFT_RES_K = (3, 7, 11)
HIFT_SRC_K = (7, 7, 11)


def synth_hift_weights(dims: HiftDims = HIFT_FULL, seed: int = 999, prefix: str = "mel2wav.") -> Dict[str, np.ndarray]:
    """A synthetic vocoder checkpoint under the keys load_hifigan_weights reads (R/hifigan.R:640-777): weight-normalised
    convolutions as `parametrizations.weight.original0 / original1` pairs (g [out, 1, 1], v [out, in, k]; conv-transpose:
    g [in, 1, 1], v [in, out, k]), plain tensors for source_downs, the linears and the Snake alphas."""
    rng = np.random.Generator(np.random.PCG64(seed))
    w: Dict[str, np.ndarray] = {}

    def pconv(name, a, b, k, gain=0.8):       # v [a, b, k]; effective row norm = g
        v = rng.standard_normal((a, b, k), dtype=F32)
        g = (F32(gain) * (F32(1) + F32(0.2) * rng.standard_normal((a, 1, 1), dtype=F32))).astype(F32)
        w[prefix + name + ".parametrizations.weight.original0"] = np.abs(g).astype(F32)
        w[prefix + name + ".parametrizations.weight.original1"] = v
        return name

    def bias(name, n, std=0.05):
        w[prefix + name + ".bias"] = (rng.standard_normal(n, dtype=F32) * F32(std)).astype(F32)

    c = dims.f0_ch
    for i in range(5):
        pconv(f"f0_predictor.condnet.{2 * i}", c, dims.mel if i == 0 else c, 3)
        bias(f"f0_predictor.condnet.{2 * i}", c)
    w[prefix + "f0_predictor.classifier.weight"] = (rng.standard_normal((1, c), dtype=F32) * F32(160.0 / np.sqrt(c))).astype(F32)
    w[prefix + "f0_predictor.classifier.bias"] = np.array([25.0], dtype=F32)       # f0 around 0..200 Hz: voiced and unvoiced frames
    w[prefix + "m_source.l_linear.weight"] = (rng.standard_normal((1, dims.n_harm), dtype=F32) * F32(2.0)).astype(F32)
    w[prefix + "m_source.l_linear.bias"] = np.array([0.01], dtype=F32)
    pconv("conv_pre", dims.base, dims.mel, 7)
    bias("conv_pre", dims.base)

    def resblock(p, ch, k):
        for j in range(3):
            pconv(f"{p}.convs1.{j}", ch, ch, k, gain=0.5)
            bias(f"{p}.convs1.{j}", ch)
            pconv(f"{p}.convs2.{j}", ch, ch, k, gain=0.5)
            bias(f"{p}.convs2.{j}", ch)
            w[prefix + f"{p}.activations1.{j}.alpha"] = (F32(1) + F32(0.3) * rng.standard_normal(ch, dtype=F32)).astype(F32)
            w[prefix + f"{p}.activations2.{j}.alpha"] = (F32(1) + F32(0.3) * rng.standard_normal(ch, dtype=F32)).astype(F32)

    total = int(np.prod([u for u, _ in HIFT_UPS]))
    cum = 1
    for i, (u, k) in enumerate(HIFT_UPS):
        cin, cout = dims.base >> i, dims.base >> (i + 1)
        pconv(f"ups.{i}", cin, cout, k, gain=1.5)          # conv-transpose layout [in, out, k]
        bias(f"ups.{i}", cout)
        cum *= u
        du = total // cum
        kd = 1 if du == 1 else 2 * du
        w[prefix + f"source_downs.{i}.weight"] = (rng.standard_normal((cout, dims.n_fft + 2, kd), dtype=F32) *
                                                  F32(0.5 / np.sqrt((dims.n_fft + 2) * kd))).astype(F32)
        bias(f"source_downs.{i}", cout)
        resblock(f"source_resblocks.{i}", cout, HIFT_SRC_K[i])
        for j, rk in enumerate(HIFT_RES_K):
            resblock(f"resblocks.{i * 3 + j}", cout, rk)
    pconv("conv_post", dims.n_fft + 2, dims.base >> 3, 7, gain=0.6)
    bias("conv_post", dims.n_fft + 2)
    return w


def ensure_hift_checkpoint(dims: HiftDims = HIFT_FULL, seed: int = 999, root: str | None = None) -> str:
    root = root or os.environ.get("T3MI_SYNTH_DIR", "/tmp/t3mi_synth")
    os.makedirs(root, exist_ok=True)
    path = os.path.join(root, f"s3gen_hift_b{dims.base}_f{dims.f0_ch}_mel{dims.mel}_s{seed}.safetensors")
    if not os.path.exists(path):
        write_safetensors(synth_hift_weights(dims, seed), path)
    return path


def synth_hift_inputs(dims: HiftDims, T: int, seed: int):
    """mel [mel, T], harmonic start phases [n_harm] in (-pi, pi), source noise [n_harm, T * 480] (standard normal)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    mel = rng.standard_normal((dims.mel, T), dtype=F32)
    phase = rng.uniform(-np.pi, np.pi, dims.n_harm).astype(F32)
    noise = rng.standard_normal((dims.n_harm, T * 480), dtype=F32)
    return mel, phase, noise
