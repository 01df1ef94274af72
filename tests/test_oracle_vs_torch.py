"""Pin the numpy oracle to CPU libtorch -- the library the reference's arithmetic
actually lives in (R package `torch`, DESCRIPTION:19) -- by running the same op
sequence (linear, scaled_dot_product_attention, softmax, sort, cumsum) through
python `torch` on CPU.  CPU only; skipped when torch is unavailable."""
import math

import numpy as np
import pytest

from chatterbox_amd import synth
from oracle import t3_oracle as O

torch = pytest.importorskip("torch")
Fn = torch.nn.functional


def _cfg(dims):
    return O.LlamaConfig(hidden_size=dims.hidden_size, intermediate_size=dims.intermediate_size,
                         num_hidden_layers=dims.num_hidden_layers,
                         num_attention_heads=dims.num_attention_heads, head_dim=dims.head_dim,
                         speech_vocab=dims.speech_vocab, max_speech_pos=dims.max_speech_pos,
                         start_speech_token=dims.start_speech_token,
                         stop_speech_token=dims.stop_speech_token)


def torch_forward(w, cfg, x, past):
    """R/llama.R:422-546 with torch ops (growing KV by cat, additive -65504 mask)."""
    tw = {k: torch.from_numpy(v) for k, v in w.items()}
    x = torch.from_numpy(x)
    R, S, d = x.shape
    H, Dh = cfg.num_attention_heads, cfg.head_dim
    pos0 = 0 if past is None else past[0][0].shape[2]
    cos, sin = O.rope_tables(cfg, pos0 + S)
    cos = torch.from_numpy(cos[pos0:pos0 + S])[None, None]
    sin = torch.from_numpy(sin[pos0:pos0 + S])[None, None]
    mask = None
    if S > 1:
        mask = torch.triu(torch.full((S, S), -65504.0), diagonal=1)[None, None]

    def rms(t, wt):
        v = t.pow(2).mean(-1, keepdim=True)
        return wt * (t * torch.rsqrt(v + cfg.rms_norm_eps))

    def rot(t):
        h = t.shape[-1] // 2
        return torch.cat([-t[..., h:], t[..., :h]], -1)

    new = []
    for i in range(cfg.num_hidden_layers):
        p = f"tfmr.layers.{i}."
        h = rms(x, tw[p + "input_layernorm.weight"])
        q = Fn.linear(h, tw[p + "self_attn.q_proj.weight"]).view(R, S, H, Dh).transpose(1, 2)
        k = Fn.linear(h, tw[p + "self_attn.k_proj.weight"]).view(R, S, H, Dh).transpose(1, 2)
        v = Fn.linear(h, tw[p + "self_attn.v_proj.weight"]).view(R, S, H, Dh).transpose(1, 2)
        q = q * cos + rot(q) * sin
        k = k * cos + rot(k) * sin
        if past is not None:
            k = torch.cat([past[i][0], k], 2)
            v = torch.cat([past[i][1], v], 2)
        new.append((k, v))
        a = Fn.scaled_dot_product_attention(q, k, v, attn_mask=mask)
        a = a.transpose(1, 2).contiguous().view(R, S, d)
        x = x + Fn.linear(a, tw[p + "self_attn.o_proj.weight"])
        h = rms(x, tw[p + "post_attention_layernorm.weight"])
        g = Fn.linear(h, tw[p + "mlp.gate_proj.weight"])
        u = Fn.linear(h, tw[p + "mlp.up_proj.weight"])
        x = x + Fn.linear(g * torch.sigmoid(g) * u, tw[p + "mlp.down_proj.weight"])
    x = rms(x, tw["tfmr.norm.weight"])
    return x, new


def test_model_forward_matches_torch():
    dims = synth.TINY
    cfg = _cfg(dims)
    w = synth.synth_llama_weights(dims, 7)
    emb = synth.synth_embeds(2, 9, dims.hidden_size, 11)
    m = O.LlamaOracle(w, cfg)
    h_o, past_o = m.forward(emb)
    with torch.no_grad():
        h_t, past_t = torch_forward(w, cfg, emb, None)
    np.testing.assert_allclose(h_o, h_t.numpy(), rtol=0, atol=2e-5)
    # one decode step on top of the prefill cache
    nxt = synth.synth_embeds(2, 1, dims.hidden_size, 12)
    h_o2, _ = m.forward(nxt, past_o)
    with torch.no_grad():
        h_t2, _ = torch_forward(w, cfg, nxt, past_t)
    np.testing.assert_allclose(h_o2, h_t2.numpy(), rtol=0, atol=2e-5)
    lo = m.speech_logits(h_o2[:, -1])
    lt = Fn.linear(h_t2[:, -1], torch.from_numpy(w["speech_head.weight"])).numpy()
    np.testing.assert_allclose(lo, lt, rtol=0, atol=2e-5)


@pytest.mark.parametrize("variant", ["r", "cpp"])
@pytest.mark.parametrize("top_p,min_p", [(0.95, 0.05), (0.5, 0.0), (0.0, 0.05), (1.0, 0.0)])
def test_sampling_tail_matches_torch(variant, top_p, min_p):
    rng = np.random.default_rng(3)
    lg = (rng.standard_normal(8194) * 3).astype(np.float32)
    seen = [5, 17, 6560]
    p = O.SampleParams(top_p=top_p, min_p=min_p, variant=variant)
    t = O.sampling_tail(lg, seen, p, 0.0)
    # torch restatement of R/t3.R:529-563 / src/t3_decode.cpp:318-357 (0-indexed)
    x = torch.from_numpy(lg.copy())
    ids = torch.tensor(sorted(set(seen)))
    x[ids] = x[ids] / p.repetition_penalty
    x = x / p.temperature
    pr = torch.softmax(x, -1)
    thr = p.min_p * pr.max()
    pr = torch.where(pr < thr, torch.zeros_like(pr), pr)
    pr = pr / pr.sum()
    sp, si = torch.sort(pr, descending=True, stable=True)
    cs = torch.cumsum(sp, -1)
    mask = cs > p.top_p
    if variant == "r":
        mask[0] = False
    else:
        mask = torch.cat([torch.zeros(1, dtype=torch.bool), mask[:-1]])
    sp = torch.where(mask, torch.zeros_like(sp), sp)
    sp = sp / sp.sum()
    nz = int((sp > 0).sum())
    assert nz == len(t.sorted_idx)
    assert np.array_equal(si[:nz].numpy(), t.sorted_idx)
    np.testing.assert_allclose(sp[:nz].numpy(), t.sorted_probs, rtol=2e-6, atol=1e-9)


def test_turbo_forward_and_tail_match_torch():
    """GPT-2 blocks (LayerNorm, fused c_attn + bias, GELU-tanh) and the turbo sampler vs libtorch ops."""
    dims = synth.TURBO_TINY
    w = synth.synth_turbo_weights(dims, 9)
    cfg = O.TurboConfig(hidden_size=dims.hidden_size, n_head=dims.n_head, n_layer=dims.n_layer,
                        speech_vocab=dims.speech_vocab, start_speech_token=dims.start_speech_token,
                        stop_speech_token=dims.stop_speech_token)
    m = O.GPT2Oracle(w, cfg)
    emb = synth.synth_embeds(1, 7, dims.hidden_size, 3)
    tw = {k: torch.from_numpy(v) for k, v in w.items()}

    def tforward(x, past):
        x = torch.from_numpy(x)
        R, S, d = x.shape
        H = cfg.n_head
        new = []
        for i in range(cfg.n_layer):
            p = f"tfmr.h.{i}."
            h = Fn.layer_norm(x, (d,), tw[p + "ln_1.weight"], tw[p + "ln_1.bias"], 1e-5)
            q, k, v = Fn.linear(h, tw[p + "attn.c_attn.weight"], tw[p + "attn.c_attn.bias"]).split(d, dim=2)
            q, k, v = [t.view(R, S, H, d // H).transpose(1, 2) for t in (q, k, v)]
            if past is not None:
                k = torch.cat([past[i][0], k], 2)
                v = torch.cat([past[i][1], v], 2)
            new.append((k, v))
            a = Fn.scaled_dot_product_attention(q, k, v, is_causal=(S > 1 and past is None))
            a = a.transpose(1, 2).contiguous().view(R, S, d)
            x = x + Fn.linear(a, tw[p + "attn.c_proj.weight"], tw[p + "attn.c_proj.bias"])
            h = Fn.layer_norm(x, (d,), tw[p + "ln_2.weight"], tw[p + "ln_2.bias"], 1e-5)
            h = Fn.gelu(Fn.linear(h, tw[p + "mlp.c_fc.weight"], tw[p + "mlp.c_fc.bias"]), approximate="tanh")
            x = x + Fn.linear(h, tw[p + "mlp.c_proj.weight"], tw[p + "mlp.c_proj.bias"])
        return Fn.layer_norm(x, (d,), tw["tfmr.ln_f.weight"], tw["tfmr.ln_f.bias"], 1e-5), new

    h_o, past_o = m.forward(emb)
    with torch.no_grad():
        h_t, past_t = tforward(emb, None)
    np.testing.assert_allclose(h_o, h_t.numpy(), rtol=0, atol=3e-5)
    nxt = synth.synth_embeds(1, 1, dims.hidden_size, 4)
    h_o2, _ = m.forward(nxt, past_o)
    with torch.no_grad():
        h_t2, _ = tforward(nxt, past_t)
    np.testing.assert_allclose(h_o2, h_t2.numpy(), rtol=0, atol=3e-5)

    # sampler (R/t3.R:1688-1742) on wide logits
    rng = np.random.default_rng(5)
    lg = (rng.standard_normal(dims.speech_vocab) * 3).astype(np.float32)
    for top_k, top_p in [(50, 0.95), (1000, 0.8), (0, 1.0), (20, 1.0)]:
        p = O.TurboParams(top_k=top_k, top_p=top_p)
        t = O.turbo_sampling_tail(lg, [3, 11], p, 0.0)
        x = torch.from_numpy(lg.copy())
        ids = torch.tensor([3, 11])
        sel = x[ids]
        x[ids] = torch.where(sel > 0, sel / 1.2, sel * 1.2)
        x = x / 0.8
        if 0 < top_k < x.numel():
            x[x < torch.topk(x, top_k)[0].min()] = -float("inf")
        if top_p < 1.0:
            sl, si = torch.sort(x, descending=True, stable=True)
            cp = torch.cumsum(torch.softmax(sl, -1), -1)
            msk = cp > top_p
            msk[0] = False
            rem = torch.zeros_like(msk).scatter(0, si, msk)
            x[rem] = -float("inf")
        pr = torch.softmax(x, -1)
        sp, si = torch.sort(pr, descending=True, stable=True)
        nz = int((pr > 0).sum()) if top_p < 1.0 or top_k > 0 else x.numel()
        nz = min(nz, int(torch.isfinite(x).sum()))
        assert nz == len(t.sorted_idx)
        assert np.array_equal(si[:nz].numpy(), t.sorted_idx)
        np.testing.assert_allclose(sp[:nz].numpy(), t.sorted_probs, rtol=3e-6, atol=1e-9)
