"""CPU checks of the prompt-construction oracle: layout facts pinned by the reference's own
fixture (outputs/cond_enc_debug.safetensors -> tests/golden/ref_prompt_embeds.npz) and an op-for-op run of the
conditioning encoder on CPU libtorch.  Values against REAL weights are unpinned (weights absent)."""
import os

import numpy as np
import pytest

from chatterbox_amd import synth
from oracle import t3_oracle as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _inputs(dims, n_cond=150, n_text=6, seed=3):
    rng = np.random.default_rng(seed)
    spk = (rng.standard_normal(dims.speaker_embed) * 0.1).astype(np.float32)
    cond = rng.integers(0, dims.start_speech_token, n_cond)
    text = rng.integers(1, min(dims.text_vocab, 255), n_text)
    return spk, cond, text


def test_prompt_layout_matches_the_reference_fixture():
    """The fixture is the real prefill input of a 6-token sentence: [2, 43, 1024] = 34 conditioning rows
    (speaker 1 + perceiver 32 + emotion 1) + 8 text rows (start, 6, stop) + BOS; both rows share conditioning
    and BOS.  The oracle must build the same layout for the same token counts."""
    g = np.load(os.path.join(GOLD, "ref_prompt_embeds.npz"))
    ref, lc = g["input_embeds"], int(g["len_cond"][0])
    assert ref.shape == (2, 43, 1024) and lc == 34
    assert np.array_equal(ref[0, :lc], ref[1, :lc]) and np.array_equal(ref[0, -1], ref[1, -1])
    w = synth.synth_llama_weights(synth.TINY)
    spk, cond, text = _inputs(synth.TINY)
    e, lc2 = O.prepare_input_embeds(w, spk, cond, 0.5, text, 0.5, synth.TINY.start_speech_token)
    assert e.shape[:2] == ref.shape[:2] and lc2 == lc
    assert np.array_equal(e[0, :lc], e[1, :lc]) and np.array_equal(e[0, -1], e[1, -1])
    assert np.all(e[1, lc:-1] == 0) and np.abs(e[0, lc:-1]).min() > 0          # R/t3.R:383-386 (Q8)
    # pieces: text rows = text_emb + text_pos_emb, BOS = speech_emb[start] + speech_pos_emb[0]
    tt = np.concatenate([[255], text, [0]])
    assert np.array_equal(e[0, lc:-1], w["text_emb.weight"][tt] + w["text_pos_emb.emb.weight"][:8])
    assert np.array_equal(e[0, -1], w["speech_emb.weight"][synth.TINY.start_speech_token] + w["speech_pos_emb.emb.weight"][0])
    # without CFG: one row (R/t3.R:480-491)
    e1, _ = O.prepare_input_embeds(w, spk, cond, 0.5, text, 0.0, synth.TINY.start_speech_token)
    assert e1.shape == (1, 43, synth.TINY.hidden_size) and np.array_equal(e1[0], e[0])
    # no emotion / no prompt tokens shorten the conditioning (R/t3.R:296-311)
    assert O.prepare_input_embeds(w, spk, cond, None, text, 0.5, synth.TINY.start_speech_token)[1] == 33
    assert O.prepare_input_embeds(w, spk, None, 0.5, text, 0.5, synth.TINY.start_speech_token)[1] == 2


def test_turbo_prompt_layout():
    """R/t3.R:1447-1477: speaker row + raw prompt-token embeddings + raw text embeddings + BOS, one row."""
    dims = synth.TURBO_TINY
    w = synth.synth_turbo_weights(dims)
    rng = np.random.default_rng(5)
    spk = (rng.standard_normal(dims.speaker_embed) * 0.1).astype(np.float32)
    cond = rng.integers(0, dims.start_speech_token, 375)
    text = rng.integers(0, dims.text_vocab, 30)
    e, lc = O.prepare_input_embeds_turbo(w, spk, cond, text, dims.start_speech_token)
    assert e.shape == (1, 1 + 375 + 30 + 1, dims.hidden_size) and lc == 376          # SURVEY §8d cfg 5
    assert np.array_equal(e[0, 1:376], w["speech_emb.weight"][cond])
    assert np.array_equal(e[0, 376:406], w["text_emb.weight"][text])
    assert np.array_equal(e[0, -1], w["speech_emb.weight"][dims.start_speech_token])


def test_cond_encoder_matches_libtorch_op_for_op():
    """The same op sequence as R/t3.R:161-195, 227-241, 278-315 on CPU libtorch (python torch)."""
    torch = pytest.importorskip("torch")
    F = torch.nn.functional
    dims = synth.TINY
    w = synth.synth_llama_weights(dims)
    t = {k: torch.from_numpy(v) for k, v in w.items() if k.startswith(("cond_enc.", "speech_"))}
    spk, cond, _ = _inputs(dims)
    d, H = dims.hidden_size, 4

    def block(x1, x2):
        p = "cond_enc.perceiver.attn."
        n1 = F.layer_norm(x1, (d,), t[p + "norm.weight"], t[p + "norm.bias"])
        n2 = F.layer_norm(x2, (d,), t[p + "norm.weight"], t[p + "norm.bias"])
        q = F.linear(n1, t[p + "to_q.weight"], t[p + "to_q.bias"]).view(1, -1, H, d // H).transpose(1, 2)
        k = F.linear(n2, t[p + "to_k.weight"], t[p + "to_k.bias"]).view(1, -1, H, d // H).transpose(1, 2)
        v = F.linear(n2, t[p + "to_v.weight"], t[p + "to_v.bias"]).view(1, -1, H, d // H).transpose(1, 2)
        a = torch.softmax(torch.matmul(q, k.transpose(2, 3)) * (1.0 / np.sqrt(d // H)), dim=-1)
        o = torch.matmul(a, v).transpose(1, 2).contiguous().view(1, -1, d)
        return x1 + F.linear(o, t[p + "proj_out.weight"], t[p + "proj_out.bias"])

    with torch.no_grad():
        ids = torch.from_numpy(cond)
        emb = (t["speech_emb.weight"][ids] + t["speech_pos_emb.emb.weight"][: ids.numel()])[None]
        q0 = t["cond_enc.perceiver.pre_attention_query"]
        pre = block(q0, emb)
        perc = block(pre, pre)
        s = F.linear(torch.from_numpy(spk)[None], t["cond_enc.spkr_enc.weight"], t["cond_enc.spkr_enc.bias"])[:, None]
        emo = F.linear(torch.tensor(0.5).view(1, 1, 1), t["cond_enc.emotion_adv_fc.weight"])
        want = torch.cat([s, perc, emo], dim=1)[0].numpy()
    prompt = (w["speech_emb.weight"][cond] + w["speech_pos_emb.emb.weight"][: cond.size]).astype(np.float32)
    got = O.cond_enc(w, spk, prompt, 0.5, use_perceiver=True)
    assert got.shape == want.shape == (34, d)
    assert np.abs(got - want).max() <= 2e-5 * np.abs(want).max()


def test_text_start_stop_tokens_match_the_reference_fixture():
    """outputs/t3_reference.safetensors holds the text ids upstream fed to T3 for one sentence, WITH the start / stop
    text tokens: 255 first, 0 last (R/t3.R:14-15, 473-477) -- the constants the oracle and the library add."""
    g = np.load(os.path.join(GOLD, "ref_tokens.npz"))
    tt = g["text_tokens_with_sot_eot"].reshape(-1)
    assert int(tt[0]) == O.START_TEXT_TOKEN == 255 and int(tt[-1]) == O.STOP_TEXT_TOKEN == 0
    assert tt.size == 37 and np.all(tt[1:-1] < 704) and not np.any(tt[1:-1] == 255)
    # S0 of that sentence under the reference's layout: 34 conditioning rows + 37 text rows + BOS = 72, the shape of
    # initial_embeds in outputs/t3_debug.safetensors
    w = synth.synth_llama_weights(synth.TINY)
    spk, cond, _ = _inputs(synth.TINY)
    e, lc = O.prepare_input_embeds(w, spk, cond, 0.5, tt[1:-1] % synth.TINY.text_vocab, 0.5, synth.TINY.start_speech_token)
    assert e.shape[:2] == (2, 72) and lc == 34
