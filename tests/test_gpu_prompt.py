"""GPU parity of the prompt builder through the C ABI: prepare_input_embeds on device against
the oracle's restatement of R/t3.R:361-403 / 1447-1477, and t3_inference(model, cond, text_tokens) end to end.
fp32 bar: 1e-4 of the largest magnitude (the reference's own bar for this stage, scripts/test_t3_cond.R:191)."""
import numpy as np
import pytest

import chatterbox_amd as cb
from chatterbox_amd import _lib, synth
from oracle import t3_oracle as O

from helpers import oracle_cfg, turbo_cfg

pytestmark = pytest.mark.gpu


def _inputs(dims, n_cond, n_text, seed):
    rng = np.random.default_rng(seed)
    spk = (rng.standard_normal(dims.speaker_embed) * 0.1).astype(np.float32)
    cond = rng.integers(0, dims.start_speech_token, n_cond)
    text = rng.integers(1, min(dims.text_vocab, 255), n_text)
    return spk, cond, text


def _close(got, want, tol=1e-4):
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= tol * np.abs(want).max(), np.abs(got - want).max()


@pytest.fixture(scope="module")
def tiny():
    m = cb.load_t3(synth.ensure_checkpoint(synth.TINY), start_speech_token=synth.TINY.start_speech_token,
                   stop_speech_token=synth.TINY.stop_speech_token)
    yield m, synth.synth_llama_weights(synth.TINY)
    m.close()


@pytest.mark.parametrize("n_cond,n_text,cfg,emotion", [(150, 6, 0.5, 0.5), (150, 17, 0.0, 0.25), (37, 1, 0.5, None),
                                                       (0, 9, 0.5, 0.9), (300, 40, 0.5, 0.5)])
def test_prepare_input_embeds_matches_oracle(tiny, n_cond, n_text, cfg, emotion):
    m, w = tiny
    spk, cond, text = _inputs(synth.TINY, n_cond, n_text, 11 + n_cond + n_text)
    want, lc = O.prepare_input_embeds(w, spk, cond if n_cond else None, emotion, text, cfg, synth.TINY.start_speech_token)
    got, lc2 = cb.prepare_input_embeds(m, cb.t3_cond(spk, cond if n_cond else None, emotion_adv=emotion), text, cfg)
    assert lc2 == lc
    _close(got, want)
    # exact pieces: gathers and the zeroed unconditional text span involve no arithmetic beyond one add
    assert np.array_equal(got[:, lc:], want[:, lc:])
    if cfg > 0:
        assert np.array_equal(got[0, :lc], got[1, :lc]) and np.all(got[1, lc:-1] == 0)


def test_t3_inference_from_tokens_equals_the_embeds_path(tiny):
    """t3_inference(model, cond, text_tokens) == t3_inference(model, prepare_input_embeds(...)) == oracle, greedy."""
    m, w = tiny
    spk, cond, text = _inputs(synth.TINY, 150, 8, 21)
    c = cb.t3_cond(spk, cond, emotion_adv=0.5)
    a = cb.t3_inference(m, c, text, max_new_tokens=24, top_p=0.0)
    e, _ = cb.prepare_input_embeds(m, c, text, 0.5)
    b = cb.t3_inference(m, e, max_new_tokens=24, top_p=0.0)
    assert np.array_equal(a, b) and a.size > 0
    om = O.LlamaOracle(w, oracle_cfg(synth.TINY))
    oe, _ = O.prepare_input_embeds(w, spk, cond, 0.5, text, 0.5, synth.TINY.start_speech_token)
    want = O.t3_inference(om, oe, O.SampleParams(max_new_tokens=24, top_p=0.0))
    assert np.array_equal(a, np.asarray(want.tokens if hasattr(want, "tokens") else want))
    # the cpp-backend form and the batch form take the same conditioning
    cpp = cb.t3_inference_cpp(m, c, text, max_new_tokens=24, top_p=0.0)
    assert cpp.size <= 24
    spk2, cond2, text2 = _inputs(synth.TINY, 150, 15, 22)
    outs = cb.t3_inference_batch(m, [c, cb.t3_cond(spk2, cond2)], [text, text2], max_new_tokens=24, top_p=0.0)
    assert np.array_equal(outs[0], a)
    assert np.array_equal(outs[1], cb.t3_inference(m, cb.t3_cond(spk2, cond2), text2, max_new_tokens=24, top_p=0.0))


def test_turbo_prompt_and_inference_from_tokens():
    dims = synth.TURBO_TINY
    w = synth.synth_turbo_weights(dims)
    m = cb.load_t3_turbo(synth.ensure_turbo_checkpoint(dims), start_speech_token=dims.start_speech_token,
                         stop_speech_token=dims.stop_speech_token)
    try:
        rng = np.random.default_rng(9)
        spk = (rng.standard_normal(dims.speaker_embed) * 0.1).astype(np.float32)
        cond = rng.integers(0, dims.start_speech_token, 375)
        text = rng.integers(0, dims.text_vocab, 30)
        want, lc = O.prepare_input_embeds_turbo(w, spk, cond, text, dims.start_speech_token)
        got, lc2 = cb.prepare_input_embeds(m, cb.t3_cond(spk, cond), text, 0.0)
        assert lc2 == lc == 376 and got.shape == (1, 407, dims.hidden_size)
        _close(got, want)
        assert np.array_equal(got[0, 1:], want[0, 1:])          # pure gathers
        a = cb.t3_inference_turbo(m, cb.t3_cond(spk, cond), text, max_new_tokens=12, top_k=1, top_p=1.0)
        b = cb.t3_inference_turbo(m, got, max_new_tokens=12, top_k=1, top_p=1.0)
        assert np.array_equal(a, b) and a.size > 0
    finally:
        m.close()


def test_prompt_error_paths(tiny, tmp_path):
    m, _ = tiny
    spk, cond, text = _inputs(synth.TINY, 150, 6, 1)
    with pytest.raises(_lib.T3miError, match="speaker_emb"):
        cb.prepare_input_embeds(m, cb.t3_cond(spk[:100], cond), text, 0.5)
    with pytest.raises(_lib.T3miError, match="cond_prompt_speech_tokens"):
        cb.prepare_input_embeds(m, cb.t3_cond(spk, np.array([5, synth.TINY.speech_vocab])), text, 0.5)
    with pytest.raises(_lib.T3miError, match="text_tokens"):
        cb.prepare_input_embeds(m, cb.t3_cond(spk, cond), [synth.TINY.text_vocab], 0.5)
    with pytest.raises(_lib.T3miError, match="longer than the learned text position table"):
        cb.prepare_input_embeds(m, cb.t3_cond(spk, cond), np.ones(synth.TINY.max_text_pos, dtype=np.int64), 0.5)
    # a decode-only checkpoint (no text_emb / cond_enc tensors) opens fine and says so when asked for a prompt
    w = {k: v for k, v in synth.synth_llama_weights(synth.TINY).items()
         if not k.startswith(("text_", "cond_enc."))}
    p = synth.write_safetensors(w, str(tmp_path / "decode_only.safetensors"))
    m2 = cb.load_t3(p, start_speech_token=synth.TINY.start_speech_token, stop_speech_token=synth.TINY.stop_speech_token)
    try:
        with pytest.raises(_lib.T3miError, match="no prompt tensors"):
            cb.prepare_input_embeds(m2, cb.t3_cond(spk, cond), text, 0.5)
    finally:
        m2.close()


def test_full_size_prompt_matches_oracle():
    """d = 1024, 150 conditioning tokens, 4 x 256 perceiver heads: the real T3 shapes (R/t3.R:11-36)."""
    dims = synth.T3Dims(num_hidden_layers=1)
    w = synth.synth_llama_weights(dims)
    m = cb.load_t3(synth.ensure_checkpoint(dims))
    try:
        spk, cond, text = _inputs(dims, 150, 15, 4)
        want, lc = O.prepare_input_embeds(w, spk, cond, 0.5, text, 0.5, dims.start_speech_token)
        got, lc2 = cb.prepare_input_embeds(m, cb.t3_cond(spk, cond), text, 0.5)
        assert lc == lc2 == 34 and got.shape == (2, 52, 1024)           # S0 = 52 for 15 text tokens
        _close(got, want)
    finally:
        m.close()


def test_batch_of_token_prompts_matches_single_calls(tiny):
    """Ten utterances with their own speaker vectors, prompt tokens and text lengths in ONE call (the MFMA
    many-row path: 20 rows) give what ten single calls give -- every utterance's staged inputs stay its own."""
    m, _ = tiny
    conds, texts = [], []
    for u in range(10):
        spk, cond, text = _inputs(synth.TINY, 150 if u % 3 else 80 + u, 3 + 2 * u, 100 + u)
        conds.append(cb.t3_cond(spk, cond, emotion_adv=0.1 * u))
        texts.append(text)
    outs = cb.t3_inference_batch(m, conds, texts, max_new_tokens=12, top_p=0.0, mask_stop=True)
    assert len(outs) == 10
    for u in range(10):
        e, _ = cb.prepare_input_embeds(m, conds[u], texts[u], 0.5)
        single = cb.Session(m, [e], max_new_tokens=12, top_p=0.0, mask_stop=True)
        single.decode(12)
        want = single.tokens(0)
        single.end()
        # the batch runs the linears on the MFMA path, the single call on the GEMV path: fp32 rounding may differ
        # in the last bits, greedy tokens of these well-separated distributions do not
        assert np.array_equal(outs[u], want), u
