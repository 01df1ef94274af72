"""GPU parity of the Turbo T3 (GPT-2 backbone, R/t3.R:1190-1742) through the C ABI vs the CPU oracle.

Same bar as the Llama path: fp32 logits within 1e-3 relative, identical greedy tokens; sampled runs
compare the filtered distribution and the draw of each step under teacher forcing."""
import numpy as np
import pytest

import chatterbox_amd as cb
from chatterbox_amd import _lib, synth
from oracle import t3_oracle as O
from tests.helpers import assert_dist_close, draw_is_consistent, rel_err, turbo_oracle

pytestmark = pytest.mark.gpu

LOGIT_TOL = 1e-3


def _open(dims, seed, head_scale=1.0, dtype="f32"):
    path = synth.ensure_turbo_checkpoint(dims, seed, head_scale=head_scale)
    return cb.load_t3_turbo(path, device=0, dtype=dtype, start_speech_token=dims.start_speech_token,
                            stop_speech_token=dims.stop_speech_token)


@pytest.fixture(scope="module")
def tiny():
    dims = synth.TURBO_TINY
    model = _open(dims, 4321)
    yield dims, model, turbo_oracle(dims, 4321)
    model.close()


@pytest.fixture(scope="module")
def tiny_peaked():
    dims = synth.TURBO_TINY
    model = _open(dims, 77, head_scale=8.0)
    yield dims, model, turbo_oracle(dims, 77, head_scale=8.0)
    model.close()


def test_turbo_teacher_forced_logits_and_distributions(tiny):
    """Flat distributions: top-k = 100 of 900 via the radix select, then top-p over the survivors."""
    dims, model, om = tiny
    emb = synth.synth_embeds(1, 13, dims.hidden_size, 3)
    n = 20
    p = O.TurboParams(max_new_tokens=n, top_k=100, top_p=0.9, seed=7)
    ref = O.t3_inference_turbo(om, emb, p, mask_stop=True)
    assert len(ref.tokens) == n + 1                       # first token from the prefill + n more
    with cb.Session(model, [emb], max_new_tokens=n, top_k=100, top_p=0.9, seed=7, mask_stop=True,
                    forced=np.array(ref.tokens)) as s:
        for i in range(n + 1):
            assert rel_err(s.logits(), ref.raw_logits[i]) < LOGIT_TOL, f"step {i}"
            s.decode(1)
            ids, pr = s.dist(0)
            if assert_dist_close(ids, pr, ref.tails[i]):
                assert draw_is_consistent(int(s.drawn()[0]), ref.tails[i]), f"draw differs at step {i}"
        assert list(s.tokens(0)) == ref.tokens
        ntok, done = s.status()
        assert ntok[0] == n + 1 and done[0] == 2


@pytest.mark.parametrize("top_k,top_p", [(1000, 0.95), (0, 1.0), (5, 1.0), (50, 0.5)])
def test_turbo_sampling_tail_rules(tiny_peaked, top_k, top_p):
    """Reference defaults (top_k >= V: no top-k), no filter at all, top-k only, both filters."""
    dims, model, om = tiny_peaked
    emb = synth.synth_embeds(1, 9, dims.hidden_size, 5)
    n = 24
    p = O.TurboParams(max_new_tokens=n, top_k=top_k, top_p=top_p, seed=13)
    ref = O.t3_inference_turbo(om, emb, p, mask_stop=True)
    with cb.Session(model, [emb], max_new_tokens=n, top_k=top_k, top_p=top_p, seed=13, mask_stop=True,
                    forced=np.array(ref.tokens)) as s:
        for i in range(n + 1):
            s.decode(1)
            ids, pr = s.dist(0)
            if assert_dist_close(ids, pr, ref.tails[i]):
                assert draw_is_consistent(int(s.drawn()[0]), ref.tails[i]), f"draw differs at step {i}"


def test_turbo_step0_penalises_column_before_bos(tiny_peaked):
    """Q17: the first sample penalises 0-based column BOS-1 only; later steps the generated ids."""
    dims, model, om = tiny_peaked
    emb = synth.synth_embeds(1, 6, dims.hidden_size, 8)
    p = O.TurboParams(max_new_tokens=3, top_k=0, top_p=1.0, repetition_penalty=3.0, seed=2)
    ref = O.t3_inference_turbo(om, emb, p, mask_stop=True)
    with cb.Session(model, [emb], max_new_tokens=3, top_k=0, top_p=1.0, repetition_penalty=3.0, seed=2,
                    mask_stop=True, forced=np.array(ref.tokens)) as s:
        for i in range(4):
            s.decode(1)
            ids, pr = s.dist(0)
            assert_dist_close(ids, pr, ref.tails[i])
            col = dims.start_speech_token - 1 if i == 0 else ref.tokens[0]
            po = dict(zip([int(x) for x in ref.tails[i].sorted_idx], ref.tails[i].sorted_probs))
            pg = dict(zip([int(x) for x in ids], pr))
            assert abs(pg[col] - po[col]) <= 2e-3 * po[col] + 1e-7


def test_turbo_greedy_generate_matches_oracle(tiny_peaked):
    """top_p = 0 keeps only the first sorted entry (R/t3.R:1732-1733): deterministic; tokens identical.
    Runs until the stop token or max_new_tokens + 1 tokens; the wrapper strips a trailing stop token."""
    dims, model, om = tiny_peaked
    for seed in (1, 2):
        emb = synth.synth_embeds(1, 11, dims.hidden_size, 40 + seed)
        p = O.TurboParams(max_new_tokens=30, top_p=0.0)
        ref = O.t3_inference_turbo(om, emb, p, keep_logits=False)
        want = ref.tokens[:-1] if ref.hit_eos else ref.tokens
        got = cb.t3_inference_turbo(model, emb, max_new_tokens=30, top_p=0.0)
        assert list(got) == want


def test_turbo_batch_rows_are_independent(tiny_peaked):
    """Ragged prompts decoded together (GEMV path <= 4 rows, MFMA path above) equal the single runs."""
    dims, model, om = tiny_peaked
    embs = [synth.synth_embeds(1, 5 + 3 * u, dims.hidden_size, 60 + u) for u in range(6)]
    single = [cb.t3_inference_turbo(model, e, max_new_tokens=16, top_p=0.0) for e in embs[:3]]
    got3 = cb.t3_inference_turbo_batch(model, embs[:3], max_new_tokens=16, top_p=0.0)
    for a, b in zip(single, got3):
        assert list(a) == list(b)
    got6 = cb.t3_inference_turbo_batch(model, embs, max_new_tokens=16, top_p=0.0)
    for u, e in enumerate(embs):
        ref = O.t3_inference_turbo(om, e, O.TurboParams(max_new_tokens=16, top_p=0.0), keep_logits=False)
        want = ref.tokens[:-1] if ref.hit_eos else ref.tokens
        assert list(got6[u]) == want, f"utterance {u}"


def test_turbo_error_paths(tiny):
    dims, model, _ = tiny
    emb = synth.synth_embeds(1, 4, dims.hidden_size, 1)
    with pytest.raises(Exception):
        cb.t3_inference(model, emb, max_new_tokens=4, cfg_weight=0.0)     # Llama sampler on a Turbo model
    with pytest.raises(Exception):
        cb.load_t3(synth.ensure_turbo_checkpoint(dims, 4321))             # Turbo file opened as Llama


def test_turbo_full_size_logits_fp32_and_bf16():
    """GPT-2 medium shapes (24 x 1024, V = 6563): prefill + 3 teacher-forced steps against the oracle;
    bf16 weights within the looser bf16 bound."""
    dims = synth.TURBO_FULL
    om = turbo_oracle(dims, 4321)
    emb = synth.synth_embeds(1, 12, dims.hidden_size, 17)
    p = O.TurboParams(max_new_tokens=3, top_p=0.0)
    ref = O.t3_inference_turbo(om, emb, p)
    for dtype, tol in (("f32", LOGIT_TOL), ("bf16", 3e-2)):
        model = _open(dims, 4321, dtype=dtype)
        try:
            with cb.Session(model, [emb], max_new_tokens=3, top_p=0.0, forced=np.array(ref.tokens)) as s:
                for i in range(len(ref.tokens)):
                    assert rel_err(s.logits(), ref.raw_logits[i]) < tol, f"{dtype} step {i}"
                    s.decode(1)
                if dtype == "f32":
                    assert list(s.tokens(0)) == ref.tokens
        finally:
            model.close()


@pytest.mark.parametrize("dtype,tol", [("f32", LOGIT_TOL), ("bf16", 3e-2)])
def test_turbo_70_rows_128_row_gemm_tiles(dtype, tol):
    """70 sentences, one row each (no CFG): the GPT-2 linears on the 128-row MFMA tiles -- LayerNorm rows, biases in
    the epilogues, GELU, no RoPE; bf16: bf16 activation rows and KV cache.  Teacher-forced against the oracle."""
    dims = synth.TURBO_TINY
    om = turbo_oracle(dims, 4321)
    model = _open(dims, 4321, dtype=dtype)
    try:
        U, n = 70, 4
        embs = [synth.synth_embeds(1, 3 + (5 * u) % 19, dims.hidden_size, 700 + u) for u in range(U)]
        refs = [O.t3_inference_turbo(om, e, O.TurboParams(max_new_tokens=n, top_p=0.0), mask_stop=True) for e in embs]
        forced = np.array([r.tokens for r in refs])
        with cb.Session(model, embs, max_new_tokens=n, top_p=0.0, mask_stop=True, forced=forced) as s:
            for i in range(n + 1):
                lg = s.logits()
                for u in range(U):
                    assert rel_err(lg[u:u + 1], refs[u].raw_logits[i]) < tol, f"{dtype} sentence {u} step {i}"
                s.decode(1)
    finally:
        model.close()


@pytest.mark.parametrize("U,n", [(300, 3), (257, 1)])
def test_turbo_300_rows_register_blocked_gemm(U, n):
    """300 (257) sentences, bf16 weights: the GPT-2 linears on gemm_wide_kernel (biases, GELU, LayerNorm rows, K-split
    c_proj, a partly filled second row chunk -- one single row of it at 257).  Teacher-forced against the oracle
    within the bf16 bar."""
    dims = synth.TURBO_TINY
    om = turbo_oracle(dims, 4321)
    model = _open(dims, 4321, dtype="bf16")
    try:
        embs = [synth.synth_embeds(1, 3 + (5 * u) % 19, dims.hidden_size, 1700 + u) for u in range(U)]
        refs = [O.t3_inference_turbo(om, e, O.TurboParams(max_new_tokens=n, top_p=0.0), mask_stop=True) for e in embs]
        forced = np.array([r.tokens for r in refs])
        with cb.Session(model, embs, max_new_tokens=n, top_p=0.0, mask_stop=True, forced=forced) as s:
            for i in range(n + 1):
                lg = s.logits()
                for u in range(U):
                    assert rel_err(lg[u:u + 1], refs[u].raw_logits[i]) < 3e-2, f"sentence {u} step {i}"
                s.decode(1)
    finally:
        model.close()


def test_turbo_bf16_8_sentences_one_staging_pass():
    """cfg 5's batched shape at reduced size: 8 sentences (8 rows), bf16 weights -- the one-staging-pass form of the
    128-row GEMM with biases, GELU and LayerNorm rows.  Teacher-forced against the oracle within the bf16 bar."""
    dims = synth.TURBO_TINY
    om = turbo_oracle(dims, 4321)
    model = _open(dims, 4321, dtype="bf16")
    try:
        U, n = 8, 5
        embs = [synth.synth_embeds(1, 6 + 3 * u, dims.hidden_size, 2700 + u) for u in range(U)]
        refs = [O.t3_inference_turbo(om, e, O.TurboParams(max_new_tokens=n, top_p=0.0), mask_stop=True) for e in embs]
        forced = np.array([r.tokens for r in refs])
        with cb.Session(model, embs, max_new_tokens=n, top_p=0.0, mask_stop=True, forced=forced) as s:
            for i in range(n + 1):
                lg = s.logits()
                for u in range(U):
                    assert rel_err(lg[u:u + 1], refs[u].raw_logits[i]) < 3e-2, f"sentence {u} step {i}"
                s.decode(1)
    finally:
        model.close()
