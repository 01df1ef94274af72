"""Continuous batching (t3mi_session_refill, chatterbox_amd.t3_inference_continuous): slots of a running session take new utterances.

The reference has no counterpart (tts_chunked is serial, R/tts.R:466-474); the bar is self-consistency with the paths that are
pinned elsewhere: with fp32 weights a row's arithmetic does not depend on what the other rows of the batch hold, and utterance i
draws from the random stream (seed, first_utterance_id + i) wherever it runs, so a refilled utterance must produce exactly the
tokens -- greedy AND sampled -- and the logits it produces alone, and the slots that were not touched must not notice."""
import numpy as np
import pytest

import chatterbox_amd as cb
from chatterbox_amd import _lib, synth
from tests.helpers import rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tiny():
    dims = synth.TINY
    m = cb.load_t3(synth.ensure_checkpoint(dims, 1234), device=0, start_speech_token=dims.start_speech_token,
                   stop_speech_token=dims.stop_speech_token)
    yield dims, m
    m.close()


def _emb(dims, s0, seed):
    return synth.synth_embeds(2, s0, dims.hidden_size, seed)


@pytest.mark.parametrize("U", [2, 6])      # 4 rows: the GEMV path; 12 rows: the MFMA GEMM path
@pytest.mark.parametrize("top_p", [0.0, 0.95])
def test_refilled_slot_equals_the_utterance_alone(tiny, U, top_p):
    dims, model = tiny
    embs = [_emb(dims, 7 + (u % 4), 100 + u) for u in range(U)]
    new = {1: _emb(dims, 12, 900), U - 2 if U > 2 else 0: _emb(dims, 5, 901)}
    kw = dict(max_new_tokens=24, top_p=top_p, seed=9, mask_stop=True, max_cache_len=48)
    # the session left alone
    with cb.Session(model, embs, **kw) as s:
        s.decode(20)
        base = [s.tokens(u) for u in range(U)]
    # the same session with two slots refilled after 6 steps
    slots = sorted(new)
    with cb.Session(model, embs, **kw) as s:
        s.decode(6)
        head = [s.tokens(u) for u in range(U)]
        s.refill(slots, [new[u] for u in slots], max_new_tokens=[9, 24], utterance_ids=[70, 71])
        lg = s.logits()
        s.decode(14)
        n, done = s.status()
        got = [s.tokens(u) for u in range(U)]
    for u in range(U):
        assert list(head[u]) == list(base[u][:6])
        if u not in new:
            assert list(got[u]) == list(base[u]), f"slot {u} was not refilled and changed"
    assert n[slots[0]] == 9 and done[slots[0]] == 2          # its own limit
    assert n[slots[1]] == 14 and done[slots[1]] == 0
    # each new utterance alone, with its global index as first_utterance_id
    for k, u in enumerate(slots):
        with cb.Session(model, [new[u]], first_utterance_id=70 + k, **kw) as s1:
            l1 = s1.logits()
            s1.decode(14)
            alone = s1.tokens(0)
        assert rel_err(lg[2 * u: 2 * u + 2], l1) < 1e-5       # first logits (another batch size: another kernel may serve the head)
        m = min(len(alone), len(got[u]))
        assert m >= 9 and list(got[u][:m]) == list(alone[:m]), (u, got[u], alone)


def test_refill_bf16_weights_on_the_gemv_path():
    """bf16 weights with 4 rows: decode on the GEMV kernels (bf16 weights widened in registers, fp32 rows and cache), prefill of the
    new prompt through the mixed-type GEMM (GemmBF16W) exactly as t3mi_session_begin does -- the refilled utterance's greedy tokens
    are those of the utterance alone."""
    dims = synth.TINY
    model = cb.load_t3(synth.ensure_checkpoint(dims, 1234), device=0, dtype="bf16", start_speech_token=dims.start_speech_token,
                       stop_speech_token=dims.stop_speech_token)
    try:
        embs = [_emb(dims, 8, 11), _emb(dims, 6, 12)]
        new = _emb(dims, 10, 13)
        kw = dict(max_new_tokens=16, top_p=0.0, mask_stop=True, max_cache_len=40)
        with cb.Session(model, embs, **kw) as s:
            s.decode(5)
            s.refill([0], [new], utterance_ids=[9])
            s.decode(11)
            got0, got1 = s.tokens(0), s.tokens(1)
        with cb.Session(model, [new], first_utterance_id=9, **kw) as s:
            s.decode(11)
            alone = s.tokens(0)
        with cb.Session(model, embs, **kw) as s:
            s.decode(16)
            base1 = s.tokens(1)
        assert list(got0) == list(alone) and len(alone) == 11
        assert list(got1) == list(base1)
    finally:
        model.close()


def test_refill_refuses_what_it_cannot_do(tiny):
    dims, model = tiny
    embs = [_emb(dims, 6, 1), _emb(dims, 6, 2)]
    with cb.Session(model, embs, max_new_tokens=8, top_p=0.0, mask_stop=True) as s:
        s.decode(2)
        with pytest.raises(_lib.T3miError):
            s.refill([2], [embs[0]])                          # no such slot
        with pytest.raises(_lib.T3miError):
            s.refill([0, 0], [embs[0], embs[1]])              # listed twice
        with pytest.raises(_lib.T3miError):
            s.refill([0], [_emb(dims, 40, 3)])                # does not fit the cache rows (6 + 8 slots)
        with pytest.raises(_lib.T3miError):
            s.refill([0], [embs[0]], max_new_tokens=[0])
        with pytest.raises(ValueError):
            s.refill([0], [embs[0][:1]])                      # one row for a CFG session
        s.refill([1], [embs[0]])                              # and still works afterwards
        s.decode(3)
        assert s.status()[0][1] == 3
    forced = np.zeros((2, 4), dtype=np.int32)
    with cb.Session(model, embs, max_new_tokens=8, top_p=0.0, forced=forced) as s:
        with pytest.raises(_lib.T3miError):
            s.refill([0], [embs[0]])                          # teacher-forced session


@pytest.mark.parametrize("top_p", [0.0, 0.9])
def test_continuous_equals_batch(tiny, top_p):
    """11 utterances of ragged length through 4 slots == the one-call batch, truncated at each utterance's own limit."""
    dims, model = tiny
    n = 11
    embs = [_emb(dims, 5 + (u * 3) % 7, 300 + u) for u in range(n)]
    lim = [5 + (u * 7) % 23 for u in range(n)]
    kw = dict(top_p=top_p, seed=4, mask_stop=True, first_utterance_id=1000)
    ref = cb.t3_inference_batch(model, embs, max_new_tokens=max(lim), **kw)
    st = {}
    got = cb.t3_inference_continuous(model, embs, slots=4, poll_steps=3, max_new_tokens=lim, stats=st, **kw)
    assert st["refill_calls"] >= 2 and 0.3 < st["occupancy"] <= 1.0
    for u in range(n):
        assert list(got[u]) == list(ref[u][: lim[u]]), (u, got[u], ref[u][: lim[u]])
    # the queue started longest first: another schedule, the same tokens (utterance i keeps the stream (seed, 1000 + i))
    st2 = {}
    lf = cb.t3_inference_continuous(model, embs, slots=4, poll_steps=3, max_new_tokens=lim, longest_first=True, stats=st2, **kw)
    for u in range(n):
        assert list(lf[u]) == list(ref[u][: lim[u]]), (u, lf[u], ref[u][: lim[u]])
    assert st2["steps"] <= st["steps"]
    # one slot: the serial order of tts_chunked
    one = cb.t3_inference_continuous(model, embs[:3], slots=1, poll_steps=4, max_new_tokens=lim[:3], **kw)
    for u in range(3):
        assert list(one[u]) == list(ref[u][: lim[u]])


def test_continuous_stop_token_ends_an_utterance(tiny):
    """Without mask_stop the stop token ends an utterance wherever it falls, and its slot is handed on.  The synthetic model never
    favours its nominal stop id, so the model is reopened with a stop id that the greedy streams do contain (at a different step in
    each utterance); greedy, because a flat sampled distribution turns the last-bit differences between batch sizes (another GEMM
    kernel serves 10 rows than 32) into other draws."""
    dims, model = tiny
    n = 16
    embs = [_emb(dims, 6 + u % 3, 500 + u) for u in range(n)]
    free = cb.t3_inference_batch(model, embs, max_new_tokens=60, top_p=0.0, mask_stop=True)
    ids, cnt = np.unique(np.concatenate([np.unique(t[2:]) for t in free]), return_counts=True)
    stop = int(ids[np.argmax(cnt)])                       # the id that occurs (after step 2) in the most utterances
    assert cnt.max() >= 3, "the greedy streams share no token: pick other seeds"
    m2 = cb.load_t3(synth.ensure_checkpoint(dims, 1234), device=0, start_speech_token=dims.start_speech_token, stop_speech_token=stop)
    try:
        ref = cb.t3_inference_batch(m2, embs, max_new_tokens=60, top_p=0.0)
        st = {}
        got = cb.t3_inference_continuous(m2, embs, slots=5, poll_steps=7, max_new_tokens=60, top_p=0.0, stats=st)
    finally:
        m2.close()
    ends = sorted(len(r) for r in ref)
    assert ends[0] < 60 and len(set(ends)) >= 3, ends      # utterances do end early, at different steps
    assert st["refill_calls"] >= 2
    for u in range(n):
        assert list(got[u]) == list(ref[u]), (u, got[u], ref[u])
        assert len(ref[u]) == 60 or ref[u][-1] == stop


def test_continuous_from_tokens_and_turbo():
    """The token front door (prompts built on the device) and the Turbo variant."""
    dims = synth.TINY
    model = cb.load_t3(synth.ensure_checkpoint(dims, 1234), device=0, start_speech_token=dims.start_speech_token,
                       stop_speech_token=dims.stop_speech_token)
    rng = np.random.default_rng(3)
    try:
        conds = [cb.t3_cond((rng.standard_normal(dims.speaker_embed) * 0.1).astype(np.float32),
                            rng.integers(0, dims.start_speech_token, 150)) for _ in range(5)]
        texts = [rng.integers(1, min(dims.text_vocab, 255), 4 + u) for u in range(5)]
        kw = dict(max_new_tokens=10, top_p=0.0, mask_stop=True)
        ref = cb.t3_inference_batch(model, conds, texts, **kw)
        got = cb.t3_inference_continuous(model, conds, texts, slots=2, poll_steps=4, **kw)
        for u in range(5):
            assert list(got[u]) == list(ref[u])
    finally:
        model.close()
    td = synth.TURBO_TINY
    tm = cb.load_t3_turbo(synth.ensure_turbo_checkpoint(td, 4321), device=0, start_speech_token=td.start_speech_token,
                          stop_speech_token=td.stop_speech_token)
    try:
        embs = [synth.synth_embeds(1, 9 + u, td.hidden_size, 40 + u) for u in range(6)]
        lim = [6, 11, 4, 9, 12, 7]
        ref = cb.t3_inference_turbo_batch(tm, embs, max_new_tokens=12, top_k=1, mask_stop=True, keep_stop=True)
        got = cb.t3_inference_continuous(tm, embs, slots=2, poll_steps=3, max_new_tokens=lim, top_k=1, mask_stop=True, keep_stop=True)
        for u in range(6):
            assert list(got[u]) == list(ref[u][: lim[u] + 1]), (u, got[u], ref[u])
    finally:
        tm.close()


def test_refill_full_size_bf16_many_rows():
    """Full-size model, bf16 weights + bf16 cache, 12 utterances (24 rows, the activation-stationary GEMMs): two slots refilled; the
    new utterances' logits stay inside the bf16 bar of the same utterances in a fresh session, the untouched rows do not change."""
    dims = synth.FULL
    model = cb.load_t3(synth.ensure_checkpoint(dims, 1234), device=0, dtype="bf16")
    try:
        embs = [_emb(dims, 20 + u % 5, 600 + u) for u in range(12)]
        new = [_emb(dims, 31, 700), _emb(dims, 17, 701)]
        kw = dict(max_new_tokens=12, top_p=0.0, mask_stop=True, max_cache_len=64)
        with cb.Session(model, embs, **kw) as s:
            s.decode(8)
            base = [s.tokens(u) for u in range(12)]
        with cb.Session(model, embs, **kw) as s:
            s.decode(3)
            s.refill([4, 9], new)
            lg = s.logits()
            s.decode(5)
            got = [s.tokens(u) for u in range(12)]
        for u in range(12):
            if u not in (4, 9):
                assert list(got[u]) == list(base[u])
        fresh = embs[:4] + [new[0]] + embs[5:9] + [new[1]] + embs[10:]
        with cb.Session(model, fresh, **kw) as s:
            lf = s.logits()
        for u in (4, 9):
            assert rel_err(lg[2 * u: 2 * u + 2], lf[2 * u: 2 * u + 2]) < 3e-2
        assert np.array_equal(lg[:8], _first_logits_after(model, embs, kw, 3)[:8])
    finally:
        model.close()


def _first_logits_after(model, embs, kw, steps):
    with cb.Session(model, embs, **kw) as s:
        s.decode(steps)
        return s.logits()


def test_continuous_random_jobs(tiny):
    """Thirty small jobs with random shapes -- 1 to 40 utterances, 1 to 12 slots (more slots than utterances too), polls of 1 to 9
    steps, prompts of 4 to 20 positions, limits of 1 to 30 tokens (some below the poll interval), both queue orders: every one equals
    the one-call batch cut at each utterance's limit.  Greedy: the batch runs 2 to 80 rows and the slots 2 to 24, i.e. other GEMM
    kernels whose sums differ in the last bit, and a sampled draw that sits on a CDF step then differs (seen once in 30 sampled
    jobs); sampled equality at equal kernels is test_continuous_equals_batch's and test_refilled_slot_equals_the_utterance_alone's."""
    dims, model = tiny
    rng = np.random.default_rng(2024)
    for job in range(30):
        n = int(rng.integers(1, 41))
        slots = int(rng.integers(1, 13))
        poll = int(rng.integers(1, 10))
        embs = [_emb(dims, int(rng.integers(4, 21)), 5000 + 50 * job + u) for u in range(n)]
        lim = [int(v) for v in rng.integers(1, 31, n)]
        kw = dict(top_p=0.0, seed=job, mask_stop=True, first_utterance_id=int(rng.integers(0, 1000)))
        ref = cb.t3_inference_batch(model, embs, max_new_tokens=max(lim), **kw)
        st = {}
        got = cb.t3_inference_continuous(model, embs, slots=slots, poll_steps=poll, max_new_tokens=lim, longest_first=bool(job % 3 == 0),
                                         stats=st, **kw)
        assert st["slots"] == min(slots, n)
        for u in range(n):
            assert list(got[u]) == list(ref[u][: lim[u]]), (job, n, slots, poll, u, lim[u], got[u], ref[u])
