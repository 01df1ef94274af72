"""The reference's own recorded logits through the DEVICE sampling tail (missing #2).

/root/reference/scripts/debug_t3_step.R:85-235 replays outputs/t3_debug.safetensors -- `raw_logits_step0 [2, 8194]`
-> CFG combine -> repetition penalty -> temperature -> min-p -> top-p -> draw -- against the upstream Python run.  The
weights that produced those logits are not in the container, so the backbone cannot reproduce them; but the sampling
tail can be fed them directly: t3mi_session_set_logits overwrites the pending logits of a session on a small synthetic
model with the real vocabulary (V = 8194, start 6561, stop 6562), the next decode step runs `sampler_kernel` on the
reference's vector, and t3mi_session_filtered_logits / _dist / _drawn return what the kernel computed.  The fixture
(tests/golden/ref_t3_debug.npz) is data extracted from the reference's file by scripts/make_golden_from_reference.py.

Bars: CFG combine bit-exact (KAT-1); penalty + temperature within 2e-6 of `logits_after_temp` off the two columns where
upstream's HF rule and the R rule differ (KAT-2) and bit-exact against the R rule restated in the oracle;
the R-rule support of 31 tokens with head 1708 / 1709 / 3689 (KAT-4); the cpp rule keeps the crossing token (Q4);
top_p = 0 gives 1708 = the token upstream sampled (KAT-5)."""
import os

import numpy as np
import pytest

import chatterbox_amd as cb
from chatterbox_amd import _lib, synth
from oracle import t3_oracle as O
from tests.helpers import assert_dist_close

pytestmark = pytest.mark.gpu

F32 = np.float32
# a small backbone under the REAL speech vocabulary and special ids (R/t3.R:17-20)
DIMS = synth.T3Dims(hidden_size=512, intermediate_size=1024, num_hidden_layers=1, num_attention_heads=8, head_dim=64,
                    speech_vocab=8194, max_speech_pos=64, start_speech_token=6561, stop_speech_token=6562,
                    text_vocab=300, max_text_pos=80)


@pytest.fixture(scope="module")
def dbg(golden_dir):
    return np.load(os.path.join(golden_dir, "ref_t3_debug.npz"))


@pytest.fixture(scope="module")
def model():
    m = cb.load_t3(synth.ensure_checkpoint(DIMS, 77), device=0)
    assert m.vocab == 8194
    yield m
    m.close()


def _step(model, raw, **kw):
    """One decode step whose sampling tail reads `raw` [2, 8194] instead of the model's own logits."""
    emb = synth.synth_embeds(2, 6, DIMS.hidden_size, 5)
    kw.setdefault("max_new_tokens", 4)
    with cb.Session(model, [emb], **kw) as s:
        s.keep_filtered()
        s.set_logits(raw)
        assert np.array_equal(s.logits(), raw)
        s.decode(1)
        ids, pr = s.dist(0)
        return s.filtered_logits(0), ids, pr, int(s.drawn()[0]), int(s.tokens(0)[0])


def test_kat1_cfg_combine_on_device_is_bit_exact(model, dbg):
    """scripts/debug_t3_step.R:114-127: logits_cfg = cond + 0.5 (cond - uncond) == `logits_after_cfg`, every bit."""
    raw = np.ascontiguousarray(dbg["raw_logits_step0"], dtype=F32)
    filt, *_ = _step(model, raw, cfg_weight=0.5, temperature=1.0, repetition_penalty=1.0, top_p=0.0)
    assert np.array_equal(filt, dbg["logits_after_cfg"][0])
    # and with a weight whose product is not exact in fp32: still the oracle's rounding sequence
    filt2, *_ = _step(model, raw, cfg_weight=0.3, temperature=1.0, repetition_penalty=1.0, top_p=0.0)
    assert np.array_equal(filt2, O.cfg_combine(raw, 0.3))


def test_kat2_penalty_and_temperature_on_device(model, dbg):
    """scripts/debug_t3_step.R:131-151: after the penalty and / 0.8 the row equals `logits_after_temp` (upstream: HF sign-aware
    rule on id 6561) everywhere except columns 6560 / 6561, where t3_inference's own rule applies (R/t3.R:529-532: column
    BOS - 1 = 6560 divided regardless of sign, Q2/Q3).  Against the oracle's restatement of that rule: bit-exact."""
    raw = np.ascontiguousarray(dbg["raw_logits_step0"], dtype=F32)
    filt, *_ = _step(model, raw, cfg_weight=0.5, temperature=0.8, repetition_penalty=1.2, top_p=0.0)
    ref = dbg["logits_after_temp"][0]
    keep = np.ones(8194, bool)
    keep[[6560, 6561]] = False
    assert np.abs(filt[keep] - ref[keep]).max() < 2e-6
    c = dbg["logits_after_cfg"][0]
    assert filt[6560] == F32(F32(c[6560] / F32(1.2)) / F32(0.8))
    assert filt[6561] == F32(c[6561] / F32(0.8))
    want = O.sampling_tail(c, [6560], O.SampleParams(temperature=0.8, repetition_penalty=1.2, top_p=0.0), 0.0).logits_after_temp
    assert np.array_equal(filt, want)


def test_kat4_r_rule_support_on_device(model, dbg):
    """scripts/debug_t3_step.R:154-199 with t3_inference's rule (min-p in probability space, R/t3.R:543-563): 31 tokens survive
    min_p .05 / top_p .9, headed by 1708 (.1572), 1709 (.1043), 3689 (.0684); the whole support and its probabilities equal
    the oracle's, and the draw is the oracle's for the same uniform."""
    raw = np.ascontiguousarray(dbg["raw_logits_step0"], dtype=F32)
    kw = dict(cfg_weight=0.5, temperature=0.8, repetition_penalty=1.2, min_p=0.05, top_p=0.9)
    for seed in (0, 3, 12345):
        _, ids, pr, drawn, tok = _step(model, raw, seed=seed, **kw)
        assert len(ids) == 31 and list(ids[:3]) == [1708, 1709, 3689]
        np.testing.assert_allclose(pr[:3], [0.1572, 0.1043, 0.0684], atol=5e-4)
        t = O.sampling_tail(dbg["logits_after_cfg"][0], [6560], O.SampleParams(**{k: v for k, v in kw.items() if k != "cfg_weight"}),
                            O.counter_uniform(seed, 0, 0))
        assert list(ids) == list(t.sorted_idx)
        assert assert_dist_close(ids, pr, t)
        assert drawn == t.token == tok
    # min-p alone (top_p = 1): the support scripts/debug_t3_step.R:166 counts as "Non-inf count after min_p"
    _, ids, pr, *_ = _step(model, raw, **dict(kw, top_p=1.0))
    lt = dbg["logits_after_temp"][0].astype(np.float64)
    p = np.exp(lt - lt.max())
    p /= p.sum()
    assert abs(len(ids) - int((p >= 0.05 * p.max()).sum())) <= 1 and ids[0] == 1708


def test_q4_cpp_rule_keeps_the_crossing_token_on_device(model, dbg):
    """src/t3_decode.cpp:318-357: empty seen set at step 0 and the shifted top-p mask -- one token more than the R rule on the
    same logits (the one whose cumulative mass crosses top_p), identical up to it."""
    raw = np.ascontiguousarray(dbg["raw_logits_step0"], dtype=F32)
    kw = dict(cfg_weight=0.5, temperature=0.8, repetition_penalty=1.2, min_p=0.05, top_p=0.9)
    _, ids_c, pr_c, drawn, _ = _step(model, raw, sampler=_lib.SAMPLER_CPP, **kw)
    _, ids_r, _, _, _ = _step(model, raw, **kw)
    t = O.sampling_tail(dbg["logits_after_cfg"][0], [], O.SampleParams(temperature=0.8, repetition_penalty=1.2, min_p=0.05,
                                                                        top_p=0.9, variant="cpp"), O.counter_uniform(0, 0, 0))
    assert len(ids_c) == 32 == len(t.sorted_idx) and list(ids_c) == list(t.sorted_idx)
    assert list(ids_c[:31]) == list(ids_r)
    assert assert_dist_close(ids_c, pr_c, t) and drawn == t.token


@pytest.mark.parametrize("sampler", [_lib.SAMPLER_R, _lib.SAMPLER_CPP])
def test_kat5_top_p_zero_gives_the_token_upstream_sampled(model, dbg, sampler):
    """Q6: top_p = 0 leaves sorted index 1 only -- 1708, which is also `sampled_tokens[0]` of the upstream run."""
    raw = np.ascontiguousarray(dbg["raw_logits_step0"], dtype=F32)
    _, ids, pr, drawn, tok = _step(model, raw, cfg_weight=0.5, temperature=0.8, repetition_penalty=1.2, min_p=0.05, top_p=0.0,
                                   sampler=sampler)
    assert list(ids) == [1708] and pr[0] == 1.0 and drawn == tok == 1708 == int(dbg["sampled_tokens"][0])


def test_last_step_support_of_the_fixture(model, dbg):
    """`probs_step0` is the filtered distribution of the LAST debug step: 18 non-zeros.  Its logits are not in the
    file, but any logits whose softmax it is reproduce it: log(p) on the support, far below elsewhere.  Through the device tail
    with the filters off (min_p 0, top_p 1, T 1, no penalty, no CFG) the distribution must come back: same 18 ids, same
    probabilities."""
    pr = dbg["probs_step0"][0].astype(np.float64)
    lg = np.full((2, 8194), -80.0, dtype=F32)
    nz = np.nonzero(pr)[0]
    lg[0, nz] = np.log(pr[nz]).astype(F32)
    lg[1] = lg[0]
    emb = synth.synth_embeds(1, 6, DIMS.hidden_size, 5)
    with cb.Session(model, [emb], max_new_tokens=4, cfg_weight=0.0, temperature=1.0, repetition_penalty=1.0, min_p=1e-6,
                    top_p=1.0) as s:
        s.set_logits(lg[:1])
        s.decode(1)
        ids, p = s.dist(0)
    assert sorted(ids) == sorted(nz) and len(ids) == 18
    got = dict(zip(ids, p))
    assert max(abs(got[i] - pr[i]) for i in nz) < 1e-6
    assert int(dbg["sampled_tokens"][5]) in got


def test_hooks_reject_misuse(model):
    emb = synth.synth_embeds(2, 6, DIMS.hidden_size, 5)
    with cb.Session(model, [emb], max_new_tokens=4, top_p=0.0) as s:
        with pytest.raises(_lib.T3miError) as e:
            s.filtered_logits(0)
        assert e.value.code == _lib.ERR_STATE
        s.decode(1)
        with pytest.raises(_lib.T3miError) as e:
            s.keep_filtered()      # the step graph is already captured
        assert e.value.code == _lib.ERR_STATE
        with pytest.raises(ValueError):
            s.set_logits(np.zeros((1, 8194), F32))


@pytest.mark.parametrize("masked", [0, 1, 2, 7])
def test_full_support_order_on_device(model, dbg, masked):
    """min_p = 0 and a top_p no cumulative sum exceeds: nothing is filtered and the kernel sorts the whole vocabulary (R/t3.R:551,
    descending probability; equal probabilities in ascending id order, as the oracle's stable sort).  8194 and 8193 entries sit
    just above 8192, where the kernel moves the last entries of the order to the tail by selection and sorts 8192 slots; 8192
    and 8187 take the plain sort.  The reference's logits quantised to steps of 0.5, so that hundreds of entries tie."""
    raw = np.ascontiguousarray(np.round(dbg["raw_logits_step0"] * 2) / 2, dtype=F32)
    gone = [17, 4096, 8193, 8192, 1, 5000, 6561][:masked]
    raw[:, gone] = F32(-1e4)      # probability underflows to 0: not part of the support
    kw = dict(cfg_weight=0.5, temperature=0.8, repetition_penalty=1.0, min_p=0.0, top_p=2.0)
    _, ids, pr, drawn, _ = _step(model, raw, seed=11, **kw)
    t = O.sampling_tail(O.cfg_combine(raw, 0.5), [], O.SampleParams(**{k: v for k, v in kw.items() if k != "cfg_weight"}),
                        O.counter_uniform(11, 0, 0))
    n = int((np.asarray(t.sorted_probs) > 0).sum())
    assert n == 8194 - masked == len(ids)
    assert list(ids) == [int(i) for i in t.sorted_idx[:n]]
    assert len(set(int(i) for i in ids[-8:]) & set(gone)) == 0
    np.testing.assert_allclose(pr, np.asarray(t.sorted_probs[:n], dtype=np.float64), rtol=2e-3, atol=1e-12)
    assert drawn in [int(i) for i in ids]
