"""Known-answer tests that pin the CPU oracle to the reference's own fixtures
(KAT-1..KAT-5).  CPU only."""
import os

import numpy as np
import pytest

from oracle import t3_oracle as O

F32 = np.float32


@pytest.fixture(scope="module")
def dbg(golden_dir):
    return np.load(os.path.join(golden_dir, "ref_t3_debug.npz"))


def test_kat1_cfg_combine_bit_exact(dbg):
    # outputs/t3_debug.safetensors: logits_after_cfg == c + 0.5 (c - u) exactly
    got = O.cfg_combine(dbg["raw_logits_step0"], 0.5)
    assert np.array_equal(got, dbg["logits_after_cfg"][0])


def test_kat2_temperature_after_penalty(dbg):
    # upstream applied the HF sign-aware penalty on id 6561; the R rule (Q2/Q3)
    # differs only on columns 6560/6561 -> compare everywhere else.
    p = O.SampleParams(temperature=0.8, repetition_penalty=1.2, variant="r")
    seen = [6560]
    lg = O.rep_penalty(dbg["logits_after_cfg"][0], seen, p.repetition_penalty)
    lg = (lg / F32(p.temperature)).astype(F32)
    ref = dbg["logits_after_temp"][0]
    keep = np.ones(lg.shape[0], bool)
    keep[[6560, 6561]] = False
    assert np.abs(lg[keep] - ref[keep]).max() < 2e-6
    # and the R rule really divides column 6560 regardless of sign
    c = dbg["logits_after_cfg"][0][6560]
    assert lg[6560] == F32(F32(c / F32(1.2)) / F32(0.8))


def test_kat3_rope_bands():
    cfg = O.LlamaConfig()
    assert O.rope_band_table(cfg) == "H" * 15 + "M" * 3 + "L" * 14
    inv = O.llama3_inv_freq(cfg)
    np.testing.assert_allclose(inv[15:18], [1.371894e-3, 5.24846e-4, 1.785078e-4], rtol=2e-6)
    np.testing.assert_allclose(inv[18], 7.784655e-5, rtol=2e-6)
    cos, sin = O.rope_tables(cfg, 8)
    assert cos.shape == (8, 64) and np.array_equal(cos[:, :32], cos[:, 32:])
    assert np.all(cos[0] == 1) and np.all(sin[0] == 0)


def test_kat4_r_rule_tail_support(dbg):
    p = O.SampleParams(temperature=0.8, min_p=0.05, top_p=0.9, repetition_penalty=1.2, variant="r")
    t = O.sampling_tail(dbg["logits_after_cfg"][0], [6560], p, u=0.0)
    assert len(t.sorted_idx) == 31
    assert list(t.sorted_idx[:3]) == [1708, 1709, 3689]
    np.testing.assert_allclose(t.sorted_probs[:3], [0.1572, 0.1043, 0.0684], atol=5e-4)
    assert abs(float(t.sorted_probs.sum()) - 1.0) < 1e-5
    # token upstream actually sampled at step 0 is the head of the support
    assert int(dbg["sampled_tokens"][0]) == 1708 and t.token == 1708


@pytest.mark.parametrize("variant", ["r", "cpp"])
def test_kat5_top_p_zero_is_argmax(dbg, variant):
    p = O.SampleParams(top_p=0.0, variant=variant)
    seen = [6560] if variant == "r" else []
    t = O.sampling_tail(dbg["logits_after_cfg"][0], seen, p, u=0.73)
    lg = t.logits_after_temp
    assert len(t.sorted_idx) == 1 and t.token == int(np.argmax(lg))
    assert t.sorted_probs[0] == 1.0


def test_q4_variants_differ_by_crossing_token(dbg):
    pr = O.SampleParams(top_p=0.9, variant="r")
    pc = O.SampleParams(top_p=0.9, variant="cpp")
    tr = O.sampling_tail(dbg["logits_after_cfg"][0], [], pr, 0.0)
    tc = O.sampling_tail(dbg["logits_after_cfg"][0], [], pc, 0.0)
    assert len(tc.sorted_idx) == len(tr.sorted_idx) + 1
    assert np.array_equal(tc.sorted_idx[:-1], tr.sorted_idx)


def test_filtered_distribution_fixture_shape(dbg):
    # probs_step0 is the filtered distribution of the LAST debug step:
    # 18 non-zeros summing to 1 that contain sampled_tokens[5]
    pr = dbg["probs_step0"][0]
    assert np.count_nonzero(pr) == 18 and abs(pr.sum() - 1) < 1e-5
    assert pr[int(dbg["sampled_tokens"][5])] > 0


def test_reference_token_streams_are_well_formed(golden_dir):
    t = np.load(os.path.join(golden_dir, "ref_tokens.npz"))
    for k in ("speech_tokens", "python_generated"):
        s = t[k]
        assert s[-1] == 6562 and np.all(s[:-1] < 6561)
    tt = t["text_tokens_with_sot_eot"]
    assert tt[0] == 255 and tt[-1] == 0


def test_counter_uniform_range_and_determinism():
    us = [O.counter_uniform(7, u, s) for u in range(4) for s in range(64)]
    assert all(0.0 <= x < 1.0 for x in us) and len(set(us)) > 250
    assert O.counter_uniform(7, 1, 2) == O.counter_uniform(7, 1, 2)


def test_kat6_drop_invalid_tokens_on_the_reference_streams(golden_dir):
    """KAT-6 (R/s3tokenizer.R:678-687): both token streams the reference ships end in the stop token 6562 and hold
    nothing else from the special range, so drop_invalid_tokens removes exactly the last id."""
    g = np.load(os.path.join(golden_dir, "ref_tokens.npz"))
    for k in ("speech_tokens", "python_generated"):
        t = g[k].astype(np.int64)
        kept = O.drop_invalid_tokens(t)
        assert t[-1] == 6562 and list(kept) == list(t[:-1]) and kept.max() < O.SPEECH_VOCAB_SIZE
    assert list(O.drop_invalid_tokens([6561, 5, 6560, 8193, 0, 6562, 7])) == [5, 6560, 0, 7]
    assert O.drop_invalid_tokens([]).size == 0
