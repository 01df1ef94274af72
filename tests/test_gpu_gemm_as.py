"""The activation-stationary bf16 GEMM (batch_kernels.hip, gemm_as_kernel; fragment-major weight copy) against the
LDS-staged kernels it replaces for batches of up to 1024 rows, and against the fp32 path.  (T3MI_AS=<mask> makes
exactly the named GEMMs take it at every row count the kernel accepts; without the variable each GEMM has a tuned
row range, t3mi.cpp frag().)

The two bf16 kernels multiply the same bf16 operands and accumulate in fp32; they differ only in the order of the
sums (K slices per wave, reduced in LDS).  A sum that differs in its last fp32 bit can flip the bf16 rounding of an
activation (1 part in 256), and 30 layers carry such flips forward, so the two bf16 paths agree to AS_VS_STAGED_TOL of
max |logit| (half the bf16 bar; measured up to 0.7 %), and each stays inside the bf16 bar against the fp32 path, the
new kernel no further from it than the old one (within 1.5x).  Every shape class of the launch heuristic is covered:
16- and 32-row chunks, 1 to 32 chunks, a ragged last chunk, the K-split residual GEMM with 1024-k and 512-k slabs (d_ff 4096 at full size, K = 1024 unsplit at tiny size), the
4- and 2-sub-tile blocks, a head whose vocabulary is not a multiple of 32 (mask 16: last tile partly empty), and
Turbo's bias / GELU / no-RoPE epilogues."""
import os

import numpy as np
import pytest

import chatterbox_amd as cb
from chatterbox_amd import synth
from tests.helpers import rel_err

pytestmark = pytest.mark.gpu

AS_VS_STAGED_TOL = 1.5e-2   # same bf16 operands, different summation order
BF16_LOGIT_TOL = 3e-2     # against the fp32 path (tests/test_gpu_scale.py)


def _open(path, turbo, mask, dims, dtype="bf16"):
    old = os.environ.get("T3MI_AS")
    os.environ["T3MI_AS"] = str(mask)     # read once, when the model is opened
    try:
        return (cb.load_t3_turbo if turbo else cb.load_t3)(path, device=0, dtype=dtype, start_speech_token=dims.start_speech_token,
                                                           stop_speech_token=dims.stop_speech_token)
    finally:
        if old is None:
            del os.environ["T3MI_AS"]
        else:
            os.environ["T3MI_AS"] = old


def _run(model, embs, n, forced=None):
    kw = dict(max_new_tokens=n + 2, top_p=0.0, mask_stop=True)
    if forced is not None:
        kw["forced"] = forced
    out = []
    with cb.Session(model, embs, **kw) as s:
        out.append(s.logits())
        for _ in range(n):
            s.decode(1)
            out.append(s.logits())
        toks = np.array([s.tokens(u)[:n + 1] for u in range(len(embs))])   # Turbo: n + 1 (the first comes from the prefill)
    return out, toks


CASES = [
    # (name, turbo, dims, utterances): llama rows = 2 x utterances, Turbo rows = utterances
    ("llama-tiny", False, synth.TINY, [3, 12, 24, 40, 64, 100, 260, 500]),
    ("llama-full", False, synth.FULL, [5, 16, 30, 44, 150]),
    ("turbo-tiny", True, synth.TURBO_TINY, [7, 24, 50, 100, 300, 1000]),
]


@pytest.mark.parametrize("name,turbo,dims,utts", CASES, ids=[c[0] for c in CASES])
def test_activation_stationary_gemm_equals_staged_gemm(name, turbo, dims, utts):
    path = synth.ensure_turbo_checkpoint(dims, 4321) if turbo else synth.ensure_checkpoint(dims, 1234)
    n = 3
    rpu = 1 if turbo else 2
    m32 = _open(path, turbo, 0, dims, dtype="f32")
    staged = _open(path, turbo, 0, dims)
    as_all = _open(path, turbo, 31, dims)      # every GEMM, the head included
    as_dflt = _open(path, turbo, 13, dims)     # the default set
    try:
        for U in utts:
            embs = [synth.synth_embeds(rpu, 9 + (u % 5), dims.hidden_size, 700 + u) for u in range(U)]
            ref, forced = _run(m32, embs, n)
            a, _ = _run(staged, embs, n, forced)
            b, _ = _run(as_all, embs, n, forced)
            c, _ = _run(as_dflt, embs, n, forced)
            for i in range(n + 1):
                assert np.isfinite(b[i]).all() and np.isfinite(c[i]).all()
                assert rel_err(b[i], a[i]) < AS_VS_STAGED_TOL, (name, U, i, rel_err(b[i], a[i]))
                assert rel_err(c[i], a[i]) < AS_VS_STAGED_TOL, (name, U, i, rel_err(c[i], a[i]))
                eb, ea = rel_err(b[i], ref[i]), rel_err(a[i], ref[i])
                assert eb < BF16_LOGIT_TOL and eb < 1.5 * ea + 1e-3, (name, U, i, eb, ea)
    finally:
        for m in (m32, staged, as_all, as_dflt):
            m.close()


F32_AS_VS_STAGED_TOL = 1e-4   # exact fp32 products on both sides, sums in another order (observed ~1e-6)


@pytest.mark.parametrize("name,turbo,dims,utts", [("llama-tiny", False, synth.TINY, [3, 12, 24, 40, 100, 300]),
                                                   ("llama-full", False, synth.FULL, [5, 30]),
                                                   ("turbo-tiny", True, synth.TURBO_TINY, [7, 50, 300])],
                         ids=["llama-tiny", "llama-full", "turbo-tiny"])
def test_fp32_activation_stationary_gemm_equals_staged_gemm(name, turbo, dims, utts):
    """fp32 operands through the same kernel (four v_mfma_f32_16x16x4_f32 per 64-byte step, exact fp32 products): every
    GEMM forced onto it, and the default row ranges, against the LDS-staged fp32 kernels; greedy tokens of the default
    dispatch equal the staged kernels' wherever the top two logits are not within the tolerance of each other."""
    path = synth.ensure_turbo_checkpoint(dims, 4321) if turbo else synth.ensure_checkpoint(dims, 1234)
    n = 3
    rpu = 1 if turbo else 2
    staged = _open(path, turbo, 0, dims, dtype="f32")
    as_all = _open(path, turbo, 31, dims, dtype="f32")
    as_dflt = (cb.load_t3_turbo if turbo else cb.load_t3)(path, device=0, dtype="f32", start_speech_token=dims.start_speech_token,
                                                          stop_speech_token=dims.stop_speech_token)
    try:
        for U in utts:
            embs = [synth.synth_embeds(rpu, 9 + (u % 5), dims.hidden_size, 700 + u) for u in range(U)]
            a, forced = _run(staged, embs, n)
            b, _ = _run(as_all, embs, n, forced)
            c, _ = _run(as_dflt, embs, n, forced)
            for i in range(n + 1):
                assert rel_err(b[i], a[i]) < F32_AS_VS_STAGED_TOL, (name, U, i, rel_err(b[i], a[i]))
                assert rel_err(c[i], a[i]) < F32_AS_VS_STAGED_TOL, (name, U, i, rel_err(c[i], a[i]))
    finally:
        for m in (staged, as_all, as_dflt):
            m.close()


@pytest.mark.parametrize("dtype,tol", [("bf16", AS_VS_STAGED_TOL), ("f32", F32_AS_VS_STAGED_TOL)])
def test_split_rms_norm_equals_the_norm_launch(dtype, tol):
    """Llama's post-attention RMS norm split over o_proj's epilogue (w_ln * x rows + per-tile sums of x^2) and gate_up's row
    scale, against the same kernels with the norm as a launch of its own (T3MI_NORM_SPLIT=0): full-size model, 6 to 128
    rows (16- and 32-row chunks, a ragged last chunk).  bf16: the split form rounds w * x where the launch rounds
    w * x * scale -- the same error class; fp32: sums in another order only."""
    dims = synth.FULL
    path = synth.ensure_checkpoint(dims, 1234)
    old = os.environ.get("T3MI_NORM_SPLIT")
    os.environ["T3MI_NORM_SPLIT"] = "0"
    try:
        plain = cb.load_t3(path, device=0, dtype=dtype)
    finally:
        if old is None:
            del os.environ["T3MI_NORM_SPLIT"]
        else:
            os.environ["T3MI_NORM_SPLIT"] = old
    split = cb.load_t3(path, device=0, dtype=dtype)
    # and with the pre-attention norm split too (bit 2: down_proj with K = 4096 unsplit writes the next layer's w_ln * x and sums of
    # squares, the next qkv scales its rows before RoPE; bf16, 17 to 128 rows; measured slower, off by default)
    os.environ["T3MI_NORM_SPLIT"] = "3"
    try:
        both = cb.load_t3(path, device=0, dtype=dtype)
    finally:
        if old is None:
            del os.environ["T3MI_NORM_SPLIT"]
        else:
            os.environ["T3MI_NORM_SPLIT"] = old
    ran_split = ran_both = False
    try:
        for U in (3, 11, 27, 64):
            embs = [synth.synth_embeds(2, 9 + (u % 5), dims.hidden_size, 900 + u) for u in range(U)]
            a, forced = _run(plain, embs, 3)
            b, _ = _run(split, embs, 3, forced)
            c, _ = _run(both, embs, 3, forced)
            for i in range(4):
                assert np.isfinite(b[i]).all() and np.isfinite(c[i]).all()
                assert rel_err(b[i], a[i]) < tol, (dtype, U, i, rel_err(b[i], a[i]))
                assert rel_err(c[i], a[i]) < tol, (dtype, U, i, rel_err(c[i], a[i]))
            ran_split = ran_split or any(not np.array_equal(b[i], a[i]) for i in range(4))
            ran_both = ran_both or any(not np.array_equal(c[i], b[i]) for i in range(4))
        assert ran_split, "bit-identical logits at every size: the split form did not run"   # (it sums and rounds in another order)
        assert ran_both == (dtype == "bf16"), "the pre-attention split runs for bf16 operands only"
    finally:
        plain.close()
        split.close()
        both.close()


@pytest.mark.parametrize("dtype,tol", [("bf16", AS_VS_STAGED_TOL), ("f32", F32_AS_VS_STAGED_TOL)])
def test_split_layer_norm_equals_the_norm_launch(dtype, tol):
    """Turbo's ln_2 (LayerNorm, R/t3.R:1190-1204) split the same way: attn.c_proj's blocks compute the new row mean as ln_1's
    row mean + attention row . column means of the weight + mean(bias), write w_ln * (x - mean) and per-tile sums of
    (x - mean)^2, and c_fc scales its rows and adds its bias with W . b_ln folded in -- against the same kernels with the
    LayerNorm as a launch (T3MI_NORM_SPLIT=0).  Full-size Turbo, 6 to 128 rows (one row per sentence)."""
    dims = synth.TURBO_FULL
    path = synth.ensure_turbo_checkpoint(dims, 4321)
    old = os.environ.get("T3MI_NORM_SPLIT")
    os.environ["T3MI_NORM_SPLIT"] = "0"
    try:
        plain = cb.load_t3_turbo(path, device=0, dtype=dtype)
    finally:
        if old is None:
            del os.environ["T3MI_NORM_SPLIT"]
        else:
            os.environ["T3MI_NORM_SPLIT"] = old
    split = cb.load_t3_turbo(path, device=0, dtype=dtype)
    ran_split = False
    try:
        for U in (6, 11, 27, 54, 128):
            embs = [synth.synth_embeds(1, 9 + (u % 5), dims.hidden_size, 900 + u) for u in range(U)]
            a, forced = _run(plain, embs, 3)
            b, _ = _run(split, embs, 3, forced)
            for i in range(4):
                assert np.isfinite(b[i]).all()
                assert rel_err(b[i], a[i]) < tol, (dtype, U, i, rel_err(b[i], a[i]))
            ran_split = ran_split or any(not np.array_equal(b[i], a[i]) for i in range(4))
        assert ran_split, "bit-identical logits at every size: the split form did not run"   # (it sums and rounds in another order)
    finally:
        plain.close()
        split.close()


@pytest.mark.parametrize("dtype,tol", [("bf16", AS_VS_STAGED_TOL), ("f32", F32_AS_VS_STAGED_TOL)])
def test_split_layer_norm_with_large_row_means(dtype, tol, tmp_path):
    """The same comparison on a tiny Turbo model whose statistics make every term of the mean formula large: token embeddings
    with mean 2 (spread 0.02), attn.c_proj weights with column means 0.01 (as large as their spread) and a c_proj bias with
    mean 0.3 -- the row mean the split derives (ln_1's mean + attention row . column means + mean(bias)) must centre the rows
    as well as the LayerNorm launch's own two-pass mean does."""
    dims = synth.TURBO_TINY
    w = synth.synth_turbo_weights(dims, 4321)
    w["speech_emb.weight"] = (w["speech_emb.weight"] + np.float32(2.0)).astype(np.float32)
    for l in range(dims.n_layer):
        p = f"tfmr.h.{l}."
        w[p + "attn.c_proj.weight"] = (w[p + "attn.c_proj.weight"] + np.float32(0.01)).astype(np.float32)
        w[p + "attn.c_proj.bias"] = (w[p + "attn.c_proj.bias"] + np.float32(0.3)).astype(np.float32)
    path = synth.write_safetensors(w, str(tmp_path / "turbo_offset.safetensors"))
    kw = dict(device=0, dtype=dtype, start_speech_token=dims.start_speech_token, stop_speech_token=dims.stop_speech_token)
    old = os.environ.get("T3MI_NORM_SPLIT")
    os.environ["T3MI_NORM_SPLIT"] = "0"
    try:
        plain = cb.load_t3_turbo(path, **kw)
    finally:
        if old is None:
            del os.environ["T3MI_NORM_SPLIT"]
        else:
            os.environ["T3MI_NORM_SPLIT"] = old
    split = cb.load_t3_turbo(path, **kw)
    ran_split = False
    try:
        for U in (7, 20, 33, 100):
            embs = [synth.synth_embeds(1, 9 + (u % 5), dims.hidden_size, 900 + u) + np.float32(1.5) for u in range(U)]
            a, forced = _run(plain, embs, 3)
            b, _ = _run(split, embs, 3, forced)
            for i in range(4):
                assert np.isfinite(b[i]).all()
                assert rel_err(b[i], a[i]) < tol, (dtype, U, i, rel_err(b[i], a[i]))
            ran_split = ran_split or any(not np.array_equal(b[i], a[i]) for i in range(4))
        assert ran_split, "bit-identical logits at every size: the split form did not run"   # (it sums and rounds in another order)
    finally:
        plain.close()
        split.close()


def test_activation_stationary_gemm_is_deterministic():
    """The K slices meet in LDS in a fixed order and the K-split slabs are folded in a fixed order: two sessions over
    the same batch give bit-identical logits."""
    dims = synth.FULL
    path = synth.ensure_checkpoint(dims, 1234)
    embs = [synth.synth_embeds(2, 20, dims.hidden_size, 40 + u) for u in range(40)]
    m = _open(path, False, 31, dims)
    try:
        a, ta = _run(m, embs, 4)
        b, tb = _run(m, embs, 4)
        assert np.array_equal(ta, tb)
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
    finally:
        m.close()


def _open_pf(path, turbo, dims, pf):
    old = os.environ.get("T3MI_PF")
    if pf is None:
        os.environ.pop("T3MI_PF", None)
    else:
        os.environ["T3MI_PF"] = pf      # read once, when the model is opened
    try:
        return (cb.load_t3_turbo if turbo else cb.load_t3)(path, device=0, dtype="bf16", start_speech_token=dims.start_speech_token,
                                                           stop_speech_token=dims.stop_speech_token)
    finally:
        if old is None:
            os.environ.pop("T3MI_PF", None)
        else:
            os.environ["T3MI_PF"] = old


@pytest.mark.parametrize("name,turbo,dims,utts", [("llama-full", False, synth.FULL, [3, 4, 6, 8, 12]),
                                                   ("turbo-full", True, synth.TURBO_FULL, [5, 8, 12, 16, 24])],
                         ids=["llama-full", "turbo-full"])
def test_weight_prefetch_changes_no_bit(name, turbo, dims, utts):
    """The weight prefetch from idle CUs (t3k::PfArgs; extra grid rows of the attention / qkv / o_proj / gate_up launches at up to
    16 activation rows) only loads: logits and tokens are bit-identical with it off (T3MI_PF=0), at its defaults, with the most
    blocks it accepts on every launch and the qkv launch forced to carry it (mask bit 32), and with the XCD mapping off.  Row
    counts on both sides of the 16-row limit and of the 128 / 256 (item, head) block thresholds."""
    path = synth.ensure_turbo_checkpoint(dims, 4321) if turbo else synth.ensure_checkpoint(dims, 1234)
    rpu = 1 if turbo else 2
    models = [_open_pf(path, turbo, dims, pf) for pf in ("0", None, "1024,256,63,1,1024,16", "64,128,15,0,64,16")]
    try:
        for U in utts:
            embs = [synth.synth_embeds(rpu, 9 + (u % 5), dims.hidden_size, 300 + u) for u in range(U)]
            ref, toks = _run(models[0], embs, 4)
            for m in models[1:]:
                got, t2 = _run(m, embs, 4)
                assert np.array_equal(t2, toks), (name, U)
                for i in range(5):
                    assert np.array_equal(got[i], ref[i]), (name, U, i)
    finally:
        for m in models:
            m.close()
