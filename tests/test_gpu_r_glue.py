"""The R glue (r-package/src/r_glue.c) executed: there is no R in the build image, so the glue is compiled against a
minimal test double of the R C API (tests/r_mock: vectors with names, external pointers with finalizers, Rf_error as a
non-local exit) and driven the way `.Call` drives it -- named config list as R/t3.R:1022-1034 builds it, t3_cond as
plain vectors, a column-major [rows, S0, d] array -- against the same library the ctypes path uses.  Results must be
identical to the ctypes path; every error path must free its C state and leave the handle usable."""
import ctypes as C
import os

import numpy as np
import pytest

import chatterbox_amd as cb
from chatterbox_amd import synth

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "r_mock", "librglue_mock.so")


class MockCfg(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("max_tokens", "sampler", "top_k", "max_cache_len", "start_token", "stop_token", "omit_required")] + \
               [(n, C.c_double) for n in ("temperature", "cfg_weight", "top_p", "min_p", "rep_penalty", "seed")] + \
               [(n, C.c_int) for n in ("slots", "poll_steps", "n_each")] + [("each", C.POINTER(C.c_int32))]


@pytest.fixture(scope="module")
def glue():
    if not os.path.exists(LIB):
        pytest.fail(f"{LIB} is not built (make -C tests/r_mock, or __graft_entry__.build())")
    lib = C.CDLL(LIB)
    lib.mock_last_error.restype = C.c_char_p
    lib.mock_last_warning.restype = C.c_char_p
    dims = synth.TINY
    path = synth.ensure_checkpoint(dims, 1234)
    assert lib.mock_open(path.encode(), 0, 0, 0, None) == 0, lib.mock_last_error()
    model = cb.load_t3(path, device=0, start_speech_token=dims.start_speech_token, stop_speech_token=dims.stop_speech_token)
    yield lib, dims, model
    model.close()
    assert lib.mock_close() == 0                      # the finalizer closed the handle and cleared the pointer


def _cfg(dims, **kw):
    c = MockCfg(max_tokens=12, sampler=0, top_k=0, max_cache_len=0, start_token=dims.start_speech_token,
                stop_token=dims.stop_speech_token, omit_required=0, temperature=0.8, cfg_weight=0.5, top_p=0.0, min_p=0.05,
                rep_penalty=1.2, seed=0.0)
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def _tokens_call(lib, dims, conds, texts, cfg, wrong_types=0):
    n = len(conds)
    spk = np.ascontiguousarray(np.stack([c[0] for c in conds]).astype(np.float64))
    ct = [np.ascontiguousarray(c[1], dtype=np.int32) for c in conds]
    tx = [np.ascontiguousarray(t, dtype=np.int32) for t in texts]
    IP = C.POINTER(C.c_int32)
    ctp = (IP * n)(*[a.ctypes.data_as(IP) for a in ct])
    txp = (IP * n)(*[a.ctypes.data_as(IP) for a in tx])
    nct = (C.c_int32 * n)(*[a.size for a in ct])
    ntx = (C.c_int32 * n)(*[a.size for a in tx])
    emo = (C.c_double * n)(*[float("nan") if c[2] is None else c[2] for c in conds])
    cap = cfg.max_tokens + 1
    out = np.zeros((n, cap), np.int32)
    n_out = (C.c_int32 * n)()
    rc = lib.mock_generate_tokens(n, spk.ctypes.data_as(C.POINTER(C.c_double)), spk.shape[1], ctp, nct, emo, txp, ntx, C.byref(cfg),
                                  wrong_types, out.ctypes.data_as(IP), cap, n_out)
    return rc, [list(out[u, :n_out[u]]) for u in range(n)]


def test_generate_tokens_through_the_glue_equals_the_ctypes_path(glue):
    lib, dims, model = glue
    rng = np.random.default_rng(5)
    conds = [((rng.standard_normal(dims.speaker_embed) * 0.1).astype(np.float32), rng.integers(0, dims.start_speech_token, 20 + 7 * u),
              0.5 if u != 1 else None) for u in range(3)]
    texts = [rng.integers(1, 250, 4 + 3 * u) for u in range(3)]
    rc, got = _tokens_call(lib, dims, conds, texts, _cfg(dims))
    assert rc == 0, lib.mock_last_error()
    want = cb.t3_inference_batch(model, [cb.t3_cond(s, t, emotion_adv=e) for s, t, e in conds], texts, max_new_tokens=12, top_p=0.0)
    assert got == [list(map(int, w)) for w in want]
    assert lib.mock_protect_balance() == 0
    # sampled, cpp rule, seed and cache cap read by name from the list
    cfg = _cfg(dims, top_p=0.9, sampler=1, max_cache_len=64, seed=7.0, max_tokens=30)
    rc, got = _tokens_call(lib, dims, conds[:1], texts[:1], cfg)
    assert rc == 0, lib.mock_last_error()
    want = cb.t3_inference_cpp(model, cb.t3_cond(*conds[0][:2], emotion_adv=conds[0][2]), texts[0], max_new_tokens=30, top_p=0.9,
                               max_cache_len=64, seed=7)
    stop = dims.stop_speech_token
    g = got[0][:got[0].index(stop)] if stop in got[0] else got[0]
    assert g == list(map(int, want))


def test_generate_from_a_column_major_array(glue):
    lib, dims, model = glue
    emb = synth.synth_embeds(2, 9, dims.hidden_size, 31)
    cfg = _cfg(dims)
    out = np.zeros(cfg.max_tokens + 1, np.int32)
    n = C.c_int32(0)
    rc = lib.mock_generate(emb.ctypes.data_as(C.POINTER(C.c_float)), 2, 9, dims.hidden_size, C.byref(cfg), 0,
                           out.ctypes.data_as(C.POINTER(C.c_int32)), out.size, C.byref(n))
    assert rc == 0, lib.mock_last_error()
    want = cb.t3_inference(model, emb, max_new_tokens=12, top_p=0.0)
    assert list(out[:n.value]) == list(map(int, want))
    # dims that do not match the array: refused with a message, C buffers freed, the handle still works
    rc = lib.mock_generate(emb.ctypes.data_as(C.POINTER(C.c_float)), 2, 9, dims.hidden_size, C.byref(cfg), 1,
                           out.ctypes.data_as(C.POINTER(C.c_int32)), out.size, C.byref(n))
    assert rc == -1 and b"numeric [rows, S0, d] array" in lib.mock_last_error()
    rc = lib.mock_generate(emb.ctypes.data_as(C.POINTER(C.c_float)), 2, 9, dims.hidden_size, C.byref(cfg), 0,
                           out.ctypes.data_as(C.POINTER(C.c_int32)), out.size, C.byref(n))
    assert rc == 0 and list(out[:n.value]) == list(map(int, want))


def test_generate_batch_ragged_and_error_paths(glue):
    lib, dims, model = glue
    embs = [synth.synth_embeds(2, 6 + 2 * u, dims.hidden_size, 60 + u) for u in range(3)]
    FP, IP = C.POINTER(C.c_float), C.POINTER(C.c_int32)
    ptrs = (FP * 3)(*[e.ctypes.data_as(FP) for e in embs])
    s0 = (C.c_int32 * 3)(*[e.shape[1] for e in embs])
    cfg = _cfg(dims)
    cap = cfg.max_tokens + 1
    out = np.zeros((3, cap), np.int32)
    n_out = (C.c_int32 * 3)()
    rc = lib.mock_generate_batch(3, ptrs, 2, s0, dims.hidden_size, C.byref(cfg), -1, out.ctypes.data_as(IP), cap, n_out)
    assert rc == 0, lib.mock_last_error()
    want = cb.t3_inference_batch(model, embs, max_new_tokens=12, top_p=0.0)
    assert [list(out[u, :n_out[u]]) for u in range(3)] == [list(map(int, w)) for w in want]
    # a mis-shaped array in the MIDDLE of the batch (after earlier conversions were malloc'ed) must not leak or crash
    for bad in (0, 1, 2):
        rc = lib.mock_generate_batch(3, ptrs, 2, s0, dims.hidden_size, C.byref(cfg), bad, out.ctypes.data_as(IP), cap, n_out)
        assert rc == -1 and b"bad arguments" in lib.mock_last_error()
    # a config list without its required entries: the reference's own message (src/t3_decode.cpp:50)
    rc = lib.mock_generate_batch(3, ptrs, 2, s0, dims.hidden_size, C.byref(_cfg(dims, omit_required=1)), -1, out.ctypes.data_as(IP), cap, n_out)
    assert rc == -1 and b"Missing config parameter: rep_penalty" in lib.mock_last_error()
    # library-side refusal (cfg_weight > 0 with ... a negative temperature) comes back as the library's text
    rc = lib.mock_generate_batch(3, ptrs, 2, s0, dims.hidden_size, C.byref(_cfg(dims, temperature=-1.0)), -1, out.ctypes.data_as(IP), cap, n_out)
    assert rc == -1 and b"temperature" in lib.mock_last_error()
    rc = lib.mock_generate_batch(3, ptrs, 2, s0, dims.hidden_size, C.byref(cfg), -1, out.ctypes.data_as(IP), cap, n_out)
    assert rc == 0 and lib.mock_protect_balance() == 0


def test_text_ids_of_the_wrong_r_type_are_refused(glue):
    lib, dims, model = glue
    rng = np.random.default_rng(9)
    conds = [((rng.standard_normal(dims.speaker_embed) * 0.1).astype(np.float32), rng.integers(0, dims.start_speech_token, 10), 0.5)]
    rc, _ = _tokens_call(lib, dims, conds, [rng.integers(1, 250, 5)], _cfg(dims), wrong_types=1)
    assert rc == -1 and b"text_tokens integer" in lib.mock_last_error()


# ---- S3Gen through the glue (r-package/src/r_glue_s3gen.c) -------------------------------------------------------------

def _s3gen_call(lib, fd, hd, tok, ptok, pfeat, xvec, z, phase, snoise, bad=0, finalize=1):
    from chatterbox_amd import synth as sy
    FP, IP = C.POINTER(C.c_float), C.POINTER(C.c_int32)
    f = lambda a: None if a is None else np.ascontiguousarray(a, dtype=np.float32)
    tok32, ptok32 = np.ascontiguousarray(tok, dtype=np.int32), np.ascontiguousarray(ptok, dtype=np.int32)
    pf, xv, zz, ph, sn = f(pfeat), f(xvec), f(z[0]), f(phase), f(snoise)
    cap = 2 * tok32.size
    mel = np.zeros(hd.mel * cap, np.float32)
    audio = np.zeros(cap * 480, np.float32)
    n = C.c_int32(0)
    p = lambda a, t=FP: None if a is None else a.ctypes.data_as(t)
    rc = lib.mock_s3gen(sy.ensure_flow_checkpoint(fd, 555).encode(), sy.ensure_hift_checkpoint(hd, 999).encode(), p(tok32, IP), tok32.size,
                        p(ptok32, IP), ptok32.size, p(pf), pf.shape[0], hd.mel, p(xv), xv.size, p(zz), zz.shape[1], 10, C.c_double(0.7), 0,
                        finalize, p(ph), hd.n_harm, p(sn), bad, p(mel), cap, C.byref(n), p(audio))
    T = n.value
    return rc, mel[:hd.mel * T].reshape(1, hd.mel, T), audio[:T * 480]


def test_s3gen_through_the_glue_equals_the_ctypes_path():
    """tokens -> mel -> waveform through t3mi_r_flow_open / _flow_inference / _hift_open / _hift_inference, driven as `.Call`
    drives them (integer ids, doubles for everything else, NULL for absent draws): the same bits as the ctypes path, whose
    parity with the oracle tests/test_gpu_cfm.py holds; every refusal frees its C state and the finalizers close both handles."""
    if not os.path.exists(LIB):
        pytest.fail(f"{LIB} is not built (make -C tests/r_mock, or __graft_entry__.build())")
    lib = C.CDLL(LIB)
    lib.mock_last_error.restype = C.c_char_p
    fd, hd = synth.FLOW_TINY, synth.HIFT_TINY
    tok, ptok, pfeat, xvec, z = synth.synth_flow_inputs(fd, 12, 7, 4)
    rng = np.random.default_rng(1)
    phase = rng.uniform(-np.pi, np.pi, hd.n_harm).astype(np.float32)
    snoise = rng.standard_normal((hd.n_harm, 24 * 480)).astype(np.float32)
    rc, mel, audio = _s3gen_call(lib, fd, hd, tok, ptok, pfeat, xvec, z, phase, snoise)
    assert rc == 0, lib.mock_last_error()
    with cb.load_s3gen_flow(synth.ensure_flow_checkpoint(fd, 555)) as flow, cb.load_hift_vocoder(synth.ensure_hift_checkpoint(hd, 999)) as voc:
        want_mel = flow.inference(tok, ptok, pfeat, xvec, z)
        want = voc.inference(want_mel, phase, snoise)["audio"]
        want_quiet = voc.inference(want_mel)["audio"]
    assert mel.shape == want_mel.shape and np.array_equal(mel, want_mel)
    assert np.array_equal(audio, want)
    # NULL phase / noise = the zero-phase, noise-free source
    rc, _, audio = _s3gen_call(lib, fd, hd, tok, ptok, pfeat, xvec, z, None, None)
    assert rc == 0 and np.array_equal(audio, want_quiet)
    # refusals: ids of the wrong R type, a short prompt mel, a short mel for the vocoder; an id outside the table is clamped (R/s3gen.R:833)
    rc, _, _ = _s3gen_call(lib, fd, hd, tok, ptok, pfeat, xvec, z, phase, snoise, bad=1)
    assert rc == -1 and b"integer vectors" in lib.mock_last_error()
    rc, _, _ = _s3gen_call(lib, fd, hd, tok, ptok, pfeat, xvec, z, phase, snoise, bad=2)
    assert rc == -1 and b"stated sizes" in lib.mock_last_error()
    rc, _, _ = _s3gen_call(lib, fd, hd, tok, ptok, pfeat, xvec, z, phase, snoise, bad=4)
    assert rc == -1 and b"stated sizes" in lib.mock_last_error()
    rc, mel3, _ = _s3gen_call(lib, fd, hd, tok, ptok, pfeat, xvec, z, phase, snoise, bad=3)
    t3 = np.array(tok).copy(); t3[0] = fd.vocab - 1
    with cb.load_s3gen_flow(synth.ensure_flow_checkpoint(fd, 555)) as flow:
        assert rc == 0 and np.array_equal(mel3, flow.inference(t3, ptok, pfeat, xvec, z))
    assert lib.mock_protect_balance() == 0


def test_continuous_batching_through_the_glue(glue):
    """`slots` / `poll_steps` / `max_tokens_each` in the config list of t3mi_r_generate_tokens (what t3_inference_batch(..., slots =)
    in R/t3_native.R passes): 7 utterances through 3 slots == the ctypes path == the one-call batch cut at each limit."""
    lib, dims, model = glue
    rng = np.random.default_rng(8)
    n = 7
    conds = [((rng.standard_normal(dims.speaker_embed) * 0.1).astype(np.float32), rng.integers(0, dims.start_speech_token, 15 + 5 * u), 0.5)
             for u in range(n)]
    texts = [rng.integers(1, 250, 3 + 2 * u) for u in range(n)]
    lim = np.array([5, 12, 3, 9, 12, 7, 10], dtype=np.int32)
    cfg = _cfg(dims, slots=3, poll_steps=4, n_each=n, each=lim.ctypes.data_as(C.POINTER(C.c_int32)))
    rc, got = _tokens_call(lib, dims, conds, texts, cfg)
    assert rc == 0, lib.mock_last_error()
    assert lib.mock_protect_balance() == 0
    cs = [cb.t3_cond(c[0], c[1], emotion_adv=c[2]) for c in conds]
    want = cb.t3_inference_continuous(model, cs, texts, slots=3, poll_steps=4, max_new_tokens=list(lim), top_p=0.0, seed=0)
    ref = cb.t3_inference_batch(model, cs, texts, max_new_tokens=12, top_p=0.0, seed=0)
    for u in range(n):
        assert got[u] == list(want[u]) == list(ref[u][: lim[u]]), (u, got[u], want[u], ref[u])
    # a limit list of the wrong length is refused before the library is called
    bad = _cfg(dims, slots=3, poll_steps=4, n_each=2, each=lim.ctypes.data_as(C.POINTER(C.c_int32)))
    rc, _ = _tokens_call(lib, dims, conds, texts, bad)
    assert rc == -1 and b"t3mi_generate_tokens" in lib.mock_last_error()
