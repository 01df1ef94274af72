"""CPU-only checks of the drop-in boundary: the C-ABI library loads, exports every symbol
include/t3mi.h declares, and fails loudly (no CPU fallback) without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from chatterbox_amd import _lib, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_and_binding_list_the_same_symbols():
    hdr = open(os.path.join(ROOT, "include", "t3mi.h")).read()
    declared = set(re.findall(r"\b(t3mi_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    for name in _lib.SYMBOLS:
        assert getattr(lib, name) is not None


def test_default_params_match_t3_inference_defaults():
    lib = _lib.load()
    p = _lib.Params()
    lib.t3mi_default_params(C.byref(p))
    # R/t3.R:461-463
    assert p.max_new_tokens == 1000
    assert abs(p.temperature - 0.8) < 1e-7 and abs(p.cfg_weight - 0.5) < 1e-7
    assert abs(p.top_p - 0.95) < 1e-7 and abs(p.min_p - 0.05) < 1e-7
    assert abs(p.repetition_penalty - 1.2) < 1e-7
    assert (p.start_token, p.stop_token, p.sampler) == (6561, 6562, _lib.SAMPLER_R)


def test_open_fails_loudly_without_gpu_or_file(tmp_path):
    lib = _lib.load()
    h = C.c_void_p()
    if lib.t3mi_device_count() == 0:
        path = synth.ensure_checkpoint(synth.TINY, 1234, str(tmp_path))
        rc = lib.t3mi_open(path.encode(), 0, 0, 0, C.byref(h))
        assert rc == _lib.ERR_HIP and b"no CPU fallback" in lib.t3mi_last_error()
    else:
        rc = lib.t3mi_open(b"/nonexistent/x.safetensors", 0, 0, 0, C.byref(h))
        assert rc == _lib.ERR_IO
    assert not h.value
    assert lib.t3mi_session_decode(None, 1) == _lib.ERR_ARG


def test_synthetic_checkpoint_layout_roundtrip(tmp_path):
    """The plain writer produces a file the `safetensors` package reads back bit-exactly
    with the reference's key names (R/t3.R:1065-1097, R/llama.R:566-607)."""
    st = pytest.importorskip("safetensors.numpy")
    w = synth.synth_llama_weights(synth.TINY, 5)
    p = synth.write_safetensors(w, str(tmp_path / "t.safetensors"))
    back = st.load_file(p)
    assert set(back) == set(w)
    for k in w:
        assert np.array_equal(back[k], w[k])
    assert "tfmr.layers.1.mlp.down_proj.weight" in w and "speech_pos_emb.emb.weight" in w


def test_sentence_split_uses_ascii_whitespace_like_R():
    """R/tts.R:462: strsplit(text, "(?<=[.!?])\\s+", perl = TRUE) -- PCRE without UCP: a no-break space or an em space after
    the full stop is not white space there."""
    from chatterbox_amd import split_sentences
    assert split_sentences("One.\u00a0Two. Three!\u2003Four?\tFive") == ["One.\u00a0Two.", "Three!\u2003Four?", "Five"]
    assert split_sentences("A. B.  ") == ["A.", "B."]
    assert split_sentences("") == []


def test_seed_default_follows_the_global_generator():
    """Without `seed` the wrappers draw one from numpy's global generator (the reference draws from torch's, R/t3.R:566):
    np.random.seed governs, consecutive calls differ; an explicit seed is passed through."""
    import numpy as np
    from chatterbox_amd import t3
    np.random.seed(5)
    a, b = t3._seed(None), t3._seed(None)
    np.random.seed(5)
    assert t3._seed(None) == a and a != b and t3._seed(17) == 17
