"""t3mi_prepare_input_embeds_static: the entry point that allocates nothing (working memory in a file-scope
`static __device__` array, its per-GPU address taken with hipGetSymbolAddress at t3mi_open).  Called through the C ABI
and checked against the existing entry point t3mi_prepare_input_embeds, bit for bit (same kernels, same order), on two
calls in a row -- the second with a smaller input, so nothing of the first call's contents may leak into it -- and a
third, larger again; inputs beyond the stated maximum are rejected with a clear error."""
import ctypes as C

import numpy as np
import pytest

import chatterbox_amd as cb
from chatterbox_amd import _lib, synth

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _static(model, cond, text, cfg_weight):
    tt = np.ascontiguousarray(text, dtype=np.int32).reshape(-1)
    cs = cond.c_struct()
    rows, s0, lc = C.c_int32(0), C.c_int32(0), C.c_int32(0)
    _lib.check(model._lib.t3mi_prompt_shape(model._h, C.byref(cs), tt.size, float(cfg_weight), C.byref(rows), C.byref(s0),
                                            C.byref(lc)))
    out = torch.full((rows.value, s0.value, model.d_model), float("nan"), dtype=torch.float32, device="cuda:0")
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info(0)             # hipMemGetInfo: the device's free bytes, whoever allocated
    rc = model._lib.t3mi_prepare_input_embeds_static(model._h, C.byref(cs), tt.ctypes.data_as(C.POINTER(C.c_int32)), tt.size,
                                                     float(cfg_weight), model.start_speech_token, C.c_void_p(out.data_ptr()),
                                                     out.numel())
    free1, _ = torch.cuda.mem_get_info(0)
    _lib.check(rc)
    assert free1 == free0, f"the static entry point changed the device's free memory by {free0 - free1} bytes"
    torch.cuda.synchronize()
    return out.cpu().numpy()


@pytest.mark.parametrize("dims", [synth.TINY, synth.FULL], ids=["tiny", "full"])
def test_static_entry_point_matches_the_allocating_one(dims):
    model = cb.load_t3(synth.ensure_checkpoint(dims, 1234), device=0, start_speech_token=dims.start_speech_token,
                       stop_speech_token=dims.stop_speech_token)
    try:
        rng = np.random.default_rng(17)
        spk = (rng.standard_normal(dims.speaker_embed) * 0.1).astype(np.float32)
        calls = [(150, 40, 0.5), (37, 5, 0.5), (90, 21, 0.0)]      # (conditioning tokens, text tokens, cfg_weight): big, smaller, other
        for n_cond, n_text, cfgw in calls:
            cond = cb.t3_cond(spk, rng.integers(0, dims.start_speech_token, n_cond), emotion_adv=0.3)
            text = rng.integers(1, 250, n_text)
            want, _ = cb.prepare_input_embeds(model, cond, text, cfg_weight=cfgw)
            got = _static(model, cond, text, cfgw)
            assert got.shape == want.shape and np.isfinite(got).all()
            assert np.array_equal(got, want), f"{n_cond} / {n_text}: max diff {np.abs(got - want).max()}"
        if dims is synth.FULL:
            # beyond the stated maximum (3990 + 200 + 3 ids > 4096; 16 M scratch floats > 2 M): rejected with the numbers
            cond = cb.t3_cond(spk, rng.integers(0, dims.start_speech_token, 3990), emotion_adv=0.3)
            with pytest.raises(_lib.T3miError) as e:
                _static(model, cond, rng.integers(1, 250, 200), 0.5)
            assert e.value.code == _lib.ERR_ARG and "exceed the fixed working memory" in str(e.value)
            # a HOST pointer in the device-buffer slot (the sibling entry point takes one there) is refused, not dereferenced
            host = np.zeros((2, 64, dims.hidden_size), dtype=np.float32)
            small = cb.t3_cond(spk, rng.integers(0, dims.start_speech_token, 10), emotion_adv=0.3)
            cs = small.c_struct()
            t5 = np.arange(1, 6, dtype=np.int32)
            rc = model._lib.t3mi_prepare_input_embeds_static(model._h, C.byref(cs), t5.ctypes.data_as(C.POINTER(C.c_int32)), 5, 0.5,
                                                             model.start_speech_token, C.c_void_p(host.ctypes.data), host.size)
            assert rc == _lib.ERR_ARG and b"device memory" in model._lib.t3mi_last_error()
            # ... and the allocating entry point still takes it
            want, _ = cb.prepare_input_embeds(model, cond, rng.integers(1, 250, 200), cfg_weight=0.5)
            assert np.isfinite(want).all()
    finally:
        model.close()
