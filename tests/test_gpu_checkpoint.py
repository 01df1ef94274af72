"""Weight pipeline on the GPU: checkpoints stored as F16 and BF16 load through the native reader and give
the logits of the oracle run on the SAME converted weights (the conversion is the only difference from the F32 file);
the packed bf16 cache gives bit-identical results, is tied to its source file, and a damaged cache is replaced."""
import os
import time

import numpy as np
import pytest

import chatterbox_amd as cb
from chatterbox_amd import _lib, synth
from oracle import t3_oracle as O
from tests.helpers import oracle_cfg, rel_err

pytestmark = pytest.mark.gpu
LOGIT_TOL = 1e-3


def _first_logits(model, emb, n=3):
    with cb.Session(model, [emb], max_new_tokens=n + 1, top_p=0.0, mask_stop=True) as s:
        out = [s.logits()]
        s.decode(n)
        out.append(s.logits())
        toks = list(s.tokens(0))
    return out, toks


@pytest.mark.parametrize("dt,conv", [("F16", lambda a: a.astype(np.float16).astype(np.float32)), ("BF16", synth.bf16_round)],
                         ids=["F16", "BF16"])
def test_half_precision_checkpoints_load_like_their_converted_weights(tmp_path, dt, conv):
    dims = synth.TINY
    w = synth.synth_llama_weights(dims, 1234)
    path = synth.write_safetensors(w, str(tmp_path / f"tiny_{dt}.safetensors"), dt)
    wc = {k: conv(v) for k, v in w.items()}                       # what the file holds, as float32
    om = O.LlamaOracle(wc, oracle_cfg(dims))
    emb = synth.synth_embeds(2, 13, dims.hidden_size, 77)
    ref = O.t3_inference(om, emb, O.SampleParams(max_new_tokens=4, top_p=0.0), mask_stop=True)
    model = cb.load_t3(path, device=0, start_speech_token=dims.start_speech_token, stop_speech_token=dims.stop_speech_token)
    try:
        lg, toks = _first_logits(model, emb, 3)
        assert rel_err(lg[0], ref.raw_logits[0]) < LOGIT_TOL and rel_err(lg[1], ref.raw_logits[3]) < LOGIT_TOL
        assert toks[:3] == ref.tokens[:3]
        # the prompt tensors (text_emb / cond_enc.*) went through the same conversion
        rng = np.random.default_rng(1)
        spk = (rng.standard_normal(dims.speaker_embed) * 0.1).astype(np.float32)
        ctok, text = rng.integers(0, dims.start_speech_token, 30), rng.integers(1, 250, 7)
        want, _ = O.prepare_input_embeds(wc, spk, ctok, 0.5, text, 0.5, dims.start_speech_token)
        got, _ = cb.prepare_input_embeds(model, cb.t3_cond(spk, ctok, emotion_adv=0.5), text, cfg_weight=0.5)
        assert np.abs(got - want).max() < 1e-4 * np.abs(want).max()
    finally:
        model.close()
    # an F32 file and a BF16 file of the same weights give the same bf16-storage model, bit for bit
    if dt == "BF16":
        p32 = synth.write_safetensors(w, str(tmp_path / "tiny_F32.safetensors"))
        a = cb.load_t3(p32, device=0, dtype="bf16", start_speech_token=dims.start_speech_token, stop_speech_token=dims.stop_speech_token)
        b = cb.load_t3(path, device=0, dtype="bf16", start_speech_token=dims.start_speech_token, stop_speech_token=dims.stop_speech_token)
        try:
            # (norms / embeddings stay fp32 in the F32 file and are bf16-rounded in the BF16 file: compare with those equalised)
            la, _ = _first_logits(a, emb, 2)
            lb, _ = _first_logits(b, emb, 2)
            assert rel_err(lb[0], la[0]) < 3e-2
        finally:
            a.close()
            b.close()


@pytest.mark.parametrize("variant", ["llama", "turbo"])
def test_packed_bf16_cache(tmp_path, variant):
    if variant == "llama":
        dims = synth.TINY
        src = synth.write_safetensors(synth.synth_llama_weights(dims, 1234), str(tmp_path / "t3.safetensors"))
        emb = synth.synth_embeds(2, 11, dims.hidden_size, 5)
        load = lambda **kw: cb.load_t3(src, device=0, dtype="bf16", start_speech_token=dims.start_speech_token,   # noqa: E731
                                       stop_speech_token=dims.stop_speech_token, **kw)
    else:
        dims = synth.TURBO_TINY
        src = synth.write_safetensors(synth.synth_turbo_weights(dims, 4321), str(tmp_path / "t3.safetensors"))
        emb = synth.synth_embeds(1, 11, dims.hidden_size, 5)
        load = lambda **kw: cb.load_t3_turbo(src, device=0, dtype="bf16", start_speech_token=dims.start_speech_token,   # noqa: E731
                                             stop_speech_token=dims.stop_speech_token, **kw)
    cache = src + ".t3mi-bf16"

    def run(**kw):
        m = load(**kw)
        try:
            lg, toks = _first_logits(m, emb, 3)
            return m.info.cache_state, lg, toks
        finally:
            m.close()

    st0, base, tk0 = run()                                   # no cache
    assert st0 == _lib.CACHE_NONE and not os.path.exists(cache)
    st1, a, tk1 = run(cache=True)                            # written
    assert st1 == _lib.CACHE_WRITTEN and os.path.getsize(cache) > 64
    st2, b, tk2 = run(cache=True)                            # used
    assert st2 == _lib.CACHE_USED
    for x in (a, b):
        assert np.array_equal(x[0], base[0]) and np.array_equal(x[1], base[1])
    assert tk0 == tk1 == tk2
    # fp32 storage never touches the cache
    m = (cb.load_t3 if variant == "llama" else cb.load_t3_turbo)(src, device=0, dtype="f32", cache=True,
                                                                 start_speech_token=dims.start_speech_token,
                                                                 stop_speech_token=dims.stop_speech_token)
    assert m.info.cache_state == _lib.CACHE_NONE
    m.close()
    # damaged payload (one flipped byte): checksum fails, the file is replaced, results unchanged
    raw = bytearray(open(cache, "rb").read())
    raw[len(raw) // 2] ^= 0x40
    open(cache, "wb").write(bytes(raw))
    st3, c, _ = run(cache=True)
    assert st3 == _lib.CACHE_REPLACED and np.array_equal(c[1], base[1])
    assert run(cache=True)[0] == _lib.CACHE_USED
    # truncated cache
    open(cache, "wb").write(open(cache, "rb").read()[:-100])
    assert run(cache=True)[0] == _lib.CACHE_REPLACED
    # the source changed (new mtime): the cache is stale
    later = time.time() + 5
    os.utime(src, (later, later))
    assert run(cache=True)[0] == _lib.CACHE_REPLACED
    assert run(cache=True)[0] == _lib.CACHE_USED
    # a same-architecture fine-tune copied over the checkpoint with size AND mtime preserved (cp -p): only the content
    # fingerprint in the header can tell -- the cache is stale, the new weights are the ones that run
    st_src = os.stat(src)
    w2 = synth.synth_llama_weights(dims, 4242) if variant == "llama" else synth.synth_turbo_weights(dims, 4242)
    synth.write_safetensors(w2, src)
    os.utime(src, ns=(st_src.st_atime_ns, st_src.st_mtime_ns))
    assert os.stat(src).st_size == st_src.st_size and os.stat(src).st_mtime_ns == st_src.st_mtime_ns
    st6, f, _ = run(cache=True)
    assert st6 == _lib.CACHE_REPLACED and not np.array_equal(f[1], base[1])
    assert run(cache=True)[0] == _lib.CACHE_USED
    base = f                                                 # the checkpoint now holds the second set of weights
    # a cache of ANOTHER checkpoint under this name is refused
    other = synth.write_safetensors(synth.synth_llama_weights(synth.TINY, 99) if variant == "llama" else
                                    synth.synth_turbo_weights(synth.TURBO_TINY, 99), str(tmp_path / "other.safetensors"))
    mo = (cb.load_t3 if variant == "llama" else cb.load_t3_turbo)(other, device=0, dtype="bf16", cache=cache,
                                                                  start_speech_token=dims.start_speech_token,
                                                                  stop_speech_token=dims.stop_speech_token)
    assert mo.info.cache_state == _lib.CACHE_REPLACED
    mo.close()
    # an unwritable cache path is not an error
    st5, e, _ = run(cache=str(tmp_path / "no_such_dir" / "x.bin"))
    assert st5 == _lib.CACHE_NONE and np.array_equal(e[1], base[1])


def test_open_rejects_bad_files_before_touching_the_gpu(tmp_path):
    p = str(tmp_path / "trunc.safetensors")
    good = synth.write_safetensors(synth.synth_llama_weights(synth.TINY, 1234), str(tmp_path / "good.safetensors"))
    open(p, "wb").write(open(good, "rb").read()[:200000])
    with pytest.raises(_lib.T3miError) as e:
        cb.load_t3(p, device=0)
    assert e.value.code == _lib.ERR_IO
    with pytest.raises(_lib.T3miError) as e:
        cb.load_t3_turbo(good, device=0)                     # a Llama checkpoint opened as Turbo
    assert e.value.code == _lib.ERR_IO
