"""Weight pipeline on CPU: t3mi_inspect validates a checkpoint without a GPU -- header, every tensor name /
shape / dtype the loaders read (R/t3.R:1065-1145, R/llama.R:566-607; Turbo R/t3.R:1508-1549) -- and rejects truncated,
mis-shaped and foreign files with T3MI_ERR_IO instead of reading out of bounds."""
import json
import os
import struct

import numpy as np
import pytest

import chatterbox_amd as cb
from chatterbox_amd import _lib, synth


@pytest.fixture(scope="module")
def tiny_files(tmp_path_factory):
    d = tmp_path_factory.mktemp("ckpt")
    w = synth.synth_llama_weights(synth.TINY, 1234)
    paths = {dt: synth.write_safetensors(w, str(d / f"tiny_{dt}.safetensors"), dt) for dt in ("F32", "F16", "BF16")}
    tw = synth.synth_turbo_weights(synth.TURBO_TINY, 4321)
    paths["turbo"] = synth.write_safetensors(tw, str(d / "turbo.safetensors"))
    return d, w, paths


def _rewrite(src, dst, mutate):
    """Copy a safetensors file with its JSON header passed through `mutate(header) -> header`."""
    raw = open(src, "rb").read()
    n = struct.unpack("<Q", raw[:8])[0]
    hdr = mutate(json.loads(raw[8:8 + n]))
    hj = json.dumps(hdr, separators=(",", ":")).encode()
    hj += b" " * ((8 - len(hj) % 8) % 8)
    open(dst, "wb").write(struct.pack("<Q", len(hj)) + hj + raw[8 + n:])
    return dst


@pytest.mark.parametrize("dt", ["F32", "F16", "BF16"])
def test_inspect_reads_the_dimensions(tiny_files, dt):
    _, _, paths = tiny_files
    dims = synth.TINY
    info = cb.inspect_checkpoint(paths[dt])
    assert (info["n_layers"], info["d_model"], info["n_heads"], info["d_ff"], info["vocab"], info["max_pos"]) == (
        dims.num_hidden_layers, dims.hidden_size, dims.num_attention_heads, dims.intermediate_size, dims.speech_vocab,
        dims.max_speech_pos)
    assert info["has_prompt_tensors"] == 1 and info["device"] == -1


def test_inspect_turbo_and_variant_mismatch(tiny_files):
    _, _, paths = tiny_files
    t = synth.TURBO_TINY
    info = cb.inspect_checkpoint(paths["turbo"], _lib.VARIANT_TURBO)
    assert (info["n_layers"], info["d_model"], info["vocab"]) == (t.n_layer, t.hidden_size, t.speech_vocab)
    for path, variant in ((paths["turbo"], _lib.VARIANT_LLAMA), (paths["F32"], _lib.VARIANT_TURBO)):
        with pytest.raises(_lib.T3miError) as e:
            cb.inspect_checkpoint(path, variant)
        assert e.value.code == _lib.ERR_IO and "lacks" in str(e.value)


def test_truncated_and_damaged_files_are_rejected(tiny_files):
    d, _, paths = tiny_files
    raw = open(paths["F32"], "rb").read()
    n = struct.unpack("<Q", raw[:8])[0]
    cases = {
        "empty": b"",
        "seven_bytes": raw[:7],
        "header_only": raw[:8 + n],                                   # every data_offsets range is beyond the file
        "half": raw[:len(raw) // 2],
        "minus_one_byte": raw[:-1],
        "huge_header_len": struct.pack("<Q", 1 << 40) + raw[8:],
        "zero_header_len": struct.pack("<Q", 0) + raw[8:],
        "not_json": raw[:8] + b"\xff" * n + raw[8 + n:],
        "unterminated": raw[:8] + raw[8:8 + n].rstrip(b" ")[:-1].ljust(n, b" ") + raw[8 + n:],   # closing brace cut off
    }
    for name, blob in cases.items():
        p = str(d / f"bad_{name}.safetensors")
        open(p, "wb").write(blob)
        with pytest.raises(_lib.T3miError) as e:
            cb.inspect_checkpoint(p)
        assert e.value.code == _lib.ERR_IO, name
    with pytest.raises(_lib.T3miError) as e:
        cb.inspect_checkpoint(str(d / "does_not_exist.safetensors"))
    assert e.value.code == _lib.ERR_IO


def test_mis_shaped_and_foreign_tensors_are_rejected(tiny_files):
    d, _, paths = tiny_files
    key = "tfmr.layers.1.mlp.down_proj.weight"

    def expect(mutate, needle):
        p = _rewrite(paths["F32"], str(d / "mut.safetensors"), mutate)
        with pytest.raises(_lib.T3miError) as e:
            cb.inspect_checkpoint(p)
        assert e.value.code == _lib.ERR_IO and needle in str(e.value), str(e.value)

    def swap_shape(h):
        h[key]["shape"] = h[key]["shape"][::-1]        # [in, out] instead of [out, in]: same byte size, wrong layout
        return h
    expect(swap_shape, key)

    def drop(h):
        del h["tfmr.layers.0.self_attn.k_proj.weight"]
        return h
    expect(drop, "k_proj")

    def int_dtype(h):
        h["tfmr.norm.weight"]["dtype"] = "I32"
        return h
    expect(int_dtype, "tfmr.norm.weight")

    def grow(h):
        h[key]["shape"][0] += 1                         # shape no longer matches the byte range
        return h
    expect(grow, key)

    def negative(h):
        h[key]["shape"][0] = -h[key]["shape"][0]
        return h
    expect(negative, "negative")

    def overflow(h):
        h[key]["shape"] = [1 << 40, 1 << 40]
        return h
    expect(overflow, key)

    def out_of_range(h):
        o = h[key]["data_offsets"]
        h[key]["data_offsets"] = [o[0] + (1 << 40), o[1] + (1 << 40)]
        return h
    expect(out_of_range, key)

    def bad_prompt(h):
        h["cond_enc.perceiver.attn.to_q.weight"]["shape"] = [1, h["cond_enc.perceiver.attn.to_q.weight"]["shape"][0] ** 2]
        return h
    expect(bad_prompt, "to_q")


def test_writer_round_trip_against_numpy(tiny_files):
    """The F16 / BF16 test files hold what numpy says they hold (the conversions the reader undoes)."""
    _, w, paths = tiny_files
    k = "tfmr.layers.0.self_attn.q_proj.weight"
    for dt, conv in (("F16", lambda a: a.astype(np.float16).astype(np.float32)), ("BF16", synth.bf16_round)):
        raw = open(paths[dt], "rb").read()
        n = struct.unpack("<Q", raw[:8])[0]
        h = json.loads(raw[8:8 + n])[k]
        blob = raw[8 + n + h["data_offsets"][0]:8 + n + h["data_offsets"][1]]
        if dt == "F16":
            got = np.frombuffer(blob, np.float16).astype(np.float32)
        else:
            got = (np.frombuffer(blob, np.uint16).astype(np.uint32) << 16).view(np.float32)
        assert np.array_equal(got.reshape(w[k].shape), conv(w[k]))
    x = np.array([1.0, 1.00390625, 1.01171875, -3.14159, 1e-40, np.inf], np.float32)   # ties go to even
    assert np.array_equal(synth.to_bf16_bits(x)[:3], np.array([0x3F80, 0x3F80, 0x3F82], np.uint16))
