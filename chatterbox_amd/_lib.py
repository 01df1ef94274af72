"""ctypes binding of libt3mi.so (include/t3mi.h).  The library is built in-tree
(`make -C chatterbox_amd/csrc`, or `__graft_entry__.build()`); there is no CPU
fallback: a missing library or a missing GPU is a loud error."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("T3MI_LIB", os.path.join(_HERE, "libt3mi.so"))   # T3MI_LIB: an alternative build (experiments)

OK = 0
ERR_ARG, ERR_IO, ERR_HIP, ERR_STATE = -1, -2, -3, -4
WARN_CACHE_FULL = 1
VARIANT_LLAMA, VARIANT_TURBO = 0, 1
W_F32, W_BF16 = 0, 1
CACHE_NONE, CACHE_WRITTEN, CACHE_USED, CACHE_REPLACED = 0, 1, 2, 3
SAMPLER_R, SAMPLER_CPP, SAMPLER_TURBO = 0, 1, 2


class Params(C.Structure):
    _fields_ = [
        ("max_new_tokens", C.c_int32),
        ("temperature", C.c_float),
        ("cfg_weight", C.c_float),
        ("top_p", C.c_float),
        ("min_p", C.c_float),
        ("repetition_penalty", C.c_float),
        ("start_token", C.c_int32),
        ("stop_token", C.c_int32),
        ("sampler", C.c_int32),
        ("top_k", C.c_int32),
        ("max_cache_len", C.c_int32),
        ("mask_stop", C.c_int32),
        ("seed", C.c_uint64),
        ("first_utterance_id", C.c_int64),
        ("drop_invalid_from", C.c_int32),
        ("reserved_", C.c_int32),
    ]


class Info(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("variant", "n_layers", "d_model", "n_heads", "head_dim", "d_ff", "vocab", "max_pos",
                 "wdtype", "device")] + [("weight_bytes", C.c_int64), ("cache_state", C.c_int32),
                                         ("has_prompt_tensors", C.c_int32)]


class Cond(C.Structure):
    """t3mi_cond: the conditioning t3_inference() receives (R/t3.R:103-113)."""
    _fields_ = [("speaker_emb", C.POINTER(C.c_float)), ("n_speaker", C.c_int32),
                ("cond_prompt_speech_tokens", C.POINTER(C.c_int32)), ("n_cond_tokens", C.c_int32),
                ("emotion_adv", C.c_float), ("has_emotion_adv", C.c_int32)]


class CfmInfo(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("mel", "hidden", "n_mid", "n_tfm", "heads", "head_dim", "d_ff", "time_in", "time_dim",
                                         "meanflow", "device", "reserved_")] + [("weight_bytes", C.c_int64)]


class FlowInfo(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("vocab", "d_model", "heads", "d_ff", "n_blocks", "n_up_blocks", "spk_embed_dim", "mel")] + \
               [("weight_bytes", C.c_int64)]


class HiftInfo(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("mel", "base_channels", "f0_channels", "n_harmonics", "samples_per_frame", "sampling_rate",
                                         "device", "reserved_")] + [("weight_bytes", C.c_int64)]


class KStat(C.Structure):
    _fields_ = [("name", C.c_char * 32), ("launches", C.c_int32), ("total_us", C.c_float),
                ("bytes", C.c_double)]


# every symbol include/t3mi.h declares: (restype, argtypes)
_FP = C.POINTER(C.c_float)
_IP = C.POINTER(C.c_int32)
_H = C.c_void_p
SYMBOLS = {
    "t3mi_default_params": (None, [C.POINTER(Params)]),
    "t3mi_last_error": (C.c_char_p, []),
    "t3mi_device_count": (C.c_int, []),
    "t3mi_open": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(_H)]),
    "t3mi_open_cached": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_char_p, C.POINTER(_H)]),
    "t3mi_inspect": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(Info)]),
    "t3mi_close": (C.c_int, [_H]),
    "t3mi_get_info": (C.c_int, [_H, C.POINTER(Info)]),
    "t3mi_generate": (C.c_int, [_H, _FP, C.c_int, C.c_int, C.POINTER(Params), _IP, C.c_int32, _IP]),
    "t3mi_generate_batch": (C.c_int, [_H, C.c_int, C.c_int, C.POINTER(C.c_void_p), _IP, C.POINTER(Params),
                                      _IP, C.c_int32, _IP]),
    "t3mi_prompt_shape": (C.c_int, [_H, C.POINTER(Cond), C.c_int, C.c_float, _IP, _IP, _IP]),
    "t3mi_prepare_input_embeds": (C.c_int, [_H, C.POINTER(Cond), _IP, C.c_int, C.c_float, C.c_int32, _FP, C.c_size_t]),
    "t3mi_prepare_input_embeds_static": (C.c_int, [_H, C.POINTER(Cond), _IP, C.c_int, C.c_float, C.c_int32, C.c_void_p, C.c_size_t]),
    "t3mi_generate_tokens": (C.c_int, [_H, C.POINTER(Cond), _IP, C.c_int, C.POINTER(Params), _IP, C.c_int32, _IP]),
    "t3mi_generate_batch_tokens": (C.c_int, [_H, C.c_int, C.POINTER(Cond), C.POINTER(_IP), _IP, C.POINTER(Params),
                                             _IP, C.c_int32, _IP]),
    "t3mi_session_begin_tokens": (C.c_int, [_H, C.c_int, C.POINTER(Cond), C.POINTER(_IP), _IP, C.POINTER(Params)]),
    "t3mi_session_begin": (C.c_int, [_H, C.c_int, C.c_int, C.POINTER(C.c_void_p), _IP, C.POINTER(Params),
                                     C.c_int]),
    "t3mi_session_force": (C.c_int, [_H, _IP, C.c_int]),
    "t3mi_generate_continuous": (C.c_int, [_H, C.c_int, C.c_int, C.POINTER(C.c_void_p), _IP, _IP, C.c_int, C.c_int, C.c_int,
                                           C.POINTER(Params), _IP, C.c_int32, _IP, _IP]),
    "t3mi_generate_continuous_tokens": (C.c_int, [_H, C.c_int, C.POINTER(Cond), C.POINTER(_IP), _IP, _IP, C.c_int, C.c_int, C.c_int,
                                                  C.POINTER(Params), _IP, C.c_int32, _IP, _IP]),
    "t3mi_session_refill": (C.c_int, [_H, C.c_int, _IP, C.POINTER(C.c_void_p), _IP, _IP, C.POINTER(C.c_int64), C.c_int]),
    "t3mi_session_refill_tokens": (C.c_int, [_H, C.c_int, _IP, C.POINTER(Cond), C.POINTER(_IP), _IP, _IP, C.POINTER(C.c_int64)]),
    "t3mi_session_decode": (C.c_int, [_H, C.c_int]),
    "t3mi_session_decode_timed": (C.c_int, [_H, C.c_int, _FP]),
    "t3mi_session_sync": (C.c_int, [_H]),
    "t3mi_session_status": (C.c_int, [_H, _IP, _IP]),
    "t3mi_session_tokens": (C.c_int, [_H, C.c_int, _IP, C.c_int32, _IP]),
    "t3mi_session_logits": (C.c_int, [_H, _FP]),
    "t3mi_session_set_logits": (C.c_int, [_H, _FP]),
    "t3mi_session_keep_filtered": (C.c_int, [_H]),
    "t3mi_session_filtered_logits": (C.c_int, [_H, C.c_int, _FP]),
    "t3mi_session_dist": (C.c_int, [_H, C.c_int, _IP, _FP, C.c_int32, _IP]),
    "t3mi_session_drawn": (C.c_int, [_H, _IP]),
    "t3mi_session_profile_step": (C.c_int, [_H, C.POINTER(KStat), C.c_int, C.POINTER(C.c_int)]),
    "t3mi_session_end": (C.c_int, [_H]),
    "t3mi_stream": (C.c_void_p, [_H]),
    "t3mi_cfm_open": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.POINTER(_H)]),
    "t3mi_cfm_close": (C.c_int, [_H]),
    "t3mi_cfm_get_info": (C.c_int, [_H, C.POINTER(CfmInfo)]),
    "t3mi_cfm_estimator": (C.c_int, [_H, C.c_int, C.c_int, _FP, _FP, _FP, _FP, _FP, _FP, _FP, _FP]),
    "t3mi_flow_open": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(_H)]),
    "t3mi_flow_get_info": (C.c_int, [_H, C.POINTER(FlowInfo)]),
    "t3mi_flow_encoder": (C.c_int, [_H, C.c_int, _FP, _FP]),
    "t3mi_flow_inference": (C.c_int, [_H, _IP, C.c_int, _IP, C.c_int, _FP, C.c_int, _FP, _FP, C.c_int, C.c_int, C.c_double, C.c_int,
                                      C.c_int, _FP, C.c_int, C.POINTER(C.c_int), _FP]),
    "t3mi_hift_open": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int, C.POINTER(_H)]),
    "t3mi_hift_close": (C.c_int, [_H]),
    "t3mi_hift_get_info": (C.c_int, [_H, C.POINTER(HiftInfo)]),
    "t3mi_hift_inference": (C.c_int, [_H, C.c_int, _FP, _FP, _FP, _FP, C.c_int, _FP, _FP, _FP, _FP]),
    "t3mi_cfm_solve": (C.c_int, [_H, C.c_int, _FP, _FP, _FP, _FP, _FP, C.c_int, C.c_double, C.c_int, _FP, _FP]),
}

_lib = None


class T3miError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"t3mi error {code}: {msg}")
        self.code = code


def load() -> C.CDLL:
    """Load libt3mi.so and bind every declared symbol.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make -C chatterbox_amd/csrc` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the T3 path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int) -> int:
    if rc < 0:
        raise T3miError(rc, load().t3mi_last_error().decode("utf-8", "replace"))
    return rc
