"""Shared helpers for the parity tests (oracle configuration from synthetic dims)."""
import numpy as np

from chatterbox_amd import synth
from oracle import t3_oracle as O


def oracle_cfg(dims: synth.T3Dims) -> O.LlamaConfig:
    return O.LlamaConfig(hidden_size=dims.hidden_size, intermediate_size=dims.intermediate_size,
                         num_hidden_layers=dims.num_hidden_layers,
                         num_attention_heads=dims.num_attention_heads, head_dim=dims.head_dim,
                         speech_vocab=dims.speech_vocab, max_speech_pos=dims.max_speech_pos,
                         start_speech_token=dims.start_speech_token,
                         stop_speech_token=dims.stop_speech_token)


def oracle_model(dims: synth.T3Dims, seed: int = 1234) -> O.LlamaOracle:
    return O.LlamaOracle(synth.synth_llama_weights(dims, seed), oracle_cfg(dims))


def rel_err(a: np.ndarray, b: np.ndarray) -> float:
    """max |a-b| / max |b|  (the 1e-3 fp32 bar of BASELINE.json is applied to this)."""
    return float(np.abs(a.astype(np.float64) - b.astype(np.float64)).max() / np.abs(b).max())
