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


def oracle_model(dims: synth.T3Dims, seed: int = 1234, head_scale: float = 1.0) -> O.LlamaOracle:
    return O.LlamaOracle(synth.synth_llama_weights(dims, seed, head_scale), oracle_cfg(dims))


def assert_dist_close(ids, probs, tail, rtol=2e-3):
    """Filtered distribution of one step: GPU (ids, probs) vs the oracle's TailResult.

    fp32 rounding differs between the two sides, so (a) two tokens whose probabilities agree to
    ~1e-6 relative may swap places and (b) the token AT the top-p / min-p cut may fall on
    either side.  Everything else must agree: same ids apart from <= 2 cut-line tokens, the
    same probability per id (rtol), GPU order non-increasing."""
    ids = [int(i) for i in ids]
    oid = [int(i) for i in tail.sorted_idx]
    pg = dict(zip(ids, [float(x) for x in probs]))
    po = dict(zip(oid, [float(x) for x in tail.sorted_probs]))
    assert len(pg) == len(ids), "duplicate ids in the GPU support"
    diff = set(pg) ^ set(po)
    assert len(diff) <= 2, f"supports differ by {len(diff)} tokens"
    common = set(pg) & set(po)
    pmin = min(po[i] for i in common)
    for i in diff:                       # only tokens at the cut line may differ
        p = pg.get(i, po.get(i))
        assert p <= pmin * (1 + 1e-3) + 1e-9, f"token {i} (p={p}) differs but is not at the cut"
    tol = rtol if not diff else rtol + 2 * max(pg.get(i, po.get(i)) for i in diff)
    for i in common:
        assert abs(pg[i] - po[i]) <= tol * po[i] + 1e-7, (i, pg[i], po[i])
    pr = np.asarray(probs, dtype=np.float64)
    assert np.all(pr[:-1] >= pr[1:] * (1 - 1e-6)), "GPU support is not sorted by probability"
    assert abs(pr.sum() - 1.0) < 1e-4
    return not diff      # True when the supports are identical (the draw is then comparable)


def draw_is_consistent(drawn: int, tail) -> bool:
    """The GPU's draw equals the oracle's, or the uniform target sits within float rounding of a
    CDF step (then either neighbour is a correct inverse-CDF answer)."""
    if drawn == tail.token:
        return True
    cum = tail.cum.astype(np.float64)
    near = np.abs(cum - tail.target) <= 4e-6 * max(float(cum[-1]), 1e-30)
    return bool(near.any())


def rel_err(a: np.ndarray, b: np.ndarray) -> float:
    """max |a-b| / max |b|  (the 1e-3 fp32 bar of BASELINE.json is applied to this)."""
    return float(np.abs(a.astype(np.float64) - b.astype(np.float64)).max() / np.abs(b).max())


def turbo_cfg(dims: synth.TurboDims) -> O.TurboConfig:
    return O.TurboConfig(hidden_size=dims.hidden_size, n_head=dims.n_head, n_layer=dims.n_layer,
                         speech_vocab=dims.speech_vocab, start_speech_token=dims.start_speech_token,
                         stop_speech_token=dims.stop_speech_token)


def turbo_oracle(dims: synth.TurboDims, seed: int = 4321, head_scale: float = 1.0) -> O.GPT2Oracle:
    return O.GPT2Oracle(synth.synth_turbo_weights(dims, seed, head_scale), turbo_cfg(dims))
