"""chatterbox_amd -- MI355X-native T3 speech-token decoder behind chatterbox's
`t3_inference()` interface (hand-written gfx950 HIP kernels + a C-ABI library).

The T3 decode hot path and its "next" rows live here (prompt construction, batched and continuous
decoding, weight pipeline, S3Gen = conformer encoder + CFM flow decoder + HiFT vocoder); the rest of the reference
(tokenizers, voice encoders, download, HTTP front end) is out of scope."""
from .t3 import (SPEECH_VOCAB_SIZE, Session, T3Cond, T3Model, drop_invalid_tokens, inspect_checkpoint, load_t3, load_t3_turbo,  # noqa: F401
                 prepare_input_embeds, prompt_length, split_sentences, t3_cond, t3_inference, t3_inference_batch,
                 t3_inference_continuous, t3_inference_cpp,
                 t3_inference_traced, t3_inference_turbo, t3_inference_turbo_batch, tts_chunked)

from .s3gen import (CfmEstimator, HiftVocoder, S3GenFlow, causal_cfm_forward, load_cfm_estimator, load_hift_vocoder,  # noqa: F401,E402
                    load_s3gen_flow, s3gen_inference, synthesize, synthesize_one)  # noqa: F401,E402
from .pipeline import tts_pipelined, tts_stream  # noqa: F401,E402

__version__ = "0.1.0"
