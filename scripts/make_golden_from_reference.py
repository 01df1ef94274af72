"""Extract the reference's own T3 fixtures (data only) into tests/golden/.

Run in the build container only (reads /root/reference/outputs/*.safetensors,
which are tensors dumped by the reference authors from upstream Python; SURVEY
§8c).  The GPU box has no /root/reference, so tests read the committed .npz.
"""
import os
import numpy as np
from safetensors.numpy import load_file

REF = "/root/reference/outputs"
OUT = os.path.join(os.path.dirname(__file__), "..", "tests", "golden")


def main():
    os.makedirs(OUT, exist_ok=True)
    dbg = load_file(os.path.join(REF, "t3_debug.safetensors"))
    ce = load_file(os.path.join(REF, "cond_enc_debug.safetensors"))
    ref = load_file(os.path.join(REF, "t3_reference.safetensors"))
    pg = load_file(os.path.join(REF, "python_generated.safetensors"))
    np.savez_compressed(
        os.path.join(OUT, "ref_t3_debug.npz"),
        raw_logits_step0=dbg["raw_logits_step0"],
        logits_after_cfg=dbg["logits_after_cfg"],
        logits_after_temp=dbg["logits_after_temp"],
        probs_step0=dbg["probs_step0"],
        sampled_tokens=dbg["sampled_tokens"],
    )
    np.savez_compressed(
        os.path.join(OUT, "ref_prompt_embeds.npz"),
        # realistic prefill inputs: [2,43,1024] (cond 34 + text 8 + BOS 1)
        input_embeds=ce["input_embeds"],
        len_cond=ce["len_cond"],
    )
    np.savez_compressed(
        os.path.join(OUT, "ref_tokens.npz"),
        text_tokens_with_sot_eot=ref["text_tokens_with_sot_eot"],
        speech_tokens=ref["speech_tokens"],
        python_generated=pg["generated_tokens"],
    )
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
