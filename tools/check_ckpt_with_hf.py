#!/usr/bin/env python3
"""Build-container / CPU only (imports `transformers`): load checkpoints SAVED BY THE ENGINE into HuggingFace
Transformers and check that the on-disk contract of the drop-in holds in that direction too (N2, SURVEY.md §8f;
the reference's `load_saved` and the ASR pipeline read these files with `from_pretrained`,
R/src/coral/wav2vec2.py:253-305, R/src/coral/whisper.py:234-267, R/src/coral/evaluate.py:145-155):

  * every tensor name and shape matches the HF module's state dict (no missing / unexpected keys),
  * weight-norm parameters arrive as `parametrizations.weight.original0 / original1`,
  * Whisper's `proj_out` is tied to `embed_tokens` after loading,
  * HF's fp32 forward on the saved input reproduces the logits / loss the engine computed in bf16.

    python tools/check_ckpt_with_hf.py [tests/golden]
"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]


def check_w2v2(d: Path, z) -> dict:
    from safetensors.torch import load_file
    from transformers import Wav2Vec2ForCTC

    model, info = Wav2Vec2ForCTC.from_pretrained(str(d), output_loading_info=True, attn_implementation="eager")
    assert not info["missing_keys"] and not info["unexpected_keys"] and not info["mismatched_keys"], info
    sd_file = load_file(str(d / "model.safetensors"))
    sd_hf = model.state_dict()
    assert set(sd_file) == set(sd_hf), (set(sd_file) ^ set(sd_hf))
    for k, v in sd_file.items():
        assert tuple(v.shape) == tuple(sd_hf[k].shape), k
        assert torch.equal(v.float(), sd_hf[k].float()), k
    wn = [k for k in sd_file if "parametrizations.weight.original" in k]
    assert len(wn) == 2 and not any(k.endswith("weight_g") or k.endswith("weight_v") for k in sd_file)
    model.eval()
    with torch.no_grad():
        out = model(input_values=torch.from_numpy(z["input_values"]), attention_mask=torch.from_numpy(z["attention_mask"]),
                    labels=torch.from_numpy(z["labels"]))
    err = float((out.logits - torch.from_numpy(z["logits"])).abs().max())
    rel = abs(float(out.loss) - float(z["loss"])) / float(z["loss"])
    assert err <= 5e-2 and rel <= 5e-3, (err, rel)   # bf16 engine vs HF fp32; logits of magnitude ~3 here
    return dict(tensors=len(sd_file), logits_max_abs_err=err, loss_rel_err=rel)


def check_whisper(d: Path, z) -> dict:
    from safetensors.torch import load_file
    from transformers import WhisperForConditionalGeneration

    model, info = WhisperForConditionalGeneration.from_pretrained(str(d), output_loading_info=True,
                                                                   attn_implementation="eager")
    missing = [k for k in info["missing_keys"] if k != "proj_out.weight"]  # tied: never stored
    assert not missing and not info["unexpected_keys"] and not info["mismatched_keys"], info
    assert model.proj_out.weight.data_ptr() == model.model.decoder.embed_tokens.weight.data_ptr()  # tied head
    sd_file = load_file(str(d / "model.safetensors"))
    sd_hf = model.state_dict()
    for k, v in sd_file.items():
        kk = k if k in sd_hf else "model." + k
        assert tuple(v.shape) == tuple(sd_hf[kk].shape), k
    g = torch.Generator().manual_seed(int(z["feats_seed"]))
    for n in (4000, 3300):
        torch.randn(n, generator=g)
    feats = torch.randn(2, 80, 3000, generator=g) * 0.5
    assert np.array_equal(feats[:, :, ::100].numpy(), z["feats_slice"])
    model.eval()
    with torch.no_grad():
        out = model(input_features=feats, labels=torch.from_numpy(z["labels"]))
    err = float((out.logits - torch.from_numpy(z["logits"])).abs().max())
    rel = abs(float(out.loss) - float(z["loss"])) / float(z["loss"])
    assert err <= 5e-2 and rel <= 1e-2, (err, rel)
    return dict(tensors=len(sd_file), logits_max_abs_err=err, loss_rel_err=rel)


def main(root):
    root = Path(root)
    res = {}
    if (root / "engine_ckpt_w2v2").exists():
        res["wav2vec2"] = check_w2v2(root / "engine_ckpt_w2v2", np.load(root / "engine_ckpt_w2v2.npz"))
    if (root / "engine_ckpt_whisper").exists():
        res["whisper"] = check_whisper(root / "engine_ckpt_whisper", np.load(root / "engine_ckpt_whisper.npz"))
    print(res)
    return res


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else ROOT / "tests" / "golden")
