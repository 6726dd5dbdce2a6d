#!/usr/bin/env python3
"""Runs on the GPU box: save a tiny wav2vec2 and a tiny Whisper checkpoint WITH THE ENGINE (`save_pretrained`) after
one optimiser step each, together with the engine's outputs on a fixed input.  The directories are committed under
tests/golden/engine_ckpt_* and checked from the transformers side by tools/check_ckpt_with_hf.py (N2).

    gpurun -- 'python tools/make_engine_ckpt.py gpurun_out/engine_ckpt'
"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def main(out_dir):
    from coral_amd.modeling import Wav2Vec2ForCTC
    from coral_amd.trainer import DataParallelTrainer
    from coral_amd.whisper import WhisperShape
    from coral_amd.whisper_setup import WhisperForConditionalGeneration
    from coral_amd.wav2vec2 import Wav2Vec2Shape

    out = Path(out_dir)
    out.mkdir(parents=True, exist_ok=True)
    # --- wav2vec2 -----------------------------------------------------------------------------------------------
    shape = Wav2Vec2Shape(hidden_size=64, num_hidden_layers=2, num_attention_heads=4, intermediate_size=128,
                          conv_dim=(512, 32, 32, 32, 32, 32, 32), num_conv_pos_embeddings=16,
                          num_conv_pos_embedding_groups=4)
    model = Wav2Vec2ForCTC(shape, "cuda:0")
    model.init_weights(123)
    g = torch.Generator().manual_seed(5)
    waves = [(0.1 * torch.randn(n, generator=g)) for n in (4000, 3300)]
    iv = torch.zeros(2, 4000)
    am = torch.zeros(2, 4000, dtype=torch.long)
    for b, w in enumerate(waves):
        iv[b, :len(w)] = (w - w.mean()) / torch.sqrt(w.var(unbiased=False) + 1e-7)
        am[b, :len(w)] = 1
    labels = torch.tensor([[3, 7, 7, 1], [9, 2, -100, -100]])
    tr = DataParallelTrainer(model, learning_rate=1e-3, warmup_steps=0, max_steps=10)
    model.train()
    tr.train_step([dict(input_values=iv, attention_mask=am, labels=labels)])
    tr.finish()
    model.eval()
    res = model(iv, am, labels)
    model.save_pretrained(out / "engine_ckpt_w2v2")
    np.savez_compressed(out / "engine_ckpt_w2v2.npz", input_values=iv.numpy(), attention_mask=am.numpy(),
                        labels=labels.numpy(), logits=res.logits.float().cpu().numpy(), loss=float(res.loss))
    # --- Whisper ------------------------------------------------------------------------------------------------
    ws = WhisperShape(d_model=64, encoder_layers=2, decoder_layers=2, encoder_attention_heads=4,
                      decoder_attention_heads=4, encoder_ffn_dim=128, decoder_ffn_dim=128, num_mel_bins=80, vocab_size=200,
                      max_target_positions=64, pad_token_id=150, decoder_start_token_id=151, eos_token_id=150)
    wm = WhisperForConditionalGeneration(ws, "cuda:0")
    gd = torch.Generator(device="cuda:0").manual_seed(321)
    from coral_amd.whisper import sinusoid_positions

    for n in wm.engine.exported_names():
        v = wm.engine.store.view(n)
        if n.endswith("layer_norm.weight"):
            v.normal_(1.0, 0.05, generator=gd)
        elif n.endswith("encoder.embed_positions.weight"):
            v.copy_(sinusoid_positions(*v.shape).to(v.device))
        else:
            v.normal_(0.0, 0.05, generator=gd)
    wm.engine.refresh_compute_weights()
    wm.engine.refresh_derived()
    feats = torch.randn(2, 80, 3000, generator=g) * 0.5
    wl = torch.randint(0, 150, (2, 9), generator=g)
    wtr = DataParallelTrainer(wm, learning_rate=1e-3, warmup_steps=0, max_steps=10)
    wm.train()
    wtr.train_step([dict(input_features=feats, labels=wl)])
    wtr.finish()
    wm.eval()
    wres = wm(feats, labels=wl)
    wm.save_pretrained(out / "engine_ckpt_whisper")
    np.savez_compressed(out / "engine_ckpt_whisper.npz", labels=wl.numpy(), feats_seed=5,
                        feats_slice=feats[:, :, ::100].numpy(), logits=wres["logits"].float().cpu().numpy(),
                        loss=float(wres["loss"]))
    print("saved", sorted(p.name for p in out.iterdir()))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/engine_ckpt")
