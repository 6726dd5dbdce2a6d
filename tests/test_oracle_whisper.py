"""Pin oracle/whisper_ref.py to HF Transformers fixtures (tools/gen_goldens.py): log-mel front end
(both mel sizes, reflect-padded head frames, the per-clip max-8 floor) and a small
WhisperForConditionalGeneration (encoder states, teacher-forced logits, CE loss, gradients, greedy ids)."""
import numpy as np
import torch

from oracle import whisper_ref as w


def _clips():
    rng = np.random.RandomState(5)
    t = np.arange(59_200) / 16000.0
    return [(0.3 * np.sin(2 * np.pi * 440 * t) + 0.05 * rng.randn(len(t))).astype(np.float32),
            (0.1 * rng.randn(480_000)).astype(np.float32)]


def test_mel_filters_and_logmel_match_hf(golden_dir):
    z = np.load(golden_dir / "logmel.npz")
    clips = _clips()
    assert len(clips[0]) == int(z["clip0_len"])
    for mels in (80, 128):
        np.testing.assert_allclose(w.mel_filter_bank(mels), z[f"filters{mels}"], atol=1e-7)
        feats = np.stack([w.log_mel(w.pad_or_trim(c), mels) for c in clips])
        assert feats.shape == (2, mels, 3000)
        np.testing.assert_allclose(feats[:, :, ::25], z[f"feat{mels}_sub"], atol=1e-4)   # 1e-4
        np.testing.assert_allclose(feats[:, :, :40], z[f"feat{mels}_head"], atol=1e-4)
        stats = np.array([[f.mean(), f.std(), f.min(), f.max()] for f in feats])
        np.testing.assert_allclose(stats, z[f"feat{mels}_stats"], atol=1e-4)


def tiny_cfg():
    return w.WhisperConfig(d_model=64, encoder_layers=2, decoder_layers=2, encoder_attention_heads=4,
                           decoder_attention_heads=4, encoder_ffn_dim=128, decoder_ffn_dim=128, num_mel_bins=80,
                           vocab_size=200, max_target_positions=64, pad_token_id=150, decoder_start_token_id=151,
                           eos_token_id=150)


def test_whisper_tiny_matches_hf(golden_dir):
    z = np.load(golden_dir / "whisper_tiny.npz")
    c = tiny_cfg()
    P = {k: v.requires_grad_(True) for k, v in w.synth_params(c).items()}
    g = torch.Generator().manual_seed(9)
    feats = torch.randn(2, 80, 3000, generator=g) * 0.5
    labels = torch.from_numpy(z["labels"])
    enc = w.encoder(feats, P, c)
    np.testing.assert_allclose(enc.detach()[:, ::100].numpy(), z["enc_slice"], atol=2e-5)
    loss, logits = w.forward_loss(feats, labels, P, c)
    np.testing.assert_allclose(logits.detach().numpy(), z["logits"], atol=2e-5)
    assert abs(float(loss) - float(z["loss"])) < 1e-5
    loss.backward()
    np.testing.assert_allclose(P["model.decoder.layers.1.fc1.weight"].grad.numpy(), z["grad_fc1"], atol=1e-6)
    assert abs(float(P["model.decoder.embed_tokens.weight"].grad.norm()) - float(z["gradnorm_embed"])) < 1e-4 * float(z["gradnorm_embed"])
    assert abs(float(P["model.encoder.conv1.weight"].grad.norm()) - float(z["gradnorm_conv1"])) < 1e-4 * float(z["gradnorm_conv1"])
    with torch.no_grad():
        ids = w.greedy_generate(feats, {k: v.detach() for k, v in P.items()}, c, [151, 160, 161, 162], 24,
                                suppress=[170, 171], begin_suppress=[20, 150])
    assert ids == z["greedy_ids"].tolist()
    assert len(set(ids[0][4:])) >= 4  # the fixture is not a degenerate repeat
