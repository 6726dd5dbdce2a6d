"""ORACLE (test infrastructure only) — CPU restatement of the Whisper path CoRal drives.

Checker, never product: only tests/, smoke() and bench.py's cpu_baseline leg may import it.
Restates in NumPy / plain torch fp32 (no `transformers` import):

  log-mel front end      $TF/models/whisper/feature_extraction_whisper.py:95-103,135-168
                         mel bank: $TF/audio_utils.py:638-731 (slaney scale, slaney norm)
  encoder                $TF/models/whisper/modeling_whisper.py:55-64 (sinusoids), 592-646
  attention              :284-356  (q scaled by hd^-0.5 before the product, k_proj has no bias)
  encoder/decoder layers :379-413, 448-505 (pre-LN; self-attn, cross-attn, FFN with GELU)
  decoder                :690-795 (token + learned position embeddings, causal mask)
  LM head + loss         :68-81 (shift_tokens_right), 1063-1087 (tied proj_out, CE ignore -100)
  greedy generation      $TF/models/whisper/generation_whisper.py:383,1455,1774-1812 — forced prefix,
                         suppress / begin-suppress token processors, argmax until EOS or max_length
                         (CoRal clears forced_decoder_ids and suppress_tokens, R/src/coral/whisper.py:103-107)

Pinned by tests/golden/logmel.npz and tests/golden/whisper_tiny.npz (tools/gen_goldens.py, HF 5.15.0).
"""

from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np
import torch
import torch.nn.functional as F

N_FFT, HOP, N_SAMPLES = 400, 160, 480_000


# ---- log-mel ----------------------------------------------------------------------------------
def _hz_to_mel_slaney(f):
    f = np.asarray(f, dtype=np.float64)
    lin = 3.0 * f / 200.0
    log = 15.0 + np.log(np.maximum(f, 1e-10) / 1000.0) * (27.0 / np.log(6.4))
    return np.where(f >= 1000.0, log, lin)


def _mel_to_hz_slaney(m):
    m = np.asarray(m, dtype=np.float64)
    lin = 200.0 * m / 3.0
    log = 1000.0 * np.exp((np.log(6.4) / 27.0) * (m - 15.0))
    return np.where(m >= 15.0, log, lin)


def mel_filter_bank(n_mels: int, n_freq: int = 201, sr: int = 16_000, fmin=0.0, fmax=8000.0) -> np.ndarray:
    """[n_freq, n_mels] triangular filters, Slaney scale + Slaney (area) normalisation."""
    mel_pts = np.linspace(_hz_to_mel_slaney(fmin), _hz_to_mel_slaney(fmax), n_mels + 2)
    hz = _mel_to_hz_slaney(mel_pts)
    fft_freqs = np.linspace(0, sr // 2, n_freq)
    diff = np.diff(hz)
    slopes = hz[None, :] - fft_freqs[:, None]
    down = -slopes[:, :-2] / diff[:-1]
    up = slopes[:, 2:] / diff[1:]
    fb = np.maximum(0.0, np.minimum(down, up))
    fb *= (2.0 / (hz[2:n_mels + 2] - hz[:n_mels]))[None, :]
    return fb.astype(np.float32)


def log_mel(wave: np.ndarray, n_mels: int = 80) -> np.ndarray:
    """wave f32 [N] (already padded/truncated, N % 160 == 0) -> f32 [n_mels, N/160]."""
    x = np.asarray(wave, dtype=np.float64)
    pad = N_FFT // 2
    xp = np.pad(x, (pad, pad), mode="reflect")
    n_frames = 1 + (len(xp) - N_FFT) // HOP
    idx = np.arange(N_FFT)[None, :] + HOP * np.arange(n_frames)[:, None]
    win = 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(N_FFT) / N_FFT)  # periodic Hann
    spec = np.fft.rfft(xp[idx] * win[None, :], n=N_FFT, axis=1)
    power = (spec.real ** 2 + spec.imag ** 2)[:-1]  # drop the last frame
    mel = power @ mel_filter_bank(n_mels).astype(np.float64)
    logm = np.log10(np.maximum(mel, 1e-10)).T
    logm = np.maximum(logm, logm.max() - 8.0)
    return ((logm + 4.0) / 4.0).astype(np.float32)


def pad_or_trim(wave: np.ndarray, n: int = N_SAMPLES) -> np.ndarray:
    w = np.asarray(wave, dtype=np.float32)[:n]
    return np.pad(w, (0, n - len(w)))


# ---- model --------------------------------------------------------------------------------------
@dataclass
class WhisperConfig:
    d_model: int = 384
    encoder_layers: int = 4
    decoder_layers: int = 4
    encoder_attention_heads: int = 6
    decoder_attention_heads: int = 6
    encoder_ffn_dim: int = 1536
    decoder_ffn_dim: int = 1536
    num_mel_bins: int = 80
    vocab_size: int = 51865
    max_source_positions: int = 1500
    max_target_positions: int = 448
    pad_token_id: int = 50257
    decoder_start_token_id: int = 50258
    eos_token_id: int = 50257
    layer_norm_eps: float = 1e-5


# CoRal model keys (R/config/model/whisper-*.yaml) -> architectures (public HF config.json values)
CORAL_SHAPES = {
    "whisper-xxsmall": dict(d_model=384, encoder_layers=4, decoder_layers=4, encoder_attention_heads=6,
                            decoder_attention_heads=6, encoder_ffn_dim=1536, decoder_ffn_dim=1536),
    "whisper-xsmall": dict(d_model=512, encoder_layers=6, decoder_layers=6, encoder_attention_heads=8,
                           decoder_attention_heads=8, encoder_ffn_dim=2048, decoder_ffn_dim=2048),
    "whisper-small": dict(d_model=768, encoder_layers=12, decoder_layers=12, encoder_attention_heads=12,
                          decoder_attention_heads=12, encoder_ffn_dim=3072, decoder_ffn_dim=3072),
    "whisper-medium": dict(d_model=1024, encoder_layers=24, decoder_layers=24, encoder_attention_heads=16,
                           decoder_attention_heads=16, encoder_ffn_dim=4096, decoder_ffn_dim=4096),
    "whisper-large": dict(d_model=1280, encoder_layers=32, decoder_layers=32, encoder_attention_heads=20,
                          decoder_attention_heads=20, encoder_ffn_dim=5120, decoder_ffn_dim=5120,
                          num_mel_bins=128, vocab_size=51866),
    "whisper-large-turbo": dict(d_model=1280, encoder_layers=32, decoder_layers=4, encoder_attention_heads=20,
                                decoder_attention_heads=20, encoder_ffn_dim=5120, decoder_ffn_dim=5120,
                                num_mel_bins=128, vocab_size=51866),
}


def param_shapes(c: WhisperConfig) -> dict[str, tuple]:
    d = c.d_model
    s = {"model.encoder.conv1.weight": (d, c.num_mel_bins, 3), "model.encoder.conv1.bias": (d,),
         "model.encoder.conv2.weight": (d, d, 3), "model.encoder.conv2.bias": (d,),
         "model.encoder.embed_positions.weight": (c.max_source_positions, d)}

    def attn(p):
        for n in ("q_proj", "v_proj", "out_proj"):
            s[p + n + ".weight"] = (d, d)
            s[p + n + ".bias"] = (d,)
        s[p + "k_proj.weight"] = (d, d)

    for l in range(c.encoder_layers):
        p = f"model.encoder.layers.{l}."
        attn(p + "self_attn.")
        for n, shp in (("self_attn_layer_norm", (d,)), ("final_layer_norm", (d,))):
            s[p + n + ".weight"], s[p + n + ".bias"] = shp, shp
        s[p + "fc1.weight"], s[p + "fc1.bias"] = (c.encoder_ffn_dim, d), (c.encoder_ffn_dim,)
        s[p + "fc2.weight"], s[p + "fc2.bias"] = (d, c.encoder_ffn_dim), (d,)
    s["model.encoder.layer_norm.weight"], s["model.encoder.layer_norm.bias"] = (d,), (d,)
    s["model.decoder.embed_tokens.weight"] = (c.vocab_size, d)
    s["model.decoder.embed_positions.weight"] = (c.max_target_positions, d)
    for l in range(c.decoder_layers):
        p = f"model.decoder.layers.{l}."
        attn(p + "self_attn.")
        attn(p + "encoder_attn.")
        for n in ("self_attn_layer_norm", "encoder_attn_layer_norm", "final_layer_norm"):
            s[p + n + ".weight"], s[p + n + ".bias"] = (d,), (d,)
        s[p + "fc1.weight"], s[p + "fc1.bias"] = (c.decoder_ffn_dim, d), (c.decoder_ffn_dim,)
        s[p + "fc2.weight"], s[p + "fc2.bias"] = (d, c.decoder_ffn_dim), (d,)
    s["model.decoder.layer_norm.weight"], s["model.decoder.layer_norm.bias"] = (d,), (d,)
    return s


def sinusoids(length: int, channels: int, max_timescale: float = 10000.0) -> torch.Tensor:
    inc = math.log(max_timescale) / (channels // 2 - 1)
    inv = torch.exp(-inc * torch.arange(channels // 2))
    t = torch.arange(length).view(-1, 1) * inv.view(1, -1)
    return torch.cat([t.sin(), t.cos()], dim=1)


def _name_seed(name: str, seed: int) -> int:
    h = 1469598103934665603
    for ch in name.encode():
        h = ((h ^ ch) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return (h ^ (seed * 0x9E3779B97F4A7C15)) & 0x7FFFFFFFFFFFFFFF


def synth_params(c: WhisperConfig, seed: int = 4242) -> dict[str, torch.Tensor]:
    """Name-keyed seeded parameters (same role as wav2vec2_ref.synth_params)."""
    out = {}
    for name, shape in param_shapes(c).items():
        g = torch.Generator().manual_seed(_name_seed(name, seed))
        if name == "model.encoder.embed_positions.weight":
            t = sinusoids(*shape)
        elif name.endswith("layer_norm.weight"):
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif name.endswith(".bias"):
            t = 0.05 * torch.randn(shape, generator=g)
        elif "embed_" in name:
            t = 0.02 * torch.randn(shape, generator=g)  # small: keeps greedy decoding context-dependent
        else:
            fan_in = int(np.prod(shape[1:]))
            t = torch.randn(shape, generator=g) / math.sqrt(fan_in)
        out[name] = t.float()
    return out


def _attn(xq, xkv, P, p, H, causal=False, pmask=None):
    """pmask: injected dropout mask on the attention probabilities (keep / (1 - p), [B, H, Tq, Tk]) or None -
    `nn.functional.dropout(attn_weights, p=dropout)` at $TF/models/whisper/modeling_whisper.py:234."""
    B, Tq, d = xq.shape
    Tk = xkv.shape[1]
    hd = d // H
    q = (F.linear(xq, P[p + "q_proj.weight"], P[p + "q_proj.bias"]) * hd ** -0.5).view(B, Tq, H, hd).transpose(1, 2)
    k = F.linear(xkv, P[p + "k_proj.weight"]).view(B, Tk, H, hd).transpose(1, 2)
    v = F.linear(xkv, P[p + "v_proj.weight"], P[p + "v_proj.bias"]).view(B, Tk, H, hd).transpose(1, 2)
    s = q @ k.transpose(-1, -2)
    if causal:
        m = torch.ones(Tq, Tk, dtype=torch.bool).tril(Tk - Tq)
        s = s.masked_fill(~m, float("-inf"))
    pr = torch.softmax(s, -1)
    if pmask is not None:
        pr = pr * pmask
    o = (pr @ v).transpose(1, 2).reshape(B, Tq, d)
    return F.linear(o, P[p + "out_proj.weight"], P[p + "out_proj.bias"])


def _ln(x, P, p, eps):
    return F.layer_norm(x, (x.shape[-1],), P[p + ".weight"], P[p + ".bias"], eps)


def _drop(x, masks, key):
    """Training-time hidden-state dropout with an INJECTED multiplicative mask (keep / (1 - p) per element, same shape as
    x): `nn.functional.dropout(hidden_states, p=self.dropout)` at $TF/models/whisper/modeling_whisper.py:398,406 (encoder
    layer), :479,493,502 (decoder layer), :625,763 (embedded inputs).  masks None / key absent = evaluation."""
    if masks is None or key not in masks:
        return x
    return x * masks[key].reshape(x.shape)


def encoder(input_features: torch.Tensor, P: dict, c: WhisperConfig, keep=None, masks=None) -> torch.Tensor:
    """input_features f32 [B, mels, 3000] -> [B, 1500, d].  keep[l] False = the layer is skipped
    (training-time LayerDrop, $TF/models/whisper/modeling_whisper.py:626-634); masks: see `_drop`
    (keys "enc_embed", "enc{l}.attn", "enc{l}.ffn")."""
    x = F.gelu(F.conv1d(input_features, P["model.encoder.conv1.weight"], P["model.encoder.conv1.bias"], padding=1))
    x = F.gelu(F.conv1d(x, P["model.encoder.conv2.weight"], P["model.encoder.conv2.bias"], stride=2, padding=1))
    h = _drop(x.permute(0, 2, 1) + P["model.encoder.embed_positions.weight"], masks, "enc_embed")
    for l in range(c.encoder_layers):
        if keep is not None and not keep[l]:
            continue
        p = f"model.encoder.layers.{l}."
        x = _ln(h, P, p + "self_attn_layer_norm", c.layer_norm_eps)
        h = h + _drop(_attn(x, x, P, p + "self_attn.", c.encoder_attention_heads, pmask=(masks or {}).get(f"enc{l}.probs")),
                      masks, f"enc{l}.attn")
        y = _ln(h, P, p + "final_layer_norm", c.layer_norm_eps)
        y = F.linear(F.gelu(F.linear(y, P[p + "fc1.weight"], P[p + "fc1.bias"])), P[p + "fc2.weight"], P[p + "fc2.bias"])
        h = h + _drop(y, masks, f"enc{l}.ffn")
    return _ln(h, P, "model.encoder.layer_norm", c.layer_norm_eps)


def decoder(input_ids: torch.Tensor, enc: torch.Tensor, P: dict, c: WhisperConfig, keep=None, masks=None) -> torch.Tensor:
    """input_ids i64 [B, L], enc [B, 1500, d] -> logits [B, L, V] (tied projection); keep as in
    `encoder` (:771-779); masks keys "dec_embed", "dec{l}.self", "dec{l}.cross", "dec{l}.ffn"."""
    L = input_ids.shape[1]
    h = _drop(P["model.decoder.embed_tokens.weight"][input_ids] + P["model.decoder.embed_positions.weight"][:L], masks,
              "dec_embed")
    for l in range(c.decoder_layers):
        if keep is not None and not keep[l]:
            continue
        p = f"model.decoder.layers.{l}."
        x = _ln(h, P, p + "self_attn_layer_norm", c.layer_norm_eps)
        h = h + _drop(_attn(x, x, P, p + "self_attn.", c.decoder_attention_heads, causal=True,
                            pmask=(masks or {}).get(f"dec{l}.self_probs")), masks, f"dec{l}.self")
        x = _ln(h, P, p + "encoder_attn_layer_norm", c.layer_norm_eps)
        h = h + _drop(_attn(x, enc, P, p + "encoder_attn.", c.decoder_attention_heads,
                            pmask=(masks or {}).get(f"dec{l}.cross_probs")), masks, f"dec{l}.cross")
        y = _ln(h, P, p + "final_layer_norm", c.layer_norm_eps)
        y = F.linear(F.gelu(F.linear(y, P[p + "fc1.weight"], P[p + "fc1.bias"])), P[p + "fc2.weight"], P[p + "fc2.bias"])
        h = h + _drop(y, masks, f"dec{l}.ffn")
    h = _ln(h, P, "model.decoder.layer_norm", c.layer_norm_eps)
    return F.linear(h, P["model.decoder.embed_tokens.weight"])


def shift_tokens_right(labels: torch.Tensor, pad_id: int, start_id: int) -> torch.Tensor:
    out = labels.new_zeros(labels.shape)
    out[:, 1:] = labels[:, :-1]
    out[:, 0] = start_id
    return out.masked_fill(out == -100, pad_id)


def forward_loss(input_features, labels, P, c: WhisperConfig, enc_keep=None, dec_keep=None, masks=None):
    """WhisperForConditionalGeneration.forward(input_features, labels) -> (loss, logits)."""
    dec_in = shift_tokens_right(labels, c.pad_token_id, c.decoder_start_token_id)
    logits = decoder(dec_in, encoder(input_features, P, c, enc_keep, masks), P, c, dec_keep, masks)
    loss = F.cross_entropy(logits.reshape(-1, c.vocab_size), labels.reshape(-1), ignore_index=-100)
    return loss, logits


def greedy_generate(input_features, P, c: WhisperConfig, prefix: list[int], max_length: int,
                    suppress: list[int] | None = None, begin_suppress: list[int] | None = None) -> list[list[int]]:
    """Greedy decoding with a forced prefix: at every step the next token is the argmax of the last
    position's logits after -inf-ing `suppress` (and `begin_suppress` at the first generated
    position); a finished row keeps emitting pad (= eos); stops at max_length or when all rows ended."""
    enc = encoder(input_features, P, c)
    B = input_features.shape[0]
    ids = torch.tensor([prefix] * B, dtype=torch.long)
    done = torch.zeros(B, dtype=torch.bool)
    while ids.shape[1] < max_length and not bool(done.all()):
        lg = decoder(ids, enc, P, c)[:, -1].clone()
        if suppress:
            lg[:, suppress] = float("-inf")
        if begin_suppress and ids.shape[1] == len(prefix):
            lg[:, begin_suppress] = float("-inf")
        nxt = lg.argmax(-1)
        nxt = torch.where(done, torch.full_like(nxt, c.pad_token_id), nxt)
        ids = torch.cat([ids, nxt[:, None]], 1)
        done |= nxt == c.eos_token_id
    return ids.tolist()
