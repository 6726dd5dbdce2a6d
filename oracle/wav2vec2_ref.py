"""ORACLE (test infrastructure only) — CPU restatement of the wav2vec2 CTC path CoRal drives.

This file is the *checker*, never the product: only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import it.  It restates, in plain torch fp32 ops (no
`transformers` import), the arithmetic of HuggingFace Transformers that
R/src/coral/wav2vec2.py:104-126 instantiates and R/src/coral/finetune.py:60-79 trains:

  zero_mean_unit_var_norm   $TF/models/wav2vec2/feature_extraction_wav2vec2.py:77-97
  feature encoder           $TF/models/wav2vec2/modeling_wav2vec2.py:275-299, 409-419
  feature projection        :429-434
  SpecAugment / padding     :1272-1316, 752-755
  positional conv embedding :326-379   (weight_norm dim=2, grouped conv, drop last, GELU)
  stable-LN encoder layers  :631-654, 741-802 ; attention :438-463, 500-548 ; FFN :565-572
  lengths / masks           :997-1036
  CTC head                  :1697-1728 (log_softmax fp32 -> F.ctc_loss blank=pad, zero_infinity)
  greedy decode             R/src/coral/compute_metrics.py:62-70,
                            $TF/models/wav2vec2/tokenization_wav2vec2.py:297-358

Pinned against goldens generated from transformers 5.15.0 by tools/gen_goldens.py
(tests/golden/*.npz, checked in tests/test_oracle_goldens.py).  ($TF line numbers: v5.15.0.)
"""

from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np
import torch
import torch.nn.functional as F


@dataclass
class W2V2Config:
    """Shape hyper-parameters (XLS-R family: stable layer norm, conv LayerNorm, conv bias)."""

    hidden_size: int = 1024
    num_hidden_layers: int = 24
    num_attention_heads: int = 16
    intermediate_size: int = 4096
    conv_dim: tuple = (512,) * 7
    conv_kernel: tuple = (10, 3, 3, 3, 3, 2, 2)
    conv_stride: tuple = (5, 2, 2, 2, 2, 2, 2)
    num_conv_pos_embeddings: int = 128
    num_conv_pos_embedding_groups: int = 16
    vocab_size: int = 46
    pad_token_id: int = 45
    layer_norm_eps: float = 1e-5
    ctc_loss_reduction: str = "sum"
    ctc_zero_infinity: bool = True

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads


# CoRal's model= keys (R/config/model/wav2vec2-{small,medium,large}.yaml) -> XLS-R shapes.
CORAL_SHAPES = {
    "wav2vec2-small": dict(hidden_size=1024, num_hidden_layers=24, intermediate_size=4096),
    "wav2vec2-medium": dict(hidden_size=1280, num_hidden_layers=48, intermediate_size=5120),
    "wav2vec2-large": dict(hidden_size=1920, num_hidden_layers=48, intermediate_size=7680),
}


def param_shapes(cfg: W2V2Config) -> dict[str, tuple]:
    """HF state_dict names -> shapes for Wav2Vec2ForCTC (the parameters the path touches)."""
    d, f = cfg.hidden_size, cfg.intermediate_size
    s: dict[str, tuple] = {}
    cin = 1
    for i, (co, k) in enumerate(zip(cfg.conv_dim, cfg.conv_kernel)):
        p = f"wav2vec2.feature_extractor.conv_layers.{i}."
        s[p + "conv.weight"] = (co, cin, k)
        s[p + "conv.bias"] = (co,)
        s[p + "layer_norm.weight"] = (co,)
        s[p + "layer_norm.bias"] = (co,)
        cin = co
    s["wav2vec2.feature_projection.layer_norm.weight"] = (cin,)
    s["wav2vec2.feature_projection.layer_norm.bias"] = (cin,)
    s["wav2vec2.feature_projection.projection.weight"] = (d, cin)
    s["wav2vec2.feature_projection.projection.bias"] = (d,)
    s["wav2vec2.masked_spec_embed"] = (d,)
    K, G = cfg.num_conv_pos_embeddings, cfg.num_conv_pos_embedding_groups
    s["wav2vec2.encoder.pos_conv_embed.conv.bias"] = (d,)
    s["wav2vec2.encoder.pos_conv_embed.conv.parametrizations.weight.original0"] = (1, 1, K)
    s["wav2vec2.encoder.pos_conv_embed.conv.parametrizations.weight.original1"] = (d, d // G, K)
    s["wav2vec2.encoder.layer_norm.weight"] = (d,)
    s["wav2vec2.encoder.layer_norm.bias"] = (d,)
    for l in range(cfg.num_hidden_layers):
        p = f"wav2vec2.encoder.layers.{l}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            s[p + f"attention.{n}.weight"] = (d, d)
            s[p + f"attention.{n}.bias"] = (d,)
        s[p + "layer_norm.weight"] = (d,)
        s[p + "layer_norm.bias"] = (d,)
        s[p + "feed_forward.intermediate_dense.weight"] = (f, d)
        s[p + "feed_forward.intermediate_dense.bias"] = (f,)
        s[p + "feed_forward.output_dense.weight"] = (d, f)
        s[p + "feed_forward.output_dense.bias"] = (d,)
        s[p + "final_layer_norm.weight"] = (d,)
        s[p + "final_layer_norm.bias"] = (d,)
    s["lm_head.weight"] = (cfg.vocab_size, d)
    s["lm_head.bias"] = (cfg.vocab_size,)
    return s


def _name_seed(name: str, seed: int) -> int:
    h = 1469598103934665603
    for ch in name.encode():
        h = ((h ^ ch) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return (h ^ (seed * 0x9E3779B97F4A7C15)) & 0x7FFFFFFFFFFFFFFF


def synth_params(cfg: W2V2Config, seed: int = 4242) -> dict[str, torch.Tensor]:
    """Deterministic, name-keyed random parameters (no checkpoint exists offline).

    Used by the golden generator (to overwrite the HF model's init) and by every test, so the
    weights never need to be stored.  Scales are chosen to keep activations O(1) through the
    stack (Kaiming-like for convs/linears; LayerNorm weights near 1).
    """
    out = {}
    for name, shape in param_shapes(cfg).items():
        g = torch.Generator().manual_seed(_name_seed(name, seed))
        if name.endswith("layer_norm.weight"):
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif name.endswith("original0"):
            t = 1.0 + 0.25 * torch.rand(shape, generator=g)
        elif name.endswith(".bias") or name.endswith("masked_spec_embed"):
            t = 0.05 * torch.randn(shape, generator=g)
        else:
            fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
            t = torch.randn(shape, generator=g) * (1.0 / math.sqrt(fan_in))
        out[name] = t.float()
    return out


# --------------------------------------------------------------------------------------------
def zero_mean_unit_var_norm(arrays: list[np.ndarray], pad_to: int | None = None):
    """Wav2Vec2FeatureExtractor(do_normalize=True, return_attention_mask=True) + pad.

    ($TF/models/wav2vec2/feature_extraction_wav2vec2.py:77-97,99-236; longest/max_length
    padding with 0.0, R/src/coral/data_collators.py:72-77.)  Returns (input_values f32 [B,N],
    attention_mask i32 [B,N]).
    """
    n = max(len(a) for a in arrays) if pad_to is None else pad_to
    vals = np.zeros((len(arrays), n), dtype=np.float32)
    mask = np.zeros((len(arrays), n), dtype=np.int32)
    for i, a in enumerate(arrays):
        a = np.asarray(a, dtype=np.float32)
        vals[i, : len(a)] = (a - a.mean()) / np.sqrt(a.var() + 1e-7)
        mask[i, : len(a)] = 1
    return vals, mask


def feat_extract_output_lengths(lengths, cfg: W2V2Config):
    """floor((n - k)/s) + 1 per conv layer ($TF/.../modeling_wav2vec2.py:997-1016)."""
    out = torch.as_tensor(lengths).clone().long()
    for k, s in zip(cfg.conv_kernel, cfg.conv_stride):
        out = torch.div(out - k, s, rounding_mode="floor") + 1
    return out


def feature_encoder(x: torch.Tensor, P: dict, cfg: W2V2Config, collect: dict | None = None):
    """7 x [Conv1d -> LayerNorm(C) -> GELU]; x [B,N] -> [B,T,C] (channels-last)."""
    h = x[:, None, :]
    for i, s in enumerate(cfg.conv_stride):
        p = f"wav2vec2.feature_extractor.conv_layers.{i}."
        h = F.conv1d(h, P[p + "conv.weight"], P[p + "conv.bias"], stride=s)
        h = h.transpose(1, 2)
        h = F.layer_norm(h, (h.shape[-1],), P[p + "layer_norm.weight"], P[p + "layer_norm.bias"],
                         cfg.layer_norm_eps)
        h = F.gelu(h)
        if collect is not None:
            collect[f"conv{i}"] = h
        h = h.transpose(1, 2)
    return h.transpose(1, 2)


def pos_conv_weight(P: dict) -> torch.Tensor:
    """weight_norm(dim=2): w = g * v / ||v|| with the norm over dims (0, 1)."""
    g = P["wav2vec2.encoder.pos_conv_embed.conv.parametrizations.weight.original0"]
    v = P["wav2vec2.encoder.pos_conv_embed.conv.parametrizations.weight.original1"]
    return g * v / v.norm(p=2, dim=(0, 1), keepdim=True)


def pos_conv_embed(h: torch.Tensor, P: dict, cfg: W2V2Config) -> torch.Tensor:
    K, G = cfg.num_conv_pos_embeddings, cfg.num_conv_pos_embedding_groups
    y = F.conv1d(h.transpose(1, 2), pos_conv_weight(P),
                 P["wav2vec2.encoder.pos_conv_embed.conv.bias"], padding=K // 2, groups=G)
    if K % 2 == 0:
        y = y[:, :, :-1]
    return F.gelu(y).transpose(1, 2)


def attention(x, P, pre, cfg: W2V2Config, key_mask):
    """Bidirectional MHA: softmax(q k^T / sqrt(hd) + mask) v ; key_mask bool [B,T] or None."""
    B, T, d = x.shape
    H, hd = cfg.num_attention_heads, cfg.head_dim
    q = F.linear(x, P[pre + "q_proj.weight"], P[pre + "q_proj.bias"]).view(B, T, H, hd).transpose(1, 2)
    k = F.linear(x, P[pre + "k_proj.weight"], P[pre + "k_proj.bias"]).view(B, T, H, hd).transpose(1, 2)
    v = F.linear(x, P[pre + "v_proj.weight"], P[pre + "v_proj.bias"]).view(B, T, H, hd).transpose(1, 2)
    s = torch.matmul(q, k.transpose(-1, -2)) * (hd ** -0.5)
    if key_mask is not None:
        s = s.masked_fill(~key_mask[:, None, None, :], torch.finfo(s.dtype).min)
    p = torch.softmax(s, dim=-1)
    o = torch.matmul(p, v).transpose(1, 2).reshape(B, T, d)
    return F.linear(o, P[pre + "out_proj.weight"], P[pre + "out_proj.bias"])


def encoder_layer(h, P, l: int, cfg: W2V2Config, key_mask):
    """Wav2Vec2EncoderLayerStableLayerNorm: h += Attn(LN(h)); h += FFN(LN(h))."""
    p = f"wav2vec2.encoder.layers.{l}."
    eps = cfg.layer_norm_eps
    x = F.layer_norm(h, (h.shape[-1],), P[p + "layer_norm.weight"], P[p + "layer_norm.bias"], eps)
    h = h + attention(x, P, p + "attention.", cfg, key_mask)
    x = F.layer_norm(h, (h.shape[-1],), P[p + "final_layer_norm.weight"],
                     P[p + "final_layer_norm.bias"], eps)
    x = F.gelu(F.linear(x, P[p + "feed_forward.intermediate_dense.weight"],
                        P[p + "feed_forward.intermediate_dense.bias"]))
    x = F.linear(x, P[p + "feed_forward.output_dense.weight"],
                 P[p + "feed_forward.output_dense.bias"])
    return h + x


def forward_logits(input_values, attention_mask, P, cfg: W2V2Config, mask_time=None,
                   mask_feature=None, collect: dict | None = None):
    """input_values f32 [B,N], attention_mask [B,N] (or None) -> logits f32 [B,T,V].

    mask_time bool [B,T] / mask_feature bool [B,d]: explicit SpecAugment masks (the reference
    draws them on the host with np.random, :101-217; parity runs inject or disable them).
    """
    eps = cfg.layer_norm_eps
    feats = feature_encoder(input_values, P, cfg, collect)
    B, T, _ = feats.shape
    frame_mask = None
    if attention_mask is not None:
        flen = feat_extract_output_lengths(attention_mask.sum(-1), cfg)
        frame_mask = torch.arange(T)[None, :] < flen[:, None]
    x = F.layer_norm(feats, (feats.shape[-1],), P["wav2vec2.feature_projection.layer_norm.weight"],
                     P["wav2vec2.feature_projection.layer_norm.bias"], eps)
    h = F.linear(x, P["wav2vec2.feature_projection.projection.weight"],
                 P["wav2vec2.feature_projection.projection.bias"])
    if collect is not None:
        collect["proj"] = h
    if mask_time is not None:
        h = torch.where(mask_time[:, :, None], P["wav2vec2.masked_spec_embed"].to(h.dtype), h)
    if mask_feature is not None:
        h = h.masked_fill(mask_feature[:, None, :], 0.0)
    if frame_mask is not None:
        h = h * frame_mask[:, :, None].to(h.dtype)
    h = h + pos_conv_embed(h, P, cfg)
    if collect is not None:
        collect["posconv"] = h
    for l in range(cfg.num_hidden_layers):
        h = encoder_layer(h, P, l, cfg, frame_mask)
        if collect is not None:
            collect[f"layer{l}"] = h
    h = F.layer_norm(h, (h.shape[-1],), P["wav2vec2.encoder.layer_norm.weight"],
                     P["wav2vec2.encoder.layer_norm.bias"], eps)
    if collect is not None:
        collect["final"] = h
    return F.linear(h, P["lm_head.weight"], P["lm_head.bias"])


# --------------------------------------------------------------------------------------------
def ctc_nll(log_probs: torch.Tensor, targets: list[int], t_in: int, blank: int) -> torch.Tensor:
    """-log p(targets | log_probs[:t_in]) by the alpha recursion in log space (differentiable).

    Restates aten/src/ATen/native/LossCTC.cpp (what F.ctc_loss runs on CPU).  log_probs [T,V].
    "-inf" is represented by -1e30 so autograd through logsumexp stays NaN-free; an
    infeasible alignment returns +inf.
    """
    NEG = -1e30
    ext = [blank]
    for c in targets:
        ext += [int(c), blank]
    S = len(ext)
    if t_in <= 0:
        return torch.tensor(float("inf"), dtype=log_probs.dtype)
    ext_t = torch.tensor(ext, dtype=torch.long)
    can_skip = torch.zeros(S, dtype=torch.bool)
    for s in range(2, S):
        can_skip[s] = ext[s] != blank and ext[s] != ext[s - 2]
    neg = torch.full((S,), NEG, dtype=log_probs.dtype)
    mask0 = torch.zeros(S, dtype=torch.bool)
    mask0[: min(2, S)] = True
    alpha = torch.where(mask0, log_probs[0, ext_t], neg)
    for t in range(1, t_in):
        a1 = torch.cat([neg[:1], alpha[:-1]])
        a2 = torch.where(can_skip, torch.cat([neg[:2], alpha[:-2]])[:S], neg)
        alpha = torch.logsumexp(torch.stack([alpha, a1, a2]), dim=0) + log_probs[t, ext_t]
        alpha = torch.clamp(alpha, min=NEG)
    tail = alpha[-1:] if S == 1 else alpha[-2:]
    ll = torch.logsumexp(tail, dim=0)
    if float(ll.detach()) < -1e29:
        return torch.tensor(float("inf"), dtype=log_probs.dtype)
    return -ll


def ctc_loss(logits: torch.Tensor, labels: torch.Tensor, input_lengths, cfg: W2V2Config):
    """Wav2Vec2ForCTC loss tail ($TF/.../modeling_wav2vec2.py:1705-1728).

    logits [B,T,V] (any float dtype), labels i64 [B,L] with -100 padding.  Returns
    (loss scalar, per-utterance nll [B] after zero_infinity).
    """
    lp = torch.log_softmax(logits.float(), dim=-1)
    nlls = []
    for b in range(logits.shape[0]):
        tg = [int(c) for c in labels[b].tolist() if c >= 0]
        nll = ctc_nll(lp[b], tg, int(input_lengths[b]), cfg.pad_token_id)
        if cfg.ctc_zero_infinity and torch.isinf(nll):
            nll = torch.zeros((), dtype=lp.dtype)
        nlls.append(nll)
    nlls_t = torch.stack(nlls)
    if cfg.ctc_loss_reduction == "sum":
        loss = nlls_t.sum()
    else:  # "mean": divide by target lengths (clamped to 1), then batch mean
        tl = torch.tensor([max(1, int((labels[b] >= 0).sum())) for b in range(len(nlls))])
        loss = (nlls_t / tl).mean()
    return loss, nlls_t


def forward_loss(input_values, attention_mask, labels, P, cfg: W2V2Config, **kw):
    """model(input_values, attention_mask, labels) -> (loss, logits) like Wav2Vec2ForCTC.forward."""
    logits = forward_logits(input_values, attention_mask, P, cfg, **kw)
    am = attention_mask if attention_mask is not None else torch.ones_like(input_values, dtype=torch.long)
    in_len = feat_extract_output_lengths(am.sum(-1), cfg)
    loss, nll = ctc_loss(logits, labels, in_len, cfg)
    return loss, logits, nll


# --------------------------------------------------------------------------------------------
def greedy_ctc_ids(logits: np.ndarray, blank: int) -> list[list[int]]:
    """argmax -> collapse repeats -> drop blank (compute_metrics.py:68-69 + tokenizer
    convert_tokens_to_string grouping, tokenization_wav2vec2.py:311-323)."""
    out = []
    for row in np.asarray(logits).argmax(-1):
        ids, prev = [], None
        for c in row.tolist():
            if c != prev:
                if c != blank:
                    ids.append(int(c))
                prev = c
        out.append(ids)
    return out


def coral_vocab(characters_to_keep: str = "abcdefghijklmnopqrstuvwxyzæøå0123456789éü") -> dict:
    """R/src/coral/wav2vec2.py:308-329 (dump_vocabulary): sorted unique characters + '|',
    then <s>, </s>, <unk>, <pad> appended by the tokenizer (R/src/coral/wav2vec2.py:64-72)."""
    chars = sorted(set(characters_to_keep + "|"))
    vocab = {c: i for i, c in enumerate(chars)}
    for tok in ("<s>", "</s>", "<unk>", "<pad>"):
        vocab[tok] = len(vocab)
    return vocab


def ids_to_text(ids: list[int], vocab: dict) -> str:
    """Token ids (already collapsed, blank-free) -> string, as convert_tokens_to_string does
    ($TF/models/wav2vec2/tokenization_wav2vec2.py:316-352): '|' -> ' ', join, strip (no
    whitespace collapsing; non-pad special tokens are emitted verbatim because
    `batch_decode` is called without skip_special_tokens, compute_metrics.py:69)."""
    inv = {i: c for c, i in vocab.items()}
    return "".join(" " if inv[i] == "|" else inv[i] for i in ids if inv[i] != "<pad>").strip()
