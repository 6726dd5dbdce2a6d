"""Which bf16 roundings produce the CTC-loss error of a 48-layer XLS-R forward?  CPU emulation: the oracle's fp32
forward with bf16 round-trips injected at chosen groups of sites, on the five utterances of
tests/test_fulldepth_gpu.py (one full 10 s utterance + four ragged ones), loss error against the plain fp32 oracle.

  w    every weight matrix (conv 1-6, projection, weight-normed positional conv, q|k|v, out, fc1, fc2, lm_head)
  in   every GEMM input activation (LayerNorm outputs, attention context, GELU output, conv block outputs)
  res  the residual stream after every add (what an engine with a bf16 residual stream stores)
  qkv  the stored q | k | v and the attention probabilities
  y    the conv stack's pre-norm outputs

python tools/dev_bf16_emulation.py [model-key]      (build container or GPU box host cores; ~5 minutes for wav2vec2-large)"""
import sys
import time
from pathlib import Path

import torch
import torch.nn.functional as F

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from oracle import wav2vec2_ref as ref  # noqa: E402

key = sys.argv[1] if len(sys.argv) > 1 else "wav2vec2-large"
cfg = ref.W2V2Config(**ref.CORAL_SHAPES[key])
P = ref.synth_params(cfg)


def bf(x):
    return x.to(torch.bfloat16).to(torch.float32)


def forward(iv, am, sites: set, Pw: dict):
    """forward_logits of the oracle with rounding hooks; Pw = the parameter dict to use for matrices."""
    r = {k: (bf if k in sites else (lambda x: x)) for k in ("in", "res", "qkv", "y")}
    eps = cfg.layer_norm_eps
    h = iv[:, None, :]
    for i, s in enumerate(cfg.conv_stride):
        p = f"wav2vec2.feature_extractor.conv_layers.{i}."
        h = F.conv1d(h, Pw[p + "conv.weight"] if i else P[p + "conv.weight"], P[p + "conv.bias"], stride=s)
        if i:
            h = r["y"](h)
        h = h.transpose(1, 2)
        h = F.gelu(F.layer_norm(h, (h.shape[-1],), P[p + "layer_norm.weight"], P[p + "layer_norm.bias"], eps))
        h = (r["in"](h) if i < 6 else h).transpose(1, 2)
    feats = h.transpose(1, 2)
    B, T, _ = feats.shape
    flen = ref.feat_extract_output_lengths(am.sum(-1), cfg)
    fmask = torch.arange(T)[None, :] < flen[:, None]
    x = r["in"](F.layer_norm(feats, (feats.shape[-1],), P["wav2vec2.feature_projection.layer_norm.weight"],
                             P["wav2vec2.feature_projection.layer_norm.bias"], eps))
    h = r["res"](F.linear(x, Pw["wav2vec2.feature_projection.projection.weight"], P["wav2vec2.feature_projection.projection.bias"]))
    h = h * fmask[:, :, None].to(h.dtype)
    K, G = cfg.num_conv_pos_embeddings, cfg.num_conv_pos_embedding_groups
    y = F.conv1d(h.transpose(1, 2), Pw["__posconv"], P["wav2vec2.encoder.pos_conv_embed.conv.bias"], padding=K // 2, groups=G)
    h = r["res"](h + F.gelu(y[:, :, :-1]).transpose(1, 2))
    H, hd, d = cfg.num_attention_heads, cfg.head_dim, cfg.hidden_size
    for l in range(cfg.num_hidden_layers):
        p = f"wav2vec2.encoder.layers.{l}."
        x = r["in"](F.layer_norm(h, (d,), P[p + "layer_norm.weight"], P[p + "layer_norm.bias"], eps))
        q, k, v = (r["qkv"](F.linear(x, Pw[p + f"attention.{n}_proj.weight"], P[p + f"attention.{n}_proj.bias"]))
                   .view(B, T, H, hd).transpose(1, 2) for n in ("q", "k", "v"))
        s = torch.matmul(q, k.transpose(-1, -2)) * (hd ** -0.5)
        s = s.masked_fill(~fmask[:, None, None, :], torch.finfo(s.dtype).min)
        pr = torch.softmax(s, dim=-1)
        # (the fused kernel normalises with the fp32 row sum and multiplies bf16 probabilities into V)
        o = r["in"](torch.matmul(r["qkv"](pr), v).transpose(1, 2).reshape(B, T, d))
        h = r["res"](h + F.linear(o, Pw[p + "attention.out_proj.weight"], P[p + "attention.out_proj.bias"]))
        x = r["in"](F.layer_norm(h, (d,), P[p + "final_layer_norm.weight"], P[p + "final_layer_norm.bias"], eps))
        x = r["in"](F.gelu(F.linear(x, Pw[p + "feed_forward.intermediate_dense.weight"], P[p + "feed_forward.intermediate_dense.bias"])))
        h = r["res"](h + F.linear(x, Pw[p + "feed_forward.output_dense.weight"], P[p + "feed_forward.output_dense.bias"]))
    h = r["in"](F.layer_norm(h, (d,), P["wav2vec2.encoder.layer_norm.weight"], P["wav2vec2.encoder.layer_norm.bias"], eps))
    return F.linear(h, Pw["lm_head.weight"], P["lm_head.bias"]), flen


def losses(logits, flen, labels):
    lp = torch.log_softmax(logits.double(), -1).transpose(0, 1)
    tl = (labels >= 0).sum(-1)
    tg = torch.cat([labels[b, :tl[b]] for b in range(labels.shape[0])])
    return F.ctc_loss(lp, tg, flen, tl, blank=cfg.pad_token_id, reduction="none", zero_infinity=True)


# the utterances of tests/test_fulldepth_gpu.py
g = torch.Generator().manual_seed(4242)
x = (0.1 * torch.randn(160_000, generator=g)).clamp(-1, 1)
iv1, am1 = ref.zero_mean_unit_var_norm([(x / x.abs().max()).numpy()])
lab1 = torch.randint(0, 42, (1, 96), generator=g)
waves = []
for n in [160_000, 131_200, 99_840, 147_520]:
    w_ = (0.1 * torch.randn(n, generator=g)).clamp(-1, 1)
    waves.append((w_ / w_.abs().max()).numpy())
iv4, am4 = ref.zero_mean_unit_var_norm(waves)
lab4 = torch.full((4, 90), -100, dtype=torch.int64)
for b, L in enumerate((90, 70, 48, 81)):
    lab4[b, :L] = torch.randint(0, 42, (L,), generator=g)
batches = [(torch.from_numpy(iv1), torch.from_numpy(am1).long(), lab1), (torch.from_numpy(iv4), torch.from_numpy(am4).long(), lab4)]

P32 = dict(P)
P32["__posconv"] = ref.pos_conv_weight(P)
P16 = {k: (bf(v) if v.dim() >= 2 else v) for k, v in P32.items()}
variants = [("fp32", set(), P32), ("w", set(), P16), ("w+in", {"in"}, P16), ("w+in+qkv", {"in", "qkv"}, P16),
            ("w+in+qkv+res", {"in", "qkv", "res"}, P16), ("all (+y)", {"in", "qkv", "res", "y"}, P16),
            ("in+qkv+res (fp32 weights)", {"in", "qkv", "res"}, P32), ("res only", {"res"}, P32)]
base = None
with torch.no_grad():
    for name, sites, Pw in variants:
        t0 = time.time()
        ls = torch.cat([losses(*forward(iv, am, sites, Pw), lab) for iv, am, lab in batches])
        if base is None:
            base = ls
            print(f"{key}: fp32 losses " + ", ".join(f"{float(v):.3f}" for v in ls) + f"   ({time.time() - t0:.0f} s per variant)")
            continue
        rel = (ls - base) / base
        print(f"{name:28s} signed rel err per utterance: " + "  ".join(f"{float(v):+.2e}" for v in rel) +
              f"   | five together {float((ls.sum() - base.sum()) / base.sum()):+.2e}", flush=True)
