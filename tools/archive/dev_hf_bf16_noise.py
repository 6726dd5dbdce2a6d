"""Per-stage noise of the REFERENCE's own bf16 path: HF Transformers under torch.autocast(bfloat16) (what `bf16=True`
runs, R/src/coral/wav2vec2.py:183-193) against its fp32 path at the XLS-R-2B shape, on the utterance of
tests/test_fulldepth_gpu.py.  For every stage the relative RMS error ||a - b|| / ||b|| (conv0..6, feature projection,
positional conv, every sixth layer, final LayerNorm, logits), the loss error and its first-order decomposition
(sum g * delta over the logits; the part carried by the per-class time-mean of delta).  Writes
tests/golden/w2v2_cfg2_bf16_noise.npz, which tools/dev_depth_drift.py prints beside the engine's own figures.
Runs in the build container only (imports transformers):  python tools/dev_hf_bf16_noise.py [model-key]"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))
from gen_goldens import hf_w2v2  # noqa: E402
from oracle import wav2vec2_ref as ref  # noqa: E402

key = sys.argv[1] if len(sys.argv) > 1 else "wav2vec2-large"
cfg = ref.W2V2Config(**ref.CORAL_SHAPES[key])
g = torch.Generator().manual_seed(4242)
x = (0.1 * torch.randn(160_000, generator=g)).clamp(-1, 1)
iv, am = ref.zero_mean_unit_var_norm([(x / x.abs().max()).numpy()])
iv, am = torch.from_numpy(iv), torch.from_numpy(am).long()
labels = torch.randint(0, 42, (1, 96), generator=g)
model = hf_w2v2(cfg).eval()
L = cfg.num_hidden_layers
stages = {}


def hook(name, pick=lambda o: o):
    def f(_m, _i, o):
        stages.setdefault(cur[0], {})[name] = pick(o).detach().float().clone()
    return f


cur = ["fp32"]
w2 = model.wav2vec2
for i, cl in enumerate(w2.feature_extractor.conv_layers):
    cl.register_forward_hook(hook(f"conv{i}", lambda o: o.transpose(1, 2)))  # -> [B, T, C]
w2.feature_projection.register_forward_hook(hook("proj", lambda o: o[0]))
w2.encoder.pos_conv_embed.register_forward_hook(hook("posconv_branch"))
for l in list(range(0, L, 6)) + [L - 1]:
    w2.encoder.layers[l].register_forward_hook(hook(f"layer{l}", lambda o: o[0] if isinstance(o, tuple) else o))
w2.encoder.layer_norm.register_forward_hook(hook("final"))

with torch.no_grad():
    a = model(input_values=iv, attention_mask=am, labels=labels)
    cur[0] = "bf16"
    with torch.autocast("cpu", dtype=torch.bfloat16):
        b = model(input_values=iv, attention_mask=am, labels=labels)
la, lb = a.logits.float(), b.logits.float()
out = {}
print(f"{key}: per-stage relative RMS error of HF bf16-autocast against HF fp32")
for name, ra in stages["fp32"].items():
    rb = stages["bf16"][name]
    dl_ = rb.double() - ra.double()
    rel = float(dl_.norm() / ra.double().norm())
    T = dl_.shape[-2]
    coh = float(dl_.reshape(-1, T, dl_.shape[-1]).mean(dim=1).norm() * T ** 0.5 / dl_.norm())
    out["rel_" + name], out["coh_" + name] = rel, coh
    print(f"  {name:16s} {rel:.4e}   coherence {coh:5.2f}")
dl_ = lb.double() - la.double()
rel_logits = float(dl_.norm() / la.double().norm())
out["rel_logits"] = rel_logits
out["coh_logits"] = float(dl_.mean(dim=1).norm() * dl_.shape[1] ** 0.5 / dl_.norm())
print(f"  {'logits':16s} {rel_logits:.4e}   coherence {out['coh_logits']:5.2f}   max-abs {float((la - lb).abs().max()):.4f}")
rel = (float(b.loss) - float(a.loss)) / abs(float(a.loss))
# first-order decomposition of the loss difference: g = d loss / d logits at the fp32 logits
lr = la.clone().requires_grad_(True)
ref.ctc_loss(lr, labels, [lr.shape[1]], cfg)[0].backward()
gq = lr.grad.double()
dl = (lb - la).double()
first = float((gq * dl).sum())
mean_t = dl.mean(dim=1, keepdim=True)  # per-class shift common to all frames
coherent = float((gq * mean_t).sum())
print(f"  loss fp32 {float(a.loss):.4f}  bf16 {float(b.loss):.4f}  signed rel {rel:+.3e}; first-order sum(g*delta) = {first:+.4f} "
      f"of {float(b.loss) - float(a.loss):+.4f}, of which the per-class time-mean of delta carries {coherent:+.4f}")
out.update(loss_fp32=float(a.loss), loss_bf16=float(b.loss), logits_fp32=la.numpy(), logits_bf16=lb.numpy(),
           dlogits_fp32=lr.grad.numpy())
# the four ragged utterances of tests/test_fulldepth_gpu.py: per-utterance loss of both HF paths
waves = []
for n in [160_000, 131_200, 99_840, 147_520]:
    w_ = (0.1 * torch.randn(n, generator=g)).clamp(-1, 1)
    waves.append((w_ / w_.abs().max()).numpy())
iv4, am4 = ref.zero_mean_unit_var_norm(waves)
iv4, am4 = torch.from_numpy(iv4), torch.from_numpy(am4).long()
lab4 = torch.full((4, 90), -100, dtype=torch.int64)
for b_, L_ in enumerate((90, 70, 48, 81)):
    lab4[b_, :L_] = torch.randint(0, 42, (L_,), generator=g)
stages.clear()
with torch.no_grad():
    la4 = model(input_values=iv4, attention_mask=am4).logits.float()
    with torch.autocast("cpu", dtype=torch.bfloat16):
        lb4 = model(input_values=iv4, attention_mask=am4).logits.float()
flen4 = ref.feat_extract_output_lengths(am4.sum(-1), cfg)
nll_a = ref.ctc_loss(la4, lab4, flen4, cfg)[1]
nll_b = ref.ctc_loss(lb4, lab4, flen4, cfg)[1]
rel4 = ((nll_b - nll_a) / nll_a).tolist()
tot_a, tot_b = float(a.loss) + float(nll_a.sum()), float(b.loss) + float(nll_b.sum())
print("  four ragged utterances, HF bf16 vs HF fp32, signed rel per utterance: " + ", ".join(f"{v:+.2e}" for v in rel4) +
      f"; the five together {(tot_b - tot_a) / tot_a:+.2e}")
out.update(nll4_fp32=nll_a.numpy(), nll4_bf16=nll_b.numpy())
dst = ROOT / "tests" / "golden" / ("w2v2_cfg2_bf16_noise.npz" if key == "wav2vec2-large" else f"w2v2_{key}_bf16_noise.npz")
np.savez_compressed(dst, **{k: np.asarray(v) for k, v in out.items()})
print("wrote", dst)
