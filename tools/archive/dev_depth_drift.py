"""Where does the bf16 engine drift from the fp32 oracle through the depth of XLS-R-2B?  One 10 s utterance (the case of
tests/test_fulldepth_gpu.py): per stage the relative RMS error against the oracle, beside it the same figure for the
REFERENCE's own bf16 path (HF under torch.autocast(bfloat16) against HF fp32: tests/golden/w2v2_cfg2_bf16_noise.npz,
written by tools/dev_hf_bf16_noise.py in the build container), and the "coherence" of the error - the norm of its
time-mean times sqrt(T) over its norm: 1 for noise that is independent from frame to frame, up to sqrt(T) = 22 for an
offset common to all frames.  Then the loss error and its first-order decomposition sum(g * delta) over the logits,
with the part carried by the per-class time-mean of delta.  python tools/dev_depth_drift.py [key]"""
import os
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from coral_amd.wav2vec2 import CORAL_W2V2_SHAPES, Wav2Vec2CTCEngine, Wav2Vec2Shape  # noqa: E402
from oracle import wav2vec2_ref as ref  # noqa: E402  (a development tool, like the tests: the oracle is the checker)

key = sys.argv[1] if len(sys.argv) > 1 else "wav2vec2-large"
cfg = ref.W2V2Config(**ref.CORAL_SHAPES[key])
P = ref.synth_params(cfg)
g = torch.Generator().manual_seed(4242)
x = (0.1 * torch.randn(160_000, generator=g)).clamp(-1, 1)
iv, am = ref.zero_mean_unit_var_norm([(x / x.abs().max()).numpy()])
iv, am = torch.from_numpy(iv), torch.from_numpy(am).long()
labels = torch.randint(0, 42, (1, 96), generator=g)
eng = Wav2Vec2CTCEngine(Wav2Vec2Shape(**CORAL_W2V2_SHAPES[key]), "cuda:0")
eng.load_state_dict(P)
eng.zero_grad()
out = eng(iv, am, labels)
torch.cuda.synchronize()
w = eng._saved["w"]
col = {}
with torch.no_grad():
    loss_ref, logits_ref, _ = ref.forward_loss(iv, am, labels, P, cfg, collect=col)
noise = ROOT / "tests" / "golden" / ("w2v2_cfg2_bf16_noise.npz" if key == "wav2vec2-large" else f"w2v2_{key}_bf16_noise.npz")
hf = dict(np.load(noise)) if noise.exists() else {}
print(f"{key}  CA_CONV_F32={os.environ.get('CA_CONV_F32', '1')}   (HF bf16 column: autocast vs fp32, same utterance and weights)")


def cmp(name, a, b, hf_key=None):
    """a, b: [..., T, C] (time second to last)"""
    a, b = a.double(), b.double()
    dl = a - b
    rel = float(dl.norm() / b.norm())
    T = dl.shape[-2]
    coh = float(dl.reshape(-1, T, dl.shape[-1]).mean(dim=1).norm() * T ** 0.5 / dl.norm())
    s = float((a.flatten() @ b.flatten()) / (b.flatten() @ b.flatten()))
    h = hf.get("rel_" + (hf_key or name))
    hc = hf.get("coh_" + (hf_key or name))
    print(f"{name:12s} rel err {rel:.4e}  (HF bf16 {float(h):.4e})" if h is not None else f"{name:12s} rel err {rel:.4e}  (HF bf16    -     )",
          f" coherence {coh:5.2f}" + (f" (HF {float(hc):5.2f})" if hc is not None else ""),
          f" best-fit scale {s:.5f}  |ref| rms {float(b.pow(2).mean().sqrt()):.3f}")


L, d = cfg.num_hidden_layers, cfg.hidden_size
B = 1
for i in range(7):
    C = cfg.conv_dim[i]
    cmp(f"conv{i}", w["a"][i].float().cpu().view(B, -1, C), col[f"conv{i}"])
cmp("proj", w["h0"].float().cpu().view(B, -1, d), col["proj"])
cmp("posconv", w["h"][0].float().cpu().view(B, -1, d), col["posconv"])
for l in list(range(0, L, 6)) + [L - 1]:
    cmp(f"layer{l}", w["h"][l + 1].float().cpu().view(B, -1, d), col[f"layer{l}"])
cmp("final", w["hf"].float().cpu().view(B, -1, d), col["final"])
le_ = out.logits.float().cpu()
cmp("logits", le_, logits_ref)
print("loss", float(out.loss), float(loss_ref), "signed rel", (float(out.loss) - float(loss_ref)) / float(loss_ref))
# first-order decomposition of the loss difference
lr = logits_ref.clone().requires_grad_(True)
ref.ctc_loss(lr, labels, [lr.shape[1]], cfg)[0].backward()
gq = lr.grad.double()
dl = (le_ - logits_ref).double()
mean_t = dl.mean(dim=1, keepdim=True)
print(f"first-order sum(g*delta) = {float((gq * dl).sum()):+.4f} of {float(out.loss) - float(loss_ref):+.4f}; the per-class "
      f"time-mean of delta carries {float((gq * mean_t).sum()):+.4f}")
V = cfg.vocab_size
if "logits_bf16" in hf:
    dh = torch.from_numpy(hf["logits_bf16"] - hf["logits_fp32"]).double()
    mh = dh.mean(dim=1, keepdim=True)
    print(f"HF bf16: sum(g*delta) = {float((gq * dh).sum()):+.4f}, time-mean part {float((gq * mh).sum()):+.4f}; "
          f"oracle vs HF fp32 logits max-abs {float((logits_ref - torch.from_numpy(hf['logits_fp32'])).abs().max()):.2e}")
    sg = gq.sum(dim=1).flatten()
    print("per class: sum_t g | engine time-mean delta | HF-bf16 time-mean delta   (largest |sum_t g| first)")
    for v in sg.abs().argsort(descending=True)[:10].tolist():
        print(f"   class {v:2d}: {float(sg[v]):+9.3f} | {float(mean_t[0, 0, v]):+.5f} | {float(mh[0, 0, v]):+.5f}")
    print(f"   rms over classes of the time-mean delta: engine {float(mean_t.pow(2).mean().sqrt()):.5f}, HF bf16 {float(mh.pow(2).mean().sqrt()):.5f}; "
          f"correlation {float((mean_t.flatten() @ mh.flatten()) / (mean_t.norm() * mh.norm())):+.3f}")
# the CTC gradient on both sides' own logits
cmp("dlogits", w["dlogits"].float().cpu().view(1, -1, w["Vp"])[:, :, :V], lr.grad)
# and the oracle's CTC on the ENGINE's logits: separates the CTC kernel from the logits drift
le = out.logits.float().cpu().clone().requires_grad_(True)
l2 = ref.ctc_loss(le, labels, [le.shape[1]], cfg)[0]
l2.backward()
cmp("dlogits@eng", w["dlogits"].float().cpu().view(1, -1, w["Vp"])[:, :, :V], le.grad)
print("oracle CTC on engine logits", float(l2.detach()), "engine", float(out.loss))
dst = ROOT / "gpurun_out"
dst.mkdir(exist_ok=True)
np.savez_compressed(dst / f"engine_logits_{key}_convf32_{os.environ.get('CA_CONV_F32', '1')}.npz", logits=le_.numpy(), ref=logits_ref.numpy(),
                    final=w["hf"].float().cpu().numpy(), final_ref=col["final"].numpy())
