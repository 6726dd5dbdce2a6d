"""Where does the bf16 engine drift from the fp32 oracle through the depth of XLS-R-2B?  One 10 s utterance (the case of
tests/test_fulldepth_gpu.py): per layer the relative error of the residual stream, its best-fit scale against the
oracle's, and the same for the final LayerNorm output, the logits and the CTC gradient.  python tools/dev_depth_drift.py [key]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
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


def cmp(name, a, b):
    a, b = a.double().flatten(), b.double().flatten()
    s = float((a @ b) / (b @ b))
    rel = float((a - b).norm() / b.norm())
    res = float((a - s * b).norm() / b.norm())
    print(f"{name:12s} rel err {rel:.4e}  best-fit scale {s:.5f}  residual after scaling {res:.4e}  |ref| rms {float(b.pow(2).mean().sqrt()):.3f} max {float(b.abs().max()):.1f}")


L, d = cfg.num_hidden_layers, cfg.hidden_size
cmp("posconv", w["h"][0].float().cpu().view(1, -1, d), col["posconv"])
for l in list(range(0, L, 6)) + [L - 1]:
    cmp(f"layer{l}", w["h"][l + 1].float().cpu().view(1, -1, d), col[f"layer{l}"])
cmp("final LN", w["hf"].float().cpu().view(1, -1, d), col["final"])
cmp("logits", out.logits.float().cpu(), logits_ref)
print("loss", float(out.loss), float(loss_ref), "rel", abs(float(out.loss) - float(loss_ref)) / float(loss_ref))
# the CTC gradient on both sides' own logits
lr = logits_ref.clone().requires_grad_(True)
ref.ctc_loss(lr, labels, [lr.shape[1]], cfg)[0].backward()
V = cfg.vocab_size
cmp("dlogits", w["dlogits"].float().cpu().view(1, -1, w["Vp"])[:, :, :V], lr.grad)
# and the oracle's CTC on the ENGINE's logits: separates the CTC kernel from the logits drift
le = out.logits.float().cpu().clone().requires_grad_(True)
l2 = ref.ctc_loss(le, labels, [le.shape[1]], cfg)[0]
l2.backward()
cmp("dlogits@eng", w["dlogits"].float().cpu().view(1, -1, w["Vp"])[:, :, :V], le.grad)
print("oracle CTC on engine logits", float(l2), "engine", float(out.loss))
