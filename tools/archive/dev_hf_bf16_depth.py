"""How far does the REFERENCE's own bf16 path (transformers under torch.autocast(bfloat16), what `bf16=True` in
R/src/coral/wav2vec2.py:183-193 runs) drift from its fp32 path at the XLS-R-2B shape?  One 10 s utterance, the inputs
of tests/test_fulldepth_gpu.py.  Runs in the build container only (imports transformers): python tools/dev_hf_bf16_depth.py [model-key]"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
sys.path.insert(0, str(Path(__file__).resolve().parent))
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
t0 = time.time()
with torch.no_grad():
    a = model(input_values=iv, attention_mask=am, labels=labels)
    t1 = time.time()
    with torch.autocast("cpu", dtype=torch.bfloat16):
        b = model(input_values=iv, attention_mask=am, labels=labels)
t2 = time.time()
la, lb = a.logits.float(), b.logits.float()
err = float((la - lb).abs().max())
cos = float((la.flatten().double() @ lb.flatten().double()) / (la.norm().double() * lb.norm().double()))
rel = abs(float(a.loss) - float(b.loss)) / abs(float(a.loss))
print(f"{key}: HF fp32 loss {float(a.loss):.4f} ({t1 - t0:.0f} s), HF bf16-autocast loss {float(b.loss):.4f} ({t2 - t1:.0f} s): "
      f"rel {rel:.2e}; logits max-abs diff {err:.4f}, cosine {cos:.6f}")
