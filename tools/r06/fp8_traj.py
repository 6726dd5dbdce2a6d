"""Round 6 (review item 6, second half of its "done"): whisper-large-turbo (configs[4]'s shape), 8 x 30 s, six optimiser steps
from the same initialisation and the same batches, bf16 against enable_fp8_forward(): per-step loss and pre-clip gradient
norm, and the cosine between the two runs' gradients at every step (identical parameters at step 1; the trajectories then
drift apart with their own updates, so later cosines also contain that drift)."""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from coral_amd.trainer import DataParallelTrainer  # noqa: E402

dev = torch.device("cuda:0")
B = 8
runs = {}
for mode in ("bf16", "fp8"):
    torch.manual_seed(0)
    eng, shape, waves, labels = bench.whisper_setup_engine("whisper-large-turbo", dev, 0, B)
    if mode == "fp8":
        eng.enable_fp8_forward()
    tr = DataParallelTrainer(eng, learning_rate=6e-6, betas=(0.9, 0.98), warmup_steps=2, max_steps=100)
    hist = []
    for step in range(6):
        feats = eng.log_mel(waves)
        loss = tr.train_step([dict(input_features=feats, labels=labels)])
        tr.finish()
        torch.cuda.synchronize()
        hist.append((float(loss), float(tr.grad_norm()), eng.store.g32.clone() if step in (0, 1, 5) else None))
    runs[mode] = hist
    tr.close()
    del eng, tr
    torch.cuda.empty_cache()
print("step   loss bf16   loss fp8   rel diff   |g| bf16    |g| fp8    cos(g bf16, g fp8)")
for i, (a, b) in enumerate(zip(runs["bf16"], runs["fp8"])):
    cos = ""
    if a[2] is not None:
        cos = f"{torch.nn.functional.cosine_similarity(a[2].flatten().double(), b[2].flatten().double(), dim=0).item():.5f}"
    print(f"{i + 1:4d} {a[0]:11.4f} {b[0]:10.4f} {abs(a[0] - b[0]) / abs(a[0]):10.2e} {a[1]:10.3f} {b[1]:10.3f}    {cos}")
