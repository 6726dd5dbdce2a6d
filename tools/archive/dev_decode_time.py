"""Marginal per-token cost of Whisper greedy decoding (eager K|V-cache loop vs HIP-graph replay)."""
import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from coral_amd.whisper import CORAL_WHISPER_SHAPES, WhisperEngine, WhisperShape

name = sys.argv[1] if len(sys.argv) > 1 else "whisper-medium"
shape = WhisperShape(**CORAL_WHISPER_SHAPES[name])
eng = WhisperEngine(shape, "cuda:0")
g = torch.Generator(device="cuda:0").manual_seed(1)
for n in eng.exported_names():
    v = eng.store.view(n)
    if n.endswith("layer_norm.weight"):
        v.fill_(1.0)
    elif n.endswith(".bias"):
        v.zero_()
    else:
        v.normal_(0.0, 0.02, generator=g)
eng.refresh_compute_weights()
feats = torch.randn(8, shape.num_mel_bins, 3000, device="cuda:0") * 0.5
prefix = [50258, 50285, 50359, 50363]
for use_graph in (False, True):
    res = {}
    for L in (36, 132):
        eng.generate(feats, prefix, L, use_graph=use_graph)  # warm
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = eng.generate(feats, prefix, L, use_graph=use_graph)
        torch.cuda.synchronize()
        res[L] = (time.perf_counter() - t0, len(out[0]))
    (ta, la), (tb, lb) = res[36], res[132]
    print(f"{name} graph={use_graph}: {ta*1e3:.1f} ms for {la} tokens, {tb*1e3:.1f} ms for {lb} tokens -> "
          f"{(tb-ta)/(max(1, lb-la))*1e3:.2f} ms/token marginal")
