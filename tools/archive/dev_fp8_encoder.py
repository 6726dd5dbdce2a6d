"""Whisper encoder forward (8 x 30 s) in bf16 and with the fp8 q|k|v / fc1 projections."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from coral_amd.whisper import CORAL_WHISPER_SHAPES, WhisperEngine, WhisperShape
for name in sys.argv[1:] or ["whisper-medium", "whisper-large-turbo"]:
    eng = WhisperEngine(WhisperShape(**CORAL_WHISPER_SHAPES[name]), "cuda:0")
    g = torch.Generator(device="cuda:0").manual_seed(1)
    for n in eng.exported_names():
        v = eng.store.view(n)
        if n.endswith("layer_norm.weight"): v.fill_(1.0)
        elif n.endswith(".bias"): v.zero_()
        else: v.normal_(0.0, 0.02, generator=g)
    eng.refresh_compute_weights()
    feats = torch.randn(8, eng.s.num_mel_bins, 3000, device="cuda:0") * 0.3
    res = {}
    for mode in ("bf16", "fp8"):
        eng.enable_fp8_encoder(mode == "fp8")
        for _ in range(2): out = eng.encode(feats)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): out = eng.encode(feats)
        e1.record(); torch.cuda.synchronize()
        res[mode] = (e0.elapsed_time(e1) / 5, out.float().clone())
    rel = ((res["fp8"][1] - res["bf16"][1]).norm() / res["bf16"][1].norm()).item()
    print(f"{name}: encoder forward bf16 {res['bf16'][0]:.2f} ms, fp8 q|k|v + fc1 {res['fp8'][0]:.2f} ms, relative difference of the states {rel:.4f}")
    del eng
