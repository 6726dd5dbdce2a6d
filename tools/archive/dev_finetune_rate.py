"""Throughput of the path a CoRal user runs - `finetune_asr_model.py model=wav2vec2-large datasets=synthetic` (raw PCM
examples -> CoralTrainer -> device input pipeline with augmentation -> engine) - next to bench.py's `value` for the same
shape (8 x 10 s per step, SpecAugment + activation dropout on).  python tools/dev_finetune_rate.py [model] [steps]"""
import sys
import tempfile
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "scripts"))
import finetune_asr_model  # noqa: E402

model = sys.argv[1] if len(sys.argv) > 1 else "wav2vec2-large"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 24
with tempfile.TemporaryDirectory() as tmp:
    t0 = time.time()
    res = finetune_asr_model.main([f"model={model}", "datasets=synthetic", f"models_dir={tmp}", "model_id=rate",
                                   f"max_steps={steps}", "total_batch_size=8", "per_device_batch_size=8", "padding=max_length",
                                   "max_seconds_per_example=10.0", "min_seconds_per_example=10.0", "logging_steps=8",
                                   "eval_steps=100000", "save_total_limit=0", "model.layerdrop=0.0"])
    torch.cuda.synchronize()
    hist = [h for h in res["history"] if "loss" in h]
    a, b = hist[1], hist[-1]  # (the first logged step carries the warm-up: allocation, first launches)
    n = b["step"] - a["step"]
    dt = b["elapsed"] - a["elapsed"]
    tr = res["trainer"]
    print(f"{model}: {n} steps of 8 x 10 s in {dt:.3f} s = {dt / n * 1e3:.2f} ms/step = {80.0 * n / dt:.1f} audio-s/s "
          f"(pipeline batches {tr.pipeline_batches}, augmentation {'on' if tr._pipe.augment is not None else 'off'}; "
          f"whole run incl. init and final save {time.time() - t0:.0f} s)")
