"""whisper-large-turbo step, bf16 and fp8 alternating INSIDE one process (what bench.py's also_turbo does once):
python tools/dev_turbo_inproc.py"""
import sys
import types
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench  # noqa: E402

args = types.SimpleNamespace(batch=8, steps=4, warmup=2, grad_wire="fp32", zero_stage=0, decode_tokens=32)
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
for i in range(3):
    for fp8 in (False, True):
        r = bench.whisper_measure("whisper-large-turbo", args, 1, 0, dev, decode=False, fp8=fp8, B=8, steps=4, warmup=2)
        print("fp8 " if fp8 else "bf16", round(r["ms_per_step"], 2), flush=True)
        del r
        torch.cuda.empty_cache()
