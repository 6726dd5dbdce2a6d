"""AdamW kernel time at the XLS-R-2B parameter count (tuning aid)."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from coral_amd import ops
dev = "cuda:0"
n = 2_160_000_000
p = torch.randn(n, device=dev); m = torch.zeros(n, device=dev); v = torch.zeros(n, device=dev); g = torch.randn(n, device=dev)
p16 = torch.empty(n, dtype=torch.bfloat16, device=dev); gn = torch.ones(1, device=dev)
def step(i): ops.adamw_step(p, m, v, g, p16, n, 1e-4, 0.9, 0.98, 1e-8, 0.0, i, grad_scale=1.0, max_norm=1.0, gnorm_sq=gn)
for i in range(1, 4): step(i)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(4, 12): step(i)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 8
print(f"AdamW {n/1e9:.2f} G params: {ms:.2f} ms = {30.0 * n / ms / 1e9:.2f} TB/s")
