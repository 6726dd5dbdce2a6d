"""Host cost of a kernel launch on the main stream while a side stream is busy / idle."""
import sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from coral_amd import ops
dev = "cuda:0"
x = torch.zeros(3992 * 1920, dtype=torch.bfloat16, device=dev)
y = torch.zeros_like(x)
g = torch.ones(1920, device=dev); b = torch.zeros(1920, device=dev)
big = torch.zeros(600_000_000, device=dev)   # 2.4 GB: one pass ~1 ms
side = torch.cuda.Stream()


def main_launches(n=200):
    t0 = time.perf_counter()
    for _ in range(n):
        ops.layernorm_fwd(x, g, b, y, None, 3992, 1920)
    return (time.perf_counter() - t0) / n * 1e6


torch.cuda.synchronize()
print(f"side idle:            {main_launches():6.1f} us/launch (host)")
torch.cuda.synchronize()
with torch.cuda.stream(side):
    for _ in range(30):
        big.add_(1.0)
print(f"side busy (30 passes): {main_launches():6.1f} us/launch (host)")
torch.cuda.synchronize()
# side busy, with cross-stream events as the optimizer does
evs = []
for _ in range(30):
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        big.add_(1.0)
        e = torch.cuda.Event(); e.record(); evs.append(e)
print(f"side busy + events:    {main_launches():6.1f} us/launch (host)")
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    with torch.cuda.stream(side):
        big[:1000].add_(1.0)
t1 = time.perf_counter()
print(f"side-stream tiny launches while idle: {(t1 - t0) / 50 * 1e6:.1f} us each")
torch.cuda.synchronize()
