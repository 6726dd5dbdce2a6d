import sys, time, torch
sys.path.insert(0, "/root/repo")
import bench
from coral_amd import whisper as W
dev = torch.device("cuda:0")
eng, shape, waves, _ = bench.whisper_setup_engine("whisper-medium", dev, 0, 16)
feats = eng.log_mel(waves)
prefix = [50258, 50285, 50359, 50363]
eng.generate(feats, prefix, 44)
orig_graph = torch.cuda.graph
tot = [0.0]
class Timed:
    def __init__(self, g): self.cm = orig_graph(g)
    def __enter__(self):
        torch.cuda.synchronize(); self.t = time.perf_counter(); return self.cm.__enter__()
    def __exit__(self, *a):
        r = self.cm.__exit__(*a); torch.cuda.synchronize(); tot[0] += time.perf_counter() - self.t; return r
torch.cuda.graph = Timed
for n in (12, 44, 44, 229):
    tot[0] = 0.0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    enc = eng.encode(feats); kv = eng.cross_kv(enc); torch.cuda.synchronize(); t1 = time.perf_counter()
    ids = eng.generate(feats, prefix, n); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"max_length {n}: encode+cross_kv {1e3*(t1-t0):.2f} ms, generate {1e3*(t2-t1):.2f} ms of which graph capture {1e3*tot[0]:.2f} ms")
