"""Where does the host wait inside a training step?  Wraps every coral_amd.ops entry point with a host timer and
prints the longest calls of one steady-state step with their position in the launch sequence."""
import sys, time
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import bench
from coral_amd import ops, specaugment
from coral_amd.trainer import DataParallelTrainer
from coral_amd.wav2vec2 import CORAL_W2V2_SHAPES, Wav2Vec2CTCEngine, Wav2Vec2Shape

dev = torch.device("cuda:0")
shape = Wav2Vec2Shape(**CORAL_W2V2_SHAPES["wav2vec2-large"], activation_dropout=0.1, layerdrop=0.0)
eng = Wav2Vec2CTCEngine(shape, dev)
bench.init_random_(eng, 4242)
tr = DataParallelTrainer(eng, learning_rate=1e-4, betas=(0.9, 0.98), max_grad_norm=1.0, warmup_steps=1000, max_steps=100000)
batch, _ = bench.synth_batch(8, 10.0, 0, dev)
B, N = batch["input_values"].shape
T = eng.conv_lengths(N)[-1]
rng = np.random.RandomState(1)


def mk():
    mb = dict(batch)
    mt, mf = specaugment.sample_masks(B, T, shape.hidden_size, [T] * B, 0.5, 10, 0.5, 64, rng=rng)
    mb["mask_time"] = torch.from_numpy(mt)
    mb["mask_feature"] = torch.from_numpy(mf)
    return [mb]


for _ in range(3):
    tr.train_step(mk())
torch.cuda.synchronize()
rec = []
import types
for name in dir(ops):
    fn = getattr(ops, name)
    if isinstance(fn, types.FunctionType) and not name.startswith("_") and name not in ("lib", "check"):
        def wrap(f, nm):
            def g(*a, **k):
                t0 = time.perf_counter()
                r = f(*a, **k)
                rec.append((time.perf_counter() - t0, nm, len(rec)))
                return r
            return g
        setattr(ops, name, wrap(fn, name))
for _ in range(2):
    tr.train_step(mk())
rec.clear()
t0 = time.perf_counter()
tr.train_step(mk())
host = time.perf_counter() - t0
torch.cuda.synchronize()
tot = sum(r[0] for r in rec)
print(f"host time of the step {host * 1e3:.1f} ms; inside ops calls {tot * 1e3:.1f} ms over {len(rec)} calls; median call {sorted(r[0] for r in rec)[len(rec) // 2] * 1e6:.1f} us")
for dt, nm, i in sorted(rec, reverse=True)[:15]:
    print(f"  call #{i:5d} {nm:24s} {dt * 1e3:8.3f} ms")
big = sum(r[0] for r in rec if r[0] > 200e-6)
print(f"calls over 200 us: {sum(1 for r in rec if r[0] > 200e-6)} totalling {big * 1e3:.1f} ms")
