"""Host-side cost of one training step: time spent inside train_step() (enqueue only, no sync) against the GPU
step time, and a cProfile of the hot Python frames.  usage: python tools/dev_host_time.py [model]"""
import cProfile
import pstats
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import bench  # noqa: E402
from coral_amd import specaugment  # noqa: E402
from coral_amd.trainer import DataParallelTrainer  # noqa: E402
from coral_amd.wav2vec2 import CORAL_W2V2_SHAPES, Wav2Vec2CTCEngine, Wav2Vec2Shape  # noqa: E402

model = sys.argv[1] if len(sys.argv) > 1 else "wav2vec2-large"
dev = torch.device("cuda:0")
shape = Wav2Vec2Shape(**CORAL_W2V2_SHAPES[model], activation_dropout=0.1, layerdrop=0.0)
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
host, tmk = 0.0, 0.0
t0 = time.perf_counter()
for _ in range(8):
    a = time.perf_counter()
    b = mk()
    c = time.perf_counter()
    tr.train_step(b)
    d = time.perf_counter()
    tmk += c - a
    host += d - c
torch.cuda.synchronize()
wall = time.perf_counter() - t0
print(f"{model}: wall {wall / 8 * 1e3:.2f} ms/step, host inside train_step {host / 8 * 1e3:.2f} ms/step, batch prep {tmk / 8 * 1e3:.2f} ms/step")
pr = cProfile.Profile()
pr.enable()
for _ in range(4):
    tr.train_step(mk())
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(18)
