"""Main-stream time between the end of a step's backward (+ gradient norm) and the first conv kernel of the next
forward, measured with events in an un-profiled run."""
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
ends, starts, firsts = [], [], []
orig_opt = tr.optimizer_step
orig_conv0 = ops.conv0_fwd
orig_fwd = eng.forward


def opt():
    e = torch.cuda.Event(enable_timing=True); e.record(); ends.append(e)
    orig_opt()


def conv0(*a, **k):
    e = torch.cuda.Event(enable_timing=True); e.record(); starts.append(e)
    return orig_conv0(*a, **k)


def fwd(*a, **k):
    e = torch.cuda.Event(enable_timing=True); e.record(); firsts.append(e)
    return orig_fwd(*a, **k)


tr.optimizer_step = opt
ops.conv0_fwd = conv0
eng.forward = fwd
tr.model = eng
for _ in range(6):
    tr.train_step(mk())
torch.cuda.synchronize()
for i in range(1, 5):
    print(f"step {i}: backward end -> forward entry {ends[i].elapsed_time(firsts[i + 1]):6.2f} ms, -> conv0 {ends[i].elapsed_time(starts[i + 1]):6.2f} ms; "
          f"step period {starts[i].elapsed_time(starts[i + 1]):6.2f} ms")
