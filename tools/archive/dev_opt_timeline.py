"""Where the optimiser sits under the next forward (XLS-R-2B bench step, no profiler: profilers serialise the streams):
events on the main and optimiser streams around forward / backward / optimiser of a few steady-state steps.
    [CA_ADAMW_BLOCKS=256] python tools/dev_opt_timeline.py"""
import os
import sys
import types
from pathlib import Path

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import torch  # noqa: E402

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import bench  # noqa: E402
from coral_amd import specaugment  # noqa: E402
from coral_amd.trainer import DataParallelTrainer  # noqa: E402
from coral_amd.wav2vec2 import CORAL_W2V2_SHAPES, Wav2Vec2CTCEngine, Wav2Vec2Shape  # noqa: E402
import numpy as np  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
shape = Wav2Vec2Shape(**CORAL_W2V2_SHAPES["wav2vec2-large"], activation_dropout=0.1, layerdrop=0.0)
eng = Wav2Vec2CTCEngine(shape, dev)
bench.init_random_(eng, 4242)
tr = DataParallelTrainer(eng, learning_rate=1e-4, betas=(0.9, 0.98), max_grad_norm=1.0, warmup_steps=1000, max_steps=100_000)
batch, lens = bench.synth_batch(8, 10.0, 0, dev)
B, N = batch["input_values"].shape
T = eng.conv_lengths(N)[-1]
fl = [eng.conv_lengths(int(n))[-1] for n in lens]
rng = np.random.RandomState(1)


def mb():
    m = dict(batch)
    mt, mf = specaugment.sample_masks(B, T, shape.hidden_size, fl, 0.5, 10, 0.5, 64, rng=rng)
    m["mask_time"], m["mask_feature"] = torch.from_numpy(mt), torch.from_numpy(mf)
    return [m]


E = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731
marks = []
orig_model_call = tr.model.__class__.__call__
state = {}


def model_call(self, *a, **k):
    e0 = E(); e0.record()
    out = orig_model_call(self, *a, **k)
    e1 = E(); e1.record()
    state["fwd"] = (e0, e1)
    return out


tr.model.__class__.__call__ = model_call
orig_opt = tr.optimizer_step


def opt_step():
    e0 = E(); e0.record()
    orig_opt()
    e1 = E(); e1.record()  # main stream after launching the optimiser
    e2 = E(); e2.record(tr.opt_stream) if tr.opt_stream is not None else e2.record()
    state["opt"] = (e0, e1, e2)


tr.optimizer_step = opt_step
for _ in range(3):
    tr.train_step(mb())
torch.cuda.synchronize()
recs = []
for _ in range(6):
    prev_opt = state.get("opt")
    tr.train_step(mb())
    recs.append((prev_opt, state["fwd"], state["opt"]))
torch.cuda.synchronize()
fw, op, st, bw = [], [], [], []
for prev_opt, (f0, f1), (o0, o1, o2) in recs[1:]:
    p0, p1, p2 = prev_opt
    fw.append(f0.elapsed_time(f1)); op.append(p0.elapsed_time(p2)); st.append(p0.elapsed_time(o0)); bw.append(f1.elapsed_time(o0))
m = lambda x: sum(x) / len(x)  # noqa: E731
print(f"forward {m(fw):6.2f} | optimiser start -> last kernel done {m(op):6.2f} | backward {m(bw):6.2f} | step {m(st):6.2f}")
