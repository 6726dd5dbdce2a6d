"""Per-shape GEMM time inside one training step (monkeypatches ops.gemm with event pairs; tuning aid)."""
import collections
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench  # noqa: E402
from coral_amd import ops  # noqa: E402
from coral_amd.trainer import DataParallelTrainer  # noqa: E402
from coral_amd.wav2vec2 import CORAL_W2V2_SHAPES, Wav2Vec2CTCEngine, Wav2Vec2Shape  # noqa: E402

model = sys.argv[1] if len(sys.argv) > 1 else "wav2vec2-large"
dev = torch.device("cuda:0")
shape = Wav2Vec2Shape(**CORAL_W2V2_SHAPES[model], activation_dropout=0.1, layerdrop=0.0)
eng = Wav2Vec2CTCEngine(shape, dev)
bench.init_random_(eng, 4242)
tr = DataParallelTrainer(eng, overlap_optimizer=False)
batch, _ = bench.synth_batch(8, 10.0, 0, dev)
for _ in range(2):
    tr.train_step([dict(batch)])
torch.cuda.synchronize()
rec = []
orig = ops.gemm


def timed(A, B, C, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    orig(A, B, C, **kw)
    e1.record()
    key = (kw["M"], kw["N"], kw["K"], kw.get("a_layout", 0), kw.get("b_layout", 0), kw.get("batch1", 1) * kw.get("batch2", 1),
           kw.get("epilogue", 0))
    rec.append((key, e0, e1))


ops.gemm = timed
tr.train_step([dict(batch)])
torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for key, e0, e1 in rec:
    a = agg[key]
    a[0] += 1
    a[1] += e0.elapsed_time(e1)
tot = sum(v[1] for v in agg.values())
print(f"total {tot:.2f} ms over {len(rec)} launches")
for key, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
    M, N, K, al, bl, nb, epi = key
    fl = 2.0 * M * N * K * nb * n
    print(f"M{M:7d} N{N:6d} K{K:6d} {'NT NN TN TT'.split()[al * 2 + bl]} b{nb:3d} epi{epi} x{n:3d}: {ms:7.2f} ms {ms / n * 1e3:7.1f} us "
          f"{fl / ms / 1e9:7.1f} TFLOP/s")
