"""Which host lines launch torch fill kernels inside a training step (dev aid): wraps the zero / fill entry points for
one step of the chosen model and prints the callers.  usage: python tools/dev_trace_fills.py [whisper-medium|wav2vec2-large]"""
import collections
import sys
import traceback
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import bench  # noqa: E402

model = sys.argv[1] if len(sys.argv) > 1 else "whisper-medium"
counts = collections.Counter()
on = [False]


def wrap(owner, name):
    orig = getattr(owner, name)

    def f(*a, **k):
        if on[0]:
            fr = [x for x in traceback.extract_stack()[:-1] if "/coral_amd/" in x.filename or x.filename.endswith("bench.py")]
            if fr:
                counts[(name, f"{Path(fr[-1].filename).name}:{fr[-1].lineno}")] += 1
        return orig(*a, **k)

    setattr(owner, name, f)


for owner, name in ((torch, "zeros"), (torch, "full"), (torch, "zeros_like"), (torch, "ones"), (torch.Tensor, "zero_"),
                    (torch.Tensor, "fill_"), (torch.Tensor, "new_zeros"), (torch.Tensor, "masked_fill"), (torch, "arange"),
                    (torch.Tensor, "copy_"), (torch.Tensor, "clamp"), (torch.Tensor, "to")):
    wrap(owner, name)

if model.startswith("whisper"):
    eng, shape, waves, labels = bench.whisper_setup_engine(model, "cuda:0", 0, 8)
    from coral_amd.trainer import DataParallelTrainer
    tr = DataParallelTrainer(eng, learning_rate=1e-5, warmup_steps=0, max_steps=100, max_grad_norm=1.0)
    for i in range(3):
        on[0] = i == 2
        tr.train_step([dict(input_features=eng.log_mel(waves), labels=labels)])
    torch.cuda.synchronize()
on[0] = False
for (name, where), c in sorted(counts.items(), key=lambda kv: -kv[1])[:40]:
    print(f"{c:5d}  {name:12s} {where}")
