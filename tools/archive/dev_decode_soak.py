"""Soak / race check of the graph-replayed greedy decoder with the key split active (4 and 8 clips: 4 and 2 workgroups per
(clip, head)) and without it (16): the same clips decoded repeatedly must give the same ids every time, and the batch-8
run must agree with its clips decoded in the batch of 16.   python tools/dev_decode_soak.py [repeats]"""
import os
import sys
from pathlib import Path

os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import torch  # noqa: E402

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import bench  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
eng, shape, waves, labels = bench.whisper_setup_engine("whisper-medium", dev, 0, 16)
prefix = [50258, 50285, 50359, 50363]
feats16 = eng.log_mel(waves)
ref = {}
for B in (16, 8, 4):
    feats = feats16[:B].contiguous()
    first = eng.generate(feats, prefix, 4 + 48)
    bad = 0
    for i in range(reps):
        out = eng.generate(feats, prefix, 4 + 48)
        bad += out != first
    ref[B] = first
    print(f"B={B}: {reps} repeats, {bad} differ from the first run; distinct tokens in clip 0: {len(set(first[0]))}")
    assert bad == 0
same8 = sum(a == b for a, b in zip(ref[8], ref[16][:8]))
same4 = sum(a == b for a, b in zip(ref[4], ref[16][:4]))
print(f"clips identical between batch 8 (split 2) and batch 16 (no split): {same8}/8; batch 4 (split 4) and 16: {same4}/4")
