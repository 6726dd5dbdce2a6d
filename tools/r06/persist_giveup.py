"""Round 6: ca_whisper_decode_token when it cannot have every CU - 16 idle workgroups holding 96 KiB of LDS each
(ca_debug_cu_hog, on a side stream, resident for several seconds) sit on 16 CUs while the launch starts.  The launch
must come back (bounded spins), say so in `status`, and `generate` must raise; afterwards, with the CUs free again, the
same state decodes normally.  Prints the wall time of the launch that gave up."""
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from coral_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
B = 8
eng, shape, waves, _ = bench.whisper_setup_engine("whisper-xxsmall", dev, 0, B)
kv = eng.cross_kv(eng.encode(eng.log_mel(waves)))
cache = eng.new_decode_cache(B, 44)
g = eng._graph_state(cache, kv, shape.pad_token_id, shape.eos_token_id)
sup = torch.zeros(shape.vocab_size, dtype=torch.uint8, device=dev)
base = eng.decode_step(torch.tensor([[50258, 50285, 50359, 50363]] * B, dtype=torch.int64, device=dev), kv, cache).contiguous()
ops.argmax_masked(base, sup, g["nxt"], B, shape.vocab_size, shape.vocab_size)
g["tok"].copy_(g["nxt"]); g["pos"].fill_(4); g["klen"].fill_(5)
ps = eng._persistent_state(cache, g, sup)
ops.whisper_decode_token(ps["desc"])
torch.cuda.synchronize()
print("free chip: status", ps["status"].tolist(), "pos", g["pos"].tolist()[:2])
side = torch.cuda.Stream()
hold_s = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for _ in range(hold_s // 2):  # (the hog holds for at most 2 s per launch: a chain of them)
    ops.check(ops.lib().ca_debug_cu_hog(16, 256, 96 * 1024, 2000.0, side.cuda_stream), "ca_debug_cu_hog")
time.sleep(0.2)
t0 = time.time()
ops.whisper_decode_token(ps["desc"])
torch.cuda.current_stream().synchronize()
dt = time.time() - t0
print(f"16 CUs held: the launch came back after {dt:.2f} s, status {ps['status'].tolist()}, pos {g['pos'].tolist()[:2]}")
side.synchronize()
ps["status"].zero_()
ops.whisper_decode_token(ps["desc"])
torch.cuda.synchronize()
print("free chip again: status", ps["status"].tolist(), "pos", g["pos"].tolist()[:2])
