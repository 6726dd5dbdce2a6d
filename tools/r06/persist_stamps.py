"""Round 6: where a persistent decode launch spends its time - per phase, the time from the previous phase's end to the
seam being passed (wait) and from there to the phase's end (work), median / max over workgroups, summed per phase kind.
usage: python tools/r06/persist_stamps.py [model] [batch]"""
import os
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from coral_amd import ops  # noqa: E402

model = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].isdigit() else "whisper-medium"
B = int([a for a in sys.argv[1:] if a.isdigit()][0]) if any(a.isdigit() for a in sys.argv[1:]) else 16
dev = torch.device("cuda:0")
prefix = [50258, 50285, 50359, 50363]
eng, shape, waves, _ = bench.whisper_setup_engine(model, dev, 0, B)
kv = eng.cross_kv(eng.encode(eng.log_mel(waves)))
cache = eng.new_decode_cache(B, 44)
g = eng._graph_state(cache, kv, shape.pad_token_id, shape.eos_token_id)
sup = torch.zeros(shape.vocab_size, dtype=torch.uint8, device=dev)
base = eng.decode_step(torch.tensor([prefix] * B, dtype=torch.int64, device=dev), kv, cache).contiguous()
ops.argmax_masked(base, sup, g["nxt"], B, shape.vocab_size, shape.vocab_size)
g["tok"].copy_(g["nxt"]); g["pos"].fill_(4); g["klen"].fill_(5)
ps = eng._persistent_state(cache, g, sup)
G = torch.cuda.get_device_properties(dev).multi_processor_count
L = shape.decoder_layers
# thread 0 of every workgroup appends the shader clock at fixed points (csrc/decode.hip, dk_t): per layer 50 stamps
LAB = {"A": ["publish + ring + seam", "LayerNorm image written", "projection + epilogue done"],
       "B": ["publish + ring + seam", "attention done"],
       "C": ["publish + ring + seam", "(no LayerNorm)", "rows gathered, projection + epilogue done"]}
LAB["D"] = LAB["G"] = LAB["A"]
LAB["F"] = LAB["H"] = LAB["C"]
LAB["E"] = ["publish + ring + first tiles + seam", "attention done"]
import os
FINE = os.environ.get("CA_DECODE_STAMP_FINE", "0") != "0"
if FINE:  # every phase: ... its own work ..., then "published", "ring advanced"; in front: "at the seam's barrier"
    for k in list(LAB):
        LAB[k] = ["at the seam's barrier (from the previous 'ring advanced')", "barrier passed"] + LAB[k][1:] + ["published", "ring advanced"]
ORDER = "ABCDEFGH"
per_layer = sum(len(LAB[k]) for k in ORDER)
nst = per_layer * L + 64
buf = torch.zeros(G * nst, dtype=torch.int64, device=dev)
for _ in range(3):
    ops.whisper_decode_token(ps["desc"])
ops.lib().ca_debug_decode_stamps(buf.data_ptr(), nst)
ops.whisper_decode_token(ps["desc"])
torch.cuda.synchronize()
ops.lib().ca_debug_decode_stamps(None, 0)
t = buf.view(G, nst).cpu().double()
if os.environ.get("STAMP_WG"):  # one workgroup's view (small batches: workgroups without an item skip phases, their stamp
    t = t[int(os.environ["STAMP_WG"]):int(os.environ["STAMP_WG"]) + 1]  # sequences do not line up with the others')
    G = 1
# calibrate ticks -> us with the event-timed launch
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    ops.whisper_decode_token(ps["desc"])
e1.record()
torch.cuda.synchronize()
us_launch = e0.elapsed_time(e1) * 100.0
first = per_layer - (2 if FINE else 1)  # layer 0 has no seam stamp(s) in phase A
last_layer_end = first + per_layer * (L - 1)
span_ticks = float(t[:, last_layer_end - 1].max() - t[:, 0].min())
print(f"{model} B={B}: launch {us_launch:.1f} us (events); layers span {span_ticks:.0f} ticks; status {ps['status'].tolist()}")
tick_us = None
# head + pick ~ the rest; estimate ticks per us from layers' share: report in ticks AND in us assuming the layers take
# (launch - head) ... simply print ticks / 1000 and the ratio to the layer total
lay = t[:, first:last_layer_end].view(G, L - 1, per_layer)
prev_end = torch.cat([t[:, first - 1:first].unsqueeze(1).expand(G, 1, 1), lay[:, :-1, -1:]], 1)  # previous layer's last stamp
lay_full = torch.cat([prev_end, lay], 2)  # [G, L-1, 1 + per_layer]
dt = lay_full[:, :, 1:] - lay_full[:, :, :-1]
med = dt.median(0).values.mean(0)  # median over workgroups, mean over layers
w0 = dt[0].mean(0)
mx = dt.max(0).values.mean(0)
tot = float(med.sum())
print(f"  per layer (median workgroup): {tot:.0f} ticks = {tot / span_ticks * (L - 1) * 100:.0f} % of the layers' span / layer")
k = 0
for ph in ORDER:
    for lab in LAB[ph]:
        print(f"  {ph} {lab:38s} {float(med[k]):8.0f} ticks  ({100 * float(med[k]) / tot:5.1f} %)   slowest workgroup {float(mx[k]):8.0f}")
        k += 1
