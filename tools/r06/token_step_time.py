"""Round 6: the per-token decode step alone (no log-mel, no encoder): the captured launch sequence of
WhisperEngine._token_step replayed from a HIP graph, us per token.
usage: python tools/r06/token_step_time.py [model] [batches...]   (env: CA_* switches as in tools/ENV.md)
Prints one line per batch: us/token, GB/s of algorithmic bytes, fraction of 8 TB/s, and a checksum of the ids decoded
(A/B runs of kernel variants must print the same checksum)."""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from coral_amd import ops  # noqa: E402

model = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].isdigit() else "whisper-medium"
batches = [int(a) for a in sys.argv[1:] if a.isdigit()] or [8, 16]
dev = torch.device("cuda:0")
prefix = [50258, 50285, 50359, 50363]
for B in batches:
    eng, shape, waves, _ = bench.whisper_setup_engine(model, dev, 0, B)
    feats = eng.log_mel(waves)
    enc = eng.encode(feats)
    kv = eng.cross_kv(enc)
    Lmax = 4 + 64
    cache = eng.new_decode_cache(B, Lmax)
    g = eng._graph_state(cache, kv, shape.pad_token_id, shape.eos_token_id)
    sup = torch.zeros(shape.vocab_size, dtype=torch.uint8, device=dev)
    ids0 = torch.tensor([prefix] * B, dtype=torch.int64, device=dev)
    base = eng.decode_step(ids0, kv, cache).contiguous()
    ops.argmax_masked(base, sup, g["nxt"], B, shape.vocab_size, shape.vocab_size)
    g["tok"].copy_(g["nxt"]); g["pos"].fill_(4); g["klen"].fill_(5)
    eng._token_step(cache, g, sup)
    torch.cuda.synchronize()
    n = 40
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        eng._token_step(cache, g, sup)
    graph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        graph.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    byt = bench.whisper_decode_bytes_per_token(eng, shape, B)
    chk = int(g["out"].to(torch.int64).sum())
    print(f"{model} B={B}: {us:8.1f} us/token  {byt / us / 1e3:7.1f} GB/s  frac {byt / (us * 1e-6) / 8e12:.3f}  ids checksum {chk}", flush=True)
    del eng, kv, cache, g, graph
    torch.cuda.empty_cache()
