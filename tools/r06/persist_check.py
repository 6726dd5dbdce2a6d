"""Round 6: ca_whisper_decode_token (one persistent launch per token) against the launch sequence it replaces, from the
same decode state: logits, picked tokens, the K|V cache rows written and the bookkeeping must be bit-identical.
usage: python tools/r06/persist_check.py [model] [batches...]   env NTOK (tokens compared, default 6)"""
import os
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from coral_amd import ops  # noqa: E402

model = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].isdigit() else "whisper-medium"
batches = [int(a) for a in sys.argv[1:] if a.isdigit()] or [8, 16]
ntok = int(os.environ.get("NTOK", "6"))
dev = torch.device("cuda:0")
prefix = [50258, 50285, 50359, 50363]
bad = 0
for B in batches:
    eng, shape, waves, _ = bench.whisper_setup_engine(model, dev, 0, B)
    enc = eng.encode(eng.log_mel(waves))
    kv = eng.cross_kv(enc)
    Lmax = 4 + 40
    sup = torch.zeros(shape.vocab_size, dtype=torch.uint8, device=dev)
    sup[torch.randint(0, shape.vocab_size, (500,), device=dev)] = 1

    def fresh():
        cache = eng.new_decode_cache(B, Lmax)
        g = eng._graph_state(cache, kv, shape.pad_token_id, shape.eos_token_id)
        ids0 = torch.tensor([prefix] * B, dtype=torch.int64, device=dev)
        base = eng.decode_step(ids0, kv, cache).contiguous()
        ops.argmax_masked(base, sup, g["nxt"], B, shape.vocab_size, shape.vocab_size)
        g["tok"].copy_(g["nxt"]); g["pos"].fill_(4); g["klen"].fill_(5)
        return cache, g

    ca, ga = fresh()
    cb, gb = fresh()
    ps = eng._persistent_state(cb, gb, sup)
    if ps is None:
        print(f"{model} B={B}: persistent step not supported here"); bad += 1; continue
    for t in range(ntok):
        eng._token_step_launches(ca, ga, sup)
        ops.whisper_decode_token(ps["desc"])
        torch.cuda.synchronize()
        st = ps["status"].tolist()
        V = shape.vocab_size
        la, lb = ga["logits"][:, :V], gb["logits"][:, :V]
        same_logits = bool(torch.equal(la.view(torch.int32), lb.view(torch.int32)))
        nd = int((la != lb).sum())
        md = float((la - lb).abs().max())
        same_tok = bool(torch.equal(ga["nxt"], gb["nxt"])) and bool(torch.equal(ga["tok"], gb["tok"]))
        same_book = all(bool(torch.equal(ga[k], gb[k])) for k in ("pos", "klen", "done", "out"))
        same_cache = all(bool(torch.equal(x, y)) for x, y in zip(ca["kv"], cb["kv"]))
        ok = st[0] == 0 and same_logits and same_tok and same_book and same_cache
        bad += 0 if ok else 1
        print(f"{model} B={B} token {t}: status {st} logits {'same' if same_logits else f'DIFFER ({nd} values, max {md:.3e})'} "
              f"tokens {'same' if same_tok else 'DIFFER'} bookkeeping {'same' if same_book else 'DIFFER'} "
              f"cache {'same' if same_cache else 'DIFFER'}", flush=True)
        if not ok:
            rows = (la != lb).any(1).nonzero().flatten().tolist()
            print("   rows that differ:", rows, "tok(launches)", ga["tok"].tolist(), "tok(persistent)", gb["tok"].tolist(),
                  "done", ga["done"].tolist(), "pos", ga["pos"].tolist()[:4])
            for li, (x, y) in enumerate(zip(ca["kv"], cb["kv"])):
                if not torch.equal(x, y):
                    dd = (x.view(B, Lmax, -1) != y.view(B, Lmax, -1))
                    print("   first cache layer that differs:", li, "clips", dd.any(2).any(1).nonzero().flatten().tolist(),
                          "positions", dd.any(2).any(0).nonzero().flatten().tolist(), "cols(min,max)",
                          int(dd.any(0).any(0).nonzero().min()), int(dd.any(0).any(0).nonzero().max()))
                    break
        if st[0] != 0:
            break
    del eng, kv, ca, cb, ga, gb, ps
    torch.cuda.empty_cache()
print("PERSIST_CHECK", "OK" if bad == 0 else f"FAILED ({bad})")
sys.exit(0 if bad == 0 else 1)
