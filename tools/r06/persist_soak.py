"""Round 6: soak of the persistent decode launch - many whole generations (fresh synthetic clips each round) through
`WhisperEngine.generate` with one launch per token, each compared with the launch sequence's ids; counts launches, give-ups
(status word) and differing generations.
usage: python tools/r06/persist_soak.py [model] [rounds] [batches...]    (CA_DECODE_STRICT=1 is set: a give-up raises)"""
import os
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
os.environ["CA_DECODE_STRICT"] = "1"
import bench  # noqa: E402

args = sys.argv[1:]
model = args[0] if args and not args[0].isdigit() else "whisper-medium"
nums = [int(a) for a in args if a.isdigit()]
rounds = nums[0] if nums else 10
batches = nums[1:] or [16, 8, 1]
dev = torch.device("cuda:0")
prefix = [50258, 50285, 50359, 50363]
max_length = int(os.environ.get("MAXLEN", "225"))
launches = diffs = gens = 0
t0 = time.time()
for B in batches:
    eng, shape, waves, _ = bench.whisper_setup_engine(model, dev, 0, B)
    g = torch.Generator(device="cpu").manual_seed(1234 + B)
    for r in range(rounds):
        w = torch.zeros(B, 480_000)
        for b in range(B):  # clips of 3 .. 29 s, padded to 30 s
            n = 16_000 * (3 + (r + b) % 27)
            w[b, :n] = (0.02 + 0.02 * torch.rand(1, generator=g).item()) * torch.randn(n, generator=g)
        feats = eng.log_mel(w)
        sup = None
        if os.environ.get("EOS_MODE") == "1":
            # everything suppressed but 24 tokens and eos: clips finish at different, random steps - the finished-row
            # bookkeeping, the host's late all-finished check and the trimming get exercised
            allowed = set(torch.randint(0, 50000, (24,), generator=g).tolist()) | {shape.eos_token_id}
            sup = [t for t in range(shape.vocab_size) if t not in allowed]
        os.environ["CA_DECODE_PERSISTENT"] = "0"
        ref = eng.generate(feats, prefix, max_length, suppress_tokens=sup)
        os.environ["CA_DECODE_PERSISTENT"] = "1"
        got = eng.generate(feats, prefix, max_length, suppress_tokens=sup)
        if os.environ.get("EOS_MODE") == "1" and r < 3:
            print(f"  round {r}: lengths {sorted(len(x) for x in got)[:4]} .. {max(len(x) for x in got)}, "
                  f"finished rows {sum(shape.eos_token_id in x for x in got)} of {B}", flush=True)
        gens += 1
        launches += max(len(x) for x in got) - len(prefix) - 1
        if got != ref:
            diffs += 1
            print(f"{model} B={B} round {r}: ids DIFFER", flush=True)
    print(f"{model} B={B}: {rounds} generations to {max_length} done, {time.time() - t0:.0f} s", flush=True)
    del eng
    torch.cuda.empty_cache()
print(f"soak {model}: {gens} generations, ~{launches} persistent launches, 0 give-ups (strict mode would have raised), "
      f"{diffs} generations differing from the launch sequence")
sys.exit(1 if diffs else 0)
