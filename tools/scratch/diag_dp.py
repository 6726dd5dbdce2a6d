import sys, tempfile
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import test_dp_gpu as T
import dp_worker
tmp = Path(tempfile.mkdtemp())
rep = T._run_two_ranks(tmp / "rep", "bf16", steps=3)
zer = T._run_two_ranks(tmp / "zero", "bf16", steps=3, zero=2)
print("norms rep", rep[0]["norms"], "zer", zer[0]["norms"])
print("losses rep", rep[0]["losses"], "zer", zer[0]["losses"])
d = (zer[0]["p32"] - rep[0]["p32"]).abs()
eng, _ = dp_worker.build_case()
idx = eng.store.index
top = torch.topk(d, 12)
for v, i in zip(top.values.tolist(), top.indices.tolist()):
    name = [n for n, (off, shp) in idx.items() if off <= i < off + int(torch.tensor(shp).prod())]
    print(f"{v:.3e} @ {i} {name} rep p {rep[0]['p32'][i]:.6e} zer p {zer[0]['p32'][i]:.6e} m {rep[0]['m'][i]:.3e}/{zer[0]['m'][i]:.3e} v {rep[0]['v'][i]:.3e}/{zer[0]['v'][i]:.3e}")
print("count > 1e-3:", int((d > 1e-3).sum()), "of", d.numel(), "; > 1e-4:", int((d > 1e-4).sum()))
