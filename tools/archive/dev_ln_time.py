"""LayerNorm forward / backward kernel time at the training shapes."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from coral_amd import ops
dev = "cuda:0"
for M, C in ((3992, 1920), (3992, 1024), (12000, 1024), (12000, 1280)):
    x = torch.randn(M, C, device=dev).to(torch.bfloat16)
    dy = torch.randn(M, C, device=dev).to(torch.bfloat16)
    dres = torch.randn(M, C, device=dev).to(torch.bfloat16)
    g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    y = torch.empty_like(x); dx = torch.empty_like(x)
    st = torch.empty(M, 2, device=dev)
    dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    part = torch.empty(ops.layernorm_bwd_partial_floats(M, C), device=dev)
    def t(fn, n=50):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    tf = t(lambda: ops.layernorm_fwd(x, g, b, y, st, M, C))
    tb = t(lambda: ops.layernorm_bwd(dy, x, g, b, st, dres, dx, dg, db, part, M, C))
    tb2 = t(lambda: ops.layernorm_bwd(dy, x, g, b, st, dres, dx, None, None, part, M, C))  # (the engine's form: partials only)
    print(f"M{M} C{C}: fwd {tf:.1f} us, bwd (+reduce) {tb:.1f} us, bwd alone {tb2:.1f} us  [CA_LN_BWD_GRID={__import__('os').environ.get('CA_LN_BWD_GRID', '512')}]")
