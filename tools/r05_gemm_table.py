"""Forced-kernel table of the GEMM launch shapes of the three training workloads (round-5 review item 1c):
   python tools/r05_gemm_table.py [workload ...] [--iters N] [--kernels 0,1,2,3,5]
Every launch shape of a transformer layer with the epilogue it carries in the step, stand-alone (20 back-to-back
launches, HIP events), once per kernel: 0 = the automatic choice, 1 = S (128x128, 2 workgroups / CU), 2 = L (256x128,
3-stage ring), 3 = X (256x256, persistent), 5 = M (128x128 on 8 waves).  (6 was kernel P - kernel L's tile in persistent
workgroups with a tile's outputs stored under the next tile's main loop: profiles/r05_gemm_kernel_p.txt, removed.)
Output: one line per (shape, kernel) with tiles / rounds, us and TFLOP/s."""
import argparse
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from coral_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("workloads", nargs="*", default=["xlsr2b_b8", "xlsr300m_b64", "whisper_large_b8"])
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--kernels", default="0,1,2,3,5")
ap.add_argument("--only", default="", help="substring filter on the shape name")
args = ap.parse_args()
dev = "cuda:0"
kernels = [int(k) for k in args.kernels.split(",")]
KNAME = {0: "auto", 1: "S", 2: "L", 3: "X", 5: "M", 6: "P"}
TILE = {1: (128, 128, 512), 2: (256, 128, 256), 3: (256, 256, 256), 5: (128, 128, 256), 6: (256, 128, 256)}


def layer_shapes(M, d, f, qkv_fused=True):
    """(name, M, N, K, a_layout, b_layout, epilogue kind) of one transformer layer's launches."""
    s = []
    if qkv_fused:
        s.append(("q|k|v fwd +bias", M, 3 * d, d, 0, 0, "bias"))
    else:
        s.append(("q (or k, v) fwd +bias", M, d, d, 0, 0, "bias"))
    s += [("out-proj fwd +bias+residual", M, d, d, 0, 0, "res"),
          ("fc1 fwd +bias+GELU+dropout, 2 outputs", M, f, d, 0, 0, "gelu"),
          ("fc2 fwd +bias+residual", M, d, f, 0, 0, "res"),
          ("fc2 dgrad +GELU'+dropout", M, f, d, 0, 1, "dgelu"),
          ("fc1 dgrad", M, d, f, 0, 1, "plain"),
          ("out-proj dgrad", M, d, d, 0, 1, "plain"),
          ("q|k|v dgrad", M, d, 3 * d, 0, 1, "plain") if qkv_fused else ("q dgrad", M, d, d, 0, 1, "plain"),
          ("fc1 wgrad fp32", f, d, M, 1, 1, "wgrad"),
          ("fc2 wgrad fp32", d, f, M, 1, 1, "wgrad"),
          ("q|k|v wgrad fp32", 3 * d, d, M, 1, 1, "wgrad") if qkv_fused else ("q wgrad fp32", d, d, M, 1, 1, "wgrad"),
          ("out-proj wgrad fp32", d, d, M, 1, 1, "wgrad")]
    return s


WORKLOADS = {
    "xlsr2b_b8": ("XLS-R-2B (wav2vec2-large), 8 x 10 s: M = 3992, d = 1920, ffn = 7680", layer_shapes(3992, 1920, 7680)),
    "xlsr300m_b64": ("XLS-R-300M (wav2vec2-small), 64 x 10 s: M = 31936, d = 1024, ffn = 4096", layer_shapes(31936, 1024, 4096)),
    "whisper_large_b8": ("whisper-large encoder, 8 x 30 s: M = 12000, d = 1280, ffn = 5120", layer_shapes(12000, 1280, 5120)),
}


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def build(M, N, K, al, bl, kind):
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    B = (0.05 * torch.randn(N, K, device=dev)).to(torch.bfloat16)
    if al:
        A = A.t().contiguous()
    if bl:
        B = B.t().contiguous()
    kw = dict(M=M, N=N, K=K, a_layout=al, b_layout=bl, lda=(M if al else K), ldb=(N if bl else K), ldc=N)
    if kind == "wgrad":
        G = torch.zeros(M * N, dtype=torch.float32, device=dev)
        slots = torch.zeros(ops.sumsq_slots(M, N), dtype=torch.float32, device=dev)
        return lambda: ops.gemm(A, B, G, out_f32=True, accumulate=False, stream_out=True, c_sumsq=slots, c_sumsq_off=0, **kw)
    Cd = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    if kind == "plain":
        return lambda: ops.gemm(A, B, Cd, **kw)
    bias = torch.randn(N, device=dev)
    if kind == "bias":
        return lambda: ops.gemm(A, B, Cd, bias=bias, **kw)
    R = torch.randn(M, N, device=dev).to(torch.bfloat16)
    if kind == "res":
        return lambda: ops.gemm(A, B, Cd, bias=bias, R=R, ldr=N, epilogue=ops.EPI_RESIDUAL, **kw)
    if kind == "gelu":
        C2 = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        return lambda: ops.gemm(A, B, Cd, bias=bias, C2=C2, c2_off=0, epilogue=ops.EPI_GELU, dropout_p=0.1, dropout_seed=7,
                                stream_out=True, **kw)
    if kind == "dgelu":
        return lambda: ops.gemm(A, B, Cd, R=R, ldr=N, epilogue=ops.EPI_DGELU, dropout_p=0.1, dropout_seed=7, **kw)
    raise ValueError(kind)


print(f"# {torch.cuda.get_device_name(0)}; random bf16 operands; {args.iters} back-to-back launches per figure, HIP events")
for wl in args.workloads:
    title, shapes = WORKLOADS[wl]
    print(f"\n## {title}")
    print(f"{'launch':42s} {'M':>6s} {'N':>6s} {'K':>6s} form  " + "  ".join(f"{KNAME[k]:>5s} us / TF/s (tiles, rounds)" for k in kernels))
    for name, M, N, K, al, bl, kind in shapes:
        if args.only and args.only not in name:
            continue
        fn = build(M, N, K, al, bl, kind)
        cells = []
        for k in kernels:
            ops.lib().ca_gemm_force_kernel(k)
            try:
                us = timeit(fn, args.iters)
                tf = 2.0 * M * N * K / us / 1e6
                if k in TILE:
                    tm, tn, slots = TILE[k]
                    t = -(-M // tm) * -(-N // tn)
                    cells.append(f"{us:7.1f} / {tf:6.1f} ({t:4d}, {t / slots:4.2f})")
                else:
                    cells.append(f"{us:7.1f} / {tf:6.1f}" + " " * 13)
            except Exception as e:  # a kernel that does not take the shape
                cells.append(f"{'-':>7s} ({str(e)[:16]})")
            ops.lib().ca_gemm_force_kernel(0)
        form = "NT NN TN TT".split()[al * 2 + bl]
        print(f"{name:42s} {M:6d} {N:6d} {K:6d} {form}  " + "  ".join(cells), flush=True)
        del fn
        torch.cuda.empty_cache()
