"""Development check of ca_gemm_bf16 on a GPU box (all four operand layouts, tails, batch,
epilogues) against torch fp32 matmul.  Run: python tools/dev_gemm_check.py"""
import ctypes as C
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from coral_amd._lib import CaGemmDesc, LIB_PATH  # noqa: E402

lib = C.CDLL(str(LIB_PATH))
lib.ca_gemm_bf16.restype = C.c_int
lib.ca_gemm_bf16.argtypes = [C.POINTER(CaGemmDesc), C.c_void_p]
lib.ca_last_error.restype = C.c_char_p
dev = torch.device("cuda:0")


def run(M, N, K, al, bl, batch=1, epi=0, out_f32=0, bias=False, alpha=1.0, acc=False):
    g = torch.Generator(device="cpu").manual_seed(M * 7 + N * 3 + K + al * 2 + bl)
    A = (torch.randn(batch, M, K, generator=g) * 0.5).to(torch.bfloat16)
    B = (torch.randn(batch, N, K, generator=g) * 0.5).to(torch.bfloat16)
    ref = torch.matmul(A.float(), B.float().transpose(1, 2)) * alpha
    Ad = (A if al == 0 else A.transpose(1, 2).contiguous()).to(dev)
    Bd = (B if bl == 0 else B.transpose(1, 2).contiguous()).to(dev)
    d = CaGemmDesc()
    d.A, d.B = Ad.data_ptr(), Bd.data_ptr()
    d.M, d.N, d.K = M, N, K
    d.a_layout, d.b_layout = al, bl
    d.lda = K if al == 0 else M
    d.ldb = K if bl == 0 else N
    d.ldc = N
    d.ldr = N
    d.batch1, d.batch2 = 1, batch
    d.sA2, d.sB2, d.sC2, d.sR2 = M * K, N * K, M * N, M * N
    d.epilogue, d.out_f32, d.alpha = epi, out_f32, alpha
    d.accumulate = 1 if acc else 0
    bias_t = None
    if bias:
        bias_t = torch.randn(N, generator=g).to(dev)
        d.bias = bias_t.data_ptr()
        ref = ref + bias_t.cpu()
    Rt = None
    if epi in (2, 3):
        Rt = torch.randn(batch, M, N, generator=g).to(torch.bfloat16).to(dev)
        d.R = Rt.data_ptr()
    Cd = torch.zeros(batch, M, N, dtype=torch.float32 if out_f32 else torch.bfloat16, device=dev)
    if acc:
        Cd.fill_(1.0)
        ref = ref + 1.0
    C2 = torch.zeros(batch, M, N, dtype=torch.bfloat16, device=dev)
    d.C, d.C2 = Cd.data_ptr(), C2.data_ptr()
    if epi == 2:
        ref = ref + Rt.float().cpu()
    if epi == 3:
        u = Rt.float().cpu()
        cdf = 0.5 * (1 + torch.erf(u / 2**0.5))
        pdf = torch.exp(-0.5 * u * u) / (2 * 3.141592653589793) ** 0.5
        ref = ref * (cdf + u * pdf)
    rc = lib.ca_gemm_bf16(C.byref(d), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert rc == 0, lib.ca_last_error()
    out = Cd.float().cpu()
    tol = 2e-2 * (K**0.5) * 0.25 * abs(alpha) + (0.05 if not out_f32 else 0.0)
    err = (out - ref).abs().max().item()
    ok = err < tol
    if epi == 1:
        gref = torch.nn.functional.gelu(ref)
        e2 = (C2.float().cpu() - gref).abs().max().item()
        ok = ok and e2 < tol
    print(f"M{M} N{N} K{K} al{al} bl{bl} b{batch} epi{epi} f32{out_f32} err {err:.4g} tol {tol:.3g} {'OK' if ok else 'FAIL'}")
    return ok


def perf(M, N, K, al, bl, iters=20):
    A = torch.randn(M, K, device=dev).to(torch.bfloat16)
    B = torch.randn(N, K, device=dev).to(torch.bfloat16)
    if al:
        A = A.t().contiguous()
    if bl:
        B = B.t().contiguous()
    Cd = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    d = CaGemmDesc()
    d.A, d.B, d.C = A.data_ptr(), B.data_ptr(), Cd.data_ptr()
    d.M, d.N, d.K = M, N, K
    d.a_layout, d.b_layout = al, bl
    d.lda = K if al == 0 else M
    d.ldb = K if bl == 0 else N
    d.ldc = N
    d.batch1 = d.batch2 = 1
    d.alpha = 1.0
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        lib.ca_gemm_bf16(C.byref(d), st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        lib.ca_gemm_bf16(C.byref(d), st)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"perf M{M} N{N} K{K} al{al} bl{bl}: {ms*1e3:.1f} us  {2*M*N*K/ms/1e9:.1f} TFLOP/s")
    t0 = time.time()
    Af = A if not al else A.t()
    Bf = B if not bl else B.t()
    for _ in range(3):
        torch.matmul(Af, Bf.t())
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        torch.matmul(Af, Bf.t())
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"   (hipBLASLt via torch for scale: {ms*1e3:.1f} us  {2*M*N*K/ms/1e9:.1f} TFLOP/s)")


if __name__ == "__main__":
    allok = True
    for al in (0, 1):
        for bl in (0, 1):
            allok &= run(128, 128, 64, al, bl)
            allok &= run(256, 384, 512, al, bl)
            allok &= run(1000, 120, 504, al, bl, batch=3)
            allok &= run(504, 1920, 4000 if al else 120, al, bl, out_f32=1, alpha=0.25)
    allok &= run(500, 1024, 256, 0, 0, epi=1, bias=True)
    allok &= run(500, 1024, 256, 0, 0, epi=2, bias=True)
    allok &= run(500, 1024, 256, 0, 1, epi=3)
    allok &= run(500, 46, 256, 0, 0, out_f32=1)
    allok &= run(504, 48, 256, 1, 1, out_f32=1, acc=True)
    print("ALL OK" if allok else "SOME FAILED")
    for al, bl in ((0, 0), (0, 1), (1, 1)):
        perf(3992, 7680, 1920, al, bl)
        perf(4096, 4096, 4096, al, bl)
    sys.exit(0 if allok else 1)
