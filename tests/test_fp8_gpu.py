"""fp8 (OCP e4m3) forward GEMM and its quantiser (BASELINE.json configs[4]) against a torch restatement:
the quantised bytes are bit-exact (same scale, round to nearest even, saturation), the GEMM matches the product of
the dequantised operands within the bf16 rounding of its output."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    from coral_amd import ops as o

    o.lib()
    return o


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def quantize_ref(x_bf16):
    xf = x_bf16.float()
    am = xf.abs().max()
    scale = torch.tensor(448.0) / am
    q = (xf * scale).clamp(-448.0, 448.0).to(torch.float8_e4m3fn)
    return q, (am / 448.0)


def quantize_dev(ops, x):
    q = torch.zeros(x.shape, dtype=torch.uint8, device=DEV)
    inv = torch.zeros(1, dtype=torch.float32, device=DEV)
    ws = torch.zeros(1, dtype=torch.float32, device=DEV)
    ops.quantize_fp8(x, q, inv, ws)
    return q, inv


@pytest.mark.parametrize("shape,scale", [((300, 256), 1.0), ((64, 1280), 30.0), ((1000, 336), 1e-3)])
def test_quantize_fp8_is_bit_exact(ops, shape, scale):
    x = rnd(*shape, seed=1, scale=scale).to(torch.bfloat16)
    x[0, 0] = 0.0
    q, inv = quantize_dev(ops, x.to(DEV))
    qr, invr = quantize_ref(x)
    torch.cuda.synchronize()
    assert torch.equal(q.cpu(), qr.view(torch.uint8))
    assert abs(inv.item() - invr.item()) <= 1e-7 * abs(invr.item())


@pytest.mark.parametrize("M,N,K", [(128, 128, 128), (300, 264, 256), (1000, 520, 336), (3000, 1280, 1280), (70, 48, 16)])
def test_gemm_fp8_matches_dequantised_product(ops, M, N, K):
    x = rnd(M, K, seed=2, scale=0.7).to(torch.bfloat16)
    w = rnd(N, K, seed=3, scale=0.05).to(torch.bfloat16)
    bias = rnd(N, seed=4)
    xq, sx = quantize_dev(ops, x.to(DEV))
    wq, sw = quantize_dev(ops, w.to(DEV))
    Np = (N + 7) // 8 * 8
    out = torch.zeros(M, Np, dtype=torch.bfloat16, device=DEV)
    ops.gemm_fp8(xq, wq, out, a_scale=sx, b_scale=sw, M=M, N=N, K=K, lda=K, ldb=K, ldc=Np, bias=bias.to(DEV))
    torch.cuda.synchronize()
    xr, sxr = quantize_ref(x)
    wr, swr = quantize_ref(w)
    ref = (xr.float() @ wr.float().t()) * (sxr * swr) + bias
    err = (out[:, :N].float().cpu() - ref).abs().max().item()
    assert err <= 1e-2 * max(1.0, ref.abs().max().item()), err
    # and it is close to the unquantised product (e4m3: 3 mantissa bits)
    full = x.float() @ w.float().t() + bias
    rel = (out[:, :N].float().cpu() - full).norm() / full.norm()
    assert rel < 0.06, rel


def test_gemm_fp8_gelu_epilogue_and_fp32_output(ops):
    M, N, K = 260, 256, 384
    x = rnd(M, K, seed=5, scale=0.5).to(torch.bfloat16)
    w = rnd(N, K, seed=6, scale=0.05).to(torch.bfloat16)
    bias = rnd(N, seed=7, scale=0.1)
    xq, sx = quantize_dev(ops, x.to(DEV))
    wq, sw = quantize_dev(ops, w.to(DEV))
    u = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
    gl = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
    ops.gemm_fp8(xq, wq, u, C2=gl, a_scale=sx, b_scale=sw, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, bias=bias.to(DEV),
                 epilogue=ops.EPI_GELU)
    f32 = torch.zeros(M, N, dtype=torch.float32, device=DEV)
    ops.gemm_fp8(xq, wq, f32, a_scale=sx, b_scale=sw, M=M, N=N, K=K, lda=K, ldb=K, ldc=N)
    torch.cuda.synchronize()
    xr, sxr = quantize_ref(x)
    wr, swr = quantize_ref(w)
    pre = (xr.float() @ wr.float().t()) * (sxr * swr)
    assert (f32.cpu() - pre).abs().max().item() <= 2e-4 * max(1.0, pre.abs().max().item())
    assert (u.float().cpu() - (pre + bias)).abs().max().item() <= 1e-2 * max(1.0, pre.abs().max().item())
    want = torch.nn.functional.gelu(pre + bias)
    assert (gl.float().cpu() - want).abs().max().item() <= 1e-2 * max(1.0, want.abs().max().item())
