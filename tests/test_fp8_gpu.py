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


@pytest.mark.parametrize("force", [1, 3])  # the 128 x 128 and the 256 x 256 kernel
@pytest.mark.parametrize("M,N,K", [(128, 128, 128), (300, 264, 256), (1000, 520, 336), (3000, 1280, 1280), (70, 48, 16)])
def test_gemm_fp8_matches_dequantised_product(ops, M, N, K, force):
    ops.lib().ca_gemm_force_kernel(force)
    try:
        _gemm_fp8_case(ops, M, N, K)
    finally:
        ops.lib().ca_gemm_force_kernel(0)


def _gemm_fp8_case(ops, M, N, K):
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


@pytest.mark.parametrize("M,C", [(300, 1024), (77, 1280), (64, 64)])
def test_layernorm_fwd_fp8_rows(ops, M, C):
    """ca_layernorm_fwd_fp8: the bf16 output is ca_layernorm_fwd's, the bytes are e4m3(y * 448 / amax_row) and the
    row scales amax_row / 448 - bit-exact from the kernel's own y."""
    x = (rnd(M, C, seed=11, scale=2.0) + 0.5).to(torch.bfloat16).to(DEV)
    gamma = (1.0 + 0.2 * rnd(C, seed=12)).to(DEV)
    beta = (0.3 * rnd(C, seed=13)).to(DEV)
    y0 = torch.zeros(M, C, dtype=torch.bfloat16, device=DEV)
    ops.layernorm_fwd(x, gamma, beta, y0, None, M, C, 1e-5)
    y = torch.zeros_like(y0)
    q = torch.zeros(M, C, dtype=torch.uint8, device=DEV)
    rs = torch.zeros(M, dtype=torch.float32, device=DEV)
    ops.layernorm_fwd_fp8(x, gamma, beta, y, q, rs, M, C, 1e-5)
    torch.cuda.synchronize()
    assert torch.equal(y, y0)
    yf = y.float().cpu()
    am = yf.abs().amax(dim=1, keepdim=True)
    qr = (yf * (torch.tensor(448.0) / am)).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).view(torch.uint8)
    assert torch.equal(q.cpu(), qr)
    assert torch.allclose(rs.cpu(), (am / 448.0).squeeze(1), rtol=1e-7, atol=0)


@pytest.mark.parametrize("force", [1, 3])
def test_gemm_fp8_with_row_scales(ops, force):
    ops.lib().ca_gemm_force_kernel(force)
    try:
        _row_scale_case(ops)
    finally:
        ops.lib().ca_gemm_force_kernel(0)


def _row_scale_case(ops):
    M, N, K = 333, 392, 1024
    x = (rnd(M, K, seed=14) * torch.linspace(0.1, 30.0, M)[:, None]).to(torch.bfloat16)  # rows of very different size
    w = rnd(N, K, seed=15, scale=0.05).to(torch.bfloat16)
    g, b = torch.ones(K, device=DEV), torch.zeros(K, device=DEV)
    y = torch.zeros(M, K, dtype=torch.bfloat16, device=DEV)
    q = torch.zeros(M, K, dtype=torch.uint8, device=DEV)
    rs = torch.zeros(M, dtype=torch.float32, device=DEV)
    ops.layernorm_fwd_fp8(x.to(DEV), g, b, y, q, rs, M, K, 1e-5)
    wq, sw = quantize_dev(ops, w.to(DEV))
    out = torch.zeros(M, N, dtype=torch.float32, device=DEV)
    ops.gemm_fp8(q, wq, out, a_row_scale=rs, b_scale=sw, M=M, N=N, K=K, lda=K, ldb=K, ldc=N)
    torch.cuda.synchronize()
    deq = q.cpu().view(torch.float8_e4m3fn).float() * rs.cpu()[:, None]
    wr, swr = quantize_ref(w)
    ref = (deq @ wr.float().t()) * swr
    assert (out.cpu() - ref).abs().max().item() <= 2e-4 * max(1.0, ref.abs().max().item())
    full = y.float().cpu() @ w.float().t()
    assert ((out.cpu() - full).norm() / full.norm()).item() < 0.05


def test_whisper_encoder_with_fp8_weights_tracks_the_bf16_encoder():
    """enable_fp8_encoder(): q|k|v and fc1 of every encoder layer on the fp8 path.  With random weights a dot product
    carries the operands' e4m3 rounding noise (3 mantissa bits, about 3 % RMS each) at full strength, so the states
    differ from the bf16 encoder's by a few per cent (7 % over the 24 layers of whisper-medium); greedy decoding
    still runs and switching the option off restores the bf16 result exactly."""
    from coral_amd.whisper import CORAL_WHISPER_SHAPES, WhisperEngine, WhisperShape

    kw = dict(CORAL_WHISPER_SHAPES["whisper-medium"])
    kw.update(encoder_layers=3, decoder_layers=2)
    eng = WhisperEngine(WhisperShape(**kw), DEV)
    g = torch.Generator(device=DEV).manual_seed(1)
    for n in eng.exported_names():
        v = eng.store.view(n)
        if n.endswith("layer_norm.weight"):
            v.fill_(1.0)
        elif n.endswith(".bias"):
            v.zero_()
        else:
            v.normal_(0.0, 0.02, generator=g)
    eng.refresh_compute_weights()
    feats = torch.randn(2, 80, 3000) * 0.3
    ref = eng.encode(feats).float().clone()
    eng.enable_fp8_encoder()
    out = eng.encode(feats).float()
    rel = ((out - ref).norm() / ref.norm()).item()
    assert 0.0 < rel < 0.10, rel  # (> 0: the fp8 path really ran)
    ids = eng.generate(feats, [50258, 50285, 50359, 50363], 8)
    assert ids[0][:4] == [50258, 50285, 50359, 50363]
    eng.enable_fp8_encoder(False)
    assert torch.equal(eng.encode(feats).float(), ref)


def test_whisper_training_with_fp8_forward_projections():
    """enable_fp8_forward(): loss and gradients stay close to the bf16 step's (the backward is the bf16 one), the e4m3
    weight copies follow the optimiser (refresh_bucket) and the loss on a fixed batch still goes down."""
    import numpy as np

    from coral_amd.trainer import DataParallelTrainer
    from coral_amd.whisper import WhisperShape
    from coral_amd.whisper_train import WhisperTrainEngine
    from oracle import whisper_ref as w

    kw = dict(d_model=64, encoder_layers=2, decoder_layers=2, encoder_attention_heads=4, decoder_attention_heads=4,
              encoder_ffn_dim=128, decoder_ffn_dim=128, num_mel_bins=80, vocab_size=200, max_target_positions=64,
              pad_token_id=150, decoder_start_token_id=151, eos_token_id=150)
    c = w.WhisperConfig(**kw)
    g = torch.Generator().manual_seed(5)
    batch = dict(input_features=torch.randn(2, 80, 3000, generator=g) * 0.5, labels=torch.randint(0, 150, (2, 10), generator=g))
    res = {}
    for mode in ("bf16", "fp8"):
        eng = WhisperTrainEngine(WhisperShape(**kw), DEV)
        eng.load_state_dict(w.synth_params(c))
        if mode == "fp8":
            eng.enable_fp8_forward()
        eng.zero_grad()
        out = eng(**batch)
        eng.backward()
        torch.cuda.synchronize()
        res[mode] = (float(out.loss), eng.store.g32.clone())
    (l0, g0), (l1, g1) = res["bf16"], res["fp8"]
    assert abs(l1 - l0) <= 0.02 * abs(l0), (l0, l1)
    cos = torch.nn.functional.cosine_similarity(g0.flatten(), g1.flatten(), dim=0).item()
    assert cos > 0.98, cos
    assert not torch.equal(g0, g1)  # the fp8 path really ran
    # The gradient entering fc1 joins the fp8 path only once a backward has measured its amax (delayed scaling): turn the
    # scales over as the trainer does after a step (same weights: the e4m3 copies come out the same), run the step again
    # - now with every fp8 piece active - and hold its gradients against the bf16 step's as well
    assert not eng._fp8_train["du_ready"][0]
    eng.refresh_bucket(next(iter(eng.store.buckets)))
    assert eng._fp8_train["du_ready"][0]
    eng.zero_grad()
    out = eng(**batch)
    eng.backward()
    torch.cuda.synchronize()
    g2 = eng.store.g32
    cos2 = torch.nn.functional.cosine_similarity(g0.flatten(), g2.flatten(), dim=0).item()
    print(f"fp8 step vs bf16 step: gradient cosine {cos:.5f} (first step), {cos2:.5f} (all pieces, e4m3 gradient into fc1)")
    assert abs(float(out.loss) - l0) <= 0.02 * abs(l0) and cos2 > 0.97, cos2
    assert not torch.equal(g1, g2)
    eng = WhisperTrainEngine(WhisperShape(**kw), DEV)
    eng.load_state_dict(w.synth_params(c))
    eng.enable_fp8_forward()
    tr = DataParallelTrainer(eng, learning_rate=3e-3, warmup_steps=2, max_steps=30)
    p8_before = eng._fp8_train["p8"].clone()
    losses = [float(tr.train_step([batch])) for _ in range(14)]
    tr.finish()
    torch.cuda.synchronize()
    assert np.isfinite(losses).all() and losses[-1] < 0.7 * losses[1], losses
    f8 = eng._fp8_train
    assert not torch.equal(p8_before, f8["p8"])  # re-quantised after the optimiser steps
    # delayed scaling: the GELU outputs' scales left their first guess (448 / 16) for measured ones, the weights' scales
    # follow the weights, and fc2 ran on the fp8 path (its e4m3 copy exists)
    nw = f8["nw"]
    assert f8["ffn2"] and f8["out8"] and f8["dgrad"] and bool((f8["scale"][nw:] != 28.0).all()) and bool((f8["scale"][:nw] > 1.0).all())
    assert int(f8["p8t"].count_nonzero()) > 0 and int(f8["dy8"].count_nonzero()) > 0  # the fp8 data gradients ran
    assert f8["dgrad_fc1"] and f8["du_ready"][0] and int(f8["du8"].count_nonzero()) > 0  # ... fc1's too, after the first step
    for name, n in (("fc2", 64 * 128), ("self_attn.out_proj", 64 * 64)):
        off = eng.store.off(f"model.encoder.layers.1.{name}.weight")
        assert int(f8["p8"][off:off + n].count_nonzero()) > 0


def test_delayed_quantiser_and_amax_rotation(ops):
    """ca_quantize_fp8_delayed: ONE pass with last step's scale, bit-exact against torch's e4m3 cast of x * scale
    (saturating), this step's amax left in the amax word; ca_fp8_amax_rotate turns it into scale / inv_scale with the
    margin and clears the word; a word that was not written keeps its old scale."""
    x = rnd(777, 264, seed=3, scale=0.3).to(torch.bfloat16)
    xd = x.to(DEV)
    q = torch.zeros(x.shape, dtype=torch.uint8, device=DEV)
    S = ops.FP8_AMAX_SLOTS
    amax = torch.zeros(2 * S, dtype=torch.int32, device=DEV)  # an accumulator = S words, the amax their maximum
    scale = torch.tensor([100.0, 7.0], device=DEV)   # too large on purpose: values beyond 448 / 100 saturate
    inv = torch.tensor([0.01, 1.0 / 7.0], device=DEV)
    ops.quantize_fp8_delayed(xd, q, scale[0:1], amax)
    want = (x.float() * 100.0).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).view(torch.uint8)
    assert torch.equal(q.cpu(), want)
    am = float(x.float().abs().max())
    assert amax[:S].view(torch.float32).max().item() == am and int(amax[S:].abs().sum()) == 0
    ops.fp8_amax_rotate(amax, scale, inv, 2, margin=2.0)
    torch.cuda.synchronize()
    assert abs(scale[0].item() - 448.0 / (2 * am)) <= 1e-6 * 448.0 / am and abs(inv[0].item() - 2 * am / 448.0) <= 1e-7
    assert scale[1].item() == 7.0 and int(amax.abs().sum()) == 0  # untouched accumulator: old scale kept; used one cleared
    ops.quantize_fp8_delayed(xd, q, scale[0:1], amax)  # second pass, now inside the range
    s = scale[0].item()
    assert torch.equal(q.cpu(), (x.float() * s).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).view(torch.uint8))


@pytest.mark.parametrize("force", [1, 3])
@pytest.mark.parametrize("fp8_in", [False, True])
@pytest.mark.parametrize("grad", [False, True])
def test_gelu_epilogue_third_output_in_fp8(ops, force, fp8_in, grad):
    """CaGemmDesc.C8: the GELU output of a projection (grad = False, CA_EPI_GELU) or the gradient v * GELU'(R) of a data
    gradient (grad = True, CA_EPI_DGELU) also as e4m3 with a given per-tensor scale, from the tile that computes it (bf16
    and fp8 input GEMM, 128 x 128 and 256 x 256 kernels, interior and ragged tiles, with dropout): the bytes are the e4m3
    cast of the kernel's own fp32 value times the scale - checked against the bf16 output to one e4m3 step - and the amax
    accumulator holds max |value|."""
    from coral_amd.ops import EPI_DGELU, EPI_GELU

    M, N, K = 1000, 520, 256
    a = rnd(M, K, seed=1, scale=0.5).to(torch.bfloat16)
    w = rnd(N, K, seed=2, scale=0.2).to(torch.bfloat16)
    bias = rnd(N, seed=3, scale=0.1)
    u = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
    g = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV)
    g8 = torch.zeros(M, N, dtype=torch.uint8, device=DEV)
    sc = torch.tensor([37.0], device=DEV)
    amax = torch.zeros(ops.FP8_AMAX_SLOTS, dtype=torch.int32, device=DEV)
    kw = dict(M=M, N=N, K=K, lda=K, ldb=K, ldc=N, dropout_p=0.1, dropout_seed=11, C8=g8, c8_scale=sc, c8_amax=amax)
    if grad:
        R = rnd(M, N, seed=5, scale=1.5).to(torch.bfloat16).to(DEV)
        kw.update(epilogue=EPI_DGELU, R=R, ldr=N)
        out = u  # the gradient is the first output
    else:
        kw.update(C2=g, bias=bias.to(DEV), epilogue=EPI_GELU)
        out = g
    ops.lib().ca_gemm_force_kernel(force)
    try:
        if fp8_in:
            qa, ia = quantize_dev(ops, a.to(DEV))
            qw, iw = quantize_dev(ops, w.to(DEV))
            ops.gemm_fp8(qa, qw, u, a_scale=ia, b_scale=iw, **kw)
        else:
            ops.gemm(a.to(DEV), w.to(DEV), u, **kw)
    finally:
        ops.lib().ca_gemm_force_kernel(0)
    torch.cuda.synchronize()
    gf = out.float().cpu()
    got = g8.cpu().view(torch.float8_e4m3fn).float() / 37.0
    # e4m3 carries 3 mantissa bits: within one step (2^-3 relative, 2^-9 absolute after the scale) of the bf16 output
    tol = gf.abs() * 0.0625 + gf.abs() * 0.008 + 2.0 ** -9 / 37.0 * 2
    assert bool(((got - gf).abs() <= tol).all()), float(((got - gf).abs() - tol).max())
    assert float((got == 0).float().mean()) < 0.2  # dropout zeros + tiny values only
    am = amax.view(torch.float32).max().item()
    assert abs(am - float(gf.abs().max())) <= 0.01 * am  # (fp32 value vs its bf16 rounding)


@pytest.mark.parametrize("hd,H,T", [(64, 4, 333), (16, 4, 1500), (120, 2, 260)])
def test_attention_output_also_in_fp8(ops, hd, H, T):
    """CaAttnDesc.O8: the forward's output stage also writes the context as e4m3 of the bf16 output times a given scale -
    bit-exact against torch's e4m3 cast of the kernel's own bf16 output (saturating), max |O| in the amax accumulator,
    the bf16 output itself unchanged by the option."""
    B, d = 2, H * hd
    qkv = rnd(B, T, 3 * d, seed=4, scale=1.0).to(torch.bfloat16).to(DEV)
    Tqp = (T + 31) // 32 * 32
    lse = torch.zeros(B, H, Tqp, device=DEV)
    kw = dict(B=B, H=H, Tq=T, Tk=T, hd=hd, Tqp=Tqp, scale=hd ** -0.5, ldo=d, sob=T * d, ldq=3 * d, ldk=3 * d, ldv=3 * d,
              sqb=T * 3 * d, skb=T * 3 * d, svb=T * 3 * d, q_off=0, k_off=d, v_off=2 * d)
    O0 = torch.zeros(B, T, d, dtype=torch.bfloat16, device=DEV)
    ops.attn_fwd(qkv, qkv, qkv, O0, lse, **kw)
    O1 = torch.zeros_like(O0)
    O8 = torch.zeros(B, T, d, dtype=torch.uint8, device=DEV)
    sc = torch.tensor([300.0], device=DEV)  # (large on purpose: the biggest outputs saturate at 448)
    amax = torch.zeros(ops.FP8_AMAX_SLOTS, dtype=torch.int32, device=DEV)
    ops.attn_fwd(qkv, qkv, qkv, O1, lse, O8=O8, o8_scale=sc, o8_amax=amax, **kw)
    torch.cuda.synchronize()
    assert torch.equal(O0, O1)
    want = (O1.float().cpu() * 300.0).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).view(torch.uint8)
    assert torch.equal(O8.cpu(), want)
    assert amax.view(torch.float32).max().item() == float(O1.float().abs().max())


@pytest.mark.parametrize("rows,C,p", [(333, 1280, 0.1), (1000, 64, 0.0), (77, 4096, 0.25)])
def test_row_quantising_dropout_and_transposed_weight_copy(ops, rows, C, p):
    """The two producers of the fp8 data gradients.  ca_dropout_rows_fp8: the bf16 output is ca_dropout_bf16's bit for
    bit (the weight gradient reads it), the e4m3 rows and row scales are the per-row quantisation of exactly that
    output.  ca_quantize_fp8_transposed: the transposed e4m3 copy of a matrix with a given scale, bit-exact."""
    x = rnd(rows, C, seed=8, scale=3.0).to(torch.bfloat16)
    x[5] = 0.0  # an all-zero row: scale 1
    xd = x.to(DEV)
    y0 = torch.zeros_like(xd)
    if p > 0:
        ops.dropout(xd, y0, rows * C, p, 1234)
    else:
        y0.copy_(xd)
    y1 = torch.zeros_like(xd)
    q = torch.zeros(rows, C, dtype=torch.uint8, device=DEV)
    rs = torch.zeros(rows, dtype=torch.float32, device=DEV)
    ops.dropout_rows_fp8(xd, y1 if p > 0 else None, q, rs, rows, C, p, 1234)
    torch.cuda.synchronize()
    if p > 0:
        assert torch.equal(y0, y1)
        assert 0.5 * p < float((y1 == 0).float().mean()) < 1.5 * p + 0.01
    yf = y0.float().cpu()
    am = yf.abs().amax(dim=1)
    want_rs = torch.where(am > 0, am / 448.0, torch.ones_like(am))
    assert torch.allclose(rs.cpu(), want_rs, rtol=1e-6, atol=0)
    sc = torch.where(am > 0, 448.0 / am, torch.ones_like(am))
    want = (yf * sc[:, None]).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).view(torch.uint8)
    # (the device forms 448 / amax with its fast division: the scale may differ in the last bit, which moves a value that
    # sits on an e4m3 rounding boundary to the neighbouring code - rare, and never further than one code)
    qa, qb = q.cpu().view(torch.float8_e4m3fn).float(), want.view(torch.float8_e4m3fn).float()
    assert float((qa != qb).float().mean()) <= 5e-3
    assert bool(((qa - qb).abs() <= 0.13 * qb.abs() + 2.0 ** -9).all())
    # transposed copy (rows x C matrix -> C x rows)
    if rows % 8 == 0:
        qt = torch.zeros(C, rows, dtype=torch.uint8, device=DEV)
        s1 = torch.tensor([41.0], device=DEV)
        ops.quantize_fp8_transposed(xd, rows, C, qt, s1)
        want_t = (x.float() * 41.0).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).view(torch.uint8).t().contiguous()
        assert torch.equal(qt.cpu(), want_t)


def test_group_refresh_equals_the_single_matrix_kernels(ops):
    """ca_fp8_refresh_group (one launch for a layer's weight copies) against ca_quantize_fp8_delayed and
    ca_quantize_fp8_transposed per matrix: the same e4m3 bytes, the same amax after the rotation - on ragged tile edges,
    with and without a transposed copy, more tasks than one launch takes."""
    shapes = [(1280, 1280, True), (200, 72, True), (333, 1280, False), (64, 64, True), (8, 8, True), (3840, 1280, False),
              (72, 200, True), (1000, 64, True), (5120, 1280, True), (16, 4096, True)]
    S = ops.FP8_AMAX_SLOTS
    n = len(shapes)
    xs = [(rnd(r, c, seed=20 + i, scale=0.5 + i).to(torch.bfloat16)).to(DEV) for i, (r, c, _) in enumerate(shapes)]
    scale = torch.tensor([448.0 / (3.0 * (0.5 + i)) for i in range(n)], device=DEV)  # (some values saturate)
    amax_a = torch.zeros(n * S, dtype=torch.int32, device=DEV)
    amax_b = torch.zeros(n * S, dtype=torch.int32, device=DEV)
    qa = [torch.zeros(r, c, dtype=torch.uint8, device=DEV) for r, c, _ in shapes]
    qb = [torch.zeros(r, c, dtype=torch.uint8, device=DEV) for r, c, _ in shapes]
    ta = [torch.zeros(c, r, dtype=torch.uint8, device=DEV) if t else None for r, c, t in shapes]
    tb = [torch.zeros(c, r, dtype=torch.uint8, device=DEV) if t else None for r, c, t in shapes]
    tasks = []
    for i, (r, c, t) in enumerate(shapes):
        ops.quantize_fp8_delayed(xs[i], qa[i], scale[i:i + 1], amax_a[i * S:])
        if t:
            ops.quantize_fp8_transposed(xs[i], r, c, ta[i], scale[i:i + 1])
        tasks.append((xs[i], 0, r, c, qb[i], 0, tb[i], 0, scale[i:i + 1], amax_b[i * S:]))
    ops.fp8_refresh_group(tasks)
    torch.cuda.synchronize()
    for i, (r, c, t) in enumerate(shapes):
        assert torch.equal(qa[i], qb[i]), (i, r, c)
        if t:
            assert torch.equal(ta[i], tb[i]), (i, r, c)
        want = (xs[i].float() * scale[i]).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).view(torch.uint8)
        assert torch.equal(qb[i], want)
    ma = amax_a.view(n, S).amax(dim=1)
    mb = amax_b.view(n, S).amax(dim=1)  # (non-negative floats order like their bit patterns)
    assert torch.equal(ma, mb)
    assert torch.equal(mb.view(torch.float32).cpu(), torch.stack([x.float().abs().amax() for x in xs]).cpu())


def test_gemm_fp8_random_shapes(ops):
    rng = torch.Generator().manual_seed(99)
    for _ in range(8):
        M = int(torch.randint(1, 600, (1,), generator=rng))
        N = int(torch.randint(1, 60, (1,), generator=rng)) * 8
        K = int(torch.randint(1, 40, (1,), generator=rng)) * 16
        for force in (1, 3):
            ops.lib().ca_gemm_force_kernel(force)
            try:
                _gemm_fp8_case(ops, M, N, K)
            finally:
                ops.lib().ca_gemm_force_kernel(0)
