"""Pre-LN transformer sub-blocks (forward with saved activations + hand-written backward) as sequences
of libcoral_amd kernels.  Used by the Whisper training path; the three block kinds are exactly the
ones of `WhisperEncoderLayer` / `WhisperDecoderLayer` ($TF/models/whisper/modeling_whisper.py:379-413,
448-505): residual self-attention, residual cross-attention, residual GELU feed-forward.

Every block works on flat row-major bf16 activations [B*T, d] and a `ParamStore` (fp32 masters `p32`,
bf16 compute copies `p16`, fp32 gradients `g32`); weight gradients are accumulated in fp32.
"""

from __future__ import annotations

import torch

from . import ops
from .ops import EPI_DGELU, EPI_GELU, EPI_RESIDUAL, MNMAJOR


def _z(n, dev, dt=torch.bfloat16):
    return torch.zeros(n, dtype=dt, device=dev)


class Scratch:
    """Shared backward scratch for one (rows, d, f) problem size."""

    def __init__(self, M, d, f, dev, Mkv=0):
        self.dx = _z(M * d, dev)
        self.dctx = _z(M * d, dev)
        self.dqkv = _z(M * 3 * d, dev)
        self.du = _z(M * f, dev)
        self.dkv = _z(Mkv * 2 * d, dev) if Mkv else None
        self.dq = _z(M * d, dev) if Mkv else None  # cross-attention dq: lives until the layer's deferred weight gradients
        # dropout(dh): the gradient entering a sub-layer whose output was dropped (hidden dropout); one buffer per block
        # kind so that a layer's deferred weight gradients can read both
        self.dm = {"attn": _z(M * d, dev), "ffn": _z(M * d, dev), "cross": _z(M * d, dev)}
        npart = max(ops.layernorm_bwd_partial_floats(M, d), ops.colsum_partial_floats(max(M, Mkv), max(f, 3 * d)), 4096)
        self.part = _z(npart, dev, torch.float32)


def _masked_grad(dh, sv, sc: "Scratch", n, kind):
    """Gradient wrt the sub-layer output: dh itself, or dropout(dh) with the forward's mask when the output went
    through hidden-state dropout before the residual add."""
    p, seed = sv.get("hdrop", (0.0, 0))
    if p <= 0.0:
        return dh
    ops.dropout(dh, sc.dm[kind], n, p, seed)
    return sc.dm[kind]


def _masked_grad_fp8(dh, sv, sc: "Scratch", M, d, kind, fb):
    """_masked_grad on the fp8 path: the same bf16 dropout(dh) (or dh itself) for the weight gradient, and in the same pass
    the rows as e4m3 with one scale per row (fb = (p8t, w_off, dy8, row_scale, inv_w)) - the A operand of the sub-layer's
    data gradient through ca_gemm_fp8 against the transposed e4m3 weight copy."""
    p, seed = sv.get("hdrop", (0.0, 0))
    ops.dropout_rows_fp8(dh, sc.dm[kind] if p > 0.0 else None, fb[2], fb[3], M, d, p, seed)
    return sc.dm[kind] if p > 0.0 else dh


def _ln_bwd(st, ln, dx, sv, dh, dhin, sc: "Scratch", M, d, ln_part, pending):
    """The pre-norm's backward: dhin = dh + LN'(dx).  d gamma | d beta (adjacent in the flat buffer) either reduced right
    away, or left as partials in `ln_part` with the reduction appended to `pending`."""
    if pending is None:
        ops.layernorm_bwd(dx, sv["hin"], st.view(ln + ".weight"), None, sv["st"], dh, dhin,
                          st.view(ln + ".weight", "g32"), st.view(ln + ".bias", "g32"), sc.part, M, d)
        return
    ops.layernorm_bwd(dx, sv["hin"], st.view(ln + ".weight"), None, sv["st"], dh, dhin, None, None, ln_part, M, d)
    pending.append((ln_part, ops.layernorm_bwd_partial_floats(M, d) // (2 * d), 2 * d, 2 * d, st.g32[st.off(ln + ".weight"):], True))


class SelfAttnBlock:
    """h_out = h_in + out_proj(attn(q, k, v)),  q|k|v = LN(h_in) Wqkv^T + bqkv."""

    def __init__(self, store, ln: str, attn: str, H: int, d: int, eps: float, causal: bool, qbias: str):
        self.st, self.ln, self.attn, self.H, self.d, self.eps, self.causal = store, ln, attn, H, d, eps, causal
        self.qbias = qbias  # name of the first of the three adjacent bias vectors (q, k, v)
        # positions of this block's bias gradients in the layer's contiguous bias vector (fused column sums of a
        # grouped weight-gradient launch); the encoder layout, a decoder layer sets its own
        self.cs_qkv, self.cs_o = 0, 3 * d

    def alloc(self, B, T, dev):
        d, H = self.d, self.H
        Tqp = (T + 31) // 32 * 32
        return dict(x=_z(B * T * d, dev), st=_z(B * T * 2, dev, torch.float32), qkv=_z(B * T * 3 * d, dev),
                    ctx=_z(B * T * d, dev), lse=_z(B * H * Tqp, dev, torch.float32), Dq=_z(B * H * Tqp, dev, torch.float32),
                    Tqp=Tqp)

    def _akw(self, B, T, sv, klen):
        d, H = self.d, self.H
        hd = d // H
        ap, aseed = sv.get("adrop", (0.0, 0))
        return dict(B=B, H=H, Tq=T, Tk=T, hd=hd, Tqp=sv["Tqp"], scale=hd ** -0.5, ldq=3 * d, ldk=3 * d, ldv=3 * d, ldo=d,
                    sqb=T * 3 * d, skb=T * 3 * d, svb=T * 3 * d, sob=T * d, q_off=0, k_off=d, v_off=2 * d, klen=klen,
                    causal=self.causal, dropout_p=ap, dropout_seed=aseed)

    def forward(self, hin, hout, sv, B, T, klen=None, hdrop=(0.0, 0), adrop=(0.0, 0)):
        """hdrop: (p, seed) of the hidden-state dropout on the block's output; adrop: of the dropout on the attention
        probabilities ($TF/models/whisper/modeling_whisper.py:234)."""
        st, d = self.st, self.d
        M = B * T
        sv["adrop"] = adrop
        fp8 = getattr(self, "fp8", None)  # (p8, scale, x8, rs): forward projection on the fp8 path
        if fp8 is not None:
            p8, scale, x8, rs = fp8
            ops.layernorm_fwd_fp8(hin, st.view(self.ln + ".weight"), st.view(self.ln + ".bias"), sv["x"], x8, rs, M, d, self.eps,
                                  stats=sv["st"])
            ops.gemm_fp8(x8, p8, sv["qkv"], a_row_scale=rs, b_scale=scale, M=M, N=3 * d, K=d, lda=d, ldb=d, ldc=3 * d,
                         b_off=st.off(self.attn + "q_proj.weight"), bias=st.p32, bias_off=st.off(self.qbias))
        else:
            ops.layernorm_fwd(hin, st.view(self.ln + ".weight"), st.view(self.ln + ".bias"), sv["x"], sv["st"], M, d, self.eps)
            ops.gemm(sv["x"], st.p16, sv["qkv"], M=M, N=3 * d, K=d, lda=d, ldb=d, ldc=3 * d,
                     b_off=st.off(self.attn + "q_proj.weight"), bias=st.p32, bias_off=st.off(self.qbias))
        o8 = getattr(self, "fp8_out", None) if (fp8 is not None and T >= 100 and adrop[0] == 0.0) else None
        if o8 is not None:
            # out_proj on the fp8 path: the attention kernel's output stage also writes the context as e4m3 (delayed
            # per-tensor scale, CaAttnDesc.O8)
            ops.attn_fwd(sv["qkv"], sv["qkv"], sv["qkv"], sv["ctx"], sv["lse"], O8=o8[0], o8_scale=o8[1], o8_amax=o8[3],
                         **self._akw(B, T, sv, klen))
            ops.gemm_fp8(o8[0], fp8[0], hout, a_scale=o8[2], b_scale=o8[4], M=M, N=d, K=d, lda=d, ldb=d, ldc=d,
                         b_off=st.off(self.attn + "out_proj.weight"), bias=st.p32, bias_off=st.off(self.attn + "out_proj.bias"),
                         epilogue=EPI_RESIDUAL, R=hin, ldr=d, dropout_p=hdrop[0], dropout_seed=hdrop[1])
        else:
            ops.attn_fwd(sv["qkv"], sv["qkv"], sv["qkv"], sv["ctx"], sv["lse"], **self._akw(B, T, sv, klen))
            ops.gemm(sv["ctx"], st.p16, hout, M=M, N=d, K=d, lda=d, ldb=d, ldc=d, b_off=st.off(self.attn + "out_proj.weight"),
                     bias=st.p32, bias_off=st.off(self.attn + "out_proj.bias"), epilogue=EPI_RESIDUAL, R=hin, ldr=d,
                     dropout_p=hdrop[0], dropout_seed=hdrop[1])
        sv["hin"], sv["klen"], sv["hdrop"] = hin, klen, hdrop

    def backward(self, dh, dhin, sv, sc: Scratch, B, T, defer=None, acc=True, sq=None, ln_part=None, pending=None):
        """dh: grad wrt h_out (kept intact); dhin: output buffer for grad wrt h_in (may alias nothing of sv).
        ln_part / pending: leave the norm's d gamma | d beta partials in `ln_part` and append their second-stage
        reduction to `pending` (the caller runs a layer's reductions as one launch, ops.reduce_rows_multi).
        defer: list collecting the block's weight-gradient problems instead of launching them (the caller launches
        the whole layer's group once dh and sc.dqkv are no longer needed elsewhere).  acc=False: the weight gradients
        overwrite (first micro-batch of a step, matrices not cleared); sq: {"o": (slots, off), "qkv": ...} where the
        weight-gradient GEMMs leave their per-tile sums of squares (CaGemmDesc.c_sumsq)."""
        sq = sq or {}
        st, d = self.st, self.d
        M = B * T
        o, g32, p16 = st.off, st.g32, st.p16
        fb = getattr(self, "fp8_bwd", None) if getattr(self, "fp8", None) is not None else None
        dy = _masked_grad(dh, sv, sc, M * d, "attn") if fb is None else _masked_grad_fp8(dh, sv, sc, M, d, "attn", fb)
        # with `defer` the bias gradients travel with the problems (fused into the grouped launch or done by it)
        if defer is None:
            ops.colsum(dy, d, M, d, g32, sc.part, out_off=o(self.attn + "out_proj.bias"))
        wg = [dict(dY=dy, X=sv["ctx"], M=d, N=d, K=M, lda=d, ldb=d, c_off=o(self.attn + "out_proj.weight"), accumulate=acc,
                   sq=sq.get("o"),
                   **(dict(bias_off=o(self.attn + "out_proj.bias"), part=sc.part, cs_off=self.cs_o) if defer is not None else {}))]
        if fb is not None:  # data gradient of out_proj on the fp8 path: e4m3 dY (row scales) x transposed e4m3 weights
            ops.gemm_fp8(fb[2], fb[0], sc.dctx, a_row_scale=fb[3], b_scale=fb[4], M=M, N=d, K=d, lda=d, ldb=d, ldc=d, b_off=fb[1])
        else:
            ops.gemm(dy, p16, sc.dctx, M=M, N=d, K=d, lda=d, b_layout=MNMAJOR, ldb=d, ldc=d, b_off=o(self.attn + "out_proj.weight"))
        qkv, dqkv = sv["qkv"], sc.dqkv
        ops.attn_bwd(qkv, qkv, qkv, sv["ctx"], sv["lse"], sc.dctx, sv["Dq"], dqkv, dqkv, dqkv, lddo=d, sdob=T * d, lddq=3 * d,
                     lddk=3 * d, lddv=3 * d, sdqb=T * 3 * d, sdkb=T * 3 * d, sdvb=T * 3 * d, dq_off=0, dk_off=d, dv_off=2 * d,
                     **self._akw(B, T, sv, sv["klen"]))
        if defer is None:
            ops.colsum(dqkv, 3 * d, M, 3 * d, g32, sc.part, out_off=o(self.qbias))
        wg.append(dict(dY=dqkv, X=sv["x"], M=3 * d, N=d, K=M, lda=3 * d, ldb=d, c_off=o(self.attn + "q_proj.weight"),
                       accumulate=acc, sq=sq.get("qkv"),
                       **(dict(bias_off=o(self.qbias), part=sc.part, cs_off=self.cs_qkv) if defer is not None else {})))
        if defer is not None:
            defer.extend(wg)
        else:
            ops.wgrad_gemm_group(wg, g32)  # both weight gradients of the block in one grouped launch
        ops.gemm(dqkv, p16, sc.dx, M=M, N=d, K=3 * d, lda=3 * d, b_layout=MNMAJOR, ldb=d, ldc=d, b_off=o(self.attn + "q_proj.weight"))
        _ln_bwd(st, self.ln, sc.dx, sv, dh, dhin, sc, M, d, ln_part, pending)


class CrossAttnBlock:
    """h_out = h_in + out_proj(attn(q, K, V)) with q from LN(h_in) and K|V = enc Wkv^T + [0|bv] [B*Te, 2d]."""

    def __init__(self, store, ln: str, attn: str, H: int, d: int, eps: float):
        self.st, self.ln, self.attn, self.H, self.d, self.eps = store, ln, attn, H, d, eps

    def alloc(self, B, L, Te, dev):
        d, H = self.d, self.H
        Lqp = (L + 31) // 32 * 32
        return dict(x=_z(B * L * d, dev), st=_z(B * L * 2, dev, torch.float32), q=_z(B * L * d, dev), ctx=_z(B * L * d, dev),
                    kv=_z(B * Te * 2 * d, dev), lse=_z(B * H * Lqp, dev, torch.float32), Dq=_z(B * H * Lqp, dev, torch.float32),
                    Tqp=Lqp)

    def _akw(self, B, L, Te, sv):
        d, H = self.d, self.H
        hd = d // H
        ap, aseed = sv.get("adrop", (0.0, 0))
        return dict(B=B, H=H, Tq=L, Tk=Te, hd=hd, Tqp=sv["Tqp"], scale=hd ** -0.5, ldq=d, ldk=2 * d, ldv=2 * d, ldo=d,
                    sqb=L * d, skb=Te * 2 * d, svb=Te * 2 * d, sob=L * d, k_off=0, v_off=d, dropout_p=ap, dropout_seed=aseed)

    def project_kv(self, enc, sv, B, Te):
        st, d = self.st, self.d
        ops.gemm(enc, st.p16, sv["kv"], M=B * Te, N=2 * d, K=d, lda=d, ldb=d, ldc=2 * d, b_off=st.off(self.attn + "k_proj.weight"),
                 bias=st.p32, bias_off=st.off(self.attn + "k_proj.bias__zero"))
        sv["enc"] = enc

    def forward(self, hin, hout, sv, B, L, Te, hdrop=(0.0, 0), adrop=(0.0, 0)):
        st, d = self.st, self.d
        M = B * L
        sv["adrop"] = adrop
        ops.layernorm_fwd(hin, st.view(self.ln + ".weight"), st.view(self.ln + ".bias"), sv["x"], sv["st"], M, d, self.eps)
        ops.gemm(sv["x"], st.p16, sv["q"], M=M, N=d, K=d, lda=d, ldb=d, ldc=d, b_off=st.off(self.attn + "q_proj.weight"),
                 bias=st.p32, bias_off=st.off(self.attn + "q_proj.bias"))
        ops.attn_fwd(sv["q"], sv["kv"], sv["kv"], sv["ctx"], sv["lse"], **self._akw(B, L, Te, sv))
        ops.gemm(sv["ctx"], st.p16, hout, M=M, N=d, K=d, lda=d, ldb=d, ldc=d, b_off=st.off(self.attn + "out_proj.weight"),
                 bias=st.p32, bias_off=st.off(self.attn + "out_proj.bias"), epilogue=EPI_RESIDUAL, R=hin, ldr=d,
                 dropout_p=hdrop[0], dropout_seed=hdrop[1])
        sv["hin"], sv["hdrop"] = hin, hdrop

    def backward(self, dh, dhin, sv, sc: Scratch, denc32, B, L, Te, defer=None, cs=(0, 0), ln_part=None, pending=None,
                 acc=True):
        """denc32: fp32 [B*Te, d] accumulator of the gradient wrt the encoder states (+=).  defer: list collecting
        the two token-side weight-gradient problems (out_proj, q_proj; bias gradients fused at cs = (cs_q, cs_o) of the
        layer's bias vector) for the layer's grouped launch; the k|v projection's (K = B*Te rows) goes out at once.
        acc=False: the three weight gradients overwrite their (uncleared) slots instead of adding to them."""
        st, d = self.st, self.d
        M, Mk = B * L, B * Te
        o, g32, p16 = st.off, st.g32, st.p16
        dy = _masked_grad(dh, sv, sc, M * d, "cross")
        if defer is None:
            ops.colsum(dy, d, M, d, g32, sc.part, out_off=o(self.attn + "out_proj.bias"))
            ops.wgrad_gemm(dy, sv["ctx"], g32, M=d, N=d, K=M, lda=d, ldb=d,
                           c_off=o(self.attn + "out_proj.weight"), accumulate=acc)
        else:
            defer.append(dict(dY=dy, X=sv["ctx"], M=d, N=d, K=M, lda=d, ldb=d, c_off=o(self.attn + "out_proj.weight"),
                              accumulate=acc, bias_off=o(self.attn + "out_proj.bias"), part=sc.part, cs_off=cs[1]))
        ops.gemm(dy, p16, sc.dctx, M=M, N=d, K=d, lda=d, b_layout=MNMAJOR, ldb=d, ldc=d, b_off=o(self.attn + "out_proj.weight"))
        dq, dkv = (sc.dx if defer is None else sc.dq), sc.dkv
        ops.attn_bwd(sv["q"], sv["kv"], sv["kv"], sv["ctx"], sv["lse"], sc.dctx, sv["Dq"], dq, dkv, dkv, lddo=d, sdob=L * d,
                     lddq=d, lddk=2 * d, lddv=2 * d, sdqb=L * d, sdkb=Te * 2 * d, sdvb=Te * 2 * d, dk_off=0, dv_off=d,
                     **self._akw(B, L, Te, sv))
        # q projection
        if defer is None:
            ops.colsum(dq, d, M, d, g32, sc.part, out_off=o(self.attn + "q_proj.bias"))
            ops.wgrad_gemm(dq, sv["x"], g32, M=d, N=d, K=M, lda=d, ldb=d,
                           c_off=o(self.attn + "q_proj.weight"), accumulate=acc)
        else:
            defer.append(dict(dY=dq, X=sv["x"], M=d, N=d, K=M, lda=d, ldb=d, c_off=o(self.attn + "q_proj.weight"),
                              accumulate=acc, bias_off=o(self.attn + "q_proj.bias"), part=sc.part, cs_off=cs[0]))
        ops.gemm(dq, p16, sc.dctx, M=M, N=d, K=d, lda=d, b_layout=MNMAJOR, ldb=d, ldc=d, b_off=o(self.attn + "q_proj.weight"))
        _ln_bwd(st, self.ln, sc.dctx, sv, dh, dhin, sc, M, d, ln_part, pending)
        # k|v projection of the encoder states
        ops.colsum(dkv, 2 * d, Mk, 2 * d, g32, sc.part, out_off=o(self.attn + "k_proj.bias__zero"))
        ops.wgrad_gemm(dkv, sv["enc"], g32, M=2 * d, N=d, K=Mk, lda=2 * d, ldb=d,
                       c_off=o(self.attn + "k_proj.weight"), accumulate=acc)
        ops.gemm(dkv, p16, denc32, M=Mk, N=d, K=2 * d, lda=2 * d, b_layout=MNMAJOR, ldb=d, ldc=d,
                 b_off=o(self.attn + "k_proj.weight"), out_f32=True, accumulate=True)


class FFNBlock:
    """h_out = h_in + fc2(dropout(gelu(fc1(LN(h_in)))))."""

    def __init__(self, store, ln: str, fc1: str, fc2: str, d: int, f: int, eps: float):
        self.st, self.ln, self.fc1, self.fc2, self.d, self.f, self.eps = store, ln, fc1, fc2, d, f, eps
        self.cs_fc1, self.cs_fc2 = 4 * d, 4 * d + f  # (the encoder layer's bias vector; see SelfAttnBlock)

    def alloc(self, M, dev):
        return dict(x=_z(M * self.d, dev), st=_z(M * 2, dev, torch.float32), u=_z(M * self.f, dev), g=_z(M * self.f, dev))

    def forward(self, hin, hout, sv, M, dropout_p=0.0, seed=0, hdrop=(0.0, 0)):
        st, d, f = self.st, self.d, self.f
        fp8 = getattr(self, "fp8", None)
        fc2_8 = getattr(self, "fp8_fc2", None) if fp8 is not None else None
        if fp8 is not None:
            p8, scale, x8, rs = fp8
            ops.layernorm_fwd_fp8(hin, st.view(self.ln + ".weight"), st.view(self.ln + ".bias"), sv["x"], x8, rs, M, d, self.eps,
                                  stats=sv["st"])
            # (fc2 on the fp8 path: the GELU output also leaves fc1's epilogue as e4m3, delayed per-tensor scale)
            c8 = dict(C8=fc2_8[0], c8_scale=fc2_8[1], c8_amax=fc2_8[3]) if fc2_8 is not None else {}
            ops.gemm_fp8(x8, p8, sv["u"], C2=sv["g"], a_row_scale=rs, b_scale=scale, M=M, N=f, K=d, lda=d, ldb=d, ldc=f,
                         b_off=st.off(self.fc1 + ".weight"), bias=st.p32, bias_off=st.off(self.fc1 + ".bias"),
                         epilogue=EPI_GELU, dropout_p=dropout_p, dropout_seed=seed, stream_out=ops.STREAM_U, **c8)
        else:
            ops.layernorm_fwd(hin, st.view(self.ln + ".weight"), st.view(self.ln + ".bias"), sv["x"], sv["st"], M, d, self.eps)
            ops.gemm(sv["x"], st.p16, sv["u"], C2=sv["g"], M=M, N=f, K=d, lda=d, ldb=d, ldc=f, b_off=st.off(self.fc1 + ".weight"),
                     bias=st.p32, bias_off=st.off(self.fc1 + ".bias"), epilogue=EPI_GELU, dropout_p=dropout_p,
                     dropout_seed=seed, stream_out=ops.STREAM_U)
        if fc2_8 is not None:
            ops.gemm_fp8(fc2_8[0], fp8[0], hout, a_scale=fc2_8[2], b_scale=fc2_8[4], M=M, N=d, K=f, lda=f, ldb=f, ldc=d,
                         b_off=st.off(self.fc2 + ".weight"), bias=st.p32, bias_off=st.off(self.fc2 + ".bias"),
                         epilogue=EPI_RESIDUAL, R=hin, ldr=d, dropout_p=hdrop[0], dropout_seed=hdrop[1])
        else:
            ops.gemm(sv["g"], st.p16, hout, M=M, N=d, K=f, lda=f, ldb=f, ldc=d, b_off=st.off(self.fc2 + ".weight"), bias=st.p32,
                     bias_off=st.off(self.fc2 + ".bias"), epilogue=EPI_RESIDUAL, R=hin, ldr=d, dropout_p=hdrop[0],
                     dropout_seed=hdrop[1])
        sv["hin"], sv["drop"], sv["hdrop"] = hin, (dropout_p, seed), hdrop

    def backward(self, dh, dhin, sv, sc: Scratch, M, defer=None, acc=True, sq=None, ln_part=None, pending=None):
        sq = sq or {}
        st, d, f = self.st, self.d, self.f
        o, g32, p16 = st.off, st.g32, st.p16
        p, seed = sv["drop"]
        fb = getattr(self, "fp8_bwd", None) if getattr(self, "fp8", None) is not None else None
        dy = _masked_grad(dh, sv, sc, M * d, "ffn") if fb is None else _masked_grad_fp8(dh, sv, sc, M, d, "ffn", fb)
        if defer is None:
            ops.colsum(dy, d, M, d, g32, sc.part, out_off=o(self.fc2 + ".bias"))
        wg = [dict(dY=dy, X=sv["g"], M=d, N=f, K=M, lda=d, ldb=f, c_off=o(self.fc2 + ".weight"), accumulate=acc,
                   sq=sq.get("fc2"),
                   **(dict(bias_off=o(self.fc2 + ".bias"), part=sc.part, cs_off=self.cs_fc2) if defer is not None else {}))]
        du8 = getattr(self, "fp8_du", None) if fb is not None else None
        if fb is not None:  # fc2's data gradient on the fp8 path (GELU' epilogue as on the bf16 path)
            # (fc1's data gradient on it too: dU also leaves this epilogue as e4m3, delayed per-tensor scale - CaGemmDesc.C8)
            c8 = dict(C8=du8["buf"], c8_scale=du8["scale"], c8_amax=du8["amax"]) if du8 is not None else {}
            ops.gemm_fp8(fb[2], fb[0], sc.du, a_row_scale=fb[3], b_scale=fb[4], M=M, N=f, K=d, lda=d, ldb=d, ldc=f, b_off=fb[1],
                         epilogue=EPI_DGELU, R=sv["u"], ldr=f, dropout_p=p, dropout_seed=seed, **c8)
        else:
            ops.gemm(dy, p16, sc.du, M=M, N=f, K=d, lda=d, b_layout=MNMAJOR, ldb=f, ldc=f, b_off=o(self.fc2 + ".weight"),
                     epilogue=EPI_DGELU, R=sv["u"], ldr=f, dropout_p=p, dropout_seed=seed)
        if defer is None:
            ops.colsum(sc.du, f, M, f, g32, sc.part, out_off=o(self.fc1 + ".bias"))
        wg.append(dict(dY=sc.du, X=sv["x"], M=f, N=d, K=M, lda=f, ldb=d, c_off=o(self.fc1 + ".weight"), accumulate=acc,
                       sq=sq.get("fc1"),
                       **(dict(bias_off=o(self.fc1 + ".bias"), part=sc.part, cs_off=self.cs_fc1) if defer is not None else {})))
        if defer is not None:
            defer.extend(wg)
        else:
            ops.wgrad_gemm_group(wg, g32)
        if du8 is not None and du8["ready"][0]:
            # dX = dU W1 through ca_gemm_fp8: e4m3 dU x the transposed e4m3 copy of W1 (the scale of dU comes from the
            # previous step's amax; until one backward has measured it the bf16 GEMM below runs)
            ops.gemm_fp8(du8["buf"], fb[0], sc.dx, a_scale=du8["inv"], b_scale=du8["inv_w"], M=M, N=d, K=f, lda=f, ldb=f, ldc=d,
                         b_off=du8["w_off"])
        else:
            ops.gemm(sc.du, p16, sc.dx, M=M, N=d, K=f, lda=f, b_layout=MNMAJOR, ldb=d, ldc=d, b_off=o(self.fc1 + ".weight"))
        _ln_bwd(st, self.ln, sc.dx, sv, dh, dhin, sc, M, d, ln_part, pending)
