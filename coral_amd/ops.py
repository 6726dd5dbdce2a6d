"""Thin host wrappers over the C ABI (include/coral_amd.h): torch tensors in, kernel enqueued on
torch's current HIP stream.  PyTorch is used here for device memory and streams only — every
arithmetic op on the hot path is a kernel of libcoral_amd.so.  No fallbacks: a missing library
or a non-GPU tensor raises.
"""

from __future__ import annotations

import ctypes as C
import os

import torch

from . import _lib
from ._lib import (CaAttnDesc, EPI_DGELU, EPI_GELU, EPI_GELU_RESIDUAL, EPI_NONE, EPI_RESIDUAL, KMAJOR,
                   MNMAJOR, CaGemmDesc, CoralAmdError, check)

__all__ = ["KMAJOR", "MNMAJOR", "EPI_NONE", "EPI_GELU", "EPI_RESIDUAL", "EPI_DGELU",
           "EPI_GELU_RESIDUAL", "CoralAmdError"]

_ELT = {torch.bfloat16: 2, torch.float32: 4, torch.int32: 4, torch.uint8: 1, torch.int64: 8, torch.int16: 2,
        torch.uint32: 4}


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: torch.Tensor | None, off: int = 0) -> int | None:
    """Device pointer of `t` advanced by `off` ELEMENTS (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise CoralAmdError("coral_amd ops need device tensors (there is no CPU path)")
    return t.data_ptr() + off * _ELT[t.dtype]


def lib():
    return _lib.load()


_SIDE_STREAMS: dict = {}


def background_update_fits():
    """-> (fits, registers of the forward GEMM kernel, registers of the update kernel): can the capped-grid AdamW run
    under the next step's forward GEMMs (ca_background_update_fits, include/coral_amd.h)?"""
    regs = (C.c_int32 * 2)()
    rc = lib().ca_background_update_fits(regs)
    if rc < 0:
        check(rc, "ca_background_update_fits")
    return bool(rc), int(regs[0]), int(regs[1])


def side_stream(device, role: str, priority: int = 0):
    """The process's side stream for `role` ("wgrad", "optimizer", "exchange", "gather", "copy") on `device`: created once
    and shared by every engine / trainer of the process.  torch hands out pool streams round-robin and HIP folds them
    onto a few hardware queues, so a fresh stream per engine makes the stream -> queue assignment (hence what can overlap
    with what) depend on how many engines the process has built before: measured on whisper-large-turbo, the second
    fp8 engine of a process ran 73.7 ms/step against 72.0 for the first (weight-gradient and optimiser kernels 10-30 %
    longer), identical in every other respect.  CA_SHARED_STREAMS=0 restores a stream per engine (the A/B switch)."""
    dev = torch.device(device)
    if os.environ.get("CA_SHARED_STREAMS", "1") == "0":
        return torch.cuda.Stream(device=dev, priority=priority)
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), role, priority)
    st = _SIDE_STREAMS.get(key)
    if st is None:
        st = _SIDE_STREAMS[key] = torch.cuda.Stream(device=dev, priority=priority)
    return st


def _gemm_desc(A, B, Cout, *, M, N, K, lda, ldb, ldc, a_layout=KMAJOR, b_layout=KMAJOR, a_off=0,
         b_off=0, c_off=0, bias=None, bias_off=0, R=None, r_off=0, ldr=0, C2=None, c2_off=None,
         epilogue=EPI_NONE, out_f32=None, accumulate=False, alpha=1.0, batch1=1, batch2=1,
         sA=(0, 0), sB=(0, 0), sC=(0, 0), sR=(0, 0), sBias=(0, 0), a_kseg=0, a_kseg_stride=0, b_kseg=0,
         b_kseg_stride=0, dropout_p=0.0, dropout_seed=0, a_colsum=None, a_colsum_off=0,
         a_colsum_ld=0, c_row_index=None, c_row_mul=0, c_split_n=0, C_hi=None, c_hi_off=0, ldc_hi=0,
         c_sumsq=None, c_sumsq_off=0, stream_out=False, C8=None, c8_scale=None, c8_amax=None, a_ln=None):
    d = CaGemmDesc()
    if a_ln is not None:  # (gamma, beta, eps): A = LayerNorm(A rows) inside the skinny kernel's prologue
        d.a_ln_gamma, d.a_ln_beta, d.a_ln_eps = _p(a_ln[0]), _p(a_ln[1]), float(a_ln[2])
    d.c_stream_out = int(stream_out)
    if C8 is not None:  # third output of CA_EPI_GELU: the activation as e4m3 with a delayed per-tensor scale
        d.C8, d.c8_scale, d.c8_amax = _p(C8), _p(c8_scale), _p(c8_amax)
    if c_sumsq is not None:
        d.c_sumsq = _p(c_sumsq, c_sumsq_off)
    if c_split_n:
        d.c_split_n, d.C_hi, d.ldc_hi = c_split_n, _p(C_hi, c_hi_off), ldc_hi
    if c_row_index is not None:
        d.c_row_index, d.c_row_mul = _p(c_row_index), c_row_mul
    if a_colsum is not None:
        d.a_colsum = _p(a_colsum, a_colsum_off)
        d.a_colsum_ld = a_colsum_ld
    d.A, d.B = _p(A, a_off), _p(B, b_off)
    d.C = _p(Cout, c_off) if Cout is not None else None
    if C2 is not None:
        d.C2 = _p(C2, c_off if c2_off is None else c2_off)
    if R is not None:
        d.R = _p(R, r_off)
    if bias is not None:
        d.bias = _p(bias, bias_off)
    d.M, d.N, d.K = M, N, K
    d.a_layout, d.b_layout = a_layout, b_layout
    d.lda, d.ldb, d.ldc, d.ldr = lda, ldb, ldc, ldr
    d.a_kseg, d.b_kseg = a_kseg, b_kseg
    d.a_kseg_stride, d.b_kseg_stride = a_kseg_stride, b_kseg_stride
    d.batch1, d.batch2 = batch1, batch2
    d.sA1, d.sA2 = sA
    d.sB1, d.sB2 = sB
    d.sC1, d.sC2 = sC
    d.sR1, d.sR2 = sR
    d.sBias1, d.sBias2 = sBias
    d.epilogue = epilogue
    ref_out = Cout if Cout is not None else C2
    d.out_f32 = int(ref_out.dtype == torch.float32) if out_f32 is None else int(out_f32)
    d.accumulate = int(accumulate)
    d.alpha = alpha
    d.dropout_p = dropout_p
    d.dropout_seed = dropout_seed
    return d


def gemm(A, B, Cout, **kw):
    """C = epilogue(alpha * opA @ opB^T) — see CaGemmDesc in include/coral_amd.h."""
    d = _gemm_desc(A, B, Cout, **kw)
    check(lib().ca_gemm_bf16(C.byref(d), _stream()), "ca_gemm_bf16")


_SPLITK_WS: dict = {}
# tests (and CA_FUSE_BIAS=0) flip this to compare the fused bias gradient with the separate column-sum pass
FUSE_BIAS_GRAD = os.environ.get("CA_FUSE_BIAS", "1") == "1"
# A layer with a weight gradient on the fallback path: the others still take their bias gradients from the 256x256
# kernel when the token count is at least this (measured, interleaved on one box: whisper-large-turbo, 12 000 rows,
# 86.1 -> 84.7 ms - four column-sum passes over 12 000 x 1280 ... 5120 per layer disappear; XLS-R-1B, 3 992 rows,
# 46.4 -> 47.4 ms - there the passes are short and hide on the side stream, the fused sums slow the grouped launch)
FUSE_BIAS_PARTIAL_MIN_K = int(os.environ.get("CA_FUSE_BIAS_PARTIAL_MIN_K", "8192"))


def gemm_fp8(A8, B8, Cout, *, a_scale=None, b_scale=None, a_row_scale=None, **kw):
    """ca_gemm_fp8: A8 [M, K], B8 [N, K] uint8 tensors holding e4m3 bytes, a_scale / b_scale the device scalars
    ca_quantize_fp8 wrote, a_row_scale the [M] factors of layernorm_fwd_fp8; the other arguments as for gemm()."""
    d = _gemm_desc(A8, B8, Cout, **kw)
    d.a_scale, d.b_scale, d.a_row_scale = _p(a_scale), _p(b_scale), _p(a_row_scale)
    check(lib().ca_gemm_fp8(C.byref(d), _stream()), "ca_gemm_fp8")


FP8_AMAX_SLOTS = 64  # CA_FP8_AMAX_SLOTS: words per amax accumulator


def quantize_fp8_delayed(x, q, scale, amax_next, n=None):
    """One-pass quantisation with last step's scale (device scalar); this step's amax accumulates into amax_next."""
    check(lib().ca_quantize_fp8_delayed(_p(x), x.numel() if n is None else n, _p(q), _p(scale), _p(amax_next), _stream()),
          "ca_quantize_fp8_delayed")


def fp8_amax_rotate(amax_next, scale, inv_scale, count, margin=1.0):
    """amax words accumulated since the last rotation -> scales of the next step (zero words keep their old scale)."""
    check(lib().ca_fp8_amax_rotate(_p(amax_next), _p(scale), _p(inv_scale), count, float(margin), _stream()), "ca_fp8_amax_rotate")


def dropout_rows_fp8(x, y, q, row_scale, rows, Cn, p, seed):
    """y = dropout(x) (ca_dropout_bf16's mask; y None when p == 0) and the rows of y as e4m3 with one scale per row."""
    check(lib().ca_dropout_rows_fp8(_p(x), _p(y), _p(q), _p(row_scale), rows, Cn, float(p), int(seed), _stream()),
          "ca_dropout_rows_fp8")


def quantize_fp8_transposed(x, rows, cols, qt, scale, x_off=0, qt_off=0):
    """qt [cols, rows] = e4m3(x [rows, cols] * scale): the transposed e4m3 copy of a weight matrix."""
    check(lib().ca_quantize_fp8_transposed(_p(x, x_off), rows, cols, _p(qt, qt_off), _p(scale), _stream()),
          "ca_quantize_fp8_transposed")


def fp8_refresh_group(tasks):
    """One launch for a layer's e4m3 weight copies.  tasks: (x, x_off, rows, cols, q, q_off, qt or None, qt_off, scale,
    amax_next or None) with offsets in elements; straight copy as quantize_fp8_delayed, transposed copy as
    quantize_fp8_transposed."""
    for i in range(0, len(tasks), _lib.FP8_GROUP_MAX):
        part = tasks[i:i + _lib.FP8_GROUP_MAX]
        arr = (_lib.CaFp8RefreshTask * len(part))()
        for t, (x, x_off, rows, cols, q, q_off, qt, qt_off, scale, amax) in zip(arr, part):
            t.x_bf16, t.q_fp8, t.q_fp8_t = _p(x, x_off), _p(q, q_off), _p(qt, qt_off)
            t.scale, t.amax_next, t.rows, t.cols = _p(scale), _p(amax), rows, cols
        check(lib().ca_fp8_refresh_group(arr, len(part), _stream()), "ca_fp8_refresh_group")


def quantize_fp8(x, q, inv_scale, amax_ws, n=None):
    """bf16 tensor -> e4m3 bytes (uint8 tensor q) + inv_scale (1 float on the device)."""
    check(lib().ca_quantize_fp8(_p(x), x.numel() if n is None else n, _p(q), _p(inv_scale), _p(amax_ws), _stream()),
          "ca_quantize_fp8")


COLSUM_PARTS = 8  # rows of the fused bias-gradient partials (CaGemmDesc.a_colsum)
# CaGemmDesc.c_stream_out for outputs that are not read again soon: the FFN pre-activation (kept for the backward) and the
# weight gradients (read by the optimiser a backward later).  CA_STREAM_U=0 / CA_STREAM_WGRAD=0: default-policy stores (A/B)
STREAM_U = os.environ.get("CA_STREAM_U", "1") == "1"
STREAM_WGRAD = os.environ.get("CA_STREAM_WGRAD", "1") == "1"


def _wgrad_splits(M, N, K):
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    xt = ((M + 255) // 256) * ((N + 255) // 256)
    if xt < 160 and tiles <= 256:
        for s in (8, 4, 2):
            if tiles * s <= 512 and K % s == 0 and K // s >= 256:
                return s
    return 1


def sumsq_slots(M, N):
    """Per-tile partials CaGemmDesc.c_sumsq writes for an [M, N] fp32 output."""
    return ((M + 63) // 64) * ((N + 63) // 64)


def wgrad_gemm(dY, X, G, *, M, N, K, lda, ldb, c_off, accumulate, a_off=0, b_off=0, bias_off=None, part=None,
               cs=None, sq=None, Gb=None):
    """Weight gradient G[c_off : c_off + M*N] (+)= dY^T X  (dY [K, M] and X [K, N] token-major bf16, G fp32
    row-major [M, N]).  Bias gradient (dY.sum(0)): either `bias_off` (+ `part` workspace) for a separate column-sum
    pass into G[bias_off:], or `cs = (ws, off, ld)` to take partial column sums from the 256x256 kernel's A stream
    (CaGemmDesc.a_colsum; the caller adds the COLSUM_PARTS rows).
    Shapes that would leave most of the chip idle (fewer 128x128 tiles than half the workgroup slots) are split
    along K: the slices run as one batched GEMM into an fp32 workspace and a deterministic second pass adds them
    up (no atomics: the result does not depend on scheduling).
    sq = (slots, off): the squared norm of the result as sumsq_slots(M, N) partials at slots[off:] (from the GEMM's
    epilogue; on the split-K path the whole sum lands in slots[off] and the matrix's other slots keep their zeros).
    G may be a bf16 buffer (weight-matrix gradients kept as the reference's autocast produces them; never accumulated
    into): the bias gradient then goes to the fp32 buffer `Gb`."""
    splits = 1 if cs is not None else _wgrad_splits(M, N, K)
    g16 = G.dtype == torch.bfloat16
    if g16 and accumulate:
        raise CoralAmdError("wgrad_gemm: a bf16 gradient buffer is written, never accumulated into")
    if bias_off is not None:
        colsum(dY, lda, K, M, Gb if g16 else G, part, x_off=a_off, out_off=bias_off)
    if splits == 1:
        kw = dict(a_colsum=cs[0], a_colsum_off=cs[1], a_colsum_ld=cs[2]) if cs is not None else {}
        if sq is not None:
            kw.update(c_sumsq=sq[0], c_sumsq_off=sq[1])
        gemm(dY, X, G, M=M, N=N, K=K, a_layout=MNMAJOR, lda=lda, b_layout=MNMAJOR, ldb=ldb, ldc=N, c_off=c_off,
             a_off=a_off, b_off=b_off, out_f32=not g16, accumulate=accumulate, stream_out=STREAM_WGRAD, **kw)
        return
    ws = _SPLITK_WS.get(G.device)
    need = max(splits * M * N, M * N + 4096)  # (+ 4096: the norm pass's partials when the result is kept in bf16)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.float32, device=G.device)
        _SPLITK_WS[G.device] = ws
    Kc = K // splits
    gemm(dY, X, ws, M=M, N=N, K=Kc, a_layout=MNMAJOR, lda=lda, b_layout=MNMAJOR, ldb=ldb, ldc=N, a_off=a_off,
         b_off=b_off, out_f32=True, batch2=splits, sA=(0, Kc * lda), sB=(0, Kc * ldb), sC=(0, M * N))
    if g16:  # (tiny shapes only: add the slices in place, then round once)
        reduce_rows(ws, splits, M * N, M * N, ws, accumulate=False)
        cast_f32_bf16(ws, G, M * N, y_off=c_off)
        if sq is not None:
            cast_bf16_f32(G[c_off:], ws, M * N)  # the norm of what is stored
            sumsq(ws, M * N, sq[0][sq[1]:], ws[M * N:])
    else:
        reduce_rows(ws, splits, M * N, M * N, G[c_off:], accumulate=accumulate)
        if sq is not None:
            sumsq(G[c_off:], M * N, sq[0][sq[1]:], ws)  # (ws: free again, >= 4096 floats)
    if sq is not None:
        # (both gradient dtypes) the matrix's other slots may still hold the per-tile partials of a step that took the
        # direct path (K = B*T changes per batch with padding=longest, and with it _wgrad_splits): the norm is the sum
        # over ALL slots
        ns = sumsq_slots(M, N)
        if ns > 1:
            clear_f32(sq[0], ns - 1, off=sq[1] + 1)


def _xtiles(p):
    return ((p["M"] + 255) // 256) * ((p["N"] + 255) // 256)


def _fill(t):
    return t / (256.0 * ((t + 255) // 256))


GROUP_MAX = 8  # problems per grouped launch (X_GROUP_MAX in gemm.hip)


_PLAN_CACHE: dict = {}
_CS_DIRTY: dict = {}  # per bias workspace: the (offset, length) slices that hold fused partial column sums


def _wgrad_plan_idx(shapes: tuple):
    """wgrad_plan on (M, N, K) triples -> index lists (cached: the plan of a layer is the same every step)."""
    import itertools

    xt = [((m + 255) // 256) * ((n + 255) // 256) for m, n, _ in shapes]
    big = [i for i, (_, _, k) in enumerate(shapes) if _fill(xt[i]) >= 0.85 and k >= 512]
    rest = [i for i in range(len(shapes)) if i not in big and shapes[i][2] >= 512]
    small_k = [i for i in range(len(shapes)) if i not in big and shapes[i][2] < 512]
    groups = []
    # all of the under-filled problems together, when that fills >= 60 % of its rounds: one launch (and one pass for the
    # fused bias gradients) beats the best-filled subset plus stragglers on the split-K path
    if 1 < len(rest) <= GROUP_MAX and _fill(sum(xt[i] for i in rest)) >= 0.6:
        groups.append(list(rest))
        rest = []
    while len(rest) > 1:
        # the subset (2..GROUP_MAX problems) that fills its rounds of CUs best; stop when nothing reaches 70 %
        best, best_fill = None, 0.0
        for r in range(min(GROUP_MAX, len(rest)), 1, -1):
            for combo in itertools.combinations(rest, r):
                total = sum(xt[i] for i in combo)
                alone = sum(xt[i] / _fill(xt[i]) for i in combo)
                f = _fill(total)
                if f >= 0.7 and total / f < 0.9 * alone and (f > best_fill + 1e-9 or (abs(f - best_fill) < 1e-9 and best is not None
                                                                                 and len(combo) > len(best))):
                    best, best_fill = combo, f
        if best is None:
            break
        groups.append(list(best))
        rest = [i for i in rest if i not in best]
    return big, groups, rest + small_k


def wgrad_plan(problems: list):
    """-> (solo, groups, fallback): which weight gradients of a layer get their own launch of the 256x256 kernel,
    which share grouped launches (lists of <= GROUP_MAX) and which fall back to the general path (split-K when tiny)."""
    key = tuple((p["M"], p["N"], p["K"]) for p in problems)
    plan = _PLAN_CACHE.get(key)
    if plan is None:
        if len(_PLAN_CACHE) > 4096:
            _PLAN_CACHE.clear()
        plan = _PLAN_CACHE[key] = _wgrad_plan_idx(key)
    solo, groups, fallback = plan
    return [problems[i] for i in solo], [[problems[i] for i in g] for g in groups], [problems[i] for i in fallback]


def wgrad_gemm_group(problems: list, G, colsum_ws=None, colsum_ld=0, Gb=None) -> bool:
    """Weight gradients of one layer (dicts: dY, X, M, N, K, lda, ldb, c_off, accumulate, bias_off, part, cs_off).
    Under-filled problems that contract over the same tokens share one grouped launch of the 256x256 kernel when
    together they fill >= 70 % of a round of CUs and beat separate launches; well-filled ones keep their own launch.
    Measured: XLS-R-300M step 25.7 -> 23.6 ms, XLS-R-2B 92.0 -> 89.4 ms; Whisper blocks (128 tiles) stay on split-K.

    Bias gradients: when `colsum_ws` is given and at least one problem is served by the 256x256 kernel, each of those
    launches leaves COLSUM_PARTS rows of partial column sums at colsum_ws[p * colsum_ld + cs_off + m], the others put
    theirs into row 0, and the function returns True (the caller adds the rows with one reduce_rows); otherwise they
    are separate column-sum passes into G[bias_off:] and the function returns False."""
    solo, groups, fallback = wgrad_plan(problems)
    # Fused bias gradients need the 256x256 kernel: problems on the fallback path (split-K / smaller tiles) take a
    # separate column-sum pass instead - into row 0 of their slice of the workspace, so that the caller's ONE reduction
    # over the layer's bias vector still covers them (rows 1.. of the slice are cleared if an earlier step, with another
    # token count and plan, left fused partials there).
    fused = (FUSE_BIAS_GRAD and colsum_ws is not None and bool(solo or groups) and all("cs_off" in p for p in problems)
             and (not fallback or min(p["K"] for p in problems) >= FUSE_BIAS_PARTIAL_MIN_K))

    def bias_kw(p):
        return dict(cs=(colsum_ws, p["cs_off"], colsum_ld)) if fused else dict(bias_off=p.get("bias_off"), part=p.get("part"))

    def base(p):
        return {k: p[k] for k in ("M", "N", "K", "lda", "ldb", "c_off", "accumulate", "sq") if p.get(k) is not None or k != "sq"}

    dirty = _CS_DIRTY.setdefault(colsum_ws.data_ptr(), set()) if fused else None
    g16 = G.dtype == torch.bfloat16  # (bf16 matrix gradients: bias gradients go to the fp32 buffer Gb)
    Gbias = Gb if g16 else G
    for p in solo:
        wgrad_gemm(p["dY"], p["X"], G, **base(p), **bias_kw(p), Gb=Gbias)
        if fused:
            dirty.add((p["cs_off"], p["M"]))
    for p in fallback:
        if fused:
            key = (p["cs_off"], p["M"])
            if key in dirty:
                clear_ranges(colsum_ws, tuple((r * colsum_ld + p["cs_off"], p["M"]) for r in range(1, COLSUM_PARTS)))
                dirty.discard(key)
            colsum(p["dY"], p["lda"], p["K"], p["M"], colsum_ws, p["part"], accumulate=False, out_off=p["cs_off"])
            wgrad_gemm(p["dY"], p["X"], G, **base(p), Gb=Gbias)
        else:
            wgrad_gemm(p["dY"], p["X"], G, **base(p), **bias_kw(p), Gb=Gbias)
    for chunk in groups:
        arr = (CaGemmDesc * len(chunk))()
        for i, p in enumerate(chunk):
            if not fused and p.get("bias_off") is not None:
                colsum(p["dY"], p["lda"], p["K"], p["M"], Gbias, p["part"], out_off=p["bias_off"])
            kw = dict(a_colsum=colsum_ws, a_colsum_off=p["cs_off"], a_colsum_ld=colsum_ld) if fused else {}
            if p.get("sq") is not None:
                kw.update(c_sumsq=p["sq"][0], c_sumsq_off=p["sq"][1])
            arr[i] = _gemm_desc(p["dY"], p["X"], G, M=p["M"], N=p["N"], K=p["K"], a_layout=MNMAJOR, lda=p["lda"],
                                b_layout=MNMAJOR, ldb=p["ldb"], ldc=p["N"], c_off=p["c_off"], out_f32=not g16,
                                accumulate=p["accumulate"], stream_out=STREAM_WGRAD, **kw)
        check(lib().ca_gemm_bf16_group(arr, len(chunk), _stream()), "ca_gemm_bf16_group")
        if fused:
            dirty.update((p["cs_off"], p["M"]) for p in chunk)
    return fused


def layernorm_fwd(x, gamma, beta, y, stats, rows, Cn, eps=1e-5, act=0, x_off=0, y_off=0):
    """LayerNorm (+ GELU with act=1) over rows; fp32 tensors on either side select the fp32 forms (C <= 1024)."""
    xf, yf = x.dtype == torch.float32, y.dtype == torch.float32
    if xf or yf:
        check(lib().ca_layernorm_fwd_ex(_p(x, x_off), _p(gamma), _p(beta), _p(y, y_off), _p(stats), rows, Cn, eps, act,
                                        int(xf), int(yf), _stream()), "ca_layernorm_fwd_ex")
        return
    check(lib().ca_layernorm_fwd(_p(x, x_off), _p(gamma), _p(beta), _p(y, y_off), _p(stats), rows,
                                 Cn, eps, act, _stream()), "ca_layernorm_fwd")


def layernorm_fwd_fp8(x, gamma, beta, y, q, row_scale, rows, Cn, eps=1e-5, stats=None):
    """LayerNorm whose output is (also) written as e4m3 bytes with one scale per row (y, stats may be None)."""
    check(lib().ca_layernorm_fwd_fp8(_p(x), _p(gamma), _p(beta), _p(y), _p(q), _p(row_scale), _p(stats), rows, Cn, eps,
                                     _stream()), "ca_layernorm_fwd_fp8")


def layernorm_bwd_partial_floats(rows, Cn):
    return lib().ca_layernorm_bwd_partial_floats(rows, Cn)


def layernorm_bwd(dy, x, gamma, beta, stats, dres, dx, dgamma, dbeta, partial, rows, Cn, act=0):
    if x.dtype == torch.float32:  # the saved input was kept in fp32 (layernorm_fwd's fp32 form)
        check(lib().ca_layernorm_bwd_ex(_p(dy), _p(x), _p(gamma), _p(beta), _p(stats), _p(dres), _p(dx), _p(dgamma),
                                        _p(dbeta), _p(partial), rows, Cn, act, 1, _stream()), "ca_layernorm_bwd_ex")
        return
    check(lib().ca_layernorm_bwd(_p(dy), _p(x), _p(gamma), _p(beta), _p(stats), _p(dres), _p(dx),
                                 _p(dgamma), _p(dbeta), _p(partial), rows, Cn, act, _stream()),
          "ca_layernorm_bwd")


def colsum_partial_floats(rows, N):
    return lib().ca_colsum_partial_floats(rows, N)


def colsum(x, ld, rows, N, out, partial, accumulate=True, rowmask=None, x_off=0, out_off=0):
    check(lib().ca_colsum_bf16(_p(x, x_off), ld, rows, N, _p(rowmask), _p(out, out_off),
                               int(accumulate), _p(partial), _stream()), "ca_colsum_bf16")


def reduce_rows(partial, nparts, stride, n, out, accumulate=False):
    check(lib().ca_reduce_rows_f32(_p(partial), nparts, stride, n, _p(out), int(accumulate), _stream()),
          "ca_reduce_rows_f32")


def reduce_rows_multi(items):
    """Up to 4 reductions `out[i] (+)= sum_p partial[p * stride + i]` in one launch (ca_reduce_rows_multi):
    items = [(partial, nparts, stride, n, out, accumulate), ...]."""
    from ._lib import CaReduceDesc

    arr = (CaReduceDesc * len(items))()
    for k, (partial, nparts, stride, n, out, accumulate) in enumerate(items):
        arr[k].partial, arr[k].out = _p(partial), _p(out)
        arr[k].stride, arr[k].nparts, arr[k].n, arr[k].accumulate = stride, nparts, n, int(accumulate)
    check(lib().ca_reduce_rows_multi(arr, len(items), _stream()), "ca_reduce_rows_multi")


def dgelu_mul(dy, u, out, n):
    check(lib().ca_dgelu_mul(_p(dy), _p(u), _p(out), n, _stream()), "ca_dgelu_mul")


def dropout(x, y, n, p, seed):
    """y = x * keep / (1 - p) with the mask of (seed, flat index) - the one the EPI_RESIDUAL epilogue applies."""
    check(lib().ca_dropout_bf16(_p(x), _p(y), n, float(p), int(seed), _stream()), "ca_dropout_bf16")


def wave_normalize(x, lengths, y, B, N, eps=1e-7):
    check(lib().ca_wave_normalize(_p(x), _p(lengths), _p(y), B, N, eps, _stream()),
          "ca_wave_normalize")


def frame_lengths(attention_mask, kernels, strides, out):
    """int32 device mask [B, N] -> int32 [B] frames after the conv stack (one launch)."""
    B, N = attention_mask.shape
    n = len(kernels)
    k = (C.c_int32 * n)(*kernels)
    s = (C.c_int32 * n)(*strides)
    check(lib().ca_frame_lengths(_p(attention_mask), B, N, k, s, n, _p(out), _stream()), "ca_frame_lengths")


def pcm_prepare(pcm, lengths, y, mask, B, N, ld_in, peak_normalize=False, zero_mean_unit_var=True, eps=1e-7):
    """Raw PCM rows (int16 or fp32, device) -> normalised fp32 input_values + int32 attention_mask."""
    if pcm.dtype not in (torch.int16, torch.float32):
        raise CoralAmdError("pcm_prepare: PCM must be int16 or float32")
    check(lib().ca_pcm_prepare(_p(pcm), int(pcm.dtype == torch.int16), ld_in, _p(lengths), _p(y), _p(mask), B, N,
                               int(peak_normalize), int(zero_mean_unit_var), eps, _stream()), "ca_pcm_prepare")


def wave_scale(x, scale, y, B, N):
    check(lib().ca_wave_scale(_p(x), _p(scale), _p(y), B, N, _stream()), "ca_wave_scale")


def fir_filter(x, lengths, taps, ntaps, mode, y, B, N, max_taps):
    check(lib().ca_fir_filter(_p(x), _p(lengths), _p(taps), _p(ntaps), _p(mode), taps.shape[1], max_taps, _p(y), B, N,
                              _stream()), "ca_fir_filter")


def mix_noise(x, lengths, noise, noise_ld, noise_len, noise_off, snr_db, active, y, B, N):
    check(lib().ca_mix_noise(_p(x), _p(lengths), _p(noise), noise_ld, noise_len, _p(noise_off), _p(snr_db), _p(active),
                             _p(y), B, N, _stream()), "ca_mix_noise")


def white_noise(out, n, seed):
    check(lib().ca_white_noise(_p(out), n, seed, _stream()), "ca_white_noise")


def conv0_fwd(x, w, bias, gamma, beta, y, B, N, Cn, k, stride, eps=1e-5):
    check(lib().ca_conv0_ln_gelu_fwd(_p(x), _p(w), _p(bias), _p(gamma), _p(beta), _p(y), B, N, Cn,
                                     k, stride, eps, _stream()), "ca_conv0_ln_gelu_fwd")


def conv0_bwd_partial_floats(B, N, Cn, k, stride):
    return lib().ca_conv0_bwd_partial_floats(B, N, Cn, k, stride)


def conv0_bwd(x, w, bias, gamma, beta, dy, dw, dbias, dgamma, dbeta, partial, B, N, Cn, k,
              stride, eps=1e-5):
    check(lib().ca_conv0_ln_gelu_bwd(_p(x), _p(w), _p(bias), _p(gamma), _p(beta), _p(dy), _p(dw),
                                     _p(dbias), _p(dgamma), _p(dbeta), _p(partial), B, N, Cn, k,
                                     stride, eps, _stream()), "ca_conv0_ln_gelu_bwd")


def col2im_1d(dcol, dx, B, T, L, Cn, k, stride):
    check(lib().ca_col2im_1d(_p(dcol), _p(dx), B, T, L, Cn, k, stride, _stream()), "ca_col2im_1d")


def softmax_fwd(scores, probs, klen, BH, H, Tq, Tk, ld, causal=False):
    check(lib().ca_softmax_fwd(_p(scores), _p(probs), _p(klen), BH, H, Tq, Tk, ld, int(causal),
                               _stream()), "ca_softmax_fwd")


def softmax_bwd(dprobs, probs, dscores, scale, BH, Tq, Tk, ld):
    check(lib().ca_softmax_bwd(_p(dprobs), _p(probs), _p(dscores), scale, BH, Tq, Tk, ld,
                               _stream()), "ca_softmax_bwd")


def ctc_workspace_bytes(B, T, Lmax):
    return lib().ca_ctc_workspace_bytes(B, T, Lmax)


def ctc_loss_fwd_bwd(logits, labels, in_len, nll, grad, gscale, ws, B, T, V, ldv, Lmax, blank,
                     zero_infinity=True):
    check(lib().ca_ctc_loss_fwd_bwd(_p(logits), _p(labels), _p(in_len), _p(nll), _p(grad),
                                    _p(gscale), _p(ws), B, T, V, ldv, Lmax, blank,
                                    int(zero_infinity), _stream()), "ca_ctc_loss_fwd_bwd")


def ctc_greedy_decode(logits, in_len, raw, ids, out_len, B, T, V, ldv, blank):
    check(lib().ca_ctc_greedy_decode(_p(logits), _p(in_len), _p(raw), _p(ids), _p(out_len), B, T,
                                     V, ldv, blank, _stream()), "ca_ctc_greedy_decode")


def mask_frames(h, tmask, fmask, embed, flen, B, T, Cn):
    check(lib().ca_mask_frames(_p(h), _p(tmask), _p(fmask), _p(embed), _p(flen), B, T, Cn,
                               _stream()), "ca_mask_frames")


def regroup_pad(x, xg, B, T, G, Cg, pad):
    check(lib().ca_regroup_pad(_p(x), _p(xg), B, T, G, Cg, pad, _stream()), "ca_regroup_pad")


def posconv_partial_floats(K):
    return lib().ca_posconv_partial_floats(K)


def posconv_weight(v, g, wf, wb, norm, partial, d, Cg, K):
    check(lib().ca_posconv_weight(_p(v), _p(g), _p(wf), _p(wb), _p(norm), _p(partial), d, Cg, K,
                                  _stream()), "ca_posconv_weight")


def posconv_weight_bwd(dwf, v, g, norm, dv, dg, partial, d, Cg, K):
    check(lib().ca_posconv_weight_bwd(_p(dwf), _p(v), _p(g), _p(norm), _p(dv), _p(dg),
                                      _p(partial), d, Cg, K, _stream()), "ca_posconv_weight_bwd")


def cast_f32_bf16(x, y, n, x_off=0, y_off=0):
    check(lib().ca_cast_f32_bf16(_p(x, x_off), _p(y, y_off), n, _stream()), "ca_cast_f32_bf16")


def cast_bf16_f32(x, y, n):
    check(lib().ca_cast_bf16_f32(_p(x), _p(y), n, _stream()), "ca_cast_bf16_f32")


def transpose_f32_bf16(x, y, rows, cols):
    check(lib().ca_transpose_f32_bf16(_p(x), _p(y), rows, cols, _stream()), "ca_transpose_f32_bf16")


def conv_weight_reorder(w, wr, Co, Ci, k, w_off=0):
    check(lib().ca_conv_weight_reorder(_p(w, w_off), _p(wr), Co, Ci, k, _stream()),
          "ca_conv_weight_reorder")


def conv_weight_grad_reorder(dwr, dw, Co, Ci, k, dw_off=0):
    check(lib().ca_conv_weight_grad_reorder(_p(dwr), _p(dw, dw_off), Co, Ci, k, _stream()),
          "ca_conv_weight_grad_reorder")


def clear_f32(x, n, off=0):
    """x[off : off + n] = 0 (fp32 buffer) as one ca_clear_ranges launch."""
    clear_ranges(x, [(off, n)])


_CLEAR_TABLES: dict = {}


def clear_ranges(x, ranges):
    """Zero the listed (offset, length) ELEMENT ranges of the flat buffer `x` in one ca_clear_ranges launch (two when
    ranges of at least and under 1 MB are mixed).  The byte-range tables live on the device, cached per (buffer, ranges):
    build the list once and pass the same tuple."""
    elt = _ELT[x.dtype]
    key = (x.data_ptr(), elt, tuple(ranges))
    ent = _CLEAR_TABLES.get(key)
    if ent is None:
        if len(_CLEAR_TABLES) > 4096:
            # The tables are read by launches on side streams the caching allocator knows nothing about (weight-gradient,
            # optimiser, communication): a table must not be freed - and its memory handed to the next H2D copy - while
            # a queued ca_clear_ranges may still read it.  Retire the cache only with the device idle.
            torch.cuda.synchronize()
            _CLEAR_TABLES.clear()
        rows = [[int(a) * elt, int(n) * elt] for a, n in ranges if n > 0]
        if any(a % 4 or n % 4 for a, n in rows):
            raise CoralAmdError("clear_ranges: ranges must be multiples of 4 bytes")
        # The launch is a (blocks per range) x (ranges) grid sized by the LONGEST range: one 200-MB embedding gradient
        # among 600 bias-sized ranges made it 600 K workgroups, nearly all of them idle (194 us for what HBM does in
        # 45).  Long and short ranges therefore go out as two launches.
        big = [r for r in rows if r[1] >= (1 << 20)]
        small = [r for r in rows if r[1] < (1 << 20)]
        ent = tuple((torch.tensor(part, dtype=torch.int64, device=x.device), len(part), max(n for _, n in part))
                    for part in (big, small) if part)
        _CLEAR_TABLES[key] = ent
    for table, count, longest in ent:
        check(lib().ca_clear_ranges(_p(x), _p(table), count, longest, _stream()), "ca_clear_ranges")


def sumsq(g, n, out, partial, accumulate=False):
    check(lib().ca_sumsq_f32(_p(g), n, _p(out), int(accumulate), _p(partial), _stream()),
          "ca_sumsq_f32")


def sumsq_ranges(g, chunks, nchunks, out, partial, accumulate=False):
    """out[0] (+)= sum of g^2 over `chunks` (device int64 [nchunks, 2]: offset, length in floats)."""
    check(lib().ca_sumsq_ranges_f32(_p(g), _p(chunks), nchunks, _p(out), int(accumulate), _p(partial), _stream()),
          "ca_sumsq_ranges_f32")


def sum_f32(x, n, out, partial, accumulate=False):
    check(lib().ca_sum_f32(_p(x), n, _p(out), int(accumulate), _p(partial), _stream()), "ca_sum_f32")


def adamw_step(p, m, v, g, p16, n, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0,
               max_norm=0.0, gnorm_sq=None, max_blocks=0):
    """AdamW with the clip coefficient folded in; max_blocks > 0 caps the grid (ca_adamw_step_ex: the CU count makes
    it a background kernel that runs under the next forward's GEMMs)."""
    fn = lib().ca_adamw_step_g16 if g.dtype == torch.bfloat16 else lib().ca_adamw_step_ex  # (bf16 weight-matrix gradients)
    check(fn(_p(p), _p(m), _p(v), _p(g), _p(p16), n, lr, beta1, beta2, eps,
             weight_decay, step, grad_scale, max_norm, _p(gnorm_sq), int(max_blocks), _stream()),
          "ca_adamw_step")


def logmel_workspace_bytes(B):
    return lib().ca_logmel_workspace_bytes(B)


def logmel(wave, mel_filters, out, ws, B, N, n_mels):
    check(lib().ca_logmel(_p(wave), _p(mel_filters), _p(out), _p(ws), B, N, n_mels, _stream()),
          "ca_logmel")


def cross_entropy_fwd_bwd(logits, labels, loss_sum, count, grad, rows, V, ldv, ignore_index=-100):
    check(lib().ca_cross_entropy_fwd_bwd(_p(logits), _p(labels), _p(loss_sum), _p(count), _p(grad),
                                         rows, V, ldv, ignore_index, _stream()),
          "ca_cross_entropy_fwd_bwd")


def argmax_masked(logits, suppress, out, rows, V, ldv):
    check(lib().ca_argmax_masked(_p(logits), _p(suppress), _p(out), rows, V, ldv, _stream()),
          "ca_argmax_masked")


def argmax_advance(logits, suppress, out, rows, V, ldv, done, ids, tok, pos, klen, pad_id, eos_id):
    """argmax_masked + the greedy step's bookkeeping (record the token, finished rows take pad, move the cursors)."""
    if done.dtype != torch.bool or ids.dtype != torch.int64 or ids.dim() != 2 or ids.stride(1) != 1:
        raise CoralAmdError("argmax_advance: done must be bool, ids a row-major int64 matrix")
    check(lib().ca_argmax_advance(_p(logits), _p(suppress), _p(out), rows, V, ldv, done.data_ptr(), _p(ids), ids.stride(0),
                                  _p(tok), _p(pos), _p(klen), int(pad_id), int(eos_id), _stream()), "ca_argmax_advance")


def whisper_decode_token_supported(B, d, f, H, V) -> bool:
    """True when ca_whisper_decode_token (one persistent launch per decoded token) takes this shape on this device."""
    return bool(lib().ca_whisper_decode_token_supported(B, d, f, H, V))


def whisper_decode_token(desc):
    """One decoded token for every clip of the batch in one launch (`desc`: a filled _lib.CaDecodeDesc whose tensors the
    caller keeps alive)."""
    check(lib().ca_whisper_decode_token(C.byref(desc), _stream()), "ca_whisper_decode_token")


def embed_tokens(table, pos, ids, pos_ids, y, rows, Cn):
    check(lib().ca_embed_tokens(_p(table), _p(pos), _p(ids), _p(pos_ids), _p(y), rows, Cn,
                                _stream()), "ca_embed_tokens")


def embed_tokens_bwd(dy, ids, pos_ids, dtable, dpos, rows, Cn, dtable_off=0, dpos_off=0):
    check(lib().ca_embed_tokens_bwd(_p(dy), _p(ids), _p(pos_ids), _p(dtable, dtable_off), _p(dpos, dpos_off), rows, Cn,
                                    _stream()), "ca_embed_tokens_bwd")


def prof_begin():
    check(lib().ca_prof_begin(), "ca_prof_begin")


def prof_end():
    """-> list of 24 dicts (kernel S/L/X x segmented-K x layout NT, NN, TN, TT): ms, count, flops of the
    GEMM launches.  `kernel` is the symbol rocprofv3 reports for the same launches."""
    ms, cnt, fl = (C.c_double * 24)(), (C.c_int64 * 24)(), (C.c_double * 24)()
    check(lib().ca_prof_end(ms, cnt, fl), "ca_prof_end")
    out = []
    for k, sym in enumerate(("ca_gemm_kernel", "ca_gemm_kernel_l", "ca_gemm_kernel_x")):
        for ks in range(2):
            for v in range(4):
                i = k * 8 + ks * 4 + v
                targs = f"{v >> 1}, {v & 1}" if k == 1 else f"{v >> 1}, {v & 1}, {'true' if ks else 'false'}"
                arg = "CaGemmGroup" if k == 2 else "CaGemmDesc"  # kernel X takes the (possibly one-problem) group struct
                out.append(dict(kernel=f"void {sym}<{targs}>({arg})", ms=ms[i], count=cnt[i], flops=fl[i]))
    return out


def _attn_desc(Q, K, V, O, lse, *, B, H, Tq, Tk, hd, Tqp, scale, ldq, ldk, ldv, ldo, sqb, skb, svb, sob,
               q_off=0, k_off=0, v_off=0, o_off=0, klen=None, causal=False, dropout_p=0.0, dropout_seed=0,
               O8=None, o8_scale=None, o8_amax=None, split_ws=None):
    d = CaAttnDesc()
    if split_ws is not None:  # key split of the small-query kernel (attn_split_workspace)
        d.split_ws, d.split_ws_bytes = _p(split_ws), split_ws.numel() * _ELT[split_ws.dtype]
    if O8 is not None:  # the output also as e4m3 (delayed per-tensor scale): the fp8 operand of the out-projection
        d.O8, d.o8_scale, d.o8_amax = _p(O8, o_off), _p(o8_scale), _p(o8_amax)
    d.Q, d.K, d.V, d.O = _p(Q, q_off), _p(K, k_off), _p(V, v_off), _p(O, o_off)
    d.lse, d.klen = _p(lse), _p(klen)
    d.ldq, d.ldk, d.ldv, d.ldo = ldq, ldk, ldv, ldo
    d.sqb, d.skb, d.svb, d.sob = sqb, skb, svb, sob
    d.B, d.H, d.Tq, d.Tk, d.hd, d.Tqp, d.causal, d.scale = B, H, Tq, Tk, hd, Tqp, int(causal), scale
    d.dropout_p, d.dropout_seed = float(dropout_p), int(dropout_seed)
    return d


ATTN_SPLIT_MAX = 4  # CA_ATTN_SPLIT_MAX


def attn_split_workspace(B, H, device):
    """Zero-filled workspace of CA_ATTN_SPLIT_WS_BYTES(B, H) bytes for `split_ws=` of attn_fwd / decode_attn_qproj (Tq <= 16):
    the keys of one (clip, head) may then be dealt to several workgroups.  One launch at a time per workspace."""
    nbytes = (B * H * 4 + 255) // 256 * 256 + B * H * ATTN_SPLIT_MAX * 16 * 66 * 4
    return torch.zeros(nbytes // 4, dtype=torch.float32, device=device)


def attn_fwd(Q, K, V, O, lse, **kw):
    """Fused attention forward: O = softmax(scale Q K^T + masks) V, lse saved for the backward."""
    d = _attn_desc(Q, K, V, O, lse, **kw)
    check(lib().ca_attn_fwd(C.byref(d), _stream()), "ca_attn_fwd")


def decode_attn_qproj(x, gamma, beta, W, bias, K, V, O, *, d_model, eps, ldx, ldw, w_off=0, bias_off=0, **kw):
    """Greedy decoding, one token per clip: LayerNorm(x) -> query projection (W rows w_off.., bias) -> single-query
    attention over the K|V cache, one launch (ca_decode_attn_qproj); kw as for attn_fwd with Tq = 1 (no Q)."""
    d = _attn_desc(None, K, V, O, None, Tq=1, Tqp=32, ldq=0, sqb=0, **kw)
    check(lib().ca_decode_attn_qproj(C.byref(d), _p(x), ldx, _p(gamma), _p(beta), eps, _p(W, w_off), ldw, _p(bias, bias_off),
                                     d_model, _stream()), "ca_decode_attn_qproj")


def attn_bwd(Q, K, V, O, lse, dO, Dq, dQ, dK, dV, *, lddo, sdob, lddq, lddk, lddv, sdqb, sdkb, sdvb, do_off=0,
             dq_off=0, dk_off=0, dv_off=0, **kw):
    d = _attn_desc(Q, K, V, O, lse, **kw)
    d.dO, d.Dq = _p(dO, do_off), _p(Dq)
    d.dQ, d.dK, d.dV = _p(dQ, dq_off), _p(dK, dk_off), _p(dV, dv_off)
    d.lddo, d.sdob = lddo, sdob
    d.lddq, d.lddk, d.lddv, d.sdqb, d.sdkb, d.sdvb = lddq, lddk, lddv, sdqb, sdkb, sdvb
    check(lib().ca_attn_bwd(C.byref(d), _stream()), "ca_attn_bwd")
