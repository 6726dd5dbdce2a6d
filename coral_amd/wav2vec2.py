"""MI355X-native Wav2Vec2ForCTC engine (XLS-R family) behind CoRal's `model=wav2vec2-*` keys.

Host-side sequencing of the C-ABI kernels in libcoral_amd.so: forward, hand-written backward
and parameter/gradient storage.  It mirrors what `Wav2Vec2ForCTC.from_pretrained(...)` gives
CoRal (R/src/coral/wav2vec2.py:104-126): `model(input_values, attention_mask, labels)` ->
loss + logits ($TF/models/wav2vec2/modeling_wav2vec2.py:1667-1736), HF parameter names, CTC
with blank = pad id, reduction "sum", zero_infinity.

HBM layout
  * parameters: one flat fp32 master buffer + one flat fp32 gradient buffer + one flat bf16
    compute copy (same offsets), ordered [front | layer 0 | ... | layer L-1 | head] so that a
    data-parallel bucket is a contiguous slice (see coral_amd/trainer.py);
  * activations: channels-last [B*T, C] bf16; q,k,v of a layer live in one [B*T, 3d] matrix;
    attention scores/probabilities are [B, H, T, Tp] with Tp = T rounded up to 8;
  * conv layers 1..6 and the grouped positional conv run as implicit GEMMs over overlapping-row
    views (no im2col is ever materialised in the forward pass).
"""

from __future__ import annotations

import math
import os
from dataclasses import dataclass

import torch

from . import ops
from .staging import PinnedStager
from .ops import EPI_DGELU, EPI_GELU, EPI_GELU_RESIDUAL, EPI_NONE, EPI_RESIDUAL, KMAJOR, MNMAJOR


@dataclass
class Wav2Vec2Shape:
    """Architecture hyper-parameters (HF Wav2Vec2Config subset used by XLS-R)."""

    hidden_size: int = 1024
    num_hidden_layers: int = 24
    num_attention_heads: int = 16
    intermediate_size: int = 4096
    conv_dim: tuple = (512,) * 7
    conv_kernel: tuple = (10, 3, 3, 3, 3, 2, 2)
    conv_stride: tuple = (5, 2, 2, 2, 2, 2, 2)
    num_conv_pos_embeddings: int = 128
    num_conv_pos_embedding_groups: int = 16
    vocab_size: int = 46
    pad_token_id: int = 45
    layer_norm_eps: float = 1e-5
    ctc_loss_reduction: str = "sum"
    ctc_zero_infinity: bool = True
    activation_dropout: float = 0.0
    layerdrop: float = 0.0

    @property
    def head_dim(self):
        return self.hidden_size // self.num_attention_heads


# CoRal model keys -> XLS-R shapes (R/config/model/wav2vec2-{small,medium,large}.yaml:3)
CORAL_W2V2_SHAPES = {
    "wav2vec2-small": dict(hidden_size=1024, num_hidden_layers=24, intermediate_size=4096),
    "wav2vec2-medium": dict(hidden_size=1280, num_hidden_layers=48, intermediate_size=5120),
    "wav2vec2-large": dict(hidden_size=1920, num_hidden_layers=48, intermediate_size=7680),
}


def _r8(n: int) -> int:
    return (n + 7) // 8 * 8


# The conv stack's pre-norm tensors (outputs of the conv GEMMs 1..6) and the last conv block's output stay fp32, as under
# the reference's autocast, where nn.LayerNorm and the GELU behind it run in fp32
# ($TF/models/wav2vec2/modeling_wav2vec2.py:291-298,429-434): 0.5 GB more at B = 8 and seven bf16 roundings less in front
# of the transformer.  CA_CONV_F32=0 restores the bf16 tensors (A/B measurements only).
CONV_F32 = os.environ.get("CA_CONV_F32", "1") == "1"


def _arena_build(build, device, default_dtype):
    """Run `build(z)` twice - z(n, dt=...) first only adds up 256-byte aligned sizes, then hands out views of one
    zero-filled arena - so that a workspace costs one allocation and one fill."""
    esz = {torch.bfloat16: 2, torch.float32: 4, torch.uint8: 1, torch.int32: 4}
    total = [0]

    def measure(n, dt=default_dtype):
        total[0] += (int(n) * esz[dt] + 255) // 256 * 256
        return None

    build(measure)
    arena = torch.zeros(total[0], dtype=torch.uint8, device=device)
    pos = [0]

    def carve(n, dt=default_dtype):
        nb = int(n) * esz[dt]
        v = arena[pos[0]:pos[0] + nb].view(dt)
        pos[0] += (nb + 255) // 256 * 256
        return v

    w = build(carve)
    w["_arena"] = arena
    return w


class ParamStore:
    """Flat fp32 master / fp32 grad / bf16 compute buffers with HF-named views."""

    def __init__(self, shapes: list[tuple[str, tuple, str]], device):
        self.index: dict[str, tuple[int, tuple]] = {}
        self.buckets: dict[str, list[int]] = {}
        off = 0
        for name, shape, bucket in shapes:
            n = int(math.prod(shape))
            self.index[name] = (off, tuple(shape))
            b = self.buckets.setdefault(bucket, [off, off])
            off += _r8(n)
            b[1] = off
        self.numel = off
        self.device = device
        self.p32 = torch.zeros(off, dtype=torch.float32, device=device)
        self.g32 = torch.zeros(off, dtype=torch.float32, device=device)
        self.p16 = torch.zeros(off, dtype=torch.bfloat16, device=device)
        self._g16 = None

    @property
    def g16(self) -> torch.Tensor:
        """bf16 gradient buffer with the same offsets (allocated on first use): the weight-matrix gradients of a step that
        does not accumulate stay bf16 - what the reference's autocast produces - from the weight-gradient GEMMs to AdamW
        (engine.wgrad_bf16, trainer.py)."""
        if self._g16 is None:
            self._g16 = torch.zeros(self.numel, dtype=torch.bfloat16, device=self.device)
        return self._g16

    def off(self, name: str) -> int:
        return self.index[name][0]

    def view(self, name: str, which: str = "p32") -> torch.Tensor:
        off, shape = self.index[name]
        n = int(math.prod(shape))
        return getattr(self, which)[off:off + n].view(shape)

    def names(self):
        return list(self.index.keys())

    def refresh_bf16(self):
        ops.cast_f32_bf16(self.p32, self.p16, self.numel)


def w2v2_param_list(s: Wav2Vec2Shape) -> list[tuple[str, tuple, str]]:
    """(HF name, shape, bucket) in storage order; q/k/v of a layer are adjacent on purpose."""
    d, f = s.hidden_size, s.intermediate_size
    out = []
    cin = 1
    for i, (co, k) in enumerate(zip(s.conv_dim, s.conv_kernel)):
        p = f"wav2vec2.feature_extractor.conv_layers.{i}."
        out += [(p + "conv.weight", (co, cin, k), "front"), (p + "conv.bias", (co,), "front"),
                (p + "layer_norm.weight", (co,), "front"), (p + "layer_norm.bias", (co,), "front")]
        cin = co
    K, G = s.num_conv_pos_embeddings, s.num_conv_pos_embedding_groups
    out += [
        ("wav2vec2.feature_projection.layer_norm.weight", (cin,), "front"),
        ("wav2vec2.feature_projection.layer_norm.bias", (cin,), "front"),
        ("wav2vec2.feature_projection.projection.weight", (d, cin), "front"),
        ("wav2vec2.feature_projection.projection.bias", (d,), "front"),
        ("wav2vec2.masked_spec_embed", (d,), "front"),
        ("wav2vec2.encoder.pos_conv_embed.conv.bias", (d,), "front"),
        ("wav2vec2.encoder.pos_conv_embed.conv.parametrizations.weight.original0", (1, 1, K), "front"),
        ("wav2vec2.encoder.pos_conv_embed.conv.parametrizations.weight.original1", (d, d // G, K), "front"),
    ]
    for l in range(s.num_hidden_layers):
        p = f"wav2vec2.encoder.layers.{l}."
        b = f"layer{l}"
        # small tensors first (one contiguous slice to clear per step), then the matrices whose
        # gradients are written -- not accumulated -- by the first micro-batch's wgrad GEMMs
        out += [(p + "layer_norm.weight", (d,), b), (p + "layer_norm.bias", (d,), b),
                (p + "final_layer_norm.weight", (d,), b), (p + "final_layer_norm.bias", (d,), b)]
        # the four Linear biases are contiguous (q|k|v, out, ffn1, ffn2 = 5d + f floats): their gradients come out of
        # the weight-gradient kernels as partial column sums and are added in one pass (backward())
        for n in ("q_proj", "k_proj", "v_proj"):
            out.append((p + f"attention.{n}.bias", (d,), b))
        out += [(p + "attention.out_proj.bias", (d,), b),
                (p + "feed_forward.intermediate_dense.bias", (f,), b),
                (p + "feed_forward.output_dense.bias", (d,), b)]
        for n in ("q_proj", "k_proj", "v_proj"):
            out.append((p + f"attention.{n}.weight", (d, d), b))
        out += [(p + "attention.out_proj.weight", (d, d), b),
                (p + "feed_forward.intermediate_dense.weight", (f, d), b),
                (p + "feed_forward.output_dense.weight", (d, f), b)]
    out += [("wav2vec2.encoder.layer_norm.weight", (d,), "head"),
            ("wav2vec2.encoder.layer_norm.bias", (d,), "head"),
            ("lm_head.weight", (s.vocab_size, d), "head"), ("lm_head.bias", (s.vocab_size,), "head")]
    return out


class CTCOutput(dict):
    """`model(**batch)` result: out["loss"], out.loss, out.logits (HF CausalLMOutput-like)."""

    __getattr__ = dict.get

    def __getitem__(self, k):
        if isinstance(k, int):
            return [self["loss"], self["logits"]][k] if self.get("loss") is not None else self["logits"]
        return dict.__getitem__(self, k)


class Wav2Vec2CTCEngine:
    """Forward + backward of Wav2Vec2ForCTC as a fixed sequence of HIP kernels."""

    def __init__(self, shape: Wav2Vec2Shape, device="cuda:0", freeze_base: bool = False,
                 fused_attention: bool = True):
        self.s = shape
        self.fused_attention = fused_attention  # False: batched-GEMM + softmax kernels (A/B reference)
        self.device = torch.device(device)
        ops.lib()  # fail loudly if the HIP library is not built
        if not torch.cuda.is_available():
            raise ops.CoralAmdError("Wav2Vec2CTCEngine needs a GPU: there is no CPU path")
        s = shape
        assert s.conv_dim[0] == 512 and s.conv_kernel[0] == 10, "layer-0 kernel expects C=512,k=10"
        assert s.hidden_size % 8 == 0 and (s.hidden_size // s.num_conv_pos_embedding_groups) % 8 == 0
        assert s.head_dim % 8 == 0
        self.store = ParamStore(w2v2_param_list(s), self.device)
        self.freeze_base = freeze_base
        self.training = False
        self._ws = None
        self._ws_key = None
        self._stager = PinnedStager(self.device)
        self._saved = None
        self.step_seed = 0
        d = s.hidden_size
        G = s.num_conv_pos_embedding_groups
        K = s.num_conv_pos_embeddings
        dev = self.device
        # reordered / derived bf16 weights
        self.conv_wr = [None] + [
            torch.zeros(s.conv_dim[i] * s.conv_kernel[i] * s.conv_dim[i - 1], dtype=torch.bfloat16, device=dev)
            for i in range(1, len(s.conv_dim))]
        self.pc_wf = torch.zeros(d * K * (d // G), dtype=torch.bfloat16, device=dev)
        self.pc_wb = torch.zeros(d * K * (d // G), dtype=torch.bfloat16, device=dev)
        self.pc_norm = torch.zeros(K, dtype=torch.float32, device=dev)
        self.pc_partial = torch.zeros(ops.posconv_partial_floats(K), dtype=torch.float32, device=dev)
        self.zero_embed = torch.zeros(d, dtype=torch.bfloat16, device=dev)

    # ---- parameters ------------------------------------------------------------------------
    def load_state_dict(self, P: dict, strict: bool = True, seed: int = 4242, init_missing: bool = True) -> dict:
        """Copy HF-named fp32 tensors into the flat master buffer and refresh compute copies.

        strict=False follows `PreTrainedModel.from_pretrained` for the checkpoints CoRal finetunes from
        (R/src/coral/wav2vec2.py:107-126: a pretrained XLS-R base plus a freshly initialised CTC head): a bare
        `Wav2Vec2Model` state dict gets the `wav2vec2.` prefix, the pre-parametrize weight-norm names
        `...pos_conv_embed.conv.weight_g / weight_v` map to `parametrizations.weight.original0 / original1`,
        unexpected keys (`quantizer.*`, `project_q.*`, `project_hid.*` of the pretraining head) are ignored, and a
        missing or differently sized `lm_head.*` / `masked_spec_embed` is initialised from `seed` the way HF's
        `_init_weights` does (Linear: N(0, 0.02), zero bias; masked_spec_embed: U(0, 1)).  Anything else that is
        missing still raises; init_missing=False leaves the tolerated missing tensors as they are.  Returns {"missing": [...], "unexpected": [...]} like `load_state_dict` of torch."""
        if not strict:
            if not any(k.startswith("wav2vec2.") or k.startswith("lm_head.") for k in P):
                P = {"wav2vec2." + k: v for k, v in P.items()}
            ren = {}
            for k, v in P.items():
                if k.endswith("pos_conv_embed.conv.weight_g"):
                    k = k[:-len("weight_g")] + "parametrizations.weight.original0"
                elif k.endswith("pos_conv_embed.conv.weight_v"):
                    k = k[:-len("weight_v")] + "parametrizations.weight.original1"
                ren[k] = v
            P = ren
        names = self.store.names()
        fresh_ok = ("lm_head.weight", "lm_head.bias", "wav2vec2.masked_spec_embed")
        missing = [n for n in names if n not in P]
        if not strict:
            missing += [n for n in fresh_ok if n in P and int(P[n].numel()) != int(math.prod(self.store.index[n][1]))]
        hard = [n for n in missing if strict or n not in fresh_ok]
        if hard:
            raise KeyError(f"missing parameters: {hard[:4]}...")
        g = torch.Generator(device=self.device).manual_seed(seed)
        for n in names:
            v = self.store.view(n)
            if n in missing:
                if not init_missing:
                    continue  # keep the present value (resuming a run whose file follows HF's key set)
                if n == "lm_head.weight":
                    v.normal_(0.0, 0.02, generator=g)
                elif n == "lm_head.bias":
                    v.zero_()
                else:
                    v.uniform_(0.0, 1.0, generator=g)
                continue
            v.copy_(P[n].to(self.device, torch.float32).reshape(self.store.index[n][1]))
        self.refresh_compute_weights()
        return dict(missing=missing, unexpected=[k for k in P if k not in self.store.index])

    def state_dict(self) -> dict:
        return {n: self.store.view(n).detach().clone() for n in self.store.names()}

    def grad_dict(self) -> dict:
        return {n: self.store.view(n, "g32") for n in self.store.names()}

    # The trainer may still be updating parameter buckets on its optimiser stream when the next forward
    # starts (trainer.py: the HBM-bound AdamW overlaps the MFMA-bound forward); the forward waits for a
    # bucket's event right before the first kernel that reads its weights.
    weights_ready: dict | None = None

    def _await(self, bucket: str):
        ev = self.weights_ready.get(bucket) if self.weights_ready else None
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)

    # the four weight matrices of an encoder layer: (first parameter name, rows, columns)
    def _layer_matrices(self, l: int):
        d, f = self.s.hidden_size, self.s.intermediate_size
        pl = f"wav2vec2.encoder.layers.{l}."
        return [("qkv", pl + "attention.q_proj.weight", 3 * d, d), ("o", pl + "attention.out_proj.weight", d, d),
                ("fc1", pl + "feed_forward.intermediate_dense.weight", f, d),
                ("fc2", pl + "feed_forward.output_dense.weight", d, f)]

    def norm_plan(self):
        """Squared gradient norm without a pass over the layer weight matrices (>99 % of the buffer): their
        weight-gradient GEMMs leave per-tile sums of squares in `slots` (CaGemmDesc.c_sumsq; backward() passes the
        slices), everything else of the flat buffer is listed as chunks for ca_sumsq_ranges_f32.  None when the model
        is frozen up to the head (the norm then covers the head bucket only)."""
        if self.freeze_base:
            return None
        if getattr(self, "_norm_plan", None) is not None:
            return self._norm_plan
        st, L = self.store, self.s.num_hidden_layers
        off, soff, mats = 0, {}, []
        for l in range(L):
            for key, name, M, N in self._layer_matrices(l):
                soff[(l, key)] = off
                off += ops.sumsq_slots(M, N)
                mats.append((st.off(name), M * N))
        # complement of the matrices inside [0, numel), cut into chunks of <= 64 Ki floats
        chunks, pos = [], 0
        for a, n in sorted(mats) + [(st.numel, 0)]:
            while pos < a:
                m = min(65536, a - pos)
                chunks.append((pos, m))
                pos += m
            pos = max(pos, a + n)
        plan = dict(slots=torch.zeros(off, dtype=torch.float32, device=self.device), nslots=off, slot_off=soff,
                    chunks=torch.tensor(chunks, dtype=torch.int64, device=self.device), nchunks=len(chunks),
                    partial=torch.zeros(max(4096, len(chunks)), dtype=torch.float32, device=self.device))
        self._norm_plan = plan
        return plan

    def shard_ranges(self) -> dict:
        """{layer bucket: (first element of its weight matrices, bucket end)}: the part of every layer bucket a sharded
        optimiser (trainer.py, zero_stage) may split over the ranks - the forward reads these parameters through the
        bf16 compute copy only, the small tensors in front of them (LayerNorms, biases: read from the fp32 master) stay
        replicated."""
        st = self.store
        return {f"layer{l}": (st.off(f"wav2vec2.encoder.layers.{l}.attention.q_proj.weight"), st.buckets[f"layer{l}"][1])
                for l in range(self.s.num_hidden_layers)}

    def bf16_grad_ranges(self) -> dict:
        """{layer bucket: (lo, hi)}: the weight matrices whose gradients a single-micro-batch step may keep in bf16
        (trainer.py, ParamStore.g16): every layer's, = shard_ranges()."""
        return self.shard_ranges()

    def zero_grad(self, matrices: bool = True):
        """Clear gradients.  matrices=False clears everything except the transformer layers' weight
        matrices (>99 % of the bytes): the next backward(overwrite_matrices=True) writes those
        instead of accumulating, which saves a full read + write of the gradient buffer."""
        st = self.store
        if matrices:
            st.g32.zero_()
            self._pre_zeroed = False
            return
        if getattr(self, "_pre_zeroed", False):
            # the trainer already cleared these slices on its optimiser stream, right behind the AdamW launches that
            # consumed them (hidden under the forward instead of sitting in front of the backward)
            self._pre_zeroed = False
            return
        self.clear_small_grads()

    def clear_small_grads(self):
        """Zero everything except the transformer layers' weight matrices (see zero_grad): the front and head
        buckets and every layer's small tensors (LayerNorms, biases), as ONE launch over a cached range table."""
        st = self.store
        if getattr(self, "_small_ranges", None) is None:
            rs = [(lo, hi - lo) for lo, hi in (st.buckets[n] for n in ("front", "head"))]
            for l in range(self.s.num_hidden_layers):
                lo = st.off(f"wav2vec2.encoder.layers.{l}.layer_norm.weight")
                rs.append((lo, st.off(f"wav2vec2.encoder.layers.{l}.attention.q_proj.weight") - lo))
            self._small_ranges = tuple(rs)
        ops.clear_ranges(st.g32, self._small_ranges)

    def train(self, mode: bool = True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    def refresh_compute_weights(self):
        """fp32 masters -> bf16 copies, conv weight reorders, weight-normed pos-conv weights."""
        self.store.refresh_bf16()
        self.refresh_derived()

    # the derived weights in the order the forward needs them: the trainer records one event per part, so the conv
    # stack does not wait for the weight-normed positional conv to be rebuilt (_await("front_posconv"))
    derived_parts = ("conv", "posconv")

    def refresh_derived(self, part: str | None = None):
        """Weights whose compute copy is not a plain cast (after an optimiser step the flat bf16
        copy is already written by ca_adamw_step).  part: None = all, or one of `derived_parts`."""
        s, st = self.s, self.store
        if part in (None, "conv"):
            for i in range(1, len(s.conv_dim)):
                ops.conv_weight_reorder(st.p32, self.conv_wr[i], s.conv_dim[i], s.conv_dim[i - 1],
                                        s.conv_kernel[i],
                                        w_off=st.off(f"wav2vec2.feature_extractor.conv_layers.{i}.conv.weight"))
        if part not in (None, "posconv"):
            return
        d, G, K = s.hidden_size, s.num_conv_pos_embedding_groups, s.num_conv_pos_embeddings
        pre = "wav2vec2.encoder.pos_conv_embed.conv.parametrizations.weight."
        ops.posconv_weight(st.view(pre + "original1"), st.view(pre + "original0"), self.pc_wf, self.pc_wb,
                           self.pc_norm, self.pc_partial, d, d // G, K)

    # ---- shapes / workspaces -------------------------------------------------------------------
    def conv_lengths(self, N: int) -> list[int]:
        out, n = [], N
        for k, st in zip(self.s.conv_kernel, self.s.conv_stride):
            n = (n - k) // st + 1
            out.append(n)
        return out

    def feat_lengths(self, sample_lengths: torch.Tensor) -> torch.Tensor:
        n = sample_lengths.clone().long()
        for k, st in zip(self.s.conv_kernel, self.s.conv_stride):
            n = torch.div(n - k, st, rounding_mode="floor") + 1
        return n

    def _workspace(self, B: int, N: int):
        key = (B, N)
        if self._ws_key == key:
            return self._ws
        s, dev = self.s, self.device
        d, f, H = s.hidden_size, s.intermediate_size, s.num_attention_heads
        L = s.num_hidden_layers
        Ts = self.conv_lengths(N)
        T = Ts[-1]
        M = B * T
        Tp = _r8(T)
        G, K = s.num_conv_pos_embedding_groups, s.num_conv_pos_embeddings
        Cg = d // G
        bf, f32 = torch.bfloat16, torch.float32

        def build(z):
            # (run twice: once to measure, once to carve views out of ONE zero-filled arena - a workspace used to
            # be ~450 separate torch.zeros fills)
            w = {"B": B, "N": N, "Ts": Ts, "T": T, "M": M, "Tp": Tp}
            C0 = s.conv_dim[0]
            cdt = f32 if CONV_F32 else bf
            w["a"] = [z(B * Ts[i] * s.conv_dim[i], dt=cdt if i == 6 else bf) for i in range(7)]  # conv block outputs
            w["y"] = [None] + [z(B * Ts[i] * s.conv_dim[i], dt=cdt) for i in range(1, 7)]  # pre-LN conv outputs
            w["cstats"] = [None] + [z(B * Ts[i] * 2, dt=f32) for i in range(1, 7)]
            w["fp_stats"] = z(M * 2, dt=f32)
            w["xln"] = z(M * C0)
            w["h0"] = z(M * d)
            w["xg"] = z(B * G * (T + K) * Cg + 8 * Cg)
            w["pc_pre"] = z(M * d)
            w["h"] = [z(M * d) for _ in range(L + 1)]      # residual stream entering layer l (h[L] = out)
            w["x1"] = [z(M * d) for _ in range(L)]
            w["st1"] = [z(M * 2, dt=f32) for _ in range(L)]
            w["qkv"] = [z(M * 3 * d) for _ in range(L)]
            Tqp = (T + 31) // 32 * 32
            w["Tqp"] = Tqp
            if self.fused_attention:
                w["lse"] = [z(B * H * Tqp, dt=f32) for _ in range(L)]
                w["Dq"] = z(B * H * Tqp, dt=f32)
            else:
                w["P"] = [z(B * H * T * Tp) for _ in range(L)]
            w["ctx"] = [z(M * d) for _ in range(L)]
            w["h1"] = [z(M * d) for _ in range(L)]
            w["x2"] = [z(M * d) for _ in range(L)]
            w["st2"] = [z(M * 2, dt=f32) for _ in range(L)]
            w["u"] = [z(M * f) for _ in range(L)]
            w["g"] = [z(M * f) for _ in range(L)]
            if not self.fused_attention:
                w["S"] = z(B * H * T * Tp, dt=f32)         # transient scores / dprobs
            w["hf"] = z(M * d)
            w["stf"] = z(M * 2, dt=f32)
            Vp = _r8(s.vocab_size)
            w["Vp"] = Vp
            w["logits"] = z(M * Vp, dt=f32)
            w["dlogits"] = z(M * Vp, dt=f32)
            w["dlogits16"] = z(M * Vp)
            w["nll"] = z(B, dt=f32)
            # backward scratch
            w["dA"] = z(M * d)
            w["dB"] = z(M * d)
            w["dC"] = z(M * d)
            w["dqkv"] = z(M * 3 * d)
            # second copies of the buffers the weight gradients read (dY operands): with the weight gradients on their own
            # stream (backward()) a layer's dY must stay intact while the next layer's data gradients are being written
            w["dBr"] = [w["dB"], z(M * d), z(M * d)]
            w["dCr"] = [w["dC"], z(M * d)]
            w["dqkvr"] = [w["dqkv"], z(M * 3 * d)]
            if not self.fused_attention:
                w["dS"] = z(B * H * T * Tp)
            w["du"] = z(M * f)
            w["dur"] = [w["du"], z(M * f)]
            w["dxg"] = z(B * G * (T + K) * Cg + 8 * Cg)
            w["dwf"] = z(d * K * Cg, dt=f32)
            # partial column sums of the layer's four dY (fused bias gradients): rows that a problem with fewer than
            # COLSUM_PARTS tile columns never writes stay zero
            w["bias_ws"] = z(ops.COLSUM_PARTS * (5 * d + s.intermediate_size), dt=f32)
            nmax = max(B * Ts[i] * s.conv_kernel[i] * s.conv_dim[i - 1] for i in range(1, 7))
            w["dcol"] = z(nmax)
            w["dconv"] = [z(B * Ts[i] * s.conv_dim[i]) for i in range(7)]   # grads wrt conv block outputs
            w["dy"] = z(max(B * Ts[i] * s.conv_dim[i] for i in range(1, 7)))
            w["dwr"] = z(max(s.conv_dim[i] * s.conv_kernel[i] * s.conv_dim[i - 1] for i in range(1, 7)), dt=f32)
            w["dwr_part"] = z(B * max(s.conv_dim[i] * s.conv_kernel[i] * s.conv_dim[i - 1] for i in range(1, 7)), dt=f32)
            pf = max(
                ops.layernorm_bwd_partial_floats(B * Ts[1], 512), ops.layernorm_bwd_partial_floats(M, d),
                ops.colsum_partial_floats(M, max(f, 3 * d)), ops.colsum_partial_floats(B * Ts[1], 512),
                ops.conv0_bwd_partial_floats(B, N, C0, s.conv_kernel[0], s.conv_stride[0]), 4096)
            w["partial"] = z(pf, dt=f32)
            w["partial_w"] = z(pf, dt=f32)  # the weight-gradient stream's own scratch
            # LayerNorm-backward partials (d gamma | d beta per row block) of a layer's two norms, two layers in flight:
            # their second-stage reductions run on the weight-gradient stream (backward())
            w["ln_parts"] = ops.layernorm_bwd_partial_floats(M, d) // (2 * d)
            w["ln_partial"] = [[z(ops.layernorm_bwd_partial_floats(M, d), dt=f32) for _ in range(2)] for _ in range(2)]

            return w

        w = _arena_build(build, dev, bf)
        self._ws, self._ws_key = w, key
        return w

    # ---- forward -------------------------------------------------------------------------------
    def __call__(self, input_values, attention_mask=None, labels=None, mask_time=None,
                 mask_feature=None, layer_keep=None):
        return self.forward(input_values, attention_mask, labels, mask_time, mask_feature, layer_keep)

    def forward(self, input_values, attention_mask=None, labels=None, mask_time=None,
                mask_feature=None, layer_keep=None) -> CTCOutput:
        """input_values f32 [B,N] (already zero-mean/unit-var: the reference's feature extractor
        output), attention_mask [B,N] or None, labels i64/i32 [B,L] (-100 padded) or None."""
        s, st = self.s, self.store
        dev = self.device
        x = self._stager.to_device(input_values, torch.float32, "x")
        B, N = x.shape
        w = self._workspace(B, N)
        d, f, H, hd = s.hidden_size, s.intermediate_size, s.num_attention_heads, s.head_dim
        L = s.num_hidden_layers
        Ts, T, M, Tp, Vp = w["Ts"], w["T"], w["M"], w["Tp"], w["Vp"]
        eps = s.layer_norm_eps
        p32, p16 = st.p32, st.p16
        o = st.off

        if attention_mask is not None:
            am = self._stager.to_device(attention_mask, torch.int32, "am")
            flen = torch.empty(B, dtype=torch.int32, device=dev)
            ops.frame_lengths(am, s.conv_kernel, s.conv_stride, flen)
        else:
            if w.get("flen_full") is None:  # (cached: no fill kernel per step)
                w["flen_full"] = torch.full((B,), T, dtype=torch.int32, device=dev)
            flen = w["flen_full"]
        keep = [True] * L if layer_keep is None else list(layer_keep)
        w["flen"] = flen
        self._await("front")

        # feature encoder
        p0 = "wav2vec2.feature_extractor.conv_layers.0."
        ops.conv0_fwd(x, st.view(p0 + "conv.weight"), st.view(p0 + "conv.bias"),
                      st.view(p0 + "layer_norm.weight"), st.view(p0 + "layer_norm.bias"), w["a"][0],
                      B, N, s.conv_dim[0], s.conv_kernel[0], s.conv_stride[0], eps)
        Lin = Ts[0]
        for i in range(1, 7):
            pi = f"wav2vec2.feature_extractor.conv_layers.{i}."
            k, sd, Ci, Co = s.conv_kernel[i], s.conv_stride[i], s.conv_dim[i - 1], s.conv_dim[i]
            ops.gemm(w["a"][i - 1], self.conv_wr[i], w["y"][i], M=Ts[i], N=Co, K=k * Ci, lda=sd * Ci,
                     ldb=k * Ci, ldc=Co, bias=p32, bias_off=o(pi + "conv.bias"), batch2=B,
                     sA=(0, Lin * Ci), sC=(0, Ts[i] * Co))
            ops.layernorm_fwd(w["y"][i], st.view(pi + "layer_norm.weight"), st.view(pi + "layer_norm.bias"),
                              w["a"][i], w["cstats"][i], B * Ts[i], Co, eps, act=1)
            Lin = Ts[i]
        feats = w["a"][6]
        C6 = s.conv_dim[6]
        # feature projection
        fp = "wav2vec2.feature_projection."
        ops.layernorm_fwd(feats, st.view(fp + "layer_norm.weight"), st.view(fp + "layer_norm.bias"),
                          w["xln"], w["fp_stats"], M, C6, eps)
        ops.gemm(w["xln"], p16, w["h0"], M=M, N=d, K=C6, lda=C6, ldb=C6, ldc=d,
                 b_off=o(fp + "projection.weight"), bias=p32, bias_off=o(fp + "projection.bias"))
        # SpecAugment + padding
        # (host-sampled masks and labels go through pinned staging: a pageable .to(device) here would stall the host
        # until the GPU had drained the previous step)
        tm = self._stager.to_device(mask_time, torch.uint8, "tm") if mask_time is not None else None
        fm = self._stager.to_device(mask_feature, torch.uint8, "fm") if mask_feature is not None else None
        ops.mask_frames(w["h0"], tm, fm, p16[o("wav2vec2.masked_spec_embed"):], flen, B, T, d)
        # positional conv embedding: h = h0 + gelu(conv(h0) + b)
        self._await("front_posconv")
        G, K = s.num_conv_pos_embedding_groups, s.num_conv_pos_embeddings
        Cg = d // G
        Tpad = T + K
        ops.regroup_pad(w["h0"], w["xg"], B, T, G, Cg, K // 2)
        ops.gemm(w["xg"], self.pc_wf, w["pc_pre"], C2=w["h"][0], R=w["h0"], ldr=d, M=T, N=Cg, K=K * Cg,
                 lda=Cg, ldb=K * Cg, ldc=d, bias=p32, bias_off=o("wav2vec2.encoder.pos_conv_embed.conv.bias"),
                 epilogue=EPI_GELU_RESIDUAL, batch1=B, batch2=G, sA=(G * Tpad * Cg, Tpad * Cg),
                 sB=(0, Cg * K * Cg), sC=(T * d, Cg), sR=(T * d, Cg), sBias=(0, Cg))
        # encoder layers
        drop_p = s.activation_dropout if self.training else 0.0
        scale = hd ** -0.5
        for l in range(L):
            hin, hout = w["h"][l], w["h"][l + 1]
            if not keep[l]:
                hout.copy_(hin)
                continue
            self._await(f"layer{l}")
            pl = f"wav2vec2.encoder.layers.{l}."
            ops.layernorm_fwd(hin, st.view(pl + "layer_norm.weight"), st.view(pl + "layer_norm.bias"),
                              w["x1"][l], w["st1"][l], M, d, eps)
            ops.gemm(w["x1"][l], p16, w["qkv"][l], M=M, N=3 * d, K=d, lda=d, ldb=d, ldc=3 * d,
                     b_off=o(pl + "attention.q_proj.weight"), bias=p32, bias_off=o(pl + "attention.q_proj.bias"))
            self._attention_fwd(w, l, B, T, Tp, H, hd, d, flen, scale)
            ops.gemm(w["ctx"][l], p16, w["h1"][l], M=M, N=d, K=d, lda=d, ldb=d, ldc=d,
                     b_off=o(pl + "attention.out_proj.weight"), bias=p32,
                     bias_off=o(pl + "attention.out_proj.bias"), epilogue=EPI_RESIDUAL, R=hin, ldr=d)
            ops.layernorm_fwd(w["h1"][l], st.view(pl + "final_layer_norm.weight"),
                              st.view(pl + "final_layer_norm.bias"), w["x2"][l], w["st2"][l], M, d, eps)
            ops.gemm(w["x2"][l], p16, w["u"][l], C2=w["g"][l], M=M, N=f, K=d, lda=d, ldb=d, ldc=f,
                     b_off=o(pl + "feed_forward.intermediate_dense.weight"), bias=p32,
                     bias_off=o(pl + "feed_forward.intermediate_dense.bias"), epilogue=EPI_GELU,
                     dropout_p=drop_p, dropout_seed=self.step_seed * 1000 + l, stream_out=ops.STREAM_U)
            ops.gemm(w["g"][l], p16, hout, M=M, N=d, K=f, lda=f, ldb=f, ldc=d,
                     b_off=o(pl + "feed_forward.output_dense.weight"), bias=p32,
                     bias_off=o(pl + "feed_forward.output_dense.bias"), epilogue=EPI_RESIDUAL,
                     R=w["h1"][l], ldr=d)
        # final LN + lm_head (fp32 logits, ld = Vp)
        self._await("head")
        for l in range(L):  # dropped layers were not waited for above; the backward reads every layer's weights
            if not keep[l]:
                self._await(f"layer{l}")
        ops.layernorm_fwd(w["h"][L], st.view("wav2vec2.encoder.layer_norm.weight"),
                          st.view("wav2vec2.encoder.layer_norm.bias"), w["hf"], w["stf"], M, d, eps)
        V = s.vocab_size
        ops.gemm(w["hf"], p16, w["logits"], M=M, N=V, K=d, lda=d, ldb=d, ldc=Vp,
                 b_off=o("lm_head.weight"), bias=p32, bias_off=o("lm_head.bias"))
        logits = w["logits"].view(B, T, Vp)[:, :, :V]
        out = CTCOutput(logits=logits, loss=None)
        self._saved = dict(w=w, x=x, flen=flen, keep=keep, tm=tm, fm=fm, drop_p=drop_p, B=B, N=N,
                           has_loss=False)
        if labels is not None:
            lab = self._stager.to_device(labels, torch.int32, "lab")
            Lmax = lab.shape[1]
            in_len = flen
            key = (B, T, Lmax)
            if getattr(self, "_ctc_key", None) != key:
                self._ctc_ws = torch.zeros(ops.ctc_workspace_bytes(B, T, Lmax), dtype=torch.uint8, device=dev)
                self._ctc_key = key
            gscale = None
            if s.ctc_loss_reduction == "mean":
                tl = (lab >= 0).sum(-1).clamp(min=1).to(torch.float32)
                gscale = (1.0 / (tl * B)).contiguous()
            ops.ctc_loss_fwd_bwd(w["logits"], lab, in_len, w["nll"], w["dlogits"], gscale, self._ctc_ws,
                                 B, T, V, Vp, Lmax, s.pad_token_id, s.ctc_zero_infinity)
            nll = w["nll"]
            out["nll"] = nll
            out["loss"] = nll.sum() if gscale is None else (nll * gscale).sum()
            self._saved["has_loss"] = True
        return out

    def _attention_fwd(self, w, l, B, T, Tp, H, hd, d, flen, scale):
        if self.fused_attention:
            qkv = w["qkv"][l]
            ops.attn_fwd(qkv, qkv, qkv, w["ctx"][l], w["lse"][l], B=B, H=H, Tq=T, Tk=T, hd=hd, Tqp=w["Tqp"],
                         scale=scale, ldq=3 * d, ldk=3 * d, ldv=3 * d, ldo=d, sqb=T * 3 * d, skb=T * 3 * d,
                         svb=T * 3 * d, sob=T * d, q_off=0, k_off=d, v_off=2 * d, klen=flen)
            return
        qkv, S, P, ctx = w["qkv"][l], w["S"], w["P"][l], w["ctx"][l]
        ops.gemm(qkv, qkv, S, M=T, N=T, K=hd, lda=3 * d, ldb=3 * d, ldc=Tp, b_off=d, alpha=scale,
                 batch1=B, batch2=H, sA=(T * 3 * d, hd), sB=(T * 3 * d, hd), sC=(H * T * Tp, T * Tp))
        ops.softmax_fwd(S, P, flen, B * H, H, T, T, Tp)
        ops.gemm(P, qkv, ctx, M=T, N=hd, K=T, lda=Tp, b_layout=MNMAJOR, ldb=3 * d, b_off=2 * d, ldc=d,
                 batch1=B, batch2=H, sA=(H * T * Tp, T * Tp), sB=(T * 3 * d, hd), sC=(T * d, hd))

    # ---- backward ------------------------------------------------------------------------------
    def backward(self, loss_scale: float = 1.0, overwrite_matrices: bool = False, bucket_done=None):
        """Back-propagate d(loss*loss_scale) into the flat gradient buffer (+=).

        bucket_done(name): called as soon as every gradient of a parameter bucket ("head",
        "layer{l}" in reverse order, "front") has been enqueued — the data-parallel trainer
        starts that bucket's all-reduce on its communication stream from this hook."""
        done = bucket_done if bucket_done is not None else (lambda name: None)
        sv = self._saved
        if sv is None or not sv["has_loss"]:
            raise RuntimeError("backward() needs a forward pass with labels")
        s, st = self.s, self.store
        w, x, flen, keep = sv["w"], sv["x"], sv["flen"], sv["keep"]
        B, N = sv["B"], sv["N"]
        d, f, H, hd = s.hidden_size, s.intermediate_size, s.num_attention_heads, s.head_dim
        L = s.num_hidden_layers
        Ts, T, M, Tp, Vp = w["Ts"], w["T"], w["M"], w["Tp"], w["Vp"]
        V = s.vocab_size
        p32, p16, g32 = st.p32, st.p16, st.g32
        o = st.off
        part = w["partial"]
        acc = True  # small tensors, front and head always accumulate (zero_grad clears them)
        lacc = not overwrite_matrices  # layer weight matrices: accumulate or overwrite
        # ... and, when they are overwritten (one micro-batch per optimiser step) and the trainer asked for it, kept in
        # bf16: the reference's autocast computes a Linear's weight gradient as a bf16 matmul output and only casts it to
        # fp32 when it lands in .grad.  Halves the store of the step's dominant kernel and the optimiser's gradient read.
        self.matrix_grads_bf16 = bool(overwrite_matrices and getattr(self, "wgrad_bf16", False))
        gm = st.g16 if self.matrix_grads_bf16 else g32

        # head: dlogits (fp32) -> bf16 for the MFMA path
        dl = w["dlogits"]
        if isinstance(loss_scale, torch.Tensor):  # autograd route: the incoming gradient stays on the device
            ops.wave_scale(dl, loss_scale.reshape(1).to(torch.float32), dl, 1, M * Vp)
        elif loss_scale != 1.0:
            ops.wave_scale(dl, self._scale_scalar(loss_scale), dl, 1, M * Vp)
        ops.cast_f32_bf16(dl, w["dlogits16"], M * Vp)
        d16 = w["dlogits16"]
        ops.gemm(d16, w["hf"], g32, M=V, N=d, K=M, a_layout=MNMAJOR, lda=Vp, b_layout=MNMAJOR, ldb=d,
                 ldc=d, c_off=o("lm_head.weight"), out_f32=True, accumulate=acc)
        ops.colsum(d16, Vp, M, Vp, g32, part, out_off=o("lm_head.bias"))
        if self.freeze_base:
            done("head")
            return
        ops.gemm(d16, p16, w["dA"], M=M, N=d, K=V, lda=Vp, b_layout=MNMAJOR, ldb=d, ldc=d,
                 b_off=o("lm_head.weight"))
        dh = w["dB"]
        ops.layernorm_bwd(w["dA"], w["h"][L], st.view("wav2vec2.encoder.layer_norm.weight"), None,
                          w["stf"], None, dh, st.view("wav2vec2.encoder.layer_norm.weight", "g32"),
                          st.view("wav2vec2.encoder.layer_norm.bias", "g32"), part, M, d)
        done("head")
        # dh = gradient wrt residual stream leaving layer L-1
        other = w["dA"]
        scale = hd ** -0.5
        drop_p = sv["drop_p"]
        # Weight gradients on their own stream: dW = dY^T X feeds only the optimiser, so a layer's three weight-gradient
        # launches (MFMA-bound, 256x256 tiles, one workgroup per CU) run beside the NEXT layer's data-gradient chain
        # (128x128-tile GEMMs, attention backward, LayerNorm backward: VALU / HBM-bound kernels and GEMM tails that
        # leave matrix pipes idle).  The dY operands rotate through two (three for dh) buffers; the main stream waits
        # for the weight gradients of layer i before layer i+2 reuses their buffers.
        ws = self._wgrad_stream()
        main = torch.cuda.current_stream()
        wdone = {}
        it = 0
        nb = 5 * d + f  # q|k|v, out, ffn1, ffn2 biases: contiguous in the flat buffer (w2v2_param_list)

        def on_side(ready_event, fn):
            """Run fn on the weight-gradient stream once `ready_event` (main stream) has passed."""
            if ws is None:
                fn()
                return
            ws.wait_event(ready_event)
            with torch.cuda.stream(ws):
                fn()

        def mark():
            if ws is None:
                return None
            ev = torch.cuda.Event()
            ev.record(main)
            return ev

        plan = self.norm_plan()

        def sq(l, key):  # where the weight-gradient GEMM of matrix `key` leaves its per-tile sums of squares
            return (plan["slots"], plan["slot_off"][(l, key)]) if plan is not None else None

        for l in reversed(range(L)):
            if not keep[l]:
                if overwrite_matrices:  # dropped layer: its matrices get no gradient this step
                    lo = o(f"wav2vec2.encoder.layers.{l}.attention.q_proj.weight")
                    if self.matrix_grads_bf16:
                        gm[lo:st.buckets[f"layer{l}"][1]].zero_()
                    else:
                        ops.clear_f32(g32, st.buckets[f"layer{l}"][1] - lo, off=lo)
                    if plan is not None:
                        a0 = plan["slot_off"][(l, "qkv")]
                        a1 = plan["slot_off"][(l + 1, "qkv")] if l + 1 < L else plan["nslots"]
                        ops.clear_f32(plan["slots"], a1 - a0, off=a0)
                done(f"layer{l}")
                continue
            pl = f"wav2vec2.encoder.layers.{l}."
            hin = w["h"][l]
            if ws is not None:
                if it - 2 in wdone:
                    main.wait_event(wdone.pop(it - 2))
                dh_next, du, dh1, dqkv = w["dBr"][(it + 1) % 3], w["dur"][it & 1], w["dCr"][it & 1], w["dqkvr"][it & 1]
            else:
                dh_next, du, dh1, dqkv = dh, w["du"], w["dC"], w["dqkv"]
            # FFN2: h_out = h1 + W2 g + b2
            # (bias gradients = column sums of the same dY: taken inside the weight-gradient kernel where it runs
            # on the 256x256 tiles, see ops.wgrad_gemm)
            wg = [dict(dY=dh, X=w["g"][l], M=d, N=f, K=M, lda=d, ldb=f, part=w["partial_w"],
                       c_off=o(pl + "feed_forward.output_dense.weight"), accumulate=lacc,
                       bias_off=o(pl + "feed_forward.output_dense.bias"), cs_off=4 * d + f, sq=sq(l, "fc2"))]
            ops.gemm(dh, p16, du, M=M, N=f, K=d, lda=d, b_layout=MNMAJOR, ldb=f, ldc=f,
                     b_off=o(pl + "feed_forward.output_dense.weight"), epilogue=EPI_DGELU, R=w["u"][l],
                     ldr=f, dropout_p=drop_p, dropout_seed=self.step_seed * 1000 + l)
            # FFN1
            wg.append(dict(dY=du, X=w["x2"][l], M=f, N=d, K=M, lda=f, ldb=d, part=w["partial_w"],
                           c_off=o(pl + "feed_forward.intermediate_dense.weight"), accumulate=lacc,
                           bias_off=o(pl + "feed_forward.intermediate_dense.bias"), cs_off=4 * d, sq=sq(l, "fc1")))
            ops.gemm(du, p16, other, M=M, N=d, K=f, lda=f, b_layout=MNMAJOR, ldb=d, ldc=d,
                     b_off=o(pl + "feed_forward.intermediate_dense.weight"))
            # LN2: dh1 = dh + LN'(dx2)
            # (with the side stream, the d gamma | d beta partials of the layer's two norms are reduced there: two
            # tiny dependent launches less per layer on the critical stream)
            lnp = w["ln_partial"][it & 1]
            ops.layernorm_bwd(other, w["h1"][l], st.view(pl + "final_layer_norm.weight"), None, w["st2"][l],
                              dh, dh1, None, None, lnp[0], M, d)
            # out_proj: h1 = h + Wo ctx + bo
            wg.append(dict(dY=dh1, X=w["ctx"][l], M=d, N=d, K=M, lda=d, ldb=d, part=w["partial_w"],
                           c_off=o(pl + "attention.out_proj.weight"), accumulate=lacc,
                           bias_off=o(pl + "attention.out_proj.bias"), cs_off=3 * d, sq=sq(l, "o")))
            dctx = other
            ops.gemm(dh1, p16, dctx, M=M, N=d, K=d, lda=d, b_layout=MNMAJOR, ldb=d, ldc=d,
                     b_off=o(pl + "attention.out_proj.weight"))
            self._attention_bwd(w, l, dctx, B, T, Tp, H, hd, d, scale, dqkv)
            wg.append(dict(dY=dqkv, X=w["x1"][l], M=3 * d, N=d, K=M, lda=3 * d, ldb=d, part=w["partial_w"],
                           c_off=o(pl + "attention.q_proj.weight"), accumulate=lacc,
                           bias_off=o(pl + "attention.q_proj.bias"), cs_off=0, sq=sq(l, "qkv")))

            # the layer's weight gradients (one grouped launch plan: 240 + 240 + 184 + 64 tiles at XLS-R-2B) and the
            # reduction of their fused bias-gradient partials
            bias_fused = [False]

            def wgrads(wg=wg, pl=pl, bias_fused=bias_fused):
                bias_fused[0] = ops.wgrad_gemm_group(wg, gm, colsum_ws=w["bias_ws"], colsum_ld=nb, Gb=g32)

            on_side(mark(), wgrads)
            dx1 = other
            ops.gemm(dqkv, p16, dx1, M=M, N=d, K=3 * d, lda=3 * d, b_layout=MNMAJOR, ldb=d, ldc=d,
                     b_off=o(pl + "attention.q_proj.weight"))
            # LN1: dh_in = dh1 + LN'(dx1)
            ops.layernorm_bwd(dx1, hin, st.view(pl + "layer_norm.weight"), None, w["st1"][l], dh1, dh_next,
                              None, None, lnp[1], M, d)
            dh = dh_next

            # the layer's second stages in ONE launch: the fused bias-gradient partials of the weight-gradient kernels
            # and the d gamma | d beta partials of its two norms (weight and bias gradients of a norm are adjacent in the
            # flat buffer)
            def second_stage(pl=pl, lnp=lnp, bias_fused=bias_fused):
                items = [(lnp[0], w["ln_parts"], 2 * d, 2 * d, g32[o(pl + "final_layer_norm.weight"):], True),
                         (lnp[1], w["ln_parts"], 2 * d, 2 * d, g32[o(pl + "layer_norm.weight"):], True)]
                if bias_fused[0]:
                    items.append((w["bias_ws"], ops.COLSUM_PARTS, nb, nb, g32[o(pl + "attention.q_proj.bias"):], True))
                ops.reduce_rows_multi(items)

            if ws is None:
                second_stage()
                done(f"layer{l}")
            else:
                # the bucket is complete once the side stream has passed both its own launches and the main stream's
                # LayerNorm gradients: the hook runs with the side stream current, so a trainer that makes its
                # communication / optimiser stream wait for "the current stream" waits for all of the bucket
                ev = mark()
                ws.wait_event(ev)
                with torch.cuda.stream(ws):
                    second_stage()
                    wd = torch.cuda.Event()
                    wd.record(ws)
                    wdone[it] = wd
                    done(f"layer{l}")
            it += 1
        if ws is not None:
            main.wait_stream(ws)
        # dh: gradient wrt h[0] = h0m + gelu(pc_pre)
        G, K = s.num_conv_pos_embedding_groups, s.num_conv_pos_embeddings
        Cg = d // G
        Tpad = T + K
        dpc = w["dC"]
        ops.dgelu_mul(dh, w["pc_pre"], dpc, M * d)
        ops.colsum(dpc, d, M, d, g32, part, out_off=o("wav2vec2.encoder.pos_conv_embed.conv.bias"))
        # weight gradient in GEMM layout [G][Cg][K][Cg], then through the weight norm
        ops.gemm(dpc, w["xg"], w["dwf"], M=Cg, N=K * Cg, K=M, a_layout=MNMAJOR, lda=d, b_layout=MNMAJOR,
                 ldb=Cg, b_kseg=T, b_kseg_stride=G * Tpad * Cg, ldc=K * Cg, batch2=G, sA=(0, Cg),
                 sB=(0, Tpad * Cg), sC=(0, Cg * K * Cg), out_f32=True)
        pre = "wav2vec2.encoder.pos_conv_embed.conv.parametrizations.weight."
        ops.posconv_weight_bwd(w["dwf"], st.view(pre + "original1"), st.view(pre + "original0"), self.pc_norm,
                               st.view(pre + "original1", "g32"), st.view(pre + "original0", "g32"),
                               self.pc_partial, d, Cg, K)
        # data gradient: correlation of the (left 63 / right 64) padded dpc with flipped weights
        ops.regroup_pad(dpc, w["dxg"], B, T, G, Cg, K // 2)
        dh0 = w["dA"]
        ops.gemm(w["dxg"], self.pc_wb, dh0, M=T, N=Cg, K=K * Cg, lda=Cg, ldb=K * Cg, ldc=d, a_off=Cg,
                 epilogue=EPI_RESIDUAL, R=dh, ldr=d, batch1=B, batch2=G, sA=(G * Tpad * Cg, Tpad * Cg),
                 sB=(0, Cg * K * Cg), sC=(T * d, Cg), sR=(T * d, Cg))
        # SpecAugment / padding backward: masked rows feed masked_spec_embed, then are zeroed
        if sv["tm"] is not None:
            ops.colsum(dh0, d, M, d, g32, part, rowmask=sv["tm"], out_off=o("wav2vec2.masked_spec_embed"))
        ops.mask_frames(dh0, sv["tm"], sv["fm"], self.zero_embed, flen, B, T, d)
        # feature projection
        fp = "wav2vec2.feature_projection."
        C6 = s.conv_dim[6]
        ops.colsum(dh0, d, M, d, g32, part, out_off=o(fp + "projection.bias"))
        ops.gemm(dh0, w["xln"], g32, M=d, N=C6, K=M, a_layout=MNMAJOR, lda=d, b_layout=MNMAJOR, ldb=C6,
                 ldc=C6, c_off=o(fp + "projection.weight"), out_f32=True, accumulate=acc)
        dxln = w["dy"]
        ops.gemm(dh0, p16, dxln, M=M, N=C6, K=d, lda=d, b_layout=MNMAJOR, ldb=C6, ldc=C6,
                 b_off=o(fp + "projection.weight"))
        ops.layernorm_bwd(dxln, w["a"][6], st.view(fp + "layer_norm.weight"), None, w["fp_stats"], None,
                          w["dconv"][6], st.view(fp + "layer_norm.weight", "g32"),
                          st.view(fp + "layer_norm.bias", "g32"), part, M, C6)
        # conv stack 6..1
        for i in range(6, 0, -1):
            pi = f"wav2vec2.feature_extractor.conv_layers.{i}."
            k, sd, Ci, Co = s.conv_kernel[i], s.conv_stride[i], s.conv_dim[i - 1], s.conv_dim[i]
            Lin, Ti = Ts[i - 1], Ts[i]
            dy = w["dy"]
            ops.layernorm_bwd(w["dconv"][i], w["y"][i], st.view(pi + "layer_norm.weight"),
                              st.view(pi + "layer_norm.bias"), w["cstats"][i], None, dy,
                              st.view(pi + "layer_norm.weight", "g32"), st.view(pi + "layer_norm.bias", "g32"),
                              part, B * Ti, Co, act=1)
            ops.colsum(dy, Co, B * Ti, Co, g32, part, out_off=o(pi + "conv.bias"))
            # weight gradient: one TN GEMM per utterance (batch dimension) into fp32 partials, summed
            # afterwards -- the single long-K GEMM has only Co/128 x k*Ci/128 = 48 tiles
            ops.gemm(dy, w["a"][i - 1], w["dwr_part"], M=Co, N=k * Ci, K=Ti, a_layout=MNMAJOR, lda=Co,
                     b_layout=MNMAJOR, ldb=sd * Ci, ldc=k * Ci, out_f32=True, batch2=B, sA=(0, Ti * Co),
                     sB=(0, Lin * Ci), sC=(0, Co * k * Ci))
            ops.reduce_rows(w["dwr_part"], B, Co * k * Ci, Co * k * Ci, w["dwr"])
            ops.conv_weight_grad_reorder(w["dwr"], g32, Co, Ci, k, dw_off=o(pi + "conv.weight"))
            ops.gemm(dy, self.conv_wr[i], w["dcol"], M=B * Ti, N=k * Ci, K=Co, lda=Co, b_layout=MNMAJOR,
                     ldb=k * Ci, ldc=k * Ci)
            ops.col2im_1d(w["dcol"], w["dconv"][i - 1], B, Ti, Lin, Ci, k, sd)
        p0 = "wav2vec2.feature_extractor.conv_layers.0."
        ops.conv0_bwd(x, st.view(p0 + "conv.weight"), st.view(p0 + "conv.bias"),
                      st.view(p0 + "layer_norm.weight"), st.view(p0 + "layer_norm.bias"), w["dconv"][0],
                      st.view(p0 + "conv.weight", "g32"), st.view(p0 + "conv.bias", "g32"),
                      st.view(p0 + "layer_norm.weight", "g32"), st.view(p0 + "layer_norm.bias", "g32"),
                      part, B, N, s.conv_dim[0], s.conv_kernel[0], s.conv_stride[0], s.layer_norm_eps)
        done("front")

    def _scale_scalar(self, v: float) -> torch.Tensor:
        """A device scalar holding `v` (cached per value: 1/accum, no fill kernel per step)."""
        cache = self.__dict__.setdefault("_scalars", {})
        t = cache.get(v)
        if t is None:
            if len(cache) > 64:
                cache.clear()
            t = cache[v] = torch.full((1,), float(v), dtype=torch.float32, device=self.device)
        return t

    def _wgrad_stream(self):
        """The weight gradients' stream (None = everything on the current stream; CA_WGRAD_STREAM=0)."""
        import os

        if os.environ.get("CA_WGRAD_STREAM", "1") == "0":
            return None
        if getattr(self, "_wstream", None) is None:
            self._wstream = ops.side_stream(self.device, "wgrad", int(os.environ.get("CA_WGRAD_PRIO", "0")))
        return self._wstream

    def _attention_bwd(self, w, l, dctx, B, T, Tp, H, hd, d, scale, dqkv=None):
        dqkv = w["dqkv"] if dqkv is None else dqkv
        if self.fused_attention:
            qkv = w["qkv"][l]
            ops.attn_bwd(qkv, qkv, qkv, w["ctx"][l], w["lse"][l], dctx, w["Dq"], dqkv, dqkv, dqkv, lddo=d,
                         sdob=T * d, lddq=3 * d, lddk=3 * d, lddv=3 * d, sdqb=T * 3 * d, sdkb=T * 3 * d,
                         sdvb=T * 3 * d, dq_off=0, dk_off=d, dv_off=2 * d, B=B, H=H, Tq=T, Tk=T, hd=hd,
                         Tqp=w["Tqp"], scale=scale, ldq=3 * d, ldk=3 * d, ldv=3 * d, ldo=d, sqb=T * 3 * d,
                         skb=T * 3 * d, svb=T * 3 * d, sob=T * d, q_off=0, k_off=d, v_off=2 * d, klen=w["flen"])
            return
        qkv, P, dP, dS = w["qkv"][l], w["P"][l], w["S"], w["dS"]
        bs = dict(batch1=B, batch2=H)
        # dP = dctx V^T
        ops.gemm(dctx, qkv, dP, M=T, N=T, K=hd, lda=d, ldb=3 * d, b_off=2 * d, ldc=Tp, sA=(T * d, hd),
                 sB=(T * 3 * d, hd), sC=(H * T * Tp, T * Tp), **bs)
        ops.softmax_bwd(dP, P, dS, scale, B * H, T, T, Tp)
        # dQ = dS K ; dK = dS^T Q ; dV = P^T dctx
        ops.gemm(dS, qkv, dqkv, M=T, N=hd, K=T, lda=Tp, b_layout=MNMAJOR, ldb=3 * d, b_off=d, ldc=3 * d,
                 c_off=0, sA=(H * T * Tp, T * Tp), sB=(T * 3 * d, hd), sC=(T * 3 * d, hd), **bs)
        ops.gemm(dS, qkv, dqkv, M=T, N=hd, K=T, a_layout=MNMAJOR, lda=Tp, b_layout=MNMAJOR, ldb=3 * d,
                 b_off=0, ldc=3 * d, c_off=d, sA=(H * T * Tp, T * Tp), sB=(T * 3 * d, hd),
                 sC=(T * 3 * d, hd), **bs)
        ops.gemm(P, dctx, dqkv, M=T, N=hd, K=T, a_layout=MNMAJOR, lda=Tp, b_layout=MNMAJOR, ldb=d, ldc=3 * d,
                 c_off=2 * d, sA=(H * T * Tp, T * Tp), sB=(T * d, hd), sC=(T * 3 * d, hd), **bs)

    # ---- inference helpers -----------------------------------------------------------------------
    def greedy_decode(self, logits_full: torch.Tensor | None = None, in_len=None):
        """argmax + collapse + drop blank on the logits of the last forward (ids list per row)."""
        sv = self._saved
        w = sv["w"]
        B, T, V, Vp = sv["B"], w["T"], self.s.vocab_size, w["Vp"]
        dev = self.device
        raw = torch.empty(B, T, dtype=torch.int32, device=dev)
        ids = torch.empty(B, T, dtype=torch.int32, device=dev)
        olen = torch.empty(B, dtype=torch.int32, device=dev)
        ops.ctc_greedy_decode(w["logits"], in_len, raw, ids, olen, B, T, V, Vp, self.s.pad_token_id)
        ids_c, olen_c = ids.cpu(), olen.cpu()
        return [ids_c[b, :int(olen_c[b])].tolist() for b in range(B)], raw
