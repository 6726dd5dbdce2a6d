"""Whisper finetuning step on the MI355X engine: forward with saved activations and the hand-written
backward of `WhisperForConditionalGeneration.forward(input_features, labels)`
($TF/models/whisper/modeling_whisper.py:994-1099: shift_tokens_right -> encoder -> decoder -> tied
proj_out -> CrossEntropyLoss(ignore_index=-100)), as driven by `Seq2SeqTrainer` in
R/src/coral/whisper.py:124-232.

Gradient flow: CE -> tied LM head (gradient joins the token-embedding gradient) -> decoder layers
(FFN, cross-attention, causal self-attention; the cross-attention K|V projections accumulate the
gradient wrt the encoder states in fp32) -> embeddings (atomic scatter-add) ; encoder states ->
encoder layers -> conv2 (+GELU, stride 2) -> conv1 (+GELU).  The sinusoidal encoder positions are
constants (requires_grad False in the reference, :570).  SpecAugment on the input features
(:821-862) takes host-drawn masks like the wav2vec2 path.
"""

from __future__ import annotations

import torch

from . import ops
from .staging import PinnedStager
from .blocks import CrossAttnBlock, FFNBlock, Scratch, SelfAttnBlock, _z
from .ops import EPI_GELU, EPI_GELU_RESIDUAL, MNMAJOR
from .wav2vec2 import _r8
from .whisper import WhisperEngine, WhisperShape


class WhisperTrainEngine(WhisperEngine):
    """Adds forward_train()/backward() to the inference engine; gradients land in `store.g32`."""

    def __init__(self, shape: WhisperShape, device="cuda:0", activation_dropout: float = 0.0,
                 freeze_base: bool = False, dropout: float = 0.0, attention_dropout: float = 0.0):
        super().__init__(shape, device)
        # `dropout`: the hidden-state dropout of WhisperConfig ($TF/models/whisper/modeling_whisper.py:398,406,479,493,
        # 502 after every sub-layer, :625,763 on the embedded inputs; R/config/model/whisper-large-turbo.yaml:12 sets
        # 0.1) - applied in the epilogue of the projection in front of each residual add, masks regenerated in the
        # backward from (step seed, site, element)
        self.dropout = dropout
        # dropout on the attention probabilities (:234), inside the fused attention kernels; only the reference's smoke
        # config test-whisper sets it (R/config/model/test-whisper.yaml:14)
        self.attention_dropout = attention_dropout
        self._stager = PinnedStager(self.device)
        self.freeze_base = freeze_base
        s, st = shape, self.store
        d, eps = s.d_model, s.layer_norm_eps
        self.activation_dropout = activation_dropout
        self.training = True
        self.step_seed = 0
        self.enc_blocks, self.dec_blocks = [], []
        for l in range(s.encoder_layers):
            p = f"model.encoder.layers.{l}."
            self.enc_blocks.append((
                SelfAttnBlock(st, p + "self_attn_layer_norm", p + "self_attn.", s.encoder_attention_heads, d, eps, False,
                              p + "self_attn.q_proj.bias"),
                FFNBlock(st, p + "final_layer_norm", p + "fc1", p + "fc2", d, s.encoder_ffn_dim, eps)))
        for l in range(s.decoder_layers):
            p = f"model.decoder.layers.{l}."
            self.dec_blocks.append((
                SelfAttnBlock(st, p + "self_attn_layer_norm", p + "self_attn.", s.decoder_attention_heads, d, eps, True,
                              p + "self_attn.q_proj.bias"),
                CrossAttnBlock(st, p + "encoder_attn_layer_norm", p + "encoder_attn.", s.decoder_attention_heads, d, eps),
                FFNBlock(st, p + "final_layer_norm", p + "fc1", p + "fc2", d, s.decoder_ffn_dim, eps)))
            # a decoder layer's bias vector (whisper_param_list): self q|k|v, self out, cross q, cross k|v, cross out, fc1, fc2
            sa, ca, ff = self.dec_blocks[-1]
            sa.cs_qkv, sa.cs_o, ca.cs, ff.cs_fc1, ff.cs_fc2 = 0, 3 * d, (4 * d, 7 * d), 8 * d, 8 * d + s.decoder_ffn_dim
        self._tw = None
        self._tw_key = None
        self.zero_mel = torch.zeros(s.num_mel_bins, dtype=torch.bfloat16, device=self.device)

    # ---- fp8 forward projections (BASELINE.json configs[4]) ------------------------------------------
    _fp8_train = None

    def enable_fp8_forward(self, on: bool = True, ffn2: bool | None = None):
        """Training: encoder forward projections on the fp8 matrix instruction (BASELINE configs[4]: "fp8 weights").
        q|k|v and fc1 take their inputs from LayerNorm kernels that quantise per row in the same pass; fc2 (`ffn2`, on by
        default; CA_FP8_FC2=0 switches it off) takes the GELU output as e4m3 straight from fc1's epilogue
        (CaGemmDesc.C8) and out_proj (CA_FP8_OUT=0 off) the attention output as e4m3 straight from the attention kernel's
        output stage (CaAttnDesc.O8), both with a DELAYED per-tensor scale - the scale of step t from the amax of step
        t - 1, the fp8 training recipe - so no quantisation pass exists anywhere in the forward.  The e4m3 weight copies are refreshed
        per bucket behind AdamW (refresh_bucket) in ONE pass each, also with delayed scales (ca_quantize_fp8_delayed; a
        weight matrix moves by ~1e-4 of its range per step), and all scales turn over in one launch per step
        (ca_fp8_amax_rotate).  Backward (CA_FP8_DGRAD=0 off): the data gradients of fc2 and out_proj - the two whose
        incoming gradient passes through an elementwise pass anyway (hidden-state dropout) - run on the fp8 instruction
        too: that pass also leaves dY as e4m3 with row scales (ca_dropout_rows_fp8), the weights have transposed e4m3
        copies (one launch per layer with the straight copies: ca_fp8_refresh_group).  fc1's data gradient is ALSO fp8 by
        default (CA_FP8_DGRAD_FC1=0 off): its incoming gradient dU leaves fc2's GELU' epilogue as e4m3 under ONE delayed
        per-tensor scale (margin 4, saturating clamp) - the numerically most fragile piece of the path, covered by the
        whole-gradient cosine test at full size (tests/test_fulldepth_gpu.py).  Everything else of the backward is bf16:
        weight gradients from the saved bf16 activations, the q|k|v data gradient from bf16 weights."""
        import os

        # the fp8 kernels leave no registers for a co-resident optimiser wave (253 of 256 per lane, two waves per SIMD):
        # the per-bucket AdamW + re-quantisation chain runs at full width there (trainer.py: ca_adamw_step_ex)
        # (holding ca_gemm_fp8_kernel_x to 224 registers spills inside its loop: 70.6 -> 100 ms per step, round 5)
        self.background_optimizer = not on
        if not on:
            self._fp8_train = None
            for sa, ff in self.enc_blocks:
                sa.fp8 = ff.fp8 = None
                sa.fp8_bwd = ff.fp8_bwd = None
            return
        st, dev, L = self.store, self.device, self.s.encoder_layers
        if ffn2 is None:
            ffn2 = os.environ.get("CA_FP8_FC2", "1") == "1"
        out8 = os.environ.get("CA_FP8_OUT", "1") == "1"
        dgrad = os.environ.get("CA_FP8_DGRAD", "1") == "1" and ffn2 and out8
        d_, f_ = self.s.d_model, self.s.encoder_ffn_dim
        nw = 4 * L  # weight tensors: q|k|v, out, fc1, fc2 per layer; then two activations per layer: GELU output, attention output
        na = 3 * L  # per layer: GELU output, attention output (forward), dU = the gradient entering fc1 (backward)
        f8 = dict(p8=torch.zeros(st.numel, dtype=torch.uint8, device=dev), nw=nw, na=na, ffn2=ffn2, out8=out8,
                  amax=torch.zeros((nw + na) * ops.FP8_AMAX_SLOTS, dtype=torch.int32, device=dev),
                  scale=torch.ones(nw + na, dtype=torch.float32, device=dev),
                  inv=torch.ones(nw + na, dtype=torch.float32, device=dev), x8=None, rs=None, g8=None, c8=None,
                  dgrad=dgrad, dy8=None, drs=None, du8=None, du_ready=[False], bwd_seen=False,
                  dgrad_fc1=dgrad and os.environ.get("CA_FP8_DGRAD_FC1", "1") == "1",
                  # transposed e4m3 copies for the data gradients: per layer fc2^T [f, d], out_proj^T [d, d], fc1^T [d, f]
                  p8t=torch.zeros(L * (2 * f_ * d_ + d_ * d_), dtype=torch.uint8, device=dev) if dgrad else None)
        # activations: a first guess (amax 8, margin 2) until the first step has measured them
        f8["scale"][nw:] = 448.0 / 16.0
        f8["inv"][nw:] = 16.0 / 448.0
        self._fp8_train = f8
        self._tw_key = None  # the workspace hands the blocks their staging buffers
        self.refresh_fp8()                                   # (scale 1: measures every matrix's amax)
        ops.fp8_amax_rotate(f8["amax"], f8["scale"], f8["inv"], nw, margin=1.0)
        self.refresh_fp8()                                   # the real copies
        ops.fp8_amax_rotate(f8["amax"], f8["scale"], f8["inv"], nw, margin=1.0)

    def _fp8_weights(self, l: int):
        s = self.s
        d, f = s.d_model, s.encoder_ffn_dim
        p = f"model.encoder.layers.{l}."
        return ((p + "self_attn.q_proj.weight", 3 * d * d), (p + "self_attn.out_proj.weight", d * d), (p + "fc1.weight", f * d),
                (p + "fc2.weight", f * d))

    def refresh_fp8(self, layer: int | None = None):
        f8 = self._fp8_train
        if f8 is None:
            return
        st = self.store
        d, f = self.s.d_model, self.s.encoder_ffn_dim
        S = ops.FP8_AMAX_SLOTS
        shapes = ((3 * d, d), (d, d), (f, d), (d, f))  # [rows, cols] of q|k|v, out_proj, fc1, fc2
        for l in (range(self.s.encoder_layers) if layer is None else (layer,)):
            # one launch per layer (ca_fp8_refresh_group): every matrix read once; straight copies with this step's
            # amax for the next scale, transposed copies (same scale) for the fp8 data gradients
            tasks = []
            base = l * (2 * f * d + d * d)
            t_off = {3: base, 1: base + f * d, 2: base + f * d + d * d}  # fc2^T [f, d] | out_proj^T [d, d] | fc1^T [d, f]
            for k, (name, n) in enumerate(self._fp8_weights(l)):
                if (k == 3 and not f8["ffn2"]) or (k == 1 and not f8["out8"]):
                    continue
                off, i = st.off(name), 4 * l + k
                tr = f8["dgrad"] and (k in (1, 3) or (k == 2 and f8["dgrad_fc1"]))
                rows, cols = shapes[k]
                tasks.append((st.p16, off, rows, cols, f8["p8"], off, f8["p8t"] if tr else None, t_off.get(k, 0),
                              f8["scale"][i:i + 1], f8["amax"][i * S:]))
            ops.fp8_refresh_group(tasks)

    def refresh_bucket(self, name: str):
        """Trainer hook: bucket `name` has just been updated (on the trainer's optimiser stream).  The first bucket of a
        step turns the amax words of the previous step into this step's scales."""
        f8 = self._fp8_train
        if f8 is None:
            return
        if name == next(iter(self.store.buckets)):
            nw, L = f8["nw"], self.s.encoder_layers
            S = ops.FP8_AMAX_SLOTS
            ops.fp8_amax_rotate(f8["amax"], f8["scale"], f8["inv"], nw, margin=1.0)
            ops.fp8_amax_rotate(f8["amax"][nw * S:], f8["scale"][nw:], f8["inv"][nw:], 2 * L, margin=2.0)  # activations
            ops.fp8_amax_rotate(f8["amax"][(nw + 2 * L) * S:], f8["scale"][nw + 2 * L:], f8["inv"][nw + 2 * L:], L, margin=4.0)  # gradients
            if f8["bwd_seen"]:
                f8["du_ready"][0] = True  # the gradients' scales now come from a measured backward
        if name.startswith("enc") and name[3:].isdigit():
            self.refresh_fp8(int(name[3:]))

    def trainable_range(self):
        """`freeze_feature_encoder` (R/src/coral/whisper.py:88-92) leaves only `proj_out` trainable, and
        proj_out is tied to the token embedding: the one matrix keeps both of its gradients."""
        if not self.freeze_base:
            return 0, self.store.numel
        lo = self.store.off("model.decoder.embed_tokens.weight")
        return lo, lo + self.s.vocab_size * self.s.d_model

    # the six weight matrices of an encoder layer are contiguous (q|k|v, out_proj, fc1, fc2): with matrices=False they
    # are neither cleared here nor read back by the first micro-batch's weight-gradient GEMMs, which overwrite them
    def _enc_matrix_range(self, l: int):
        st = self.store
        p = f"model.encoder.layers.{l}."
        lo = st.off(p + "self_attn.q_proj.weight")
        hi = st.off(p + "fc2.weight") + self.s.d_model * self.s.encoder_ffn_dim
        return lo, hi

    # ... and so are the ten of a decoder layer (self q|k|v|out, cross q|k|v|out, fc1, fc2), at the end of its bucket
    def _dec_matrix_range(self, l: int):
        st = self.store
        return st.off(f"model.decoder.layers.{l}.self_attn.q_proj.weight"), st.buckets[f"dec{l}"][1]

    def shard_ranges(self) -> dict:
        """{layer bucket: (first element of its weight matrices, bucket end)}: what a sharded optimiser (trainer.py,
        zero_stage) may split over the ranks - every encoder and decoder layer's matrices (92 % of whisper-large-turbo's
        parameters).  The forward reads them through the bf16 compute copy only; the small tensors in front of them,
        the convolutions and the embeddings stay replicated."""
        st = self.store
        out = {}
        for l in range(self.s.encoder_layers):
            lo, hi = self._enc_matrix_range(l)
            if hi == st.buckets[f"enc{l}"][1]:
                out[f"enc{l}"] = (lo, hi)
        for l in range(self.s.decoder_layers):  # (a decoder layer: norms and the bias vector, then its ten matrices)
            out[f"dec{l}"] = (st.off(f"model.decoder.layers.{l}.self_attn.q_proj.weight"), st.buckets[f"dec{l}"][1])
        return out

    def bf16_grad_ranges(self) -> dict:
        """{bucket: (lo, hi)}: the weight matrices whose gradients a single-micro-batch step may keep in bf16
        (trainer.py, ParamStore.g16) - the encoder layers' (the ones with per-tile norm partials, norm_plan)."""
        return {f"enc{l}": self._enc_matrix_range(l) for l in range(self.s.encoder_layers)}

    def _enc_matrices(self, l: int):
        d, f = self.s.d_model, self.s.encoder_ffn_dim
        p = f"model.encoder.layers.{l}."
        return [("qkv", p + "self_attn.q_proj.weight", 3 * d, d), ("o", p + "self_attn.out_proj.weight", d, d),
                ("fc1", p + "fc1.weight", f, d), ("fc2", p + "fc2.weight", d, f)]

    def zero_grad(self, matrices: bool = True):
        st, Le = self.store, self.s.encoder_layers
        if matrices or self.freeze_base or Le == 0:
            st.g32.zero_()
        else:
            # everything except the encoder AND decoder layers' weight matrices (each gets exactly one weight gradient per
            # backward, which overwrites in the step's first micro-batch: at whisper-medium the decoder's 1.6 GB were
            # cleared here and then read back by the accumulating epilogues), over a cached range table
            if getattr(self, "_small_ranges", None) is None:
                rs, pos = [], 0
                spans = [self._enc_matrix_range(l) for l in range(Le)] + [self._dec_matrix_range(l) for l in range(self.s.decoder_layers)]
                for lo, hi in sorted(spans):
                    rs.append((pos, lo - pos))
                    pos = hi
                rs.append((pos, st.numel - pos))
                self._small_ranges = tuple(r for r in rs if r[1] > 0)
            ops.clear_ranges(st.g32, self._small_ranges)
        plan = getattr(self, "_norm_plan", None)
        if plan is not None:
            ops.clear_f32(plan["slots"], plan["nslots"])

    def norm_plan(self):
        """Squared gradient norm without a pass over the encoder layers' weight matrices (wav2vec2.norm_plan)."""
        if self.freeze_base:
            return None
        if getattr(self, "_norm_plan", None) is not None:
            return self._norm_plan
        st = self.store
        off, soff, mats = 0, {}, []
        for l in range(self.s.encoder_layers):
            for key, name, M, N in self._enc_matrices(l):
                soff[(l, key)] = off
                off += ops.sumsq_slots(M, N)
                mats.append((st.off(name), M * N))
        if not mats:
            return None
        chunks, pos = [], 0
        for a, n in sorted(mats) + [(st.numel, 0)]:
            while pos < a:
                m = min(65536, a - pos)
                chunks.append((pos, m))
                pos += m
            pos = max(pos, a + n)
        self._norm_plan = dict(slots=torch.zeros(off, dtype=torch.float32, device=self.device), nslots=off, slot_off=soff,
                               chunks=torch.tensor(chunks, dtype=torch.int64, device=self.device), nchunks=len(chunks),
                               partial=torch.zeros(max(4096, len(chunks)), dtype=torch.float32, device=self.device))
        return self._norm_plan

    def train(self, mode: bool = True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    def refresh_derived(self):
        """conv stem weights are consumed in the reordered [Co][k][Ci] layout."""
        s, st = self.s, self.store
        ops.conv_weight_reorder(st.p32, self.conv1_wr, s.d_model, s.num_mel_bins, 3, w_off=st.off("model.encoder.conv1.weight"))
        ops.conv_weight_reorder(st.p32, self.conv2_wr, s.d_model, s.d_model, 3, w_off=st.off("model.encoder.conv2.weight"))

    def __call__(self, input_features, labels, mask_time=None, mask_feature=None, enc_keep=None, dec_keep=None):
        """Trainer-facing call: returns an object with `.loss` (device scalar) and `.logits`."""
        from .wav2vec2 import CTCOutput

        out = self.forward_train(input_features, labels, mask_time, mask_feature, enc_keep, dec_keep)
        return CTCOutput(loss=out["loss"], logits=out["logits"])

    def clear_internal_grads(self):
        """The `k_proj.bias__zero` slots exist only so q|k|v biases form one vector: keep them (and their
        gradients) at zero so the optimiser never moves them."""
        for n in self.store.names():
            if n.endswith("__zero"):
                self.store.view(n, "g32").zero_()

    # ---- workspace ---------------------------------------------------------------------------
    def _train_ws(self, B, L):
        key = (B, L)
        if self._tw_key == key:
            return self._tw
        s, dev = self.s, self.device
        d, T = s.d_model, s.max_source_positions
        Tin = 2 * T
        Me, Md = B * T, B * L
        f32 = torch.float32
        w = dict(
            xin=_z(B * (Tin + 2) * s.num_mel_bins + 64, dev), pre1=_z(B * (Tin + 2) * d + 64, dev),
            c1=_z(B * (Tin + 2) * d + 64, dev), pre2=_z(Me * d, dev),
            eh=[_z(Me * d, dev) for _ in range(2 * s.encoder_layers + 1)], enc_out=_z(Me * d, dev),
            enc_st=_z(Me * 2, dev, f32),
            enc_sv=[(sa.alloc(B, T, dev), ff.alloc(Me, dev)) for sa, ff in self.enc_blocks],
            dh=[_z(Md * d, dev) for _ in range(3 * s.decoder_layers + 1)], dec_out=_z(Md * d, dev),
            dec_st=_z(Md * 2, dev, f32),
            dec_sv=[(sa.alloc(B, L, dev), ca.alloc(B, L, T, dev), ff.alloc(Md, dev)) for sa, ca, ff in self.dec_blocks],
            logits=_z(Md * _r8(s.vocab_size), dev, f32), dlogits=_z(Md * _r8(s.vocab_size), dev, f32),
            dlogits16=_z(Md * _r8(s.vocab_size), dev),
            loss_cnt=_z(2, dev, f32),  # loss_sum (fp32) | count (int32) in adjacent words: cleared by one launch
            sc_e=Scratch(Me, d, s.encoder_ffn_dim, dev), sc_d=Scratch(Md, d, s.decoder_ffn_dim, dev, Mkv=Me),
            # (second scratch / bias workspace and three more gradient buffers: the encoder layers' weight gradients run
            # on a side stream two layers behind the data-gradient chain, see backward())
            sc_e2=Scratch(Me, d, s.encoder_ffn_dim, dev),
            g_e=[_z(Me * d, dev) for _ in range(6)], g_d=[_z(Md * d, dev) for _ in range(4)], dec_bias_ws=_z(ops.COLSUM_PARTS * (9 * d + s.decoder_ffn_dim), dev, f32),
            # d gamma | d beta partials of a layer's norms until the layer's one second-stage launch (encoder: two sets, its
            # weight-gradient stream runs a layer behind)
            ln_part_e=[[_z(ops.layernorm_bwd_partial_floats(Me, d), dev, f32) for _ in range(2)] for _ in range(2)],
            ln_part_d=[_z(ops.layernorm_bwd_partial_floats(Md, d), dev, f32) for _ in range(3)],
            bias_ws=_z(ops.COLSUM_PARTS * (5 * d + s.encoder_ffn_dim), dev, f32),
            bias_ws2=_z(ops.COLSUM_PARTS * (5 * d + s.encoder_ffn_dim), dev, f32), denc32=_z(Me * d, dev, f32), dpre=_z(B * (Tin + 2) * d + 64, dev),
            dcol=_z(Me * 3 * d, dev), dwr_part=_z(B * d * 3 * max(d, s.num_mel_bins), dev, f32),
            dwr=_z(d * 3 * max(d, s.num_mel_bins), dev, f32))
        f8 = self._fp8_train
        if f8 is not None:
            Me = B * s.max_source_positions
            f8["x8"] = torch.zeros(Me * s.d_model, dtype=torch.uint8, device=dev)
            f8["rs"] = torch.zeros(Me, dtype=torch.float32, device=dev)
            f8["g8"] = torch.zeros(Me * s.encoder_ffn_dim, dtype=torch.uint8, device=dev) if f8["ffn2"] else None
            f8["c8"] = torch.zeros(Me * s.d_model, dtype=torch.uint8, device=dev) if f8["out8"] else None
            f8["dy8"] = torch.zeros(Me * s.d_model, dtype=torch.uint8, device=dev) if f8["dgrad"] else None
            f8["drs"] = torch.zeros(Me, dtype=torch.float32, device=dev) if f8["dgrad"] else None
            f8["du8"] = torch.zeros(Me * s.encoder_ffn_dim, dtype=torch.uint8, device=dev) if f8["dgrad_fc1"] else None
            nw, S = f8["nw"], ops.FP8_AMAX_SLOTS

            def act(i, buf, w):  # (e4m3 activation, its scale / dequantisation factor / amax accumulator, the weight's factor)
                return (buf, f8["scale"][nw + i:nw + i + 1], f8["inv"][nw + i:nw + i + 1], f8["amax"][(nw + i) * S:], f8["inv"][w:w + 1])

            for l, (sa, ff) in enumerate(self.enc_blocks):
                sa.fp8 = (f8["p8"], f8["inv"][4 * l:4 * l + 1], f8["x8"], f8["rs"])
                ff.fp8 = (f8["p8"], f8["inv"][4 * l + 2:4 * l + 3], f8["x8"], f8["rs"])
                ff.fp8_fc2 = act(2 * l, f8["g8"], 4 * l + 3) if f8["ffn2"] else None       # fc2 <- GELU output
                sa.fp8_out = act(2 * l + 1, f8["c8"], 4 * l + 1) if f8["out8"] else None  # out_proj <- attention output
                ff.fp8_du = None
                if f8["dgrad"]:  # (transposed weights, offset, e4m3 dY, its row scales, the weight's dequantisation factor)
                    fd, dd = s.encoder_ffn_dim * s.d_model, s.d_model * s.d_model
                    base = l * (2 * fd + dd)
                    ff.fp8_bwd = (f8["p8t"], base, f8["dy8"], f8["drs"], f8["inv"][4 * l + 3:4 * l + 4])
                    sa.fp8_bwd = (f8["p8t"], base + fd, f8["dy8"], f8["drs"], f8["inv"][4 * l + 1:4 * l + 2])
                    if f8["dgrad_fc1"]:
                        i = nw + 2 * len(self.enc_blocks) + l
                        ff.fp8_du = dict(buf=f8["du8"], scale=f8["scale"][i:i + 1], inv=f8["inv"][i:i + 1], amax=f8["amax"][i * S:],
                                         inv_w=f8["inv"][4 * l + 2:4 * l + 3], w_off=base + fd + dd, ready=f8["du_ready"])
                else:
                    ff.fp8_bwd = sa.fp8_bwd = None
        self._tw, self._tw_key = w, key
        return w

    # ---- forward -----------------------------------------------------------------------------
    def forward_train(self, input_features, labels, mask_time=None, mask_feature=None, enc_keep=None, dec_keep=None):
        """-> dict(loss, logits).  labels i64 [B, L] with -100 padding.  enc_keep / dec_keep: host-drawn
        LayerDrop decisions (one bool per layer, $TF/models/whisper/modeling_whisper.py:626-634,771-779);
        a dropped layer is the identity in forward and backward."""
        s, st = self.s, self.store
        p32, p16, o = st.p32, st.p16, st.off
        dev = self.device
        x = self._stager.to_device(input_features, torch.float32, "x")  # pinned staging: no host stall (staging.py)
        B, mels, Tin = x.shape
        T, d = s.max_source_positions, s.d_model
        if mels != s.num_mel_bins or Tin != 2 * T:
            raise ValueError(f"Whisper expects mel input features of shape [B, {s.num_mel_bins}, {2 * T}]")
        lab = labels.to(torch.int64)
        L = lab.shape[1]
        if L > s.max_target_positions:
            raise ValueError(f"Labels' sequence length {L} cannot exceed the maximum allowed length of {s.max_target_positions} tokens.")
        dec_in = lab.new_zeros(lab.shape)
        dec_in[:, 1:] = lab[:, :-1]
        dec_in[:, 0] = s.decoder_start_token_id
        dec_in = dec_in.masked_fill(dec_in == -100, s.pad_token_id)
        w = self._train_ws(B, L)
        Me, Md = B * T, B * L
        drop = self.activation_dropout if self.training else 0.0
        hp = self.dropout if self.training else 0.0
        base = self.step_seed * 4096

        def hd(site):  # (p, seed) of one hidden-dropout site
            return (hp, base + site)

        apd = self.attention_dropout if self.training else 0.0

        def ad(site):  # (p, seed) of one attention-probability dropout site
            return (apd, base + site)
        mask_time_d = self._stager.to_device(mask_time, torch.uint8, "tm") if mask_time is not None else None
        mask_feature_d = self._stager.to_device(mask_feature, torch.uint8, "fm") if mask_feature is not None else None
        self._await("front")
        # encoder stem
        for b in range(B):
            ops.transpose_f32_bf16(x[b], w["xin"][(b * (Tin + 2) + 1) * mels:], mels, Tin)
            if mask_time is not None or mask_feature is not None:  # SpecAugment on the input features
                tm = mask_time_d[b:b + 1].contiguous() if mask_time_d is not None else None
                fm = mask_feature_d[b:b + 1].contiguous() if mask_feature_d is not None else None
                ops.mask_frames(w["xin"][(b * (Tin + 2) + 1) * mels:], tm, fm, self.zero_mel, None, 1, Tin, mels)
        ops.gemm(w["xin"], self.conv1_wr, w["pre1"], C2=w["c1"], c_off=d, c2_off=d, M=Tin, N=d, K=3 * mels, lda=mels,
                 ldb=3 * mels, ldc=d, bias=p32, bias_off=o("model.encoder.conv1.bias"), epilogue=EPI_GELU, batch2=B,
                 sA=(0, (Tin + 2) * mels), sC=(0, (Tin + 2) * d))
        ops.gemm(w["c1"], self.conv2_wr, w["pre2"], C2=w["eh"][0], M=T, N=d, K=3 * d, lda=2 * d, ldb=3 * d, ldc=d, bias=p32,
                 bias_off=o("model.encoder.conv2.bias"), epilogue=EPI_GELU_RESIDUAL, R=p16, r_off=o("model.encoder.embed_positions.weight"),
                 ldr=d, batch2=B, sA=(0, (Tin + 2) * d), sC=(0, T * d), sR=(0, 0))
        if hp > 0.0:
            ops.dropout(w["eh"][0], w["eh"][0], Me * d, *hd(1000))
        ek = [True] * s.encoder_layers if enc_keep is None else [bool(k) for k in enc_keep]
        dk = [True] * s.decoder_layers if dec_keep is None else [bool(k) for k in dec_keep]
        for l, (sa, ff) in enumerate(self.enc_blocks):
            self._await(f"enc{l}")
            if not ek[l]:
                w["eh"][2 * l + 2].copy_(w["eh"][2 * l])
                continue
            sv_a, sv_f = w["enc_sv"][l]
            sa.forward(w["eh"][2 * l], w["eh"][2 * l + 1], sv_a, B, T, hdrop=hd(256 + l), adrop=ad(768 + l))
            ff.forward(w["eh"][2 * l + 1], w["eh"][2 * l + 2], sv_f, Me, drop, self.step_seed * 4096 + l, hdrop=hd(512 + l))
        self._await("encf")
        self._await("emb")
        ops.layernorm_fwd(w["eh"][-1], st.view("model.encoder.layer_norm.weight"), st.view("model.encoder.layer_norm.bias"),
                          w["enc_out"], w["enc_st"], Me, d, s.layer_norm_eps)
        # decoder
        ids = self._stager.to_device(dec_in, torch.int32, "ids").view(-1)
        pos = torch.arange(L, dtype=torch.int32, device=dev).repeat(B)
        ops.embed_tokens(p16[o("model.decoder.embed_tokens.weight"):], p16[o("model.decoder.embed_positions.weight"):],
                         ids, pos, w["dh"][0], Md, d)
        if hp > 0.0:
            ops.dropout(w["dh"][0], w["dh"][0], Md * d, *hd(3000))
        for l, (sa, ca, ff) in enumerate(self.dec_blocks):
            self._await(f"dec{l}")
            if not dk[l]:
                w["dh"][3 * l + 3].copy_(w["dh"][3 * l])
                continue
            sv_a, sv_c, sv_f = w["dec_sv"][l]
            ca.project_kv(w["enc_out"], sv_c, B, T)
            sa.forward(w["dh"][3 * l], w["dh"][3 * l + 1], sv_a, B, L, hdrop=hd(2304 + l), adrop=ad(3072 + l))
            ca.forward(w["dh"][3 * l + 1], w["dh"][3 * l + 2], sv_c, B, L, T, hdrop=hd(2560 + l), adrop=ad(3328 + l))
            ff.forward(w["dh"][3 * l + 2], w["dh"][3 * l + 3], sv_f, Md, drop, self.step_seed * 4096 + 2048 + l,
                       hdrop=hd(2816 + l))
        self._await_all()  # decf and anything not waited for above
        ops.layernorm_fwd(w["dh"][-1], st.view("model.decoder.layer_norm.weight"), st.view("model.decoder.layer_norm.bias"),
                          w["dec_out"], w["dec_st"], Md, d, s.layer_norm_eps)
        V, Vp = s.vocab_size, _r8(s.vocab_size)
        ops.gemm(w["dec_out"], p16, w["logits"], M=Md, N=V, K=d, lda=d, ldb=d, ldc=Vp, b_off=o("model.decoder.embed_tokens.weight"))
        ops.clear_ranges(w["loss_cnt"], ((0, 2),))  # loss_sum | count (adjacent 4-byte words): one launch
        w["loss_sum"], w["count"] = w["loss_cnt"][0:1], w["loss_cnt"][1:2].view(torch.int32)
        lab32 = self._stager.to_device(lab, torch.int32, "lab").view(-1)
        ops.cross_entropy_fwd_bwd(w["logits"], lab32, w["loss_sum"], w["count"], w["dlogits"], Md, V, Vp, -100)
        cnt = w["count"].clamp(min=1).to(torch.float32)
        loss = (w["loss_sum"] / cnt)[0]
        self._saved = dict(w=w, B=B, L=L, ids=ids, pos=pos, inv_count=(1.0 / cnt), x=x, ek=ek, dk=dk,
                           embed_drop=(hd(1000), hd(3000)))
        return dict(loss=loss, logits=w["logits"].view(B, L, Vp)[:, :, :V])

    # ---- backward ----------------------------------------------------------------------------
    def backward(self, loss_scale: float = 1.0, overwrite_matrices: bool = False, bucket_done=None):
        """Gradients are accumulated into store.g32; `bucket_done(name)` is called per parameter bucket
        once its gradients are enqueued (decoder buckets first, then the embeddings, then the encoder)."""
        done = bucket_done if bucket_done is not None else (lambda name: None)
        sv = self._saved
        s, st = self.s, self.store
        w, B, L = sv["w"], sv["B"], sv["L"]
        p16, g32, o = st.p16, st.g32, st.off
        if self._fp8_train is not None:
            self._fp8_train["bwd_seen"] = True  # (this backward measures the amax of the gradients entering fc1)
        T, d = s.max_source_positions, s.d_model
        Tin = 2 * T
        Me, Md = B * T, B * L
        V, Vp = s.vocab_size, _r8(s.vocab_size)
        mels = s.num_mel_bins
        sc_e, sc_d = w["sc_e"], w["sc_d"]
        # mean over the non-ignored tokens, then bf16 for the MFMA path
        w["dlogits"].mul_(sv["inv_count"] * loss_scale)
        ops.cast_f32_bf16(w["dlogits"], w["dlogits16"], Md * Vp)
        dl = w["dlogits16"]
        # tied LM head: dE += dlogits^T hf ; dhf = dlogits E
        ops.gemm(dl, w["dec_out"], g32, M=V, N=d, K=Md, a_layout=MNMAJOR, lda=Vp, b_layout=MNMAJOR, ldb=d, ldc=d,
                 c_off=o("model.decoder.embed_tokens.weight"), out_f32=True, accumulate=True)
        ga, gb = w["g_d"][:2]
        ops.gemm(dl, p16, ga, M=Md, N=d, K=V, lda=Vp, b_layout=MNMAJOR, ldb=d, ldc=d, b_off=o("model.decoder.embed_tokens.weight"))
        ops.layernorm_bwd(ga, w["dh"][-1], st.view("model.decoder.layer_norm.weight"), None, w["dec_st"], None, gb,
                          st.view("model.decoder.layer_norm.weight", "g32"), st.view("model.decoder.layer_norm.bias", "g32"),
                          sc_d.part, Md, d)
        ops.clear_f32(w["denc32"], w["denc32"].numel())
        # Decoder layers.  The six token-side weight gradients of a layer (K = B*L rows: 16 ... 64 tiles of the 256x256
        # kernel each, 224 together at whisper-medium) go out as ONE grouped launch at the end of the layer, with their
        # bias gradients taken from the kernel's A stream and added in one pass (the layer's biases are one vector);
        # one by one they were split-K launches + reductions + column-sum passes: ~30 launches per layer.  Every dY of
        # the layer therefore stays alive until then: a ring of four residual-gradient buffers, the blocks' own scratch.
        ringd, r = w["g_d"], 1  # ringd[1] = gb holds the gradient wrt the last layer's output
        nbd = 9 * d + s.decoder_ffn_dim
        dacc = not overwrite_matrices  # decoder weight matrices: accumulate, or overwrite in a step's first micro-batch
        for l in reversed(range(s.decoder_layers)):
            if not sv["dk"][l]:
                if overwrite_matrices:  # dropped layer: its (uncleared) matrices get no gradient this step
                    lo, hi = self._dec_matrix_range(l)
                    g32[lo:hi].zero_()
                done(f"dec{l}")
                continue
            sa, ca, ff = self.dec_blocks[l]
            sv_a, sv_c, sv_f = w["dec_sv"][l]
            g0, g1, g2, g3 = (ringd[(r + i) % 4] for i in range(4))
            wg, second = [], []
            lpd = w["ln_part_d"]
            ff.backward(g0, g1, sv_f, sc_d, Md, defer=wg, acc=dacc, ln_part=lpd[0], pending=second)
            ca.backward(g1, g2, sv_c, sc_d, w["denc32"], B, L, T, defer=wg, cs=ca.cs, ln_part=lpd[1], pending=second, acc=dacc)
            sa.backward(g2, g3, sv_a, sc_d, B, L, defer=wg, acc=dacc, ln_part=lpd[2], pending=second)
            if ops.wgrad_gemm_group(wg, g32, colsum_ws=w["dec_bias_ws"], colsum_ld=nbd):
                second.append((w["dec_bias_ws"], ops.COLSUM_PARTS, nbd, nbd,
                               g32[o(f"model.decoder.layers.{l}.self_attn.q_proj.bias"):], True))
            ops.reduce_rows_multi(second)  # the layer's second stages (three norms + the bias vector): one launch
            r = (r + 3) % 4
            self.clear_internal_grads_of(f"model.decoder.layers.{l}.")
            done(f"dec{l}")
        cur = ringd[r]
        done("decf")
        (ep, eseed), (dp, dseed) = sv["embed_drop"]
        if dp > 0.0:  # dropout on the embedded decoder inputs (:763)
            ops.dropout(cur, cur, Md * d, dp, dseed)
        ops.embed_tokens_bwd(cur, sv["ids"], sv["pos"], g32, g32, Md, d, dtable_off=o("model.decoder.embed_tokens.weight"),
                             dpos_off=o("model.decoder.embed_positions.weight"))
        done("emb")
        # encoder
        ring = w["g_e"]
        ea = ring[1]
        ops.cast_f32_bf16(w["denc32"], ea, Me * d)
        ops.layernorm_bwd(ea, w["eh"][-1], st.view("model.encoder.layer_norm.weight"), None, w["enc_st"], None, ring[0],
                          st.view("model.encoder.layer_norm.weight", "g32"), st.view("model.encoder.layer_norm.bias", "g32"),
                          sc_e.part, Me, d)
        # Encoder layers.  Both blocks' dY stay alive until the layer's four weight gradients go out as one grouped
        # launch (192 tiles of the 256x256 kernel at d = 1024 instead of four split-K launches) - on a side stream beside
        # the next layers' data-gradient chain (as in the wav2vec2 engine; CA_WGRAD_STREAM=0: in line): layer i works in
        # ring buffers 2i, 2i+1, 2i+2 (mod 6) and scratch i & 1, so what a layer's weight gradients read is first
        # overwritten two layers later, behind an event.
        # (measured, interleaved on one box: whisper-large-turbo 89.5 -> 88.3 ms with the side stream, whisper-medium
        # 72.9 -> 73.9 without the rule below: at 12 000 rows the data-gradient GEMMs fill the chip by themselves, and a
        # layer's weight-gradient group that does not - 192 tiles of 256 x 256 at d = 1024 against 300 at d = 1280 - only
        # takes CUs away from them)
        xt = lambda m, n: ((m + 255) // 256) * ((n + 255) // 256)  # noqa: E731
        fe = s.encoder_ffn_dim
        group_tiles = xt(3 * d, d) + xt(d, d) + xt(fe, d) + xt(d, fe)
        wside = self._wgrad_stream() if group_tiles >= 256 else None
        main = torch.cuda.current_stream()
        scs, bws = (sc_e, w["sc_e2"]), (w["bias_ws"], w["bias_ws2"])
        wdone, it = {}, 0
        cur = ring[0]
        plan = self.norm_plan()
        eacc = not overwrite_matrices  # encoder weight matrices: accumulate, or overwrite in a step's first micro-batch
        # ... and then, when the trainer asked for it, kept in bf16 - the dtype the reference's autocast computes them in
        # (wav2vec2.py backward; NOTEBOOK R5.10)
        self.matrix_grads_bf16 = bool(overwrite_matrices and getattr(self, "wgrad_bf16", False) and plan is not None)
        gm = st.g16 if self.matrix_grads_bf16 else g32
        for l in reversed(range(s.encoder_layers)):
            sqd = ({k: (plan["slots"], plan["slot_off"][(l, k)]) for k in ("qkv", "o", "fc1", "fc2")}
                   if plan is not None else None)
            if not sv["ek"][l]:
                if overwrite_matrices:  # dropped layer: its (uncleared) matrices get no gradient this step
                    lo, hi = self._enc_matrix_range(l)
                    gm[lo:hi].zero_()
                done(f"enc{l}")
                continue
            sa, ff = self.enc_blocks[l]
            sv_a, sv_f = w["enc_sv"][l]
            if wside is not None and it - 2 in wdone:
                main.wait_event(wdone.pop(it - 2))
            sc, bw = scs[it & 1], bws[it & 1]
            cur, other, third = ring[(2 * it) % 6], ring[(2 * it + 1) % 6], ring[(2 * it + 2) % 6]
            wg, second = [], []
            lpe = w["ln_part_e"][it & 1]
            ff.backward(cur, other, sv_f, sc, Me, defer=wg, acc=eacc, sq=sqd, ln_part=lpe[0], pending=second)
            sa.backward(other, third, sv_a, sc, B, T, defer=wg, acc=eacc, sq=sqd, ln_part=lpe[1], pending=second)
            nb = 5 * d + s.encoder_ffn_dim

            def wgrads(wg=wg, bw=bw, l=l, second=second):
                if ops.wgrad_gemm_group(wg, gm, colsum_ws=bw, colsum_ld=nb, Gb=g32):
                    second.append((bw, ops.COLSUM_PARTS, nb, nb,
                                   g32[o(f"model.encoder.layers.{l}.self_attn.q_proj.bias"):], True))
                ops.reduce_rows_multi(second)  # both norms' d gamma | d beta and the bias vector: one launch
                self.clear_internal_grads_of(f"model.encoder.layers.{l}.")

            if wside is None:
                wgrads()
                done(f"enc{l}")
            else:
                ev = torch.cuda.Event()
                ev.record(main)
                wside.wait_event(ev)
                with torch.cuda.stream(wside):
                    wgrads()
                    wd = torch.cuda.Event()
                    wd.record(wside)
                    wdone[it] = wd
                    done(f"enc{l}")  # (hook runs with the side stream current: the bucket is complete behind it)
            cur = third
            it += 1
        if wside is not None:
            main.wait_stream(wside)
        other = ring[(2 * it + 1) % 6]
        done("encf")
        # conv2: h0 = dropout(gelu(pre2) + pos)
        if ep > 0.0:
            ops.dropout(cur, cur, Me * d, ep, eseed)
        dpre2 = other
        ops.dgelu_mul(cur, w["pre2"], dpre2, Me * d)
        ops.colsum(dpre2, d, Me, d, g32, sc_e.part, out_off=o("model.encoder.conv2.bias"))
        ops.gemm(dpre2, w["c1"], w["dwr_part"], M=d, N=3 * d, K=T, a_layout=MNMAJOR, lda=d, b_layout=MNMAJOR, ldb=2 * d,
                 ldc=3 * d, out_f32=True, batch2=B, sA=(0, T * d), sB=(0, (Tin + 2) * d), sC=(0, d * 3 * d))
        ops.reduce_rows(w["dwr_part"], B, d * 3 * d, d * 3 * d, w["dwr"])
        ops.conv_weight_grad_reorder(w["dwr"], g32, d, d, 3, dw_off=o("model.encoder.conv2.weight"))
        ops.gemm(dpre2, self.conv2_wr, w["dcol"], M=Me, N=3 * d, K=d, lda=d, b_layout=MNMAJOR, ldb=3 * d, ldc=3 * d)
        ops.col2im_1d(w["dcol"], w["dpre"], B, T, Tin + 2, d, 3, 2)  # gradient wrt the padded gelu(pre1)
        # conv1: rows 1..3000 of every clip (the padding rows carry no parameter gradient)
        ops.dgelu_mul(w["dpre"], w["pre1"], w["dpre"], B * (Tin + 2) * d)
        part_n = ops.colsum_partial_floats(Tin, d)
        for b in range(B):
            ops.colsum(w["dpre"], d, Tin, d, g32, sc_e.part[:part_n], x_off=(b * (Tin + 2) + 1) * d,
                       out_off=o("model.encoder.conv1.bias"))
        ops.gemm(w["dpre"], w["xin"], w["dwr_part"], M=d, N=3 * mels, K=Tin, a_layout=MNMAJOR, lda=d, a_off=d,
                 b_layout=MNMAJOR, ldb=mels, ldc=3 * mels, out_f32=True, batch2=B, sA=(0, (Tin + 2) * d),
                 sB=(0, (Tin + 2) * mels), sC=(0, d * 3 * mels))
        ops.reduce_rows(w["dwr_part"], B, d * 3 * mels, d * 3 * mels, w["dwr"])
        ops.conv_weight_grad_reorder(w["dwr"], g32, d, mels, 3, dw_off=o("model.encoder.conv1.weight"))
        done("front")

    def _wgrad_stream(self):
        """The encoder weight gradients' stream (None = everything on the current stream; CA_WGRAD_STREAM=0)."""
        import os

        if os.environ.get("CA_WGRAD_STREAM", "1") == "0":
            return None
        if getattr(self, "_wstream", None) is None:
            self._wstream = ops.side_stream(self.device, "wgrad", int(os.environ.get("CA_WGRAD_PRIO", "0")))
        return self._wstream

    def clear_internal_grads_of(self, prefix: str):
        """The `__zero` slots of one layer, as one launch over a cached range table (called per layer in the backward,
        before the layer's bucket is handed to the trainer's hook)."""
        cache = self.__dict__.setdefault("_zero_slot_ranges", {})
        rs = cache.get(prefix)
        if rs is None:
            st = self.store
            rs = cache[prefix] = tuple((st.off(n), int(st.view(n).numel())) for n in st.names()
                                       if n.startswith(prefix) and n.endswith("__zero"))
        if rs:
            ops.clear_ranges(self.store.g32, rs)

    def grad_dict(self):
        return {n: self.store.view(n, "g32") for n in self.exported_names()}
