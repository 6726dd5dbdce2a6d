"""MI355X-native Whisper engine behind CoRal's `model=whisper-*` keys: log-mel front end on the
GPU, encoder, teacher-forced decoder with the tied LM head + cross-entropy, and greedy generation.

Mirrors what `WhisperModelSetup` gives CoRal (R/src/coral/whisper.py:49-109) and what the ASR
pipeline calls (R/src/coral/evaluate.py:56-60 -> `model.generate`):
  WhisperFeatureExtractor            $TF/models/whisper/feature_extraction_whisper.py:135-168  -> ca_logmel
  WhisperEncoder.forward             $TF/models/whisper/modeling_whisper.py:592-646
  WhisperDecoder(.Layer).forward     :448-505, 690-795
  WhisperForConditionalGeneration    :994-1099 (shift_tokens_right, tied proj_out, CE ignore -100)
  greedy generate                    $TF/models/whisper/generation_whisper.py:383,1455,1774-1812

Round-1 scope: forward paths (inference, evaluation loss, greedy decode).  The conv stem runs as
overlapping-row GEMMs over a time-padded channels-last buffer (Conv1d k=3, p=1, stride 1 / 2), the
sinusoidal positions are added in the second conv's epilogue, layers reuse the wav2vec2 kernels
(the encoder layer is the same pre-LN block), the decoder adds causal and cross attention.
Greedy decoding re-runs the decoder over the growing prefix with the cross-attention K/V computed
once per clip (a KV-cached single-token step and the Whisper backward are the next items).
"""

from __future__ import annotations

import math
import os
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib, ops
from .ops import EPI_GELU, EPI_GELU_RESIDUAL, EPI_NONE, EPI_RESIDUAL, KMAJOR, MNMAJOR
from .wav2vec2 import ParamStore, _r8


@dataclass
class WhisperShape:
    d_model: int = 384
    encoder_layers: int = 4
    decoder_layers: int = 4
    encoder_attention_heads: int = 6
    decoder_attention_heads: int = 6
    encoder_ffn_dim: int = 1536
    decoder_ffn_dim: int = 1536
    num_mel_bins: int = 80
    vocab_size: int = 51865
    max_source_positions: int = 1500
    max_target_positions: int = 448
    pad_token_id: int = 50257
    decoder_start_token_id: int = 50258
    eos_token_id: int = 50257
    layer_norm_eps: float = 1e-5


# A decoded token's self-attention and FFN LayerNorms inside the following projection's prologue (CA_DECODE_LN_FUSED=0:
# their own launches - the A/B switch; the results are bit-identical)
LN_IN_GEMM = os.environ.get("CA_DECODE_LN_FUSED", "1") != "0"

# CoRal model keys -> architectures (R/config/model/whisper-*.yaml:3-5; public config.json values)
CORAL_WHISPER_SHAPES = {
    "whisper-xxsmall": dict(d_model=384, encoder_layers=4, decoder_layers=4, encoder_attention_heads=6,
                            decoder_attention_heads=6, encoder_ffn_dim=1536, decoder_ffn_dim=1536),
    "whisper-xsmall": dict(d_model=512, encoder_layers=6, decoder_layers=6, encoder_attention_heads=8,
                           decoder_attention_heads=8, encoder_ffn_dim=2048, decoder_ffn_dim=2048),
    "whisper-small": dict(d_model=768, encoder_layers=12, decoder_layers=12, encoder_attention_heads=12,
                          decoder_attention_heads=12, encoder_ffn_dim=3072, decoder_ffn_dim=3072),
    "whisper-medium": dict(d_model=1024, encoder_layers=24, decoder_layers=24, encoder_attention_heads=16,
                           decoder_attention_heads=16, encoder_ffn_dim=4096, decoder_ffn_dim=4096),
    "whisper-large": dict(d_model=1280, encoder_layers=32, decoder_layers=32, encoder_attention_heads=20,
                          decoder_attention_heads=20, encoder_ffn_dim=5120, decoder_ffn_dim=5120,
                          num_mel_bins=128, vocab_size=51866),
    "whisper-large-turbo": dict(d_model=1280, encoder_layers=32, decoder_layers=4, encoder_attention_heads=20,
                                decoder_attention_heads=20, encoder_ffn_dim=5120, decoder_ffn_dim=5120,
                                num_mel_bins=128, vocab_size=51866),
}

N_FFT, HOP, N_SAMPLES = 400, 160, 480_000


def mel_filter_bank(n_mels: int, n_freq: int = 201, sr: int = 16_000, fmin=0.0, fmax=8000.0) -> np.ndarray:
    """Slaney-scale, Slaney-normalised triangular filters [n_freq, n_mels]
    (`mel_filter_bank(201, n_mels, 0, 8000, 16000, "slaney", "slaney")`, $TF/audio_utils.py:638-731)."""
    def hz2mel(f):
        f = np.asarray(f, dtype=np.float64)
        return np.where(f >= 1000.0, 15.0 + np.log(np.maximum(f, 1e-10) / 1000.0) * (27.0 / np.log(6.4)), 3.0 * f / 200.0)

    def mel2hz(m):
        m = np.asarray(m, dtype=np.float64)
        return np.where(m >= 15.0, 1000.0 * np.exp((np.log(6.4) / 27.0) * (m - 15.0)), 200.0 * m / 3.0)

    hz = mel2hz(np.linspace(hz2mel(fmin), hz2mel(fmax), n_mels + 2))
    freqs = np.linspace(0, sr // 2, n_freq)
    slopes = hz[None, :] - freqs[:, None]
    d = np.diff(hz)
    fb = np.maximum(0.0, np.minimum(-slopes[:, :-2] / d[:-1], slopes[:, 2:] / d[1:]))
    return (fb * (2.0 / (hz[2:n_mels + 2] - hz[:n_mels]))[None, :]).astype(np.float32)


def whisper_param_list(s: WhisperShape):
    """(HF name, shape, bucket); q,k,v weights adjacent, and a zero `k_proj.bias` slot (HF has none)
    so the fused [3d] bias vector exists.  Names ending in `__zero` are internal and never exported."""
    d = s.d_model
    out = [("model.encoder.conv1.weight", (d, s.num_mel_bins, 3), "front"), ("model.encoder.conv1.bias", (d,), "front"),
           ("model.encoder.conv2.weight", (d, d, 3), "front"), ("model.encoder.conv2.bias", (d,), "front"),
           ("model.encoder.embed_positions.weight", (s.max_source_positions, d), "front")]

    for l in range(s.encoder_layers):
        p, b = f"model.encoder.layers.{l}.", f"enc{l}"
        a = p + "self_attn."
        # small tensors first, the weight matrices (read through the bf16 compute copy only) at the end of the bucket:
        # the part a sharded optimiser splits over the ranks (WhisperTrainEngine.shard_ranges, trainer.py zero_stage)
        out += [(p + "self_attn_layer_norm.weight", (d,), b), (p + "self_attn_layer_norm.bias", (d,), b),
                (p + "final_layer_norm.weight", (d,), b), (p + "final_layer_norm.bias", (d,), b),
                # the Linear biases are contiguous (q|k|v, out, fc1, fc2 = 5d + f floats): their gradients are partial
                # column sums out of the grouped weight-gradient launch, added in one pass (whisper_train.backward)
                (a + "q_proj.bias", (d,), b), (a + "k_proj.bias__zero", (d,), b), (a + "v_proj.bias", (d,), b),
                (a + "out_proj.bias", (d,), b), (p + "fc1.bias", (s.encoder_ffn_dim,), b), (p + "fc2.bias", (d,), b)]
        out += [(a + n + ".weight", (d, d), b) for n in ("q_proj", "k_proj", "v_proj")]
        out += [(a + "out_proj.weight", (d, d), b),
                (p + "fc1.weight", (s.encoder_ffn_dim, d), b), (p + "fc2.weight", (d, s.encoder_ffn_dim), b)]
    out += [("model.encoder.layer_norm.weight", (d,), "encf"), ("model.encoder.layer_norm.bias", (d,), "encf"),
            ("model.decoder.embed_tokens.weight", (s.vocab_size, d), "emb"),
            ("model.decoder.embed_positions.weight", (s.max_target_positions, d), "emb")]
    for l in range(s.decoder_layers):
        p, b = f"model.decoder.layers.{l}.", f"dec{l}"
        sa, ca = p + "self_attn.", p + "encoder_attn."
        # small tensors first (the three norms, then ALL Linear biases as one contiguous vector of 9d + f floats in the
        # order self q|k|v, self out, cross q, cross k|v, cross out, fc1, fc2: the fused bias gradients of the layer's
        # grouped weight-gradient launch are added in one pass), the weight matrices at the end of the bucket
        for n in ("self_attn_layer_norm", "encoder_attn_layer_norm", "final_layer_norm"):
            out += [(p + n + ".weight", (d,), b), (p + n + ".bias", (d,), b)]
        out += [(sa + "q_proj.bias", (d,), b), (sa + "k_proj.bias__zero", (d,), b), (sa + "v_proj.bias", (d,), b),
                (sa + "out_proj.bias", (d,), b),
                (ca + "q_proj.bias", (d,), b), (ca + "k_proj.bias__zero", (d,), b), (ca + "v_proj.bias", (d,), b),
                (ca + "out_proj.bias", (d,), b),
                (p + "fc1.bias", (s.decoder_ffn_dim,), b), (p + "fc2.bias", (d,), b)]
        for a in (sa, ca):
            out += [(a + n + ".weight", (d, d), b) for n in ("q_proj", "k_proj", "v_proj", "out_proj")]
        out += [(p + "fc1.weight", (s.decoder_ffn_dim, d), b), (p + "fc2.weight", (d, s.decoder_ffn_dim), b)]
    out += [("model.decoder.layer_norm.weight", (d,), "decf"), ("model.decoder.layer_norm.bias", (d,), "decf")]
    return out


def sinusoid_positions(length: int, channels: int, max_timescale: float = 10000.0) -> torch.Tensor:
    """The encoder's fixed position table ($TF/models/whisper/modeling_whisper.py:55-64): [sin | cos] of
    position x geometric timescales."""
    import math

    inc = math.log(max_timescale) / (channels // 2 - 1)
    inv = torch.exp(-inc * torch.arange(channels // 2, dtype=torch.float32))
    t = torch.arange(length, dtype=torch.float32).view(-1, 1) * inv.view(1, -1)
    return torch.cat([t.sin(), t.cos()], dim=1)


class WhisperEngine:
    """Forward paths of WhisperForConditionalGeneration as sequences of HIP kernels."""

    def __init__(self, shape: WhisperShape, device="cuda:0"):
        ops.lib()
        if not torch.cuda.is_available():
            raise ops.CoralAmdError("WhisperEngine needs a GPU: there is no CPU path")
        self.s = shape
        self.device = torch.device(device)
        assert shape.d_model % 8 == 0 and shape.num_mel_bins % 8 == 0
        self.store = ParamStore(whisper_param_list(shape), self.device)
        d = shape.d_model
        self.conv1_wr = torch.zeros(d * 3 * shape.num_mel_bins, dtype=torch.bfloat16, device=self.device)
        self.conv2_wr = torch.zeros(d * 3 * d, dtype=torch.bfloat16, device=self.device)
        self.mel_filters = torch.from_numpy(mel_filter_bank(shape.num_mel_bins)).to(self.device)
        self._enc_ws = {}
        self._dec_ws = {}

    # The trainer may still be updating parameter buckets on its optimiser stream (trainer.py); every
    # path that reads weights waits for the bucket events first.
    weights_ready: dict | None = None

    def _await(self, bucket: str):
        ev = self.weights_ready.get(bucket) if self.weights_ready else None
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)

    def _await_all(self):
        if self.weights_ready:
            for ev in self.weights_ready.values():
                torch.cuda.current_stream().wait_event(ev)

    # ---- parameters ------------------------------------------------------------------------
    def exported_names(self):
        return [n for n in self.store.names() if not n.endswith("__zero")]

    def load_state_dict(self, P: dict):
        missing = [n for n in self.exported_names() if n not in P]
        if missing:
            raise KeyError(f"missing parameters: {missing[:4]}...")
        for n in self.exported_names():
            self.store.view(n).copy_(P[n].to(self.device, torch.float32).reshape(self.store.index[n][1]))
        self.refresh_compute_weights()

    def state_dict(self):
        self._await_all()
        return {n: self.store.view(n).detach().clone() for n in self.exported_names()}

    def refresh_compute_weights(self):
        s, st = self.s, self.store
        st.refresh_bf16()
        ops.conv_weight_reorder(st.p32, self.conv1_wr, s.d_model, s.num_mel_bins, 3, w_off=st.off("model.encoder.conv1.weight"))
        ops.conv_weight_reorder(st.p32, self.conv2_wr, s.d_model, s.d_model, 3, w_off=st.off("model.encoder.conv2.weight"))

    # ---- front end ---------------------------------------------------------------------------
    def log_mel(self, waves: torch.Tensor) -> torch.Tensor:
        """waves f32 [B, 480000] (padded/truncated PCM) -> input_features f32 [B, mels, 3000] on the GPU."""
        B, N = waves.shape
        x = waves.to(self.device, torch.float32).contiguous()
        out = torch.empty(B, self.s.num_mel_bins, N // HOP, dtype=torch.float32, device=self.device)
        ws = torch.empty(ops.logmel_workspace_bytes(B), dtype=torch.uint8, device=self.device)
        ops.logmel(x, self.mel_filters, out, ws, B, N, self.s.num_mel_bins)
        return out

    # ---- shared blocks -----------------------------------------------------------------------
    def _lse(self, n):
        if getattr(self, "_lse_buf", None) is None or self._lse_buf.numel() < n:
            self._lse_buf = torch.zeros(n, dtype=torch.float32, device=self.device)
        return self._lse_buf

    def _self_attention(self, qkv, ctx, B, T, H, hd, d, causal):
        Tqp = (T + 31) // 32 * 32
        ops.attn_fwd(qkv, qkv, qkv, ctx, self._lse(B * H * Tqp), B=B, H=H, Tq=T, Tk=T, hd=hd, Tqp=Tqp,
                     scale=hd ** -0.5, ldq=3 * d, ldk=3 * d, ldv=3 * d, ldo=d, sqb=T * 3 * d, skb=T * 3 * d,
                     svb=T * 3 * d, sob=T * d, q_off=0, k_off=d, v_off=2 * d, causal=causal)

    def _ffn(self, w, h_in, h_out, p, M, d, f):
        st, p32, p16 = self.store, self.store.p32, self.store.p16
        o = st.off
        if M <= 128 and d <= 2048 and LN_IN_GEMM:
            # a decoded token: the LayerNorm runs in the projection's prologue (CaGemmDesc.a_ln_gamma, bit-identical)
            ops.gemm(h_in, p16, None, C2=w["g"], M=M, N=f, K=d, lda=d, ldb=d, ldc=f, b_off=o(p + "fc1.weight"),
                     bias=p32, bias_off=o(p + "fc1.bias"), epilogue=EPI_GELU,
                     a_ln=(st.view(p + "final_layer_norm.weight"), st.view(p + "final_layer_norm.bias"), self.s.layer_norm_eps))
        else:
            ops.layernorm_fwd(h_in, st.view(p + "final_layer_norm.weight"), st.view(p + "final_layer_norm.bias"),
                              w["x"], None, M, d, self.s.layer_norm_eps)
            ops.gemm(w["x"], p16, None, C2=w["g"], M=M, N=f, K=d, lda=d, ldb=d, ldc=f, b_off=o(p + "fc1.weight"),
                     bias=p32, bias_off=o(p + "fc1.bias"), epilogue=EPI_GELU)
        ops.gemm(w["g"], p16, h_out, M=M, N=d, K=f, lda=f, ldb=f, ldc=d, b_off=o(p + "fc2.weight"), bias=p32,
                 bias_off=o(p + "fc2.bias"), epilogue=EPI_RESIDUAL, R=h_in, ldr=d)

    # ---- encoder -----------------------------------------------------------------------------
    def _encoder_ws(self, B):
        if B in self._enc_ws:
            return self._enc_ws[B]
        s, dev = self.s, self.device
        d, f, H, T = s.d_model, s.encoder_ffn_dim, s.encoder_attention_heads, s.max_source_positions
        Tin = 2 * T
        Tp = _r8(T)
        z = lambda n, dt=torch.bfloat16: torch.zeros(n, dtype=dt, device=dev)  # noqa: E731
        w = dict(xin=z(B * (Tin + 2) * s.num_mel_bins + 64), c1=z(B * (Tin + 2) * d + 64), h=[z(B * T * d), z(B * T * d)],
                 x=z(B * T * d), qkv=z(B * T * 3 * d), ctx=z(B * T * d), g=z(B * T * f), out=z(B * T * d),
                 pos16=self.store.p16[self.store.off("model.encoder.embed_positions.weight"):])
        self._enc_ws[B] = w
        return w

    # ---- fp8 encoder weights (BASELINE.json configs[4]) ------------------------------------------------
    _fp8 = None

    def enable_fp8_encoder(self, on: bool = True):
        """Inference: keep the encoder's q|k|v and fc1 weights as OCP fp8 e4m3 (one scale per matrix) and run those
        two projections of every layer on the fp8 matrix instruction; their inputs are the LayerNorm outputs, which
        ca_layernorm_fwd_fp8 quantises per row in the pass that produces them.  (out_proj and fc2 stay bf16: their
        inputs come out of the attention kernel / the GELU epilogue, where a row is spread over many workgroups.)
        Call again after the weights change."""
        if not on:
            self._fp8 = None
            return
        s, st, dev = self.s, self.store, self.device
        d, f = s.d_model, s.encoder_ffn_dim
        p8 = torch.zeros(st.numel, dtype=torch.uint8, device=dev)
        scales = torch.zeros(2 * s.encoder_layers, dtype=torch.float32, device=dev)
        ws = torch.zeros(1, dtype=torch.float32, device=dev)
        for l in range(s.encoder_layers):
            p = f"model.encoder.layers.{l}."
            for k, (name, n) in enumerate(((p + "self_attn.q_proj.weight", 3 * d * d), (p + "fc1.weight", f * d))):
                off = st.off(name)
                ops.quantize_fp8(st.p16[off:off + n], p8[off:off + n], scales[2 * l + k:2 * l + k + 1], ws, n=n)
        self._fp8 = dict(p8=p8, scales=scales)

    def encode(self, input_features: torch.Tensor) -> torch.Tensor:
        """input_features f32 [B, mels, 3000] -> encoder states bf16 [B, 1500, d]."""
        self._await_all()
        s, st = self.s, self.store
        p32, p16, o = st.p32, st.p16, st.off
        x = input_features.to(self.device, torch.float32).contiguous()
        B, mels, Tin = x.shape
        T = s.max_source_positions
        if mels != s.num_mel_bins or Tin != 2 * T:
            raise ValueError(f"Whisper expects the mel input features to be of shape [B, {s.num_mel_bins}, {2 * T}], "
                             f"but found {tuple(x.shape)}")
        d, f, H = s.d_model, s.encoder_ffn_dim, s.encoder_attention_heads
        hd = d // H
        w = self._encoder_ws(B)
        M = B * T
        # channels-last, time-padded input: rows 1..3000 of each clip hold the frames
        for b in range(B):
            ops.transpose_f32_bf16(x[b], w["xin"][(b * (Tin + 2) + 1) * mels:], mels, Tin)
        # conv1 (k=3, p=1) + GELU -> padded [B, 3002, d]
        ops.gemm(w["xin"], self.conv1_wr, None, C2=w["c1"], c2_off=d, M=Tin, N=d, K=3 * mels, lda=mels, ldb=3 * mels,
                 ldc=d, bias=p32, bias_off=o("model.encoder.conv1.bias"), epilogue=EPI_GELU, batch2=B,
                 sA=(0, (Tin + 2) * mels), sC=(0, (Tin + 2) * d))
        # conv2 (k=3, s=2, p=1) + GELU + sinusoidal positions
        h = w["h"][0]
        ops.gemm(w["c1"], self.conv2_wr, None, C2=h, M=T, N=d, K=3 * d, lda=2 * d, ldb=3 * d, ldc=d, bias=p32,
                 bias_off=o("model.encoder.conv2.bias"), epilogue=EPI_GELU_RESIDUAL, R=w["pos16"], ldr=d, batch2=B,
                 sA=(0, (Tin + 2) * d), sC=(0, T * d), sR=(0, 0))
        cur = 0
        for l in range(s.encoder_layers):
            p = f"model.encoder.layers.{l}."
            hin, hmid = w["h"][cur], w["h"][1 - cur]
            fp8 = self._fp8
            if fp8 is not None:
                if "x8" not in w:
                    w["x8"] = torch.zeros(M * d, dtype=torch.uint8, device=self.device)
                    w["rs"] = torch.zeros(M, dtype=torch.float32, device=self.device)
                ops.layernorm_fwd_fp8(hin, st.view(p + "self_attn_layer_norm.weight"), st.view(p + "self_attn_layer_norm.bias"),
                                      None, w["x8"], w["rs"], M, d, s.layer_norm_eps)
                ops.gemm_fp8(w["x8"], fp8["p8"], w["qkv"], a_row_scale=w["rs"], b_scale=fp8["scales"][2 * l:2 * l + 1],
                             M=M, N=3 * d, K=d, lda=d, ldb=d, ldc=3 * d, b_off=o(p + "self_attn.q_proj.weight"),
                             bias=p32, bias_off=o(p + "self_attn.q_proj.bias"))
            else:
                ops.layernorm_fwd(hin, st.view(p + "self_attn_layer_norm.weight"), st.view(p + "self_attn_layer_norm.bias"),
                                  w["x"], None, M, d, s.layer_norm_eps)
                ops.gemm(w["x"], p16, w["qkv"], M=M, N=3 * d, K=d, lda=d, ldb=d, ldc=3 * d,
                         b_off=o(p + "self_attn.q_proj.weight"), bias=p32, bias_off=o(p + "self_attn.q_proj.bias"))
            self._self_attention(w["qkv"], w["ctx"], B, T, H, hd, d, causal=False)
            ops.gemm(w["ctx"], p16, hmid, M=M, N=d, K=d, lda=d, ldb=d, ldc=d, b_off=o(p + "self_attn.out_proj.weight"),
                     bias=p32, bias_off=o(p + "self_attn.out_proj.bias"), epilogue=EPI_RESIDUAL, R=hin, ldr=d)
            if fp8 is not None:
                ops.layernorm_fwd_fp8(hmid, st.view(p + "final_layer_norm.weight"), st.view(p + "final_layer_norm.bias"),
                                      None, w["x8"], w["rs"], M, d, s.layer_norm_eps)
                ops.gemm_fp8(w["x8"], fp8["p8"], None, C2=w["g"], a_row_scale=w["rs"],
                             b_scale=fp8["scales"][2 * l + 1:2 * l + 2], M=M, N=f, K=d, lda=d, ldb=d, ldc=f,
                             b_off=o(p + "fc1.weight"), bias=p32, bias_off=o(p + "fc1.bias"), epilogue=EPI_GELU)
                ops.gemm(w["g"], p16, hin, M=M, N=d, K=f, lda=f, ldb=f, ldc=d, b_off=o(p + "fc2.weight"), bias=p32,
                         bias_off=o(p + "fc2.bias"), epilogue=EPI_RESIDUAL, R=hmid, ldr=d)
            else:
                self._ffn(w, hmid, hin, p, M, d, f)  # result back in `hin`
        ops.layernorm_fwd(w["h"][cur], st.view("model.encoder.layer_norm.weight"), st.view("model.encoder.layer_norm.bias"),
                          w["out"], None, M, d, s.layer_norm_eps)
        return w["out"].view(B, T, d)

    # ---- decoder -----------------------------------------------------------------------------
    def cross_kv(self, enc: torch.Tensor) -> list[torch.Tensor]:
        """Per decoder layer: K|V projections of the encoder states, bf16 [B*1500, 2d] (computed once
        per clip, like the cross-attention cache at $TF/models/whisper/modeling_whisper.py:312-335)."""
        self._await_all()
        s, st = self.s, self.store
        d = s.d_model
        B, T, _ = enc.shape
        out = []
        for l in range(s.decoder_layers):
            p = f"model.decoder.layers.{l}.encoder_attn."
            kv = torch.empty(B * T * 2 * d, dtype=torch.bfloat16, device=self.device)
            ops.gemm(enc, st.p16, kv, M=B * T, N=2 * d, K=d, lda=d, ldb=d, ldc=2 * d, b_off=st.off(p + "k_proj.weight"),
                     bias=st.p32, bias_off=st.off(p + "k_proj.bias__zero"))
            out.append(kv)
        return out

    def _decoder_ws(self, B, L):
        key = (B, L)
        if key in self._dec_ws:
            return self._dec_ws[key]
        s, dev = self.s, self.device
        d, f, H, Te = s.d_model, s.decoder_ffn_dim, s.decoder_attention_heads, s.max_source_positions
        Lp, Tep = _r8(L), _r8(Te)
        z = lambda n, dt=torch.bfloat16: torch.zeros(n, dtype=dt, device=dev)  # noqa: E731
        w = dict(h=[z(B * L * d), z(B * L * d)], x=z(B * L * d), qkv=z(B * L * 3 * d), q=z(B * L * d), ctx=z(B * L * d),
                 g=z(B * L * f), hf=z(B * L * d))
        # one decoded token per clip: the cross-attention may deal a (clip, head)'s 1500 keys to several workgroups
        # (CaAttnDesc.split_ws; it decides by B x H against the CU count)
        w["split"] = ops.attn_split_workspace(B, H, dev) if L == 1 else None
        self._dec_ws[key] = w
        return w

    def decode(self, input_ids: torch.Tensor, enc: torch.Tensor, kv: list | None = None, last_only: bool = False):
        """Teacher-forced decoder: input_ids [B, L] -> fp32 logits [B, L, V] (or [B, 1, V] for the
        last position only)."""
        self._await_all()
        s, st = self.s, self.store
        p32, p16, o = st.p32, st.p16, st.off
        dev = self.device
        B, L = input_ids.shape
        if L > s.max_target_positions:
            raise ValueError(f"sequence length {L} cannot exceed the maximum allowed length of {s.max_target_positions} tokens")
        d, f, H, Te = s.d_model, s.decoder_ffn_dim, s.decoder_attention_heads, s.max_source_positions
        hd = d // H
        M = B * L
        Tep = _r8(Te)
        w = self._decoder_ws(B, L)
        kv = kv if kv is not None else self.cross_kv(enc)
        ids = input_ids.to(dev, torch.int32).contiguous().view(-1)
        pos = torch.arange(L, dtype=torch.int32, device=dev).repeat(B)
        ops.embed_tokens(p16[o("model.decoder.embed_tokens.weight"):], p16[o("model.decoder.embed_positions.weight"):],
                         ids, pos, w["h"][0], M, d)
        for l in range(s.decoder_layers):
            p = f"model.decoder.layers.{l}."
            h0, h1 = w["h"][0], w["h"][1]
            # causal self-attention
            ops.layernorm_fwd(h0, st.view(p + "self_attn_layer_norm.weight"), st.view(p + "self_attn_layer_norm.bias"),
                              w["x"], None, M, d, s.layer_norm_eps)
            ops.gemm(w["x"], p16, w["qkv"], M=M, N=3 * d, K=d, lda=d, ldb=d, ldc=3 * d,
                     b_off=o(p + "self_attn.q_proj.weight"), bias=p32, bias_off=o(p + "self_attn.q_proj.bias"))
            self._self_attention(w["qkv"], w["ctx"], B, L, H, hd, d, causal=True)
            ops.gemm(w["ctx"], p16, h1, M=M, N=d, K=d, lda=d, ldb=d, ldc=d, b_off=o(p + "self_attn.out_proj.weight"),
                     bias=p32, bias_off=o(p + "self_attn.out_proj.bias"), epilogue=EPI_RESIDUAL, R=h0, ldr=d)
            # cross-attention over the encoder states
            ops.layernorm_fwd(h1, st.view(p + "encoder_attn_layer_norm.weight"), st.view(p + "encoder_attn_layer_norm.bias"),
                              w["x"], None, M, d, s.layer_norm_eps)
            ops.gemm(w["x"], p16, w["q"], M=M, N=d, K=d, lda=d, ldb=d, ldc=d, b_off=o(p + "encoder_attn.q_proj.weight"),
                     bias=p32, bias_off=o(p + "encoder_attn.q_proj.bias"))
            Lqp = (L + 31) // 32 * 32
            ops.attn_fwd(w["q"], kv[l], kv[l], w["ctx"], self._lse(B * H * Lqp), B=B, H=H, Tq=L, Tk=Te, hd=hd, Tqp=Lqp,
                         scale=hd ** -0.5, ldq=d, ldk=2 * d, ldv=2 * d, ldo=d, sqb=L * d, skb=Te * 2 * d,
                         svb=Te * 2 * d, sob=L * d, k_off=0, v_off=d)
            ops.gemm(w["ctx"], p16, h0, M=M, N=d, K=d, lda=d, ldb=d, ldc=d, b_off=o(p + "encoder_attn.out_proj.weight"),
                     bias=p32, bias_off=o(p + "encoder_attn.out_proj.bias"), epilogue=EPI_RESIDUAL, R=h1, ldr=d)
            # feed-forward (result back in h1, then swap roles by copying the pointer order)
            self._ffn(w, h0, h1, p, M, d, f)
            w["h"][0], w["h"][1] = h1, h0
        ops.layernorm_fwd(w["h"][0], st.view("model.decoder.layer_norm.weight"), st.view("model.decoder.layer_norm.bias"),
                          w["hf"], None, M, d, s.layer_norm_eps)
        V = s.vocab_size
        Vp = _r8(V)
        if last_only:
            rows = w["hf"].view(B, L, d)[:, -1, :].contiguous()
            # (torch.empty: the GEMM writes all V columns, the Vp - V pad columns are never read - no fill kernel per call)
            logits = torch.empty(B, Vp, dtype=torch.float32, device=dev)
            ops.gemm(rows, p16, logits, M=B, N=V, K=d, lda=d, ldb=d, ldc=Vp, b_off=o("model.decoder.embed_tokens.weight"))
            return logits.view(B, 1, Vp)[:, :, :V]
        logits = torch.empty(M, Vp, dtype=torch.float32, device=dev)
        ops.gemm(w["hf"], p16, logits, M=M, N=V, K=d, lda=d, ldb=d, ldc=Vp, b_off=o("model.decoder.embed_tokens.weight"))
        self._last_logits = logits
        return logits.view(B, L, Vp)[:, :, :V]

    # ---- incremental decoding (self-attention K|V cache) ----------------------------------------
    def new_decode_cache(self, B: int, max_len: int):
        """Per decoder layer a bf16 [B, max_len, 2d] buffer holding K|V of the tokens decoded so far — the
        self-attention half of the cache HF keeps in `EncoderDecoderCache`
        ($TF/models/whisper/modeling_whisper.py:312-335, generation with use_cache)."""
        d = self.s.d_model
        return dict(kv=[torch.zeros(B * max_len * 2 * d, dtype=torch.bfloat16, device=self.device)
                        for _ in range(self.s.decoder_layers)], max_len=max_len, pos=0, B=B)

    def decode_step(self, new_ids: torch.Tensor, cross_kv: list, cache: dict) -> torch.Tensor:
        """Feed `new_ids` [B, n] (the forced prefix at position 0, then one token per call) through the
        decoder, appending their K|V to `cache`; returns fp32 logits [B, V] of the last position.
        Same arithmetic as `decode(...)[:, -1]`: each new query attends to all cached keys."""
        self._await_all()
        s, st = self.s, self.store
        p32, p16, o = st.p32, st.p16, st.off
        dev = self.device
        B, n = new_ids.shape
        pos0, Lmax = cache["pos"], cache["max_len"]
        if B != cache["B"] or pos0 + n > Lmax or pos0 + n > s.max_target_positions:
            raise ValueError("decode cache too small / batch mismatch")
        if pos0 > 0 and n != 1:
            raise ValueError("after the first call tokens are appended one at a time")
        d, f, H, Te = s.d_model, s.decoder_ffn_dim, s.decoder_attention_heads, s.max_source_positions
        hd = d // H
        M = B * n
        w = self._decoder_ws(B, n)
        ids = new_ids.to(dev, torch.int32).contiguous().view(-1)
        pos = (torch.arange(n, dtype=torch.int32, device=dev) + pos0).repeat(B)
        ops.embed_tokens(p16[o("model.decoder.embed_tokens.weight"):], p16[o("model.decoder.embed_positions.weight"):],
                         ids, pos, w["h"][0], M, d)
        Lk = pos0 + n
        nqp = (n + 31) // 32 * 32
        for l in range(s.decoder_layers):
            p = f"model.decoder.layers.{l}."
            h0, h1 = w["h"][0], w["h"][1]
            ckv = cache["kv"][l]
            ops.layernorm_fwd(h0, st.view(p + "self_attn_layer_norm.weight"), st.view(p + "self_attn_layer_norm.bias"),
                              w["x"], None, M, d, s.layer_norm_eps)
            ops.gemm(w["x"], p16, w["q"], M=M, N=d, K=d, lda=d, ldb=d, ldc=d, b_off=o(p + "self_attn.q_proj.weight"),
                     bias=p32, bias_off=o(p + "self_attn.q_proj.bias"))
            # K|V of the new tokens go straight into the cache rows (batch b, position pos0..)
            ops.gemm(w["x"], p16, ckv, M=n, N=2 * d, K=d, lda=d, ldb=d, ldc=2 * d, c_off=pos0 * 2 * d,
                     b_off=o(p + "self_attn.k_proj.weight"), bias=p32, bias_off=o(p + "self_attn.k_proj.bias__zero"),
                     batch2=B, sA=(0, n * d), sC=(0, Lmax * 2 * d))
            ops.attn_fwd(w["q"], ckv, ckv, w["ctx"], self._lse(B * H * nqp), B=B, H=H, Tq=n, Tk=Lk, hd=hd, Tqp=nqp,
                         scale=hd ** -0.5, ldq=d, ldk=2 * d, ldv=2 * d, ldo=d, sqb=n * d, skb=Lmax * 2 * d,
                         svb=Lmax * 2 * d, sob=n * d, k_off=0, v_off=d, causal=(n > 1))
            ops.gemm(w["ctx"], p16, h1, M=M, N=d, K=d, lda=d, ldb=d, ldc=d, b_off=o(p + "self_attn.out_proj.weight"),
                     bias=p32, bias_off=o(p + "self_attn.out_proj.bias"), epilogue=EPI_RESIDUAL, R=h0, ldr=d)
            ops.layernorm_fwd(h1, st.view(p + "encoder_attn_layer_norm.weight"), st.view(p + "encoder_attn_layer_norm.bias"),
                              w["x"], None, M, d, s.layer_norm_eps)
            ops.gemm(w["x"], p16, w["q"], M=M, N=d, K=d, lda=d, ldb=d, ldc=d, b_off=o(p + "encoder_attn.q_proj.weight"),
                     bias=p32, bias_off=o(p + "encoder_attn.q_proj.bias"))
            ops.attn_fwd(w["q"], cross_kv[l], cross_kv[l], w["ctx"], self._lse(B * H * nqp), B=B, H=H, Tq=n, Tk=Te, hd=hd,
                         Tqp=nqp, scale=hd ** -0.5, ldq=d, ldk=2 * d, ldv=2 * d, ldo=d, sqb=n * d, skb=Te * 2 * d,
                         svb=Te * 2 * d, sob=n * d, k_off=0, v_off=d)
            ops.gemm(w["ctx"], p16, h0, M=M, N=d, K=d, lda=d, ldb=d, ldc=d, b_off=o(p + "encoder_attn.out_proj.weight"),
                     bias=p32, bias_off=o(p + "encoder_attn.out_proj.bias"), epilogue=EPI_RESIDUAL, R=h1, ldr=d)
            self._ffn(w, h0, h1, p, M, d, f)
            w["h"][0], w["h"][1] = h1, h0
        ops.layernorm_fwd(w["h"][0], st.view("model.decoder.layer_norm.weight"), st.view("model.decoder.layer_norm.bias"),
                          w["hf"], None, M, d, s.layer_norm_eps)
        V, Vp = s.vocab_size, _r8(s.vocab_size)
        rows = w["hf"].view(B, n, d)[:, -1, :].contiguous()
        logits = torch.empty(B, Vp, dtype=torch.float32, device=dev)
        ops.gemm(rows, p16, logits, M=B, N=V, K=d, lda=d, ldb=d, ldc=Vp, b_off=o("model.decoder.embed_tokens.weight"))
        cache["pos"] = Lk
        return logits[:, :V]

    # ---- one-token step with static shapes and pointers (capturable in a HIP graph) -----------------
    def _graph_state(self, cache: dict, cross_kv: list, pad_id: int, eos_id: int):
        """Static buffers of the per-token step: everything that changes from token to token (position,
        cached length, token ids, finished flags) lives in device memory, so the launch sequence is
        identical for every token and can be replayed as one graph."""
        B, Lmax, dev = cache["B"], cache["max_len"], self.device
        s = self.s
        st = dict(
            tok=torch.zeros(B, dtype=torch.int32, device=dev), pos=torch.zeros(B, dtype=torch.int32, device=dev),
            klen=torch.zeros(B, dtype=torch.int32, device=dev), cur=torch.zeros(1, dtype=torch.int64, device=dev),
            done=torch.zeros(B, dtype=torch.bool, device=dev), nxt=torch.zeros(B, dtype=torch.int32, device=dev),
            out=torch.full((B, Lmax), pad_id, dtype=torch.int64, device=dev),
            logits=torch.zeros(B, _r8(s.vocab_size), dtype=torch.float32, device=dev),
            kvnew=torch.zeros(B * 2 * s.d_model, dtype=torch.bfloat16, device=dev),
            rows=torch.arange(B, device=dev), pad=torch.full((B,), pad_id, dtype=torch.int32, device=dev), pad_id=pad_id,
            eos=eos_id, cross=cross_kv)
        return st

    def _persistent_state(self, cache: dict, g: dict, suppress: torch.Tensor):
        """Descriptor + device tables of ca_whisper_decode_token for this decode state, or None where the launch
        sequence stays (shape / device outside the kernel's limits, CA_DECODE_PERSISTENT=0)."""
        if "persist" in g:
            return g["persist"]
        s, st = self.s, self.store
        B, Lmax = cache["B"], cache["max_len"]
        d, f, H, V = s.d_model, s.decoder_ffn_dim, s.decoder_attention_heads, s.vocab_size
        g["persist"] = None
        if os.environ.get("CA_DECODE_PERSISTENT", "1") == "0" or d != 64 * H or getattr(self, "_persistent_off", False):
            return None
        if not ops.whisper_decode_token_supported(B, d, f, H, V):
            return None
        p16, p32, o = st.p16.data_ptr(), st.p32.data_ptr(), st.off
        w16 = lambda n: p16 + 2 * o(n)  # noqa: E731
        w32 = lambda n: p32 + 4 * o(n)  # noqa: E731
        # the encoder K|V head-major for the launch's cross-attention: a (clip, head)'s keys / values as two contiguous
        # strips instead of 128-byte columns of 4 KB rows (one copy per generate: ~1 ms at 16 clips; CA_DECODE_CROSS_HM=0
        # keeps the [B, Te, 2d] buffers)
        Te = s.max_source_positions
        hm = os.environ.get("CA_DECODE_CROSS_HM", "1") != "0"
        cross = [c.view(B, Te, 2, H, 64).permute(0, 2, 3, 1, 4).contiguous() for c in g["cross"]] if hm else g["cross"]
        rows = []
        for l in range(s.decoder_layers):
            p = f"model.decoder.layers.{l}."
            rows.append([
                w32(p + "self_attn_layer_norm.weight"), w32(p + "self_attn_layer_norm.bias"),
                w16(p + "self_attn.q_proj.weight"), w32(p + "self_attn.q_proj.bias"),
                w16(p + "self_attn.out_proj.weight"), w32(p + "self_attn.out_proj.bias"),
                w32(p + "encoder_attn_layer_norm.weight"), w32(p + "encoder_attn_layer_norm.bias"),
                w16(p + "encoder_attn.q_proj.weight"), w32(p + "encoder_attn.q_proj.bias"),
                w16(p + "encoder_attn.out_proj.weight"), w32(p + "encoder_attn.out_proj.bias"),
                w32(p + "final_layer_norm.weight"), w32(p + "final_layer_norm.bias"),
                w16(p + "fc1.weight"), w32(p + "fc1.bias"), w16(p + "fc2.weight"), w32(p + "fc2.bias"),
                cache["kv"][l].data_ptr(), cross[l].data_ptr()])
        assert len(rows[0]) == len(_lib.CaDecodeLayer.FIELDS)
        table = torch.tensor(rows, dtype=torch.int64).to(self.device)
        nbytes = _lib.decode_ws_bytes(B, d, f, H, s.decoder_layers)
        ws = torch.zeros(nbytes, dtype=torch.uint8, device=self.device)
        status = torch.zeros(4, dtype=torch.int32, device=self.device)
        dsc = _lib.CaDecodeDesc()
        dsc.layers, dsc.n_layers, dsc.B, dsc.d, dsc.f, dsc.H = table.data_ptr(), s.decoder_layers, B, d, f, H
        dsc.Te, dsc.max_len, dsc.V = s.max_source_positions, Lmax, V
        dsc.embed, dsc.embed_pos = w16("model.decoder.embed_tokens.weight"), w16("model.decoder.embed_positions.weight")
        dsc.lnf_g, dsc.lnf_b = w32("model.decoder.layer_norm.weight"), w32("model.decoder.layer_norm.bias")
        dsc.eps = s.layer_norm_eps
        dsc.logits, dsc.ld_logits = g["logits"].data_ptr(), g["logits"].stride(0)
        dsc.suppress = suppress.data_ptr()
        dsc.out, dsc.done, dsc.ids, dsc.ld_ids = g["nxt"].data_ptr(), g["done"].data_ptr(), g["out"].data_ptr(), g["out"].stride(0)
        dsc.tok, dsc.pos, dsc.klen = g["tok"].data_ptr(), g["pos"].data_ptr(), g["klen"].data_ptr()
        dsc.pad_id, dsc.eos_id = g["pad_id"], g["eos"]
        dsc.ws, dsc.ws_bytes, dsc.status = ws.data_ptr(), nbytes, status.data_ptr()
        dsc.cross_head_major = 1 if hm else 0
        g["persist"] = dict(desc=dsc, table=table, ws=ws, status=status, suppress=suppress, cross=cross)
        return g["persist"]

    def _token_step(self, cache: dict, g: dict, suppress: torch.Tensor):
        """Decode the token in g["tok"] at position g["pos"], pick the next one (masked argmax), record it.
        Up to 16 clips: ONE persistent launch (ca_whisper_decode_token, csrc/decode.hip; bit-identical to the launch
        sequence below, which larger batches keep)."""
        ps = self._persistent_state(cache, g, suppress)
        if ps is not None:
            ops.whisper_decode_token(ps["desc"])
            return
        self._token_step_launches(cache, g, suppress)

    def _token_step_launches(self, cache: dict, g: dict, suppress: torch.Tensor):
        """The same step as a sequence of ~7 launches per layer."""
        s, st = self.s, self.store
        p32, p16, o = st.p32, st.p16, st.off
        B, Lmax = cache["B"], cache["max_len"]
        d, f, H, Te = s.d_model, s.decoder_ffn_dim, s.decoder_attention_heads, s.max_source_positions
        hd = d // H
        w = self._decoder_ws(B, 1)
        import os

        # LayerNorm + query projection inside the cross-attention launch: every (clip, head) workgroup streams its head's
        # 64 x d slice of Wq in its prologue (128 KB at d = 1024, a third of the K|V it then streams) - worth it while the
        # launch is short of workgroups (32 clips x 16 heads = 2 per CU: 3.05 against 3.14 ms per token), not above (64
        # clips: 4.17 against 4.02, 128: 7.34 against 6.96; round 5).  CA_DECODE_FUSED = 0 / 1 forces either.
        fz = os.environ.get("CA_DECODE_FUSED")
        ncu = torch.cuda.get_device_properties(self.device).multi_processor_count
        fused = (fz != "0" if fz is not None else B * H <= 2 * ncu) and hd <= 64 and d <= 2048
        ops.embed_tokens(p16[o("model.decoder.embed_tokens.weight"):], p16[o("model.decoder.embed_positions.weight"):],
                         g["tok"], g["pos"], w["h"][0], B, d)
        h0, h1 = w["h"][0], w["h"][1]
        for l in range(s.decoder_layers):
            p = f"model.decoder.layers.{l}."
            ckv = cache["kv"][l]
            ln_in = LN_IN_GEMM and B <= 128 and d <= 2048
            if not ln_in:
                ops.layernorm_fwd(h0, st.view(p + "self_attn_layer_norm.weight"), st.view(p + "self_attn_layer_norm.bias"),
                                  w["x"], None, B, d, s.layer_norm_eps)
            # q and the new K|V rows from one launch over the adjacent q|k|v weights: q to its buffer, K|V straight
            # into the cache at the device-side position (CaGemmDesc.c_split_n / c_row_index: the position is data,
            # not a launch argument, so the launch sequence can be replayed as a graph); the LayerNorm in front of it
            # in the same launch's prologue (CaGemmDesc.a_ln_gamma)
            ops.gemm(h0 if ln_in else w["x"], p16, w["q"], M=B, N=3 * d, K=d, lda=d, ldb=d, ldc=d,
                     b_off=o(p + "self_attn.q_proj.weight"),
                     bias=p32, bias_off=o(p + "self_attn.q_proj.bias"), c_split_n=d, C_hi=ckv, ldc_hi=2 * d,
                     c_row_index=g["pos"], c_row_mul=Lmax,
                     a_ln=(st.view(p + "self_attn_layer_norm.weight"), st.view(p + "self_attn_layer_norm.bias"),
                           s.layer_norm_eps) if ln_in else None)
            ops.attn_fwd(w["q"], ckv, ckv, w["ctx"], self._lse(B * H * 32), B=B, H=H, Tq=1, Tk=Lmax, hd=hd, Tqp=32,
                         scale=hd ** -0.5, ldq=d, ldk=2 * d, ldv=2 * d, ldo=d, sqb=d, skb=Lmax * 2 * d,
                         svb=Lmax * 2 * d, sob=d, k_off=0, v_off=d, klen=g["klen"])
            ops.gemm(w["ctx"], p16, h1, M=B, N=d, K=d, lda=d, ldb=d, ldc=d, b_off=o(p + "self_attn.out_proj.weight"),
                     bias=p32, bias_off=o(p + "self_attn.out_proj.bias"), epilogue=EPI_RESIDUAL, R=h0, ldr=d)
            if fused:
                # LayerNorm + query projection + attention over the cached encoder K|V: one launch (bit-identical to
                # the three below; CA_DECODE_FUSED=0 keeps them)
                ops.decode_attn_qproj(h1, st.view(p + "encoder_attn_layer_norm.weight"), st.view(p + "encoder_attn_layer_norm.bias"),
                                      p16, p32, g["cross"][l], g["cross"][l], w["ctx"], d_model=d, eps=s.layer_norm_eps,
                                      ldx=d, ldw=d, w_off=o(p + "encoder_attn.q_proj.weight"),
                                      bias_off=o(p + "encoder_attn.q_proj.bias"), B=B, H=H, Tk=Te, hd=hd,
                                      scale=hd ** -0.5, ldk=2 * d, ldv=2 * d, ldo=d, skb=Te * 2 * d, svb=Te * 2 * d,
                                      sob=d, k_off=0, v_off=d, split_ws=w["split"])
            else:
                if not ln_in:
                    ops.layernorm_fwd(h1, st.view(p + "encoder_attn_layer_norm.weight"), st.view(p + "encoder_attn_layer_norm.bias"),
                                      w["x"], None, B, d, s.layer_norm_eps)
                ops.gemm(h1 if ln_in else w["x"], p16, w["q"], M=B, N=d, K=d, lda=d, ldb=d, ldc=d,
                         b_off=o(p + "encoder_attn.q_proj.weight"), bias=p32, bias_off=o(p + "encoder_attn.q_proj.bias"),
                         a_ln=(st.view(p + "encoder_attn_layer_norm.weight"), st.view(p + "encoder_attn_layer_norm.bias"),
                               s.layer_norm_eps) if ln_in else None)
                ops.attn_fwd(w["q"], g["cross"][l], g["cross"][l], w["ctx"], self._lse(B * H * 32), B=B, H=H, Tq=1, Tk=Te,
                             hd=hd, Tqp=32, scale=hd ** -0.5, ldq=d, ldk=2 * d, ldv=2 * d, ldo=d, sqb=d, skb=Te * 2 * d,
                             svb=Te * 2 * d, sob=d, k_off=0, v_off=d, split_ws=w["split"])
            ops.gemm(w["ctx"], p16, h0, M=B, N=d, K=d, lda=d, ldb=d, ldc=d, b_off=o(p + "encoder_attn.out_proj.weight"),
                     bias=p32, bias_off=o(p + "encoder_attn.out_proj.bias"), epilogue=EPI_RESIDUAL, R=h1, ldr=d)
            self._ffn(w, h0, h1, p, B, d, f)
            h0, h1 = h1, h0
        ops.layernorm_fwd(h0, st.view("model.decoder.layer_norm.weight"), st.view("model.decoder.layer_norm.bias"),
                          w["hf"], None, B, d, s.layer_norm_eps)
        V, Vp = s.vocab_size, _r8(s.vocab_size)
        ops.gemm(w["hf"], p16, g["logits"], M=B, N=V, K=d, lda=d, ldb=d, ldc=Vp, b_off=o("model.decoder.embed_tokens.weight"))
        # argmax + the step's bookkeeping in one launch: out[b, pos + 1] = the token (pad for finished rows), done |= eos,
        # tok = the token, pos += 1, klen += 1
        ops.argmax_advance(g["logits"], suppress, g["nxt"], B, V, Vp, g["done"], g["out"], g["tok"], g["pos"], g["klen"],
                           g["pad_id"], g["eos"])

    # ---- model-level API -------------------------------------------------------------------------
    def forward(self, input_features, labels=None, decoder_input_ids=None):
        """-> dict(loss, logits): `WhisperForConditionalGeneration.forward(input_features, labels)`."""
        s = self.s
        enc = self.encode(input_features)
        if decoder_input_ids is None:
            if labels is None:
                raise ValueError("either labels or decoder_input_ids is required")
            lab = labels.to(torch.int64)
            dec = lab.new_zeros(lab.shape)
            dec[:, 1:] = lab[:, :-1]
            dec[:, 0] = s.decoder_start_token_id
            decoder_input_ids = dec.masked_fill(dec == -100, s.pad_token_id)
        logits = self.decode(decoder_input_ids, enc)
        out = dict(logits=logits, loss=None, encoder_last_hidden_state=enc)
        if labels is not None:
            B, L = labels.shape
            lab = labels.to(self.device, torch.int32).contiguous().view(-1)
            loss_sum = torch.zeros(1, dtype=torch.float32, device=self.device)
            count = torch.zeros(1, dtype=torch.int32, device=self.device)
            ops.cross_entropy_fwd_bwd(self._last_logits, lab, loss_sum, count, None, B * L, s.vocab_size,
                                      _r8(s.vocab_size), -100)
            out["loss"] = (loss_sum / count.clamp(min=1).to(torch.float32))[0]
        return out

    def generate(self, input_features, prefix: list[int], max_length: int, suppress_tokens=None,
                 begin_suppress_tokens=None, use_cache: bool = True, use_graph: bool = True) -> list[list[int]]:
        """Greedy decoding with a forced prefix (<|sot|><|da|><|transcribe|><|notimestamps|> in CoRal's
        evaluation): masked argmax on the GPU (ca_argmax_masked), stop at EOS / max_length."""
        s, dev = self.s, self.device
        enc = self.encode(input_features)
        kv = self.cross_kv(enc)
        B = enc.shape[0]
        V = s.vocab_size
        sup = torch.zeros(V, dtype=torch.uint8, device=dev)
        if suppress_tokens:
            sup[torch.tensor(list(suppress_tokens), device=dev)] = 1
        sup_begin = sup.clone()
        if begin_suppress_tokens:
            sup_begin[torch.tensor(list(begin_suppress_tokens), device=dev)] = 1
        if use_cache and use_graph and max_length > len(prefix) + 2:
            return self._generate_graph(kv, prefix, max_length, sup, sup_begin)
        ids = torch.tensor([prefix] * B, dtype=torch.int64, device=dev)
        done = torch.zeros(B, dtype=torch.bool, device=dev)
        nxt = torch.empty(B, dtype=torch.int32, device=dev)
        cache = self.new_decode_cache(B, max_length) if use_cache else None
        feed = ids
        while ids.shape[1] < max_length and not bool(done.all()):
            if use_cache:
                base = self.decode_step(feed, kv, cache).contiguous()  # fp32 [B, V]
            else:
                base = self.decode(ids, enc, kv, last_only=True)[:, 0, :].contiguous()
            mask = sup_begin if ids.shape[1] == len(prefix) else sup
            ops.argmax_masked(base, mask, nxt, B, V, V)
            step = torch.where(done, torch.full_like(nxt, s.pad_token_id), nxt).to(torch.int64)
            ids = torch.cat([ids, step[:, None]], 1)
            feed = step[:, None]
            done |= step == s.eos_token_id
        return ids.tolist()

    def _generate_graph(self, kv, prefix, max_length, sup, sup_begin):
        """Greedy loop with the per-token step captured once in a HIP graph and replayed: the ~350 small
        launches of a token (24-32 layers x 14 kernels) cost one graph launch instead of 350 host calls."""
        s, dev = self.s, self.device
        B, V, P = kv[0].shape[0] // (s.max_source_positions * 2 * s.d_model), s.vocab_size, len(prefix)
        cache = self.new_decode_cache(B, max_length)
        g = self._graph_state(cache, kv, s.pad_token_id, s.eos_token_id)
        # the forced prefix and the first free token run eagerly (different shapes / begin-suppress mask)
        ids0 = torch.tensor([prefix] * B, dtype=torch.int64, device=dev)
        base = self.decode_step(ids0, kv, cache).contiguous()
        ops.argmax_masked(base, sup_begin, g["nxt"], B, V, V)
        g["out"][:, :P] = ids0
        g["out"][:, P] = g["nxt"].long()
        g["done"] |= g["nxt"] == s.eos_token_id
        g["tok"].copy_(g["nxt"])
        g["pos"].fill_(P)
        g["klen"].fill_(P + 1)
        g["cur"].fill_(P + 1)
        n_done = P + 1
        if n_done < max_length and not bool(g["done"].all()):
            self._token_step(cache, g, sup)  # eager warm-up of the captured sequence (allocations, attributes)
            n_done += 1
        graphs = {}

        def replay(n):  # n token steps as ONE graph (capture records the launches without running them)
            if n not in graphs:
                torch.cuda.synchronize()
                graphs[n] = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graphs[n]):
                    for _ in range(n):
                        self._token_step(cache, g, sup)
            graphs[n].replay()

        # with one launch per token a graph of eight tokens is 8 kernels + 8 memset nodes: the gap between two graph
        # launches (~10-16 us) is paid once per eight tokens, like the host's all-finished check
        persist = g.get("persist") is not None
        chunk = 8 if persist else 1
        # The all-finished check.  Launch sequence: synchronous, every 8 tokens (the host waits, looks, launches).  One launch
        # per token: the check of a chunk is an asynchronous copy to pinned memory behind it, looked at one chunk LATER, so
        # the next chunk is already queued while the host waits - the device never idles between chunks (the synchronous
        # form cost ~50 us per token at 16 clips).  What the host sees late costs nothing: a launch that finds every clip
        # finished only records pad and returns (decode.hip), and the trimming below cuts those columns off.
        flags = torch.zeros(2, dtype=torch.bool).pin_memory() if persist else None
        pending = []  # (event, slot) of the chunks whose flag has not been looked at
        k = 0
        while n_done < max_length:
            if not persist and (n_done - P) % 8 == 2 and bool(g["done"].all()):  # host check every 8 tokens
                break
            n = chunk if (chunk > 1 and (n_done - P) % 8 == 2 and n_done + chunk <= max_length) else 1
            replay(n)
            n_done += n
            if persist and (n_done - P) % 8 == 2:
                flags[k & 1:(k & 1) + 1].copy_(g["done"].all().view(1), non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
                pending.append((ev, k & 1))
                k += 1
                if len(pending) == 2:  # the chunk before the one just queued
                    ev0, slot = pending.pop(0)
                    ev0.synchronize()
                    if bool(flags[slot]):
                        break
        out = g["out"][:, :n_done]
        if g.get("persist") is not None:
            code = int(g["persist"]["status"][0])  # (synchronises)
            if code != 0:
                msg = (f"ca_whisper_decode_token gave up at the seam in front of phase {code - 1}: the launch needs every "
                       "CU of the device (nothing else may run beside it)")
                if os.environ.get("CA_DECODE_STRICT") == "1":
                    raise ops.CoralAmdError(msg + "; CA_DECODE_PERSISTENT=0 keeps the launch sequence")
                # a launch that gave up has written no token: the ids above are not a generation.  This engine keeps
                # the launch sequence from here on (same bits) and decodes the batch again.
                import warnings

                warnings.warn("coral_amd: " + msg + "; decoding this batch again as a launch sequence and keeping that "
                              "path for this engine (CA_DECODE_STRICT=1 raises instead)")
                self._persistent_off = True
                return self._generate_graph(kv, prefix, max_length, sup, sup_begin)
        # trim like the eager loop: stop at the first column where every row had already finished
        fin = (out == s.eos_token_id).cumsum(1) > 0
        allfin = fin.all(0)
        keep = n_done
        if bool(allfin.any()):
            keep = int(torch.nonzero(allfin)[0]) + 1
        return out[:, :keep].tolist()
