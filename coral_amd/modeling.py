"""HF-shaped model objects over the engines: `Wav2Vec2ForCTC.from_pretrained / save_pretrained`,
`model(input_values, attention_mask, labels)` -> output with `.loss` / `.logits`
(the contract `Trainer` and the ASR pipeline rely on)."""

from __future__ import annotations

import json
import logging
from pathlib import Path

import numpy as np
import torch

from . import specaugment
from .autograd import attach_backward
from .wav2vec2 import CORAL_W2V2_SHAPES, Wav2Vec2CTCEngine, Wav2Vec2Shape

logger = logging.getLogger(__package__)

# hub ids CoRal configures (R/config/model/wav2vec2-*.yaml:3) -> architecture
HUB_SHAPES = {
    "facebook/wav2vec2-xls-r-300m": CORAL_W2V2_SHAPES["wav2vec2-small"],
    "facebook/wav2vec2-xls-r-1b": CORAL_W2V2_SHAPES["wav2vec2-medium"],
    "facebook/wav2vec2-xls-r-2b": CORAL_W2V2_SHAPES["wav2vec2-large"],
}


def _save_safetensors(tensors: dict, path: Path):
    from safetensors.torch import save_file

    save_file({k: v.detach().cpu().contiguous() for k, v in tensors.items()}, str(path), metadata={"format": "pt"})


def _load_safetensors(path: Path) -> dict:
    from safetensors.torch import load_file

    return load_file(str(path))


def load_checkpoint_tensors(model_dir: Path) -> dict:
    """The weights of an HF model directory: `model.safetensors`, sharded safetensors through
    `model.safetensors.index.json`, or the older `pytorch_model.bin` (many pretrained XLS-R repos ship only that)."""
    model_dir = Path(model_dir)
    if (model_dir / "model.safetensors").exists():
        return _load_safetensors(model_dir / "model.safetensors")
    idx = model_dir / "model.safetensors.index.json"
    if idx.exists():
        out = {}
        for shard in sorted(set(json.loads(idx.read_text())["weight_map"].values())):
            out.update(_load_safetensors(model_dir / shard))
        return out
    if (model_dir / "pytorch_model.bin").exists():
        return torch.load(model_dir / "pytorch_model.bin", map_location="cpu", weights_only=True)
    raise FileNotFoundError(f"{model_dir} holds neither model.safetensors nor pytorch_model.bin")


def _warn_unsupported_dropouts(overrides: dict, cfg: dict):
    """The reference passes five dropouts to `from_pretrained` (R/src/coral/wav2vec2.py:108-112); every CoRal model
    YAML sets all but `activation_dropout` to 0.  The engine implements activation dropout only (the FFN GEMM's
    epilogue): a non-zero value for one of the others must not pass silently."""
    for k in ("attention_dropout", "hidden_dropout", "feat_proj_dropout", "final_dropout"):
        v = float(overrides.get(k, 0.0) or 0.0)  # the checkpoint's own config values only matter in training
        if v != 0.0:
            raise NotImplementedError(f"{k}={v}: only activation_dropout is implemented on the MI355X engine "
                                      "(every CoRal wav2vec2 config sets the other dropouts to 0.0)")


class Wav2Vec2ForCTC:
    """`Wav2Vec2ForCTC` look-alike backed by `Wav2Vec2CTCEngine` (HIP kernels only)."""

    def __init__(self, shape: Wav2Vec2Shape, device=None, freeze_base=False, spec=None):
        device = device or f"cuda:{torch.cuda.current_device() if torch.cuda.is_available() else 0}"
        self.engine = Wav2Vec2CTCEngine(shape, device, freeze_base=freeze_base)
        self.shape = shape
        self.spec = spec or dict(apply_spec_augment=False, mask_time_prob=0.0, mask_time_length=10,
                                 mask_feature_prob=0.0, mask_feature_length=64)
        self.training = False
        self._rng = np.random.RandomState(4242)

    # ---- construction -----------------------------------------------------------------------
    @classmethod
    def from_pretrained(cls, name_or_path: str, device=None, freeze_base: bool = False, seed: int = 4242,
                        **overrides):
        path = Path(name_or_path)
        spec = {k: overrides.pop(k) for k in ("apply_spec_augment", "mask_time_prob", "mask_time_length",
                                              "mask_feature_prob", "mask_feature_length") if k in overrides}
        if path.is_dir() and (path / "config.json").exists():
            cfg = json.loads((path / "config.json").read_text())
            shape = Wav2Vec2Shape(
                hidden_size=cfg["hidden_size"], num_hidden_layers=cfg["num_hidden_layers"],
                num_attention_heads=cfg["num_attention_heads"], intermediate_size=cfg["intermediate_size"],
                conv_dim=tuple(cfg["conv_dim"]), conv_kernel=tuple(cfg["conv_kernel"]),
                conv_stride=tuple(cfg["conv_stride"]),
                num_conv_pos_embeddings=cfg["num_conv_pos_embeddings"],
                num_conv_pos_embedding_groups=cfg["num_conv_pos_embedding_groups"],
                vocab_size=overrides.get("vocab_size", cfg["vocab_size"]),
                pad_token_id=overrides.get("pad_token_id", cfg["pad_token_id"]),
                ctc_loss_reduction=overrides.get("ctc_loss_reduction", cfg.get("ctc_loss_reduction", "sum")),
                ctc_zero_infinity=overrides.get("ctc_zero_infinity", cfg.get("ctc_zero_infinity", True)),
                activation_dropout=overrides.get("activation_dropout", cfg.get("activation_dropout", 0.0)),
                layerdrop=overrides.get("layerdrop", cfg.get("layerdrop", 0.0)))
            model = cls(shape, device, freeze_base, spec)
            rep = model.engine.load_state_dict(load_checkpoint_tensors(path), strict=False, seed=seed)
            if rep["missing"]:
                logger.warning("%s: %s newly initialised (not in the checkpoint, or of another size)", path, rep["missing"])
            if rep["unexpected"]:
                logger.info("%s: %d checkpoint tensors not used by Wav2Vec2ForCTC (e.g. %s)", path,
                            len(rep["unexpected"]), rep["unexpected"][0])
            _warn_unsupported_dropouts(overrides, cfg)
            return model
        if name_or_path not in HUB_SHAPES:
            raise ValueError(f"unknown model {name_or_path!r}: not a local directory and not one of {list(HUB_SHAPES)}")
        shape = Wav2Vec2Shape(**HUB_SHAPES[name_or_path],
                              vocab_size=overrides.get("vocab_size", 46), pad_token_id=overrides.get("pad_token_id", 45),
                              ctc_loss_reduction=overrides.get("ctc_loss_reduction", "sum"),
                              ctc_zero_infinity=overrides.get("ctc_zero_infinity", True),
                              activation_dropout=overrides.get("activation_dropout", 0.0),
                              layerdrop=overrides.get("layerdrop", 0.0))
        _warn_unsupported_dropouts(overrides, {})
        logger.warning("no network / hub cache here: %s is instantiated with seeded random weights "
                       "(pass a local directory holding model.safetensors for real weights)", name_or_path)
        model = cls(shape, device, freeze_base, spec)
        model.init_weights(seed)
        return model

    def init_weights(self, seed: int = 4242):
        """Seeded random init generated on the device (HF-like scales)."""
        eng = self.engine
        g = torch.Generator(device=eng.device).manual_seed(seed)
        for name, (_, shp) in eng.store.index.items():
            v = eng.store.view(name)
            if name.endswith("layer_norm.weight") or name.endswith("original0"):
                v.fill_(1.0)
            elif name.endswith(".bias"):
                v.zero_()
            elif name.endswith("masked_spec_embed"):
                v.uniform_(0.0, 1.0, generator=g)
            else:
                fan_in = int(np.prod(shp[1:])) if len(shp) > 1 else shp[0]
                v.normal_(0.0, fan_in ** -0.5, generator=g)
        eng.refresh_compute_weights()

    def save_pretrained(self, model_dir):
        """HF layout: config.json + model.safetensors with HF parameter names."""
        model_dir = Path(model_dir)
        model_dir.mkdir(parents=True, exist_ok=True)
        s = self.shape
        cfg = dict(architectures=["Wav2Vec2ForCTC"], model_type="wav2vec2", hidden_size=s.hidden_size,
                   num_hidden_layers=s.num_hidden_layers, num_attention_heads=s.num_attention_heads,
                   intermediate_size=s.intermediate_size, conv_dim=list(s.conv_dim), conv_kernel=list(s.conv_kernel),
                   conv_stride=list(s.conv_stride), num_conv_pos_embeddings=s.num_conv_pos_embeddings,
                   num_conv_pos_embedding_groups=s.num_conv_pos_embedding_groups, vocab_size=s.vocab_size,
                   pad_token_id=s.pad_token_id, ctc_loss_reduction=s.ctc_loss_reduction,
                   ctc_zero_infinity=s.ctc_zero_infinity, feat_extract_norm="layer", conv_bias=True,
                   do_stable_layer_norm=True, hidden_act="gelu", feat_extract_activation="gelu",
                   activation_dropout=s.activation_dropout, layerdrop=s.layerdrop, layer_norm_eps=s.layer_norm_eps,
                   **{k: v for k, v in self.spec.items()})
        (model_dir / "config.json").write_text(json.dumps(cfg, indent=2))
        sd = self.engine.state_dict()
        if not (float(self.spec.get("mask_time_prob", 0.0)) > 0.0 or float(self.spec.get("mask_feature_prob", 0.0)) > 0.0):
            # HF only creates `masked_spec_embed` when one of the two probabilities is positive
            # ($TF/models/wav2vec2/modeling_wav2vec2.py, Wav2Vec2Model.__init__): keep the file's key set identical
            sd.pop("wav2vec2.masked_spec_embed", None)
        _save_safetensors(sd, model_dir / "model.safetensors")

    # ---- nn.Module-like surface --------------------------------------------------------------
    def train(self, mode=True):
        self.training = mode
        self.engine.train(mode)
        return self

    def eval(self):
        return self.train(False)

    def parameters(self):
        return [self.engine.store.view(n) for n in self.engine.store.names()]

    def num_parameters(self):
        return sum(int(np.prod(shp)) for _, shp in self.engine.store.index.values())

    def sample_spec_masks(self, B: int, T: int, frame_lengths):
        sp = self.spec
        if not (self.training and sp.get("apply_spec_augment", False)):
            return None, None
        mt, mf = specaugment.sample_masks(B, T, self.shape.hidden_size, frame_lengths, sp["mask_time_prob"],
                                          sp["mask_time_length"], sp["mask_feature_prob"],
                                          sp["mask_feature_length"], rng=self._rng)
        return (None if mt is None else torch.from_numpy(mt)), (None if mf is None else torch.from_numpy(mf))

    def sample_layer_keep(self):
        if not self.training or self.shape.layerdrop <= 0:
            return None
        return [bool(self._rng.rand() >= self.shape.layerdrop) for _ in range(self.shape.num_hidden_layers)]

    def __call__(self, input_values, attention_mask=None, labels=None, mask_time=None, mask_feature=None,
                 layer_keep=None, sample_lengths=None):
        """sample_lengths: optional host list of valid samples per row (what `attention_mask.sum(-1)` holds) - given by
        the device input pipeline, whose masks live on the GPU, so that drawing the SpecAugment spans needs no read-back."""
        B, N = input_values.shape
        if mask_time is None and mask_feature is None and self.training:
            T = self.engine.conv_lengths(N)[-1]
            if sample_lengths is not None:
                flen = [self.engine.conv_lengths(min(int(n), N))[-1] for n in sample_lengths]
            else:
                flen = [T] * B if attention_mask is None else self.engine.feat_lengths(attention_mask.sum(-1)).tolist()
            mask_time, mask_feature = self.sample_spec_masks(B, T, flen)
        if layer_keep is None:
            layer_keep = self.sample_layer_keep()
        out = self.engine(input_values, attention_mask, labels, mask_time, mask_feature, layer_keep)
        if out.get("loss") is not None:
            # `out.loss.backward()` runs the engine's backward (coral_amd/autograd.py; $TF/trainer.py:2005,2038)
            out["loss"] = attach_backward(self, out["loss"])
        return out

    # forwarded to engine.backward when the backward is started through autograd (`loss.backward()`)
    backward_kwargs: dict | None = None

    def backward(self, **kw):
        return self.engine.backward(**kw)
