"""`WhisperModelSetup` — mirror of R/src/coral/whisper.py:30-267 over `WhisperEngine`.

The byte-level BPE vocabulary of Whisper ships with the checkpoints on the HuggingFace hub, which
is unreachable here, so `load_processor` builds the feature-extraction half (GPU log-mel) and loads
the tokenizer files only from a local model directory (`tokenizers` library, `tokenizer.json`);
token-id level training / evaluation (`labels` already encoded) works without it."""

from __future__ import annotations

import json
import logging
import os
from pathlib import Path

import numpy as np
import torch

from .model_setup import ModelSetup, PreTrainedModelData, _training_args
from .coral_trainer import CoralTrainer
from . import specaugment
from .autograd import attach_backward
from .whisper import CORAL_WHISPER_SHAPES, N_SAMPLES, WhisperEngine, WhisperShape, sinusoid_positions
from .whisper_train import WhisperTrainEngine

logger = logging.getLogger(__package__)

HUB_SHAPES = {"openai/whisper-tiny": "whisper-xxsmall", "openai/whisper-base": "whisper-xsmall",
              "openai/whisper-small": "whisper-small", "openai/whisper-medium": "whisper-medium",
              "openai/whisper-large-v3": "whisper-large", "openai/whisper-large-v3-turbo": "whisper-large-turbo"}

# <|startoftranscript|><|da|><|transcribe|><|notimestamps|> in the multilingual vocabulary
# (language="danish", task="transcribe": R/src/coral/evaluate.py:59, R/src/coral/whisper.py:51-55)
DANISH_TRANSCRIBE_PREFIX = [50258, 50285, 50359, 50363]
DANISH_TRANSCRIBE_PREFIX_V3 = [50258, 50285, 50360, 50364]  # large-v3 vocabulary (51866 entries)


def prefix_ids(shape: WhisperShape):
    return DANISH_TRANSCRIBE_PREFIX_V3 if shape.vocab_size == 51866 else DANISH_TRANSCRIBE_PREFIX


class WhisperFeatureExtractorGPU:
    """pad/truncate to 30 s on the host, log-mel on the GPU (ca_logmel)."""

    def __init__(self, engine: WhisperEngine, sampling_rate: int = 16_000):
        self.engine = engine
        self.sampling_rate = sampling_rate

    def __call__(self, audios, sampling_rate: int | None = None) -> torch.Tensor:
        if sampling_rate is not None and sampling_rate != self.sampling_rate:
            raise ValueError(f"expected {self.sampling_rate} Hz audio, got {sampling_rate}")
        if isinstance(audios, np.ndarray) and audios.ndim == 1:
            audios = [audios]
        batch = np.zeros((len(audios), N_SAMPLES), dtype=np.float32)
        for i, a in enumerate(audios):
            a = np.asarray(a, dtype=np.float32)[:N_SAMPLES]
            batch[i, : len(a)] = a
        return self.engine.log_mel(torch.from_numpy(batch))


class WhisperProcessor:
    def __init__(self, feature_extractor, tokenizer=None):
        self.feature_extractor = feature_extractor
        self.tokenizer = tokenizer

    def batch_decode(self, ids, skip_special_tokens=True):
        if self.tokenizer is None:
            # no byte-level BPE files offline: render every text token as the word "t<id>", so WER on
            # the rendered strings is the token error rate (special tokens >= 50257 are skipped)
            return [" ".join(f"t{int(i)}" for i in r if not (skip_special_tokens and int(i) >= 50257)) for r in ids]
        return self.tokenizer.decode_batch([list(map(int, r)) for r in ids], skip_special_tokens=skip_special_tokens)

    def save_pretrained(self, model_dir):
        model_dir = Path(model_dir)
        model_dir.mkdir(parents=True, exist_ok=True)
        s = self.feature_extractor.engine.s
        (model_dir / "preprocessor_config.json").write_text(json.dumps(dict(
            feature_extractor_type="WhisperFeatureExtractor", feature_size=s.num_mel_bins, sampling_rate=self.feature_extractor.sampling_rate,
            hop_length=160, chunk_length=30, n_fft=400, padding_value=0.0, return_attention_mask=False), indent=2))
        if self.tokenizer is not None:
            self.tokenizer.save(str(model_dir / "tokenizer.json"))


class WhisperForConditionalGeneration:
    """HF-shaped wrapper: `model(input_features, labels)` and `model.generate(...)`.

    In training mode the call draws SpecAugment masks on the input features
    ($TF/models/whisper/modeling_whisper.py:821-862: time spans over the 3000 frames, then feature
    spans over the mel bins, zero fill) and LayerDrop decisions (:626-634, :771-779) on the host, in
    the reference's order, and runs the training forward (activations saved for `backward`)."""

    def __init__(self, shape: WhisperShape, device=None, activation_dropout: float = 0.0, freeze_base: bool = False,
                 spec=None, layerdrop: float = 0.0, dropout: float = 0.0, attention_dropout: float = 0.0):
        device = device or f"cuda:{torch.cuda.current_device() if torch.cuda.is_available() else 0}"
        self.engine = WhisperTrainEngine(shape, device, activation_dropout=activation_dropout, freeze_base=freeze_base,
                                         dropout=dropout, attention_dropout=attention_dropout)
        self.shape = shape
        self.spec = spec or dict(apply_spec_augment=False, mask_time_prob=0.0, mask_time_length=10,
                                 mask_feature_prob=0.0, mask_feature_length=64)
        self.layerdrop = layerdrop
        self.training = False
        self.engine.training = False
        self._rng = np.random

    def train(self, mode: bool = True):
        self.training = mode
        self.engine.training = mode
        return self

    def eval(self):
        return self.train(False)

    @classmethod
    def from_pretrained(cls, name_or_path: str, device=None, seed: int = 4242, **overrides):
        spec = {k: overrides.pop(k) for k in ("apply_spec_augment", "mask_time_prob", "mask_time_length",
                                              "mask_feature_prob", "mask_feature_length") if k in overrides}
        kw = dict(activation_dropout=float(overrides.pop("activation_dropout", 0.0)),
                  dropout=float(overrides.pop("dropout", 0.0)),
                  attention_dropout=float(overrides.pop("attention_dropout", 0.0)),
                  freeze_base=bool(overrides.pop("freeze_base", False)), spec=spec or None,
                  layerdrop=float(overrides.pop("encoder_layerdrop", overrides.pop("layerdrop", 0.0))))
        path = Path(name_or_path)
        if path.is_dir() and (path / "config.json").exists():
            cfg = json.loads((path / "config.json").read_text())
            fields = WhisperShape.__dataclass_fields__
            model = cls(WhisperShape(**{k: cfg[k] for k in fields if k in cfg}), device, **kw)
            from .modeling import load_checkpoint_tensors

            sd = load_checkpoint_tensors(path)
            sd = {(k if k.startswith("model.") else "model." + k): v for k, v in sd.items() if k != "proj_out.weight"}
            model.engine.load_state_dict(sd)
            model.engine.refresh_derived()
            return model
        if name_or_path not in HUB_SHAPES:
            raise ValueError(f"unknown model {name_or_path!r}")
        model = cls(WhisperShape(**CORAL_WHISPER_SHAPES[HUB_SHAPES[name_or_path]]), device, **kw)
        logger.warning("no network / hub cache here: %s is instantiated with seeded random weights", name_or_path)
        g = torch.Generator(device=model.engine.device).manual_seed(seed)
        for n in model.engine.exported_names():
            v = model.engine.store.view(n)
            if n.endswith("layer_norm.weight"):
                v.fill_(1.0)
            elif n.endswith(".bias"):
                v.zero_()
            elif n.endswith("encoder.embed_positions.weight"):  # fixed sinusoid table (:55-64, requires_grad False)
                v.copy_(sinusoid_positions(*v.shape).to(v.device))
            else:
                v.normal_(0.0, 0.02, generator=g)
        model.engine.refresh_compute_weights()
        model.engine.refresh_derived()
        return model

    def save_pretrained(self, model_dir):
        from safetensors.torch import save_file

        model_dir = Path(model_dir)
        model_dir.mkdir(parents=True, exist_ok=True)
        cfg = dict(architectures=["WhisperForConditionalGeneration"], model_type="whisper", **self.shape.__dict__)
        (model_dir / "config.json").write_text(json.dumps(cfg, indent=2))
        save_file({k: v.cpu().contiguous() for k, v in self.engine.state_dict().items()},
                  str(model_dir / "model.safetensors"), metadata={"format": "pt"})

    def sample_spec_masks(self, B: int):
        sp = self.spec
        if not (self.training and sp.get("apply_spec_augment", False)):
            return None, None
        mt, mf = specaugment.sample_masks(B, 2 * self.shape.max_source_positions, self.shape.num_mel_bins, None,
                                          sp["mask_time_prob"], sp["mask_time_length"], sp["mask_feature_prob"],
                                          sp["mask_feature_length"], rng=self._rng)
        return (None if mt is None else torch.from_numpy(mt)), (None if mf is None else torch.from_numpy(mf))

    def sample_layer_keep(self):
        if not self.training or self.layerdrop <= 0:
            return None, None
        draw = lambda n: [bool(float(torch.rand([])) >= self.layerdrop) for _ in range(n)]  # noqa: E731
        return draw(self.shape.encoder_layers), draw(self.shape.decoder_layers)

    def __call__(self, input_features, labels=None, decoder_input_ids=None, mask_time=None, mask_feature=None,
                 enc_keep=None, dec_keep=None):
        if self.training and labels is not None:
            if mask_time is None and mask_feature is None:
                mask_time, mask_feature = self.sample_spec_masks(input_features.shape[0])
            if enc_keep is None and dec_keep is None:
                enc_keep, dec_keep = self.sample_layer_keep()
            out = self.engine(input_features, labels, mask_time, mask_feature, enc_keep, dec_keep)
            if out.get("loss") is not None:  # `out["loss"].backward()` runs the engine's backward (coral_amd/autograd.py)
                out["loss"] = attach_backward(self, out["loss"])
            return out
        return self.engine.forward(input_features, labels, decoder_input_ids)

    backward_kwargs: dict | None = None

    def backward(self, **kw):
        return self.engine.backward(**kw)

    def generate(self, input_features, language="danish", task="transcribe", max_length: int = 225, **_):
        if language not in ("danish", "da") or task != "transcribe":
            raise ValueError("only language='danish', task='transcribe' (CoRal's evaluation call) is wired up")
        prefix = prefix_ids(self.shape)
        # CoRal clears `suppress_tokens`; the default begin-suppress set (blank ' ' = 220, eos) stays
        return self.engine.generate(input_features, prefix, max_length, suppress_tokens=None,
                                    begin_suppress_tokens=[220, self.shape.eos_token_id])


class WhisperModelSetup(ModelSetup):
    def __init__(self, config) -> None:
        self.config = config
        self.processor = None
        self.model = None
        self.is_main_process = os.getenv("RANK", "0") == "0"

    def load_model(self):
        m = self.config.model
        self.model = WhisperForConditionalGeneration.from_pretrained(
            m.pretrained_model_id, seed=self.config.seed, dropout=m.dropout, activation_dropout=m.activation_dropout,
            attention_dropout=m.attention_dropout, apply_spec_augment=True, mask_time_prob=m.mask_time_prob,
            mask_time_length=m.mask_time_length, mask_feature_prob=m.mask_feature_prob,
            mask_feature_length=m.mask_feature_length, encoder_layerdrop=m.layerdrop, decoder_layerdrop=m.layerdrop,
            freeze_base=bool(m.freeze_feature_encoder))
        return self.model

    def load_processor(self):
        if self.model is None:
            self.load_model()
        tok = None
        tj = Path(self.config.model_dir) / "tokenizer.json"
        if tj.exists():
            from tokenizers import Tokenizer

            tok = Tokenizer.from_file(str(tj))
        self.processor = WhisperProcessor(WhisperFeatureExtractorGPU(self.model.engine, self.config.model.sampling_rate), tok)
        return self.processor

    def load_data_collator(self):
        def collate(features):
            """DataCollatorSpeechSeq2SeqWithPadding (R/src/coral/data_collators.py:130-187): stack
            input_features, pad labels with -100, strip a leading BOS present in every row."""
            feats = torch.stack([torch.as_tensor(f["input_features"]) for f in features])
            L = max(len(f["labels"]) for f in features)
            labels = torch.full((len(features), L), -100, dtype=torch.int64)
            for i, f in enumerate(features):
                labels[i, : len(f["labels"])] = torch.as_tensor(f["labels"])
            start = self.model.shape.decoder_start_token_id
            if bool((labels[:, 0] == start).all()):
                labels = labels[:, 1:]
            return dict(input_features=feats, labels=labels)

        return collate

    def load_trainer_class(self):
        return CoralTrainer

    def load_compute_metrics(self):
        def compute(pred_ids, label_ids):
            from .metrics import cer, wer

            labels = np.array(label_ids, copy=True)
            labels[labels == -100] = self.model.shape.pad_token_id
            preds = self.processor.batch_decode(pred_ids, skip_special_tokens=True)
            labs = self.processor.batch_decode(labels, skip_special_tokens=True)
            preds = [p.lower().strip() for p in preds]
            labs = [x.lower().strip() for x in labs]
            return dict(cer=cer(preds, labs), wer=wer(preds, labs))

        return compute

    def load_training_arguments(self):
        args = _training_args(self.config, self.config.model.learning_rate)
        args.generation_max_length = self.config.model.max_length  # predict_with_generate (whisper.py:221-222)
        return args

    def load_saved(self) -> PreTrainedModelData:
        model_dir = Path(self.config.model_dir)
        if not model_dir.exists():
            raise FileNotFoundError(f"{model_dir} does not exist (no hub access in this environment)")
        self.model = WhisperForConditionalGeneration.from_pretrained(str(model_dir))
        self.load_processor()
        return PreTrainedModelData(model=self.model, processor=self.processor,
                                   data_collator=self.load_data_collator(), compute_metrics=self.load_compute_metrics())
