"""The drop-in boundary: CoRal's `ModelSetup` ABC and factory (R/src/coral/data_models.py:44-82,
R/src/coral/model_setup.py:10-31) re-implemented over the MI355X engine.

`load_model_setup(config)` dispatches on `config.model.type`; the returned object offers the same
seven loaders with the same meaning.  Instead of `transformers.Trainer` the trainer class is
`coral_amd.coral_trainer.CoralTrainer` (Trainer's constructor and `.train()`; underneath one process per GPU, RCCL
gradient all-reduce, coral_amd/trainer.py), and the
"training arguments" are a plain dataclass carrying the values `TrainingArguments` would.
"""

from __future__ import annotations

import json
import logging
import os
from abc import ABC, abstractmethod
from dataclasses import dataclass, field
from functools import partial
from pathlib import Path

import torch

from .compute_metrics import compute_error_rate_metrics
from .data_collators import DataCollatorCTCWithPadding
from .processor import CTCTokenizer, Wav2Vec2Processor, WaveformFeatureExtractor, dump_vocabulary
from .coral_trainer import CoralTrainer
from .trainer import grad_accumulation_steps

logger = logging.getLogger(__package__)


@dataclass
class PreTrainedModelData:
    """R/src/coral/data_models.py:23-41."""

    model: object
    processor: object
    data_collator: object
    compute_metrics: object


@dataclass
class TrainingArgs:
    """The subset of `TrainingArguments` CoRal sets (R/src/coral/wav2vec2.py:209-250)."""

    output_dir: str
    per_device_train_batch_size: int
    gradient_accumulation_steps: int
    learning_rate: float
    warmup_steps: int
    max_steps: int
    bf16: bool
    fp16: bool
    eval_steps: int
    save_steps: int
    save_strategy: str
    logging_steps: int
    max_grad_norm: float
    save_total_limit: int
    load_best_model_at_end: bool
    metric_for_best_model: str
    greater_is_better: bool
    seed: int
    adam_beta1: float
    adam_beta2: float
    lr_scheduler_type: str = "cosine"
    optim: str = "adamw_torch"
    dataloader_num_workers: int = 4
    dataloader_drop_last: bool = True
    ddp_find_unused_parameters: bool = False
    report_to: list = field(default_factory=list)
    ignore_data_skip: bool = False
    # what the device input path needs to know about the examples (the reference applies these per example on the host,
    # R/src/coral/data.py:254-255,704-747: training = normalise + augment)
    sampling_rate: int = 16_000
    max_seconds_per_example: float = 10.0
    padding: str = "longest"
    normalise_audio: bool = True
    augment_audio: bool = True
    device_input_pipeline: bool = True
    zero_stage: int = 0   # >0: optimiser state + update sharded over the ranks (the reference's `--zero-stage 2` launch)
    # "sum": the micro-batches of an optimiser step are NOT scaled by 1 / gradient_accumulation_steps - what the
    # reference's transformers.Trainer (>= 4.46; pinned 5.5.0) does for Wav2Vec2ForCTC / WhisperForConditionalGeneration
    # (their forwards accept **kwargs, $TF/trainer.py:496-505,1952-1954); "mean": the textbook scaling
    accumulation_loss: str = "sum"
    restart_untagged_moments: bool = False  # resume from an optimizer.safetensors without a layout tag: restart m, v from zero


class ModelSetup(ABC):
    """Base class for a model setup (same abstract methods as the reference's)."""

    @abstractmethod
    def __init__(self, config) -> None: ...

    @abstractmethod
    def load_processor(self): ...

    @abstractmethod
    def load_model(self): ...

    @abstractmethod
    def load_data_collator(self): ...

    @abstractmethod
    def load_trainer_class(self): ...

    @abstractmethod
    def load_compute_metrics(self): ...

    @abstractmethod
    def load_training_arguments(self): ...

    @abstractmethod
    def load_saved(self) -> PreTrainedModelData: ...


def _training_args(config, learning_rate) -> TrainingArgs:
    num_devices = max(torch.cuda.device_count(), 1)  # visible devices, not WORLD_SIZE (wav2vec2.py:159)
    accum = grad_accumulation_steps(config.total_batch_size, num_devices, config.per_device_batch_size)
    if config.total_batch_size // num_devices // config.per_device_batch_size == 0 and os.getenv("RANK", "0") == "0":
        logger.warning("`total_batch_size` is too small for %d devices x per_device_batch_size %d; using %d.",
                       num_devices, config.per_device_batch_size, config.per_device_batch_size * num_devices)
    bf16 = bool(config.bf16_allowed)  # MI355X always has bf16 MFMA
    if config.early_stopping:
        config.save_total_limit = max(config.save_total_limit, 1)
    ev = config.evaluation_datasets[0]
    metric = ("val_" + ev["id"].split("/")[-1] + "_" + str(ev["subset"]) + "_cer").lower().replace("-", "_")
    return TrainingArgs(
        output_dir=config.model_dir, per_device_train_batch_size=config.per_device_batch_size,
        gradient_accumulation_steps=accum, learning_rate=learning_rate, warmup_steps=config.warmup_steps,
        max_steps=config.max_steps, bf16=bf16, fp16=False, eval_steps=config.eval_steps,
        save_steps=config.save_steps, save_strategy="no" if config.save_total_limit == 0 else "steps",
        logging_steps=config.logging_steps, max_grad_norm=config.max_grad_norm,
        save_total_limit=config.save_total_limit, load_best_model_at_end=config.early_stopping,
        metric_for_best_model=metric, greater_is_better=False, seed=config.seed,
        adam_beta1=config.adam_first_momentum, adam_beta2=config.adam_second_momentum,
        dataloader_num_workers=config.dataloader_num_workers, ignore_data_skip=bool(config.get("ignore_data_skip", False)),
        sampling_rate=int(config.model.sampling_rate), max_seconds_per_example=float(config.max_seconds_per_example),
        padding=config.padding if isinstance(config.padding, str) else "longest",
        augment_audio=bool(config.get("augment_audio", True)), normalise_audio=bool(config.get("normalise_audio", True)),
        device_input_pipeline=bool(config.get("device_input_pipeline", True)),
        zero_stage=int(config.get("zero_stage", 0) or 0))


class Wav2Vec2ModelSetup(ModelSetup):
    """Model setup for Wav2Vec 2.0 models (R/src/coral/wav2vec2.py:33-305)."""

    def __init__(self, config) -> None:
        self.config = config
        self.processor = None
        self.is_main_process = os.getenv("RANK", "0") == "0"

    def load_processor(self) -> Wav2Vec2Processor:
        # rank 0 writes vocab.json, everybody reads it (the reference lets every rank write and
        # retries on JSONDecodeError, R/src/coral/wav2vec2.py:61-84)
        vocab_path = Path(self.config.model_dir) / "vocab.json"
        if self.is_main_process or not vocab_path.exists():
            dump_vocabulary(self.config.model.characters_to_keep, self.config.model_dir)
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            torch.distributed.barrier()
        tokenizer = CTCTokenizer.from_pretrained(self.config.model_dir)
        extractor = WaveformFeatureExtractor(sampling_rate=self.config.model.sampling_rate, do_normalize=True)
        self.processor = Wav2Vec2Processor(extractor, tokenizer)
        return self.processor

    def load_model(self):
        from .modeling import Wav2Vec2ForCTC

        m = self.config.model
        tok = self.processor.tokenizer
        model = Wav2Vec2ForCTC.from_pretrained(
            m.pretrained_model_id, activation_dropout=m.activation_dropout, attention_dropout=m.attention_dropout,
            hidden_dropout=m.hidden_dropout, feat_proj_dropout=m.feat_proj_dropout, final_dropout=m.final_dropout,
            apply_spec_augment=True,
            mask_time_prob=m.mask_time_prob, mask_time_length=m.mask_time_length,
            mask_feature_prob=m.mask_feature_prob, mask_feature_length=m.mask_feature_length,
            layerdrop=m.layerdrop, ctc_loss_reduction=m.ctc_loss_reduction, pad_token_id=tok.pad_token_id,
            vocab_size=len(tok.get_vocab()), ctc_zero_infinity=True,
            freeze_base=bool(m.freeze_feature_encoder), seed=self.config.seed)
        return model

    def load_data_collator(self) -> DataCollatorCTCWithPadding:
        return DataCollatorCTCWithPadding(processor=self.processor, sample_rate=self.config.model.sampling_rate,
                                          max_seconds_per_example=self.config.max_seconds_per_example,
                                          padding=self.config.padding)

    def load_trainer_class(self):
        return CoralTrainer

    def load_compute_metrics(self):
        return partial(compute_error_rate_metrics, processor=self.processor)

    def load_training_arguments(self) -> TrainingArgs:
        return _training_args(self.config, self.config.model.learning_rate)

    def load_saved(self) -> PreTrainedModelData:
        from .modeling import Wav2Vec2ForCTC

        model_dir = Path(self.config.model_dir)
        if not model_dir.exists():
            raise FileNotFoundError(f"{model_dir} does not exist (no hub access in this environment)")
        tokenizer = CTCTokenizer.from_pretrained(model_dir)
        processor = Wav2Vec2Processor(WaveformFeatureExtractor(self.config.model.sampling_rate), tokenizer)
        model = Wav2Vec2ForCTC.from_pretrained(str(model_dir))
        collator = DataCollatorCTCWithPadding(processor=processor, sample_rate=self.config.model.sampling_rate,
                                              max_seconds_per_example=self.config.max_seconds_per_example,
                                              padding=self.config.padding)
        return PreTrainedModelData(model=model, processor=processor, data_collator=collator,
                                   compute_metrics=partial(compute_error_rate_metrics, processor=processor))


def load_model_setup(config) -> ModelSetup:
    """R/src/coral/model_setup.py:10-31."""
    model_type = config.model.type
    if model_type == "wav2vec2":
        return Wav2Vec2ModelSetup(config)
    if model_type == "whisper":
        from .whisper_setup import WhisperModelSetup

        return WhisperModelSetup(config)
    raise ValueError(f"Unsupported model type: {model_type!r}")
