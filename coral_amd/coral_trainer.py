"""`CoralTrainer` — the class `ModelSetup.load_trainer_class()` returns, callable exactly as the
reference calls `transformers.Trainer` (R/src/coral/finetune.py:60-79):

    trainer = model_setup.load_trainer_class()(
        model=model, data_collator=..., args=..., compute_metrics=..., train_dataset=dataset["train"],
        eval_dataset=eval_dataset, processing_class=processor.tokenizer, callbacks=[EarlyStoppingCallback(...)])
    trainer.train(resume_from_checkpoint=config.resume_from_checkpoint)

Underneath: `DataParallelTrainer` (one process per GPU, bucketed gradient all-reduce over RCCL, fused
clip + AdamW; coral_amd/trainer.py) and the loop `Trainer._inner_training_loop` runs for CoRal's
settings ($TF/trainer.py:1678-1800: `max_steps` optimiser steps over a re-iterable stream,
`dataloader_drop_last`, evaluation every `eval_steps`, checkpoints every `save_steps` with rotation,
`load_best_model_at_end`, early stopping, resume with data skipping).

Training examples that still carry raw audio (`example["audio"]["array"]`, what the reference's dataset
holds before `process_example`, R/src/coral/data.py:704-757) take the device input path
(SURVEY.md §8f rows N1 + N4): the host only packs PCM into pinned staging memory; peak normalisation,
the augmentation chain (training only, as `augment_audio=True` there), zero-mean/unit-variance +
padding + attention mask, or Whisper's pad/trim + log-mel, all run on the GPU, one batch ahead of the
step that consumes them.  Examples that were featurised on the host (`input_values` /
`input_features`) go through the collator as in the reference.
"""

from __future__ import annotations

import json
import logging
import os
import shutil
import time
from dataclasses import dataclass, field
from pathlib import Path

import numpy as np
import torch

from .trainer import DataParallelTrainer

logger = logging.getLogger(__package__)


class EarlyStoppingCallback:
    """`transformers.EarlyStoppingCallback` look-alike (the reference passes one, R/src/coral/finetune.py:66-75).
    CoralTrainer only reads `early_stopping_patience` / `early_stopping_threshold`, so HF's own class works too."""

    def __init__(self, early_stopping_patience: int = 1, early_stopping_threshold: float = 0.0):
        self.early_stopping_patience = early_stopping_patience
        self.early_stopping_threshold = early_stopping_threshold


@dataclass
class TrainOutput:
    """What `Trainer.train` returns ($TF/trainer_utils.py TrainOutput)."""

    global_step: int
    training_loss: float
    metrics: dict = field(default_factory=dict)


def checkpoint_dirs(model_dir: Path) -> list[Path]:
    out = [d for d in Path(model_dir).glob("checkpoint-*") if d.is_dir() and d.name.split("-")[-1].isdigit()]
    return sorted(out, key=lambda d: int(d.name.split("-")[-1]))


class _RankAgreement:
    """`all(flag)` over the ranks without touching the GPU queue: one scalar MIN all-reduce on a gloo (CPU) group.
    (A device all-reduce followed by `.item()` blocks the host until everything already enqueued on the main stream
    has run - the host could no longer run ahead of the GPU.)"""

    def __init__(self):
        self.group = None
        if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            if torch.distributed.get_backend() == "gloo":
                self.group = torch.distributed.group.WORLD
            else:
                self.group = torch.distributed.new_group(backend="gloo")  # collective: every rank builds a trainer

    def __call__(self, flag: bool) -> bool:
        if self.group is None:
            return flag
        t = torch.tensor([1 if flag else 0], dtype=torch.int32)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MIN, group=self.group)
        return bool(int(t[0]))


class CoralTrainer:
    """See the module docstring.  Keyword set = the reference's call; everything else has a default."""

    def __init__(self, model=None, args=None, data_collator=None, train_dataset=None, eval_dataset=None,
                 processing_class=None, compute_metrics=None, callbacks=None, *, process_group=None,
                 compress_grads: bool = False, zero_stage: int | None = None):
        if model is None or args is None:
            raise ValueError("CoralTrainer needs `model=` and `args=`")
        self.model, self.args = model, args
        self.data_collator, self.compute_metrics = data_collator, compute_metrics
        self.train_dataset, self.eval_dataset = train_dataset, eval_dataset
        self.processing_class = processing_class
        self.callbacks = list(callbacks or [])
        self.patience = None
        for cb in self.callbacks:
            if hasattr(cb, "early_stopping_patience"):
                self.patience = int(cb.early_stopping_patience)
        a = args
        kw = {}
        if zero_stage is None:
            zero_stage = int(getattr(a, "zero_stage", 0) or 0)
        if zero_stage:
            kw["zero_stage"] = zero_stage
        self.dp = DataParallelTrainer(
            model, learning_rate=a.learning_rate, betas=(a.adam_beta1, a.adam_beta2), max_grad_norm=a.max_grad_norm,
            warmup_steps=a.warmup_steps, max_steps=a.max_steps, grad_accum=a.gradient_accumulation_steps,
            process_group=process_group, compress_grads=compress_grads,
            # the reference's Trainer (transformers 5.x) does not divide these models' losses by the accumulation count
            # (DataParallelTrainer.accum_loss; pinned by tests/golden/trainer_traj.npz); `accumulation_loss="mean"`
            # in the training arguments restores the textbook scaling
            accum_loss=getattr(a, "accumulation_loss", "sum"), **kw)
        self.engine = self.dp.engine
        self.is_seq2seq = hasattr(model, "generate")
        self.is_main = os.getenv("RANK", "0") == "0"
        self.state = dict(global_step=0, epoch=0, best_metric=None, best_step=None, bad_evals=0, log_history=[])
        self.best_dir = None
        self._agree = _RankAgreement()
        self._pipe = None           # DeviceInputPipeline, built on the first raw-audio batch
        self._staged = None         # the labels (host) of the batch whose audio is already on its way to the GPU
        self._it = None
        self.pipeline_batches = 0   # batches that came through the device input pipeline (tests assert on it)

    # ---- pass-throughs the rest of the package (and tests) use -----------------------------------------------
    opt_step = property(lambda self: self.dp.opt_step)
    grad_accum = property(lambda self: self.dp.grad_accum)
    m = property(lambda self: self.dp.m)
    v = property(lambda self: self.dp.v)
    lr = property(lambda self: self.dp.lr)

    def finish(self):
        self.dp.finish()

    def train_step(self, micro_batches):
        return self.dp.train_step(micro_batches)

    def grad_norm(self):
        return self.dp.grad_norm()

    # ---- data ----------------------------------------------------------------------------------------------------
    def _labels_of(self, feats):
        labs = [list(f["labels"]) for f in feats]
        if self.is_seq2seq:
            L = max(len(x) for x in labs)
            out = torch.full((len(labs), L), -100, dtype=torch.int64)
            for i, x in enumerate(labs):
                out[i, :len(x)] = torch.as_tensor(x)
            start = self.model.shape.decoder_start_token_id
            if bool((out[:, 0] == start).all()):  # DataCollatorSpeechSeq2SeqWithPadding (data_collators.py:183-186)
                out = out[:, 1:]
            return out
        padding = getattr(self.data_collator, "padding", "longest")
        tok = getattr(getattr(self.data_collator, "processor", None), "tokenizer", None)
        max_lab = min(getattr(tok, "model_max_length", 512), 512)
        L = max_lab if padding == "max_length" else max((len(x) for x in labs), default=0)
        out = np.full((len(labs), L), -100, dtype=np.int64)
        for i, x in enumerate(labs):
            x = x[:L]
            out[i, :len(x)] = x
        return torch.from_numpy(out)

    def _build_pipeline(self, B: int):
        from .input_pipeline import DeviceInputPipeline

        a = self.args
        sr = int(getattr(a, "sampling_rate", 16_000))
        aug = None
        if getattr(a, "augment_audio", True):
            from .augment import DeviceAugment

            aug = DeviceAugment(self.engine.device, sr, seed=int(getattr(a, "seed", 4242)) + 17 * int(os.getenv("RANK", "0") or 0),
                                background_noises=getattr(a, "background_noises", None))
        # the reference's augmentation chain itself starts with PeakNormalization(p=1) (R/src/coral/data.py:709-711):
        # augmented training audio is peak-normalised also when normalise_audio is off
        peak = bool(getattr(a, "normalise_audio", True)) or aug is not None
        if self.is_seq2seq:
            from .whisper import N_SAMPLES

            return DeviceInputPipeline(self.engine.device, B, N_SAMPLES, kind="whisper", dtype=np.float32,
                                       peak_normalize=peak, mel_filters=self.engine.mel_filters, augment=aug)
        n_max = int(sr * float(getattr(a, "max_seconds_per_example", 10.0)))
        return DeviceInputPipeline(self.engine.device, B, n_max, kind="wav2vec2", dtype=np.float32,
                                   padding=getattr(a, "padding", "longest") or "longest",
                                   peak_normalize=peak, augment=aug)

    def _pull(self, B: int):
        """B examples of the training stream, or None when the pass ran dry on some rank (`dataloader_drop_last`)."""
        feats = []
        try:
            while len(feats) < B:
                feats.append(next(self._it))
        except StopIteration:
            pass
        return feats if self._agree(len(feats) == B) else None

    def _pull_epochs(self, B: int):
        """The next per-device batch; a dry stream starts the next epoch (Trainer re-iterates an IterableDataset until
        `max_steps`, $TF/trainer.py:1678-1726), on every rank together."""
        for _ in range(2):
            feats = self._pull(B)
            if feats is not None:
                return feats
            self.state["epoch"] += 1
            self._it = iter(self.train_dataset)
        raise RuntimeError(f"the training stream yields fewer than per_device_batch_size={B} examples per epoch on "
                           "some rank: nothing to train on")

    def _stage(self, B: int):
        """Pull one batch and, when it holds raw audio, start its trip to the GPU (H2D copy on the pipeline's side
        stream).  -> ("raw", labels) or ("host", collated batch)."""
        feats = self._pull_epochs(B)
        if "audio" in feats[0] and getattr(self.args, "device_input_pipeline", True):
            if self._pipe is None:
                self._pipe = self._build_pipeline(B)
            self._pipe.submit([np.asarray(f["audio"]["array"]) for f in feats])
            return ("raw", self._labels_of(feats))
        return ("host", self.data_collator(feats))

    def next_micro_batch(self) -> dict:
        """One per-device batch for `model(**batch)`.  Raw-audio batches are staged ONE BATCH AHEAD: batch k+1 is
        packed and copied while the step of batch k computes; `pipe.get()` enqueues the device-side featurisation
        (normalise -> augment -> featurise) on the compute stream right in front of the forward."""
        B = self.args.per_device_train_batch_size
        if self._staged is None:
            self._staged = self._stage(B)
        kind, payload = self._staged
        if kind == "host":
            self._staged = None
            return payload
        batch = self._pipe.get()
        batch["labels"] = payload
        self.pipeline_batches += 1
        self._staged = self._stage(B)  # the next batch's PCM starts moving now
        if self._staged[0] != "raw":
            pass  # a mixed stream: the host-featurised batch simply waits its turn
        return batch

    # ---- checkpoints ----------------------------------------------------------------------------------------------
    def save_model(self, output_dir=None):
        """`Trainer.save_model`.  With the sharded optimiser this is a COLLECTIVE: between steps the fp32 master of the
        other ranks' slices is stale on this rank (only the bf16 copy is all-gathered), so every rank must call it - the
        masters are gathered first (`DataParallelTrainer.consolidate`), then the main process writes."""
        self.finish()
        if self.dp.zero:
            self.dp.consolidate()
        torch.cuda.synchronize()
        if self.is_main or not self.dp.zero:
            self.model.save_pretrained(output_dir or self.args.output_dir)

    def _moment_layout(self) -> str:
        """Fingerprint of the flat parameter layout the moment buffers follow: (name, offset, shape) of every tensor.
        `optimizer.safetensors` holds m and v as raw flat buffers; a file written under another layout has the same
        length and would attach moments to the wrong parameters."""
        import hashlib

        idx = self.engine.store.index
        text = ";".join(f"{n}@{off}:{'x'.join(map(str, shp))}" for n, (off, shp) in idx.items())
        return hashlib.sha256(text.encode()).hexdigest()[:32]

    def _rng_state(self) -> dict:
        """Host RNG streams the training step draws from: SpecAugment spans and LayerDrop decisions (the wav2vec2
        wrapper's own RandomState; np.random / torch's CPU generator for Whisper, as in HF) - Trainer keeps them in
        rng_state.pth so that a resumed run continues the interrupted one's mask sequence ($TF/trainer.py _save_rng_state)."""
        st = dict(numpy=np.random.get_state(), torch_cpu=torch.get_rng_state())
        rng = getattr(self.model, "_rng", None)
        if isinstance(rng, np.random.RandomState):
            st["wrapper"] = rng.get_state()
        return st

    def _set_rng_state(self, st: dict):
        np.random.set_state(st["numpy"])
        torch.set_rng_state(st["torch_cpu"])
        rng = getattr(self.model, "_rng", None)
        if "wrapper" in st and isinstance(rng, np.random.RandomState):
            rng.set_state(st["wrapper"])

    def _save_rng_state(self, d: Path):
        """`rng_state_<rank>.pth`, written by EVERY rank (HF's per-process `_save_rng_state`, $TF/trainer.py:3166-3201):
        the main rank writes its file with the checkpoint, the others after the barrier behind it (`train()`), when
        the directory exists - a resumed N-rank run continues each rank's own SpecAugment / LayerDrop sequence."""
        d.mkdir(parents=True, exist_ok=True)
        torch.save(self._rng_state(), str(d / f"rng_state_{int(os.getenv('RANK', '0') or 0)}.pth"))

    def _save_checkpoint(self, step: int, moments=None) -> Path:
        """`checkpoint-<step>/`: the model in HF layout, the optimiser moments and the trainer state
        (Trainer._save_checkpoint + rotation, $TF/trainer.py:3079,3326; `save_total_limit` never deletes the best)."""
        from safetensors.torch import save_file

        model_dir = Path(self.args.output_dir)
        d = model_dir / f"checkpoint-{step}"
        # the AdamW of the step just taken may still be running bucket by bucket on the optimiser stream: wait for it
        # BEFORE the parameters are read, or the file mixes pre- and post-update buckets
        self.finish()
        torch.cuda.synchronize()
        self.model.save_pretrained(d)
        m, v = moments if moments is not None else (self.dp.m, self.dp.v)
        save_file(dict(m=m.cpu(), v=v.cpu()), str(d / "optimizer.safetensors"), metadata={"layout": self._moment_layout()})
        self._save_rng_state(d)
        st = {k: self.state[k] for k in ("epoch", "best_metric", "best_step", "bad_evals")}
        (d / "trainer_state.json").write_text(json.dumps(dict(global_step=step, **st), indent=1))
        limit = self.args.save_total_limit
        if limit and limit > 0:
            keep = model_dir / f"checkpoint-{self.state['best_step']}" if self.state["best_step"] else None
            protected = {d, keep}
            deletable = [c for c in checkpoint_dirs(model_dir) if c not in protected]  # oldest first
            total = len(checkpoint_dirs(model_dir))
            while total > limit and deletable:
                shutil.rmtree(deletable.pop(0), ignore_errors=True)
                total -= 1
        return d

    def _load_checkpoint(self, ckpt: Path) -> dict:
        """Resume: parameters, optimiser moments and step count (the cosine schedule continues where it stopped)."""
        from safetensors.torch import load_file

        from .modeling import load_checkpoint_tensors

        sd = load_checkpoint_tensors(ckpt)
        eng = self.engine
        if any(k.startswith("model.") for k in sd) or hasattr(eng, "exported_names"):
            sd = {(k if k.startswith("model.") else "model." + k): v for k, v in sd.items() if k != "proj_out.weight"}
        if hasattr(eng, "exported_names"):
            eng.load_state_dict(sd)
        else:  # wav2vec2: `masked_spec_embed` is only in the file when SpecAugment is configured (as HF saves it)
            rep = eng.load_state_dict(sd, strict=False, init_missing=False)
            if [n for n in rep["missing"] if n != "wav2vec2.masked_spec_embed"]:
                raise KeyError(f"{ckpt}: checkpoint lacks {rep['missing']}")
        if hasattr(eng, "refresh_derived"):
            eng.refresh_derived()
        from safetensors import safe_open

        with safe_open(str(ckpt / "optimizer.safetensors"), "pt") as f:
            layout = (f.metadata() or {}).get("layout")
        if layout is not None and layout != self._moment_layout():
            raise ValueError(f"{ckpt}/optimizer.safetensors was written under another flat parameter layout "
                             f"({layout} != {self._moment_layout()}): its AdamW moments would land on the wrong parameters")
        if layout is None:
            # (a file from before the layout tag: its order cannot be verified - restart the moments rather than risk
            # attaching them to other parameters)
            if not getattr(self.args, "restart_untagged_moments", False):
                raise ValueError(f"{ckpt}/optimizer.safetensors carries no layout tag (written before round 4): its AdamW "
                                 "moments cannot be matched to the parameters.  Delete the file or pass "
                                 "`restart_untagged_moments=True` in the training arguments to restart them from zero")
            logger.warning("%s/optimizer.safetensors carries no layout tag: AdamW moments restart from zero", ckpt)
            self.dp.m.zero_()
            self.dp.v.zero_()
        else:
            opt = load_file(str(ckpt / "optimizer.safetensors"))
            self.dp.load_moments(opt["m"].to(self.dp.m.device), opt["v"].to(self.dp.v.device))
        rng_file = ckpt / f"rng_state_{int(os.getenv('RANK', '0') or 0)}.pth"
        self._resumed_rng = torch.load(str(rng_file), weights_only=False) if rng_file.exists() else None
        state = json.loads((ckpt / "trainer_state.json").read_text())
        self.dp.opt_step = int(state["global_step"])
        return state

    # ---- evaluation ---------------------------------------------------------------------------------------------
    def evaluate(self, eval_dataset=None, metric_key_prefix: str = "eval") -> dict:
        """Greedy decoding on the GPU + CER/WER on the host, keys prefixed like `Trainer.evaluate` (`eval_cer`, or
        `eval_<name>_cer` per entry when `eval_dataset` is a dict, $TF/trainer.py evaluate)."""
        from .finetune import evaluate_split, evaluate_split_seq2seq

        ds = eval_dataset if eval_dataset is not None else self.eval_dataset
        if ds is None:
            return {}
        sets = ds if isinstance(ds, dict) else {None: ds}
        out = {}
        B = getattr(self.args, "per_device_eval_batch_size", None) or self.args.per_device_train_batch_size
        for name, examples in sets.items():
            if self.is_seq2seq:
                m = evaluate_split_seq2seq(self.model, examples, self.data_collator, self.compute_metrics, B,
                                           getattr(self.args, "generation_max_length", 225))
            else:
                m = evaluate_split(self.model, examples, self.data_collator, self.compute_metrics, B)
            pre = metric_key_prefix + ("_" + name if name else "")
            out.update({f"{pre}_{k}": v for k, v in m.items()})
        return out

    def _best_metric_of(self, metrics: dict):
        """`metric_for_best_model` (the reference sets `val_<dataset>_<subset>_cer`, R/src/coral/wav2vec2.py:198-208;
        Trainer prepends `eval_`), else the CER of the first evaluation set."""
        key = getattr(self.args, "metric_for_best_model", None) or "cer"
        for k in (key, "eval_" + key):
            if k in metrics:
                return metrics[k]
        for k, v in metrics.items():
            if k.endswith("_cer"):
                return v
        return None

    # ---- the loop -------------------------------------------------------------------------------------------------
    def train(self, resume_from_checkpoint=None, ignore_data_skip: bool | None = None) -> TrainOutput:
        a = self.args
        model_dir = Path(a.output_dir)
        accum = a.gradient_accumulation_steps
        hist = self.state["log_history"]
        start_step = 0
        resume = resume_from_checkpoint
        if resume:  # True = the newest checkpoint under output_dir (Trainer.train(resume_from_checkpoint=True)); or a path
            ckpt = Path(resume) if isinstance(resume, (str, Path)) else (checkpoint_dirs(model_dir) or [None])[-1]
            if ckpt is None or not Path(ckpt).exists():
                raise FileNotFoundError(f"resume_from_checkpoint={resume!r}: no checkpoint-* directory under {model_dir}")
            st = self._load_checkpoint(Path(ckpt))
            start_step = int(st["global_step"])
            self.state.update({k: st[k] for k in ("best_metric", "best_step", "bad_evals") if k in st})
            if self.is_main:
                logger.info("resumed from %s at step %d", ckpt, start_step)
        self._it = iter(self.train_dataset)
        self._staged = None
        self.state["epoch"] = 0
        skip = getattr(a, "ignore_data_skip", False) if ignore_data_skip is None else ignore_data_skip
        if start_step and not skip:
            # Trainer skips the batches the first run consumed so that the data order continues (`ignore_data_skip`):
            # replay the batch construction itself - epoch ends, dropped tail batches, the ranks' agreement and the
            # augmentation draws then fall exactly where they fell in the first run
            for _ in range(start_step * accum):
                self.next_micro_batch()
        if start_step and getattr(self, "_resumed_rng", None) is not None:
            # the SpecAugment / LayerDrop streams continue where the interrupted run left them (HF restores rng_state.pth)
            self._set_rng_state(self._resumed_rng)
            self._resumed_rng = None
        t0 = time.time()
        step = start_step - 1
        loss_sum, loss_n = 0.0, 0
        lower_is_better = not getattr(a, "greater_is_better", False)
        for step in range(start_step, a.max_steps):
            micro = [self.next_micro_batch() for _ in range(accum)]
            loss = self.dp.train_step(micro)
            if (step + 1) % a.logging_steps == 0 or step == start_step:
                lv = float(loss)
                loss_sum, loss_n = loss_sum + lv, loss_n + 1
                # (as Trainer logs them: the loss over the step's micro-batches, the gradient norm BEFORE clipping and
                # the learning rate the update just taken used, $TF/trainer.py `_maybe_log_save_evaluate`)
                hist.append(dict(step=step + 1, loss=lv, grad_norm=float(self.dp.grad_norm()),
                                 learning_rate=float(getattr(self.dp, "last_lr", self.dp.lr)), lr=self.dp.lr,
                                 epoch=self.state["epoch"], elapsed=time.time() - t0))
                if self.is_main:
                    logger.info("step %d loss %.4f", step + 1, lv)
            stop = False
            evaluated = self.eval_dataset is not None and ((step + 1) % a.eval_steps == 0 or step + 1 == a.max_steps)
            if evaluated:
                metrics = self.evaluate()
                # (history keeps the short names the reference's logs show: val_cer / val_wer for a single set)
                hist.append(dict(step=step + 1, **{("val_" + k[len("eval_"):]): v for k, v in metrics.items()}))
                cur = self._best_metric_of(metrics)
                best = self.state["best_metric"]
                better = cur is not None and (best is None or (cur < best if lower_is_better else cur > best))
                if better:
                    self.state.update(best_metric=cur, best_step=step + 1, bad_evals=0)
                else:
                    self.state["bad_evals"] += 1
                # EarlyStoppingCallback (R/src/coral/finetune.py:66-75): stop after `patience` evaluations without a new best
                stop = self.patience is not None and self.state["bad_evals"] >= self.patience
            save_now = a.save_strategy != "no" and ((step + 1) % a.save_steps == 0 or
                                                    (self.state["best_step"] == step + 1 and a.load_best_model_at_end))
            # (sharded optimiser: every rank takes part in gathering the master parameters and moments rank 0 writes)
            moments = self.dp.consolidate() if (save_now and self.dp.zero) else None
            if save_now and self.is_main:
                d = self._save_checkpoint(step + 1, moments)
                if self.state["best_step"] == step + 1:
                    self.best_dir = d
            if save_now and torch.distributed.is_available() and torch.distributed.is_initialized():
                torch.distributed.barrier()  # the main rank's directory exists (and survived the rotation)
                if not self.is_main:
                    self._save_rng_state(Path(a.output_dir) / f"checkpoint-{step + 1}")
                    torch.distributed.barrier()  # nobody resumes or rotates before every rank's file is there
                else:
                    torch.distributed.barrier()
            if stop:
                if self.is_main:
                    logger.info("early stopping at step %d (best %.4f at step %s)", step + 1, self.state["best_metric"],
                                self.state["best_step"])
                break
        self.finish()  # the last optimiser step may still be running on the trainer's side stream
        if self.dp.zero:
            self.dp.consolidate()  # the master parameters `model.save_pretrained` reads are complete on every rank again
        torch.cuda.synchronize()
        self.state["global_step"] = step + 1
        if a.load_best_model_at_end and self.state["best_step"] and self.state["best_step"] != step + 1:
            best = model_dir / f"checkpoint-{self.state['best_step']}"
            if best.exists():  # `load_best_model_at_end` (R/src/coral/wav2vec2.py:233): the saved model is the best one
                kept = self.dp.opt_step
                self._load_checkpoint(best)
                self.dp.opt_step = kept
                if self.is_main:
                    logger.info("loaded the best model (step %d)", self.state["best_step"])
        return TrainOutput(step + 1, loss_sum / max(1, loss_n), dict(train_runtime=time.time() - t0))
