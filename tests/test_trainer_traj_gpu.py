"""SURVEY.md row A11 pinned to the reference's OWN training loop: `transformers.Trainer` (R/src/coral/finetune.py:60-79;
$TF/trainer.py:1892-1963,1778-1796) ran six optimiser steps on the tiny wav2vec2 and the tiny Whisper in the build
container (`tools/gen_goldens.py trainer_traj`: CPU fp32, dropout 0, gradient_accumulation_steps 2, max_grad_norm 1.0,
cosine schedule with 2 warm-up steps, AdamW 0.9 / 0.98); `CoralTrainer` is driven through the same example stream here.
What the trajectory pins together: how the micro-batches of a step combine (transformers 5.x does NOT divide these
models' losses by the accumulation count), the gradient norm before clipping, the clip, the warm-up + cosine rate the
update uses, AdamW - and, through six updates, the parameters they produce.

Tolerances: learning rate exact (1e-12); per-step loss <= 1e-2 rel (bf16 engine against an fp32 CPU run; step 1 agrees
to ~1e-3, later steps carry the bf16 parameter rounding of the earlier updates); gradient norm <= 5e-2 rel; final
watched parameters cosine >= 0.999 against the reference's."""
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _args(tmp_path, B, accum, steps, lr, warmup):
    from coral_amd.model_setup import TrainingArgs

    return TrainingArgs(output_dir=str(tmp_path), per_device_train_batch_size=B, gradient_accumulation_steps=accum,
                        learning_rate=lr, warmup_steps=warmup, max_steps=steps, bf16=True, fp16=False, eval_steps=10_000,
                        save_steps=10_000, save_strategy="no", logging_steps=1, max_grad_norm=1.0, save_total_limit=0,
                        load_best_model_at_end=False, metric_for_best_model="cer", greater_is_better=False, seed=4242,
                        adam_beta1=0.9, adam_beta2=0.98, augment_audio=False, normalise_audio=False)


def _check(z, key, hist, final, lr, warmup, steps):
    from coral_amd.trainer import cosine_lr

    loss, gn, lrs = z[f"{key}:loss"], z[f"{key}:grad_norm"], z[f"{key}:lr_logged"]
    assert bool(z[f"{key}:model_accepts_loss_kwargs"])  # (why the reference does not scale by 1 / accumulation steps)
    logs = [h for h in hist if "loss" in h]
    assert [h["step"] for h in logs] == list(range(1, steps + 1))
    for k, h in enumerate(logs):
        assert abs(h["learning_rate"] - lrs[k]) <= 1e-12 and abs(cosine_lr(k, lr, warmup, steps) - lrs[k]) <= 1e-12
        assert abs(h["loss"] - loss[k]) <= 1e-2 * abs(loss[k]), (k, h["loss"], loss[k])
        assert abs(h["grad_norm"] - gn[k]) <= 5e-2 * gn[k], (k, h["grad_norm"], gn[k])
    assert abs(logs[0]["loss"] - loss[0]) <= 2e-3 * abs(loss[0])  # nothing has been updated yet: kernels only
    worst = 1.0
    for name, ((ref_final, init), got) in final.items():
        a, b, i0 = torch.as_tensor(ref_final).double().flatten(), got.double().flatten(), init.double().flatten()
        cos = float((a * b).sum() / (a.norm() * b.norm()))
        worst = min(worst, cos)
        assert cos >= 0.999, (name, cos)
        # ... and they MOVED as the reference's did: the update itself (final - initial) points the same way
        d0, d1 = a - i0, b - i0
        assert float((d0 * d1).sum() / (d0.norm() * d1.norm() + 1e-30)) >= 0.98, name
    return worst


def test_wav2vec2_trainer_trajectory_matches_transformers_trainer(golden_dir, tmp_path):
    from coral_amd.coral_trainer import CoralTrainer
    from coral_amd.modeling import Wav2Vec2ForCTC
    from coral_amd.wav2vec2 import Wav2Vec2Shape
    from oracle import wav2vec2_ref as ref  # test infrastructure: seeded parameters + the host featuriser

    z = np.load(golden_dir / "trainer_traj.npz")
    B, accum, steps, lr, warmup = (z["w2v2:hparams"][i] for i in range(5))
    B, accum, steps, warmup = int(B), int(accum), int(steps), int(warmup)
    kw = dict(hidden_size=128, num_hidden_layers=2, num_attention_heads=4, intermediate_size=256)
    model = Wav2Vec2ForCTC(Wav2Vec2Shape(**kw, activation_dropout=0.0, layerdrop=0.0), DEV)
    P = ref.synth_params(ref.W2V2Config(**kw))
    model.engine.load_state_dict(P)
    # the example stream of the golden run, regenerated from its seeds (tools/gen_goldens.py gen_trainer_traj)
    lens, lab_lens, labels = z["w2v2:lens"], z["w2v2:lab_lens"], z["w2v2:labels"]
    g = torch.Generator().manual_seed(77)
    waves = []
    for n in lens:
        x = (0.1 * torch.randn(int(n), generator=g)).clamp(-1, 1)
        waves.append((x / x.abs().max()).numpy())
    examples = []
    for w, lab, L in zip(waves, labels, lab_lens):
        iv, _ = ref.zero_mean_unit_var_norm([w])
        examples.append({"input_values": iv[0].astype(np.float32), "labels": [int(t) for t in lab[:int(L)]]})

    def collate(feats):  # DataCollatorCTCWithPadding, padding="longest" (R/src/coral/data_collators.py:62-95)
        n = max(len(f["input_values"]) for f in feats)
        iv = torch.zeros(len(feats), n)
        am = torch.zeros(len(feats), n, dtype=torch.long)
        Lm = max(len(f["labels"]) for f in feats)
        lab = torch.full((len(feats), Lm), -100, dtype=torch.long)
        for i, f in enumerate(feats):
            k = len(f["input_values"])
            iv[i, :k] = torch.from_numpy(f["input_values"])
            am[i, :k] = 1
            lab[i, :len(f["labels"])] = torch.tensor(f["labels"])
        return {"input_values": iv, "attention_mask": am, "labels": lab}

    collate.padding = "longest"
    trainer = CoralTrainer(model=model, args=_args(tmp_path, B, accum, steps, float(lr), warmup), data_collator=collate,
                           train_dataset=examples)
    assert trainer.dp.accum_loss == "sum"
    out = trainer.train()
    assert out.global_step == steps
    torch.cuda.synchronize()
    sd = model.engine.state_dict()
    final = {}
    for k in z.files:
        if k.startswith("w2v2:final:"):
            name = k[len("w2v2:final:"):]
            final[name] = ((z[k], P[name].float()), sd[name].float().cpu())
    worst = _check(z, "w2v2", trainer.state["log_history"], final, float(lr), warmup, steps)
    print("wav2vec2 trajectory: worst final-parameter cosine", worst)


def test_whisper_trainer_trajectory_matches_transformers_trainer(golden_dir, tmp_path):
    from coral_amd.coral_trainer import CoralTrainer
    from coral_amd.whisper import WhisperShape
    from coral_amd.whisper_setup import WhisperForConditionalGeneration
    from oracle import whisper_ref as w

    z = np.load(golden_dir / "trainer_traj.npz")
    B, accum, steps, lr, warmup = (z["whisper:hparams"][i] for i in range(5))
    B, accum, steps, warmup = int(B), int(accum), int(steps), int(warmup)
    kw = dict(d_model=64, encoder_layers=2, decoder_layers=2, encoder_attention_heads=4, decoder_attention_heads=4,
              encoder_ffn_dim=128, decoder_ffn_dim=128, num_mel_bins=80, vocab_size=200, max_target_positions=64,
              pad_token_id=150, decoder_start_token_id=151, eos_token_id=150)
    model = WhisperForConditionalGeneration(WhisperShape(**kw), DEV)
    P = w.synth_params(w.WhisperConfig(**kw))
    model.engine.load_state_dict(P)
    n = B * accum * steps
    g = torch.Generator().manual_seed(int(z["whisper:feats_seed"]))
    feats = (torch.randn(n, 80, 3000, generator=g) * 0.5).to(torch.float16).float()
    labels, lab_lens = z["whisper:labels"], z["whisper:lab_lens"]
    examples = [{"input_features": feats[i].numpy(), "labels": [int(t) for t in labels[i, :int(lab_lens[i])]]} for i in range(n)]

    def collate(fs):  # DataCollatorSpeechSeq2SeqWithPadding (R/src/coral/data_collators.py:145-187)
        Lm = max(len(f["labels"]) for f in fs)
        lab = torch.full((len(fs), Lm), -100, dtype=torch.long)
        for i, f in enumerate(fs):
            lab[i, :len(f["labels"])] = torch.tensor(f["labels"])
        return {"input_features": torch.stack([torch.from_numpy(f["input_features"]) for f in fs]), "labels": lab}

    trainer = CoralTrainer(model=model, args=_args(tmp_path, B, accum, steps, float(lr), warmup), data_collator=collate,
                           train_dataset=examples)
    out = trainer.train()
    assert out.global_step == steps
    torch.cuda.synchronize()
    sd = model.engine.state_dict()
    final = {}
    for k in z.files:
        if k.startswith("whisper:final:"):
            name = k[len("whisper:final:"):]
            final[name] = ((z[k], P[name].float()), sd[name].float().cpu())
    worst = _check(z, "whisper", trainer.state["log_history"], final, float(lr), warmup, steps)
    print("whisper trajectory: worst final-parameter cosine", worst)
