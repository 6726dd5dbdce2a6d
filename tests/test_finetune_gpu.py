"""The reference's own smoke test (R/tests/test_finetune.py:8-10: `finetune(config)` must run) on
the MI355X engine with the offline `synthetic` dataset, plus the save -> load_saved -> evaluate
round trip (HF safetensors layout) and a learning check: the loss on a fixed batch goes down."""
import json

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_finetune_entry_point_and_roundtrip(tmp_path, monkeypatch):
    import sys
    from pathlib import Path

    sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "scripts"))
    import finetune_asr_model

    from coral_amd import modeling
    from coral_amd.config import load_config
    from coral_amd.evaluate import evaluate
    from coral_amd.wav2vec2 import CORAL_W2V2_SHAPES

    # a small architecture behind the test model key so the smoke test takes seconds
    monkeypatch.setitem(modeling.HUB_SHAPES, "facebook/wav2vec2-xls-r-300m",
                        dict(hidden_size=128, num_hidden_layers=2, intermediate_size=256, num_attention_heads=4))
    res = finetune_asr_model.main(["model=test-wav2vec2", "datasets=synthetic", f"models_dir={tmp_path}",
                                   "model_id=smoke", "max_steps=2", "total_batch_size=2", "per_device_batch_size=2",
                                   "max_seconds_per_example=2.0", "min_seconds_per_example=1.0",
                                   "logging_steps=1", "eval_steps=2"])
    hist = res["history"]
    assert any("loss" in h for h in hist) and any("val_cer" in h for h in hist)
    # the path a CoRal user runs IS the measured one: the training batches went raw PCM -> device input pipeline
    assert res["trainer"].pipeline_batches == 2 and res["trainer"]._pipe is not None
    mdir = tmp_path / "smoke"
    assert (mdir / "model.safetensors").exists() and (mdir / "config.json").exists() and (mdir / "vocab.json").exists()
    cfg = json.loads((mdir / "config.json").read_text())
    assert cfg["architectures"] == ["Wav2Vec2ForCTC"] and cfg["vocab_size"] == 46 and cfg["pad_token_id"] == 45
    from safetensors.torch import load_file

    sd = load_file(str(mdir / "model.safetensors"))
    assert "wav2vec2.encoder.pos_conv_embed.conv.parametrizations.weight.original1" in sd
    assert sd["lm_head.weight"].shape == (46, 128)
    # frozen base (test config): only lm_head moved away from its init
    ev = load_config("evaluation", [f"model_id={mdir}", "batch_size=4", "store_results=false"])
    scores = evaluate(ev)
    assert 0.0 <= scores["cer"] and scores["n"] == 8


def test_training_reduces_loss_on_a_fixed_batch():
    from coral_amd.trainer import DataParallelTrainer
    from coral_amd.wav2vec2 import Wav2Vec2CTCEngine, Wav2Vec2Shape
    from oracle import wav2vec2_ref as ref

    kw = dict(hidden_size=128, num_hidden_layers=2, num_attention_heads=4, intermediate_size=256)
    eng = Wav2Vec2CTCEngine(Wav2Vec2Shape(**kw), "cuda:0")
    eng.load_state_dict(ref.synth_params(ref.W2V2Config(**kw)))
    g = torch.Generator().manual_seed(0)
    waves = [(0.1 * torch.randn(n, generator=g)).numpy() for n in (8000, 6400, 7000, 8000)]
    iv, am = ref.zero_mean_unit_var_norm(waves)
    labels = torch.randint(0, 42, (4, 6), generator=g)
    batch = dict(input_values=torch.from_numpy(iv), attention_mask=torch.from_numpy(am).long(), labels=labels)
    tr = DataParallelTrainer(eng, learning_rate=2e-3, warmup_steps=2, max_steps=40)
    losses = [float(tr.train_step([batch])) for _ in range(25)]
    assert np.isfinite(losses).all()
    assert losses[-1] < 0.6 * losses[1], losses
    assert tr.grad_norm() > 0


def test_gradient_norm_from_the_weight_gradient_epilogues_matches_a_pass_over_the_buffer(monkeypatch):
    """N = 1, whole model trainable: the squared norm assembled from the weight-gradient GEMMs' per-tile sums of squares
    (CaGemmDesc.c_sumsq) plus one pass over the small tensors equals the norm of the flat gradient buffer, with and
    without layerdrop in either accumulation micro-batch, and agrees with the pass that reads the buffer again
    (CA_FUSED_NORM=0) to fp32 summation order.  Reference: clip_grad_norm_ in $TF/trainer.py:1778-1796."""
    from coral_amd.trainer import DataParallelTrainer
    from coral_amd.wav2vec2 import Wav2Vec2CTCEngine, Wav2Vec2Shape
    from oracle import wav2vec2_ref as ref

    kw = dict(hidden_size=256, num_hidden_layers=3, num_attention_heads=4, intermediate_size=328)  # ragged 64-tiles
    g = torch.Generator().manual_seed(0)
    waves = [(0.1 * torch.randn(n, generator=g)).numpy() for n in (8000, 6400, 7000, 8000)]
    iv, am = ref.zero_mean_unit_var_norm(waves)
    labels = torch.randint(0, 42, (4, 6), generator=g)
    mb = lambda idx, **kw2: dict(input_values=torch.from_numpy(iv[idx]), attention_mask=torch.from_numpy(am[idx]),  # noqa: E731
                                 labels=labels[idx], **kw2)
    out = {}
    for fused, overlap in (("1", True), ("1", False), ("0", True)):
        monkeypatch.setenv("CA_FUSED_NORM", fused)
        eng = Wav2Vec2CTCEngine(Wav2Vec2Shape(**kw), "cuda:0")
        eng.load_state_dict(ref.synth_params(ref.W2V2Config(**kw)))
        tr = DataParallelTrainer(eng, learning_rate=1e-3, warmup_steps=1, max_steps=100, max_grad_norm=0.05, grad_accum=2,
                                 overlap_optimizer=overlap)
        assert (tr._norm_plan() is not None) == (fused == "1")
        norms = []
        for step in range(3):
            # two accumulation micro-batches; in step 1 the second one drops layer 1, in step 2 the first one does
            keep = [[True, step != 2, True], [True, step != 1, True]]
            tr.train_step([mb([0, 1], layer_keep=keep[0]), mb([2, 3], layer_keep=keep[1])])
            norms.append(tr.grad_norm())
            if not overlap:  # against the buffer itself (without the overlapped optimiser nothing clears gradients early)
                torch.cuda.synchronize()
                direct = float(eng.store.g32.double().pow(2).sum().sqrt())
                assert abs(norms[-1] - direct) <= 1e-5 * direct, (step, norms[-1], direct)
        tr.finish()
        torch.cuda.synchronize()
        out[(fused, overlap)] = (norms, eng.store.p32.clone())
    (n1, p1), (n2, p2), (n0, p0) = out[("1", True)], out[("1", False)], out[("0", True)]
    assert all(a > 0.05 for a in n0), n0  # the clip is active, so the norm matters
    # across runs only the first two steps are comparable: the two ways of summing differ in the 7th digit, the
    # parameters after step 1 therefore by ~1e-7, and a handful of bf16 weight copies that flip by one ulp move the third
    # step's gradients by 1e-3 (each run agrees with its own buffer above, at every step)
    assert np.allclose(n0[:2], n1[:2], rtol=1e-6) and np.allclose(n0[:2], n2[:2], rtol=1e-6), (n0, n1, n2)
    assert n1 == n2 and torch.equal(p1, p2)  # the overlapped optimiser changes nothing


def test_single_micro_batch_steps_keep_the_matrix_gradients_in_bf16(monkeypatch):
    """One micro-batch per optimiser step on one GPU: the layers' weight-matrix gradients go from the weight-gradient
    GEMMs to AdamW as bf16 (what the reference's autocast computes), everything else through the fp32 buffer.  Against the
    fp32 route (CA_WGRAD_BF16=0): the first step's loss is the same number, its gradient norm agrees to bf16 rounding of
    the matrices, the runs stay together; a dropped layer's matrices are cleared in the buffer that is read."""
    from coral_amd.trainer import DataParallelTrainer
    from coral_amd.wav2vec2 import Wav2Vec2CTCEngine, Wav2Vec2Shape
    from oracle import wav2vec2_ref as ref

    kw = dict(hidden_size=128, num_hidden_layers=3, num_attention_heads=4, intermediate_size=256)
    g = torch.Generator().manual_seed(0)
    waves = [(0.1 * torch.randn(n, generator=g)).numpy() for n in (8000, 6400, 7000, 8000)]
    iv, am = ref.zero_mean_unit_var_norm(waves)
    labels = torch.randint(0, 42, (4, 6), generator=g)
    out = {}
    for mode, overlap in (("0", True), ("1", True), ("1", False)):
        monkeypatch.setenv("CA_WGRAD_BF16", mode)
        eng = Wav2Vec2CTCEngine(Wav2Vec2Shape(**kw), "cuda:0")
        eng.load_state_dict(ref.synth_params(ref.W2V2Config(**kw)))
        tr = DataParallelTrainer(eng, learning_rate=1e-3, warmup_steps=1, max_steps=100, max_grad_norm=0.05, overlap_optimizer=overlap)
        losses, norms = [], []
        for step in range(4):
            keep = [True, step != 2, True]  # step 2 drops layer 1
            losses.append(float(tr.train_step([dict(input_values=torch.from_numpy(iv), attention_mask=torch.from_numpy(am),
                                                    labels=labels, layer_keep=keep)])))
            norms.append(tr.grad_norm())
            assert eng.matrix_grads_bf16 == (mode == "1")
            if step == 2 and mode == "1":
                tr.finish()
                torch.cuda.synchronize()
                lo, hi = eng.shard_ranges()["layer1"]
                assert float(eng.store.g16[lo:hi].float().abs().sum()) == 0.0
        tr.finish()
        torch.cuda.synchronize()
        out[(mode, overlap)] = (losses, norms, eng.store.p32.clone())
    (l0, n0, p0), (l1, n1, p1), (l2, n2, p2) = out[("0", True)], out[("1", True)], out[("1", False)]
    assert l1 == l2 and n1 == n2 and torch.equal(p1, p2)  # the overlapped optimiser changes nothing
    assert l0[0] == l1[0] and abs(n0[0] - n1[0]) <= 2e-3 * n0[0], (l0, l1, n0, n1)
    assert all(a > 0.05 for a in n0), n0  # the clip is active
    assert np.allclose(l0, l1, rtol=2e-3) and np.allclose(n0, n1, rtol=2e-2), (l0, l1, n0, n1)
    assert float((p0 - p1).abs().max()) <= 4.5e-3  # (4 steps of lr 1e-3: a sign flip of a noise-level element costs 2 lr per step)


def test_early_gradient_norm_pass_matches_the_single_pass(monkeypatch):
    """N = 1: the squared gradient norm taken in two pieces (tail of the flat buffer on the side stream during the
    backward, head behind it) gives the same clip factor as one pass: same parameters after the steps up to fp32
    summation order.  (The path engines without a norm plan take: CA_FUSED_NORM=0 here.)"""
    monkeypatch.setenv("CA_FUSED_NORM", "0")
    from coral_amd.trainer import DataParallelTrainer
    from coral_amd.wav2vec2 import Wav2Vec2CTCEngine, Wav2Vec2Shape
    from oracle import wav2vec2_ref as ref

    kw = dict(hidden_size=128, num_hidden_layers=4, num_attention_heads=4, intermediate_size=256)
    g = torch.Generator().manual_seed(0)
    waves = [(0.1 * torch.randn(n, generator=g)).numpy() for n in (8000, 6400, 7000, 8000)]
    iv, am = ref.zero_mean_unit_var_norm(waves)
    labels = torch.randint(0, 42, (4, 6), generator=g)
    out = {}
    for frac in (0.0, 0.5):
        eng = Wav2Vec2CTCEngine(Wav2Vec2Shape(**kw), "cuda:0")
        eng.load_state_dict(ref.synth_params(ref.W2V2Config(**kw)))
        tr = DataParallelTrainer(eng, learning_rate=1e-3, warmup_steps=1, max_steps=100, max_grad_norm=0.05)
        tr.early_fraction = frac
        norms = []
        for _ in range(3):
            tr.train_step([dict(input_values=torch.from_numpy(iv), attention_mask=torch.from_numpy(am), labels=labels)])
            norms.append(tr.grad_norm())
        tr.finish()
        torch.cuda.synchronize()
        out[frac] = (norms, eng.store.p32.clone())
    (n0, p0), (n1, p1) = out[0.0], out[0.5]
    assert all(a > 0.05 for a in n0), n0  # the clip is active, so the norm matters
    assert np.allclose(n0, n1, rtol=1e-5), (n0, n1)
    assert torch.allclose(p0, p1, rtol=0, atol=2e-6)


def test_checkpoints_rotate_and_resume_continues_the_run(tmp_path, monkeypatch):
    """`save_steps` / `save_total_limit` / `resume_from_checkpoint` (R/src/coral/wav2vec2.py:224-236,244,
    R/src/coral/finetune.py:79): a run resumed from its step-4 checkpoint ends, bit for bit, with the
    parameters and losses of the uninterrupted 6-step run (no host-drawn randomness in this configuration, data order continued), and the
    synthetic stream covers per-device batch x accumulation x steps examples (it restarts instead of running dry)."""
    import sys
    from pathlib import Path

    sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "scripts"))
    import finetune_asr_model

    from coral_amd import modeling

    monkeypatch.setitem(modeling.HUB_SHAPES, "facebook/wav2vec2-xls-r-300m",
                        dict(hidden_size=128, num_hidden_layers=2, intermediate_size=256, num_attention_heads=4))
    common = ["model=test-wav2vec2", "datasets=synthetic", f"models_dir={tmp_path}", "total_batch_size=4",
              "per_device_batch_size=2", "max_seconds_per_example=2.0", "min_seconds_per_example=1.0", "logging_steps=1",
              "eval_steps=100", "model.freeze_feature_encoder=false", "model.mask_time_prob=0.0",
              "model.mask_feature_prob=0.0", "model.activation_dropout=0.0", "model.layerdrop=0.0", "warmup_steps=2",
              "save_steps=2", "save_total_limit=2"]
    full = finetune_asr_model.main(common + ["model_id=full", "max_steps=6"])
    assert full["steps_done"] == 6 and full["trainer"].grad_accum == 2  # 2 micro-batches per step, 24 examples used
    names = sorted(p.name for p in (tmp_path / "full").glob("checkpoint-*"))
    assert names == ["checkpoint-4", "checkpoint-6"]                     # rotation keeps the newest two
    a = full["model"].engine.store.p32.clone()
    la = [h["loss"] for h in full["history"] if "loss" in h]
    # exact continuation: same schedule in both runs (max_steps=6 from the start, stopped by a save at step 4)
    part2 = finetune_asr_model.main(common + ["model_id=full", "max_steps=6", "resume_from_checkpoint=" + str(tmp_path / "full" / "checkpoint-4")])
    assert torch.equal(part2["model"].engine.store.p32, a)
    assert part2["steps_done"] == 6 and part2["trainer"].opt_step == 6
    assert [h["step"] for h in part2["history"] if "loss" in h] == [5, 6]
    assert [h["loss"] for h in part2["history"] if "loss" in h] == la[4:]
    # resume_from_checkpoint=true picks the newest checkpoint (step 6 = the end of the run): nothing left to do
    done = finetune_asr_model.main(common + ["model_id=full", "max_steps=6", "resume_from_checkpoint=true"])
    assert done["steps_done"] == 6 and torch.equal(done["model"].engine.store.p32, a)


def _tiny_config(tmp_path, monkeypatch, extra=()):
    import sys
    from pathlib import Path

    sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "scripts"))
    from coral_amd import modeling
    from coral_amd.config import load_config

    monkeypatch.setitem(modeling.HUB_SHAPES, "facebook/wav2vec2-xls-r-300m",
                        dict(hidden_size=128, num_hidden_layers=2, intermediate_size=256, num_attention_heads=4))
    return load_config("asr_finetuning", ["model=test-wav2vec2", "datasets=synthetic", f"models_dir={tmp_path}",
                                          "model_id=t", "max_steps=3", "total_batch_size=2", "per_device_batch_size=2",
                                          "max_seconds_per_example=2.0", "min_seconds_per_example=1.0", "logging_steps=1",
                                          "eval_steps=3", "model.freeze_feature_encoder=false", *extra])


def test_trainer_class_takes_the_reference_call_verbatim(tmp_path, monkeypatch):
    """`load_trainer_class()(model=, data_collator=, args=, compute_metrics=, train_dataset=, eval_dataset=,
    processing_class=, callbacks=)` then `.train(resume_from_checkpoint=)` - the call of R/src/coral/finetune.py:60-79,
    keyword for keyword, with transformers' own EarlyStoppingCallback class where it is importable.  The training
    batches must come through the device input pipeline (raw PCM -> GPU normalise / augment / featurise)."""
    from coral_amd.data import load_data_for_finetuning
    from coral_amd.model_setup import load_model_setup

    try:
        from transformers.trainer_callback import EarlyStoppingCallback
    except Exception:  # noqa: BLE001
        from coral_amd.coral_trainer import EarlyStoppingCallback
    config = _tiny_config(tmp_path, monkeypatch, ["early_stopping=true", "early_stopping_patience=5"])
    model_setup = load_model_setup(config)
    processor = model_setup.load_processor()
    model = model_setup.load_model()
    dataset = load_data_for_finetuning(config, processor)
    p0 = model.engine.store.p32.clone()
    trainer = model_setup.load_trainer_class()(
        model=model,
        data_collator=model_setup.load_data_collator(),
        args=model_setup.load_training_arguments(),
        compute_metrics=model_setup.load_compute_metrics(),
        train_dataset=dataset["train"],
        eval_dataset=dataset["val"],
        processing_class=getattr(processor, "tokenizer"),
        callbacks=[EarlyStoppingCallback(early_stopping_patience=config.early_stopping_patience)],
    )
    out = trainer.train(resume_from_checkpoint=config.resume_from_checkpoint)
    assert out.global_step == 3 and np.isfinite(out.training_loss)
    assert trainer.pipeline_batches == 3 and trainer._pipe is not None and trainer._pipe.augment is not None
    assert not torch.equal(model.engine.store.p32, p0)
    hist = trainer.state["log_history"]
    assert [h["step"] for h in hist if "loss" in h] == [1, 2, 3] and any("val_cer" in h for h in hist)
    m = trainer.evaluate()
    assert set(m) == {"eval_cer", "eval_wer"} and m["eval_cer"] >= 0.0
    model.save_pretrained(config.model_dir)
    assert (tmp_path / "t" / "model.safetensors").exists()


def test_loss_backward_drives_the_engine_backward():
    """`model(**batch)["loss"]` is a scalar with autograd ($TF/trainer.py:2005,2038).  `loss.backward()`
    fills the same flat gradient buffer, bit for bit, as `engine.backward()`; `(loss / 2).backward()` scales it by
    exactly 1/2 (the incoming gradient is applied on the device); a second backward through the same graph and a
    backward after a newer forward raise instead of silently re-running kernels."""
    from coral_amd.modeling import Wav2Vec2ForCTC
    from coral_amd.wav2vec2 import Wav2Vec2Shape
    from oracle import wav2vec2_ref as ref

    kw = dict(hidden_size=128, num_hidden_layers=2, num_attention_heads=4, intermediate_size=256)
    model = Wav2Vec2ForCTC(Wav2Vec2Shape(**kw), DEV)
    model.engine.load_state_dict(ref.synth_params(ref.W2V2Config(**kw)))
    g = torch.Generator().manual_seed(3)
    waves = [(0.1 * torch.randn(n, generator=g)).numpy() for n in (8000, 6400)]
    iv, am = ref.zero_mean_unit_var_norm(waves)
    batch = dict(input_values=torch.from_numpy(iv), attention_mask=torch.from_numpy(am).long(),
                 labels=torch.randint(0, 42, (2, 6), generator=g))
    eng = model.engine
    eng.zero_grad()
    out = model(**batch)
    assert out["loss"].requires_grad and out.loss.grad_fn is not None and out[0] is out["loss"]
    eng.backward()
    torch.cuda.synchronize()
    direct = eng.store.g32.clone()
    assert float(direct.abs().sum()) > 0
    eng.zero_grad()
    out = model(**batch)
    out.loss.backward()
    torch.cuda.synchronize()
    assert torch.equal(eng.store.g32, direct)
    with pytest.raises(RuntimeError):
        out.loss.backward()
    eng.zero_grad()
    (model(**batch)["loss"] / 2).backward()
    torch.cuda.synchronize()
    assert torch.equal(eng.store.g32, direct * 0.5)
    stale = model(**batch)
    model(**batch)
    with pytest.raises(RuntimeError):
        stale.loss.backward()
    with torch.no_grad():
        assert not model(**batch).loss.requires_grad


def test_checkpoint_on_a_save_only_step_holds_the_finished_update(tmp_path, monkeypatch):
    """A checkpoint written on a step that does not evaluate must contain the parameters AFTER that step's AdamW (which
    may still be running bucket by bucket on the optimiser stream when the save starts): same file contents as a run
    with the optimiser on the main stream (CA_OPT_OVERLAP=0)."""
    from safetensors.torch import load_file

    import finetune_asr_model  # noqa: F401  (path set up by _tiny_config)

    res = {}
    for overlap in ("1", "0"):
        monkeypatch.setenv("CA_OPT_OVERLAP", overlap)
        cfg = _tiny_config(tmp_path / overlap, monkeypatch, ["save_steps=1", "save_total_limit=5", "eval_steps=100",
                                                             "model.mask_time_prob=0.0", "model.mask_feature_prob=0.0",
                                                             "model.activation_dropout=0.0", "model.layerdrop=0.0"])
        from coral_amd.finetune import finetune

        finetune(cfg)
        res[overlap] = [load_file(str(tmp_path / overlap / "t" / f"checkpoint-{k}" / "model.safetensors")) for k in (1, 2, 3)]
    for a, b in zip(res["1"], res["0"]):
        assert a.keys() == b.keys()
        for k in a:
            assert torch.equal(a[k], b[k]), k


def test_fused_gradient_norm_when_a_matrix_alternates_between_direct_and_split_k():
    """A weight gradient whose K = B*T changes per batch can take the direct GEMM in one step (per-tile sums of squares
    in every slot) and the split-K fallback in the next (the whole sum in slot 0): the other slots must not keep the
    previous step's partials.  Checked on the op itself: slots after (direct, then split-K) sum to the split-K norm."""
    from coral_amd import ops

    M, N = 192, 192  # 9 slots of 64 x 64; 4 tiles of 128 -> the split-K rule applies when K divides
    slots = torch.zeros(ops.sumsq_slots(M, N) + 3, dtype=torch.float32, device=DEV)
    G = torch.zeros(M * N, dtype=torch.float32, device=DEV)
    g = torch.Generator(device=DEV).manual_seed(0)
    for K in (1001, 2048, 1001, 4096):
        dY = torch.randn(K, M, device=DEV, generator=g).to(torch.bfloat16)
        X = torch.randn(K, N, device=DEV, generator=g).to(torch.bfloat16)
        ops.wgrad_gemm(dY, X, G, M=M, N=N, K=K, lda=M, ldb=N, c_off=0, accumulate=False, sq=(slots, 1))
        torch.cuda.synchronize()
        want = float(G.double().pow(2).sum())
        got = float(slots[1:1 + ops.sumsq_slots(M, N)].double().sum())
        assert abs(got - want) <= 1e-5 * want, (K, ops._wgrad_splits(M, N, K), got, want)
        assert float(slots[0]) == 0.0 and float(slots[-2:].abs().sum()) == 0.0
    assert {ops._wgrad_splits(M, N, K) > 1 for K in (1001, 2048)} == {False, True}
