#!/usr/bin/env python3
"""Headline benchmark: audio-seconds/sec of CTC finetuning steps (fwd + bwd + gradient all-reduce +
clip + AdamW) for CoRal's `model=wav2vec2-large` (XLS-R-2B shape) on synthetic 16 kHz audio,
8 x 10 s utterances per GPU, bf16 MFMA compute / fp32 master weights.  BASELINE.json configs[1]
(N=1) and configs[2] (N=8, weak scaling).

    python bench.py --gpus 1 --steps 8 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (contract in the task statement): metric/value/unit/... plus
  roofline     — dominant GEMM template, live hipEvent timing through ca_prof_begin/end
  cpu_baseline — the oracle (oracle/wav2vec2_ref.py) timed on this box's host cores (N=1 only)
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

MFMA_BF16_DENSE_PEAK_TFLOPS = 2500.0  # /opt/skills/guides/MI355X_MICROARCH.md: ~2.5 PF dense (spec)


def fwd_gflop_per_utt(shape, T: int, conv_lens) -> float:
    """cnn + featproj + posconv + L*T*(8 d^2 + 4 d ffn) + L*4 T^2 d + 2 d V T."""
    d, f, L, V = shape.hidden_size, shape.intermediate_size, shape.num_hidden_layers, shape.vocab_size
    cnn, cin = 0.0, 1
    for co, k, n in zip(shape.conv_dim, shape.conv_kernel, conv_lens):
        cnn += 2.0 * n * co * cin * k
        cin = co
    featproj = 2.0 * T * cin * d
    K, G = shape.num_conv_pos_embeddings, shape.num_conv_pos_embedding_groups
    posconv = 2.0 * T * d * (d // G) * K
    enc = L * T * (8.0 * d * d + 4.0 * d * f) + L * 4.0 * T * T * d
    head = 2.0 * d * V * T
    return (cnn + featproj + posconv + enc + head) / 1e9


def init_random_(engine, seed: int):
    """Seeded random-init weights generated on the device (no checkpoints exist offline)."""
    g = torch.Generator(device=engine.device).manual_seed(seed)
    st = engine.store
    for name, (off, shape) in st.index.items():
        v = st.view(name)
        if name.endswith("layer_norm.weight"):
            v.normal_(1.0, 0.05, generator=g)
        elif name.endswith("original0"):
            v.uniform_(1.0, 1.25, generator=g)
        elif name.endswith(".bias") or name.endswith("masked_spec_embed"):
            v.normal_(0.0, 0.02, generator=g)
        else:
            fan_in = 1
            for s_ in shape[1:]:
                fan_in *= s_
            v.normal_(0.0, fan_in ** -0.5, generator=g)
    engine.refresh_compute_weights()


def synth_batch(B, seconds, rank, device, ragged=False):
    """0.1*randn clipped, peak-normalised, then the feature extractor's zero-mean /
    unit-variance (done on the GPU by ca_wave_normalize); labels U{0..41}, length U{20..120}."""
    from coral_amd import ops

    N = int(16000 * seconds)
    g = torch.Generator().manual_seed(4242 + rank)
    x = (0.1 * torch.randn(B, N, generator=g)).clamp_(-1, 1)
    x = x / x.abs().amax(dim=1, keepdim=True)
    lens = torch.full((B,), N, dtype=torch.int32)
    if ragged:
        lens = torch.randint(16000, N + 1, (B,), generator=g, dtype=torch.int32)
    am = (torch.arange(N)[None, :] < lens[:, None]).to(torch.int32)
    xd, y = x.to(device), torch.empty(B, N, device=device)
    ops.wave_normalize(xd, lens.to(device), y, B, N)
    tl = torch.randint(20, 121, (B,), generator=g)
    labels = torch.full((B, int(tl.max())), -100, dtype=torch.int32)
    for b in range(B):
        labels[b, :tl[b]] = torch.randint(0, 42, (int(tl[b]),), generator=g, dtype=torch.int32)
    return dict(input_values=y, attention_mask=am.to(device), labels=labels.to(device)), lens


def cpu_baseline(model_key: str, seconds: float = 2.0, max_threads: int = 32):
    """Time the oracle (fp32 torch CPU restatement of the HF path) on this box's host cores on a
    BOUNDED sample of the same workload: ONE utterance of `seconds` s through the bench's
    architecture (fwd + bwd incl. CTC).  Threads are capped at 32: torch's CPU GEMMs get slower,
    not faster, when oversubscribed on the 256-thread GPU host (a 10 s sample took 647 s there)."""
    import numpy as np

    from oracle import wav2vec2_ref as ref

    cores = min(os.cpu_count() or 1, max_threads)
    torch.set_num_threads(cores)
    cfg = ref.W2V2Config(**ref.CORAL_SHAPES[model_key])
    g = torch.Generator().manual_seed(1)
    P = {}
    for name, shape in ref.param_shapes(cfg).items():
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
        if name.endswith("layer_norm.weight") or name.endswith("original0"):
            t = torch.ones(shape)
        elif name.endswith(".bias") or name.endswith("masked_spec_embed"):
            t = torch.zeros(shape)
        else:
            t = torch.empty(shape).normal_(0.0, fan_in ** -0.5, generator=g)
        P[name] = t.requires_grad_(True)
    N = int(16000 * seconds)
    x = 0.1 * torch.randn(1, N, generator=g)
    x = (x - x.mean()) / x.std()
    labels = torch.randint(0, 42, (1, max(2, int(6 * seconds))), generator=g)
    t0 = time.time()
    loss, _, _ = ref.forward_loss(x, None, labels, P, cfg)
    loss.backward()
    dt = time.time() - t0
    return {"value": round(seconds / dt, 4), "unit": "audio-seconds/sec", "cores": cores, "kind": "port",
            "sample": f"1 x {seconds:g} s utterance, {model_key} shape, fwd+bwd+CTC fp32, {dt:.1f} s wall "
                      f"(oracle/wav2vec2_ref.py, torch {torch.__version__} CPU, {cores} threads)"}


def whisper_bench(args, world, rank, device):
    """Secondary workload (BASELINE.json configs[3]/[4] shapes, bf16): Whisper teacher-forced finetune
    step on 30 s clips (log-mel on the GPU inside the step) or greedy decoding (--decode)."""
    import numpy as np

    from coral_amd import ops
    from coral_amd.trainer import DataParallelTrainer
    from coral_amd.whisper import CORAL_WHISPER_SHAPES, N_SAMPLES, WhisperShape
    from coral_amd.whisper_train import WhisperTrainEngine

    shape = WhisperShape(**CORAL_WHISPER_SHAPES[args.model])
    eng = WhisperTrainEngine(shape, device, activation_dropout=0.1)
    g = torch.Generator(device=device).manual_seed(4242)
    for n in eng.exported_names():
        v = eng.store.view(n)
        if n.endswith("layer_norm.weight"):
            v.fill_(1.0)
        elif n.endswith(".bias"):
            v.zero_()
        elif n == "model.encoder.embed_positions.weight":
            from coral_amd.whisper_setup import WhisperForConditionalGeneration  # noqa: F401
            T, d = shape.max_source_positions, shape.d_model
            inc = np.log(10000.0) / (d // 2 - 1)
            inv = torch.exp(-inc * torch.arange(d // 2, device=device))
            t = torch.arange(T, device=device)[:, None] * inv[None, :]
            v.copy_(torch.cat([t.sin(), t.cos()], 1))
        else:
            v.normal_(0.0, 0.02, generator=g)
    eng.refresh_compute_weights()
    B = args.batch
    gen = torch.Generator().manual_seed(4242 + rank)
    waves = torch.zeros(B, N_SAMPLES)
    for b in range(B):
        n = int(torch.randint(112_000, N_SAMPLES + 1, (1,), generator=gen))
        waves[b, :n] = 0.1 * torch.randn(n, generator=gen)
    waves = waves.to(device)
    tl = torch.randint(20, 121, (B,), generator=gen)
    labels = torch.full((B, int(tl.max())), -100, dtype=torch.int64)
    for b in range(B):
        labels[b, :tl[b]] = torch.randint(0, 50257, (int(tl[b]),), generator=gen)
    trainer = DataParallelTrainer(eng, learning_rate=6e-6, betas=(0.9, 0.98), warmup_steps=1000, max_steps=100_000,
                                  compress_grads=(args.grad_wire == "bf16"))

    if args.decode and args.fp8_encoder:
        eng.enable_fp8_encoder()
    if args.fp8_forward and not args.decode:
        eng.enable_fp8_forward()

    def step():
        feats = eng.log_mel(waves)  # front end on the GPU, inside the step
        if args.decode:
            return eng.generate(feats, [50258, 50285, 50359, 50363], 4 + args.decode_tokens)
        return trainer.train_step([dict(input_features=feats, labels=labels)])

    def sync():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    sync()
    dt = time.perf_counter() - t0
    if rank == 0:
        audio_s = world * B * 30.0 * args.steps
        mode = f"greedy decode, {args.decode_tokens} new tokens" if args.decode else "finetune step fwd+bwd+clip+AdamW, teacher-forced"
        print(json.dumps({
            "metric": f"audio-seconds/sec ({mode}), {args.model}, 30 s clips", "value": round(audio_s / dt, 2),
            "unit": "audio-seconds/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16 + fp8 e4m3 (encoder q|k|v, fc1 forward)" if ((args.decode and args.fp8_encoder) or (args.fp8_forward and not args.decode)) else "bf16",
            "data": "synthetic",
            "config": {"workload": f"{args.model} {mode}, {B} x 30 s per GPU, log-mel on GPU", "global_batch": world * B,
                       "label_len": int(labels.shape[1]), "parallelism": f"dp{world}"}}), flush=True)
    if world > 1:
        if args.check_replicas and not args.decode:
            torch.cuda.synchronize()
            p = eng.store.p32
            lo, hi = p.clone(), p.clone()
            torch.distributed.all_reduce(lo, op=torch.distributed.ReduceOp.MIN)
            torch.distributed.all_reduce(hi, op=torch.distributed.ReduceOp.MAX)
            spread = float((hi - lo).abs().max())
            if rank == 0:
                print(json.dumps({"replica_param_spread": spread}), flush=True)
            assert spread == 0.0, f"replicas diverged: {spread}"
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--model", default="wav2vec2-large", help="CoRal model key (wav2vec2-small/medium/large)")
    ap.add_argument("--batch", type=int, default=8, help="utterances per GPU")
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-specaugment", action="store_true")
    ap.add_argument("--from-host-pcm", action="store_true",
                    help="feed every step from raw int16 PCM in host memory through the device input pipeline "
                         "(pinned staging + side-stream H2D + on-GPU normalisation): the PCIe-inclusive rate")
    ap.add_argument("--gemm-breakdown", action="store_true", help="print the per-kernel GEMM timing table to stderr")
    ap.add_argument("--decode", action="store_true", help="whisper models: time greedy decoding instead of training")
    ap.add_argument("--decode-tokens", type=int, default=32)
    ap.add_argument("--fp8-forward", action="store_true",
                    help="whisper finetune step: encoder q|k|v and fc1 forward projections in fp8 (DESIGN.md 4.4)")
    ap.add_argument("--fp8-encoder", action="store_true",
                    help="whisper --decode: encoder q|k|v and fc1 projections with fp8 weights (DESIGN.md 4.4)")
    ap.add_argument("--grad-wire", default="bf16", choices=["bf16", "fp32"],
                    help="dtype of the gradient all-reduce at N>1 (bf16 = DDP bf16_compress_hook equivalent)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for 1-GPU debugging of the N>1 logic)")
    ap.add_argument("--check-replicas", action="store_true", help="after the run, verify every rank holds identical parameters")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    ndev = torch.cuda.device_count()
    local_rank = local_rank % max(1, ndev)  # (gloo debugging may put several ranks on one GPU)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=device)
        else:
            torch.distributed.init_process_group(args.backend)

    from coral_amd import ops, specaugment
    from coral_amd.trainer import DataParallelTrainer
    from coral_amd.wav2vec2 import CORAL_W2V2_SHAPES, Wav2Vec2CTCEngine, Wav2Vec2Shape

    if args.model.startswith("whisper"):
        return whisper_bench(args, world, rank, device)

    # CoRal model YAML values (R/config/model/wav2vec2-large.yaml:12-22); layerdrop is forced to 0
    # in the multi-GPU regime (R/src/scripts/finetune_asr_model.py:48-54) and kept 0 at N=1 so the
    # per-GPU work is identical at every N (weak scaling).
    shape = Wav2Vec2Shape(**CORAL_W2V2_SHAPES[args.model], activation_dropout=0.1, layerdrop=0.0)
    eng = Wav2Vec2CTCEngine(shape, device)
    init_random_(eng, 4242)
    trainer = DataParallelTrainer(eng, learning_rate=1e-4, betas=(0.9, 0.98), max_grad_norm=1.0,
                                  warmup_steps=1000, max_steps=100_000, compress_grads=(args.grad_wire == "bf16"))
    batch, lens = synth_batch(args.batch, args.seconds, rank, device)
    B, N = batch["input_values"].shape
    Ts = eng.conv_lengths(N)
    T = Ts[-1]
    import numpy as np

    rng = np.random.RandomState(4242 + rank)

    pipe = None
    if args.from_host_pcm:
        from coral_amd.input_pipeline import DeviceInputPipeline

        pipe = DeviceInputPipeline(device, B, N, dtype=np.int16, padding="max_length")
        pcm = [(np.clip(0.1 * rng.randn(N), -1, 1) * 32767).astype(np.int16) for _ in range(B)]
        pipe.submit(pcm)

    def make_step_batch():
        mb = dict(batch)
        if pipe is not None:  # this step's batch was staged during the previous step; stage the next one now
            mb.update(pipe.get())
            pipe.submit(pcm)
        if not args.no_specaugment:
            mt, mf = specaugment.sample_masks(B, T, shape.hidden_size, [T] * B, 0.5, 10, 0.5, 64, rng=rng)
            mb["mask_time"] = torch.from_numpy(mt)
            mb["mask_feature"] = torch.from_numpy(mf)
        return [mb]

    def sync():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        loss = trainer.train_step(make_step_batch())
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = trainer.train_step(make_step_batch())
    sync()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device=device)
    if world > 1:
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
    dt = float(tmax.item())
    loss_val = float(loss)

    # live roofline: two more steps with every GEMM launch bracketed by hipEvents on its stream
    ops.prof_begin()
    for _ in range(2):
        trainer.train_step(make_step_batch())
    torch.cuda.synchronize()
    prof = ops.prof_end()
    dom = max(prof, key=lambda r: r["ms"])
    tot_ms = sum(r["ms"] for r in prof)
    tot_fl = sum(r["flops"] for r in prof)

    if rank == 0 and args.gemm_breakdown:
        for r in sorted(prof, key=lambda r: -r["ms"]):
            if r["count"]:
                print(f"{r['kernel']:55s} {r['count'] // 2:5d} launches/step {r['ms'] / 2:8.2f} ms/step "
                      f"{r['flops'] / (r['ms'] * 1e-3) / 1e12:7.1f} TFLOP/s", file=sys.stderr)
    if rank == 0:
        audio_s = world * B * args.seconds * args.steps
        step_ms = dt / args.steps * 1e3
        fwd = fwd_gflop_per_utt(shape, T, Ts)
        step_tflop = 3.0 * fwd * B / 1e3
        ach = dom["flops"] / (dom["ms"] * 1e-3) / 1e12 if dom["ms"] > 0 else 0.0
        traffic = None
        pmc = ROOT / "profiles" / "pmc_traffic.json"
        if pmc.exists():
            try:
                traffic = json.loads(pmc.read_text()).get(dom["kernel"])
            except Exception:
                traffic = None
        out = {
            "metric": "audio-seconds/sec (CTC finetune step: fwd+bwd+allreduce+clip+AdamW), wav2vec2-large, 10 s utterances",
            "value": round(audio_s / dt, 2), "unit": "audio-seconds/sec", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(step_ms, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
            "data": "synthetic",
            "config": {"workload": f"{args.model} (XLS-R shape d={shape.hidden_size} L={shape.num_hidden_layers} "
                                   f"ffn={shape.intermediate_size}) CTC finetune, {B} x {args.seconds:g} s per GPU, "
                                   f"SpecAugment {'off' if args.no_specaugment else 'on'}, activation_dropout 0.1, "
                                   "layerdrop 0 (multi-GPU rule), fp32 master + AdamW + clip 1.0"
                                   + (", inputs from host int16 PCM through the device input pipeline" if pipe is not None else "")
                                   + (f", gradient all-reduce on {args.grad_wire} wire, per-layer buckets overlapped with backward" if world > 1 else ""),
                       "global_batch": world * B, "frames_per_utt": T, "parallelism": f"dp{world}",
                       "loss": round(loss_val, 3)},
            "roofline": {"bound": "mfma", "kernel": dom["kernel"], "achieved": round(ach, 1),
                         "peak": MFMA_BF16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(ach / MFMA_BF16_DENSE_PEAK_TFLOPS, 4), "traffic": traffic,
                         "launches": dom["count"] // 2, "avg_us": round(dom["ms"] * 1e3 / max(1, dom["count"]), 2),
                         "all_gemm_tflops": round(tot_fl / (tot_ms * 1e-3) / 1e12, 1) if tot_ms else 0.0,
                         "gemm_ms_per_step": round(tot_ms / 2, 2),
                         "step_algorithmic_tflop": round(step_tflop, 2),
                         "step_frac_of_peak": round(step_tflop / (step_ms * 1e-3) / MFMA_BF16_DENSE_PEAK_TFLOPS, 4)},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.model)
        print(json.dumps(out), flush=True)
    if world > 1:
        if args.check_replicas:
            # DDP invariant: identical parameters on every rank after identical (averaged) updates
            p = eng.store.p32
            lo, hi = p.clone(), p.clone()
            torch.distributed.all_reduce(lo, op=torch.distributed.ReduceOp.MIN)
            torch.distributed.all_reduce(hi, op=torch.distributed.ReduceOp.MAX)
            spread = float((hi - lo).abs().max())
            if rank == 0:
                print(json.dumps({"replica_param_spread": spread, "loss_rank0": loss_val}), flush=True)
            assert spread == 0.0, f"replicas diverged: {spread}"
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
