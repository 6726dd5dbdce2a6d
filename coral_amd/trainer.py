"""Data-parallel CTC finetuning step: the MI355X replacement for the inner loop CoRal gets from
`transformers.Trainer` + accelerate/DDP ($TF/trainer.py:1678-1800 `_run_epoch`, :1892-1963
`training_step`, :1778-1796 clip + optimizer step; launched by R/src/coral/finetune.py:60-79).

One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI).  Utterances are
sharded across ranks; the only exchange is the gradient all-reduce, issued per parameter bucket
(head, layer L-1 ... layer 0, front — contiguous slices of the flat fp32 gradient buffer) on a
dedicated HIP stream from the engine's backward hooks, so communication of layer l overlaps the
backward of layers < l.  DDP semantics are kept: gradients are averaged over ranks
(sum all-reduce, then 1/world folded into the fused AdamW kernel), the global-norm clip and the
AdamW update (beta=(0.9, 0.98), cosine schedule with warm-up; R/src/coral/wav2vec2.py:216-240,
R/config/asr_finetuning.yaml:64-75) run replicated on every rank.
"""

from __future__ import annotations

import math
import os

import torch

from . import ops


def cosine_lr(step: int, base_lr: float, warmup_steps: int, max_steps: int) -> float:
    """transformers.get_cosine_schedule_with_warmup (SchedulerType.COSINE,
    R/src/coral/wav2vec2.py:217): `step` = number of optimiser steps already taken."""
    if step < warmup_steps:
        return base_lr * step / max(1, warmup_steps)
    progress = (step - warmup_steps) / max(1, max_steps - warmup_steps)
    return base_lr * max(0.0, 0.5 * (1.0 + math.cos(math.pi * progress)))


def grad_accumulation_steps(total_batch_size: int, num_devices: int, per_device_batch_size: int) -> int:
    """R/src/coral/wav2vec2.py:159-181: total // devices // per-device, at least 1."""
    return max(1, total_batch_size // max(1, num_devices) // per_device_batch_size)


class _Handles:
    """Several asynchronous collectives waited for as one."""

    def __init__(self, hs):
        self.hs = hs

    def wait(self):
        for h in self.hs:
            h.wait()


class _Done:
    """Handle of a collective that is ordered by the communication stream itself (the C-ABI path)."""

    def wait(self):
        pass


class CommContext:
    """One RCCL communicator + its communication stream behind the C ABI (include/coral_amd.h: ca_comm_*).  The
    rendezvous - handing rank 0's 128-byte unique id to every rank - rides on the torch.distributed group that launched
    the ranks; every collective afterwards is a ca_* call enqueued on the context's own HIP stream."""

    def __init__(self, device, ident: bytes, rank: int, world: int):
        """`ident`: rank 0's 128-byte unique id (`new_unique_id()`), handed to every rank by the caller."""
        import ctypes as C

        lib = ops.lib()
        self.rank, self.world = rank, world
        self.ctx = None
        with torch.cuda.device(device):
            ctx = C.c_void_p()
            ops.check(lib.ca_comm_init(C.byref(ctx), ident, rank, world), "ca_comm_init")
        self.ctx = ctx
        self.lib = lib
        self.stream = torch.cuda.ExternalStream(lib.ca_comm_stream(ctx), device=device)

    @staticmethod
    def new_unique_id() -> bytes:
        """Binds RCCL through the C ABI (ca_comm_unique_id) and returns a fresh id; raises where it cannot."""
        import ctypes as C

        buf = C.create_string_buffer(128)
        ops.check(ops.lib().ca_comm_unique_id(buf), "ca_comm_unique_id")
        return bytes(buf.raw)

    def close(self, abort: bool = False):
        """Give the communicator and its stream back (ca_comm_destroy: waits for the stream first; `abort`:
        ca_comm_abort, for a context whose peers never joined a collective)."""
        ctx, self.ctx = self.ctx, None
        if ctx is not None:
            (self.lib.ca_comm_abort if abort else self.lib.ca_comm_destroy)(ctx)

    def __del__(self):
        # Not `close()`: ca_comm_destroy waits for the context's stream, and a garbage-collected context (or one collected
        # at interpreter shutdown) may still have a collective in flight whose peers are gone - that wait would never end.
        # An owner that is done with a context calls close(); a context that is merely dropped is left to the process's end.
        self.ctx = None

    @staticmethod
    def _dt(t):
        return {torch.float32: 0, torch.bfloat16: 1}[t.dtype]

    def after_current(self):
        """The communication stream waits for what is enqueued on torch's current stream."""
        ops.check(self.lib.ca_comm_after(self.ctx, torch.cuda.current_stream().cuda_stream), "ca_comm_after")

    def before_current(self):
        """torch's current stream waits for the collectives enqueued so far."""
        ops.check(self.lib.ca_comm_before(self.ctx, torch.cuda.current_stream().cuda_stream), "ca_comm_before")

    def all_reduce(self, t):
        ops.check(self.lib.ca_allreduce_bucket(self.ctx, t.data_ptr(), t.numel(), self._dt(t)), "ca_allreduce_bucket")

    def reduce_scatter(self, t):
        """In place over t (world equal slices): this rank's slice ends with the sum over ranks."""
        ops.check(self.lib.ca_reduce_scatter_bucket(self.ctx, t.data_ptr(), t.numel() // self.world, self._dt(t)),
                  "ca_reduce_scatter_bucket")

    def all_gather(self, t):
        """In place over t: every rank's slice -> all of t on every rank."""
        ops.check(self.lib.ca_allgather_bucket(self.ctx, t.data_ptr(), t.numel() // self.world, self._dt(t)),
                  "ca_allgather_bucket")


class GradSync:
    """Bucketed gradient all-reduce over a flat fp32 buffer (device-agnostic, so the N>1 logic is
    testable with gloo on CPU).  `start(name)` launches the asynchronous SUM all-reduce of one
    contiguous bucket — on a side HIP stream when the buffer lives on a GPU — and `finish()` waits
    for all of them.  The 1/world averaging is NOT applied here (it is folded into the fused AdamW
    kernel / `scale_`), so the wire carries plain sums like DDP's buckets.

    compress=True sends bf16 on the wire (the DDP `bf16_compress_hook` trade: half the xGMI bytes —
    4.3 GB instead of 8.6 GB per step for XLS-R-2B — for one bf16 rounding of each rank's gradient);
    the fp32 buffer is refilled from the reduced bf16 values."""

    def __init__(self, flat_grad: torch.Tensor, buckets: dict, process_group=None, compress: bool = False,
                 force: bool = False, shard: dict | None = None):
        """shard: {bucket name: (mlo, hi)} - the part of a bucket that is REDUCE-SCATTERED instead of all-reduced
        (sharded optimiser, see DataParallelTrainer zero_stage): rank r ends up with the sum over ranks of its slice
        [mlo + r * per, mlo + (r + 1) * per), per = (hi - mlo) / world, and nothing defined elsewhere in [mlo, hi); the
        rest of the bucket [lo, mlo) is all-reduced as before."""
        self.g = flat_grad
        self.buckets = buckets
        self.pg = process_group
        self.world = 1
        self.rank = 0
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            self.world = torch.distributed.get_world_size(process_group)
            self.rank = torch.distributed.get_rank(process_group)
        self.shard = dict(shard or {})
        for name, (mlo, hi) in self.shard.items():
            lo, bhi = buckets[name]
            if not (lo <= mlo <= hi == bhi) or (hi - mlo) % (8 * max(1, self.world)):
                raise ValueError(f"shard range of bucket {name!r} must end the bucket and divide into 32-byte aligned slices")
        # reduce_scatter_tensor exists on RCCL; the CPU test backend (gloo) has none: there the slice is cut out of an
        # all-reduce (the same sums; at two ranks the same bits)
        self._has_rs = (torch.distributed.is_available() and torch.distributed.is_initialized()
                        and torch.distributed.get_backend(process_group) == "nccl")
        # force (or CA_DP_FORCE=1): run the whole exchange path with a process group of one rank - how the RCCL
        # plumbing (communication stream, asynchronous handles, per-bucket callbacks) is exercised on a 1-GPU box
        force = force or os.environ.get("CA_DP_FORCE", "0") == "1"
        self.active = self.world > 1 or (force and torch.distributed.is_available() and torch.distributed.is_initialized())
        self.on_gpu = flat_grad.is_cuda
        # RCCL ranks exchange through the C ABI (ca_allreduce_bucket / ca_reduce_scatter_bucket on the context's own
        # stream: include/coral_amd.h); torch.distributed collectives remain for the gloo CPU tests (CA_COMM_CAPI=0
        # forces them on RCCL too: the A/B switch)
        self.capi = None
        if self.active and self.on_gpu and self._has_rs and os.environ.get("CA_COMM_CAPI", "1") != "0":
            self.capi = self._make_capi(flat_grad.device, process_group)
        if self.capi is not None:
            self.comm_stream = self.capi.stream
        else:
            self.comm_stream = ops.side_stream(flat_grad.device, "exchange") if (self.on_gpu and self.active) else None
        self.compress = compress and self.active
        self.g16 = torch.empty_like(flat_grad, dtype=torch.bfloat16) if self.compress else None
        self._pending = []
        self.launched: list[str] = []

    @staticmethod
    def _make_capi(device, pg):
        """A CommContext all ranks agree on, or None (every rank then uses torch.distributed's collectives).  The context
        is created and, with more than one rank, checked once against torch.distributed on a small buffer; the ranks
        exchange their verdicts, so that one rank's failure moves ALL of them to the fallback - never a mixed group."""
        import logging

        dist = torch.distributed
        world, rank = dist.get_world_size(pg), dist.get_rank(pg)
        log = logging.getLogger(__package__)

        def agree(ok: bool) -> bool:
            """True only if EVERY rank says so - exchanged over torch.distributed's own communicator on the current
            stream, which at no point of this function waits for the context under test."""
            if world == 1:
                return ok
            flag = torch.tensor([1.0 if ok else 0.0], device=device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=pg)
            return bool(flag.item() > 0.5)

        def fallback(ctx=None, abort=False):
            if ctx is not None:
                ctx.close(abort=abort)
            log.warning("gradient exchange falls back to torch.distributed's RCCL collectives")
            return None

        # 1. every rank binds RCCL through the C ABI (its own throw-away id proves it); rank 0's id is the one used.
        #    The broadcast happens WHATEVER rank 0 found - its id or a None sentinel - so no rank is left waiting in it.
        ident, ok = None, True
        try:
            ident = CommContext.new_unique_id()
        except Exception as e:  # noqa: BLE001
            log.warning("C-ABI collectives unavailable (%s)", e)
            ok = False
        box = [ident]
        if world > 1:
            src = dist.get_global_rank(pg, 0) if pg is not None else 0
            dist.broadcast_object_list(box, src=src, group=pg)
        if not agree(ok and box[0] is not None):
            return fallback()
        # 2. the communicator (collective: every rank passed step 1, so every rank calls it)
        ctx = None
        try:
            ctx = CommContext(device, box[0], rank, world)
        except Exception as e:  # noqa: BLE001
            log.warning("C-ABI communicator not created (%s)", e)
            ok = False
        if not agree(ok):
            return fallback(ctx, abort=True)
        # 3. one all-reduce against torch.distributed's.  The verdict on the ENQUEUE is exchanged before anything
        #    waits for the context's stream: a rank that could not enqueue leaves its peers' kernels spinning, and
        #    those are aborted, not waited for.
        if world > 1:
            x = (torch.arange(4096, dtype=torch.float32, device=device) % 61) * (rank + 1)
            want = x.clone()
            dist.all_reduce(want, group=pg)
            try:
                ctx.after_current()
                ctx.all_reduce(x)
            except Exception as e:  # noqa: BLE001
                log.warning("C-ABI all-reduce not enqueued (%s)", e)
                ok = False
            if not agree(ok):
                return fallback(ctx, abort=True)
            ctx.before_current()
            if not agree(bool(torch.equal(x, want))):
                return fallback(ctx)
        return ctx

    def _to_wire(self, lo, hi):
        if self.on_gpu:
            ops.cast_f32_bf16(self.g[lo:hi], self.g16[lo:hi], hi - lo)  # enqueued on the current (= comm) stream
        else:
            self.g16[lo:hi].copy_(self.g[lo:hi])

    def _from_wire(self, lo, hi):
        if self.on_gpu:
            ops.cast_bf16_f32(self.g16[lo:hi], self.g[lo:hi], hi - lo)
        else:
            self.g[lo:hi].copy_(self.g16[lo:hi])

    def slice_of(self, name: str):
        """This rank's slice (a, b) of bucket `name`'s sharded part."""
        mlo, hi = self.shard[name]
        per = (hi - mlo) // self.world
        return mlo + self.rank * per, mlo + (self.rank + 1) * per

    def _launch(self, lo, hi, name=None):
        buf = self.g
        if self.compress:
            self._to_wire(lo, hi)
            buf = self.g16
        SUM = torch.distributed.ReduceOp.SUM
        if self.capi is not None:  # (called with the communication stream current: ordered by the stream itself)
            if name is None or name not in self.shard:
                self.capi.all_reduce(buf[lo:hi])
            else:
                mlo, _ = self.shard[name]
                if mlo > lo:
                    self.capi.all_reduce(buf[lo:mlo])
                self.capi.reduce_scatter(buf[mlo:hi])
            return _Done()
        if name is None or name not in self.shard:
            return torch.distributed.all_reduce(buf[lo:hi], op=SUM, group=self.pg, async_op=True)
        mlo, _ = self.shard[name]
        hs = []
        if mlo > lo:
            hs.append(torch.distributed.all_reduce(buf[lo:mlo], op=SUM, group=self.pg, async_op=True))
        a, b = self.slice_of(name)
        if self._has_rs:  # in place: the output slice sits at input + rank * count, RCCL's in-place form
            hs.append(torch.distributed.reduce_scatter_tensor(buf[a:b], buf[mlo:hi], op=SUM, group=self.pg, async_op=True))
        else:
            hs.append(torch.distributed.all_reduce(buf[mlo:hi], op=SUM, group=self.pg, async_op=True))
        return _Handles(hs)

    def start(self, name: str, post=None):
        """post(name): run on the communication stream as soon as the bucket's reduced gradients are in
        the fp32 buffer (the trainer computes the bucket's squared norm there, off the critical path)."""
        if not self.active:
            return
        lo, hi = self.buckets[name]
        self.launched.append(name)

        def go():
            h = self._launch(lo, hi, name)
            if post is None:
                return h
            h.wait()  # stream-level wait on a GPU, blocking on the CPU backends
            if self.compress:
                self._from_wire(lo, hi)
            post(name)
            return None

        if self.comm_stream is not None:
            if self.capi is not None:
                self.capi.after_current()  # (ca_comm_after: the bucket's grads are enqueued)
            else:
                self.comm_stream.wait_stream(torch.cuda.current_stream())  # the bucket's grads are enqueued
            with torch.cuda.stream(self.comm_stream):
                h = go()
        else:
            h = go()
        self._pending.append((h, lo, hi))

    def start_all(self):
        for name in self.buckets:
            self.start(name)

    def finish(self):
        def drain():
            for h, lo, hi in self._pending:
                if h is None:
                    continue  # completed on the communication stream in start()
                h.wait()
                if self.compress:
                    self._from_wire(lo, hi)

        if self.comm_stream is not None:
            with torch.cuda.stream(self.comm_stream):
                drain()
            if self.capi is not None:
                self.capi.before_current()
            else:
                torch.cuda.current_stream().wait_stream(self.comm_stream)
        else:
            drain()
        self._pending.clear()
        self.launched.clear()

    def close(self, abort: bool = False):
        """Give the C-ABI communicator (and its stream) back; `abort`: without waiting for its stream."""
        if self.capi is not None:
            self.capi.close(abort=abort)
            self.capi = None
            self.comm_stream = None  # (it wrapped the communicator's HIP stream, gone with it)

    def scale_(self):
        """DDP-mean semantics for host-side consumers: g /= world."""
        if self.world > 1:
            self.g.mul_(1.0 / self.world)


def shard_indices(n_items: int, rank: int, world: int) -> list[int]:
    """Utterance indices of `rank` for one global batch: contiguous equal shards (the per-device
    batches accelerate hands each DDP rank); n_items must be divisible by world."""
    if n_items % world:
        raise ValueError(f"global batch {n_items} is not divisible by world size {world}")
    per = n_items // world
    return list(range(rank * per, (rank + 1) * per))


class DataParallelTrainer:
    """Owns optimiser state (flat fp32 m, v) and the communication stream for one engine."""

    def __init__(self, engine, learning_rate=1e-4, betas=(0.9, 0.98), eps=1e-8, weight_decay=0.0,
                 max_grad_norm=1.0, warmup_steps=1000, max_steps=100_000, grad_accum=1,
                 process_group=None, overlap=True, compress_grads=False, overlap_optimizer=True, zero_stage=0,
                 accum_loss: str = "mean"):
        """zero_stage > 0 (N > 1): the reference's production launch mode (`accelerate launch --use-deepspeed
        --zero-stage 2`, R/makefile:79-84,94-99,109-114) in this engine's terms - the weight-matrix part of every layer
        bucket (> 99 % of the parameters) is REDUCE-SCATTERED instead of all-reduced, each rank keeps AdamW moments for
        and updates only its 1/N slice, and the bf16 compute copy of the slices is all-gathered bucket by bucket under
        the next forward.  Per step and rank at XLS-R-2B, N = 8: AdamW traffic 64.8 -> 8.1 GB, wire 2 x 7/8 x 8.64 GB
        (all-reduce) -> 7/8 x 8.64 GB (reduce-scatter, fp32) + 7/8 x 4.32 GB (all-gather, bf16).  The small tensors of
        a layer, the front and the head bucket stay replicated.  The update is the same arithmetic on the same sums:
        parameters equal the replicated trainer's (tests/test_dp_gloo.py, tests/test_dp_gpu.py)."""
        self.model = engine                          # HF-shaped wrapper or the bare engine
        engine = getattr(engine, "engine", engine)  # the kernel-sequencing engine underneath
        self.engine = engine
        self.lr, self.betas, self.eps, self.wd = learning_rate, betas, eps, weight_decay
        self.max_grad_norm = max_grad_norm
        self.warmup_steps, self.max_steps = warmup_steps, max_steps
        self.grad_accum = grad_accum
        # How the micro-batches of one optimiser step combine.  "mean": loss and gradients of each micro-batch scaled by
        # 1 / grad_accum (Trainer.training_step for a model WITHOUT **kwargs in its forward, $TF/trainer.py:1952-1954).
        # "sum": no scaling - what transformers >= 4.46 (the reference pins 5.5.0) does for Wav2Vec2ForCTC and
        # WhisperForConditionalGeneration, whose forwards take **kwargs (`model_accepts_loss_kwargs`) while their own
        # losses ignore `num_items_in_batch`: the logged loss is the SUM over the micro-batches and so is the gradient
        # the clip sees (tools/gen_goldens.py trainer_traj pins it; CoralTrainer's default).
        if accum_loss not in ("mean", "sum"):
            raise ValueError(f"accum_loss must be 'mean' or 'sum', not {accum_loss!r}")
        self.accum_loss = accum_loss
        self.opt_step = 0
        st = engine.store
        if hasattr(engine, "trainable_range"):
            lo, hi = engine.trainable_range()
        else:
            lo, hi = st.buckets["head"] if engine.freeze_base else (0, st.numel)
        self.train_range = (lo, hi)
        world = 1
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            world = torch.distributed.get_world_size(process_group)
        zero_stage = int(os.environ.get("CA_ZERO_STAGE", zero_stage) or 0)
        shard = {}
        if zero_stage and (world > 1 or os.environ.get("CA_DP_FORCE", "0") == "1") and hasattr(engine, "shard_ranges") \
                and (lo, hi) == (0, st.numel):
            shard = {n: r for n, r in engine.shard_ranges().items() if (r[1] - r[0]) % (8 * world) == 0 and r[1] > r[0]}
        self.sync = GradSync(st.g32, st.buckets, process_group, compress=compress_grads, shard=shard)
        self.zero = bool(shard) and self.sync.active
        if self.zero and world > 1 and st.p32.is_cuda and not self._collectives_selfcheck(st.device, process_group, self.sync.capi):
            # (never silently wrong: a backend whose in-place reduce-scatter / all-gather does not behave as RCCL
            # documents falls back to the replicated path, and says so)
            import logging

            logging.getLogger(__package__).warning("sharded optimiser: the collective self-check failed; using replicated DDP")
            self.sync.close(abort=True)  # (a rank that could not enqueue leaves its peers' kernels spinning)
            self.sync = GradSync(st.g32, st.buckets, process_group, compress=compress_grads)
            self.zero = False
        self.world = self.sync.world
        self.dist = self.sync.active  # gradients are exchanged (world > 1, or a forced one-rank group)
        if self.dist and st.p32.is_cuda:
            # A collective's kernel holds CUs for most of the backward: the persistent GEMM launches hand out every tile
            # dynamically (a workgroup that only starts once others have exited finds nothing left and exits) and, with
            # CA_COMPUTE_CUS=<n>, size themselves and the tile-shape rule to n CUs (include/coral_amd.h:
            # ca_gemm_set_compute_cus; measured beside an emulated ring kernel: tools/r05_hog_gemm.py)
            ncu = torch.cuda.get_device_properties(st.device).multi_processor_count
            ops.lib().ca_gemm_set_compute_cus(int(os.environ.get("CA_COMPUTE_CUS", ncu)))
            self._set_compute_cus = True  # (process-global library state: close() puts the default back)
        self.overlap = overlap and self.dist
        # AdamW moments.  Replicated: the parameters' own offsets.  Sharded: a compact buffer holding, bucket by bucket,
        # the replicated part [lo, mlo) and this rank's slice of the sharded part - 1/N of the state.
        self._state_off = None
        self._ag_stream = None
        self._ag_comm = None
        if self.zero:
            self._state_off, n = {}, 0
            for name, (blo, bhi) in st.buckets.items():
                if name in shard:
                    a, b = self.sync.slice_of(name)
                    self._state_off[name] = (n, n + (shard[name][0] - blo))
                    n += (shard[name][0] - blo) + (b - a)
                else:
                    self._state_off[name] = (n, None)
                    n += bhi - blo
            self.m = torch.zeros(n, dtype=torch.float32, device=st.device)
            self.v = torch.zeros(n, dtype=torch.float32, device=st.device)
        else:
            self.m = torch.zeros_like(st.p32)
            self.v = torch.zeros_like(st.p32)
        self.gnorm_sq = torch.zeros(1, dtype=torch.float32, device=st.device)
        self.partial = torch.zeros(4096, dtype=torch.float32, device=st.device)
        # AdamW is HBM-bound, the next forward MFMA-bound: run the update bucket by bucket on a side stream
        # and let the next forward wait per bucket (engine._await) instead of for the whole optimiser.
        # (CA_OPT_OVERLAP=0: everything on one stream - the regime the per-kernel profiles are taken in)
        self.overlap_optimizer = (overlap_optimizer and os.environ.get("CA_OPT_OVERLAP", "1") != "0"
                                  and st.p32.is_cuda and hasattr(engine, "_await")
                                  and (lo, hi) == (0, st.numel))
        self.opt_stream = None
        if self.overlap_optimizer:
            # the HBM-bound update only fills what the MFMA-bound forward leaves free: lowest queue priority
            prio = int(os.environ.get("CA_OPT_PRIO", "0"))
            self.opt_stream = ops.side_stream(st.device, "optimizer", prio)
        self.opt_done = None
        # ... and as a BACKGROUND kernel: one workgroup per CU, whose waves fit beside the forward GEMMs' in the register
        # file, so the two really run at the same time (ca_adamw_step_ex; CA_OPT_BG_BLOCKS=0: full grid - the A/B switch)
        self.bg_blocks = 0
        if self.overlap_optimizer:
            bg = os.environ.get("CA_OPT_BG_BLOCKS")
            self.bg_blocks = int(bg) if bg is not None else torch.cuda.get_device_properties(st.device).multi_processor_count
            if bg is None:
                # ... which holds only while the loaded kernels' register counts say so (a compiler that gives the forward
                # kernel 8 registers more turns the capped update into a 19 ms stall, tests/test_build.py): asked of the
                # code object itself; otherwise the full-grid update, which alternates with the GEMMs
                fits, rx, ru = ops.background_update_fits()
                if not fits:
                    import warnings

                    warnings.warn(f"coral_amd: the forward GEMM kernel holds {rx} registers per lane and the update "
                                  f"kernel {ru}: the update cannot run beside it; using the full-grid update")
                    self.bg_blocks = 0
        # per-bucket squared gradient norms, computed on the side stream as the buckets complete
        self.bucket_index = {name: i for i, name in enumerate(st.buckets)}
        self.bucket_sq = torch.zeros(len(st.buckets), dtype=torch.float32, device=st.device)
        self.shard_norm = os.environ.get("CA_SHARD_NORM", "1") != "0"
        self._rank = self.sync.rank
        self._norms_ready = False
        # N = 1: the squared norm of the gradients that are final early in the backward is taken on the side stream
        # while the rest of the backward runs (one HBM-bound pass beside MFMA-bound GEMMs), see _early_norm
        self.partial_early = torch.zeros(4096, dtype=torch.float32, device=st.device)
        self.early_fraction = float(os.environ.get("CA_EARLY_NORM", "0.8"))
        self._early_lo = None   # [self._early_lo, hi) is already inside gnorm_sq
        self._done = {}

    # ---- one optimiser step ----------------------------------------------------------------------
    def train_step(self, micro_batches) -> float | torch.Tensor:
        """micro_batches: list (len = grad_accum) of dicts with input_values / attention_mask /
        labels (+ optional mask_time / mask_feature / layer_keep).  Returns the loss tensor (device) of this rank
        over the micro-batches, combined as `accum_loss` says (see __init__)."""
        eng = self.engine
        self.model.train()
        total = None
        n = len(micro_batches)
        rank = int(os.environ.get("RANK", "0")) % 64
        # One micro-batch per optimiser step on one GPU: the layers' weight-matrix gradients stay bf16 - the dtype the
        # reference's autocast computes them in ($TF/trainer.py training_step under torch.autocast: a Linear's weight
        # gradient is the bf16 output of a bf16 matmul, cast to fp32 only when it is accumulated into .grad) - from the
        # weight-gradient GEMM's epilogue to AdamW.  With accumulation or several ranks the fp32 buffer is what is
        # accumulated into / reduced, as in the reference.  CA_WGRAD_BF16=0: fp32 always.
        if hasattr(eng, "bf16_grad_ranges") and hasattr(type(eng.store), "g16"):  # (the class: the property allocates on first read)
            eng.wgrad_bf16 = (n == 1 and not self.dist and not self.zero and not eng.freeze_base
                              and os.environ.get("CA_WGRAD_BF16", "1") != "0" and self._norm_plan() is not None)
        for i, mb in enumerate(micro_batches):
            # fresh activation-dropout masks per micro-batch and rank (HF draws new masks in every forward); the
            # backward of micro-batch i runs before the next forward and regenerates the masks from the same seed
            eng.step_seed = (self.opt_step * n + i) * 64 + rank
            out = self.model(**mb)
            if i == 0:  # the previous optimiser step may still be reading the gradients
                self.finish()
                eng.zero_grad(matrices=eng.freeze_base)
            last = i == n - 1
            hook = None
            self._norms_ready = False
            # Per-bucket norms during the backward pay off where the buckets are being all-reduced anyway (N > 1: the
            # squared norm rides behind each bucket's reduction on the communication stream).  At N = 1 the side-stream
            # kernels only contend with the backward GEMMs for HBM (+20 % on the weight-gradient kernel for 0.2 ms).
            if last and self.dist and self.overlap and (self.overlap_optimizer or self.zero):
                hook = self._bucket_ready
                self._norms_ready = True
            elif self.dist and last and self.overlap:
                hook = self.sync.start
            elif not self.dist and self._norm_plan() is not None:
                hook = None  # the squared norm comes out of the weight-gradient GEMMs (optimizer_step)
            elif not self.dist and last and self.overlap_optimizer and self.early_fraction > 0:
                self._done, self._early_lo = {}, None
                hook = self._early_norm
            scale = 1.0 / n if self.accum_loss == "mean" else 1.0
            eng.backward(loss_scale=scale, overwrite_matrices=(i == 0), bucket_done=hook)
            loss = out.loss.detach()  # (the autograd route of coral_amd/autograd.py is not used here)
            total = loss * scale if total is None else total + loss * scale
        if self.dist and not self.overlap:
            self.sync.start_all()
        self.sync.finish()
        self.optimizer_step()
        if hasattr(eng, "wgrad_bf16"):
            # the choice belongs to THIS step: a later direct engine.backward(overwrite_matrices=True) (the autograd
            # route, tests reading store.g32) must find fp32 matrix gradients, not stale ones beside a bf16 buffer
            eng.wgrad_bf16 = False
        return total

    def _norm_plan(self):
        """The engine's plan for a gradient norm without a pass over the weight matrices (wav2vec2 engine, whole
        model trainable, CA_FUSED_NORM != 0), else None."""
        if os.environ.get("CA_FUSED_NORM", "1") == "0" or not hasattr(self.engine, "norm_plan"):
            return None
        if self.train_range != (0, self.engine.store.numel):
            return None
        return self.engine.norm_plan()

    def _bucket_sumsq(self, name: str):
        """Squared norm of a bucket's (reduced) gradients.  With several ranks every rank holds the same reduced
        gradients, so each takes the squared norm of its 1/world slice of the bucket only and the slices' sums are
        added over ranks in optimizer_step (one scalar all-reduce, identical on every rank): 1/world of the bytes."""
        lo, hi = self.engine.store.buckets[name]
        i = self.bucket_index[name]
        if self.zero and name in self.sync.shard:
            # sharded bucket: this rank's slice of the matrices + its 1/world piece of the replicated small tensors
            g = self.engine.store.g32
            a, b = self.sync.slice_of(name)
            ops.sumsq(g[a:b], b - a, self.bucket_sq[i:i + 1], self.partial)
            mlo = self.sync.shard[name][0]
            per = (-(-(mlo - lo) // self.world) + 7) // 8 * 8
            a2 = min(mlo, lo + self._rank * per)
            b2 = min(mlo, a2 + per)
            if b2 > a2:
                ops.sumsq(g[a2:b2], b2 - a2, self.bucket_sq[i:i + 1], self.partial, accumulate=True)
            return
        if self.world > 1 and (self.shard_norm or self.zero):
            per = -(-(hi - lo) // self.world)
            per = (per + 7) // 8 * 8  # slices start 32-byte aligned
            a = min(hi, lo + self._rank * per)
            b = min(hi, a + per)
            if b > a:
                ops.sumsq(self.engine.store.g32[a:b], b - a, self.bucket_sq[i:i + 1], self.partial)
            else:
                ops.clear_f32(self.bucket_sq, 1, off=i)
            return
        ops.sumsq(self.engine.store.g32[lo:hi], hi - lo, self.bucket_sq[i:i + 1], self.partial)

    def _bucket_ready(self, name: str):
        """Backward hook: all gradients of bucket `name` are enqueued.  N>1: all-reduce it on the
        communication stream, then its squared norm there; N=1: squared norm on the optimiser stream."""
        if self.dist:
            self.sync.start(name, post=self._bucket_sumsq)
            return
        self.opt_stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.opt_stream):
            self._bucket_sumsq(name)

    def _early_norm(self, name: str):
        """Backward hook (N = 1).  Buckets complete roughly from the end of the flat gradient buffer towards its start;
        once the completed tail [x, hi) covers `early_fraction` of the gradients its squared norm is started on the
        side stream (the split depends only on the bucket layout: deterministic) and only the head of the buffer is
        left for the pass behind the backward."""
        if self._early_lo is not None:
            return
        a, b = self.engine.store.buckets[name]
        lo, hi = self.train_range
        self._done[b] = a  # completed interval, keyed by its end
        x = hi
        while x in self._done:
            x = self._done[x]
        if hi - x < self.early_fraction * (hi - lo) or x <= lo:
            return
        self._early_lo = x
        self.opt_stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.opt_stream):
            ops.sumsq(self.engine.store.g32[x:hi], hi - x, self.gnorm_sq, self.partial_early)

    def finish(self):
        """Make the current stream wait for an optimiser step that is still running on the side stream."""
        if self.opt_done is not None:
            torch.cuda.current_stream().wait_event(self.opt_done)
            self.opt_done = None

    def close(self):
        """Teardown: wait for the optimiser, give the C-ABI communicator and its stream back (one RCCL communicator per
        trainer otherwise stays behind - bench workloads and tests build several per process)."""
        self.finish()
        if torch.cuda.is_available() and self.engine.store.p32.is_cuda:
            torch.cuda.synchronize()
        self._ag_comm = None
        self.sync.close()
        if getattr(self, "_set_compute_cus", False):
            ops.lib().ca_gemm_set_compute_cus(0)  # later single-GPU trainers of this process launch as a fresh process would
            self._set_compute_cus = False

    def optimizer_step(self):
        eng, st = self.engine, self.engine.store
        lo, hi = self.train_range
        n = hi - lo
        lr = cosine_lr(self.opt_step, self.lr, self.warmup_steps, self.max_steps)
        self.last_lr = lr  # the rate THIS update uses (what Trainer logs as `learning_rate` for the step)
        self.opt_step += 1
        plan = None if self.dist else self._norm_plan()
        if self.zero and not self._norms_ready:  # (no per-bucket hooks ran: take the sharded norms here)
            for name in st.buckets:
                self._bucket_sumsq(name)
            self._norms_ready = True
        if plan is not None:
            # N = 1: per-tile sums of squares from the weight-gradient GEMMs' epilogues + one pass over the < 1 % of the
            # buffer that is not a layer weight matrix (instead of reading all 8.6 GB of gradients again: 1.5 ms of
            # HBM traffic beside the backward at XLS-R-2B)
            ops.sumsq_ranges(st.g32, plan["chunks"], plan["nchunks"], self.gnorm_sq, plan["partial"])
            ops.sum_f32(plan["slots"], plan["nslots"], self.gnorm_sq, self.partial, accumulate=True)
        elif self._norms_ready:  # every bucket's squared norm was produced during the backward
            if not self.dist and self.opt_stream is not None:
                torch.cuda.current_stream().wait_stream(self.opt_stream)
            self.gnorm_sq.copy_(self.bucket_sq.sum().reshape(1))
            if self.world > 1 and (self.shard_norm or self.zero):  # add the ranks' slices (the same result on every rank)
                if self.sync.capi is not None:
                    self.sync.capi.after_current()
                    self.sync.capi.all_reduce(self.gnorm_sq)
                    self.sync.capi.before_current()
                else:
                    torch.distributed.all_reduce(self.gnorm_sq, op=torch.distributed.ReduceOp.SUM, group=self.sync.pg)
            self._norms_ready = False
        elif self._early_lo is not None:  # the tail's squared norm is already in gnorm_sq (side stream)
            torch.cuda.current_stream().wait_stream(self.opt_stream)
            ops.sumsq(st.g32[lo:self._early_lo], self._early_lo - lo, self.gnorm_sq, self.partial, accumulate=True)
            self._early_lo = None
        else:
            ops.sumsq(st.g32[lo:hi], n, self.gnorm_sq, self.partial)

        def adam(a, b, so, g=None):
            if b > a:
                g = st.g32 if g is None else g
                ops.adamw_step(st.p32[a:b], self.m[so:so + b - a], self.v[so:so + b - a], g[a:b], st.p16[a:b], b - a,
                               lr, self.betas[0], self.betas[1], self.eps, self.wd, self.opt_step,
                               grad_scale=1.0 / self.world, max_norm=self.max_grad_norm, gnorm_sq=self.gnorm_sq,
                               max_blocks=self.bg_blocks if getattr(self.engine, "background_optimizer", True) else 0)

        # this step's weight-matrix gradients are in the bf16 buffer (train_step): [lo, hi) inside a layer's bucket
        g16_ranges = eng.bf16_grad_ranges() if (getattr(eng, "matrix_grads_bf16", False) and getattr(eng, "wgrad_bf16", False)) else {}

        def update(a, b, name=None):
            if not self.zero:
                if g16_ranges and name is None:  # (whole-range call: bucket by bucket, the matrices from the bf16 buffer)
                    for bname, (ba, bb) in st.buckets.items():
                        update(max(a, ba), min(b, bb), bname)
                    return None
                if name in g16_ranges:
                    mlo, mhi = g16_ranges[name]
                    adam(a, min(b, mlo), a)
                    adam(max(a, mlo), min(b, mhi), max(a, mlo), st.g16)
                    return adam(max(a, mhi), b, max(a, mhi))
                return adam(a, b, a)
            rep_off, sl_off = self._state_off[name]
            if sl_off is None:
                return adam(a, b, rep_off)
            mlo, bhi = self.sync.shard[name]
            adam(a, mlo, rep_off)           # replicated small tensors
            sa, sb = self.sync.slice_of(name)
            adam(sa, sb, sl_off)            # this rank's slice of the matrices (fp32 master, moments, bf16 copy)
            # the bf16 slices travel on a stream of their own: the AdamW of the next bucket does not wait for this
            # bucket's all-gather, only the forward's per-bucket wait does (the event below is recorded behind it)
            cur = torch.cuda.current_stream()
            if self._ag_stream is None:
                # C ABI: the gathers ride on the exchange context's own stream - ONE communicator per rank, so no two
                # collectives of this trainer are ever in flight in an order that could differ between ranks (by the
                # time the optimiser runs, the step's reduce-scatters have been waited for: nothing queues in front)
                self._ag_comm = self.sync.capi
                self._ag_stream = self._ag_comm.stream if self._ag_comm is not None else ops.side_stream(st.device, "gather")
            self._ag_stream.wait_stream(cur)
            with torch.cuda.stream(self._ag_stream):
                self._allgather_bf16(mlo, bhi, sa, sb)
            return self._ag_stream

        rebucket = getattr(eng, "refresh_bucket", None)  # engines with derived per-bucket weight copies (fp8)
        if not self.overlap_optimizer:
            if self.zero:
                for name, (a, b) in st.buckets.items():
                    update(a, b, name)
                if self._ag_stream is not None:
                    torch.cuda.current_stream().wait_stream(self._ag_stream)
            else:
                update(lo, hi)
            if not eng.freeze_base:
                eng.refresh_derived()
            if rebucket is not None:
                for name in st.buckets:
                    rebucket(name)
            return
        # buckets in the order the next forward consumes them; one event per bucket
        self.opt_stream.wait_stream(torch.cuda.current_stream())
        events = {}
        with torch.cuda.stream(self.opt_stream):
            order = sorted(st.buckets.items(), key=lambda kv: kv[1][0])
            for name, (a, b) in order:
                gathered_on = update(a, b, name)
                if gathered_on is not None and rebucket is not None:
                    self.opt_stream.wait_stream(gathered_on)  # (derived per-bucket copies read the gathered weights)
                if rebucket is not None:
                    rebucket(name)
                parts = getattr(eng, "derived_parts", None) if name == "front" else None
                if name == "front" and not parts:
                    eng.refresh_derived()
                if parts:
                    eng.refresh_derived(parts[0])
                ev = torch.cuda.Event()
                ev.record(gathered_on if (gathered_on is not None and rebucket is None) else self.opt_stream)
                events[name] = ev
                for part in (parts or ())[1:]:  # later parts of the derived weights get events of their own
                    eng.refresh_derived(part)
                    ev = torch.cuda.Event()
                    ev.record(self.opt_stream)
                    events[f"front_{part}"] = ev
            if "front" not in events:
                eng.refresh_derived()
            if hasattr(eng, "clear_small_grads") and not eng.freeze_base and self.grad_accum >= 1:
                # the gradients are consumed: clear what the next backward accumulates into (front / head buckets and
                # the layers' small tensors) here, under the next forward
                eng.clear_small_grads()
                eng._pre_zeroed = True
            if self._ag_stream is not None:
                self.opt_stream.wait_stream(self._ag_stream)  # `finish()` = every gather has landed too
            self.opt_done = torch.cuda.Event()
            self.opt_done.record(self.opt_stream)
        eng.weights_ready = events

    @staticmethod
    def _collectives_selfcheck(device, pg, capi=None) -> bool:
        """One-time check, on the real process group, of the two in-place forms the sharded optimiser relies on:
        reduce-scatter with the output slice inside the input buffer (at input + rank x count) and all-gather with the
        input slice inside the output buffer, against their defining sums - through the C ABI's ca_* collectives when
        `capi` (a CommContext) is given, else torch.distributed's."""
        dist = torch.distributed
        world, rank = dist.get_world_size(pg), dist.get_rank(pg)
        n = 4096
        base = torch.arange(world * n, dtype=torch.float32, device=device) % 257
        x = base * (rank + 1)
        want = base * (world * (world + 1) // 2)

        def agree(ok: bool) -> bool:
            """Every rank's verdict, over torch.distributed's own communicator (never the context under test)."""
            flag = torch.tensor([1.0 if ok else 0.0], device=device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=pg)
            return bool(flag.item() > 0.5)

        if capi is None and dist.get_backend(pg) != "nccl":
            return True  # (gloo: GradSync cuts the slices out of all-reduces, nothing in place to check)
        if capi is not None:
            # C ABI: enqueue under a guard, agree that EVERY rank enqueued, only then wait for the context's stream (a
            # rank whose ca_* call raised has enqueued nothing: its peers' kernels would spin, and so would a verdict
            # exchanged behind them)
            for which in ("reduce_scatter", "all_gather"):
                ok = True
                try:
                    if which == "reduce_scatter":
                        buf = x.clone()
                        capi.after_current()
                        capi.reduce_scatter(buf)
                    else:
                        buf = torch.zeros(world * n, dtype=torch.bfloat16, device=device)
                        buf[rank * n:(rank + 1) * n] = (base[rank * n:(rank + 1) * n]).to(torch.bfloat16)
                        capi.after_current()
                        capi.all_gather(buf)
                except Exception:  # noqa: BLE001
                    ok = False
                if not agree(ok):
                    return False  # (the caller rebuilds GradSync without the sharded path; the context is left to abort)
                capi.before_current()
                if which == "reduce_scatter":
                    good = torch.equal(buf[rank * n:(rank + 1) * n], want[rank * n:(rank + 1) * n])
                else:
                    good = torch.equal(buf, base.to(torch.bfloat16))
                if not agree(bool(good)):
                    return False
            return True
        ok = True
        try:
            buf = x.clone()
            dist.reduce_scatter_tensor(buf[rank * n:(rank + 1) * n], buf, group=pg)
            ok = torch.equal(buf[rank * n:(rank + 1) * n], want[rank * n:(rank + 1) * n])
            g = torch.zeros(world * n, dtype=torch.bfloat16, device=device)
            g[rank * n:(rank + 1) * n] = (base[rank * n:(rank + 1) * n]).to(torch.bfloat16)
            dist.all_gather_into_tensor(g, g[rank * n:(rank + 1) * n], group=pg)
            ok = ok and torch.equal(g, base.to(torch.bfloat16))
        except Exception:  # noqa: BLE001
            ok = False
        try:
            return agree(bool(ok))
        except Exception:  # noqa: BLE001
            return False

    def _allgather_bf16(self, mlo, hi, a, b):
        """Every rank's freshly updated bf16 slice -> the whole [mlo, hi) of the compute copy, on the current stream."""
        p16, pg = self.engine.store.p16, self.sync.pg
        if getattr(self, "_ag_comm", None) is not None:  # C ABI (ca_allgather_bucket, in place, on the gather context's stream)
            self._ag_comm.all_gather(p16[mlo:hi])
            return
        if self.sync._has_rs:  # RCCL: in place (the input slice sits at output + rank * count)
            torch.distributed.all_gather_into_tensor(p16[mlo:hi], p16[a:b], group=pg)
            return
        per = b - a
        outs = [p16[mlo + r * per: mlo + (r + 1) * per] for r in range(self.world)]
        mine = p16[a:b].clone()
        torch.distributed.all_gather(outs, mine, group=pg)

    def consolidate(self):
        """Sharded optimiser: bring the fp32 master parameters of the other ranks' slices up to date on this rank
        (checkpoints and `save_pretrained` read the master buffer; between steps only the bf16 copy is complete) and
        return full-size (m, v) moment buffers in the parameters' layout.  A collective: every rank calls it."""
        st = self.engine.store
        if not self.zero:
            return self.m, self.v
        self.finish()
        m = torch.zeros_like(st.p32)
        v = torch.zeros_like(st.p32)
        for name, (lo, hi) in st.buckets.items():
            rep_off, sl_off = self._state_off[name]
            if sl_off is None:
                m[lo:hi], v[lo:hi] = self.m[rep_off:rep_off + hi - lo], self.v[rep_off:rep_off + hi - lo]
                continue
            mlo, _ = self.sync.shard[name]
            m[lo:mlo], v[lo:mlo] = self.m[rep_off:rep_off + mlo - lo], self.v[rep_off:rep_off + mlo - lo]
            a, b = self.sync.slice_of(name)
            m[a:b], v[a:b] = self.m[sl_off:sl_off + b - a], self.v[sl_off:sl_off + b - a]
            per = b - a
            for buf in (st.p32, m, v):
                outs = [buf[mlo + r * per: mlo + (r + 1) * per] for r in range(self.world)]
                torch.distributed.all_gather(outs, buf[a:b].clone(), group=self.sync.pg)
        return m, v

    def load_moments(self, m_full, v_full):
        """Inverse of `consolidate` for resuming: take this rank's part of full-size moment buffers."""
        if not self.zero:
            self.m.copy_(m_full)
            self.v.copy_(v_full)
            return
        st = self.engine.store
        for name, (lo, hi) in st.buckets.items():
            rep_off, sl_off = self._state_off[name]
            if sl_off is None:
                self.m[rep_off:rep_off + hi - lo], self.v[rep_off:rep_off + hi - lo] = m_full[lo:hi], v_full[lo:hi]
                continue
            mlo, _ = self.sync.shard[name]
            self.m[rep_off:rep_off + mlo - lo], self.v[rep_off:rep_off + mlo - lo] = m_full[lo:mlo], v_full[lo:mlo]
            a, b = self.sync.slice_of(name)
            self.m[sl_off:sl_off + b - a], self.v[sl_off:sl_off + b - a] = m_full[a:b], v_full[a:b]

    def grad_norm(self) -> float:
        """Global gradient norm of the last step (after the DDP mean), host scalar."""
        return float(self.gnorm_sq.sqrt().item()) / self.world
