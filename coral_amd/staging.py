"""Host -> device transfers that do not stall the host.

`tensor.to(device)` from pageable host memory is a synchronous copy: the host waits until the stream has reached
it, i.e. until the GPU has drained everything enqueued before (the previous step's backward), and only then goes on
enqueueing - every kernel launch after that is exposed.  The engines' per-step inputs (SpecAugment masks, labels,
host-collated features) therefore go through a small ring of pinned staging buffers and an asynchronous copy."""
from __future__ import annotations

import torch


class PinnedStager:
    """`to_device(t, dtype, key)`: t (host or device tensor) -> contiguous device tensor of `dtype`.  Host tensors
    are copied into one of `depth` pinned buffers kept per key and sent with a non-blocking copy on the current
    stream; a buffer is reused only after the copy that last read it has completed."""

    def __init__(self, device, depth: int = 3):
        self.device = torch.device(device)
        self.depth = depth
        self._rings: dict = {}

    def to_device(self, t: torch.Tensor, dtype: torch.dtype, key: str) -> torch.Tensor:
        if t.device.type != "cpu":
            return t.to(self.device, dtype).contiguous()
        src = t.to(dtype).contiguous()
        n = src.numel()
        ring = self._rings.get((key, dtype))
        if ring is None or ring["cap"] < n:
            cap = max(n, 1)
            ring = dict(cap=cap, buf=[torch.empty(cap, dtype=dtype).pin_memory() for _ in range(self.depth)],
                        ev=[None] * self.depth, idx=0)
            self._rings[(key, dtype)] = ring
        k = ring["idx"]
        ring["idx"] = (k + 1) % self.depth
        if ring["ev"][k] is not None:
            ring["ev"][k].synchronize()  # `depth` transfers ago: long done
        ring["buf"][k][:n].copy_(src.view(-1))
        out = torch.empty(src.shape, dtype=dtype, device=self.device)
        out.view(-1).copy_(ring["buf"][k][:n], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        ring["ev"][k] = ev
        return out
