"""Data-parallel layer (new work: the reference is single-process).

One process per GPU; every rank holds a full replica and its own shard of the batch; after backward the flat
gradient arena is averaged with a handful of large all-reduces (RCCL over xGMI when the backend is "nccl").
Buckets are contiguous slices of the arena (no packing copies); they are issued on a side stream so the reduction
of early buckets overlaps whatever still runs on the compute stream; the optimizer waits on the last bucket's event.
xGMI is point-to-point (7 links per GPU), so buckets are large (default 64 MiB) -- per-call latency, not bandwidth,
is what many small buckets would pay for.
"""
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def bucket_ranges(n: int, bucket_elems: int, align: int = 64) -> List[Tuple[int, int]]:
    """Split [0, n) into contiguous ranges of ~bucket_elems (multiples of `align`)."""
    if n <= 0:
        return []
    step = max(align, (bucket_elems // align) * align)
    out, s = [], 0
    while s < n:
        e = min(n, s + step)
        out.append((s, e))
        s = e
    return out


class GradReducer:
    """Averages a flat fp32 gradient buffer across ranks."""

    def __init__(self, flat_grad: torch.Tensor, bucket_bytes: int = 64 << 20, process_group=None, force: bool = False):
        self.flat = flat_grad
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.force = force and dist.is_initialized()      # issue the collectives even with one rank (single-GPU smoke test of the path)
        self.bucket_elems = bucket_bytes // 4
        self.ranges = bucket_ranges(flat_grad.numel(), self.bucket_elems)
        self.backend = dist.get_backend(process_group) if dist.is_initialized() else None
        self.side = torch.cuda.Stream(device=flat_grad.device) if flat_grad.is_cuda else None
        self._done = None
        # ncclAvg needs NCCL/RCCL >= 2.10; probe once (this also brings the communicator up before the first timed step) and
        # fall back to SUM followed by a scale when the backend refuses it
        self.use_avg = False
        if (self.world > 1 or self.force) and self.backend == "nccl":
            try:
                probe = torch.ones(8, dtype=torch.float32, device=flat_grad.device)
                dist.all_reduce(probe, op=dist.ReduceOp.AVG, group=process_group)
                self.use_avg = bool(torch.allclose(probe, torch.ones_like(probe)))
            except Exception:
                self.use_avg = False

    def split_ranges(self, arena, first_names) -> tuple:
        """Partition [0, numel) into the element ranges of the parameters whose names satisfy `first_names(name)` (reduced
        first, while the rest of backward still runs) and the complement.  Parameters are contiguous in the arena, so a
        handful of merged ranges result."""
        marks = []
        for name, p in arena.order:
            h = p._mrg
            if h.grad is None:
                continue
            size = (h.numel + 63) // 64 * 64
            marks.append((h.offset, min(h.offset + size, self.flat.numel()), bool(first_names(name))))
        marks.sort()
        first, rest = [], []
        for s, e, m in marks:
            tgt = first if m else rest
            if tgt and tgt[-1][1] == s:
                tgt[-1] = (tgt[-1][0], e)
            else:
                tgt.append((s, e))
        return first, rest

    def reduce_async(self, ranges=None, then=None):
        """Launch the bucket all-reduces (side stream on GPU).  Call wait() before reading the gradients.  `ranges`: element
        ranges to reduce (default: the whole buffer); each is cut into buckets.  `then`: callable issued on the same stream right
        after them."""
        if self.world == 1 and not self.force:
            if then is not None:
                then()
            return
        use_avg = self.use_avg
        op = dist.ReduceOp.AVG if use_avg else dist.ReduceOp.SUM
        if self.side is not None:
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(self.flat.device))
            with torch.cuda.stream(self.side):
                self.side.wait_event(ready)
                for s, e in self._buckets(ranges):
                    dist.all_reduce(self.flat[s:e], op=op, group=self.group)
                self._done = torch.cuda.Event()
                self._done.record(self.side)
        else:
            for s, e in self._buckets(ranges):
                dist.all_reduce(self.flat[s:e], op=op, group=self.group)
        if not use_avg:
            if self.side is not None:
                with torch.cuda.stream(self.side):
                    for s, e in (ranges if ranges is not None else [(0, self.flat.numel())]):
                        self.flat[s:e].div_(self.world)
            else:
                for s, e in (ranges if ranges is not None else [(0, self.flat.numel())]):
                    self.flat[s:e].div_(self.world)
        if then is not None:                 # work that only needs THESE gradients (e.g. their AdamW update), behind the collectives
            if self.side is not None:
                with torch.cuda.stream(self.side):
                    then()
            else:
                then()
        if self.side is not None:
            self._done = torch.cuda.Event()
            self._done.record(self.side)

    def _buckets(self, ranges):
        if ranges is None:
            return list(reversed(self.ranges))       # arena tail (decoders / last blocks) finishes first in backward
        out = []
        for s, e in ranges:
            out.extend((s + a, s + b) for a, b in bucket_ranges(e - s, self.bucket_elems))
        return out

    def wait(self):
        if self._done is not None:
            torch.cuda.current_stream(self.flat.device).wait_event(self._done)
            self._done = None

    def reduce(self):
        self.reduce_async()
        self.wait()


def broadcast_params(flat_param: torch.Tensor, src: int = 0, process_group=None):
    """Initial weight broadcast from rank 0 (all ranks then hold identical replicas)."""
    if dist.is_initialized() and dist.get_world_size(process_group) > 1:
        dist.broadcast(flat_param, src=src, group=process_group)


def all_reduce_scalar_mean(t: torch.Tensor, process_group=None) -> torch.Tensor:
    """Logging only: mean of a scalar (loss, grad-norm) over ranks."""
    if dist.is_initialized() and dist.get_world_size(process_group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=process_group)
        t = t / dist.get_world_size(process_group)
    return t
