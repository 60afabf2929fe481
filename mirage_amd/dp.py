"""Data-parallel layer (new work: the reference is single-process).

One process per GPU; every rank holds a full replica and its own shard of the batch; after backward the flat
gradient arena is averaged with a handful of large all-reduces (RCCL over xGMI when the backend is "nccl").
Buckets are contiguous slices of the arena (no packing copies); they are issued on a side stream so the reduction
of early buckets overlaps whatever still runs on the compute stream; the optimizer waits on the last bucket's event.
xGMI is point-to-point (7 links per GPU), so buckets are large (default 64 MiB) -- per-call latency, not bandwidth,
is what many small buckets would pay for.

Exchange format (SURVEY.md section 5 / 8e leaves it open; no 8-GPU node was available to measure, so both knobs default to the
conservative choice and can be A/B-ed on the first real node without code changes):
  MRG_DP_COMM  = allreduce (default): one ncclAllReduce per bucket, RCCL picks ring / tree / direct itself;
               = rs_ag: explicit reduce-scatter + all-gather per bucket (each rank reduces 1/world of the bucket, all 7 xGMI links
                 carry one shard each in both phases -- the "direct" schedule of SURVEY.md section 8e's budget row).
  MRG_DP_DTYPE = fp32 (default): gradients travel as produced (393 MB Base / 1.27 GB Large per step);
               = bf16: buckets are cast to bf16 into a staging buffer, summed in bf16 on the wire, cast back and divided in fp32
                 (half the xGMI bytes, two extra HBM passes over the bucket, ~3 significant digits per addend).
  MRG_DP_UPDATE = replicated (default): every rank applies AdamW to the whole arena after the exchange;
               = sharded: per bucket reduce-scatter -> AdamW on THIS rank's 1/world shard -> all-gather of the updated fp32 parameters
                 (+ a local re-cast of the bf16 shadow).  Same wire bytes as reduce-scatter + all-gather of the gradients, the optimizer's
                 HBM traffic (30 B per parameter: p, g, m, v read, p, m, v, shadow written) divided by world -- Large: 8.9 GB -> 1.1 GB
                 per rank and step at 8 ranks -- and the moments of a shard live on its owner only (gather_moments() before a
                 checkpoint).  Needs a norm-independent update (no clip_grad / skip_grad) and the fp32 wire format.
"""
import os
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
        self.timing = False        # bench.py: record an event pair around every wait() -- the time the compute stream is blocked on the
        self.exposed = []          # exchange, i.e. the communication the backward pass did not hide
        self.comm = os.environ.get("MRG_DP_COMM", "allreduce")
        self.wire = os.environ.get("MRG_DP_DTYPE", "fp32")
        self.update = os.environ.get("MRG_DP_UPDATE", "replicated")
        if self.comm not in ("allreduce", "rs_ag") or self.wire not in ("fp32", "bf16") or self.update not in ("replicated", "sharded"):
            raise ValueError(f"GradReducer: MRG_DP_COMM={self.comm!r} / MRG_DP_DTYPE={self.wire!r} / MRG_DP_UPDATE={self.update!r} "
                             "(allreduce|rs_ag, fp32|bf16, replicated|sharded)")
        if self.update == "sharded" and self.wire != "fp32":
            raise ValueError("GradReducer: MRG_DP_UPDATE=sharded exchanges fp32 shards (MRG_DP_DTYPE=bf16 is not implemented for it)")
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        self.moments_synced = True   # False between a sharded update and the next gather_moments()
        self._plan = {}            # sharded update: (s, e) -> (main_end, sub) of every bucket exchanged so far (gather_moments walks it)
        active = dist.is_initialized() and (self.world > 1 or self.force)
        self._stage16 = torch.empty(self.bucket_elems, dtype=torch.bfloat16, device=flat_grad.device) if (active and self.wire == "bf16") else None
        self._shard = None
        if active and (self.comm == "rs_ag" or self.update == "sharded"):
            per = (self.bucket_elems + self.world - 1) // self.world
            self._shard = torch.empty(per, dtype=torch.bfloat16 if self.wire == "bf16" else torch.float32, device=flat_grad.device)
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

    def _exchange(self, s: int, e: int, op):
        """Sum (or average, when `op` is AVG) of flat[s:e] over the ranks, in place, in the configured wire format."""
        buf = self.flat[s:e]
        n = e - s
        if self.wire == "bf16":
            wire = self._stage16[:n]
            wire.copy_(buf)
            wop = dist.ReduceOp.SUM
        else:
            wire, wop = buf, op
        if self.comm == "rs_ag" and n % self.world == 0:
            shard = self._shard[:n // self.world]
            dist.reduce_scatter_tensor(shard, wire, op=wop, group=self.group)
            dist.all_gather_into_tensor(wire, shard, group=self.group)
        else:
            dist.all_reduce(wire, op=wop, group=self.group)
        if self.wire == "bf16":
            buf.copy_(wire)
            if op == dist.ReduceOp.AVG:
                buf.mul_(1.0 / self.world)

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
                    self._exchange(s, e, op)
                self._done = torch.cuda.Event()
                self._done.record(self.side)
        else:
            for s, e in self._buckets(ranges):
                self._exchange(s, e, op)
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

    # -- sharded update (MRG_DP_UPDATE=sharded) -----------------------------------------------------------------------------
    def shard_plan(self, s: int, e: int):
        """-> (main_end, sub): flat[s:main_end) is cut into `world` shards of `sub` elements (a multiple of 64: the update kernel's
        alignment), rank r owning [s + r sub, s + (r + 1) sub); [main_end, e) is the ragged tail (< 64 world elements), which every
        rank reduces and updates in full."""
        sub = (e - s) // (self.world * 64) * 64
        return s + sub * self.world, sub

    def _exchange_update(self, s: int, e: int, op, flat_p: torch.Tensor, update, recast):
        main_end, sub = self.shard_plan(s, e)
        self._plan[(s, e)] = (main_end, sub)
        own = (s + self.rank * sub, s + (self.rank + 1) * sub)
        inplace = self.backend == "nccl"           # RCCL runs both collectives in place when the shard sits at rank * count of the buffer
        if sub:
            if inplace:
                dist.reduce_scatter_tensor(self.flat[own[0]:own[1]], self.flat[s:main_end], op=op, group=self.group)
            else:
                tmp = self._shard[:sub]
                dist.reduce_scatter_tensor(tmp, self.flat[s:main_end].clone(), op=op, group=self.group)
                self.flat[own[0]:own[1]].copy_(tmp)
            if op != dist.ReduceOp.AVG:
                self.flat[own[0]:own[1]].div_(self.world)
        if main_end < e:
            dist.all_reduce(self.flat[main_end:e], op=op, group=self.group)
            if op != dist.ReduceOp.AVG:
                self.flat[main_end:e].div_(self.world)
        # AdamW of the owned shard and of the tail; the tail's sum of g^2 is counted by rank 0 only (the per-rank sums are added up)
        update([own] if sub else [], [(main_end, e)] if main_end < e else [], self.rank == 0)
        if sub:
            if inplace:
                dist.all_gather_into_tensor(flat_p[s:main_end], flat_p[own[0]:own[1]], group=self.group)
            else:
                dist.all_gather_into_tensor(flat_p[s:main_end], flat_p[own[0]:own[1]].clone(), group=self.group)
            if recast is not None:
                recast(s, main_end)                # bf16 shadow of the gathered parameters (the owner's update wrote its own shard only)

    def exchange_update_async(self, ranges, flat_p: torch.Tensor, update, recast=None):
        """Sharded update of the element ranges `ranges` (default: the whole buffer), bucket by bucket on the communication stream:
        reduce-scatter of the gradients -> update(owned_ranges, tail_ranges, count_tails) on this rank's shard -> all-gather of the updated
        fp32 parameters into `flat_p` -> recast(s, e) of whatever mirrors them (the bf16 shadow).  Call wait() before anything reads the
        parameters.  After it the gradient buffer holds the averaged gradient only inside the ranges this rank owns."""
        if self.world == 1 and not self.force:
            rs = ranges if ranges is not None else [(0, self.flat.numel())]
            update(list(rs), [], True)
            return
        self.moments_synced = False          # from here on a shard's moments are current on its owner only (FusedAdamW.state_dict checks)
        op = dist.ReduceOp.AVG if self.use_avg else dist.ReduceOp.SUM
        if self.side is not None:
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(self.flat.device))
            with torch.cuda.stream(self.side):
                self.side.wait_event(ready)
                for s, e in self._buckets(ranges):
                    self._exchange_update(s, e, op, flat_p, update, recast)
                self._done = torch.cuda.Event()
                self._done.record(self.side)
        else:
            for s, e in self._buckets(ranges):
                self._exchange_update(s, e, op, flat_p, update, recast)

    def finish_norm(self, grad_norm: torch.Tensor):
        """sharded update: every rank's gradient norm covers the ranges it updated; the global norm is the root of the summed squares"""
        if self.world == 1 and not self.force:
            return
        grad_norm.pow_(2)
        dist.all_reduce(grad_norm, op=dist.ReduceOp.SUM, group=self.group)
        grad_norm.sqrt_()

    def gather_moments(self, *moments: torch.Tensor):
        """sharded update: bring every rank's copy of the optimizer moments up to date (all-gather over the recorded bucket plan) --
        collective, call it on every rank before a checkpoint is written."""
        if self.world == 1 and not self.force:
            return
        for (s, e), (main_end, sub) in sorted(self._plan.items()):
            if not sub:
                continue
            own = (s + self.rank * sub, s + (self.rank + 1) * sub)
            for m in moments:
                dist.all_gather_into_tensor(m[s:main_end], m[own[0]:own[1]].clone(), group=self.group)
        self.moments_synced = True

    def _buckets(self, ranges):
        if ranges is None:
            return list(reversed(self.ranges))       # arena tail (decoders / last blocks) finishes first in backward
        out = []
        for s, e in ranges:
            out.extend((s + a, s + b) for a, b in bucket_ranges(e - s, self.bucket_elems))
        return out

    def wait(self):
        if self._done is not None:
            cur = torch.cuda.current_stream(self.flat.device)
            if self.timing:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(cur)
                cur.wait_event(self._done)
                e1.record(cur)
                self.exposed.append((e0, e1))
            else:
                cur.wait_event(self._done)
            self._done = None

    def exposed_ms(self) -> float:
        """sum of the blocked intervals recorded since `timing` was switched on (call after a synchronize); clears the record"""
        t = sum(a.elapsed_time(b) for a, b in self.exposed)
        self.exposed = []
        return t

    def describe(self) -> dict:
        return {"backend": self.backend, "comm": self.comm, "wire_dtype": self.wire, "update": self.update, "bucket_mib": self.bucket_elems * 4 / 2 ** 20,
                "buckets": len(self.ranges), "avg_op": bool(self.use_avg), "bytes_per_step": int(self.flat.numel()) * (2 if self.wire == "bf16" else 4)}

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
