"""Step boundary: parameter groups, cosine schedules and a fused flat-arena AdamW.

Reference pieces mirrored: mutils/optim_factory.py:33-92 get_parameter_groups (decay / no_decay membership,
including the quirk that `class_emb` / `task_embeddings` DO get weight decay), :95-209 create_optimizer (AdamW),
mutils/native_scaler.py:64-88 cosine_scheduler, :46-61 get_grad_norm_, :10-34 clip / skip semantics,
run_pretraining.py:496-515 (lr = blr * batch / 256, per-step lr/wd arrays), :683-688 (per-step assignment).
"""
import math
import os
from typing import Dict, List, Optional

import numpy as np
import torch

from . import _lib as L
from . import kernels as K
from .arena import build_arena


def cosine_scheduler(base_value, final_value, epochs, niter_per_ep, warmup_epochs=0, start_warmup_value=0, warmup_steps=-1):
    """Per-iteration cosine schedule with linear warm-up from start_warmup_value (mutils/native_scaler.py:64-88)."""
    warmup_iters = warmup_epochs * niter_per_ep
    if warmup_steps > 0:
        warmup_iters = warmup_steps
    warmup = np.array([])
    if warmup_epochs > 0:
        warmup = np.linspace(start_warmup_value, base_value, warmup_iters)
    n = epochs * niter_per_ep - warmup_iters
    iters = np.arange(n)
    schedule = np.array([final_value + 0.5 * (base_value - final_value) * (1 + math.cos(math.pi * i / n)) for i in iters])
    schedule = np.concatenate((warmup, schedule))
    assert len(schedule) == epochs * niter_per_ep
    return schedule


def get_parameter_groups(model, weight_decay=1e-5, skip_list=()):
    """decay / no_decay split (mutils/optim_factory.py:33-92 without layer decay): returns group dicts with `params`,
    `weight_decay`, `lr_scale` plus `names` for inspection."""
    groups: Dict[str, dict] = {}
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        if len(p.shape) == 1 or name.endswith(".bias") or name in skip_list:
            g, wd = "no_decay", 0.0
        else:
            g, wd = "decay", weight_decay
        if g not in groups:
            groups[g] = {"weight_decay": wd, "params": [], "names": [], "lr_scale": 1.0}
        groups[g]["params"].append(p)
        groups[g]["names"].append(name)
    return list(groups.values())


class PartialSumSlots:
    """Host-side map from an update range [s, e) of the arena to its FIXED slice of a partial-sum buffer (pure bookkeeping, no device
    state): a range is given a slice the first time it is seen and keeps it for ever, so that nothing about the placement changes from
    step to step -- eager launches, captured graphs and their mix (data parallel: stage updates issued eagerly between graph replays,
    the rest inside the optimizer graph) write and sum the same addresses on every replay.  The registered ranges are meant to form ONE
    partition of [0, n): a new range that overlaps registered ones retires those (the caller zeroes their slices; they are never
    handed out again, because captured graphs keep the addresses they were built with)."""

    def __init__(self, capacity: int):
        self.capacity, self.used = int(capacity), 0
        self.slots: Dict[tuple, tuple] = {}          # (s, e) -> (offset, length)

    def get(self, s: int, e: int, length: int):
        """-> (offset, length, retired): the slice of [s, e) and the (offset, length) slices that this registration retired"""
        slot = self.slots.get((s, e))
        retired = []
        if slot is None:
            for (s1, e1), sl in list(self.slots.items()):
                if s1 < e and s < e1:
                    retired.append(sl)
                    del self.slots[(s1, e1)]
            if self.used + length > self.capacity:
                raise RuntimeError("FusedAdamW: partial-sum buffer exhausted (too many distinct update ranges)")
            slot = self.slots[(s, e)] = (self.used, int(length))
            self.used += int(length)
        return slot[0], slot[1], retired

    def is_partition(self, n: int) -> bool:
        """do the registered ranges tile [0, n) without gap or overlap (the last range may run past n: alignment padding)?"""
        cov = sorted(self.slots)
        return bool(cov) and cov[0][0] == 0 and cov[-1][1] >= n and all(x[1] == y[0] for x, y in zip(cov, cov[1:]))


class AdamWFusion:
    """AdamW inside the grouped weight-gradient launches (mrg_wgrad_group_desc.fuse; csrc/gemm8p.hip epi_kind 9).  A weight gradient that
    comes out of such a launch is final, so the tile's epilogue also updates the parameter elements it covers -- the optimizer pass over
    those elements (read p, g, m, v; write p, m, v, shadow) disappears from the end of the step, where the last group's update ran alone
    on an idle matrix unit (C2 175 us, C4 440 us).  Only without clipping / skipping (the update must not depend on the global gradient
    norm), on one GPU (no gradient exchange between the weight gradient and the update), and for outputs that are contiguous views into
    the gradient arena.  The per-element arithmetic is mrg_adamw's (csrc/common.h adamw_elem): a fused and an unfused step leave
    bit-identical parameters and moments.  `ranges` collects the arena ranges updated so far in the current backward pass; the caller
    (trainer.TrainStep) keeps them out of its own update launches."""

    def __init__(self, opt: "FusedAdamW"):
        assert opt.fuse_norm, "the fused update needs the gradient norm to be log-only (no clip_grad / skip_grad)"
        self.opt = opt
        self.hp = torch.zeros(4, dtype=torch.float32, device=opt.arena.flat_p.device)
        self.ranges: List[tuple] = []
        # MRG_FUSE_OPT=2: only in the groups launched from a join (the exposed last group of a backward pass); 1: in every group
        self.only_exposed = os.environ.get("MRG_FUSE_OPT", "1") == "2"

    def begin_step(self):
        """before the backward pass of a step: this step's hyper-parameters (device side, replayable) and an empty range list"""
        o = self.opt
        self.ranges = []
        K.adamw_hp(o.lr_table, o.wd_table, o.step_dev, o.betas[0], o.betas[1], self.hp)

    def descriptor(self, ts):
        """mrg_adamw_fuse for the tasks of one grouped launch, or None when one of their outputs is not an arena view"""
        o, a = self.opt, self.opt.arena
        g0 = a.flat_g.data_ptr()
        items = []
        for t in ts:
            out, M, N, ldc = t[7], t[2], t[3], t[8]
            off_b = out.data_ptr() - g0
            if off_b < 0 or off_b % 256 or ldc != N or N % 4 or (M * N) % 64 or off_b // 4 + M * N > a.n_trainable:
                return None
            items.append((off_b // 4, off_b // 4 + M * N, ((M + 255) // 256) * ((N + 255) // 256)))
        fz = L.AdamWFuse()
        fz.g0, fz.p, fz.m, fz.v = g0, a.flat_p.data_ptr(), o.exp_avg.data_ptr(), o.exp_avg_sq.data_ptr()
        fz.shadow_bf16, fz.hp, fz.n_decay = a.flat_bf16.data_ptr(), self.hp.data_ptr(), a.n_decay
        fz.beta1, fz.beta2, fz.eps = o.betas[0], o.betas[1], o.eps
        for i, (s0, e0, tiles) in enumerate(items):
            fz.ss[i] = o._ss_slot(s0, e0, 8 * tiles).data_ptr()
            self.ranges.append((s0, e0))
        return fz

    @staticmethod
    def subtract(ranges, done):
        """ranges minus the union of `done` (both lists of half-open intervals)"""
        out = []
        done = sorted(done)
        for s0, e0 in ranges:
            cur = s0
            for ds, de in done:
                if de <= cur or ds >= e0:
                    continue
                if ds > cur:
                    out.append((cur, ds))
                cur = max(cur, de)
            if cur < e0:
                out.append((cur, e0))
        return out


class FusedAdamW:
    """AdamW over the model's flat arenas: one gradient-norm reduction + one update launch per step; the update also
    refreshes the bf16 shadow weights.  lr / weight-decay come from device-resident per-step tables indexed by a device
    iteration counter, so `step()` is hipGraph-replayable and needs no host->device traffic.

    step_dev = int32[2] on the device: [0] the iteration counter (schedule index), [1] the number of updates applied (torch AdamW's
    per-parameter state['step']; a skipped iteration, native_scaler.py:24-29, advances only [0]).

    param_groups keeps the reference's shape ([{'params', 'weight_decay', 'lr', 'lr_scale'}...]) for logging code;
    state_dict() / load_state_dict() speak torch.optim.AdamW's format (what mutils/checkpoint.py:14-20,62-66 stores and loads).
    """

    def __init__(self, model, lr_schedule, wd_schedule, betas=(0.9, 0.95), eps=1e-8, clip_grad: Optional[float] = None,
                 skip_grad: Optional[float] = None):
        params = list(model.parameters())
        if not params or not params[0].is_cuda:
            raise RuntimeError("FusedAdamW: move the model to the GPU first (there is no CPU path)")
        arena = getattr(model, "_mrg_arena", None)
        if arena is None or not arena.owns(model):
            arena = build_arena(model, params[0].device)
        self.model, self.arena = model, arena
        dev = arena.device
        self.lr_table = torch.as_tensor(np.asarray(lr_schedule, dtype=np.float32), device=dev)
        self.wd_table = torch.as_tensor(np.asarray(wd_schedule, dtype=np.float32), device=dev)
        self._lr_host = np.asarray(lr_schedule, dtype=np.float64)
        self._wd_host = np.asarray(wd_schedule, dtype=np.float64)
        self.step_dev = torch.zeros(2, dtype=torch.int32, device=dev)
        self.exp_avg = torch.zeros(arena.n_trainable, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(arena.n_trainable, dtype=torch.float32, device=dev)
        self.grad_norm = torch.zeros(1, dtype=torch.float32, device=dev)
        self.betas, self.eps = betas, eps
        # native_scaler.py:19-34: clip_grad takes precedence, skip_grad is only consulted without it
        self.clip_grad = clip_grad
        self.skip_grad = skip_grad if clip_grad is None else None
        self._host_step = 0
        # without clip / skip the gradient norm is only logged: the AdamW launches emit partial sums of g^2 over their ranges
        # (the gradient is read there anyway) and one tiny kernel finishes them -- no separate pass over the gradient arena
        self.fuse_norm = not self.clip_grad and not self.skip_grad and os.environ.get("MRG_FUSE_NORM", "1") != "0"
        # Partial-sum slots are FIXED per element range (decided the first time a range is updated, never moved): a range always
        # writes the same slice of `_ss`, and the finish kernel sums every slot handed out so far.  Nothing here depends on a host
        # cursor, so eager calls, captured graphs and any mix of the two (data parallel: stage updates issued eagerly between
        # graph replays, the rest inside the optimizer graph) stay consistent across replays.
        self.reducer = None         # ... which also hands over its dp.GradReducer (state_dict checks reducer.moments_synced)
        self.sharded = False        # set by trainer.TrainStep when the data-parallel exchange also carries the update (MRG_DP_UPDATE=sharded)
        self._ss = torch.zeros(1 << 19, dtype=torch.float32, device=dev) if self.fuse_norm else None     # <= 4096 partials per launch, <= 128 ranges
        self._slots = PartialSumSlots(1 << 19)
        self.param_groups = [
            {"params": [arena.params[n] for n in arena.names_decay], "names": arena.names_decay, "weight_decay": float(self._wd_host[0]),
             "lr": float(self._lr_host[0]), "lr_scale": 1.0},
            {"params": [arena.params[n] for n in arena.names_no_decay], "names": arena.names_no_decay, "weight_decay": 0.0,
             "lr": float(self._lr_host[0]), "lr_scale": 1.0},
        ]

    def zero_grad(self, set_to_none: bool = False):
        """Gradients are overwritten by every backward pass, so nothing has to be cleared; the views stay attached."""
        self.arena.reattach_grads()

    def set_iteration(self, it: int):
        """Position in the lr / wd schedules (start of an epoch / resume); the AdamW step count is left alone."""
        self.step_dev[0:1].fill_(int(it))
        self._host_step = int(it)

    def step_ranges(self, ranges, count: bool = True):
        """AdamW over element ranges [s, e) of the arena (64-element aligned) with the CURRENT step counters: the element updates
        are independent, so any partition of the arena gives bit-identical results to the single launch.  Only without clip / skip
        (the update must not depend on the global gradient norm).  count=False: the ranges' g^2 stay out of this rank's gradient norm
        (data parallel with a sharded update: a range every rank updates is counted by rank 0 only)."""
        a = self.arena
        for s, e in ranges:
            nd = max(0, min(e, a.n_decay) - s)
            ss = self._ss_slot(s, e) if (self.fuse_norm and count) else None
            K.adamw(a.flat_p[s:e], a.flat_g[s:e], self.exp_avg[s:e], self.exp_avg_sq[s:e], a.flat_bf16[s:e], nd, self.lr_table,
                    self.wd_table, self.step_dev, self.betas[0], self.betas[1], self.eps, sumsq_out=ss)

    def _ss_slot(self, s, e, length=None):
        """The fixed slice of the partial-sum buffer that the update of [s, e) writes (PartialSumSlots); slices of ranges this one
        supersedes are zeroed so that the finish kernel, which sums every slice ever handed out, does not count them.  length: number of
        partial sums the writer leaves (default: one per workgroup of an mrg_adamw launch over the range)."""
        off, n, retired = self._slots.get(s, e, K.adamw_nblocks(e - s) if length is None else int(length))
        for o1, n1 in retired:
            self._ss[o1:o1 + n1].zero_()
        return self._ss[off:off + n]

    def step(self, done_ranges=None):
        """global grad norm (native_scaler.py:46-61) -> AdamW (+clip / skip) -> counters.  `done_ranges`: sorted, disjoint
        element ranges already updated this step through step_ranges() (early, overlapped with backward)."""
        from .engine import join_side_stream
        join_side_stream()
        skipped = getattr(self.model, "_mrg_unused_adapters", None)
        if skipped:
            # the reference's optimizer skips parameters whose .grad is None (a modality that was not in the batch); one fused launch over the
            # arena cannot, and stepping them anyway would apply weight decay and momentum decay the reference does not
            raise RuntimeError(f"FusedAdamW.step: input adapter(s) {skipped} had trainable parameters but were not in the batch of the last backward "
                               "pass; pass every modality, or freeze the unused adapter (requires_grad_(False)) before building the optimizer")
        a = self.arena
        g = a.flat_g
        if not self.fuse_norm:
            K.grad_norm(g, self.grad_norm)
        if done_ranges or self.fuse_norm:
            assert not self.clip_grad and not self.skip_grad
            rest, pos = [], 0
            for s, e in (done_ranges or []):
                if s > pos:
                    rest.append((pos, s))
                pos = max(pos, e)
            if pos < a.n_trainable:
                rest.append((pos, a.n_trainable))
            self.step_ranges(rest)
            if self.fuse_norm:
                # sharded data-parallel update (dp.GradReducer.exchange_update_async): this rank updated -- and counted -- its shards only
                if not self.sharded and not self._slots.is_partition(a.n_trainable):
                    raise RuntimeError("FusedAdamW: the update ranges of this step do not partition the arena (gradient norm would be wrong)")
                K.sumsq_finish(self._ss, self._slots.used, self.grad_norm)
        else:
            K.adamw(a.flat_p[:a.n_trainable], g, self.exp_avg, self.exp_avg_sq, a.flat_bf16[:a.n_trainable], a.n_decay, self.lr_table,
                    self.wd_table, self.step_dev, self.betas[0], self.betas[1], self.eps, grad_norm_t=self.grad_norm,
                    clip_norm=self.clip_grad or 0.0, skip_norm=self.skip_grad or 0.0)
        K.step_inc(self.step_dev, self.grad_norm, self.clip_grad or 0.0, self.skip_grad or 0.0)
        if not torch.cuda.is_current_stream_capturing():
            a.mark_fresh()
            i = min(self._host_step, len(self._lr_host) - 1)
            self.param_groups[0]["lr"] = self.param_groups[1]["lr"] = float(self._lr_host[i])
            self.param_groups[0]["weight_decay"] = float(self._wd_host[i])
            self._host_step += 1
        return self.grad_norm

    # -- warm-up support ---------------------------------------------------------------------------
    def snapshot(self):
        """Everything a step modifies (weights, bf16 shadow, moments, counters), for iterations that must leave no trace
        (allocator / tuner warm-up before hipGraph capture, trainer.TrainStep.warmup_and_capture)."""
        a = self.arena
        return (a.flat_p.clone(), a.flat_bf16.clone(), self.exp_avg.clone(), self.exp_avg_sq.clone(), self.step_dev.clone(), self._host_step)

    def restore(self, snap):
        a = self.arena
        p, b, m, v, st, hs = snap
        a.flat_p.copy_(p); a.flat_bf16.copy_(b); self.exp_avg.copy_(m); self.exp_avg_sq.copy_(v); self.step_dev.copy_(st)
        self._host_step = hs
        a.mark_fresh()

    # -- checkpoint format: torch.optim.AdamW's -----------------------------------------------------
    def _index_order(self):
        """Parameter numbering of the reference's optimizer: create_optimizer -> get_parameter_groups (mutils/optim_factory.py:33-92)
        forms the groups in order of first appearance in named_parameters() and torch numbers parameters group after group."""
        groups = get_parameter_groups(self.model, 1.0, self.model.no_weight_decay() if hasattr(self.model, "no_weight_decay") else ())
        return groups

    def state_dict(self):
        """{'state': {i: {'step', 'exp_avg', 'exp_avg_sq'}}, 'param_groups': [...]} exactly as torch.optim.AdamW.state_dict() of the
        reference's optimizer would hold it (checkpoint['optimizer'], mutils/checkpoint.py:14-20), so that a reference run can resume
        from our checkpoint.  Moments are copied out of the flat arena per parameter (CPU tensors)."""
        red = getattr(self, "reducer", None)
        if self.sharded and red is not None and not red.moments_synced:
            # MRG_DP_UPDATE=sharded: exp_avg / exp_avg_sq of a shard are current on its owner only; serialising them as they are would write
            # stale moments that nothing detects on resume
            raise RuntimeError("FusedAdamW.state_dict: the sharded update left this rank's copy of the moments stale; call "
                               "reducer.gather_moments(optimizer.exp_avg, optimizer.exp_avg_sq) on EVERY rank first (a collective)")
        groups = self._index_order()
        it = int(self.step_dev[0].item()); t_adam = int(self.step_dev[1].item())
        i_lr = min(max(it - 1, 0), len(self._lr_host) - 1)
        state, pgs, idx = {}, [], 0
        for gr in groups:
            ids = []
            for name in gr["names"]:
                h = self.arena.params[name]._mrg
                if t_adam > 0:
                    shape = self.arena.params[name].shape
                    state[idx] = {"step": torch.tensor(float(t_adam), dtype=torch.float32),
                                  "exp_avg": self.exp_avg[h.offset:h.offset + h.numel].view(shape).detach().cpu().clone(),
                                  "exp_avg_sq": self.exp_avg_sq[h.offset:h.offset + h.numel].view(shape).detach().cpu().clone()}
                ids.append(idx)
                idx += 1
            decay = gr["weight_decay"] > 0
            pgs.append({"weight_decay": float(self._wd_host[i_lr]) if decay else 0.0, "lr_scale": 1.0, "lr": float(self._lr_host[i_lr]),
                        "betas": (float(self.betas[0]), float(self.betas[1])), "eps": float(self.eps), "amsgrad": False, "maximize": False,
                        "foreach": None, "capturable": False, "differentiable": False, "fused": None, "decoupled_weight_decay": True,
                        "params": ids})
        return {"state": state, "param_groups": pgs,
                # extras (ignored by torch.optim.Optimizer.load_state_dict): parameter names per index and the iteration counter
                "mrg_param_names": [n for gr in groups for n in gr["names"]], "mrg_iteration": it}

    def load_state_dict(self, sd):
        """Accepts torch.optim.AdamW's state_dict (a reference checkpoint, or ours).  Parameter i of the file is matched to the i-th
        parameter of the reference's group order; shapes are verified (and names, when the file carries them)."""
        if "state" not in sd or "param_groups" not in sd:
            raise ValueError("FusedAdamW.load_state_dict: expected a torch.optim.AdamW state_dict ({'state', 'param_groups'})")
        groups = self._index_order()
        names = [n for gr in groups for n in gr["names"]]
        n_file = sum(len(g["params"]) for g in sd["param_groups"])
        if n_file != len(names):
            raise ValueError(f"FusedAdamW.load_state_dict: checkpoint has {n_file} parameters, the model {len(names)}")
        if [len(g["params"]) for g in sd["param_groups"]] != [len(gr["names"]) for gr in groups]:
            raise ValueError("FusedAdamW.load_state_dict: parameter-group sizes differ from the model's decay / no_decay split")
        if "mrg_param_names" in sd and list(sd["mrg_param_names"]) != names:
            raise ValueError("FusedAdamW.load_state_dict: parameter names / order differ from the model's")
        file_ids = [i for g in sd["param_groups"] for i in g["params"]]
        steps = set()
        self.exp_avg.zero_(); self.exp_avg_sq.zero_()
        for fid, name in zip(file_ids, names):
            st = sd["state"].get(fid)
            if st is None:
                continue
            p = self.arena.params[name]
            h = p._mrg
            if tuple(st["exp_avg"].shape) != tuple(p.shape):
                raise ValueError(f"FusedAdamW.load_state_dict: {name}: moment shape {tuple(st['exp_avg'].shape)} vs parameter {tuple(p.shape)}")
            self.exp_avg[h.offset:h.offset + h.numel].copy_(st["exp_avg"].reshape(-1).to(self.exp_avg.device, torch.float32))
            self.exp_avg_sq[h.offset:h.offset + h.numel].copy_(st["exp_avg_sq"].reshape(-1).to(self.exp_avg.device, torch.float32))
            steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError(f"FusedAdamW.load_state_dict: parameters carry different step counts {sorted(steps)}: one fused counter cannot hold them")
        t_adam = steps.pop() if steps else 0
        it = int(sd.get("mrg_iteration", t_adam))
        self.step_dev.copy_(torch.tensor([it, t_adam], dtype=torch.int32))
        self._host_step = it


def create_optimizer(args, model, lr_schedule=None, wd_schedule=None):
    """AdamW as run_pretraining.py:504 builds it (opt='adamw', betas (0.9, 0.95), eps 1e-8, wd 0.05), fused."""
    if args.opt.lower().split("_")[-1] != "adamw":
        raise NotImplementedError("mirage_amd.create_optimizer: only AdamW (the pretraining optimizer) is fused")
    if lr_schedule is None:
        lr_schedule = [args.lr]
    if wd_schedule is None:
        wd_schedule = [args.weight_decay]
    return FusedAdamW(model, lr_schedule, wd_schedule, betas=tuple(getattr(args, "opt_betas", None) or (0.9, 0.999)),
                      eps=getattr(args, "opt_eps", None) or 1e-8, clip_grad=getattr(args, "clip_grad", None),
                      skip_grad=getattr(args, "skip_grad", None))
