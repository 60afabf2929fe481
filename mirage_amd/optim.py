"""Step boundary: parameter groups, cosine schedules and a fused flat-arena AdamW.

Reference pieces mirrored: mutils/optim_factory.py:33-92 get_parameter_groups (decay / no_decay membership,
including the quirk that `class_emb` / `task_embeddings` DO get weight decay), :95-209 create_optimizer (AdamW),
mutils/native_scaler.py:64-88 cosine_scheduler, :46-61 get_grad_norm_, :10-34 clip / skip semantics,
run_pretraining.py:496-515 (lr = blr * batch / 256, per-step lr/wd arrays), :683-688 (per-step assignment).
"""
import math
from typing import Dict, List, Optional

import numpy as np
import torch

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


class FusedAdamW:
    """AdamW over the model's flat arenas: one gradient-norm reduction + one update launch per step; the update also
    refreshes the bf16 shadow weights.  lr / weight-decay come from device-resident per-step tables indexed by a device
    step counter, so `step()` is hipGraph-replayable and needs no host->device traffic.

    param_groups keeps the reference's shape ([{'params', 'weight_decay', 'lr', 'lr_scale'}...]) for logging code.
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
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        self.exp_avg = torch.zeros(arena.n_trainable, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(arena.n_trainable, dtype=torch.float32, device=dev)
        self.grad_norm = torch.zeros(1, dtype=torch.float32, device=dev)
        self.betas, self.eps = betas, eps
        self.clip_grad, self.skip_grad = clip_grad, skip_grad
        self._host_step = 0
        self.param_groups = [
            {"params": [arena.params[n] for n in arena.names_decay], "names": arena.names_decay, "weight_decay": float(self._wd_host[0]),
             "lr": float(self._lr_host[0]), "lr_scale": 1.0},
            {"params": [arena.params[n] for n in arena.names_no_decay], "names": arena.names_no_decay, "weight_decay": 0.0,
             "lr": float(self._lr_host[0]), "lr_scale": 1.0},
        ]

    def zero_grad(self, set_to_none: bool = False):
        """Gradients are overwritten by every backward pass, so nothing has to be cleared; the views stay attached."""
        self.arena.reattach_grads()

    def step_ranges(self, ranges):
        """AdamW over element ranges [s, e) of the arena (64-element aligned) with the CURRENT step counter: the element updates are
        independent, so any partition of the arena gives bit-identical results to the single launch.  Only without clip / skip
        (the update must not depend on the global gradient norm)."""
        a = self.arena
        for s, e in ranges:
            nd = max(0, min(e, a.n_decay) - s)
            K.adamw(a.flat_p[s:e], a.flat_g[s:e], self.exp_avg[s:e], self.exp_avg_sq[s:e], a.flat_bf16[s:e], nd, self.lr_table,
                    self.wd_table, self.step_dev, self.betas[0], self.betas[1], self.eps)

    def step(self, done_ranges=None):
        """global grad norm (native_scaler.py:46-61) -> AdamW (+clip / skip) -> step counter + 1.  `done_ranges`: sorted, disjoint
        element ranges already updated this step through step_ranges() (early, overlapped with backward)."""
        from .engine import join_side_stream
        join_side_stream()
        a = self.arena
        g = a.flat_g
        K.grad_norm(g, self.grad_norm)
        if done_ranges:
            assert not self.clip_grad and not self.skip_grad
            rest, pos = [], 0
            for s, e in done_ranges:
                if s > pos:
                    rest.append((pos, s))
                pos = max(pos, e)
            if pos < a.n_trainable:
                rest.append((pos, a.n_trainable))
            self.step_ranges(rest)
        else:
            K.adamw(a.flat_p[:a.n_trainable], g, self.exp_avg, self.exp_avg_sq, a.flat_bf16[:a.n_trainable], a.n_decay, self.lr_table,
                    self.wd_table, self.step_dev, self.betas[0], self.betas[1], self.eps, grad_norm_t=self.grad_norm,
                    clip_norm=self.clip_grad or 0.0, skip_norm=self.skip_grad or 0.0)
        K.step_inc(self.step_dev)
        if not torch.cuda.is_current_stream_capturing():
            a.mark_fresh()
            i = min(self._host_step, len(self._lr_host) - 1)
            self.param_groups[0]["lr"] = self.param_groups[1]["lr"] = float(self._lr_host[i])
            self.param_groups[0]["weight_decay"] = float(self._wd_host[i])
            self._host_step += 1
        return self.grad_norm

    def state_dict(self):
        return {"exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq, "step": self.step_dev, "betas": self.betas, "eps": self.eps,
                "names_decay": self.arena.names_decay, "names_no_decay": self.arena.names_no_decay}

    def load_state_dict(self, sd):
        self.exp_avg.copy_(sd["exp_avg"]); self.exp_avg_sq.copy_(sd["exp_avg_sq"]); self.step_dev.copy_(sd["step"])
        self._host_step = int(self.step_dev.item())


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
