"""Flat parameter arena: every parameter of a model lives in ONE fp32 buffer (decay group first, then no-decay,
then frozen), with a same-offset fp32 gradient arena and a bf16 shadow arena.

Why (MI355X-first): the optimizer becomes one HBM-streaming launch instead of 296/440 per-tensor kernel chains
(mutils/optim_factory.py:180 + native_scaler.py:46-61), the bf16 GEMM operands are refreshed by that same launch
(no per-Linear autocast casts), and data-parallel all-reduce works on a few large contiguous bucket slices.
nn.Parameter objects stay ordinary Parameters with the reference's names/shapes (state_dict parity): their
`.data` and `.grad` are views into the arenas.
"""
from typing import Dict, Iterable, List, Optional, Tuple

import torch

from . import kernels as K

ALIGN = 64  # elements: 256 B fp32 / 128 B bf16 alignment per parameter


class _Handle:
    """Per-parameter record hung on the Parameter as `_mrg`."""
    __slots__ = ("bf16", "grad", "version", "ptr", "offset", "numel", "arena")

    def __init__(self):
        self.bf16 = None
        self.grad = None
        self.version = -1
        self.ptr = 0
        self.offset = -1
        self.numel = 0
        self.arena = None


def _is_decay(name: str, p: torch.nn.Parameter, skip: Iterable[str]) -> bool:
    # mutils/optim_factory.py:47-57: 1-D / .bias / names in the skip list get no weight decay
    return not (p.ndim == 1 or name.endswith(".bias") or name in skip)


class ParamArena:
    def __init__(self, named_params: List[Tuple[str, torch.nn.Parameter]], skip_list: Iterable[str], device: torch.device):
        skip = set(skip_list)
        decay = [(n, p) for n, p in named_params if p.requires_grad and _is_decay(n, p, skip)]
        nodecay = [(n, p) for n, p in named_params if p.requires_grad and not _is_decay(n, p, skip)]
        frozen = [(n, p) for n, p in named_params if not p.requires_grad]
        self.order = decay + nodecay + frozen
        self.names_decay = [n for n, _ in decay]
        self.names_no_decay = [n for n, _ in nodecay]
        off = 0
        offsets = []
        self.n_decay = self.n_trainable = 0
        for gi, group in enumerate((decay, nodecay, frozen)):
            for n, p in group:
                offsets.append(off)
                off += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
            if gi == 0:
                self.n_decay = off
            if gi == 1:
                self.n_trainable = off
        self.total = off
        self.device = device
        self.flat_p = torch.zeros(self.total, dtype=torch.float32, device=device)
        self.flat_g = torch.zeros(self.n_trainable, dtype=torch.float32, device=device)
        self.flat_bf16 = torch.zeros(self.total, dtype=torch.bfloat16, device=device)
        self.params: Dict[str, torch.nn.Parameter] = {}
        for (n, p), o in zip(self.order, offsets):
            numel = p.numel()
            view = self.flat_p[o:o + numel].view(p.shape)
            with torch.no_grad():
                view.copy_(p.detach().to(device=device, dtype=torch.float32))
            p.data = view
            h = _Handle()
            h.bf16 = self.flat_bf16[o:o + numel].view(p.shape)
            h.offset, h.numel, h.arena = o, numel, self
            if p.requires_grad:
                h.grad = self.flat_g[o:o + numel].view(p.shape)
                p.grad = h.grad
            h.ptr = view.data_ptr()
            p._mrg = h
            self.params[n] = p
        self._stamp = None
        self.refresh_shadow(force=True)

    # -- shadow freshness -------------------------------------------------------------------------
    def _version_stamp(self) -> int:
        return sum(p._version for p in self.params.values())

    def refresh_shadow(self, force: bool = False) -> None:
        """Re-cast fp32 -> bf16 if any parameter was modified through torch since the last cast.  (The fused
        optimizer updates the shadow itself and calls mark_fresh().)"""
        if torch.cuda.is_current_stream_capturing():
            return
        stamp = self._version_stamp()
        if force or stamp != self._stamp:
            K.cast_bf16(self.flat_p, self.flat_bf16)
            self._stamp = stamp

    def mark_fresh(self) -> None:
        self._stamp = self._version_stamp()

    def owns(self, model: torch.nn.Module) -> bool:
        for p in model.parameters():
            h = getattr(p, "_mrg", None)
            if h is None or h.arena is not self or p.data_ptr() != h.ptr:
                return False
        return True

    def reattach_grads(self) -> None:
        for p in self.params.values():
            h = p._mrg
            if h.grad is not None and p.grad is not h.grad:
                p.grad = h.grad


def build_arena(model: torch.nn.Module, device: Optional[torch.device] = None) -> ParamArena:
    named = list(model.named_parameters())
    dev = device if device is not None else named[0][1].device
    skip = model.no_weight_decay() if hasattr(model, "no_weight_decay") else ()
    arena = ParamArena(named, skip, dev)
    model._mrg_arena = arena
    return arena


def shadow(p: torch.nn.Parameter) -> torch.Tensor:
    """bf16 copy of a parameter (arena view when the owning model has an arena, else a cached private cast)."""
    h = getattr(p, "_mrg", None)
    if h is not None and h.arena is not None and h.ptr == p.data_ptr():
        return h.bf16
    if h is None or h.arena is not None:
        h = _Handle()
        p._mrg = h
    if h.bf16 is None or h.version != p._version or h.ptr != p.data_ptr():
        if not p.is_cuda:
            raise RuntimeError("mirage_amd: parameters must live on the GPU; there is no CPU path")
        src = p.detach().contiguous().float()
        h.bf16 = torch.empty(p.shape, dtype=torch.bfloat16, device=p.device)
        K.cast_bf16(src, h.bf16)
        h.version, h.ptr = p._version, p.data_ptr()
    return h.bf16


def grad_buffer(p: torch.nn.Parameter) -> Optional[torch.Tensor]:
    """fp32 tensor the backward kernels write this parameter's gradient into (overwrite semantics)."""
    if not p.requires_grad:
        return None
    h = getattr(p, "_mrg", None)
    if h is not None and h.grad is not None and h.ptr == p.data_ptr():
        if p.grad is not h.grad:
            p.grad = h.grad
        return h.grad
    if p.grad is None or p.grad.dtype != torch.float32 or not p.grad.is_contiguous():
        p.grad = torch.zeros_like(p, dtype=torch.float32, memory_format=torch.contiguous_format)
    return p.grad
