"""One optimizer step of the pretraining loop (run_pretraining.py:683-742 train_one_epoch body) as a replayable unit.

TrainStep owns static input buffers and, on one GPU, captures masks -> forward -> masked losses -> backward ->
grad-norm -> AdamW into a single hipGraph (torch.cuda.CUDAGraph): ~600 kernel launches become one graph launch and
no host synchronisation is left in the step (the reference has 5+ `.item()` syncs per step).  With several ranks the
collective stays outside the graphs: graph A (forward+backward), bucketed RCCL all-reduce on a side stream, graph B
(optimizer).
"""
from typing import Dict, Optional

import torch

from .dp import GradReducer


class TrainStep:
    def __init__(self, model, loss_fns: Dict[str, torch.nn.Module], optimizer, batch_example: Dict[str, torch.Tensor],
                 num_encoded_tokens: int, alphas: float = 1.0, sample_tasks_uniformly: bool = False, use_graph: bool = True,
                 reducer: Optional[GradReducer] = None, loss_on_unmasked: bool = False):
        self.model, self.loss_fns, self.opt = model, loss_fns, optimizer
        self.K, self.alphas, self.uniform = num_encoded_tokens, alphas, sample_tasks_uniformly
        self.static_x = {k: torch.empty_like(v) for k, v in batch_example.items()}
        self.in_domains = list(model.input_adapters.keys())
        self.loss_on_unmasked = loss_on_unmasked      # run_pretraining.py:718-721: loss over every token instead of the masked ones
        self.loss = None
        self.task_losses = {}
        self.reducer = reducer
        self.use_graph = use_graph
        self.g_fb = self.g_opt = None

    # -- pieces ----------------------------------------------------------------------------------------
    def _fwd_bwd(self):
        x = self.static_x
        preds, masks = self.model({d: x[d] for d in self.in_domains}, num_encoded_tokens=self.K, alphas=self.alphas,
                                  sample_tasks_uniformly=self.uniform)
        self.task_losses = {t: self.loss_fns[t](preds[t], x[t], mask=None if self.loss_on_unmasked else masks.get(t)) for t in preds}
        self.loss = sum(self.task_losses.values())
        self.opt.zero_grad()
        self.loss.backward()

    def _opt(self):
        self.opt.step()

    def _eager(self):
        self._fwd_bwd()
        if self.reducer is not None:
            self.reducer.reduce()
        self._opt()

    def warmup_and_capture(self, warm_iters: int = 2):
        """A few eager iterations on a side stream (allocator warm-up, workspace sizing), then graph capture."""
        if not self.use_graph:
            return
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(warm_iters):
                self._eager()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        if self.reducer is None:
            self.g_fb = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g_fb, capture_error_mode="thread_local"):
                self._fwd_bwd()
                self._opt()
        else:
            self.g_fb = torch.cuda.CUDAGraph()
            # thread_local: the RCCL watchdog thread of torch.distributed polls events while we capture; in the default
            # "global" mode any such call from another thread invalidates the capture
            with torch.cuda.graph(self.g_fb, capture_error_mode="thread_local"):
                self._fwd_bwd()
            self.g_opt = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g_opt, pool=self.g_fb.pool(), capture_error_mode="thread_local"):
                self._opt()
        self.opt.arena.mark_fresh()

    def load(self, batch: Dict[str, torch.Tensor]):
        for k, v in batch.items():
            self.static_x[k].copy_(v, non_blocking=True)

    def step(self, batch: Optional[Dict[str, torch.Tensor]] = None):
        if batch is not None:
            self.load(batch)
        if self.g_fb is None:
            self._eager()
        elif self.g_opt is None:
            self.g_fb.replay()
        else:
            self.g_fb.replay()
            self.reducer.reduce()
            self.g_opt.replay()
        return self.loss
