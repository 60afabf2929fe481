"""One optimizer step of the pretraining loop (run_pretraining.py:683-742 train_one_epoch body) as a replayable unit.

TrainStep owns the input slots (device addresses of the batch to read; step(batch) consumes the batch in place) plus staging buffers for
load(), and, on one GPU, captures masks -> forward -> masked losses -> backward ->
grad-norm -> AdamW into a single hipGraph (torch.cuda.CUDAGraph): ~600 kernel launches become one graph launch and
no host synchronisation is left in the step (the reference has 5+ `.item()` syncs per step).  With several ranks the
collective stays outside the graphs: graph A (forward+backward), bucketed RCCL all-reduce on a side stream, graph B
(optimizer).
"""
from typing import Dict, Optional

import torch

import os

from .dp import GradReducer

IN_PLACE_INPUTS = os.environ.get("MRG_INPUT_SLOTS", "1") != "0"   # step(batch) re-points the input slots at the batch (0: staging copy, as load())
# ... and, optionally, the lanes run from the loss forward straight into the backward pass without a join in between (each task loss is
# seeded with the constant gradient 1 instead of differentiating the sum).  Bit-identical, but measured neutral at C2 (9.494 vs 9.495 ms)
# and +5 % SLOWER at C4 (21.72 vs 20.64 ms, three adapters: adapter 0 lives on the calling stream, and without the join the two lanes
# run their backward beside its forward tail instead of beside its backward) -- off by default, kept as an A/B knob
# AdamW inside the grouped weight-gradient launches (optim.AdamWFusion; MRG_FUSE_OPT=0: separate update launches as before)
FUSE_OPT = os.environ.get("MRG_FUSE_OPT", "1") != "0"
LANE_DEFER_JOIN = os.environ.get("MRG_LANE_DEFER_JOIN", "0") != "0"
LANE_LOSSES = os.environ.get("MRG_LANE_LOSSES", "1") != "0"   # the masked losses run on the output adapters' lane streams (engine.LaneLosses)
LOSS_SEED_ONES = os.environ.get("MRG_LOSS_SEED_ONES", "1") != "0"   # with the lane losses: the backward pass is seeded with d(sum)/d(loss_t) = 1 per task and the logged sum is formed after it
LOSS_FUSED = os.environ.get("MRG_LOSS_FUSED", "1") != "0"     # ... and every lane forms loss + loss gradient in one pass over its predictions (engine.LaneLosses.fuse_bwd)
TOKEN_PREDS = os.environ.get("MRG_TOKEN_PREDS", "1") != "0"   # predictions stay in the decoders' token layout inside the step (engine.PredLayout)


class TrainStep:
    def __init__(self, model, loss_fns: Dict[str, torch.nn.Module], optimizer, batch_example: Dict[str, torch.Tensor],
                 num_encoded_tokens: int, alphas: float = 1.0, sample_tasks_uniformly: bool = False, use_graph: bool = True,
                 reducer: Optional[GradReducer] = None, loss_on_unmasked: bool = False, overlap: bool = True,
                 feed_mask_noise: bool = False):
        self.model, self.loss_fns, self.opt = model, loss_fns, optimizer
        self.K, self.alphas, self.uniform = num_encoded_tokens, alphas, sample_tasks_uniformly
        # the step reads its inputs through input slots (kernels.InputSlots): `static_x` are the placeholders the captured launches
        # were recorded with AND the staging buffers of load(); step(batch) re-points the slots at the batch itself -- no copy
        # The placeholders have the dtype the batch-reading kernels take (fp32 images, int64 label maps), whatever the example batch
        # holds: a placeholder of another dtype would be cast by the engine into a NEW tensor that no slot stands for, and every step
        # would silently train on the staging buffer.  Batches of another dtype are converted by the staging copy (step -> load).
        # ... chosen by the KIND of the modality's adapter, not by the example's dtype: an image modality whose example arrives as uint8 would
        # otherwise get an int64 placeholder that binds later int64 batches directly and trains on a cast copy no slot stands for
        def _dtype_of(k, v):
            if k in model.input_adapters:
                return torch.int64 if hasattr(model.input_adapters[k], "class_emb") else torch.float32
            return torch.float32 if v.is_floating_point() else torch.int64     # a target-only key: the loss kernels read fp32 images / int64 labels
        self.static_x = {k: torch.empty(v.shape, dtype=_dtype_of(k, v), device=v.device) for k, v in batch_example.items()}
        from . import kernels as _K
        self.slots = _K.InputSlots(self.static_x)
        self.in_domains = list(model.input_adapters.keys())
        missing = [d for d in self.in_domains if d not in batch_example]
        if missing:
            # the step drives EVERY input adapter of the model; a trainable adapter without input could not be skipped by the fused /
            # early / sharded updates anyway (FusedAdamW.step refuses it: the reference's optimizer skips parameters without a gradient)
            raise ValueError(f"TrainStep: the example batch lacks the modalities {missing} of the model's input adapters; build the model with "
                             "the adapters in use (or pass every modality)")
        # parity runs: the three random inputs of the mask generator (Dirichlet shares p [B, n], per-task
        # noise u [B, T], tie-break noise g [B, T]) come from static buffers filled by step(noise=...) instead of the device RNG
        self.static_noise = None
        if feed_mask_noise:
            first = next(iter(batch_example.values()))
            B, dev = first.shape[0], first.device
            T = model._dom_start(self.in_domains)[-1]
            self.static_noise = (torch.empty(B, len(self.in_domains), device=dev), torch.empty(B, T, device=dev), torch.empty(B, T, device=dev))
        self.loss_on_unmasked = loss_on_unmasked      # run_pretraining.py:718-721: loss over every token instead of the masked ones
        self.loss = None
        self.task_losses = {}
        self.preds, self.masks = {}, {}
        self.reducer = reducer
        self.use_graph = use_graph
        self.g_fb = self.g_opt = self.g_b = None
        self.g_mids = []
        # with several ranks the backward pass is cut at the middle of the encoder: everything above the cut (decoders, upper
        # encoder blocks) is all-reduced on the side stream while the lower half is still being differentiated
        self.split = None
        self.r_first = self.r_rest = None
        self.cuts, self.stage_ranges = [], []
        if reducer is not None and overlap and len(model.encoder) >= 2:
            import os as _os
            L_ = len(model.encoder)
            ncut = int(_os.environ.get("MRG_DP_CUTS", "2"))
            # cut points (block indices, ascending): backward reaches the highest first.  Two cuts = three stages: decoders + upper
            # half | third quarter | first quarter + embeddings, so that only the last quarter's all-reduce has nothing to hide under
            cuts = sorted({c for c in ([L_ // 2] if ncut <= 1 or L_ < 4 else [L_ // 4, L_ // 2]) if 0 < c < L_})
            self.cuts = cuts
            self.split = cuts[-1]
            model._mrg_split_at = list(cuts)
            bounds = cuts + [L_]

            def stage_pred(j):                       # stage 0 = above the highest cut (and the output adapters), then downwards
                lo, hi = bounds[len(cuts) - 1 - j], bounds[len(cuts) - j]
                def pred(name):
                    if j == 0 and name.startswith("output_adapters."):
                        return True
                    if name.startswith("encoder."):
                        return lo <= int(name.split(".")[1]) < hi
                    return False
                return pred
            preds_ = [stage_pred(j) for j in range(len(cuts))]
            self.stage_ranges = [reducer.split_ranges(optimizer.arena, pr)[0] for pr in preds_]
            self.r_first = self.stage_ranges[0]
            self.r_rest = reducer.split_ranges(optimizer.arena, lambda n: any(pr(n) for pr in preds_))[1]
        # single GPU, no clipping: the AdamW update of an encoder block does not depend on anything but that block's gradients, so it
        # is issued on the side stream as soon as they are final and overlaps the rest of the backward pass (HBM-bound update
        # under MFMA-bound GEMMs) instead of running alone at the end of the step
        self._early = None
        import os
        from .optim import FusedAdamW
        self._first_merged = None            # union of the ranges updated early in data-parallel mode (None: single update)
        self._stage_merged = []
        # MRG_DP_UPDATE=sharded: the exchange carries the update (reduce-scatter -> AdamW on this rank's shard -> all-gather of the
        # parameters, dp.GradReducer.exchange_update_async); the optimizer graph only finishes the norm and advances the counters
        self._sharded = False
        if reducer is not None and getattr(reducer, "update", "replicated") == "sharded":
            if not (isinstance(optimizer, FusedAdamW) and optimizer.fuse_norm):
                raise RuntimeError("MRG_DP_UPDATE=sharded needs FusedAdamW without clip_grad / skip_grad (the update must not depend on the global gradient norm)")
            self._sharded = True
            optimizer.sharded = True
            optimizer.reducer = reducer          # state_dict() refuses stale moments (reducer.moments_synced)
        if (self.split and not self._sharded and os.environ.get("MRG_EARLY_OPT", "1") != "0" and isinstance(optimizer, FusedAdamW)
                and not optimizer.clip_grad and not optimizer.skip_grad):
            self._stage_merged = [self._merge(r) for r in self.stage_ranges]
            self._first_merged = self._merge([r for rs in self.stage_ranges for r in rs])
        if (reducer is None and os.environ.get("MRG_EARLY_OPT", "1") != "0" and isinstance(optimizer, FusedAdamW)
                and not optimizer.clip_grad and not optimizer.skip_grad and hasattr(model, "encoder")):
            arena = optimizer.arena
            per_blk = {}
            for i, blk in enumerate(model.encoder):
                pre = f"encoder.{i}."
                rs = []
                for name, p in arena.order:
                    if name.startswith(pre) and p.requires_grad:
                        h = p._mrg
                        s0, e0 = h.offset, min((h.offset + h.numel + 63) // 64 * 64, arena.n_trainable)
                        if rs and rs[-1][1] == s0:
                            rs[-1] = (rs[-1][0], e0)
                        else:
                            rs.append((s0, e0))
                per_blk[id(blk)] = rs
            if per_blk and getattr(model, "output_adapters", None) is not None:
                rs = []
                for name, p in arena.order:
                    if name.startswith("output_adapters.") and p.requires_grad:
                        h = p._mrg
                        rs.append((h.offset, min((h.offset + h.numel + 63) // 64 * 64, arena.n_trainable)))
                if rs:
                    per_blk["decoders"] = self._merge(rs)
            if per_blk:
                self._early = per_blk
                self._early_all = self._merge([r for rs in per_blk.values() for r in rs])
        # ... and the weight gradients that leave a grouped launch are final there: that launch's epilogue applies AdamW to them itself
        # (optim.AdamWFusion), so the update of the LAST group's blocks no longer runs alone after the backward pass
        self._fusion = None
        if self._early is not None and FUSE_OPT and optimizer.fuse_norm:
            from .optim import AdamWFusion
            self._fusion = AdamWFusion(optimizer)

    def _exchange(self, ranges, j=None):
        """gradient exchange of `ranges` (None: the whole arena) -- with the update inside it when it is sharded, else with stage j's AdamW
        behind it (`then`)"""
        if self._sharded:
            a = self.opt.arena
            from . import kernels as _K
            self.reducer.exchange_update_async(
                ranges, a.flat_p,
                lambda owned, tails, count_tails: (self.opt.step_ranges(owned), self.opt.step_ranges(tails, count=count_tails)),
                lambda s0, e0: _K.cast_bf16(a.flat_p[s0:e0], a.flat_bf16[s0:e0]))
        else:
            self.reducer.reduce_async(ranges, then=self._stage_then(j) if j is not None else None)

    def _stage_then(self, j):
        """several ranks, no clipping: AdamW of a stage's parameters runs on the communication stream right behind their all-reduce,
        i.e. under the following stages of the backward pass; the optimizer graph then covers the rest."""
        if not self._stage_merged:
            return None
        rs = self._stage_merged[j]
        return lambda: self.opt.step_ranges(rs)

    @staticmethod
    def _merge(ranges):
        out = []
        for s0, e0 in sorted(ranges):
            if out and out[-1][1] >= s0:
                out[-1] = (out[-1][0], max(out[-1][1], e0))
            else:
                out.append((s0, e0))
        return out

    def _early_cb(self, blks):
        key = lambda b: b if isinstance(b, str) else id(b)
        rs = self._merge([r for b in blks for r in self._early.get(key(b), [])])
        if self._fusion is not None and self._fusion.ranges:
            rs = self._fusion.subtract(rs, self._fusion.ranges)      # already updated by the grouped weight-gradient launches
        if rs:
            self.opt.step_ranges(rs)
            self._early_seen += len([b for b in blks if key(b) in self._early])

    # -- pieces ----------------------------------------------------------------------------------------
    def _forward(self, x):
        """model forward with the predictions left in the decoders' token layout when every loss can read it (engine.PredLayout)"""
        from . import engine as E
        from .criterion import MaskedCrossEntropyLoss, MaskedMSELoss
        ours = all(isinstance(f, (MaskedMSELoss, MaskedCrossEntropyLoss)) for f in self.loss_fns.values())
        E.PredLayout.tokens = TOKEN_PREDS and ours
        self._lane_losses = bool(LANE_LOSSES and E.PredLayout.tokens and getattr(self.model, "output_adapters", None) is not None
                                 and set(self.model.output_adapters.keys()) <= set(self.loss_fns) and all(t in x for t in self.model.output_adapters.keys()))
        if self._lane_losses:
            from . import _lib as L
            E.LaneLosses.defer_join = self._defer_join = bool(LANE_DEFER_JOIN and self.reducer is None)
            if getattr(self, "_one", None) is None:
                self._one = torch.ones((), dtype=torch.float32, device=self.static_x[self.in_domains[0]].device)
            E.LaneLosses.fuse_bwd, E.LaneLosses.unit = LOSS_FUSED and LOSS_SEED_ONES, self._one
            E.LaneLosses.spec = {t: (L.MRG_LOSS_MSE if isinstance(self.loss_fns[t], MaskedMSELoss) else L.MRG_LOSS_CE, self.loss_fns[t].scale_factor,
                                     x[t], not self.loss_on_unmasked, float(getattr(self.loss_fns[t], "label_smoothing", 0.0)),
                                     bool(getattr(self.loss_fns[t], "norm_pix", False)))
                                 for t in self.model.output_adapters.keys()}
        try:
            return self.model({d: x[d] for d in self.in_domains}, num_encoded_tokens=self.K, alphas=self.alphas,
                              sample_tasks_uniformly=self.uniform, mask_noise=self.static_noise)
        finally:
            E.PredLayout.tokens = False
            E.LaneLosses.spec = None
            E.LaneLosses.defer_join = False
            E.LaneLosses.fuse_bwd = False

    def _losses(self, preds, masks, x):
        """task losses of a forward pass: computed by the adapters' lanes already (engine.LaneLosses) or here, as the reference does"""
        if self._lane_losses:
            from . import engine as E
            self.preds = dict(E.LaneLosses.preds)
            return dict(preds)
        self.preds = preds
        return {t: self.loss_fns[t](preds[t], x[t], mask=None if self.loss_on_unmasked else masks.get(t)) for t in preds}

    def pred_images(self) -> Dict[str, torch.Tensor]:
        """the last step's predictions as images [B, C, H, W] (self.preds holds what the step itself produced)"""
        from . import engine as E
        return {t: E.PredLayout.to_images(p, self.model.output_adapters[t]) for t, p in self.preds.items()}

    def _refuse_unused_adapters(self):
        """FusedAdamW.step refuses a trainable input adapter that was not in the batch (the reference's optimizer skips parameters without
        a gradient; one launch over the arena cannot).  Here the refusal comes BEFORE the backward pass: early block-wise updates, AdamW in
        the weight-gradient epilogues and the sharded exchange all modify the arena during it, and a raise at step() would leave a
        half-applied step behind."""
        skipped = getattr(self.model, "_mrg_unused_adapters", None)
        if skipped:
            raise RuntimeError(f"TrainStep: input adapter(s) {skipped} have trainable parameters but are not in the batch; pass every modality, "
                               "or freeze the unused adapter (requires_grad_(False)) before building the optimizer")

    def _fwd_bwd(self):
        from . import kernels as _K
        _K.timestamp("step_start")
        x = self.static_x
        preds, masks = self._forward(x)
        self._refuse_unused_adapters()
        self.task_losses = self._losses(preds, masks, x)
        self.masks = masks                            # static tensors of the captured step (parity tests read them and self.preds after a replay)
        self.opt.zero_grad()
        seeded = self._lane_losses and LOSS_SEED_ONES
        if seeded:
            # the task losses come from the adapters' lanes (engine.LaneLosses): each is seeded with d(sum)/d(loss_t) = 1 -- the gradient
            # the sum's backward would hand it, as one constant tensor (the lanes have formed their loss gradients under that contract
            # already, LaneLosses.fuse_bwd) -- and the logged sum is formed after the backward pass, off the critical chain
            ls = list(self.task_losses.values())
            one = self._one
            run_backward = lambda: torch.autograd.backward(ls, [one] * len(ls))
        else:
            self.loss = sum(self.task_losses.values())
            run_backward = lambda: self.loss.backward()
        if self._early is not None:
            from . import engine as E
            E._Side.early_opt, E._Side.finished, self._early_seen = self._early_cb, [], 0
            if self._fusion is not None and E._Side.enabled:
                self._fusion.begin_step()
                _K.WgradGroup.fuse = self._fusion
            try:
                run_backward()
                E.join_side_stream()             # issues the update of the last blocks as well
            finally:
                E._Side.early_opt, E._Side.finished = None, []
                _K.WgradGroup.fuse = None
            if self._early_seen != len(self._early):      # e.g. side stream disabled: fall back to the single update
                raise RuntimeError(f"early optimizer: {self._early_seen} of {len(self._early)} encoder blocks were updated")
        else:
            run_backward()
        if seeded:
            self.loss = sum(t.detach() for t in self.task_losses.values())

    def _fwd_bwd_upper(self):
        """forward + losses + backward down to the activation at the encoder cut; the gradients above the cut are final on return."""
        from . import engine as E
        x = self.static_x
        preds, masks = self._forward(x)
        self._refuse_unused_adapters()
        self.task_losses = self._losses(preds, masks, x)
        self.opt.zero_grad()
        self._hs = list(self.model._mrg_h_cuts)
        if self._lane_losses and LOSS_SEED_ONES:             # seeded with the constant 1 per task (see _fwd_bwd)
            ls = list(self.task_losses.values())
            self._g = torch.autograd.grad(ls, self._hs[-1], grad_outputs=[self._one] * len(ls), retain_graph=True)[0]
        else:
            self.loss = sum(self.task_losses.values())
            self._g = torch.autograd.grad(self.loss, self._hs[-1], retain_graph=True)[0]
        E.join_side_stream()

    def _bwd_mid(self, k):
        """backward from the cut activation k+1 down to cut activation k (parameter gradients are written by the engine)."""
        from . import engine as E
        self._g = torch.autograd.grad(self._hs[k + 1], self._hs[k], self._g, retain_graph=True)[0]
        E.join_side_stream()

    def _bwd_lower(self):
        self._hs[0].backward(self._g)
        if self._lane_losses and LOSS_SEED_ONES:
            self.loss = sum(t.detach() for t in self.task_losses.values())
        self._hs = self._g = None
        self.model._mrg_h_cuts = None
        self.model._mrg_h_mid = None

    def _opt(self):
        from . import kernels as _K
        _K.timestamp("opt_start")
        if self._sharded:
            self.opt.step(done_ranges=[(0, self.opt.arena.n_trainable)])      # every range was updated inside the exchange
        elif self._early is not None:
            self.opt.step(done_ranges=self._early_all)
        elif self._first_merged is not None:
            self.opt.step(done_ranges=self._first_merged)
        else:
            self.opt.step()
        _K.timestamp("step_end")

    def _eager(self):
        if self.split:
            self._fwd_bwd_upper()
            self._exchange(self.stage_ranges[0], 0)
            for j in range(1, len(self.cuts)):
                self._bwd_mid(len(self.cuts) - 1 - j)
                self._exchange(self.stage_ranges[j], j)
            self._bwd_lower()
            self._exchange(self.r_rest)
            self.reducer.wait()
        else:
            self._fwd_bwd()
            if self.reducer is not None:
                self._exchange(None)
                self.reducer.wait()
        self._opt()
        if self._sharded:
            self.reducer.finish_norm(self.opt.grad_norm)

    def warmup_and_capture(self, warm_iters: int = 2):
        """A few eager iterations on a side stream (allocator warm-up, workspace sizing, GEMM autotuning), then graph capture.
        The warm-up iterations leave no trace: weights, bf16 shadow, AdamW moments and both step counters are restored afterwards
        (the reference takes no such extra steps), so the first replay is the first optimizer step."""
        if not self.use_graph:
            return
        snap = self.opt.snapshot() if hasattr(self.opt, "snapshot") else None
        # ... including the random streams: the warm-up iterations draw mask noise (and stochastic-depth factors), so without this
        # the first real step would see different draws depending on warm_iters
        dev = next(iter(self.static_x.values())).device
        rng_cpu, rng_dev = torch.get_rng_state(), torch.cuda.get_rng_state(dev)
        from .engine import DropoutRNG
        drop_state = DropoutRNG.state(dev).clone()
        mask_state = self.model.mask_rng(dev).clone() if hasattr(self.model, "mask_rng") else None
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(warm_iters):
                self._eager()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        if snap is not None:
            self.opt.restore(snap)
            torch.cuda.synchronize()
        torch.set_rng_state(rng_cpu)
        torch.cuda.set_rng_state(rng_dev, dev)
        DropoutRNG.state(dev).copy_(drop_state)
        if mask_state is not None:
            self.model.mask_rng(dev).copy_(mask_state)
        if self.reducer is None:
            self.g_fb = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g_fb, capture_error_mode="thread_local"):
                self._fwd_bwd()
                self._opt()
        else:
            self.g_fb = torch.cuda.CUDAGraph()
            # thread_local: the RCCL watchdog thread of torch.distributed polls events while we capture; in the default
            # "global" mode any such call from another thread invalidates the capture
            if self.split:
                with torch.cuda.graph(self.g_fb, capture_error_mode="thread_local"):
                    self._fwd_bwd_upper()
                self.g_mids = []
                for j in range(1, len(self.cuts)):
                    gm = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gm, pool=self.g_fb.pool(), capture_error_mode="thread_local"):
                        self._bwd_mid(len(self.cuts) - 1 - j)
                    self.g_mids.append(gm)
                self.g_b = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.g_b, pool=self.g_fb.pool(), capture_error_mode="thread_local"):
                    self._bwd_lower()
            else:
                with torch.cuda.graph(self.g_fb, capture_error_mode="thread_local"):
                    self._fwd_bwd()
            self.g_opt = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g_opt, pool=self.g_fb.pool(), capture_error_mode="thread_local"):
                self._opt()
        self.opt.arena.mark_fresh()

    def load(self, batch: Dict[str, torch.Tensor]):
        """copy a batch into the staging buffers and read the next steps from there"""
        held = self.slots._held or {}
        for k, dst in self.static_x.items():
            if k in batch:
                v = batch[k]
                if tuple(v.shape) != tuple(dst.shape):                       # copy_ would broadcast silently
                    raise ValueError(f"TrainStep.load: {k} has shape {tuple(v.shape)}, the step was built for {tuple(dst.shape)}")
                dst.copy_(v, non_blocking=True)
            elif k in held and held[k] is not dst:
                dst.copy_(held[k], non_blocking=True)                        # an input bound in place earlier and not named now stays what it was
        self.slots.bind(self.static_x)

    def step(self, batch: Optional[Dict[str, torch.Tensor]] = None, noise=None):
        """One training step.  `batch` (device tensors shaped like the example batch, contiguous) is read IN PLACE: the input slots are
        pointed at it (one tiny launch) and it is kept alive until the next step; the caller must not overwrite it on another stream
        before this step has run.  Without `batch` the step re-reads what is bound (after load(): the staging buffers)."""
        if batch is not None:
            if not any(k in self.static_x for k in batch):
                raise ValueError(f"TrainStep.step: the batch has none of the inputs the step was built for ({sorted(self.static_x)}); got {sorted(batch)}")
            # extra keys (targets of other tasks, metadata) are ignored, as load() does
            if not IN_PLACE_INPUTS:
                self.load(batch)
            else:
                # a batch that names only some inputs re-points those and leaves the others bound as they are (load() semantics)
                held = self.slots._held or self.static_x
                full = {k: batch.get(k, held[k]) for k in self.slots.names}
                try:
                    self.slots.bind(full)
                except ValueError:
                    self.load(batch)                   # other dtype / layout / device: converted by the staging copy
        if noise is not None:
            if self.static_noise is None:
                raise RuntimeError("TrainStep.step(noise=...): construct the TrainStep with feed_mask_noise=True")
            for dst, src in zip(self.static_noise, noise):
                dst.copy_(src, non_blocking=True)
        if self.g_fb is None:
            self._eager()
        elif self.g_opt is None:
            self.g_fb.replay()
        elif self.g_b is not None:
            self.g_fb.replay()
            # each stage's all-reduce overlaps the following stages of the backward pass (and, without clipping, takes the stage's
            # AdamW along)
            self._exchange(self.stage_ranges[0], 0)
            for j, gm in enumerate(self.g_mids, start=1):
                gm.replay()
                self._exchange(self.stage_ranges[j], j)
            self.g_b.replay()
            self._exchange(self.r_rest)
            self.reducer.wait()
            self.g_opt.replay()
            if self._sharded:
                self.reducer.finish_norm(self.opt.grad_norm)
        else:
            self.g_fb.replay()
            self._exchange(None)
            self.reducer.wait()
            self.g_opt.replay()
            if self._sharded:
                self.reducer.finish_norm(self.opt.grad_norm)
        return self.loss
