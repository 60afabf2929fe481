"""Forward / backward orchestration of the hot path on top of the HIP kernels.

Each stage (gather-then-embed, encoder block, decoders, masked loss) is an explicit forward function that returns the
activations its hand-written backward needs, plus a torch.autograd.Function shell so that callers keep the
reference's `loss.backward()` protocol (run_pretraining.py:735-742 -> mutils/native_scaler.py:17).
Parameter gradients are written IN PLACE (overwrite) into each Parameter's fp32 `.grad` buffer (an arena view for
arena-backed models); the Functions return None for parameters.

Dtype map (mirrors the reference's autocast map, SURVEY.md Appendix B, with bf16 instead of fp16): fp32 residual
stream / LayerNorm statistics / softmax / accumulators / parameters / gradients; bf16 GEMM operands and outputs.
"""
from typing import Dict, List, Optional, Sequence, Tuple

import os
import torch

from . import _lib as L
from . import kernels as K
from .arena import grad_buffer, shadow

BF16, F32 = torch.bfloat16, torch.float32
KS = L.MRG_OP_KSTRIDED
FUSE_EPI_COLSUM = True    # bias gradient of the producing layer from the dgrad GEMM's epilogue (column sums of dx per tile)
EMBED_WGRAD_GATHER = True # patch-embedding weight gradient: compact the gradient rows, then a plain k-strided GEMM (A/B: MRG_EMBED_GATHER=0)
EMBED_MULTI = os.environ.get("MRG_EMBED_MULTI", "1") != "0"   # the modalities' patch projections in one launch (mrg_gemm_multi)
EMBED_LANES = os.environ.get("MRG_EMBED_LANES", "0") != "0"   # the modalities' patch projections side by side on lane streams: measured neutral (C2 9.52 / 9.54 vs 9.54 / 9.51 ms, C4 20.46 vs 20.53), the two cross-stream edges cost what the shared round saves
CHAIN_END_FLUSH = os.environ.get("MRG_CHAIN_END_FLUSH", "0") != "0"   # experiment: side-stream fork at the end of the encoder's dgrad chain (neutral)
ROWFUSE = int(os.environ.get("MRG_ROWFUSE", "0"))   # the output adapters' row-local chains in one launch each (csrc/rowfuse.hip): bit 0 = norm + projection, bit 1 = proj + norm + mlp
FUSE_ATTN_BIAS = True     # q / k / v bias gradients from inside the attention-backward kernels (else: a column-sum pass over d_qkv)


# ------------------------------------------------------------------------------------------- side stream for wgrad
class _Side:
    """Weight / bias gradients are off the critical path of the backward pass (nothing downstream reads them until the
    optimizer), while the dgrad chain is a sequence of mostly latency-bound launches.  They are therefore issued on a second
    HIP stream and overlap the chain; `join()` (called at the end of the backward pass and by the optimizer) makes the main
    stream wait for them.  Works under hipGraph capture (fork / join through events)."""
    enabled = True
    stream = None
    keep = []          # tensors read by side-stream kernels stay referenced until the join
    pending = False
    queue = []         # deferred jobs: one fork (event + cross-stream edge) per flush instead of one per Linear -- inside a
                       # hipGraph every cross-stream edge costs ~16 us of dependent-launch latency on this runtime
    # blocks per fork.  With grouped weight gradients (kernels.WgradGroup) a flush is ONE launch whose efficiency grows with its tile
    # count, so the forks are coarser than when every Linear had its own split-K launches (was 4 with finer forks towards the end):
    # same box, C2 9.79 -> 9.69 ms, C4 21.27 -> 20.8 ms
    blocks_per_flush = int(os.environ.get("MRG_SIDE_BLOCKS", "6"))
    # the group queued by the LAST blocks has nothing left to overlap with: 0 (default) no special case (12 blocks -> 6, 6); 1 the old
    # halving taper (4, 4, 2, 1, 1 at 4 per flush); 2 the last MRG_SIDE_TAIL blocks form their own group (12 blocks -> 6, 4, 2).
    # Measured same box: 0 and 2 tie (C2 9.87 / 9.88 ms, C4 21.18 / 21.35): the grouped launch is as fast alone as shared
    taper = int(os.environ.get("MRG_SIDE_TAPER", "0"))
    tail_blocks = int(os.environ.get("MRG_SIDE_TAIL", "2"))
    reduce_early = True
    inline = os.environ.get("MRG_SIDE_INLINE", "0") == "1"   # experiment: the queued work goes out on the calling stream (serial), same grouping
    # the LAST flush of a backward pass (join) is exposed: weight-gradient group, its reductions, the AdamW of its blocks, one after the
    # other on an otherwise idle chip.  tail_split = n > 0 (with the early optimizer): the group is cut before its last n blocks and the
    # AdamW of the first part (HBM-bound, small workgroups) runs on a third stream under the second part's group
    tail_split = int(os.environ.get("MRG_TAIL_SPLIT", "0"))
    opt_stream = None
    opt_pending = False
    marks = []         # (queue length, block) at every block_done since the last flush: where the queue can be cut between blocks
    _blocks = 0
    origin = None      # stream the queued jobs were produced on (the decoders' lanes queue from their own streams)
    early_opt = None   # optional callable(list of finished blocks): optimizer update of blocks whose gradients are final, issued on
    finished = []      # the side stream right after the reductions of a flush (installed per step by trainer.TrainStep)

    @classmethod
    def run(cls, fn, *tensors):
        """Queue fn for the side stream; it is issued at the next flush() (end of a transformer block's backward, end of a
        decoder's backward, or join())."""
        if not cls.enabled:
            K.WgradGroup.begin()                 # same grouping as on the side stream, launched at the block's end (block_done / join)
            fn()
            return
        cur = torch.cuda.current_stream()
        if cls.queue and cls.origin != cur:
            cls.flush()                      # jobs queued from another stream must fork from THAT stream
        cls.origin = cur
        cls.queue.append(fn)
        cls.keep.extend(tensors)

    @classmethod
    def flush(cls):
        if not cls.queue:
            return
        main = cls.origin
        if cls.inline:
            cls.stream = main
        elif cls.stream is None or cls.stream.device != main.device:
            cls.stream = torch.cuda.Stream(device=main.device)
        ev = torch.cuda.Event()
        ev.record(main)
        jobs, cls.queue = cls.queue, []
        marks, cls.marks = cls.marks, []
        parts = [(jobs, None)]
        if (K.WgradGroup.exposed and cls.tail_split > 0 and cls.early_opt is not None and cls.reduce_early and len(marks) > cls.tail_split
                and [m[1] for m in marks] == cls.finished[-len(marks):]):
            n_a = len(marks) - cls.tail_split
            cut = marks[n_a - 1][0]
            parts = [(jobs[:cut], [m[1] for m in marks[:n_a]]), (jobs[cut:], None)]
        tag = K.timestamp_seq("side_flush")
        with torch.cuda.stream(cls.stream):
            if cls.stream != main:
                cls.stream.wait_event(ev)
            K.timestamp(f"{tag}.start")
            for part, early_blks in parts:
                K.WgradGroup.begin()             # the plain weight gradients of this flush go out as ONE grouped launch without split-K
                try:
                    for fn in part:
                        fn()
                except BaseException:
                    K.WgradGroup.collecting, K.WgradGroup.tasks = False, []
                    raise
                K.WgradGroup.flush()
                K.timestamp(f"{tag}.wgrad_end")
                if cls.reduce_early:
                    # finish the split-K slabs / column-sum partials produced so far (the event above covers the LayerNorm partials of
                    # the calling stream): only the last group's reductions are then left for the tail of the step
                    K.Deferred.flush()
                    if early_blks is not None:   # first part of a split tail: its AdamW goes under the second part's group
                        cls.finished = [b for b in cls.finished if not any(b is e for e in early_blks)]
                        if cls.opt_stream is None or cls.opt_stream.device != main.device:
                            cls.opt_stream = torch.cuda.Stream(device=main.device)
                        ev_a = torch.cuda.Event()
                        ev_a.record(cls.stream)
                        with torch.cuda.stream(cls.opt_stream):
                            cls.opt_stream.wait_event(ev_a)
                            cls.early_opt(early_blks)
                        cls.opt_pending = True
                    elif cls.early_opt is not None and cls.finished:
                        done, cls.finished = cls.finished, []
                        cls.early_opt(done)      # their gradients are final now: AdamW for these blocks overlaps the rest of backward
            K.timestamp(f"{tag}.end")
        cls.pending = True

    @classmethod
    def block_done(cls, remaining=None, blk=None):
        """End of one transformer block's backward.  `remaining` = blocks of the same stack still to come: the side work
        queued by the LAST blocks has nothing left to overlap with, so the forks get finer towards the end
        (12 blocks -> flush after 4, 8, 10, 11, 12) instead of leaving a third of the weight-gradient GEMMs exposed."""
        cls._blocks += 1
        if not cls.enabled and cls.early_opt is not None:
            K.WgradGroup.flush()                 # the optimizer update below needs this block's gradients now
        if cls.early_opt is not None and blk is not None:
            if cls.enabled:
                cls.finished.append(blk)
                cls.marks.append((len(cls.queue), blk))
            else:
                cls.early_opt([blk])             # no side stream, no deferred reductions: the gradients are final already
        need = cls.blocks_per_flush
        if remaining is not None and cls.taper == 1:
            need = min(cls.blocks_per_flush, max(1, remaining))
        elif remaining is not None and cls.taper == 2 and remaining == cls.tail_blocks:
            need = 1
        if cls._blocks >= need:
            cls._blocks = 0
            cls.flush()
            if not cls.enabled:
                K.WgradGroup.flush()             # side stream off: same grouping cadence, on the calling stream

    @classmethod
    def begin_backward(cls):
        """Entry of a hand-written backward pass: from here until join() the slice reductions behind weight / bias / LayerNorm
        gradients are collected and issued as one batched launch (kernels.Deferred)."""
        if cls.enabled:
            K.Deferred.active = True

    @classmethod
    def join(cls):
        K.WgradGroup.exposed = True              # the caller waits for this group: it may take the whole chip
        try:
            cls.flush()
            K.WgradGroup.flush()                 # side stream off: what the last blocks queued
        finally:
            K.WgradGroup.exposed = False
        cls._blocks = 0
        if K.Deferred.tasks:
            main = torch.cuda.current_stream()
            if cls.pending:
                ev = torch.cuda.Event()
                ev.record(main)                  # LayerNorm / column-sum partials were produced on the calling stream
                with torch.cuda.stream(cls.stream):
                    if cls.stream != main:
                        cls.stream.wait_event(ev)
                    K.Deferred.flush()
            else:
                K.Deferred.flush()
        if cls.early_opt is not None and cls.finished:       # reduce_early off: nothing was final before this point
            done, cls.finished = cls.finished, []
            if cls.pending:
                with torch.cuda.stream(cls.stream):
                    cls.early_opt(done)
            else:
                cls.early_opt(done)
        K.Deferred.active = False
        if cls.opt_pending:
            torch.cuda.current_stream().wait_stream(cls.opt_stream)
            cls.opt_pending = False
        if cls.pending:
            if cls.stream != torch.cuda.current_stream():
                torch.cuda.current_stream().wait_stream(cls.stream)
            cls.keep.clear()
            cls.pending = False
        K.Deferred.release()


def join_side_stream():
    _Side.join()


# ------------------------------------------------------------------------------------------- linear helpers
def lin_fwd(x16: torch.Tensor, lin, *, out_dtype=BF16, resid: Optional[torch.Tensor] = None, want_bf16_copy: bool = False,
            gelu: bool = False, save_pre: bool = False, row_scale: Optional[torch.Tensor] = None, dropout=None):
    """y = x W^T + b with fused epilogue.  x16: bf16 [..., K].  Returns y (and bf16 copy / pre-activation if asked).
    dropout: (rng, p, stream id) -> nn.Dropout(p) on (x W^T + b) before the row factor / residual (kernels.gemm)."""
    W = shadow(lin.weight)
    N = W.shape[0]
    Kin = W.numel() // N
    M = x16.numel() // Kin
    oshape = x16.shape[:-1] + (N,)
    out = K.new(oshape, out_dtype, x16.device)
    out2 = K.new(oshape, BF16, x16.device) if want_bf16_copy else None
    # with save_pre the epilogue stores gelu'(pre-activation) in fp16 (values in [-0.13, 1.13], 11 significant bits) instead of
    # the pre-activation itself: the backward GEMM then multiplies by it and needs no erf / exp (-18 us per fc2 dgrad at C2)
    pre = K.new(oshape, torch.float16, x16.device) if (gelu and save_pre) else None
    act = L.MRG_ACT_NONE if not gelu else (L.MRG_ACT_GELU_SAVE_GRAD if pre is not None else L.MRG_ACT_GELU)
    K.gemm(x16, W, M, N, Kin, lda=Kin, ldb=Kin, out=out, ldc=N, out2=out2, bias=lin.bias,
           resid=resid, act=act, aux_out=pre, row_scale=row_scale, dropout=dropout)
    return out, out2, pre


def lin_bwd(dy16: torch.Tensor, x16: torch.Tensor, lin, *, need_dx: bool = True, gelu_pre: Optional[torch.Tensor] = None,
            dx_dtype=BF16, dx_bf16_copy: bool = False, bias_grad: bool = True, a_map=None, k_dev=None, m_rows: Optional[int] = None,
            dx_accumulate_into: Optional[torch.Tensor] = None, dx_colsum: Optional[torch.Tensor] = None):
    """Backward of y = x W^T + b given dy (bf16 [M, N]) and x (bf16 [M, K]).
    Writes dW (and db) in place; returns dx (bf16 or fp32, optionally both) or None.  dx_colsum: fp32 [K] destination for the
    column sums of dx (the bias gradient of the layer that produced x's pre-activation), formed in the dgrad epilogue."""
    Wp = lin.weight
    W = shadow(Wp)
    N = W.shape[0]
    Kin = W.numel() // N
    M = m_rows if m_rows is not None else dy16.numel() // N
    def _param_grads():
        if Wp.requires_grad:
            gw = grad_buffer(Wp)
            K.gemm(dy16, x16, N, Kin, M, a_mode=KS, b_mode=KS, lda=N, ldb=Kin, out=gw.view(N, Kin), ldc=Kin,
                   a_map=a_map, k_dev=k_dev, splitk=K.pick_splitk(N, Kin, M), tune_split=True)
        if bias_grad and lin.bias is not None and lin.bias.requires_grad:
            K.colsum(dy16, grad_buffer(lin.bias), M=M, row_map=a_map, m_dev=k_dev)
    _Side.run(_param_grads, dy16, x16, a_map, k_dev)
    if not need_dx:
        return None, None
    dshape = dy16.shape[:-1] + (Kin,)
    if dx_accumulate_into is not None:
        dx = dx_accumulate_into
        acc = True
    else:
        dx = torch.empty(dshape, dtype=dx_dtype, device=dy16.device)
        acc = False
    dx2 = torch.empty(dshape, dtype=BF16, device=dy16.device) if dx_bf16_copy else None
    K.gemm(dy16, W, M, Kin, N, b_mode=KS, lda=N, ldb=Kin, out=dx, ldc=Kin, out2=dx2, accumulate=acc,
           act=L.MRG_ACT_NONE if gelu_pre is None else (L.MRG_ACT_MUL_AUX_F16 if gelu_pre.dtype == torch.float16 else L.MRG_ACT_GELU_BWD),
           aux_in=gelu_pre, colsum=dx_colsum if FUSE_EPI_COLSUM else None)
    if dx_colsum is not None and not FUSE_EPI_COLSUM:
        K.colsum(dx.view(-1, Kin), dx_colsum, M=M)
    return dx, dx2


def to_bf16(x32: torch.Tensor) -> torch.Tensor:
    out = torch.empty(x32.shape, dtype=BF16, device=x32.device)
    K.cast_bf16(x32, out)
    return out


# ------------------------------------------------------------------------------------------- dropout state
class DropoutRNG:
    """{seed, step} per device for the Philox dropout masks (csrc/common.h dropout_apply4).  `begin_forward` advances the step ON THE
    DEVICE (an in-place add: hipGraph-replayable, every replay draws new masks) and returns a snapshot tensor; the forward GEMMs and the
    backward mask kernels of that pass read the snapshot, so a later forward cannot change the masks an earlier backward regenerates."""
    _state = {}
    _seeded_from = {}      # device -> the torch.initial_seed() the state was derived from (None after an explicit set())

    @classmethod
    def state(cls, device) -> torch.Tensor:
        """Seeded from torch.initial_seed() on first use and again whenever that seed has changed since (torch.manual_seed(other); not
        inside a stream capture), like the mask stream (model.mask_rng): two runs in one process after the same manual_seed calls
        draw the same masks.  Process-wide (one stream per device, shared by all models); `set` pins it and checkpoints may store it."""
        device = torch.device(device)
        src = torch.initial_seed()
        st = cls._state.get(device)
        if st is None:
            st = cls._state[device] = torch.tensor([src & 0x7FFFFFFFFFFFFFFF, 0], dtype=torch.int64, device=device)
            cls._seeded_from[device] = src
        elif (cls._seeded_from.get(device) is not None and cls._seeded_from[device] != src
              and not (device.type == "cuda" and torch.cuda.is_current_stream_capturing())):
            st.copy_(torch.tensor([src & 0x7FFFFFFFFFFFFFFF, 0], dtype=torch.int64))
            cls._seeded_from[device] = src
        return st

    @classmethod
    def set(cls, device, seed: int, step: int = 0):
        """pin the stream (tests, resume): the NEXT forward uses step + 1; a later torch.manual_seed no longer reseeds it"""
        cls.state(device).copy_(torch.tensor([int(seed), int(step)], dtype=torch.int64))
        cls._seeded_from[torch.device(device)] = None

    @classmethod
    def begin_forward(cls, device) -> torch.Tensor:
        st = cls.state(device)
        st[1:2] += 1
        return st.clone()


# ------------------------------------------------------------------------------------------- transformer block
def attn_dropout_args(att):
    """(rng snapshot, p, stream id) of an Attention / CrossAttention module whose attn_drop > 0, else None.  The reference hands attn_drop to
    F.scaled_dot_product_attention as dropout_p (mirage/utils.py:181-185, 216-220), which applies it in eval mode too:
    so does this.  The snapshot is the one the owning model's forward pass took (model.MIRAGEModel.forward -> att._mrg_adrop_rng); a module
    used on its own takes one per call."""
    p = float(getattr(att, "attn_drop", 0.0) or 0.0)
    if p <= 0.0:
        return None
    rng = getattr(att, "_mrg_adrop_rng", None)
    if rng is None:
        rng = DropoutRNG.begin_forward(att.proj.weight.device)
    return (rng, p, int(getattr(att, "attn_drop_stream", ATTN_DROP_STREAM0)))


ATTN_DROP_STREAM0 = 0x10000      # mask streams of the attention-probability dropout: encoder block i 0x10000 + i; output adapter a:
                                 # cross-attention 0x20000 + 256 a, its block j 0x20000 + 256 a + 1 + j (proj / fc2 dropout use 2 i, 2 i + 1)


def block_fwd(blk, x32: torch.Tensor, save: bool, want_bf16_out: bool = False, dp=None):
    """Block.forward (mirage/utils.py:259-262): x + drop_path(attn(norm1(x))); x + drop_path(mlp(norm2(x))).  x32: fp32 [B, N, D].
    dp = (rows_attn, rows_mlp[, drop]): per-row stochastic-depth factors floor(keep + u) / keep (mirage/utils.py:103-118) or None, and
    `drop` = (rng snapshot, p, stream base) for nn.Dropout(p) after proj / fc2 (mirage/utils.py:158,187) or None; all of it is fused
    into the proj / fc2 epilogues.  dp is None when the block has neither (every BASELINE config)."""
    drop = dp[2] if (dp is not None and len(dp) > 2) else None
    d_att = (drop[0], drop[1], drop[2]) if drop is not None else None
    d_mlp = (drop[0], drop[1], drop[2] + 1) if drop is not None else None
    B, N, D = x32.shape
    H = blk.attn.num_heads
    hd = D // H
    eps = blk.norm1.eps
    st = (N * 3 * D, 3 * D)
    adrop = attn_dropout_args(blk.attn)
    # 256-wide blocks (the output adapters' decoder_transformer) may run their row-local chains as one launch each (csrc/rowfuse.hip)
    rf = ROWFUSE if (dp is None and K.Halves.mode == 0 and blk.attn.qkv.bias is not None and K.rowfuse_ok(B * N, D, 3 * D)
                     and blk.mlp.fc1.weight.shape[0] == 1024) else 0
    if rf & 1:
        qkv, ln1, mean1, rstd1 = K.rowfuse_ln_proj(x32, blk.norm1, blk.attn.qkv, shadow(blk.attn.qkv.weight), save)
    else:
        ln1, mean1, rstd1 = K.ln_fwd(x32, blk.norm1.weight, blk.norm1.bias, eps)
        qkv, _, _ = lin_fwd(ln1, blk.attn.qkv)
    o, lse = K.attn_fwd(qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:], B, H, N, N, hd, blk.attn.scale, st, st, st, need_lse=save, drop=adrop)
    if rf & 2:
        x_out, x_out16, sv = K.rowfuse_mlp(o, blk.attn.proj, shadow(blk.attn.proj.weight), x32, blk.norm2, blk.mlp.fc1, shadow(blk.mlp.fc1.weight),
                                           blk.mlp.fc2, shadow(blk.mlp.fc2.weight), save, want_bf16_out)
        x_mid, ln2, mean2, rstd2, h_pre, h_act = sv if save else (None,) * 6
    else:
        x_mid, _, _ = lin_fwd(o, blk.attn.proj, out_dtype=F32, resid=x32, row_scale=dp[0] if dp else None, dropout=d_att)
        ln2, mean2, rstd2 = K.ln_fwd(x_mid, blk.norm2.weight, blk.norm2.bias, blk.norm2.eps)
        h_act, _, h_pre = lin_fwd(ln2, blk.mlp.fc1, gelu=True, save_pre=save)
        x_out, x_out16, _ = lin_fwd(h_act, blk.mlp.fc2, out_dtype=F32, resid=x_mid, want_bf16_copy=want_bf16_out,
                                    row_scale=dp[1] if dp else None, dropout=d_mlp)
    saved = (x32, ln1, mean1, rstd1, qkv, o, lse, x_mid, ln2, mean2, rstd2, h_pre, h_act, dp, adrop) if save else None
    return x_out, x_out16, saved


def block_bwd(blk, saved, dx32: torch.Tensor, dx16: Optional[torch.Tensor], remaining: Optional[int] = None):
    """Returns (dx_in fp32, dx_in bf16).  fc2 / proj bias gradients come out of the LayerNorm backward kernels
    (column sums of the residual-stream gradient)."""
    x_in, ln1, mean1, rstd1, qkv, o, lse, x_mid, ln2, mean2, rstd2, h_pre, h_act, dp, adrop = saved
    B, N, D = x_in.shape
    H = blk.attn.num_heads
    hd = D // H
    drop = dp[2] if (dp is not None and len(dp) > 2) else None

    def branch_grad(g32, rows, sid_off):
        """gradient entering a branch whose output was masked (dropout) and / or scaled per row (stochastic depth) before the add"""
        if drop is not None:
            return K.dropout_bwd_bf16(g32, (drop[0], drop[1], drop[2] + sid_off), row_scale=rows)
        return K.row_scale_bf16(g32, rows)
    if dp is not None:
        # stochastic depth / dropout: the branch sees mask * row_scale * dy, the residual path the unscaled dy; the branch bias gradients
        # are then column sums of the scaled gradient, not of the residual-stream gradient the LayerNorm kernels sum
        dxs16 = branch_grad(dx32, dp[1], 1)
        d_pre, _ = lin_bwd(dxs16, h_act, blk.mlp.fc2, gelu_pre=h_pre, bias_grad=True, dx_colsum=grad_buffer(blk.mlp.fc1.bias))
    else:
        if dx16 is None:
            dx16 = to_bf16(dx32)
        d_pre, _ = lin_bwd(dx16, h_act, blk.mlp.fc2, gelu_pre=h_pre, bias_grad=False, dx_colsum=grad_buffer(blk.mlp.fc1.bias))
    d_ln2, _ = lin_bwd(d_pre, ln2, blk.mlp.fc1, bias_grad=False)      # fc1 bias gradient = column sums of d_pre, formed above
    dx_mid32, dx_mid16 = K.ln_bwd(d_ln2, x_mid, mean2, rstd2, blk.norm2.weight, dx32, grad_buffer(blk.norm2.weight),
                                  grad_buffer(blk.norm2.bias), None if dp is not None else grad_buffer(blk.mlp.fc2.bias))
    if dp is not None:
        d_o, _ = lin_bwd(branch_grad(dx_mid32, dp[0], 0), o, blk.attn.proj, bias_grad=True)
    else:
        d_o, _ = lin_bwd(dx_mid16, o, blk.attn.proj, bias_grad=False)
    d_qkv = torch.empty_like(qkv)
    st = (N * 3 * D, 3 * D)
    gb = grad_buffer(blk.attn.qkv.bias) if (blk.attn.qkv.bias is not None and FUSE_ATTN_BIAS and adrop is None) else None
    K.attn_bwd(qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:], o, lse, d_o, d_qkv[..., :D], d_qkv[..., D:2 * D], d_qkv[..., 2 * D:],
               B, H, N, N, hd, blk.attn.scale, st, st, st, st, st, st,
               bias_grads=(gb[:D], gb[D:2 * D], gb[2 * D:]) if gb is not None else None,   # qkv bias gradient from inside the kernels
               drop=adrop)
    d_ln1, _ = lin_bwd(d_qkv, ln1, blk.attn.qkv, bias_grad=gb is None)
    dx_in32, dx_in16 = K.ln_bwd(d_ln1, x_in, mean1, rstd1, blk.norm1.weight, dx_mid32, grad_buffer(blk.norm1.weight),
                                grad_buffer(blk.norm1.bias), None if dp is not None else grad_buffer(blk.attn.proj.bias))
    _Side.block_done(remaining, blk)
    return dx_in32, dx_in16


class BlockFn(torch.autograd.Function):
    """Stand-alone Block (mirage/utils.py:226-262) for callers that use the module directly."""

    @staticmethod
    def forward(ctx, x, blk, save, dp, *params):
        x32 = x.contiguous().float()
        out, _, saved = block_fwd(blk, x32, save, dp=dp)
        ctx.blk, ctx.saved = blk, saved
        return out

    @staticmethod
    def backward(ctx, dout):
        _Side.begin_backward()
        dx32, _ = block_bwd(ctx.blk, ctx.saved, dout.contiguous().float(), None)
        ctx.saved = None
        _Side.join()
        return (dx32, None, None, None) + (None,) * 12


# The encoder forward is one serial chain of launches that alternate between matrix-bound (qkv, fc1), memory-bound (LayerNorm,
# residual epilogues) and latency-bound (attention) kernels, each with its own tile-count tail.  Two half-batch chains on two streams
# fill each other's tails and phase gaps (profiles/r02_experiments.md: encoder forward -7.6 % at C2).  The activations the backward
# pass needs are full-batch tensors: chain A allocates them and works on the first halves, chain B on the second (kernels.Halves),
# so the backward pass (full batch, weight gradients over all rows in one GEMM) is unchanged.
# MRG_TWO_CHAINS: 0 off; 1 forward-only passes only (feature extraction, C5: -3 %); 2 (default since round 3) also the training forward:
# same box, 30 steps, two repetitions: C2 9.71 / 9.75 -> 9.60 / 9.59 ms (-1.3 %), C4 21.05 / 21.06 -> 20.58 / 20.55 ms (-2.3 %).
# Two overlapping half-size launches have no exclusive duration, so bench.py's instrumented roofline step (and the rocprof traces it is
# compared with) run ONE chain -- the per-launch numbers describe the full-batch kernels, the step time the production schedule.
TWO_CHAINS = int(os.environ.get("MRG_TWO_CHAINS", "2"))
TWO_CHAINS_MIN_ROWS = int(os.environ.get("MRG_TWO_CHAINS_MIN_ROWS", "16"))
TWO_CHAINS_GUARD = os.environ.get("MRG_TWO_CHAINS_GUARD", "1") != "0"      # s2 waits for chain A of block i - 2 before chain B of block i
_CHAIN_STREAM = {}


_CHAIN_SETTLED = set()          # (shape, save) of eager two-chain passes that ran without a single first-use autotuning


def _encoder_fwd_two_chains(x: torch.Tensor, blocks, save: bool, want_bf16: bool):
    """Cross-stream lifetimes: chain B (stream s2) reads and writes the second halves of tensors that chain A allocated on the main
    stream.  Three rules keep the caching allocator from handing such a block on while the other stream still works on it (under graph
    capture nothing else would order them): (i) the full-batch tensors of block i stay referenced here for `lag` blocks, and are
    released only after (ii) the main stream has waited on s2's end-of-block-i event; (iii) every lag-th block s2 waits for chain A's
    end of the block before, so that chain B never writes into recycled memory that chain A (running ahead on the main stream) still
    reads.  The chains overlap freely within the lag.  Tensors that chain B allocates itself belong to s2's pool and are recorded on
    the main stream by the final wait_stream."""
    main = torch.cuda.current_stream()
    s2 = _CHAIN_STREAM.get(x.device)
    if s2 is None:
        s2 = _CHAIN_STREAM[x.device] = torch.cuda.Stream(device=x.device)
    capturing = torch.cuda.is_current_stream_capturing()
    # eager passes: first-use autotuning must not time its candidates against the other chain, so the chains alternate with host
    # synchronisations until one whole pass of this shape has run without tuning anything; after that they overlap as in a graph
    skey = (tuple(x.shape), bool(save), len(blocks))
    sync_blocks = (not capturing) and K.TUNE["enabled"] and skey not in _CHAIN_SETTLED
    tuned_before = len(K.TUNE["log"])
    h = x.shape[0] // 2
    xa, xb = x[:h], x[h:]
    n = len(blocks)
    ev = torch.cuda.Event()
    ev.record(main)
    s2.wait_event(ev)
    saved_all = []
    x_in_full = x
    xa16 = None
    keep = []                                   # (end-of-block event on s2, that block's full-batch tensors), oldest first
    # the guard below matters only when full-batch tensors are given back DURING the pass: a training forward saves every one of them
    # for the backward pass, nothing is recycled before the final wait_stream, and the extra edges (-0.5 % at C2) buy nothing there
    guard = TWO_CHAINS_GUARD and not save
    lag = 4 if guard else 2                     # blocks a full-batch tensor outlives its block
    ev_a_prev = None                            # end of chain A's previous block on the main stream
    for i, blk in enumerate(blocks):
        last = want_bf16 and i == n - 1
        if len(keep) >= lag:                    # block i - lag is finished on BOTH streams from here on: its tensors may be recycled
            ev_old, _fulls_old = keep.pop(0)
            main.wait_event(ev_old)
            del _fulls_old
        K.Halves.begin(1)
        try:
            xa, xa16, sa = block_fwd(blk, xa, save, want_bf16_out=last)
        finally:
            K.Halves.end()
        fulls_i = list(K.Halves.fulls)
        if guard:
            # The opposite direction of the same lag: the full tensors of block j + lag may sit in memory that block j's unsaved
            # tensors just gave back, and chain B writes their second halves from s2 -- it must not overtake chain A's (main stream)
            # reads of block j.  Every lag-th block s2 therefore waits for the end of chain A's PREVIOUS block (finished by then: chain
            # B of a block runs beside chain A of the same block, so this never stalls it): chain B of block j + lag follows such a
            # wait at a block >= j + 1, i.e. on chain A of a block >= j.  One edge per lag blocks: each cross-stream edge of a captured
            # graph costs dependent-launch latency (an edge per block: -0.5 % at C2; waiting for the CURRENT block: -1.4 %).
            if i % lag == lag - 1 and ev_a_prev is not None:          # also before the LAST block (depths 12 / 24 are multiples of the lag)
                s2.wait_event(ev_a_prev)
            ev_a_prev = torch.cuda.Event()
            ev_a_prev.record(main)
        if sync_blocks:
            main.synchronize()
        K.Halves.begin(2)
        try:
            with torch.cuda.stream(s2):
                xb, _, _ = block_fwd(blk, xb, save, want_bf16_out=last)
                ev_i = torch.cuda.Event()
                ev_i.record(s2)
        finally:
            K.Halves.end()
        if sync_blocks:
            s2.synchronize()
        keep.append((ev_i, fulls_i + [x_in_full]))
        if save:
            sv = tuple(K.Halves.full_of(t) for t in sa)
            saved_all.append((x_in_full,) + sv[1:])
        else:
            saved_all.append(None)
        x_in_full = K.Halves.full_of(xa)
    main.wait_stream(s2)
    keep.clear()
    K.Halves.fulls = []
    if not capturing and len(K.TUNE["log"]) == tuned_before:
        _CHAIN_SETTLED.add(skey)
    return x_in_full, (K.Halves.full_of(xa16) if xa16 is not None else None), saved_all


class EncoderFn(torch.autograd.Function):
    """nn.Sequential(Block x depth) (mirage/model.py:82-94,409) as one node: the (fp32, bf16) gradient pair flows
    between blocks without touching autograd."""

    @staticmethod
    def forward(ctx, x, blocks, save, want_bf16, defer_join, dps, *params):
        ctx.defer_join = defer_join
        ctx.set_materialize_grads(False)         # the bf16 twin's "gradient" stays None instead of a zero-filled [B, N, D] tensor per backward pass
        K.timestamp("enc_fwd_start")
        x32 = x
        saved_all = []
        x16 = None
        n = len(blocks)
        if (TWO_CHAINS == 2 or (TWO_CHAINS == 1 and not save)) and not dps and not any(getattr(b.attn, "attn_drop", 0.0) for b in blocks) and x.shape[0] % 2 == 0 and x.is_contiguous() and x.shape[0] * x.shape[1] >= TWO_CHAINS_MIN_ROWS:
            x32, x16, saved_all = _encoder_fwd_two_chains(x, blocks, save, want_bf16)
        else:
            for i, blk in enumerate(blocks):
                x32, x16, sv = block_fwd(blk, x32, save, want_bf16_out=(want_bf16 and i == n - 1), dp=dps[i] if dps else None)
                saved_all.append(sv)
        ctx.blocks, ctx.saved = blocks, saved_all
        K.timestamp("enc_fwd_end")
        if want_bf16:
            ctx.mark_non_differentiable(x16)
            return x32, x16
        return x32, None

    @staticmethod
    def backward(ctx, dout, _unused):
        _Side.begin_backward()
        K.timestamp("enc_bwd_start")
        dx32 = dout.contiguous()
        dx16 = _twin_of(dx32)                    # the output adapters' last proj_context dgrad writes it beside the fp32 sum
        n = len(ctx.blocks)
        for i, (blk, sv) in enumerate(zip(reversed(ctx.blocks), reversed(ctx.saved))):
            dx32, dx16 = block_bwd(blk, sv, dx32, dx16, remaining=(n - 1 - i) if _Side.taper else None)
            K.timestamp(f"enc_bwd_block_end.{n - 1 - i}")
        ctx.saved = None
        K.timestamp("enc_bwd_chain_end")
        # (forking what the last blocks queued HERE instead of at the embedding backward's join was measured neutral: section 3.3)
        if CHAIN_END_FLUSH:
            _Side.flush()
        # hand the bf16 twin of the returned gradient to the embedding backward: keyed by the tensor OBJECT (identity
        # is checked on lookup, so a recycled address can never alias); at most one entry is alive
        _TWINS.clear()
        _TWINS[dx32.data_ptr()] = (dx32, dx16)
        if not ctx.defer_join:
            _Side.join()
        return (dx32,) + (None,) * (len(ctx.needs_input_grad) - 1)


# fp32 gradient tensor -> its bf16 twin, valid only while the fp32 tensor object itself is alive and identical
_TWINS: Dict[int, Tuple[torch.Tensor, torch.Tensor]] = {}


def _twin_of(g32: torch.Tensor) -> Optional[torch.Tensor]:
    ent = _TWINS.pop(g32.data_ptr(), None)
    if ent is not None and ent[0] is g32:
        return ent[1]
    return None


# ------------------------------------------------------------------------------------------- input adapters
def _pos_tokens(adapter) -> torch.Tensor:
    """[1, D, h, w] frozen table -> token-major fp32 [h*w, D], cached (F.interpolate at identical size is an
    identity: input_adapters.py:104,232, output_adapters.py:176,221)."""
    pe = adapter.pos_emb
    cache = getattr(adapter, "_mrg_pos_tok", None)
    # a learnable table (learnable_pos_emb, mirage/input_adapters.py:72) is re-read every pass: the fused optimizer updates the arena in
    # place without touching the tensor's version counter
    if cache is None or pe.requires_grad or cache[0] != (pe.data_ptr(), pe._version):
        _, D, h, w = pe.shape
        tok = pe.detach()[0].permute(1, 2, 0).reshape(h * w, D).contiguous().float()
        adapter._mrg_pos_tok = ((pe.data_ptr(), pe._version), tok)
        return tok
    return cache[1]


def embed_fwd(model, x: Dict[str, torch.Tensor], ids_keep: torch.Tensor, save: bool):
    """input adapters (mirage/input_adapters.py:87-110, 211-238) + token select + global tokens
    (mirage/model.py:384-391), computing only the kept tokens.  The modalities and their order are x's (mirage/model.py:352-356)."""
    doms = list(x.keys())
    B, Kk = ids_keep.shape
    ng = model.num_global_tokens
    N = Kk + ng
    D = model.global_tokens.shape[-1]
    dev = ids_keep.device
    dom_start = model._dom_start(doms)
    rows, toks, srcb, counts = K.embed_plan(ids_keep, N, dom_start)
    cap = B * Kk
    x0 = torch.empty((B, N, D), dtype=F32, device=dev)
    def embed_one(d, dom):
        ad = model.input_adapters[dom]
        inp = x[dom]
        if ad.kind == "semseg":
            A = K.im2col_layermap(inp.contiguous(), ad.class_emb.weight, srcb[d], toks[d], counts[d:d + 1], cap, ad.P_H)
        else:
            A = K.im2col_image(inp.contiguous().float(), srcb[d], toks[d], counts[d:d + 1], cap, ad.P_H)
        W = shadow(ad.proj.weight)
        Kp = A.shape[1]
        pr = dict(A=A, B=W, M=cap, N=D, K=Kp, lda=Kp, ldb=Kp, out=x0, ldc=D, bias=ad.proj.bias, row_map=rows[d], add_tab=_pos_tokens(ad),
                  add_row=toks[d], m_dev=counts[d:d + 1])
        if group is not None:
            group.append(pr)
        else:
            K.gemm_multi([pr])
        return A if save else None
    # the modalities' patch projections are independent (disjoint rows of x0), each a latency chain over its 16 k-tiles on fewer tiles than
    # CUs: they leave as ONE launch (mrg_gemm_multi, MRG_EMBED_MULTI=0: one launch each).  (MRG_EMBED_LANES=1, side by side on lane
    # streams, was measured neutral: the two cross-stream edges cost what the shared round saves.)
    group = [] if (EMBED_MULTI and not EMBED_LANES and len(doms) > 1 and all(x[dom].shape[-1] % 8 == 0 for dom in doms)) else None
    saved = _Lanes.run_all([(lambda d=d, dom=dom: embed_one(d, dom)) for d, dom in enumerate(doms)], dev,
                           enabled=_Lanes.enabled and EMBED_LANES)
    if group:
        # one launch for the problems the single-stage kernel is the choice for (< 24 k-tiles, mrg_gemm's own rule); a long-K projection
        # (the layer maps' class-embedding patches) keeps its pipelined launch: inside the shared launch it would set the pace (C4: 110 vs 66 us)
        fits = lambda pr: pr["K"] % 64 == 0 and pr["K"] < 24 * 64 and pr["A"].data_ptr() % 16 == 0 and pr["B"].data_ptr() % 16 == 0
        short = [pr for pr in group if fits(pr)]
        K.gemm_multi(short) if len(short) > 1 else [K.gemm_multi([pr]) for pr in short]
        for pr in group:
            if not fits(pr):
                K.gemm_multi([pr])
    K.fill_global(x0, model.global_tokens, B, N, Kk)
    if save:
        model._mrg_unused_adapters = unused_trainable_adapters(model, doms)
    plan = (rows, toks, srcb, counts)
    return x0, (plan, saved, {k: x[k] for k in doms}) if save else None


def unused_trainable_adapters(model, doms) -> List[str]:
    """input adapters with trainable parameters whose modality is not among `doms` (the batch of this pass)"""
    return sorted(name for name, ad in model.input_adapters.items()
                  if name not in doms and any(p_.requires_grad for p_ in ad.parameters()))


def embed_bwd(model, saved_all, dx32: torch.Tensor, dx16: Optional[torch.Tensor]):
    (rows, toks, srcb, counts), A_list, inputs = saved_all
    doms = list(inputs.keys())
    for name in unused_trainable_adapters(model, inputs):
        # a modality that was not in x: torch leaves its parameters' .grad at None and the optimizer skips them; here gradients are
        # buffers that every backward pass overwrites, so the unused adapter's are cleared.  The adapter was reported at FORWARD time
        # (embed_fwd -> model._mrg_unused_adapters): the fused optimizer refuses to step it (momentum and weight decay on parameters the
        # reference's optimizer would skip), and trainer.TrainStep refuses before its backward pass, i.e. before any early / fused /
        # sharded update of the same step has touched the arena
        for p_ in model.input_adapters[name].parameters():
            if p_.requires_grad:
                grad_buffer(p_).zero_()
    B, N, D = dx32.shape
    ng = model.num_global_tokens
    Kk = N - ng
    cap = B * Kk
    if dx16 is None:
        dx16 = to_bf16(dx32)
    for d, dom in enumerate(doms):
        ad = model.input_adapters[dom]
        A = A_list[d]
        Kp = A.shape[1]
        cnt = counts[d:d + 1]
        if ad.proj.weight.requires_grad:
            gw = grad_buffer(ad.proj.weight)
            if EMBED_WGRAD_GATHER:
                # compact the modality's gradient rows first (zero beyond the device-side count): the contraction then runs over
                # contiguous rows on the LDS-DMA kernels instead of the generic gathered loader (~90 us -> ~25 us per modality,
                # at the very end of the backward pass where nothing overlaps it)
                dxc = K.gather_rows_bf16(dx16.view(B * N, D), rows[d], cnt, cap)
                K.gemm(dxc, A, D, Kp, cap, a_mode=KS, b_mode=KS, lda=D, ldb=Kp, out=gw.view(D, Kp), ldc=Kp, k_dev=cnt,
                       splitk=K.pick_splitk(D, Kp, max(cap // len(doms), 64)))
            else:
                K.gemm(dx16, A, D, Kp, cap, a_mode=KS, b_mode=KS, lda=D, ldb=Kp, out=gw.view(D, Kp), ldc=Kp, a_map=rows[d], k_dev=cnt,
                       splitk=K.pick_splitk(D, Kp, max(cap // len(doms), 64)))
        if ad.proj.bias.requires_grad:
            K.colsum(dx32.view(B * N, D), grad_buffer(ad.proj.bias), M=cap, row_map=rows[d], m_dev=cnt)
        if ad.pos_emb.requires_grad:
            dom_start = model._dom_start(doms)
            K.posemb_grad(dx32.view(B * N, D), grad_buffer(ad.pos_emb), B, dom_start[d + 1] - dom_start[d], plan=(rows[d], toks[d], srcb[d], cnt, cap))
        if ad.kind == "semseg" and ad.class_emb.weight.requires_grad:
            W = shadow(ad.proj.weight)
            dA = torch.empty((cap, Kp), dtype=BF16, device=dx32.device)
            K.gemm(dx16, W, cap, Kp, D, b_mode=KS, lda=D, ldb=Kp, out=dA, ldc=Kp, a_map=rows[d], m_dev=cnt)
            ncls, E = ad.class_emb.weight.shape
            K.class_emb_grad(dA, inputs[dom].contiguous(), srcb[d], toks[d], cnt, ncls, E, ad.P_H, grad_buffer(ad.class_emb.weight))
    if model.global_tokens.requires_grad:
        gg = grad_buffer(model.global_tokens).view(ng, D)
        flat = dx32.view(B * N, D)
        for g in range(ng):
            rm = model._global_rows(B, N, Kk + g, dx32.device)
            K.colsum(flat, gg[g], M=B, row_map=rm)


class EmbedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, x, ids_keep, save, *params):
        x0, saved = embed_fwd(model, x, ids_keep, save)
        ctx.model, ctx.saved = model, saved
        return x0

    @staticmethod
    def backward(ctx, dx):
        _Side.begin_backward()
        dx = dx.contiguous()
        embed_bwd(ctx.model, ctx.saved, dx, _twin_of(dx))
        ctx.saved = None
        _Side.join()       # the embedding backward is the last node of the graph: parameter gradients are complete after this
        return (None,) * len(ctx.needs_input_grad)


# ------------------------------------------------------------------------------------------- output adapters
def decoder_fwd(ad, enc32: torch.Tensor, enc16: torch.Tensor, ids_keep, ids_restore, dom_names: Sequence[str],
                dom_start: Sequence[int], ng: int, save: bool):
    """SpatialOutputAdapter.forward (mirage/output_adapters.py:248-296)."""
    K.timestamp(f"dec_fwd_start.{ad.task}")
    B, N, D = enc32.shape
    Kk = N - ng
    T = dom_start[-1]
    Dd = ad.dim_tokens
    H = ad.decoder.num_heads
    hd = Dd // H
    dev = enc32.device
    ctx, _, _ = lin_fwd(enc16, ad.proj_context, out_dtype=F32)                           # :272
    qdom = dom_names.index(ad.task) if (ad.use_task_queries and ad.task in dom_names) else -1
    task_embs = [ad.task_embeddings[n] if (ad.task_embeddings is not None and n in ad.task_embeddings) else None for n in dom_names]
    pos = _pos_tokens(ad)
    Tq = pos.shape[0]
    # generate_context_embeddings interpolates pos_emb to THIS adapter's grid and asserts every context task has that many
    # tokens (mirage/output_adapters.py:176-179): mixed token grids are an error in the reference too
    for i, n in enumerate(dom_names):
        nt = dom_start[i + 1] - dom_start[i]
        assert nt == Tq, f"# tokens ({nt}) != # pos. embeddings ({Tq})"
    q_te = None
    if qdom < 0 and ad.task_embeddings is not None and ad.task in ad.task_embeddings:
        q_te = ad.task_embeddings[ad.task]
    desc = K.dec_assemble_desc(ctx, ad.mask_token, task_embs, pos, ids_keep, ids_restore, dom_start, qdom, B, N, Kk, T, Tq, Dd, q_te)
    queries, context = K.dec_assemble_fwd(desc, dev)                                     # :188-246
    eps = ad.query_norm.eps
    sq, skv = (Tq * Dd, Dd), (N * 2 * Dd, 2 * Dd)
    adrop = attn_dropout_args(ad.decoder)
    blocks = list(ad.decoder_transformer) if isinstance(ad.decoder_transformer, torch.nn.Sequential) else []
    # the row-local chains as one launch each (csrc/rowfuse.hip; MRG_ROWFUSE bit 0: norm + projection, bit 1: proj + out_norm + mlp)
    rf = ROWFUSE if (K.rowfuse_ok(B * Tq, Dd, Dd) and K.rowfuse_ok(B * N, Dd, 2 * Dd) and ad.mlp.fc1.weight.shape[0] == 1024
                     and ad.decoder.q.bias is not None and ad.decoder.kv.bias is not None) else 0
    if rf & 1:
        qp, qn, mq, rq = K.rowfuse_ln_proj(queries, ad.query_norm, ad.decoder.q, shadow(ad.decoder.q.weight), save)
        kv, cn, mc, rc = K.rowfuse_ln_proj(context, ad.context_norm, ad.decoder.kv, shadow(ad.decoder.kv.weight), save)
    else:
        qn, mq, rq = K.ln_fwd(queries, ad.query_norm.weight, ad.query_norm.bias, eps)
        cn, mc, rc = K.ln_fwd(context, ad.context_norm.weight, ad.context_norm.bias, eps)
        qp, _, _ = lin_fwd(qn, ad.decoder.q)
        kv, _, _ = lin_fwd(cn, ad.decoder.kv)
    o, lse = K.attn_fwd(qp, kv[..., :Dd], kv[..., Dd:], B, H, Tq, N, hd, ad.decoder.scale, sq, skv, skv, need_lse=save, drop=adrop)
    if rf & 2:
        x2, x16, sv = K.rowfuse_mlp(o, ad.decoder.proj, shadow(ad.decoder.proj.weight), None, ad.out_norm, ad.mlp.fc1, shadow(ad.mlp.fc1.weight),
                                    ad.mlp.fc2, shadow(ad.mlp.fc2.weight), save, len(blocks) == 0)          # :279-280
        x1, on, mo, ro, h_pre, h_act = sv if save else (None,) * 6
    else:
        x1, _, _ = lin_fwd(o, ad.decoder.proj, out_dtype=F32)                                # :279 (no residual)
        on, mo, ro = K.ln_fwd(x1, ad.out_norm.weight, ad.out_norm.bias, eps)
        h_act, _, h_pre = lin_fwd(on, ad.mlp.fc1, gelu=True, save_pre=save)
        x2, x16, _ = lin_fwd(h_act, ad.mlp.fc2, out_dtype=F32, resid=x1, want_bf16_copy=(len(blocks) == 0))   # :280
    blk_saved = []
    xx = x2
    for i, blk in enumerate(blocks):                                                     # :285
        xx, x16, sv = block_fwd(blk, xx, save, want_bf16_out=(i == len(blocks) - 1))
        blk_saved.append(sv)
    y, _, _ = lin_fwd(x16, ad.out_proj)                                                  # :288
    g = ad.image_size[0] // (ad.stride_level * ad.P_H)
    if PredLayout.tokens:
        img = y.view(B, Tq, -1)              # the masked losses read (and differentiate) the token layout directly: no image copy
    else:
        img = K.patch_rearrange(y.view(B, Tq, -1), True, B, ad.num_channels, g, ad.P_H)  # :291-294
    K.timestamp(f"dec_fwd_end.{ad.task}")
    saved = None
    if save:
        saved = (desc, task_embs, queries, context, qn, mq, rq, cn, mc, rc, qp, kv, o, lse, x1, on, mo, ro, h_pre, h_act, x2,
                 blk_saved, x16, enc16, (B, N, Tq, Dd, H, hd, g, adrop))
    return img, saved


def decoder_bwd(ad, saved, d_img: torch.Tensor):
    """Backward of one adapter down to d(proj_context output); returns it in bf16.  The caller folds the proj_context dgrads of
    all adapters into one shared d_enc buffer (decoder_bwd_finish) after the per-adapter streams have joined."""
    (desc, task_embs, queries, context, qn, mq, rq, cn, mc, rc, qp, kv, o, lse, x1, on, mo, ro, h_pre, h_act, x2, blk_saved, x16,
     enc16, (B, N, Tq, Dd, H, hd, g, adrop)) = saved
    dev = d_img.device
    K.timestamp(f"dec_bwd_start.{ad.task}")
    blocks = list(ad.decoder_transformer) if isinstance(ad.decoder_transformer, torch.nn.Sequential) else []
    if d_img.dim() == 3:                     # PredLayout.tokens: the loss gradient arrives in the token layout already
        d_y = d_img.contiguous().view(B * Tq, -1)
    else:
        d_y = K.patch_rearrange(d_img.contiguous(), False, B, ad.num_channels, g, ad.P_H).view(B * Tq, -1)
    if d_y.dtype != BF16:
        d_y = to_bf16(d_y)
    dx32, dx16 = lin_bwd(d_y, x16, ad.out_proj, dx_dtype=F32, dx_bf16_copy=True)
    dx32 = dx32.view(B, Tq, Dd)
    dx16 = dx16.view(B, Tq, Dd)
    for blk, sv in zip(reversed(blocks), reversed(blk_saved)):
        dx32, dx16 = block_bwd(blk, sv, dx32, dx16)
    d_pre, _ = lin_bwd(dx16, h_act, ad.mlp.fc2, gelu_pre=h_pre, bias_grad=False, dx_colsum=grad_buffer(ad.mlp.fc1.bias))
    d_on, _ = lin_bwd(d_pre, on, ad.mlp.fc1, bias_grad=False)
    dx1_32, dx1_16 = K.ln_bwd(d_on, x1, mo, ro, ad.out_norm.weight, dx32, grad_buffer(ad.out_norm.weight),
                              grad_buffer(ad.out_norm.bias), grad_buffer(ad.mlp.fc2.bias))
    d_o, _ = lin_bwd(dx1_16, o, ad.decoder.proj, bias_grad=False)
    K.colsum(dx1_32.view(B * Tq, Dd), grad_buffer(ad.decoder.proj.bias))
    dq = torch.empty_like(qp)
    dkv = torch.empty_like(kv)
    sq, skv = (Tq * Dd, Dd), (N * 2 * Dd, 2 * Dd)
    gq, gkv = grad_buffer(ad.decoder.q.bias), grad_buffer(ad.decoder.kv.bias)
    fused = gq is not None and gkv is not None and FUSE_ATTN_BIAS and adrop is None
    K.attn_bwd(qp, kv[..., :Dd], kv[..., Dd:], o, lse, d_o, dq, dkv[..., :Dd], dkv[..., Dd:], B, H, Tq, N, hd, ad.decoder.scale,
               sq, skv, skv, sq, skv, skv, bias_grads=(gq, gkv[:Dd], gkv[Dd:]) if fused else None, drop=adrop)
    d_qn, _ = lin_bwd(dq, qn, ad.decoder.q, bias_grad=not fused)
    d_cn, _ = lin_bwd(dkv, cn, ad.decoder.kv, bias_grad=not fused)
    d_queries, _ = K.ln_bwd(d_qn, queries, mq, rq, ad.query_norm.weight, None, grad_buffer(ad.query_norm.weight),
                            grad_buffer(ad.query_norm.bias), None, want_bf16=False)
    d_context, _ = K.ln_bwd(d_cn, context, mc, rc, ad.context_norm.weight, None, grad_buffer(ad.context_norm.weight),
                            grad_buffer(ad.context_norm.bias), None, want_bf16=False)
    d_te = [grad_buffer(t) if t is not None else None for t in task_embs]
    d_ctx = K.dec_assemble_bwd(desc, d_queries, d_context, d_te, grad_buffer(ad.mask_token), dev)
    d_ctx16 = to_bf16(d_ctx)
    # proj_context: wgrad + bias + dgrad accumulated into the shared encoder-output gradient
    Wp = ad.proj_context
    D = enc16.shape[-1]
    def _pc_grads():
        if Wp.weight.requires_grad:
            K.gemm(d_ctx16, enc16, Dd, D, B * N, a_mode=KS, b_mode=KS, lda=Dd, ldb=D, out=grad_buffer(Wp.weight), ldc=D,
                   splitk=K.pick_splitk(Dd, D, B * N))
        if Wp.bias.requires_grad:
            K.colsum(d_ctx.view(B * N, Dd), grad_buffer(Wp.bias))
    _Side.run(_pc_grads, d_ctx16, enc16, d_ctx)
    K.timestamp(f"dec_bwd_end.{ad.task}")
    return d_ctx16


def decoder_bwd_finish(ad, d_ctx16: torch.Tensor, d_enc32: torch.Tensor, acc32: Optional[torch.Tensor], twin16: Optional[torch.Tensor] = None):
    """d_enc32 = acc32 (the other adapters' sum so far, or None) + proj_context dgrad of this adapter; twin16 (last adapter): the bf16 copy
    of the finished sum that the encoder's backward pass reads"""
    B, N, Dd = d_ctx16.shape
    D = d_enc32.shape[-1]
    K.gemm(d_ctx16, shadow(ad.proj_context.weight), B * N, D, Dd, b_mode=KS, lda=Dd, ldb=D, out=d_enc32, ldc=D, resid=acc32, out2=twin16)


class PredLayout:
    """What the output adapters return.  Default: images [B, C, H, W] (the reference's contract, mirage/output_adapters.py:291-294).
    `tokens = True` (set by trainer.TrainStep around its forward pass): the out_proj output [B, tokens, C*P*P] as it is -- the masked
    losses index it in place (mrg_loss_desc.pred_tokens) and write their gradient in the same layout, so the two rearrangement passes
    per task and direction disappear from the step.  `to_images` converts such a prediction for callers that want the picture."""
    tokens = False

    @staticmethod
    def to_images(pred: torch.Tensor, ad) -> torch.Tensor:
        if pred.dim() != 3:
            return pred
        g = ad.image_size[0] // (ad.stride_level * ad.P_H)
        return K.patch_rearrange(pred.contiguous(), True, pred.shape[0], ad.num_channels, g, ad.P_H)


def loss_mask(m):
    """a task's token mask as the loss kernels read it: int64 [B, tokens], rows possibly strided (the task's columns of the step's mask, in place)"""
    if m is None:
        return None
    if m.dtype != torch.int64:
        m = m.long()
    return m if m.dim() == 2 and m.stride(1) == 1 else m.contiguous()


class LaneLosses:
    """Masked losses computed ON the output adapters' lane streams (set by trainer.TrainStep around its forward pass).  The reference
    computes the task losses after the model call (run_pretraining.py:713-721); as separate autograd nodes they run one after the other
    on the calling stream between the join of the decoders' forward lanes and the fork of their backward lanes -- 150 us (C2) / 300 us
    (C4) of serial small launches.  With `spec` = {task: (kind, patch, target, use_mask)} model.forward hands each adapter its loss:
    (optionally a fifth and sixth entry: the label smoothing of a cross-entropy loss, norm_pix of an MSE loss);
    the lane that ran the adapter runs the loss forward right behind out_proj, DecodersFn returns the loss scalars, and its backward
    starts every lane with that task's loss backward.  Same kernels on the same data: bit-identical to the separate nodes
    (MRG_LANE_LOSSES=0).  `preds` keeps the (detached) predictions of the last pass for callers that read them (parity tests)."""
    spec = None
    preds = {}
    # defer_join (set with `spec` by TrainStep on one GPU): the decoders' forward lanes are not joined into the calling stream; every lane
    # runs straight on from its loss forward into its loss backward when DecodersFn.backward forks again, and that backward's join covers
    # both passes.  The caller seeds the backward with the constant gradient of each task loss (d sum / d loss_t = 1) instead of
    # differentiating the sum, and reads the loss scalars only after the backward pass.
    defer_join = False
    # fuse_bwd (trainer.TrainStep): every lane forms its loss AND the loss gradient in one pass over the predictions (mrg_masked_loss_fused;
    # the backward coefficients come from the mask at the head of the lane, mrg_loss_coef) under the contract that the caller seeds each
    # task loss with the constant `unit` (d sum / d loss_t = 1).  A backward pass seeded with anything else runs the separate gradient launch.
    fuse_bwd = False
    unit = None
    fused_hits = 0      # backward passes (per adapter) that took the forward lane's gradient as it was (tests)


class _Lanes:
    """Independent output adapters run on their own HIP streams (fork from / join into the calling stream): each adapter is a
    chain of ~25 short launches on [B*256, 256]-sized problems that cannot fill 256 CUs on its own."""
    streams = []
    enabled = True
    fwd = bwd = True        # experiment knobs (MRG_LANES_FWD / MRG_LANES_BWD = 0)

    @classmethod
    def get(cls, i, device):
        while len(cls.streams) <= i:
            cls.streams.append(torch.cuda.Stream(device=device))
        if cls.streams[i].device != device:
            cls.streams[i] = torch.cuda.Stream(device=device)
        return cls.streams[i]

    main_first = False      # experiment knob (MRG_LANES_ORDER=1): issue job 0 (calling stream) before the lane jobs
    all_lanes = False

    @classmethod
    def run_all(cls, jobs, device, enabled=None, join=True):
        """jobs: list of zero-arg callables; job 0 runs on the current stream, the others on lane streams.  join=False leaves the lanes
        un-joined: the caller guarantees that a later run_all on the same device (which forks from the calling stream again and joins)
        follows before anything on the calling stream reads what the lanes produce."""
        main = torch.cuda.current_stream(device)
        if len(jobs) == 1 or not (cls.enabled if enabled is None else enabled):
            return [job() for job in jobs]
        ev = torch.cuda.Event()
        ev.record(main)
        out = [None] * len(jobs)
        lanes = []
        if cls.all_lanes:                      # experiment knob (MRG_LANES_ALL=1): every job on a lane stream, none on the caller's
            for i, job in enumerate(jobs):
                st = cls.get(i, device)
                st.wait_event(ev)
                with torch.cuda.stream(st):
                    out[i] = job()
                lanes.append(st)
            for st in lanes:
                main.wait_stream(st)
            return out
        if cls.main_first:
            out[0] = jobs[0]()
        for i, job in enumerate(jobs[1:], start=1):
            st = cls.get(i - 1, device)
            st.wait_event(ev)
            with torch.cuda.stream(st):
                out[i] = job()
            lanes.append(st)
        if not cls.main_first:
            out[0] = jobs[0]()
        if join:
            for st in lanes:
                main.wait_stream(st)
        return out


class DecodersFn(torch.autograd.Function):
    """One or more output adapters in one node: their proj_context dgrads accumulate straight into one d_enc buffer."""

    @staticmethod
    def forward(ctx, enc32, enc16, adapters, dom_names, dom_start, ng, ids_keep, ids_restore, save, defer_join, loss_spec, *params):
        """loss_spec: None, or one (kind, patch, target, mask or None, label smoothing, norm_pix) per adapter (LaneLosses): the node then returns the task LOSSES"""
        ctx.defer_join = defer_join
        enc32 = enc32.contiguous().float()
        if enc16 is None:
            enc16 = to_bf16(enc32)
        doms_l, start_l = list(dom_names), list(dom_start)

        fuse = loss_spec is not None and LaneLosses.fuse_bwd and LaneLosses.unit is not None and save

        # fewer tokens are kept than a modality has => every sample has masked tokens in every modality: the fused launch then forms each
        # sample's coefficient from its own mask row (mrg_loss_desc.nvalid) and the lane needs no mrg_loss_coef launch at its head
        n_keep = ids_keep.shape[1]

        def job(i, ad):
            coef = None
            if fuse:
                kind, patch, target, mask, smooth, npix, *own = loss_spec[i]
                H, W = target.shape[-2:]
                ntok_i = (H // patch) * (W // patch)
                # own[0]: the mask is the mask generator's own for THESE ids_keep (MIRAGEModel._forward says so); a mask from elsewhere
                # carries no such guarantee (a sample without masked pixels would get coefficient 0 while the others still divide by B)
                if not (mask is None or (own and own[0] and n_keep < ntok_i)):
                    coef = K.loss_coef(mask, target.shape[0], ntok_i, patch * patch, enc32.device)
            img, sv = decoder_fwd(ad, enc32, enc16, ids_keep, ids_restore, doms_l, start_l, ng, save)
            if loss_spec is None:
                return img, sv, None
            kind, patch, target, mask, smooth, npix = loss_spec[i][:6]
            d = K.loss_desc(kind, img, target, mask, patch, smoothing=smooth, norm_pix=npix)
            if fuse:
                loss, partial, dpred, finish = K.masked_loss_fused(d, coef, img.device)
                return loss, sv, (d, coef, partial, img, dpred, finish)
            loss, coef, partial = K.masked_loss_fwd(d, img.device)
            return loss, sv, (d, coef, partial, img, None, None)
        jobs = [(lambda i=i, ad=ad: job(i, ad)) for i, ad in enumerate(adapters)]
        res = _Lanes.run_all(jobs, enc32.device, enabled=_Lanes.enabled and _Lanes.fwd,
                             join=not (loss_spec is not None and LaneLosses.defer_join and _Lanes.bwd))
        outs = [r[0] for r in res]
        saved = [r[1] for r in res]
        ctx.adapters, ctx.saved, ctx.shape = adapters, saved, enc32.shape
        ctx.loss_saved = [r[2] for r in res] if loss_spec is not None else None
        if loss_spec is not None:
            LaneLosses.preds = {ad.task: r[2][3] for ad, r in zip(adapters, res)}
        return tuple(outs)

    @staticmethod
    def backward(ctx, *d_imgs):
        _Side.begin_backward()
        dev = next(d for d in d_imgs if d is not None).device
        d_enc = torch.empty(ctx.shape, dtype=F32, device=dev)
        if ctx.loss_saved is not None:
            # d_imgs are the gradients of the task LOSSES (scalars): every lane opens with its loss backward
            todo = [(ad, sv, (g, ls)) for ad, sv, g, ls in zip(ctx.adapters, ctx.saved, d_imgs, ctx.loss_saved) if g is not None]

            def bwd(ad, sv, gl):
                g, (d, coef, partial, img, dpred, finish) = gl
                if finish is not None:
                    _Side.run(finish, partial)                 # the loss VALUE: nobody waits for it before the step ends
                if dpred is not None and LaneLosses.unit is not None and g.data_ptr() == LaneLosses.unit.data_ptr():
                    LaneLosses.fused_hits += 1
                    return decoder_bwd(ad, sv, dpred)          # the forward lane has written the gradient already (seed = 1)
                if coef is None:                               # (the fused launch formed its coefficients itself)
                    coef = K.loss_coef(d._refs[2], d.B, (d.H // d.P) * (d.W // d.P), d.P * d.P, img.device)
                return decoder_bwd(ad, sv, K.masked_loss_bwd(d, coef, partial, g.contiguous().float(), img))
            jobs = [(lambda ad=ad, sv=sv, gl=gl: bwd(ad, sv, gl)) for ad, sv, gl in todo]
            ctx.loss_saved = None
        else:
            todo = [(ad, sv, d_img) for ad, sv, d_img in zip(ctx.adapters, ctx.saved, d_imgs) if d_img is not None]
            jobs = [(lambda ad=ad, sv=sv, d_img=d_img: decoder_bwd(ad, sv, d_img)) for ad, sv, d_img in todo]
        d_ctxs = _Lanes.run_all(jobs, dev, enabled=_Lanes.enabled and _Lanes.bwd)
        # the adapters' proj_context dgrads form one sum: each launch adds the sum so far as its fp32 residual and writes the other of two
        # buffers (no aliasing: a launch stays repeatable for the autotuner); the last one also writes the bf16 twin the encoder reads
        d_enc16 = torch.empty(ctx.shape, dtype=BF16, device=dev) if len(todo) > 1 else None
        bufs, acc = [d_enc, torch.empty_like(d_enc) if len(todo) > 1 else None], None
        for i, ((ad, _, _), d_ctx16) in enumerate(zip(todo, d_ctxs)):
            decoder_bwd_finish(ad, d_ctx16, bufs[i % 2], acc, d_enc16 if i == len(todo) - 1 else None)
            acc = bufs[i % 2]
        d_enc = acc
        if d_enc16 is not None:
            _TWINS.clear()
            _TWINS[d_enc.data_ptr()] = (d_enc, d_enc16)
        ctx.saved = None
        if _Side.early_opt is not None:          # every output-adapter gradient is final after the next reduction flush
            if _Side.enabled:
                _Side.finished.append("decoders")
            else:
                # side stream off: the adapters' weight gradients may still sit in the group collector (two decoder blocks per adapter
                # never reach the flush cadence) -- launch them before the update that consumes them.  (Found in round 3: without a
                # warm-up iteration the early update read zero gradients; with one, the previous iteration's.)
                K.WgradGroup.flush()
                _Side.early_opt(["decoders"])
        if not ctx.defer_join:
            _Side.join()
        return (d_enc,) + (None,) * (len(ctx.needs_input_grad) - 1)


# ------------------------------------------------------------------------------------------- losses
class MaskedLossFn(torch.autograd.Function):
    """MaskedMSELoss / MaskedCrossEntropyLoss (mirage/criterion.py:31-51, 87-117)."""

    @staticmethod
    def forward(ctx, pred, target, mask, kind, patch, smoothing=0.0, norm_pix=False):
        pred = pred.contiguous()
        target = target.contiguous()
        if kind == L.MRG_LOSS_MSE:
            target = target.float()
        if mask is not None:
            mask = mask.contiguous().long()
        d = K.loss_desc(kind, pred, target, mask, patch, smoothing=smoothing, norm_pix=norm_pix)
        loss, coef, partial = K.masked_loss_fwd(d, pred.device)
        ctx.d, ctx.keep = d, (pred, target, mask, coef, partial)
        return loss

    @staticmethod
    def backward(ctx, gout):
        pred, target, mask, coef, partial = ctx.keep
        dpred = K.masked_loss_bwd(ctx.d, coef, partial, gout.contiguous().float(), pred)
        ctx.keep = None
        return dpred, None, None, None, None, None, None
