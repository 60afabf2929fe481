"""Tensor-level launchers for the HIP kernels (one python function per C-ABI entry point).

Everything here takes CUDA(=HIP) torch tensors, passes raw pointers + the current torch stream to
libmirage_hip.so, and returns torch tensors.  No computation happens in torch.
"""
import ctypes as C
import json
import os
import weakref
from typing import Dict, Optional, Sequence

import torch

from . import _lib as L
from ._lib import lib, check

BF16, F32 = torch.bfloat16, torch.float32
_cur = torch.cuda.current_stream


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _s():
    return C.c_void_p(_cur().cuda_stream)


def _need_cuda(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise RuntimeError(f"mirage_amd: {what} must live on the GPU (got {t.device}); there is no CPU path")


# ------------------------------------------------------------------------------------------------ half-batch placement
class Halves:
    """Output placement for two half-batch chains (engine.EncoderFn): while chain A (mode 1) runs, every activation it allocates
    through new() is the first half (along dim 0) of a full-batch tensor; chain B (mode 2) replays the same sequence of
    allocations and receives the second halves.  The full tensors are what the (full-batch) backward pass saves."""
    mode = 0
    fulls = []
    idx = 0

    @classmethod
    def begin(cls, mode: int):
        cls.mode = mode
        if mode == 1:
            cls.fulls = []
        cls.idx = 0

    @classmethod
    def end(cls):
        cls.mode = 0

    @staticmethod
    def full_of(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
        """full-batch tensor of a first-half activation produced under mode 1 (any later .view() of it included)"""
        if t is None or not torch.is_tensor(t):
            return t
        root = t._base if t._base is not None else t
        return root.view((2 * t.shape[0],) + tuple(t.shape[1:]))


def new(shape, dtype, device) -> torch.Tensor:
    """torch.empty for activations that are batch-major along dim 0 (see Halves)."""
    if Halves.mode == 0:
        return torch.empty(shape, dtype=dtype, device=device)
    shape = tuple(shape)
    full_shape = (2 * shape[0],) + shape[1:]
    if Halves.mode == 1:
        f = torch.empty(full_shape, dtype=dtype, device=device)
        Halves.fulls.append(f)
        return f[:shape[0]]
    f = Halves.fulls[Halves.idx]
    Halves.idx += 1
    if tuple(f.shape) != full_shape or f.dtype != dtype:
        raise RuntimeError(f"mirage_amd: half-batch chains diverged ({tuple(f.shape)} {f.dtype} vs {full_shape} {dtype})")
    return f[shape[0]:]


# ------------------------------------------------------------------------------------------------ GEMM
_ws_cache = {}


def _workspace(nbytes: int, device) -> torch.Tensor:
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    ws = _ws_cache.get(key)
    if ws is None or ws.numel() * 4 < nbytes:
        ws = torch.empty((max(nbytes, 1 << 20) + 3) // 4, dtype=F32, device=device)
        _ws_cache[key] = ws
    return ws


_tune_cache = {}
_TUNE_REPS = 6
# hits: optional set collecting the table keys a run actually uses; table_entries / table_used: size of the committed table and the
# distinct shapes of this process that were served from it (everything else was tuned at warm-up and is in "log") -- reported by
# bench.py / run_pretraining so that a run says which variant table it ran on
# MRG_TUNE=0: no table, no first-use autotuning -- every GEMM takes the heuristic kernel choice of mrg_gemm (runs that must pick the same
# kernels in separate processes: first-use autotuning is a timing, and two processes may time differently)
TUNE = {"enabled": os.environ.get("MRG_TUNE", "1") != "0", "log": [], "hits": None, "table_entries": 0, "table_used": set()}
_table_keys = set()
_TUNE_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gemm_tune_gfx950.json")


def load_tune_cache(path: str = _TUNE_FILE) -> int:
    """Pre-tuned (variant, split-K) choices for the BASELINE shapes, measured on MI355X (like a BLAS library's tuned table);
    shapes that are not in the table are tuned on first use."""
    if not os.path.exists(path):
        return 0
    with open(path) as f:
        for k, v in json.load(f).items():
            _tune_cache[tuple(json.loads(k))] = tuple(v)
            _table_keys.add(tuple(json.loads(k)))
    TUNE["table_entries"] = len(_table_keys)
    return len(_tune_cache)


def save_tune_cache(path: str = _TUNE_FILE) -> None:
    with open(path, "w") as f:
        json.dump({json.dumps([int(x) if not isinstance(x, bool) else x for x in k]): list(v) for k, v in sorted(_tune_cache.items())}, f, indent=0)


class ReduceTask(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("n", C.c_long), ("stride", C.c_long), ("nslices", C.c_int), ("reserved", C.c_int)]


class Deferred:
    """Slice reductions postponed to one batched launch (mrg_reduce_batch): the split-K slabs of weight-gradient GEMMs and the
    partial column sums of LayerNorm / bias gradients.  Nothing reads those results before the optimizer, so ~165 tiny
    kernels per training step collapse into one or two.  `active` is switched on by the backward orchestration
    (engine._Side) for the duration of a backward pass; stand-alone kernel calls reduce immediately."""
    active = False
    tasks = []        # (src tensor, byte offset, dst tensor, n, stride, nslices)
    keep = []

    @classmethod
    def add(cls, src: torch.Tensor, src_off_elems: int, dst: torch.Tensor, n: int, stride: int, nslices: int):
        cls.tasks.append((src.data_ptr() + 4 * src_off_elems, dst.data_ptr(), n, stride, nslices))
        cls.keep.append(src); cls.keep.append(dst)

    @classmethod
    def ok(cls, dst: torch.Tensor, n: int, stride: int) -> bool:
        return cls.active and dst is not None and dst.dtype == F32 and dst.is_contiguous() and n % 4 == 0 and stride % 4 == 0 \
            and dst.data_ptr() % 16 == 0

    @classmethod
    def flush(cls):
        if not cls.tasks:
            return
        arr = (ReduceTask * len(cls.tasks))()
        for i, (src, dst, n, stride, ns) in enumerate(cls.tasks):
            arr[i].src, arr[i].dst, arr[i].n, arr[i].stride, arr[i].nslices = src, dst, n, stride, ns
        check(lib.mrg_reduce_batch(C.cast(arr, C.c_void_p), len(cls.tasks), _s()), "mrg_reduce_batch")
        cls.tasks = []

    @classmethod
    def release(cls):
        cls.keep = []


TIMING = {"on": False, "records": []}     # bench.py: (kernel variant, mode, flops, start event, stop event) per GEMM launch


# Arrival counters of the in-launch split-K finish (mrg_gemm_desc.splitk_counters): regions of one zero-initialised int32 ring, one
# region per GEMM call; every launch leaves its region zero again, so a captured graph reuses its regions on every replay and the
# eager path wraps around after ~1M tiles (several steps' worth: no two launches in flight share a region)
# Measured on the C2 step (same box, 30 steps): off 10.90 ms; on for every split-K GEMM 12.79 ms; on for <= 4 slices of >= 64 tiles
# (the encoder weight gradients) 11.21 ms; <= 2 slices 10.95 ms.  The last arriver reads its tile's slabs alone (cross-XCD,
# ~60-100 GB/s per workgroup) inside the weight-gradient kernel, while the batched reduction launch does the same work with the whole
# chip off the critical path -- fewer HBM bytes (the slabs are consumed from cache), more time.  Default: off (A/B: MRG_SPLITK_INKERNEL=1).
SPLITK_INKERNEL = os.environ.get("MRG_SPLITK_INKERNEL", "0") != "0"
SPLITK_INKERNEL_MAX = int(os.environ.get("MRG_SPLITK_INKERNEL_MAX", "4"))
SPLITK_INKERNEL_MIN_TILES = int(os.environ.get("MRG_SPLITK_INKERNEL_MIN_TILES", "64"))
_cnt_ring = {}


def _splitk_counters(ntiles: int, device):
    ring = _cnt_ring.get(device)
    if ring is None:
        ring = [torch.zeros(1 << 20, dtype=torch.int32, device=device), 0]
        _cnt_ring[device] = ring
    buf, pos = ring
    n = (ntiles + 63) // 64 * 64
    if pos + n > buf.numel():
        pos = 0
    ring[1] = pos + n
    return buf.data_ptr() + 4 * pos


def _gemm_launch(d, splitk, variant, dev, defer_out=None):
    d.splitk = splitk
    d.variant = variant
    d.defer_reduce = 0
    d.splitk_counters = None
    if splitk > 1:
        if (SPLITK_INKERNEL and not d.colsum_out and splitk <= SPLITK_INKERNEL_MAX
                and ((d.M + 127) // 128) * ((d.N + 127) // 128) >= SPLITK_INKERNEL_MIN_TILES):
            # slabs in the per-stream workspace (rewritten by every split-K GEMM of the stream, so they stay cache-resident) and
            # reduced by the last-arriving workgroup of each tile inside the launch
            d.ws = _workspace(splitk * d.M * d.N * 4, dev).data_ptr()
            d.splitk_counters = _splitk_counters(((d.M + 127) // 128) * ((d.N + 127) // 128), dev)
        elif defer_out is not None:
            # own slab (kept alive until the batched reduction) instead of the shared workspace
            ws = torch.empty(splitk * d.M * d.N, dtype=F32, device=dev)
            d.ws = ws.data_ptr()
            d.defer_reduce = 1
            Deferred.add(ws, 0, defer_out, d.M * d.N, d.M * d.N, splitk)
        else:
            d.ws = _workspace(splitk * d.M * d.N * 4, dev).data_ptr()
    check(lib.mrg_gemm(C.byref(d), _s()), "mrg_gemm")


def _autotune(d, key, splitk, allow_split, dev):
    """Time the kernel variants (and, for wgrad-like problems, neighbouring split-K factors) once per problem shape and
    remember the fastest.  Runs only outside hipGraph capture; every candidate recomputes the same (idempotent) result."""
    cands = []
    splits = [splitk]
    if allow_split:
        kt = (d.K + 63) // 64
        splits = sorted({s for s in (1, 2, 4, 8, 16, 32) if s == 1 or kt // s >= 4})
    for v in (L.MRG_GEMM_GLDS3, L.MRG_GEMM_GLDS4, L.MRG_GEMM_PIPE128, L.MRG_GEMM_WS256):
        if v == L.MRG_GEMM_WS256 and d.M <= 128:
            continue
        for s_ in splits:
            cands.append((v, s_))
    # 256 x 256 8-phase kernel (gemm8p.hip): one workgroup per CU, so it needs about a CU's worth of tiles per round
    if d.K % 64 == 0 and d.M > 128 and d.N > 128 and not d.colsum_out and not d.k_dev and not d.a_map and not d.m_dev:
        t8 = ((d.M + 255) // 256) * ((d.N + 255) // 256)
        for s_ in splits:
            if t8 * s_ >= 192:
                cands.append((L.MRG_GEMM_P8, s_))
    best, best_t = (L.MRG_GEMM_AUTO, splitk), float("inf")
    for v, s_ in cands:
        _gemm_launch(d, s_, v, dev)
        t = float("inf")
        for _round in range(2):                                    # min of two timed bursts: robust to clock ramp / stragglers
            st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            st.record()
            for _ in range(_TUNE_REPS):
                _gemm_launch(d, s_, v, dev)
            en.record()
            en.synchronize()
            t = min(t, st.elapsed_time(en))
        if t < best_t:
            best, best_t = (v, s_), t
    _tune_cache[key] = best
    TUNE["log"].append((key, best, best_t / _TUNE_REPS * 1e3))
    return best


def gemm(A: torch.Tensor, B: torch.Tensor, M: int, N: int, K: int, *, a_mode=L.MRG_OP_KCONTIG, b_mode=L.MRG_OP_KCONTIG,
         lda: Optional[int] = None, ldb: Optional[int] = None, out: Optional[torch.Tensor] = None, out_dtype=BF16,
         ldc: Optional[int] = None, out2: Optional[torch.Tensor] = None, bias=None, resid=None, accumulate=False,
         act=L.MRG_ACT_NONE, aux_out=None, aux_in=None, row_map=None, add_tab=None, add_row=None, a_map=None, b_map=None,
         m_dev=None, k_dev=None, splitk: int = 1, tune_split: bool = False, row_scale=None, colsum=None, dropout=None) -> torch.Tensor:
    """colsum: optional fp32 [N] destination for the column sums of the values written to `out` (per-tile partial sums in the
    epilogue + batched reduction): the bias gradient of the layer whose output gradient this GEMM produces.
    dropout: optional (rng int64[2] = {seed, step} on the device, p, stream id): nn.Dropout(p) on (acc + bias) fused into the epilogue
    (Philox mask, regenerated by dropout_bwd_bf16 in the backward pass)."""
    _need_cuda(A, "GEMM operand")
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype, device=A.device)
    if (WgradGroup.collecting and a_mode == L.MRG_OP_KSTRIDED and b_mode == L.MRG_OP_KSTRIDED and tune_split
            and WgradGroup.eligible(A, B, M, N, K, lda, ldb, out, ldc, out2, bias, resid, accumulate, act, aux_out, aux_in, row_map, add_tab,
                                    a_map, b_map, m_dev, k_dev, row_scale, colsum)):
        WgradGroup.tasks.append((A, B, M, N, K, lda if lda is not None else A.stride(0), ldb if ldb is not None else B.stride(0), out,
                                 ldc if ldc is not None else out.stride(-2), splitk))
        return out
    d = L.GemmDesc()
    d.A, d.B = A.data_ptr(), B.data_ptr()
    d.M, d.N, d.K = M, N, K
    d.lda = lda if lda is not None else A.stride(0)
    d.ldb = ldb if ldb is not None else B.stride(0)
    d.a_mode, d.b_mode = a_mode, b_mode
    d.a_map, d.b_map, d.m_dev, d.k_dev = _p(a_map), _p(b_map), _p(m_dev), _p(k_dev)
    d.C = out.data_ptr()
    d.ldc = ldc if ldc is not None else out.stride(-2)
    d.c_dtype = L.MRG_DT_F32 if out.dtype == F32 else L.MRG_DT_BF16
    if out2 is not None:
        d.C2, d.ldc2 = out2.data_ptr(), out2.stride(-2)
    d.bias = _p(bias)
    if resid is not None:
        d.resid, d.ldr = resid.data_ptr(), resid.stride(-2)
    d.accumulate = int(accumulate)
    d.act = act
    if aux_out is not None:
        d.aux_out, d.ldaux = aux_out.data_ptr(), aux_out.stride(-2)
    if aux_in is not None:
        d.aux_in, d.ldaux = aux_in.data_ptr(), aux_in.stride(-2)
    d.row_map = _p(row_map)
    if add_tab is not None:
        d.add_tab, d.add_row, d.ldt = add_tab.data_ptr(), add_row.data_ptr(), add_tab.stride(0)
    d.row_scale = _p(row_scale)
    if dropout is not None:
        rng, p_drop, sid = dropout
        if rng.dtype != torch.int64 or rng.numel() != 2 or not rng.is_cuda:
            raise ValueError("gemm(dropout=...): rng must be an int64[2] device tensor {seed, step}")
        d.drop_rng, d.drop_p, d.drop_stream = rng.data_ptr(), float(p_drop), int(sid)
    cs_part = None
    if colsum is not None:
        cs_part = torch.empty(((M + 127) // 128, N), dtype=F32, device=A.device)
        d.colsum_out = cs_part.data_ptr()
    variant = L.MRG_GEMM_AUTO
    tunable = (TUNE["enabled"] and a_map is None and b_map is None and m_dev is None and k_dev is None and not accumulate and M * N >= (1 << 16)
               and dropout is None)
    if tunable:
        key = (a_mode, b_mode, M, N, K, act, resid is not None, out.dtype == F32, tune_split)
        hit = _tune_cache.get(key)
        if TUNE["hits"] is not None:
            TUNE["hits"].add(key)
        if hit is None and not torch.cuda.is_current_stream_capturing():
            hit = _autotune(d, key, splitk, tune_split, A.device)
        if hit is not None:
            variant, splitk = hit
            if key in _table_keys:
                TUNE["table_used"].add(key)
    if TUNE.get("force") is not None:                              # tests / tools: pin one kernel variant
        variant = TUNE["force"]
    if TIMING["on"]:
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st.record(); en.record()                                   # creates the underlying hipEvent_t handles
        d.ev_start, d.ev_stop = st.cuda_event, en.cuda_event
        TIMING["records"].append((variant, (a_mode, b_mode), 2.0 * M * N * K, st, en, (M, N, K, splitk)))
    defer = None
    if splitk > 1 and bias is None and not accumulate and out2 is None and d.ldc == N and (M * N) % 4 == 0 and Deferred.ok(out, M * N, M * N):
        defer = out
    if cs_part is not None and variant not in (L.MRG_GEMM_GLDS3, L.MRG_GEMM_GLDS4):
        variant = L.MRG_GEMM_GLDS4 if (a_mode == L.MRG_OP_KCONTIG and M * N >= 1400 * 16384) else L.MRG_GEMM_GLDS3
    _gemm_launch(d, splitk, variant, A.device, defer)
    if cs_part is not None:
        reduce_slices(cs_part, 0, colsum, N, N, cs_part.shape[0])
    return out


def gemm_multi(problems):
    """Several independent forward-layout products y = A W^T (+ bias) (+ table row at a mapped output row) in ONE launch (mrg_gemm_multi).
    problems: list of dicts with the keyword arguments of gemm() that this entry point accepts (A, B, M, N, K, lda, ldb, out, ldc, bias,
    row_map, add_tab, add_row, m_dev).  Results are those of the separate gemm() calls."""
    n = len(problems)
    if TIMING["on"] or n == 1 or n > L.MRG_MAX_DOMAINS:          # (the instrumented roofline step times every product on its own)
        for pr in problems:
            gemm(pr["A"], pr["B"], pr["M"], pr["N"], pr["K"], lda=pr.get("lda"), ldb=pr.get("ldb"), out=pr["out"], ldc=pr.get("ldc"), bias=pr.get("bias"),
                 row_map=pr.get("row_map"), add_tab=pr.get("add_tab"), add_row=pr.get("add_row"), m_dev=pr.get("m_dev"))
        return
    arr = (L.GemmDesc * n)()
    for d, pr in zip(arr, problems):
        A, B, out = pr["A"], pr["B"], pr["out"]
        _need_cuda(A, "GEMM operand")
        d.A, d.B, d.C = A.data_ptr(), B.data_ptr(), out.data_ptr()
        d.M, d.N, d.K = pr["M"], pr["N"], pr["K"]
        d.lda = pr.get("lda") or A.stride(0)
        d.ldb = pr.get("ldb") or B.stride(0)
        d.ldc = pr.get("ldc") or out.stride(-2)
        d.a_mode = d.b_mode = L.MRG_OP_KCONTIG
        d.c_dtype = L.MRG_DT_F32 if out.dtype == F32 else L.MRG_DT_BF16
        d.bias, d.row_map, d.m_dev = _p(pr.get("bias")), _p(pr.get("row_map")), _p(pr.get("m_dev"))
        if pr.get("add_tab") is not None:
            d.add_tab, d.add_row, d.ldt = pr["add_tab"].data_ptr(), pr["add_row"].data_ptr(), pr["add_tab"].stride(0)
        d.act, d.splitk = L.MRG_ACT_NONE, 1
    check(lib.mrg_gemm_multi(C.cast(arr, C.c_void_p), n, _s()), "mrg_gemm_multi")


class WgradGroup:
    """Weight gradients of several Linears in ONE launch without split-K (mrg_wgrad_group, gemm8p.hip).  While `collecting`
    (engine._Side: the side-stream jobs of a flush), gemm() calls that are plain weight gradients -- both operands contracted over
    the token axis, fp32 overwrite, nothing fused -- are queued instead of launched; flush() launches them grouped by K.  A single
    weight gradient has 9-36 output tiles of 256 x 256: alone it needs split-K slabs and a reduction to fill 256 CUs; side by side
    the gradients of a few blocks fill them with every tile contracting over all tokens (C2: 100-106 us per block against 197 us,
    C4: 130 against 215; tools/debug/exp_group_wgrad.py).  MRG_WGRAD_GROUP=0: the per-GEMM tuned split-K path."""
    enabled = os.environ.get("MRG_WGRAD_GROUP", "1") != "0"
    # smaller groups go through the per-GEMM path; a tile's k-loop is K / 64 steps of ~1.6 us (K = 16384: the output adapters' 256-wide
    # gradients, 38 tiles per adapter for 410 us beside the encoder's backward -- C2 10.05 -> 9.90 ms, C4 21.56 -> 21.44 same box;
    # 64 / 12288 = encoder blocks only)
    min_tiles = int(os.environ.get("MRG_WGRAD_GROUP_MIN_TILES", "16"))
    # optional split-K 2 inside groups with fewer tiles than this (the exposed last groups of a backward pass): measured neutral
    # fuse: set by trainer.TrainStep around a backward pass whose optimizer allows it (optim.AdamWFusion): the grouped launch then applies
    # AdamW to the parameters whose gradients it produces, and `fused` collects their arena ranges for the step's bookkeeping
    fuse = None
    # (C2 9.93 vs 9.95 ms, C4 21.66 vs 21.65: the slabs and their reduction cost what the fuller launch gains) -> off by default
    split_below = int(os.environ.get("MRG_WGRAD_GROUP_SPLIT_BELOW", "0"))
    split_max_k = int(os.environ.get("MRG_WGRAD_GROUP_SPLIT_MAX_K", "12288"))
    max_k = int(os.environ.get("MRG_WGRAD_GROUP_MAX_K", "16384"))
    exposed = False
    collecting = False
    tasks = []
    origin = None      # stream the queued tasks' operands (dy, x) were produced on: the grouped launch must go out on THAT stream

    @classmethod
    def begin(cls):
        """start (or continue) collecting on the current stream; tasks collected from another stream are launched there first (the
        decoders' backward passes run on lane streams: a group may not mix producers that nothing orders against each other)"""
        cur = torch.cuda.current_stream()
        if cls.tasks and cls.origin is not None and cls.origin != cur:
            cls.flush()
        cls.origin = cur
        cls.collecting = cls.enabled

    @staticmethod
    def eligible(A, B, M, N, K, lda, ldb, out, ldc, out2, bias, resid, accumulate, act, aux_out, aux_in, row_map, add_tab, a_map, b_map,
                 m_dev, k_dev, row_scale, colsum) -> bool:
        if (out2 is not None or bias is not None or resid is not None or accumulate or act != L.MRG_ACT_NONE or aux_out is not None
                or aux_in is not None or row_map is not None or add_tab is not None or a_map is not None or b_map is not None
                or m_dev is not None or k_dev is not None or row_scale is not None or colsum is not None):
            return False
        if out.dtype != F32 or K % 64 or K > WgradGroup.max_k or M % 8 or N % 8 or M < 256 or N < 256:
            return False
        la = lda if lda is not None else A.stride(0)
        lb = ldb if ldb is not None else B.stride(0)
        return la % 8 == 0 and lb % 8 == 0 and A.data_ptr() % 16 == 0 and B.data_ptr() % 16 == 0

    @classmethod
    def flush(cls):
        """launch what was collected (on the stream it was collected from) and stop collecting"""
        cls.collecting = False
        tasks, cls.tasks = cls.tasks, []
        if not tasks:
            return
        origin, cls.origin = cls.origin, None
        cur = torch.cuda.current_stream()
        if origin is not None and origin != cur:
            with torch.cuda.stream(origin):
                cls._launch(tasks)
            cur.wait_stream(origin)         # whoever flushes goes on to read the gradients (reductions, optimizer): order it behind them
            return
        cls._launch(tasks)

    @classmethod
    def _launch(cls, tasks):
        by_k = {}
        for t in tasks:
            by_k.setdefault(t[4], []).append(t)
        for K_, ts in by_k.items():
            tiles = sum(((t[2] + 255) // 256) * ((t[3] + 255) // 256) for t in ts)
            if tiles < cls.min_tiles:
                for (A, B, M, N, K, lda, ldb, out, ldc, splitk) in ts:       # too few tiles to fill the chip without split-K
                    gemm(A, B, M, N, K, a_mode=L.MRG_OP_KSTRIDED, b_mode=L.MRG_OP_KSTRIDED, lda=lda, ldb=ldb, out=out, ldc=ldc,
                         splitk=splitk, tune_split=True)
                continue
            for i in range(0, len(ts), L.MRG_WGRAD_GROUP_MAX):
                chunk = ts[i:i + L.MRG_WGRAD_GROUP_MAX]
                ctiles = sum(((t[2] + 255) // 256) * ((t[3] + 255) // 256) for t in chunk)
                # a group that covers less than ~2/3 of the CUs (the last blocks of the backward pass, whose launch is exposed at the
                # end of the step) cuts every tile's k-range in two: slabs + the batched reduction, as the per-GEMM path does
                sk = 2 if (ctiles <= cls.split_below and K_ // 64 >= 32 and K_ <= cls.split_max_k and Deferred.active
                           and all(t[8] == t[3] and Deferred.ok(t[7], t[2] * t[3], t[2] * t[3]) for t in chunk)) else 1
                # (round 6: cutting the k-range of the output adapters' 38-tile group -- K = 16384 tokens, ~400 us on 38 CUs beside the encoder's
                # backward pass -- in four made the launch fuller and the step slower: slabs + reduction, no AdamW in the epilogue; C2 +0.1 ms,
                # C4 +-0, profiles/r06_experiments.md)
                fz = cls.fuse if (sk == 1 and cls.fuse is not None and (cls.exposed or not cls.fuse.only_exposed)) else None
                wgrad_group(chunk, K_, sk, fuse=fz)


def wgrad_group(ts, K: int, splitk: int = 1, fuse=None):
    """one mrg_wgrad_group launch: ts = [(A, B, M, N, K, lda, ldb, out, ldc, _)], out_p[M_p, N_p] = A_p^T B_p (fp32, overwritten);
    splitk > 1: slabs + deferred batched reduction (only inside a backward pass: Deferred.active)"""
    d = L.WgradGroupDesc()
    d.count, d.K = len(ts), K
    fl = 0.0
    fz = None
    if fuse is not None and splitk == 1:
        # AdamW applied by the launch (mrg_wgrad_group_desc.fuse): every output must be a contiguous view into the optimizer's gradient arena
        fz = fuse.descriptor(ts)
        if fz is not None:
            d.fuse = C.cast(C.pointer(fz), C.c_void_p)
            d._fz = fz
    if splitk > 1:
        ws = torch.empty(splitk * sum(t[2] * t[3] for t in ts), dtype=F32, device=ts[0][0].device)
        d.splitk, d.ws = splitk, ws.data_ptr()
        off = 0
        for (_, _, M, N, _, _, _, out, _, _) in ts:
            Deferred.add(ws, splitk * off, out, M * N, M * N, splitk)
            off += M * N
    for p, (A, B, M, N, _, lda, ldb, out, ldc, _s_) in enumerate(ts):
        d.A[p], d.B[p], d.C[p] = A.data_ptr(), B.data_ptr(), out.data_ptr()
        d.M[p], d.N[p], d.lda[p], d.ldb[p], d.ldc[p] = M, N, lda, ldb, ldc
        fl += 2.0 * M * N * K
    if TIMING["on"]:
        st, en = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        st.record(); en.record()
        d.ev_start, d.ev_stop = st.cuda_event, en.cuda_event
        tiles = sum(((t[2] + 255) // 256) * ((t[3] + 255) // 256) for t in ts)
        TIMING["records"].append((L.MRG_GEMM_P8_GROUP, (L.MRG_OP_KSTRIDED, L.MRG_OP_KSTRIDED), fl, st, en, (tiles * 256, 256, K, 1)))
    check(lib.mrg_wgrad_group(C.byref(d), _s()), "mrg_wgrad_group")


if os.environ.get("MRG_TUNE_FRESH") != "1":      # retuning: start from an empty table
    load_tune_cache()


def pick_splitk(M: int, N: int, K: int, target_wgs: int = 320) -> int:
    """wgrad GEMMs have few output tiles and a long K (= tokens): split K until there are ~2 workgroups per CU
    (the pipelined kernel runs 2 per CU), keeping >= 8 k-tiles per workgroup."""
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    ktiles = (K + 63) // 64
    s = 1
    while tiles * s < target_wgs and (ktiles // (s * 2)) >= 8 and s < 32:
        s *= 2
    return s


# ------------------------------------------------------------------------------------------- LayerNorm
def ln_fwd(x: torch.Tensor, gamma, beta, eps: float, out_dtype=BF16):
    _need_cuda(x, "LayerNorm input")
    D = x.shape[-1]
    M = x.numel() // D
    y = new(x.shape, out_dtype, x.device)
    mean = new((M,), F32, x.device)
    rstd = new((M,), F32, x.device)
    check(lib.mrg_ln_fwd(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr() if out_dtype == BF16 else None,
                         y.data_ptr() if out_dtype == F32 else None, mean.data_ptr(), rstd.data_ptr(), M, D, eps, _s()), "mrg_ln_fwd")
    return y, mean, rstd


def ln_bwd(dy: torch.Tensor, x, mean, rstd, gamma, dres, dgamma, dbeta, dres_colsum=None, want_f32=True, want_bf16=True):
    D = x.shape[-1]
    M = x.numel() // D
    dx32 = torch.empty(x.shape, dtype=F32, device=x.device) if want_f32 else None
    dx16 = torch.empty(x.shape, dtype=BF16, device=x.device) if want_bf16 else None
    nb = lib.mrg_ln_bwd_partial_rows(M, D)
    partials = torch.empty(nb * 3 * D, dtype=F32, device=x.device)
    outs = (dgamma, dbeta, dres_colsum)
    defer = Deferred.active and D % 4 == 0 and all(o is None or Deferred.ok(o, D, 3 * D) for o in outs) and any(o is not None for o in outs)
    check(lib.mrg_ln_bwd(dy.data_ptr() if dy.dtype == BF16 else None, dy.data_ptr() if dy.dtype == F32 else None, x.data_ptr(),
                         mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), _p(dres), _p(dx32), _p(dx16), partials.data_ptr(),
                         None if defer else _p(dgamma), None if defer else _p(dbeta), None if defer else _p(dres_colsum), M, D, _s()),
          "mrg_ln_bwd")
    if defer:
        for q, o in enumerate(outs):
            if o is not None:
                Deferred.add(partials, q * D, o, D, 3 * D, nb)
    return dx32, dx16


def rowfuse_ok(M: int, D: int, NH: int) -> bool:
    """shapes the fused row-chain kernels (csrc/rowfuse.hip) accept: 256-wide rows in whole 64-row tiles"""
    return D == 256 and M > 0 and M % 64 == 0 and NH in (256, 512, 768, 1024)


def rowfuse_ln_proj(x32: torch.Tensor, norm, lin, W16: torch.Tensor, save: bool):
    """LayerNorm(x) W^T + b in one launch (mrg_rowfuse_fwd without a0): query_norm + q, context_norm + kv, norm1 + qkv of the output
    adapters.  Returns (y bf16 [..., NH], ln bf16 or None, mean, rstd or None) -- ln / mean / rstd only when `save`."""
    D = x32.shape[-1]
    M = x32.numel() // D
    NH = W16.shape[0]
    dev = x32.device
    y = torch.empty(x32.shape[:-1] + (NH,), dtype=BF16, device=dev)
    d = L.RowFuseDesc()
    d.M, d.NH, d.eps = M, NH, norm.eps
    d.resid = x32.data_ptr()
    d.gamma, d.beta = norm.weight.data_ptr(), norm.bias.data_ptr()
    d.w1, d.b1, d.h_out = W16.data_ptr(), lin.bias.data_ptr(), y.data_ptr()
    ln = mean = rstd = None
    if save:
        ln = torch.empty(x32.shape, dtype=BF16, device=dev)
        mean = torch.empty((M,), dtype=F32, device=dev)
        rstd = torch.empty((M,), dtype=F32, device=dev)
        d.ln_out, d.mean, d.rstd = ln.data_ptr(), mean.data_ptr(), rstd.data_ptr()
    check(lib.mrg_rowfuse_fwd(C.byref(d), _s()), "mrg_rowfuse_fwd")
    return y, ln, mean, rstd


def rowfuse_mlp(o16: torch.Tensor, proj, W0: torch.Tensor, resid: Optional[torch.Tensor], norm, fc1, W1: torch.Tensor, fc2, W2: torch.Tensor,
                save: bool, want_bf16: bool):
    """x = o W0^T + b0 (+ resid); y = x + fc2(gelu(fc1(LayerNorm(x)))) in one launch (mrg_rowfuse_fwd with a0).
    Returns (y32, y16 or None, saved) with saved = (x, ln, mean, rstd, gelu'(h) fp16, gelu(h) bf16) when `save`, else None."""
    D = o16.shape[-1]
    M = o16.numel() // D
    NH = W1.shape[0]
    dev = o16.device
    y32 = torch.empty(o16.shape, dtype=F32, device=dev)
    y16 = torch.empty(o16.shape, dtype=BF16, device=dev) if want_bf16 else None
    d = L.RowFuseDesc()
    d.M, d.NH, d.eps = M, NH, norm.eps
    d.a0, d.w0, d.b0, d.resid = o16.data_ptr(), W0.data_ptr(), proj.bias.data_ptr(), _p(resid)
    d.gamma, d.beta = norm.weight.data_ptr(), norm.bias.data_ptr()
    d.w1, d.b1, d.w2, d.b2 = W1.data_ptr(), fc1.bias.data_ptr(), W2.data_ptr(), fc2.bias.data_ptr()
    d.y32, d.y16 = y32.data_ptr(), _p(y16)
    saved = None
    if save:
        x = torch.empty(o16.shape, dtype=F32, device=dev)
        ln = torch.empty(o16.shape, dtype=BF16, device=dev)
        mean = torch.empty((M,), dtype=F32, device=dev)
        rstd = torch.empty((M,), dtype=F32, device=dev)
        hshape = o16.shape[:-1] + (NH,)
        h_act = torch.empty(hshape, dtype=BF16, device=dev)
        h_grad = torch.empty(hshape, dtype=torch.float16, device=dev)
        d.x_out, d.ln_out, d.mean, d.rstd, d.h_out, d.hgrad_out = (x.data_ptr(), ln.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                                                    h_act.data_ptr(), h_grad.data_ptr())
        saved = (x, ln, mean, rstd, h_grad, h_act)
    check(lib.mrg_rowfuse_fwd(C.byref(d), _s()), "mrg_rowfuse_fwd")
    return y32, y16, saved


def colsum(x: torch.Tensor, out: torch.Tensor, M: Optional[int] = None, row_map=None, m_dev=None):
    N = x.shape[-1]
    M = M if M is not None else x.numel() // N
    nb = lib.mrg_colsum_nblocks(M)
    partials = torch.empty(nb * N, dtype=F32, device=x.device)
    defer = Deferred.ok(out, N, N)
    check(lib.mrg_colsum(x.data_ptr(), L.MRG_DT_BF16 if x.dtype == BF16 else L.MRG_DT_F32, M, N, x.stride(-2), _p(row_map),
                         _p(m_dev), partials.data_ptr(), None if defer else out.data_ptr(), _s()), "mrg_colsum")
    if defer:
        Deferred.add(partials, 0, out, N, N, nb)
    return out


# ------------------------------------------------------------------------------------------- attention
def _attn_desc(q, k, v, o, lse, B, H, Nq, Nk, hd, scale, strides):
    d = L.AttnDesc()
    d.q, d.k, d.v, d.o, d.lse = q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), _p(lse)
    (d.q_bs, d.q_rs), (d.k_bs, d.k_rs), (d.v_bs, d.v_rs), (d.o_bs, d.o_rs) = strides
    d.B, d.H, d.Nq, d.Nk, d.head_dim, d.scale = B, H, Nq, Nk, hd, scale
    return d


def _attn_drop(d, drop):
    """drop = (rng int64[2] device tensor {seed, step}, p, stream id): dropout on the attention probabilities (mrg_attn_desc.drop_*)"""
    if drop is None:
        return
    rng, p, stream = drop
    if rng.dtype != torch.int64 or rng.numel() != 2 or not rng.is_cuda:
        raise ValueError("attention dropout: rng must be an int64[2] device tensor {seed, step}")
    d.drop_rng, d.drop_p, d.drop_stream = rng.data_ptr(), float(p), int(stream)


def attn_fwd(q, k, v, B, H, Nq, Nk, hd, scale, q_strides, k_strides, v_strides, need_lse=True, drop=None):
    """q/k/v: bf16 tensors (possibly views into fused projection buffers); strides = (batch, row) in elements.
    drop: see _attn_drop (the plain fp32 path of csrc/attention_drop.hip)."""
    _need_cuda(q, "attention input")
    o = new((B, Nq, H * hd), BF16, q.device)
    lse = new((B, H, Nq), F32, q.device) if need_lse else None
    d = _attn_desc(q, k, v, o, lse, B, H, Nq, Nk, hd, scale, (q_strides, k_strides, v_strides, (Nq * H * hd, H * hd)))
    _attn_drop(d, drop)
    check(lib.mrg_attn_fwd(C.byref(d), _s()), "mrg_attn_fwd")
    return o, lse


def reduce_slices(src: torch.Tensor, src_off: int, dst: torch.Tensor, n: int, stride: int, nslices: int):
    """dst[i] = sum_s src[src_off + s * stride + i]: queued for the batched reduction while a backward pass is collecting
    (Deferred.active), otherwise one mrg_reduce_batch launch now."""
    if Deferred.ok(dst, n, stride):
        Deferred.add(src, src_off, dst, n, stride, nslices)
        return
    t = (ReduceTask * 1)()
    t[0].src, t[0].dst, t[0].n, t[0].stride, t[0].nslices = src.data_ptr() + 4 * src_off, dst.data_ptr(), n, stride, nslices
    check(lib.mrg_reduce_batch(C.cast(t, C.c_void_p), 1, _s()), "mrg_reduce_batch")


def attn_bwd(q, k, v, o, lse, d_o, dq, dk, dv, B, H, Nq, Nk, hd, scale, q_strides, k_strides, v_strides, dq_strides, dk_strides,
             dv_strides, bias_grads=None, drop=None):
    """bias_grads = (gq, gk, gv): fp32 [H*hd] destinations for the column sums of dQ / dK / dV over all rows (the bias
    gradients of the projections that produced q / k / v); formed inside the kernels per (batch, head) and summed over the batch.
    drop: the forward pass's (rng, p, stream) -- the mask is regenerated; not together with bias_grads."""
    if drop is not None and bias_grads is not None:
        raise ValueError("attn_bwd: the dropout path does not form the bias column sums")
    d = _attn_desc(q, k, v, o, lse, B, H, Nq, Nk, hd, scale, (q_strides, k_strides, v_strides, (Nq * H * hd, H * hd)))
    _attn_drop(d, drop)
    part = None
    if bias_grads is not None:
        Dm = H * hd
        part = torch.empty((B, 3 * Dm), dtype=F32, device=q.device)
        d.colsum_q, d.colsum_k, d.colsum_v = part.data_ptr(), part.data_ptr() + 4 * Dm, part.data_ptr() + 8 * Dm
        d.colsum_bs = 3 * Dm
    dsum = torch.empty((B, H, Nq), dtype=F32, device=q.device)
    d.d_o, d.do_bs, d.do_rs = d_o.data_ptr(), d_o.stride(0), d_o.stride(1)
    d.dsum = dsum.data_ptr()
    d.dq, d.dk, d.dv = dq.data_ptr(), dk.data_ptr(), dv.data_ptr()
    (d.dq_bs, d.dq_rs), (d.dk_bs, d.dk_rs), (d.dv_bs, d.dv_rs) = dq_strides, dk_strides, dv_strides
    check(lib.mrg_attn_bwd(C.byref(d), _s()), "mrg_attn_bwd")
    if part is not None:
        Dm = H * hd
        for i, g in enumerate(bias_grads):
            if g is not None:
                reduce_slices(part, i * Dm, g, Dm, 3 * Dm, B)


# ------------------------------------------------------------------------------------------- embedding
def _int_array(vals: Sequence[int]):
    return (C.c_int * len(vals))(*vals)


def gather_rows_bf16(src: torch.Tensor, row_map: torch.Tensor, count_dev: torch.Tensor, cap: int) -> torch.Tensor:
    """bf16 [*, D] -> compact bf16 [cap, D]: rows row_map[0 .. count) of src, zero beyond the (device-side) count."""
    D = src.shape[-1]
    out = torch.empty((cap, D), dtype=BF16, device=src.device)
    check(lib.mrg_gather_rows_bf16(src.data_ptr(), src.stride(-2), row_map.data_ptr(), count_dev.data_ptr(), cap, D, out.data_ptr(), _s()),
          "mrg_gather_rows_bf16")
    return out


def posemb_grad(dx32: torch.Tensor, out: torch.Tensor, B: int, T: int, plan=None):
    """out (the [1, D, h, w] gradient buffer of a learnable pos_emb) = per-token sums of the embedding-gradient rows dx32 [rows, D];
    plan = (rows, toks, srcb, count_dev, cap) of one modality (mrg_embed_plan), or None when dx32 is [B * T, D] with every token present."""
    D = dx32.shape[-1]
    if plan is None:
        check(lib.mrg_posemb_grad(dx32.data_ptr(), dx32.stride(-2), D, None, None, None, None, 0, B, T, None, out.data_ptr(), _s()),
              "mrg_posemb_grad")
        return out
    rows, toks, srcb, cnt, cap = plan
    inv = torch.empty(B * T, dtype=torch.int32, device=dx32.device)
    check(lib.mrg_posemb_grad(dx32.data_ptr(), dx32.stride(-2), D, rows.data_ptr(), toks.data_ptr(), srcb.data_ptr(), cnt.data_ptr(), cap,
                              B, T, inv.data_ptr(), out.data_ptr(), _s()), "mrg_posemb_grad")
    return out


def embed_plan(ids_keep: torch.Tensor, N: int, dom_start: Sequence[int]):
    _need_cuda(ids_keep, "ids_keep")
    B, K = ids_keep.shape
    ndom = len(dom_start) - 1
    cap = B * K
    dev = ids_keep.device
    rows = torch.empty((ndom, cap), dtype=torch.int32, device=dev)
    toks = torch.empty((ndom, cap), dtype=torch.int32, device=dev)
    srcb = torch.empty((ndom, cap), dtype=torch.int32, device=dev)
    counts = torch.empty(ndom, dtype=torch.int32, device=dev)
    check(lib.mrg_embed_plan(ids_keep.data_ptr(), B, K, N, ndom, _int_array(list(dom_start)), rows.data_ptr(), toks.data_ptr(),
                             srcb.data_ptr(), counts.data_ptr(), _s()), "mrg_embed_plan")
    return rows, toks, srcb, counts


# ------------------------------------------------------------------------------------------- input slots
_SLOT_OF_PTR: Dict[int, int] = {}      # data_ptr of a placeholder tensor -> address of its slot (see InputSlots)


def slot_of(t: Optional[torch.Tensor]) -> Optional[int]:
    """Address of the input slot a placeholder tensor stands for, or None: the wrappers of the kernels that read raw input batches
    (patch packers, class-embedding gradient, masked losses) pass it along, and the kernel then reads the batch at the address the
    slot holds when it RUNS instead of at t.data_ptr()."""
    return _SLOT_OF_PTR.get(t.data_ptr()) if (t is not None and _SLOT_OF_PTR) else None


class InputSlots:
    """Device table of input-batch base addresses (include/mirage_hip.h "Input slots").  `placeholders` {name: tensor} fix shape,
    dtype and device of each input and stand for "whatever batch is bound" wherever the engine meets them (their data_ptr is the
    registry key, so views / .contiguous() of a placeholder resolve too); `bind` points the slots at a batch without copying it.  A
    captured step therefore reads every new batch in place.  The bound batch is kept alive until the next bind."""

    def __init__(self, placeholders: Dict[str, torch.Tensor]):
        self.names = list(placeholders)
        if not 1 <= len(self.names) <= 8:
            raise ValueError("InputSlots: 1 to 8 inputs")
        self.placeholders = dict(placeholders)
        dev = next(iter(placeholders.values())).device
        for k, t in placeholders.items():
            _need_cuda(t, f"input placeholder {k}")
            if not t.is_contiguous() or t.device != dev:
                raise ValueError(f"InputSlots: placeholder {k} must be contiguous and on {dev}")
        self.table = torch.zeros(len(self.names), dtype=torch.int64, device=dev)
        self._bound_ptrs, self._held = None, None
        self._keys = []
        for i, k in enumerate(self.names):
            key = placeholders[k].data_ptr()
            if key in _SLOT_OF_PTR:
                raise RuntimeError(f"InputSlots: placeholder {k} already stands for another slot")
            _SLOT_OF_PTR[key] = self.table.data_ptr() + 8 * i
            self._keys.append(key)
        weakref.finalize(self, InputSlots._release, list(self._keys))
        self.bind(self.placeholders)

    @staticmethod
    def _release(keys):
        for k in keys:
            _SLOT_OF_PTR.pop(k, None)

    def close(self):
        InputSlots._release(self._keys)

    def bind(self, batch: Dict[str, torch.Tensor]):
        ptrs = []
        for k in self.names:
            t, ref = batch[k], self.placeholders[k]
            if t.shape != ref.shape or t.dtype != ref.dtype or t.device != ref.device or not t.is_contiguous():
                raise ValueError(f"InputSlots.bind: {k} must be a contiguous {tuple(ref.shape)} {ref.dtype} tensor on {ref.device}, "
                                 f"got {tuple(t.shape)} {t.dtype} on {t.device}" + ("" if t.is_contiguous() else " (not contiguous)"))
            if t.data_ptr() % 16:
                raise ValueError(f"InputSlots.bind: {k} is not 16-byte aligned")
            ptrs.append(t.data_ptr())
        if ptrs != self._bound_ptrs:
            arr = (C.c_void_p * len(ptrs))(*ptrs)
            check(lib.mrg_set_slots(self.table.data_ptr(), arr, len(ptrs), _s()), "mrg_set_slots")
            self._bound_ptrs = ptrs
        cur = torch.cuda.current_stream()
        for k in self.names:
            if batch[k] is not self.placeholders[k]:
                batch[k].record_stream(cur)        # produced on a loader's stream: its memory must outlive this stream's reads
        self._held = {k: batch[k] for k in self.names}


def im2col_image(img: torch.Tensor, srcb, toks, count, cap: int, P: int) -> torch.Tensor:
    B, Cc, H, W = img.shape
    out = torch.empty((cap, Cc * P * P), dtype=BF16, device=img.device)
    check(lib.mrg_im2col_image(img.data_ptr(), srcb.data_ptr(), toks.data_ptr(), count.data_ptr(), cap, Cc, H, W, P, out.data_ptr(),
                               slot_of(img), _s()), "mrg_im2col_image")
    return out


def im2col_layermap(labels: torch.Tensor, class_emb: torch.Tensor, srcb, toks, count, cap: int, P: int) -> torch.Tensor:
    B, Hs, Ws = labels.shape
    ncls, E = class_emb.shape
    out = torch.empty((cap, E * P * P), dtype=BF16, device=labels.device)
    check(lib.mrg_im2col_layermap(labels.data_ptr(), class_emb.data_ptr(), srcb.data_ptr(), toks.data_ptr(), count.data_ptr(), cap, ncls,
                                  E, Hs, Ws, P, out.data_ptr(), slot_of(labels), _s()), "mrg_im2col_layermap")
    return out


def class_emb_grad(dA: torch.Tensor, labels, srcb, toks, count, ncls: int, E: int, P: int, out: torch.Tensor):
    B, Hs, Ws = labels.shape
    partials = torch.empty(lib.mrg_class_emb_grad_slices() * ncls * E, dtype=F32, device=dA.device)
    check(lib.mrg_class_emb_grad(dA.data_ptr(), labels.data_ptr(), srcb.data_ptr(), toks.data_ptr(), count.data_ptr(), ncls, E, Hs, Ws, P,
                                 partials.data_ptr(), out.data_ptr(), slot_of(labels), _s()), "mrg_class_emb_grad")
    return out


def fill_global(x: torch.Tensor, glob: torch.Tensor, B: int, N: int, K: int):
    ng, D = glob.shape[-2], glob.shape[-1]
    check(lib.mrg_fill_global(x.data_ptr(), glob.data_ptr(), B, N, K, ng, D, _s()), "mrg_fill_global")


# --------------------------------------------------------------------------------------------- decoder
def dec_assemble_desc(ctx, mask_token, task_embs, pos, ids_keep, ids_restore, dom_start, qdom, B, N, K, T, Tq, Dd, q_task_emb=None):
    d = L.DecAssembleDesc()
    d.ctx, d.mask_token, d.pos = _p(ctx), mask_token.data_ptr(), pos.data_ptr()
    for i, t in enumerate(task_embs):
        d.task_emb[i] = _p(t)
    d.ids_keep, d.ids_restore = ids_keep.data_ptr(), ids_restore.data_ptr()
    for i, v in enumerate(dom_start):
        d.dom_start[i] = v
    d.ndom, d.qdom, d.q_task_emb = len(dom_start) - 1, qdom, _p(q_task_emb)
    d.B, d.N, d.K, d.T, d.Tq, d.Dd = B, N, K, T, Tq, Dd
    # the descriptor outlives this call (it is reused by the backward pass): keep the tensors behind its raw pointers alive
    d._refs = (ctx, mask_token, list(task_embs), pos, ids_keep, ids_restore, q_task_emb)
    return d


def dec_assemble_fwd(d: "L.DecAssembleDesc", device):
    queries = torch.empty((d.B, d.Tq, d.Dd), dtype=F32, device=device)
    context = torch.empty((d.B, d.N, d.Dd), dtype=F32, device=device)
    d.queries, d.context = queries.data_ptr(), context.data_ptr()
    check(lib.mrg_dec_assemble_fwd(C.byref(d), _s()), "mrg_dec_assemble_fwd")
    return queries, context


def dec_assemble_bwd(d: "L.DecAssembleDesc", d_queries, d_context, d_task_embs, d_mask_token, device):
    d_ctx = torch.empty((d.B, d.N, d.Dd), dtype=F32, device=device)
    nb = lib.mrg_dec_assemble_bwd_nblocks(d.B, d.N, d.Tq)
    partials = torch.empty(nb * (L.MRG_MAX_DOMAINS + 1) * d.Dd, dtype=F32, device=device)
    d.d_queries, d.d_context, d.d_ctx, d.partials = d_queries.data_ptr(), d_context.data_ptr(), d_ctx.data_ptr(), partials.data_ptr()
    for i, t in enumerate(d_task_embs):
        d.d_task_emb[i] = _p(t)
    d.d_mask_token = _p(d_mask_token)
    check(lib.mrg_dec_assemble_bwd(C.byref(d), _s()), "mrg_dec_assemble_bwd")
    return d_ctx


# ---------------------------------------------------------------------------------------------- losses
def loss_desc(kind, pred, target, mask, P, smoothing: float = 0.0, norm_pix: bool = False):
    """smoothing: label smoothing of the cross entropy (criterion.py:22,33); norm_pix: per-patch normalised MSE targets (criterion.py:92-98)"""
    _need_cuda(pred, "loss input")
    d = L.LossDesc()
    d.label_smoothing = float(smoothing) if kind == L.MRG_LOSS_CE else 0.0
    if pred.dim() == 3:
        # predictions in the decoder's token layout [B, tokens, C*P*P] (engine.DecodersFn with token_layout): geometry from the target
        H, W = target.shape[-2:]
        B, ntok, inner = pred.shape
        Cc = inner // (P * P)
        if ntok != (H // P) * (W // P) or Cc * P * P != inner or (kind == L.MRG_LOSS_MSE and target.shape[1] != Cc):
            raise ValueError(f"loss_desc: token-layout predictions {tuple(pred.shape)} do not match target {tuple(target.shape)} at patch {P}")
        d.pred_tokens = 1
    else:
        B, Cc, H, W = pred.shape
    d.kind, d.pred, d.pred_dtype = kind, pred.data_ptr(), (L.MRG_DT_BF16 if pred.dtype == BF16 else L.MRG_DT_F32)
    d.target, d.mask, d.target_slot = target.data_ptr(), _p(mask), slot_of(target)
    d.B, d.C, d.H, d.W, d.P = B, Cc, H, W, P
    if mask is not None:
        # a task's columns of the step's [B, all tokens] mask are read in place (mrg_loss_desc.mask_bs)
        if mask.dtype != torch.int64 or mask.dim() != 2 or mask.stride(1) != 1 or tuple(mask.shape) != (B, (H // P) * (W // P)):
            raise ValueError(f"loss_desc: mask must be int64 [B, tokens] with unit token stride, got {mask.dtype} {tuple(mask.shape)} strides {mask.stride()}")
        d.mask_bs = mask.stride(0)
    stats = None
    if norm_pix and kind == L.MRG_LOSS_MSE:
        stats = torch.empty(B * (H // P) * (W // P) * 2, dtype=F32, device=pred.device)
        d.norm_pix, d.stats = 1, stats.data_ptr()
    d._refs = (pred, target, mask, stats)
    return d


def masked_loss_fwd(d: "L.LossDesc", device):
    partial = torch.empty(lib.mrg_loss_partial_floats(d.B), dtype=F32, device=device)
    loss = torch.empty((), dtype=F32, device=device)
    coef = torch.empty(d.B, dtype=F32, device=device)
    d.partial, d.loss, d.coef = partial.data_ptr(), loss.data_ptr(), coef.data_ptr()
    check(lib.mrg_masked_loss_fwd(C.byref(d), _s()), "mrg_masked_loss_fwd")
    return loss, coef, partial


def masked_loss_bwd(d: "L.LossDesc", coef, partial, gout, pred):
    dpred = torch.empty_like(pred)
    d.coef, d.partial, d.gout, d.dpred = coef.data_ptr(), partial.data_ptr(), _p(gout), dpred.data_ptr()
    check(lib.mrg_masked_loss_bwd(C.byref(d), _s()), "mrg_masked_loss_bwd")
    return dpred


def loss_coef(mask, B: int, ntok: int, ppt: int, device):
    """Backward coefficients of a masked loss from the token mask alone (mrg_loss_coef): available as soon as the masks are.
    mask: int64 [B, ntok] (rows may be strided) or None (every pixel counts)."""
    coef = torch.empty(B, dtype=F32, device=device)
    if mask is not None and (mask.dtype != torch.int64 or mask.stride(1) != 1 or tuple(mask.shape) != (B, ntok)):
        raise ValueError(f"loss_coef: mask must be int64 [{B}, {ntok}] with unit token stride")
    check(lib.mrg_loss_coef(_p(mask), mask.stride(0) if mask is not None else 0, B, ntok, ppt, coef.data_ptr(), _s()), "mrg_loss_coef")
    return coef


def masked_loss_fused(d: "L.LossDesc", coef, device, gout=None):
    """The gradient of a masked loss (seeded with gout, None = 1) and the partial sums of its value in one pass over the predictions
    (mrg_masked_loss_fused).  coef: mrg_loss_coef's coefficients, or None under the caller's guarantee that EVERY sample has masked pixels
    (the launch then forms them from each sample's own mask row).  Returns (loss, partial, dpred, finish): `loss` holds the value once finish() -- the batch statistics, a tiny
    launch nothing in a training step waits for -- has run on some stream ordered behind this call."""
    partial = torch.empty(lib.mrg_loss_partial_floats(d.B), dtype=F32, device=device)
    loss = torch.full((), float("nan"), dtype=F32, device=device)     # NaN until finish() has run: a premature read is detectable
    pred = d._refs[0]
    dpred = torch.empty_like(pred)
    d.partial, d.loss, d.coef, d.gout, d.dpred = partial.data_ptr(), loss.data_ptr(), _p(coef), _p(gout), dpred.data_ptr()
    d.nvalid = 0 if coef is not None else d.B      # coef None: the caller guarantees that every sample has masked pixels
    check(lib.mrg_masked_loss_fused(C.byref(d), _s()), "mrg_masked_loss_fused")

    def finish():
        d.partial, d.loss = partial.data_ptr(), loss.data_ptr()
        check(lib.mrg_masked_loss_finish(C.byref(d), _s()), "mrg_masked_loss_finish")
    return loss, partial, dpred, finish


def patch_rearrange(src: torch.Tensor, to_image: bool, B: int, Cc: int, g: int, P: int) -> torch.Tensor:
    shape = (B, Cc, g * P, g * P) if to_image else (B, g * g, Cc * P * P)
    dst = torch.empty(shape, dtype=src.dtype, device=src.device)
    check(lib.mrg_patch_rearrange(src.data_ptr(), dst.data_ptr(), L.MRG_DT_BF16 if src.dtype == BF16 else L.MRG_DT_F32, int(to_image),
                                  B, Cc, g, P, _s()), "mrg_patch_rearrange")
    return dst


# --------------------------------------------------------------------------------------------- maskgen
def maskgen(p: torch.Tensor, u: torch.Tensor, g: torch.Tensor, K: int, dom_start: Sequence[int]):
    _need_cuda(p, "mask noise")
    B, T = g.shape
    dev = g.device
    task_mask = torch.empty((B, T), dtype=torch.int64, device=dev)
    ids_keep = torch.empty((B, K), dtype=torch.int64, device=dev)
    ids_restore = torch.empty((B, T), dtype=torch.int64, device=dev)
    check(lib.mrg_maskgen(p.data_ptr(), u.data_ptr(), g.data_ptr(), B, T, K, len(dom_start) - 1, _int_array(list(dom_start)),
                          task_mask.data_ptr(), ids_keep.data_ptr(), ids_restore.data_ptr(), _s()), "mrg_maskgen")
    return task_mask, ids_keep, ids_restore


def maskgen_rng(conc: torch.Tensor, rng: torch.Tensor, K: int, T: int, dom_start: Sequence[int], return_noise: bool = False):
    """The masks of `maskgen` with the noise drawn inside the launch (csrc/maskgen.hip): conc fp32 [B, ndom]; rng int64 [3] on the
    device = {seed, step, 0}, advanced by the launch itself.  return_noise: also (p, u, g) as drawn (tests)."""
    _need_cuda(conc, "Dirichlet concentrations")
    if rng.dtype != torch.int64 or rng.numel() != 3 or rng.device != conc.device or not rng.is_contiguous():
        raise ValueError("maskgen_rng: rng must be a contiguous int64[3] tensor {seed, step, 0} on the masks' device")
    if conc.dtype != F32 or not conc.is_contiguous():
        raise ValueError("maskgen_rng: conc must be contiguous fp32 [B, ndom]")
    B, ndom = conc.shape
    dev = conc.device
    task_mask = torch.empty((B, T), dtype=torch.int64, device=dev)
    ids_keep = torch.empty((B, K), dtype=torch.int64, device=dev)
    ids_restore = torch.empty((B, T), dtype=torch.int64, device=dev)
    noise = (torch.empty((B, ndom), dtype=F32, device=dev), torch.empty((B, T), dtype=F32, device=dev),
             torch.empty((B, T), dtype=F32, device=dev)) if return_noise else (None, None, None)
    check(lib.mrg_maskgen_rng(conc.data_ptr(), rng.data_ptr(), B, T, K, ndom, _int_array(list(dom_start)), task_mask.data_ptr(),
                              ids_keep.data_ptr(), ids_restore.data_ptr(), *[(t.data_ptr() if t is not None else None) for t in noise],
                              _s()), "mrg_maskgen_rng")
    return (task_mask, ids_keep, ids_restore) + ((noise,) if return_noise else ())


# ------------------------------------------------------------------------------------------- optimizer
def grad_norm(g: torch.Tensor, out: torch.Tensor):
    n = g.numel()
    partials = torch.empty(lib.mrg_sumsq_nblocks(n), dtype=F32, device=g.device)
    check(lib.mrg_grad_norm(g.data_ptr(), n, partials.data_ptr(), out.data_ptr(), _s()), "mrg_grad_norm")
    return out


def adamw(p, g, m, v, shadow, n_decay, lr_table, wd_table, step, beta1, beta2, eps, lr_scale=1.0, grad_norm_t=None, clip_norm=0.0,
          skip_norm=0.0, sumsq_out=None):
    """sumsq_out: optional fp32 tensor with >= adamw_nblocks(p.numel()) entries receiving this launch's partial sums of g^2."""
    d = L.AdamWDesc()
    d.p, d.g, d.m, d.v, d.shadow_bf16 = p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), _p(shadow)
    d.n, d.n_decay = p.numel(), n_decay
    d.lr_table, d.wd_table, d.step, d.table_len = lr_table.data_ptr(), wd_table.data_ptr(), step.data_ptr(), lr_table.numel()
    d.lr_scale, d.beta1, d.beta2, d.eps = lr_scale, beta1, beta2, eps
    d.grad_norm, d.clip_norm, d.skip_norm = _p(grad_norm_t), clip_norm or 0.0, skip_norm or 0.0
    d.sumsq_out = _p(sumsq_out)
    check(lib.mrg_adamw(C.byref(d), _s()), "mrg_adamw")


def adamw_hp(lr_table, wd_table, step, beta1, beta2, out: torch.Tensor, lr_scale=1.0):
    """out fp32 [4] = {lr, wd, 1 / (1 - beta1^t), 1 / sqrt(1 - beta2^t)} of the current step (mrg_adamw_hp), for the fused update of
    mrg_wgrad_group"""
    d = L.AdamWDesc()
    d.lr_table, d.wd_table, d.step, d.table_len = lr_table.data_ptr(), wd_table.data_ptr(), step.data_ptr(), lr_table.numel()
    d.lr_scale, d.beta1, d.beta2 = lr_scale, beta1, beta2
    check(lib.mrg_adamw_hp(C.byref(d), out.data_ptr(), _s()), "mrg_adamw_hp")


def adamw_nblocks(n: int) -> int:
    return lib.mrg_adamw_nblocks(n)


def sumsq_finish(partials: torch.Tensor, n: int, out: torch.Tensor):
    check(lib.mrg_sumsq_finish(partials.data_ptr(), n, out.data_ptr(), _s()), "mrg_sumsq_finish")


def step_inc(step: torch.Tensor, grad_norm_t=None, clip_norm=0.0, skip_norm=0.0):
    """step: int32[2] = (iteration counter, AdamW updates applied); the second only advances when the iteration was not skipped."""
    assert step.numel() == 2 and step.dtype == torch.int32
    check(lib.mrg_step_inc(step.data_ptr(), _p(grad_norm_t), clip_norm or 0.0, skip_norm or 0.0, _s()), "mrg_step_inc")


def cast_bf16(src: torch.Tensor, dst: torch.Tensor):
    check(lib.mrg_cast_bf16(src.data_ptr(), dst.data_ptr(), src.numel(), _s()), "mrg_cast_bf16")
    return dst


def dropout_bwd_bf16(src: torch.Tensor, dropout, row_scale: Optional[torch.Tensor] = None) -> torch.Tensor:
    """bf16(src * keep_mask / (1 - p) (* row_scale[r])) for fp32 src [..., D]: the gradient entering a branch that ended in nn.Dropout;
    `dropout` = the (rng, p, stream id) triple the forward GEMM was given."""
    rng, p_drop, sid = dropout
    D = src.shape[-1]
    M = src.numel() // D
    dst = torch.empty(src.shape, dtype=BF16, device=src.device)
    check(lib.mrg_dropout_bwd_bf16(src.data_ptr(), _p(row_scale), dst.data_ptr(), M, D, rng.data_ptr(), float(p_drop), int(sid), _s()),
          "mrg_dropout_bwd_bf16")
    return dst


def row_scale_bf16(src: torch.Tensor, row_scale: torch.Tensor) -> torch.Tensor:
    """bf16(src[r, :] * row_scale[r]) for fp32 src [..., D] flattened to rows."""
    D = src.shape[-1]
    M = src.numel() // D
    dst = torch.empty(src.shape, dtype=BF16, device=src.device)
    check(lib.mrg_row_scale_bf16(src.data_ptr(), row_scale.data_ptr(), dst.data_ptr(), M, D, _s()), "mrg_row_scale_bf16")
    return dst


# ------------------------------------------------------------------------------------------- data path (F2)
def _augment(src: torch.Tensor, params: torch.Tensor, do_affine: bool, label: bool) -> torch.Tensor:
    _need_cuda(src, "augmentation input")
    assert src.dtype == torch.uint8 and src.is_contiguous() and params.dtype == F32 and params.is_contiguous()
    B, H, W = src.shape
    assert params.shape == (B, 8)
    dst = torch.empty((B, H, W), dtype=torch.int64 if label else F32, device=src.device)
    fn = lib.mrg_augment_label if label else lib.mrg_augment_image
    check(fn(src.data_ptr(), dst.data_ptr(), params.data_ptr(), B, H, W, int(bool(do_affine)), _s()),
          "mrg_augment_label" if label else "mrg_augment_image")
    return dst


def augment_image(src_u8: torch.Tensor, params: torch.Tensor, do_affine: bool = True) -> torch.Tensor:
    """uint8 [B, H, W] -> fp32 [B, H, W] in [0, 1]: flip, intensity shift + clip, bilinear affine with zero fill."""
    return _augment(src_u8, params, do_affine, False)


def augment_label(src_u8: torch.Tensor, params: torch.Tensor, do_affine: bool = True) -> torch.Tensor:
    """uint8 [B, H, W] class ids -> int64 [B, H, W]: flip, bilinear affine + round (the reference's label path)."""
    return _augment(src_u8, params, do_affine, True)



def resize_image(x: torch.Tensor, size) -> torch.Tensor:
    """fp32 [B, Hi, Wi] -> [B, Ho, Wo], bilinear with antialiasing (TF.resize on tensors; datasets_pretrain.py:76-81)."""
    B, Hi, Wi = x.shape
    Ho, Wo = int(size[0]), int(size[1])
    out = torch.empty((B, Ho, Wo), dtype=F32, device=x.device)
    check(lib.mrg_resize_image(x.contiguous().data_ptr(), out.data_ptr(), B, Hi, Wi, Ho, Wo, _s()), "mrg_resize_image")
    return out


def resize_label(x: torch.Tensor, size) -> torch.Tensor:
    """int64 [B, Hi, Wi] -> [B, Ho, Wo], nearest (TF.resize(..., NEAREST) for label maps; datasets_pretrain.py:71-81)."""
    B, Hi, Wi = x.shape
    Ho, Wo = int(size[0]), int(size[1])
    out = torch.empty((B, Ho, Wo), dtype=torch.int64, device=x.device)
    check(lib.mrg_resize_label(x.contiguous().data_ptr(), out.data_ptr(), B, Hi, Wi, Ho, Wo, _s()), "mrg_resize_label")
    return out


# ------------------------------------------------------------------------------------------- phase timestamps (profiling aid)
TS = {"buf": None, "names": [], "on": os.environ.get("MRG_TS") == "1"}


def timestamp_seq(prefix: str) -> str:
    """MRG_TS=1: a name unique within the step for a point that is reached several times (numbered in call order since step_start)"""
    if not TS["on"]:
        return prefix
    TS["seq"] = TS.get("seq", 0) + 1
    return f"{prefix}{TS['seq']}"


def timestamp(name: str):
    """MRG_TS=1: record the device wall clock when the current stream reaches this point (graph-capturable); tools/step_phases.py"""
    if not TS["on"]:
        return
    if name == "step_start":
        TS["seq"] = 0
    if TS["buf"] is None:
        TS["buf"] = torch.zeros(256, dtype=torch.int64, device="cuda")
    if name not in TS["names"]:
        TS["names"].append(name)
    i = TS["names"].index(name)
    check(lib.mrg_debug_timestamp(TS["buf"].data_ptr() + 8 * i, _s()), "mrg_debug_timestamp")
