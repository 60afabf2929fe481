#!/usr/bin/env python3
"""Headline benchmark: MIRAGE pretraining images/sec on N MI355X of one node (BASELINE.json metric).

A "step" is one full optimizer step of the pretraining hot path on one batch of synthetic device-resident inputs:
mask generation -> gather-then-embed -> encoder -> decoders -> masked losses -> backward -> [grad all-reduce] ->
global grad-norm + fused AdamW.  Nothing is skipped inside the timed region.

    python bench.py --gpus 1 --steps 20 --warmup 5                      # C2: Base 512/32 bscan+slo, B=64, K=128
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 ... bench.py --gpus 8 ...

Prints ONE JSON line on rank 0 (see README / the driver contract) with `roofline` (dominant kernel, measured live with
events on the launch stream in an instrumented eager step) and, at N=1, `cpu_baseline` (the CPU oracle timed on the
host cores on a bounded sample of the same workload).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

MFMA_BF16_PEAK = 2.5e15      # dense bf16 MFMA peak, FLOP/s (MI355X_MICROARCH.md)

CONFIGS = {
    #        model             in_domains                              B    K    train
    "c2": ("miragepre_base", ("bscan", "slo"), 64, 128, True),
    "c3": ("miragepre_base", ("bscan", "slo", "bscanlayermap"), 64, 98, True),
    "c4": ("miragepre_large", ("bscan", "slo", "bscanlayermap"), 64, 98, True),
    "c5": ("miragepre_large", ("bscan",), 256, 256, False),
}
DIMS = {"miragepre_base": (768, 12, 12), "miragepre_large": (1024, 24, 16)}


def flops_per_image(model_name, doms, K, train):
    """Algorithmic matmul FLOPs per image (SURVEY.md section 8d / BASELINE.md section 3): encoder over N = K+1 tokens,
    patch-embed for the K kept tokens only (expected share K/T per modality), one decoder per output domain."""
    D, L, _ = DIMS[model_name]
    N = K + 1
    Dd, Tq = 256, 256
    f_enc = L * (24 * N * D * D + 4 * N * N * D)
    T = 256 * len(doms)
    f_emb_img = f_emb_lm = 0.0
    for d in doms:
        kept = K * 256 / T
        if d == "bscanlayermap":
            f_emb_lm += kept * 2 * 4096 * D
        else:
            f_emb_img += kept * 2 * 1024 * D
    if not train:
        return f_enc + f_emb_img + f_emb_lm
    f_dec = 0.0
    for d in doms:
        out = 832 if d == "bscanlayermap" else 1024
        f_dec += (2 * N * D * Dd + 4 * Tq * Dd * Dd + 4 * N * Dd * Dd + 4 * Tq * N * Dd + 16 * Tq * Dd * Dd
                  + 2 * (24 * Tq * Dd * Dd + 4 * Tq * Tq * Dd) + 2 * Tq * Dd * out)
    return 3 * (f_enc + f_dec) + 2 * f_emb_img + 3 * f_emb_lm


def synth_batches(doms, B, n, seed, device):
    rng = np.random.default_rng(seed)
    pool = []
    for _ in range(n):
        b = {}
        for d in doms:
            if d == "bscanlayermap":
                b[d] = torch.from_numpy(rng.integers(0, 13, (B, 128, 128), dtype=np.int64)).to(device)
            else:
                b[d] = torch.from_numpy(rng.random((B, 1, 512, 512), dtype=np.float32)).to(device)
        pool.append(b)
    return pool


HBM_ACHIEVABLE = 6.3e12      # B/s: what a streaming kernel sustains of the 8 TB/s HBM3E peak (MI355X_MICROARCH.md)


def algorithmic_bytes(model_name, doms, B, K, train):
    """HBM bytes ONE step has to move with the present dataflow (every operand read once, every result written once; fp32 residual stream,
    bf16 GEMM operands, activations saved for the backward pass), per GPU.  Encoder block at M = B (K + 1) rows of width D
    (profiles/history.md section 6.1, per row and column: forward 72 B = LN in 4 / out 2, qkv 2 + 6, attention 6 + 2, proj 2 + 4 + 4, LN 4 + 2,
    fc1 2 + 8 + 8, fc2 8 + 4 + 4; backward 141.9 B = dgrad chain, both LayerNorm backwards at 16 B, attention backward, the weight gradients'
    operand reads) + 30 B per weight for AdamW inside the weight-gradient epilogues (gradient, parameter, two moments read / written, bf16
    shadow); output adapters, patch embedding, losses and the rest of the optimizer: 2.0 GB per adapter at B = 64 (C2: 19.0 GB encoder + 4.0 =
    the 23.0 GB of DESIGN section 6).  Forward-only (C5): 64 B per row and column (no GELU' side tensor), weights once per block."""
    D, L, _ = DIMS[model_name]
    M = B * (K + 1)
    if not train:
        return L * (64.0 * M * D + 24.0 * D * D)
    return L * ((72.0 + 141.9) * M * D + 30.0 * 12 * D * D) + 2.0e9 * len(doms) * B / 64.0


def step_hbm(model_name, doms, B, K, train, ms_per_step, fpi, measured=None):
    """the whole step against BOTH of its floors: algorithmic bytes / achievable HBM bandwidth and algorithmic flops / bf16 MFMA peak"""
    by = algorithmic_bytes(model_name, doms, B, K, train)
    t = ms_per_step * 1e-3
    hbm_floor, mfma_floor = by / HBM_ACHIEVABLE, B * fpi / MFMA_BF16_PEAK
    out = {"algorithmic_gb": round(by / 1e9, 2), "measured_gb": measured["measured_gb"] if measured else None,
           "achieved_tbps": round(by / t / 1e12, 3), "achievable_tbps": HBM_ACHIEVABLE / 1e12, "frac_of_6.3TBps": round(by / t / HBM_ACHIEVABLE, 4),
           "hbm_floor_ms": round(hbm_floor * 1e3, 3), "mfma_floor_ms": round(mfma_floor * 1e3, 3),
           "binding_floor": "hbm" if hbm_floor > mfma_floor else "mfma", "frac_of_binding_floor": round(max(hbm_floor, mfma_floor) / t, 4),
           "flop_per_byte": round(B * fpi / by, 1), "ridge_flop_per_byte": round(MFMA_BF16_PEAK / HBM_ACHIEVABLE, 1)}
    if measured:
        out["measured_frac_of_6.3TBps"] = round(measured["measured_gb"] * 1e9 / t / HBM_ACHIEVABLE, 4)
        out["measured_source"] = measured["source"]
    return out


def step_traffic(cfg_name):
    """{'measured_gb', 'algorithmic_gb', 'ratio', 'source'} of one step from the committed PMC pass of the configuration, or None"""
    alg = {"c2": 23.0}.get(cfg_name)
    for tname in (f"r06_{cfg_name}_step_traffic.json", f"r05_{cfg_name}_step_traffic.json"):
        f = os.path.join(ROOT, "profiles", tname)
        if os.path.exists(f):
            t = json.load(open(f))
            if alg is None:
                model_name, doms, B, K, train = CONFIGS[cfg_name]
                alg = round(algorithmic_bytes(model_name, doms, B, K, train) / 1e9, 1)
            return {"measured_gb": round(t["total_gb"], 2), "read_gb": round(t["read_gb"], 2), "written_gb": round(t["written_gb"], 2),
                    "algorithmic_gb": alg, "ratio": round(t["total_gb"] / alg, 3), "source": "profiles/" + tname}
    return None


def cpu_baseline(model_name, doms, K, train, budget_s=25.0):
    """CPU oracle (parity-pinned restatement of the reference) on the host cores; bounded sample of the same workload."""
    from oracle import mirage_oracle as O
    n_threads = min(os.cpu_count() or 1, 32)   # more threads than this only oversubscribe the small CPU matmuls
    torch.set_num_threads(n_threads)
    cfg = O.make_config("base" if model_name.endswith("base") else "large", in_domains=doms, out_domains=doms if train else ())
    sd = {k: v.clone().requires_grad_(O.param_shapes(cfg)[k][1]) for k, v in O.procedural_state_dict(cfg).items()}
    B = 16 if train else 8
    x = O.synthetic_batch(cfg, B, seed=1234)
    _, _, ids_keep, ids_restore = O.synthetic_masks(cfg, B, K, seed=99)
    times = []
    t_start = time.perf_counter()
    it = 0
    while True:
        t0 = time.perf_counter()
        if train:
            for p in sd.values():
                p.grad = None
            loss, _, _, _ = O.forward_loss(cfg, sd, x, ids_keep, ids_restore)
            loss.backward()
        else:
            with torch.no_grad():
                O.forward(cfg, sd, {d: x[d] for d in doms}, ids_keep, ids_restore, decode=False)
        dt = time.perf_counter() - t0
        it += 1
        if it > 2:                                    # 2 warm-up iterations
            times.append(dt)
        if len(times) >= 3 or (time.perf_counter() - t_start > budget_s and len(times) >= 1):
            break
    best = min(times)
    # ... and once more on EVERY host thread (SURVEY section 8d says os.cpu_count()): the small CPU matmuls of this workload do not scale past
    # ~32 threads, which is why `value` is the 32-thread figure; the all-thread figure is reported beside it
    all_cores = None
    n_all = os.cpu_count() or 1
    if n_all > n_threads and os.environ.get("MRG_CPU_ALL_CORES") == "1":      # opt-in: measured once on a 256-thread host, 0.09 images/s (three orders
        # of magnitude of oversubscription: minutes per iteration), which would take the default run far beyond "a few minutes"
        torch.set_num_threads(n_all)
        t_all = []
        for i in range(3):
            t0 = time.perf_counter()
            if train:
                for p in sd.values():
                    p.grad = None
                loss, _, _, _ = O.forward_loss(cfg, sd, x, ids_keep, ids_restore)
                loss.backward()
            else:
                with torch.no_grad():
                    O.forward(cfg, sd, {d: x[d] for d in doms}, ids_keep, ids_restore, decode=False)
            if i > 0:
                t_all.append(time.perf_counter() - t0)
            if time.perf_counter() - t_start > 2.2 * budget_s and t_all:
                break
        all_cores = {"value": round(B / min(t_all), 3), "cores": n_all}
        torch.set_num_threads(n_threads)
    return {"value": round(B / best, 3), "unit": "images/sec", "cores": n_threads, "host_cores": os.cpu_count(), "kind": "port", "all_cores": all_cores,
            "sample": f"CPU oracle fp32, B={B}, {'fwd+loss+bwd' if train else 'fwd'}, 2 warm-up + {len(times)} timed iterations (best), "
                      f"{n_threads} threads of {os.cpu_count()} host cores"}


def measure(a, cfg_name, steps, warmup, batch, rank, world, dev, force_dp, ranks_seen, light=False):
    """Build one configuration from scratch, warm it up, capture it, time `steps` steps (barrier + synchronize on both sides, max over
    ranks) and run the instrumented roofline step.  Returns the JSON-ready result dict (rank 0; None elsewhere).
    light: timing only ({ms_per_step, value}, on every rank) -- the per-rank reference step of an N > 1 run."""
    from mirage_amd import builder, kernels as K, optim
    from mirage_amd import _lib as L
    from mirage_amd.dp import GradReducer, broadcast_params
    from mirage_amd.trainer import TrainStep

    model_name, doms, B, Kk, train = CONFIGS[cfg_name]
    if batch:
        B = batch
    out_doms = doms if train else ()
    if os.environ.get("MRG_BENCH_OUT"):                     # experiment knob: fewer output adapters (marginal cost of a decoder)
        out_doms = tuple(os.environ["MRG_BENCH_OUT"].split("-"))
    args = builder.make_args(doms, out_doms)
    torch.manual_seed(0)
    model = builder.get_model(args, model_name, with_decoders=train).to(dev)
    model.train(train)
    # weights come from the common seed above (and are broadcast from rank 0); mask / stochastic-depth draws use a per-rank stream
    # ("own mask RNG stream (seed+rank)")
    torch.manual_seed(4321 + rank)
    torch.cuda.manual_seed(4321 + rank)
    pool = synth_batches(doms, B, 4, 1234 + rank, dev)

    if train:
        total_steps = steps + warmup + 64
        lr = 1e-4 * (B * world) / 256                      # run_pretraining.py:497 with the global batch
        lrs = optim.cosine_scheduler(lr, 0.0, 1, total_steps, warmup_epochs=1, warmup_steps=max(1, total_steps // 10))
        wds = optim.cosine_scheduler(0.05, 0.05, 1, total_steps)
        opt = optim.FusedAdamW(model, lrs, wds, betas=(0.9, 0.95), eps=1e-8)
        broadcast_params(opt.arena.flat_p)
        opt.arena.refresh_shadow(force=True)
        reducer = GradReducer(opt.arena.flat_g, force=force_dp) if (world > 1 or force_dp) else None
        loss_fns = builder.get_losses(args)
        ts = TrainStep(model, loss_fns, opt, pool[0], Kk, use_graph=not a.no_graph, reducer=reducer,
                       overlap=os.environ.get("MRG_DP_OVERLAP", "1") != "0")
        ts.load(pool[0])
        ts.warmup_and_capture()

        def run_step(i):
            ts.step(pool[i % len(pool)])
    else:
        from mirage_amd import kernels as _Kn
        static = {d: torch.empty_like(pool[0][d]) for d in doms}
        slots = _Kn.InputSlots(static)          # the captured forward reads each pool batch in place (include/mirage_hip.h "Input slots")
        graph = None

        def fwd():
            with torch.no_grad():
                return model(static, mask_inputs=False)[0]

        for d in doms:
            static[d].copy_(pool[0][d])
        fwd()
        torch.cuda.synchronize()
        if not a.no_graph:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out_static = fwd()

        def run_step(i):
            slots.bind(pool[i % len(pool)])
            if graph is not None:
                graph.replay()
            else:
                fwd()

    multi = world > 1 or force_dp
    for i in range(warmup):
        run_step(i)
    if multi:
        dist.barrier()
    torch.cuda.synchronize()
    if multi and train and reducer is not None:
        reducer.exposed = []
        reducer.timing = True
    t0 = time.perf_counter()
    for i in range(steps):
        run_step(warmup + i)
    torch.cuda.synchronize()
    own = time.perf_counter() - t0              # this rank's own clock, before it waits for the slowest
    if multi:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    dp_info = None
    if multi:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # per-rank detail for the first real multi-GPU run: every rank's own step time and the part of it the compute stream spent
        # blocked on the gradient exchange (event pair around GradReducer.wait()), next to the exchange format that produced it
        exposed = 0.0
        if train and reducer is not None:
            reducer.timing = False
            exposed = reducer.exposed_ms() / steps
        mine = torch.tensor([1e3 * own / steps, exposed], device=dev, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
        dist.all_gather(allr, mine)
        dp_info = {"ranks_ms_per_step": [round(float(x[0]), 3) for x in allr],
                   "exposed_comm_ms_per_step": [round(float(x[1]), 3) for x in allr]}
        if train and reducer is not None:
            dp_info.update(reducer.describe())
            dp_info.update({"overlap": bool(ts.split), "encoder_cuts": list(ts.cuts), "early_adamw_per_stage": bool(ts._stage_merged)})
    ms_per_step = 1e3 * elapsed / steps
    value = B * world * steps / elapsed
    fpi = flops_per_image(model_name, doms, Kk, train)
    if light:
        return {"ms_per_step": round(ms_per_step, 3), "value": round(value, 2)}

    result = None
    if rank == 0:
        # ---- roofline of the dominant kernel: instrumented eager step, events on the launch stream -------------
        K.TIMING["records"].clear()
        K.TIMING["on"] = True
        # park the GPU behind a long spin kernel so that the whole eager step is already queued when it starts executing:
        # the per-launch events then measure kernel durations, not the python launch gaps
        torch.cuda._sleep(int(0.12 * 2.4e9))
        # weight-gradient GEMMs stay on the launch stream for this one step: in the graph-replayed steps the kernels run (almost)
        # exclusively (profiles/r01_c2_timeline.md), whereas a fully pre-queued eager step would overlap the two streams
        # maximally and inflate every per-launch duration
        from mirage_amd import engine as _E
        side_was, chains_was = _E._Side.enabled, _E.TWO_CHAINS
        _E._Side.enabled = False
        # ... and the forward runs as ONE chain of full-batch launches (production: two half-batch chains on two streams, whose
        # launches overlap pairwise and therefore have no exclusive duration to divide the flops by)
        _E.TWO_CHAINS = 0
        try:
            if train:
                # forward + losses + backward only, with the reducer detached: rank 0 runs this step alone, so nothing in it
                # may be collective (the other ranks are already waiting at the final barrier); the optimizer is skipped because
                # the gradients of a detached step are this rank's only
                keep = (ts.reducer, ts.split, ts._early, getattr(model, "_mrg_split_at", None))
                ts.reducer, ts.split, ts._early = None, None, None
                model._mrg_split_at = None
                try:
                    ts._fwd_bwd()
                    _E.join_side_stream()
                finally:
                    ts.reducer, ts.split, ts._early, model._mrg_split_at = keep
            else:
                fwd()
        finally:
            K.TIMING["on"] = False
            _E._Side.enabled, _E.TWO_CHAINS = side_was, chains_was
        torch.cuda.synchronize()
        vname = {0: "auto", 1: "gemm_glds_kernel", 2: "gemm_glds_kernel", 3: "gemm_pipe_kernel", 7: "gemm_ws256_kernel", 11: "gemm_p8_kernel",
                 100: "gemm_p8_group_kernel"}
        mname = {(0, 0): "forward x*W^T", (0, 1): "dgrad dy*W", (1, 1): "wgrad dy^T*x"}
        groups = {}
        shapes = {}
        for variant, mode, fl, s, e, shp in K.TIMING["records"]:
            sg = shapes.setdefault((mode, shp, variant), [0.0, 0])
            sg[0] += s.elapsed_time(e) * 1e-3; sg[1] += 1
            g = groups.setdefault(vname.get(variant, "auto"), [0.0, 0.0, 0, {}])
            dt = s.elapsed_time(e) * 1e-3
            g[0] += fl; g[1] += dt; g[2] += 1
            g[3][mname[mode]] = g[3].get(mname[mode], 0) + 1
        if os.environ.get("MRG_GEMM_TABLE") == "1":                   # per-shape GEMM time of one (multi-stream, eager) step
            for (mode, shp, variant), (tt, n) in sorted(shapes.items(), key=lambda kv: -kv[1][0]):
                print(f"gemm {mname[mode]:14s} {shp[0]:6d}x{shp[1]:5d}x{shp[2]:6d} sk={shp[3]:2d} v={variant} n={n:3d} total {tt * 1e3:7.3f} ms "
                      f"avg {tt / n * 1e6:7.1f} us {2.0 * shp[0] * shp[1] * shp[2] * n / tt / 1e12:6.0f} TF/s", file=sys.stderr)
        dom = max(groups, key=lambda m: groups[m][1])
        fl, tsec, n, modes = groups[dom]
        all_fl = sum(g[0] for g in groups.values()); all_t = sum(g[1] for g in groups.values())
        # HBM bytes per launch of the dominant kernel come from a separate rocprofv3 --pmc pass (counters cannot be read from
        # inside the process): the committed summary of that pass is quoted and its file named in `traffic_source`
        traffic, traffic_src = None, None
        for tname in ("r05_c2_pmc_traffic.json", "r04_c2_pmc_traffic.json", "r03_c2_pmc_traffic.json", "r02_c2_pmc_traffic.json", "r01_c2_pmc_traffic.json"):
            tfile = os.path.join(ROOT, "profiles", tname)
            if cfg_name == "c2" and os.path.exists(tfile):
                tj = json.load(open(tfile))
                cand = [v for k, v in tj.items() if k.startswith(dom.split("<")[0])]
                if cand:
                    traffic = int(sum(v["hbm_bytes_per_launch"] * v["launches"] for v in cand) / max(1, sum(v["launches"] for v in cand)))
                    traffic_src = "profiles/" + tname
                    break
        roofline = {"bound": "mfma", "kernel": dom, "achieved": round(fl / tsec / 1e12, 2), "peak": MFMA_BF16_PEAK / 1e12,
                    "unit": "TFLOP/s", "frac": round(fl / tsec / MFMA_BF16_PEAK, 4), "traffic": traffic, "traffic_source": traffic_src,
                    "launches_per_step": n, "avg_launch_us": round(1e6 * tsec / n, 2), "launch_mix": modes,
                    "all_gemm_achieved": round(all_fl / all_t / 1e12, 2),
                    # every GEMM kernel family of the instrumented step, largest total time first (the dominant one above is the first):
                    # launches, total ms, achieved TFLOP/s and fraction of the bf16 MFMA peak
                    "families": [{"kernel": m, "launches": gg[2], "ms": round(gg[1] * 1e3, 3), "achieved": round(gg[0] / gg[1] / 1e12, 1),
                                  "frac": round(gg[0] / gg[1] / MFMA_BF16_PEAK, 4)}
                                 for m, gg in sorted(groups.items(), key=lambda kv: -kv[1][1])],
                    # HBM bytes of ONE whole step (PMC pass over a graph-replayed step, profiles/r05_c2_step_traffic.json) against the bytes the
                    # step's dataflow needs (encoder dataflow 19.0 GB + adapters / optimizer ~4 GB at C2): the family-average
                    # `traffic` above hides the re-fetches, this ratio does not
                    "step_traffic": step_traffic(cfg_name),
                    # the WHOLE step against both of its floors (round 6): at the algorithmic level C2 and C4 sit below the ridge (215 / 260
                    # flop per byte against 397), i.e. the step's floor is HBM time, not MFMA time -- both fractions belong in the line
                    "hbm": step_hbm(model_name, doms, B, Kk, train, ms_per_step, fpi, step_traffic(cfg_name)),
                    "whole_step_achieved": round(value / world * fpi / 1e12, 2),
                    "whole_step_frac": round(value / world * fpi / MFMA_BF16_PEAK, 4)}
        result = {
            "metric": "pretrain images/sec (node) MIRAGE-Base 512/32 bscan+slo; MFMA % of peak" if cfg_name == "c2" else f"images/sec (node) {cfg_name}",
            "value": round(value, 2), "unit": "images/sec", "n_gpus": world, "ranks_seen": ranks_seen, "steps": steps, "warmup": warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"{cfg_name}: {model_name} 512/32 {'+'.join(doms)}, per-GPU batch {B}, {Kk} of {256 * len(doms)} tokens encoded, "
                                   f"{'fwd+loss+bwd+AdamW' if train else 'encoder forward only'}",
                       "global_batch": B * world, "parallelism": f"dp{world}", "hipgraph": not a.no_graph,
                       "forward_chains": 2 if (_E.TWO_CHAINS == 2 or (_E.TWO_CHAINS == 1 and not train)) else 1,
                       "gflop_per_image": round(fpi / 1e9, 2)},
            "mfma_frac_of_peak": round(value / world * fpi / MFMA_BF16_PEAK, 4),
            "roofline": roofline,
        }
        if train:
            result["final_loss"] = round(float(ts.loss), 5)
        if dp_info is not None:
            result["dp"] = dp_info
        if os.environ.get("MRG_TUNE_LOG") == "1":
            names = {1: "glds3", 2: "glds4", 3: "pipe128", 4: "pipe256", 5: "pipe256x3", 6: "pipe256sq", 7: "ws256", 8: "glds64", 10: "g256", 11: "p8", 12: "p8sk"}
            for key, (v, sk), us in K.TUNE["log"]:
                am, bm, M_, N_, K_ = key[:5]
                print(f"tune mode=({am},{bm}) {M_}x{N_}x{K_} act={key[5]} -> {names.get(v, v)} splitk={sk} {us:.1f} us {2.0 * M_ * N_ * K_ / us / 1e6:.0f} TF/s",
                      file=sys.stderr)
        result["gemm_table"] = {"file": os.path.basename(K._TUNE_FILE), "entries": K.TUNE["table_entries"], "shapes_from_table": len(K.TUNE["table_used"]),
                                "shapes_tuned_at_warmup": len(K.TUNE["log"])}
    return result


def main():
    # stdout carries exactly one JSON line: native libraries write there too (RCCL prints a version banner through C stdio
    # when its first communicator comes up), so fd 1 points at stderr until the result is ready
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    try:
        _run(real_stdout)
    finally:
        sys.stdout.flush()
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        os.dup2(real_stdout, 1)
        os.close(real_stdout)


def _self_launch(n: int, real_stdout: int) -> int:
    """Start n ranks of this script (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment), wait for all of them and
    relay rank 0's JSON line.  Runs before anything in this process touches the GPU; the children are fresh processes."""
    import socket
    import subprocess
    if os.environ.get("MRG_SHARE_GPU") != "1" and torch.cuda.device_count() < n:
        print(f"bench.py: --gpus {n} but only {torch.cuda.device_count()} GPUs are visible", file=sys.stderr)
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0 = procs[0].stdout.read().decode()
    rcs = [p.wait() for p in procs]
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if bad:
        print(f"bench.py: ranks failed (rank, exit code): {bad}", file=sys.stderr)
        return 1
    lines = [l for l in out0.splitlines() if l.startswith("{")]
    if len(lines) != 1:
        print(f"bench.py: rank 0 printed {len(lines)} JSON lines", file=sys.stderr)
        return 1
    _emit(real_stdout, lines[0])
    return 0


def _emit(real_stdout: int, line: str):
    sys.stdout.flush()
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    os.write(real_stdout, (line + "\n").encode())


def _run(real_stdout: int):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch override (debug)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the C4 / C5 measurements attached to the default C2 line")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: this process has not touched the GPU yet, so it only spawns the N ranks
        # (fresh child processes, one per GPU) and relays rank 0's JSON line
        sys.exit(_self_launch(a.gpus, real_stdout))
    if os.environ.get("MRG_LANES") == "0" or os.environ.get("MRG_SIDE") == "0":     # experiment knobs: stream structure
        from mirage_amd import engine as _E2
        if os.environ.get("MRG_LANES") == "0":
            _E2._Lanes.enabled = False
        if os.environ.get("MRG_SIDE") == "0":
            _E2._Side.enabled = False
    if any(os.environ.get(k) for k in ("MRG_LANES_FWD", "MRG_LANES_BWD", "MRG_LANES_ORDER", "MRG_LANES_ALL")):
        from mirage_amd import engine as _E5
        _E5._Lanes.fwd = os.environ.get("MRG_LANES_FWD", "1") != "0"
        _E5._Lanes.bwd = os.environ.get("MRG_LANES_BWD", "1") != "0"
        _E5._Lanes.main_first = os.environ.get("MRG_LANES_ORDER", "0") == "1"
        _E5._Lanes.all_lanes = os.environ.get("MRG_LANES_ALL", "0") == "1"
    if os.environ.get("MRG_TAPER") == "0" or os.environ.get("MRG_REDUCE_EARLY") == "0":
        from mirage_amd import engine as _E3
        _E3._Side.taper = os.environ.get("MRG_TAPER") != "0"
        _E3._Side.reduce_early = os.environ.get("MRG_REDUCE_EARLY") != "0"
    if os.environ.get("MRG_ATTN_BIAS") == "0" or os.environ.get("MRG_EPI_COLSUM") == "0":
        from mirage_amd import engine as _E4
        _E4.FUSE_ATTN_BIAS = os.environ.get("MRG_ATTN_BIAS") != "0"
        _E4.FUSE_EPI_COLSUM = os.environ.get("MRG_EPI_COLSUM") != "0"
    if os.environ.get("MRG_EMBED_GATHER") == "0":
        from mirage_amd import engine as _E6
        _E6.EMBED_WGRAD_GATHER = False
    if os.environ.get("MRG_SIDE_BLOCKS"):          # experiment knob: transformer blocks per side-stream fork
        from mirage_amd import engine as _E
        _E._Side.blocks_per_flush = int(os.environ["MRG_SIDE_BLOCKS"])
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                         f"(torch.distributed.run --nproc-per-node {a.gpus}) or drop WORLD_SIZE and let bench.py spawn them")
    share = os.environ.get("MRG_SHARE_GPU") == "1"      # test hook: several ranks on one GPU (use with MRG_DIST_BACKEND=gloo)
    if share:
        local_rank = 0
    elif torch.cuda.device_count() <= local_rank:        # device_count() does not initialise the GPU
        raise SystemExit(f"bench.py: rank {rank} needs GPU {local_rank} but only {torch.cuda.device_count()} are visible")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    force_dp = os.environ.get("MRG_FORCE_DP") == "1"      # single-GPU smoke test of the multi-rank code path (world size 1)
    if force_dp and world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29541")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
    if world > 1 or force_dp:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import datetime
        backend = os.environ.get("MRG_DIST_BACKEND", "nccl")
        # a rank that never joins makes the rendezvous fail after the timeout instead of producing a number for fewer GPUs
        tmo = datetime.timedelta(seconds=int(os.environ.get("MRG_DIST_TIMEOUT", "600")))
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, timeout=tmo)
        else:
            dist.init_process_group(backend, timeout=tmo)
        if dist.get_world_size() != a.gpus and not force_dp:
            raise SystemExit(f"bench.py: {dist.get_world_size()} ranks joined, --gpus {a.gpus} asked for")
    ranks_seen = dist.get_world_size() if dist.is_initialized() else 1

    from mirage_amd import kernels as K
    result = measure(a, a.config, a.steps, a.warmup, a.batch, rank, world, dev, force_dp, ranks_seen)
    if (world > 1 or force_dp) and CONFIGS[a.config][4] and os.environ.get("MRG_DP_LOCAL_REF", "1") != "0":
        # What the N-rank step is to be compared with, measured by EVERY rank on its own GPU in the same process: the same configuration
        # without the gradient exchange and with separate update launches (with N > 1 the update cannot ride in the weight-gradient
        # epilogues; other_configs.c4.unfused_update is the same number on one GPU).  step(N) / this = the cost of
        # communication that the overlap did not hide, per rank.
        import gc
        from mirage_amd import trainer as _T
        gc.collect(); torch.cuda.empty_cache()
        was, _T.FUSE_OPT = _T.FUSE_OPT, False
        try:
            loc = measure(a, a.config, a.steps, a.warmup, a.batch, rank, 1, dev, False, 1, light=True)
        finally:
            _T.FUSE_OPT = was
        mine = torch.tensor([loc["ms_per_step"]], device=dev, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
        dist.all_gather(allr, mine)
        if rank == 0 and result is not None and "dp" in result:
            result["dp"]["local_unfused_ms_per_step"] = [round(float(x[0]), 3) for x in allr]
            result["dp"]["local_unfused_note"] = ("every rank's own step without the gradient exchange and with separate update launches "
                                                  "(MRG_FUSE_OPT=0): the per-GPU step the N-rank step is to be read against")
    if rank == 0:
        model_name, doms, _, Kk, train = CONFIGS[a.config]
        # The 40 % target of BASELINE.json is stated on the Large model: the same single line also carries the Large configurations
        # (C4: the pre_mirage_98_1600e shape at per-GPU batch 64; C5: encoder-only forward at batch 256), each built from scratch in
        # this process after the headline's buffers are gone and timed exactly like the headline (warm-up, graph capture, K steps
        # between synchronisations).  The headline value / config above stay C2.
        if world == 1 and a.config == "c2" and not a.batch and not a.no_other_configs and not force_dp:
            import gc
            others = {}
            for name, (st, wu) in (("c4", (20, 5)), ("c5", (10, 3))):
                gc.collect()
                torch.cuda.empty_cache()
                K.TUNE["log"].clear()
                try:
                    r = measure(a, name, st, wu, None, rank, world, dev, False, ranks_seen)
                except Exception as exc:          # the headline must survive a failure here; the failure is reported, not hidden
                    others[name] = {"error": f"{type(exc).__name__}: {exc}"[:300]}
                    print(f"bench.py: other_configs[{name}] FAILED: {type(exc).__name__}: {exc}", file=sys.stderr)
                    continue
                rf = r["roofline"]
                others[name] = {"workload": r["config"]["workload"], "value": r["value"], "unit": r["unit"], "steps": st, "warmup": wu,
                                "ms_per_step": r["ms_per_step"], "gflop_per_image": r["config"]["gflop_per_image"],
                                "whole_step_frac": rf["whole_step_frac"], "hbm": rf["hbm"],
                                "roofline": {"kernel": rf["kernel"], "achieved": rf["achieved"], "frac": rf["frac"], "unit": rf["unit"],
                                             "launches_per_step": rf["launches_per_step"], "avg_launch_us": rf["avg_launch_us"],
                                             "families": rf["families"]},
                                "gemm_table": r.get("gemm_table")}
                if "final_loss" in r:
                    others[name]["final_loss"] = r["final_loss"]
                if name == "c4":
                    # C4 is DEFINED on 8 GPUs, and with N > 1 the update cannot ride in the weight-gradient epilogues (the exchange sits
                    # between a gradient and its update): the per-GPU step a scaling efficiency should be read against is the
                    # one with separate update launches (MRG_FUSE_OPT=0), printed beside the fused one
                    from mirage_amd import trainer as _T
                    was = _T.FUSE_OPT
                    _T.FUSE_OPT = False
                    try:
                        gc.collect(); torch.cuda.empty_cache()
                        r2 = measure(a, name, st, wu, None, rank, world, dev, False, ranks_seen)
                        others[name]["unfused_update"] = {"ms_per_step": r2["ms_per_step"], "value": r2["value"],
                                                          "note": "MRG_FUSE_OPT=0: the per-GPU step of an N > 1 run before communication"}
                    except Exception as exc:
                        others[name]["unfused_update"] = {"error": f"{type(exc).__name__}: {exc}"[:300]}
                    finally:
                        _T.FUSE_OPT = was
            result["other_configs"] = others
            gc.collect()
            torch.cuda.empty_cache()
        if world == 1 and not a.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(model_name, doms, Kk, train)
        if os.environ.get("MRG_TUNE_SAVE") == "1":
            K.save_tune_cache()
        _emit(real_stdout, json.dumps(result))
    if world > 1 or force_dp:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
