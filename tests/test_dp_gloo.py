"""Data-parallel layer on CPU with the gloo backend, world_size 2 (the GPU path uses the same code with RCCL):
bucketed gradient averaging, initial broadcast, and equivalence of 2-rank data parallelism with one process on the
concatenated batch (gradients produced by the CPU oracle)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from mirage_amd.dp import GradReducer, broadcast_params, all_reduce_scalar_mean
    from oracle import mirage_oracle as O
    torch.set_num_threads(2)
    # 1) plain averaging over several buckets
    n = 10_000
    flat = torch.arange(n, dtype=torch.float32) * (rank + 1)
    GradReducer(flat, bucket_bytes=4096).reduce()
    want = torch.arange(n, dtype=torch.float32) * (sum(range(1, world + 1)) / world)
    ok1 = torch.allclose(flat, want)
    # 1b) the same average as two partial reductions over disjoint range lists (the overlapped schedule of trainer.TrainStep)
    flat2 = torch.arange(n, dtype=torch.float32) * (rank + 1)
    r = GradReducer(flat2, bucket_bytes=4096)
    first, rest = [(2048, 4096), (8192, n)], [(0, 2048), (4096, 8192)]
    r.reduce_async(first)
    mid_ok = torch.allclose(flat2[2048:4096], want[2048:4096]) and torch.equal(flat2[:2048], torch.arange(2048, dtype=torch.float32) * (rank + 1))
    r.reduce_async(rest); r.wait()
    ok1 = ok1 and mid_ok and torch.allclose(flat2, want)
    # 1c) the alternative exchange formats (MRG_DP_COMM / MRG_DP_DTYPE): explicit reduce-scatter + all-gather, bf16 on the wire
    for comm, wire, tol in (("rs_ag", "fp32", 0.0), ("allreduce", "bf16", 8e-3), ("rs_ag", "bf16", 8e-3)):
        os.environ["MRG_DP_COMM"], os.environ["MRG_DP_DTYPE"] = comm, wire
        flat3 = (torch.arange(n, dtype=torch.float32) % 97 + 1.0) * (rank + 1)
        want3 = (torch.arange(n, dtype=torch.float32) % 97 + 1.0) * (sum(range(1, world + 1)) / world)
        r3 = GradReducer(flat3, bucket_bytes=4096)
        r3.reduce_async([(0, 4096), (4096, n)]); r3.wait()
        ok1 = ok1 and bool(((flat3 - want3).abs() <= tol * want3.abs() + 1e-6).all())
    os.environ.pop("MRG_DP_COMM"); os.environ.pop("MRG_DP_DTYPE")
    # 1d) MRG_DP_UPDATE=sharded: reduce-scatter -> AdamW on this rank's shard -> all-gather of the parameters, against the replicated
    #     update (all-reduce, then AdamW of everything on every rank).  A torch AdamW over element ranges stands in for mrg_adamw (the
    #     GPU path hands FusedAdamW.step_ranges to the same call); ragged ranges exercise the tails every rank updates in full.
    os.environ["MRG_DP_UPDATE"] = "sharded"
    nn = 10_000 + 37 * 64
    gen = torch.Generator().manual_seed(11)
    p0 = torch.randn(nn, generator=gen)
    g_r = [torch.randn(nn, generator=gen) for _ in range(world)]

    def adamw_ranges(p, g, m, v, ranges, lr=1e-2, wd=0.05, b1=0.9, b2=0.95, eps=1e-8, t=1):
        ss = 0.0
        for s0, e0 in ranges:
            gg = g[s0:e0]
            p[s0:e0].mul_(1 - lr * wd)
            m[s0:e0].mul_(b1).add_(gg, alpha=1 - b1)
            v[s0:e0].mul_(b2).addcmul_(gg, gg, value=1 - b2)
            p[s0:e0].addcdiv_(m[s0:e0] / (1 - b1 ** t), (v[s0:e0] / (1 - b2 ** t)).sqrt() + eps, value=-lr)
            ss += float((gg.double() ** 2).sum())
        return ss

    ranges4 = [(0, 4096), (4096 + 64, 9000 // 64 * 64), (9984, nn)]              # 64-aligned, lengths not multiples of 64 * world
    covered = torch.zeros(nn, dtype=torch.bool)
    for s0, e0 in ranges4:
        covered[s0:e0] = True
    g_avg = sum(g_r) / world
    p_ref, m_ref, v_ref = p0.clone(), torch.zeros(nn), torch.zeros(nn)
    ss_ref = adamw_ranges(p_ref, g_avg, m_ref, v_ref, ranges4)
    p_sh, m_sh, v_sh, g_sh = p0.clone(), torch.zeros(nn), torch.zeros(nn), g_r[rank].clone()
    r4 = GradReducer(g_sh, bucket_bytes=4096)
    counted = [0.0]

    def upd(owned, tails, count_tails):
        counted[0] += adamw_ranges(p_sh, g_sh, m_sh, v_sh, owned)
        t_ss = adamw_ranges(p_sh, g_sh, m_sh, v_sh, tails)
        if count_tails:
            counted[0] += t_ss

    recast_calls = []
    r4.exchange_update_async(ranges4, p_sh, upd, lambda s0, e0: recast_calls.append((s0, e0)))
    r4.wait()
    gn = torch.tensor([counted[0] ** 0.5])
    r4.finish_norm(gn)
    r4.gather_moments(m_sh, v_sh)
    plan_ok = all(sub % 64 == 0 and (me - s0) == sub * world and e0 - me < 64 * world for (s0, e0), (me, sub) in r4._plan.items())
    ok_sh = (plan_ok and len(recast_calls) > 0 and torch.equal(p_sh, p_ref) and torch.equal(m_sh[covered], m_ref[covered])
             and torch.equal(v_sh[covered], v_ref[covered]) and abs(float(gn[0]) - ss_ref ** 0.5) < 1e-4 * ss_ref ** 0.5
             and torch.equal(p_sh[~covered], p0[~covered]))
    # replicas bit-identical: compare against rank 0's parameters
    p_chk = p_sh.clone()
    dist.broadcast(p_chk, src=0)
    ok_sh = ok_sh and torch.equal(p_chk, p_sh)
    ok1 = ok1 and ok_sh
    os.environ.pop("MRG_DP_UPDATE")
    # 2) broadcast
    p = torch.full((100,), float(rank))
    broadcast_params(p, src=0)
    ok2 = bool((p == 0).all())
    # 3) DP == single process on the concatenated batch (equal shards; per-sample-mean losses)
    cfg = O.make_config(model="tiny", in_domains=("bscan", "slo"), input_size=64, patch_size=8, decoder_dim=64, decoder_depth=1,
                        decoder_heads=2, dim=128, depth=1, heads=2)
    shapes = O.param_shapes(cfg)
    B, K = 4, 16
    x = O.synthetic_batch(cfg, B, seed=3)
    _, _, ids_keep, ids_restore = O.synthetic_masks(cfg, B, K, seed=4)
    names = [k for k, (s, t) in shapes.items() if t]

    def flat_grads(sl):
        sd = {k: v.clone().requires_grad_(shapes[k][1]) for k, v in O.procedural_state_dict(cfg).items()}
        loss, _, _, _ = O.forward_loss(cfg, sd, {d: t[sl] for d, t in x.items()}, ids_keep[sl], ids_restore[sl])
        loss.backward()
        return torch.cat([sd[k].grad.reshape(-1) for k in names]), float(loss)

    per = B // world
    g_local, l_local = flat_grads(slice(rank * per, (rank + 1) * per))
    GradReducer(g_local, bucket_bytes=1 << 16).reduce()
    l_mean = float(all_reduce_scalar_mean(torch.tensor([l_local]))[0])
    g_full, l_full = flat_grads(slice(0, B))
    err = float((g_local - g_full).norm() / g_full.norm())
    out.put((rank, ok1, ok2, err, abs(l_mean - l_full)))
    dist.destroy_process_group()


def test_two_rank_gloo_data_parallel():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok1, ok2, err, dl in res:
        assert ok1 and ok2
        assert err < 1e-5, err
        assert dl < 1e-5, dl
