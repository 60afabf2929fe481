"""Helper process for tests/test_fullsize_gpu.py::test_two_engine_ranks_on_one_gpu (not a test module).

    python tests/_dp_engine_worker.py <rank> <world> <port> <outfile> [comm] [wire] [update]

world > 1: one data-parallel rank of the HIP engine (all ranks share GPU 0, gloo backend) running trainer.TrainStep's staged
schedule -- upper / middle / lower backward graphs with the gradient exchange between them and the optimizer graph -- on its
shard of a fixed global batch.  world == 1: the same model in one process on the whole batch (single graph, no reducer).
Writes {flat_p, flat_g, losses, grad_norm} with torch.save.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch
import torch.distributed as dist

GLOBAL_B, K, STEPS = 8, 32, 2


def config():
    from oracle import mirage_oracle as O
    return O.make_config(model="tiny", in_domains=("bscan", "slo"), input_size=64, patch_size=8, decoder_dim=64, decoder_depth=2,
                         decoder_heads=2, dim=128, depth=4, heads=2)


def main():
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    if len(sys.argv) > 5:
        os.environ["MRG_DP_COMM"] = sys.argv[5]
    if len(sys.argv) > 6:
        os.environ["MRG_DP_DTYPE"] = sys.argv[6]
    if len(sys.argv) > 7:
        os.environ["MRG_DP_UPDATE"] = sys.argv[7]
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    if world > 1:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        import datetime
        # a peer that died during start-up must surface within minutes, not after gloo's default half hour
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=240))
    from mirage_amd import builder, optim
    from mirage_amd.dp import GradReducer, broadcast_params
    from mirage_amd.trainer import TrainStep
    from oracle import mirage_oracle as O
    import test_model_gpu as T
    cfg = config()
    model, args, _ = T._build(cfg)
    model.train()
    loss_fns = builder.get_losses(args)
    opt = optim.FusedAdamW(model, [1e-3, 7e-4, 5e-4, 3e-4], [0.05] * 4, betas=(0.9, 0.95), eps=1e-8)
    reducer = None
    if world > 1:
        broadcast_params(opt.arena.flat_p)
        opt.arena.refresh_shadow(force=True)
        reducer = GradReducer(opt.arena.flat_g, bucket_bytes=1 << 20)
    per = GLOBAL_B // world
    sl = slice(rank * per, (rank + 1) * per)
    batches, noises = [], []
    for i in range(STEPS):
        x = O.synthetic_batch(cfg, GLOBAL_B, seed=300 + i)
        batches.append({d: t[sl].contiguous().to(dev) for d, t in x.items()})
        (p, u, g), _, _, _ = O.synthetic_masks(cfg, GLOBAL_B, K, seed=400 + i)
        noises.append((torch.from_numpy(p)[sl].contiguous().to(dev), torch.cat([torch.from_numpy(a) for a in u], 1)[sl].contiguous().to(dev),
                       torch.from_numpy(g)[sl].contiguous().to(dev)))
    ts = TrainStep(model, loss_fns, opt, batches[0], K, use_graph=True, reducer=reducer, feed_mask_noise=True)
    ts.load(batches[0])
    for dst, src in zip(ts.static_noise, noises[0]):
        dst.copy_(src)
    ts.warmup_and_capture(warm_iters=1)
    n_graphs = sum(g is not None for g in (ts.g_fb, ts.g_b, ts.g_opt)) + len(ts.g_mids)
    losses = []
    for i in range(STEPS):
        ts.step(batches[i], noise=noises[i])
        losses.append(float(ts.loss))
    torch.cuda.synchronize()
    if reducer is not None and reducer.update == "sharded":
        reducer.gather_moments(opt.exp_avg, opt.exp_avg_sq)          # what a checkpoint does first (run_pretraining.main)
        torch.cuda.synchronize()
    torch.save({"flat_p": opt.arena.flat_p.cpu(), "flat_g": opt.arena.flat_g.cpu(), "losses": losses, "grad_norm": float(opt.grad_norm),
                "exp_avg": opt.exp_avg.cpu(), "exp_avg_sq": opt.exp_avg_sq.cpu(), "shadow": opt.arena.flat_bf16.cpu(),
                "update": reducer.update if reducer is not None else "single",
                "n_graphs": n_graphs, "cuts": list(ts.cuts), "steps": opt.step_dev.tolist()}, out)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
