"""World-size-1 check of the overlapped data-parallel schedule (backward cut at a quarter and at the middle of the encoder, ranged
all-reduces and the stages' AdamW on the side stream between the graph segments): it must produce bit-identical losses and weights to the single-graph step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29555")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch, torch.distributed as dist
from mirage_amd import builder, optim
from mirage_amd.dp import GradReducer
from mirage_amd.trainer import TrainStep
from oracle import mirage_oracle as O

dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
use_graph = "--eager" not in sys.argv
# --soak N: N optimizer steps instead of 3 (replay safety of everything the multi-rank schedule keeps on the host: the partial-sum
# slots of the fused gradient norm, the reducer's events), and the logged gradient norm is compared at every step as well
N_STEPS = int(sys.argv[sys.argv.index("--soak") + 1]) if "--soak" in sys.argv else 3
cfg = O.make_config(model="tiny", in_domains=("bscan", "slo", "bscanlayermap"), input_size=64, patch_size=8, layermap_size=16, layermap_patch=2,
                    decoder_dim=64, decoder_depth=2, decoder_heads=2, dim=128, depth=4, heads=2)
dom = cfg["domains"]


def run(split):
    args = builder.make_args(cfg["in_domains"], cfg["out_domains"], input_size=64, patch_size=8, layermap_size=16, layermap_patch=2,
                             decoder_dim=64, decoder_depth=2, decoder_num_heads=2)
    model = builder.get_model(args, dim_tokens=128, depth=4, num_heads=2)
    model.load_state_dict(O.procedural_state_dict(cfg), strict=True)
    model = model.to(dev).train()
    opt = optim.FusedAdamW(model, [1e-3] * (N_STEPS + 8), [0.05] * (N_STEPS + 8))
    red = GradReducer(opt.arena.flat_g, bucket_bytes=1 << 20, force=True) if split else None
    batches = [{d: t.to(dev) for d, t in O.synthetic_batch(cfg, 4, seed=40 + i).items()} for i in range(3)]
    ts = TrainStep(model, builder.get_losses(args), opt, batches[0], 24, use_graph=use_graph, reducer=red)
    if split:
        assert ts.split == 2 and ts.cuts == [1, 2] and all(ts.stage_ranges) and ts.r_rest
        n1 = sum(e - s for rs in ts.stage_ranges for s, e in rs); n2 = sum(e - s for s, e in ts.r_rest)
        assert n1 + n2 == opt.arena.flat_g.numel(), (n1, n2, opt.arena.flat_g.numel())       # the stages partition the arena
    ts.load(batches[0])
    torch.manual_seed(7); torch.cuda.manual_seed(7)
    ts.warmup_and_capture(warm_iters=1)
    losses, norms = [], []
    for i in range(N_STEPS):
        ts.step(batches[i % len(batches)]); losses.append(float(ts.loss)); norms.append(float(opt.grad_norm))
    torch.cuda.synchronize()
    # the norm the reference logs (mutils/native_scaler.py:46-61) recomputed from the last step's gradient arena
    direct = float(opt.arena.flat_g[:opt.arena.n_trainable].double().norm())
    assert abs(norms[-1] - direct) <= 1e-5 * direct, (norms[-1], direct)
    return losses, norms, opt.arena.flat_p.detach().clone()


l0, n0, p0 = run(False)
l1, n1, p1 = run(True)
print("losses", l0[:3], l1[:3], "steps", N_STEPS)
assert l0 == l1, (l0, l1)
assert torch.equal(p0, p1), float((p0 - p1).abs().max())
# the two schedules partition the arena differently (per block vs per stage), so the partial sums are added in another order
worst = max(abs(a - b) / b for a, b in zip(n1, n0))
print("grad_norm worst rel", worst)
assert worst <= 1e-5, worst
dist.destroy_process_group()
print("SPLIT-OK graph" if use_graph else "SPLIT-OK eager")
