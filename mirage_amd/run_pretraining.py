#!/usr/bin/env python3
"""Pretraining driver: the counterpart of the reference's run_pretraining.py (get_args :59-350, get_model :373-448,
main :451-566, train_one_epoch :648-767) on the MI355X engine.

Same flag names / defaults for everything the BASELINE configs touch; same step protocol (per-step cosine lr / wd arrays,
lr = blr * total_batch / 256, decay / no-decay AdamW groups with betas (0.9, 0.95), loss = sum of masked task losses,
global grad-norm, clip / skip), same artefacts (`config.json`, per-epoch JSON lines in `log.txt` with the reference's
keys, `checkpoint-{epoch}.pth` = {model, optimizer, epoch, scaler, args}).  Differences: bf16 without a GradScaler
(`loss_scale` is logged as 1.0), device-side mask RNG, no `.item()` inside the step (metrics are fetched once per
`print_freq` steps), data-parallel over the ranks of one node (python -m torch.distributed.run ...).

    python -m mirage_amd.run_pretraining --data_path synthetic --in_domains bscan-slo --out_domains bscan-slo \\
        --batch_size 64 --perc_encoded_tokens 0.25 --epochs 1 --steps_per_epoch 50
"""
import argparse
import glob
import json
import math
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist
import yaml

from . import builder, optim
from .dp import GradReducer, all_reduce_scalar_mean, broadcast_params
from .trainer import TrainStep


def get_args(argv=None):
    config_parser = argparse.ArgumentParser(description="Training Config", add_help=False)
    config_parser.add_argument("-c", "--config", default="", type=str, metavar="FILE")
    p = argparse.ArgumentParser("MIRAGE pre-training (MI355X engine)", add_help=True)
    p.add_argument("--batch_size", default=256, type=int, help="Batch size per GPU")
    p.add_argument("--epochs", default=1600, type=int)
    p.add_argument("--save_ckpt_freq", default=20, type=int)
    p.add_argument("--in_domains", default="bscan-slo-bscanlayermap", type=str)
    p.add_argument("--out_domains", default="bscan-slo-bscanlayermap", type=str)
    p.add_argument("--model", default="miragepre_base", type=str)
    p.add_argument("--num_encoded_tokens", default=98, type=int)
    p.add_argument("--perc_encoded_tokens", default=None, type=float)
    p.add_argument("--num_global_tokens", default=1, type=int)
    p.add_argument("--patch_size", default=32, type=int)
    p.add_argument("--input_size", default=512, type=int)
    p.add_argument("--alphas", type=float, default=1.0)
    p.add_argument("--sample_tasks_uniformly", default=False, action="store_true")
    p.add_argument("--decoder_dim", default=256, type=int)
    p.add_argument("--decoder_depth", default=2, type=int)
    p.add_argument("--decoder_num_heads", default=8, type=int)
    p.add_argument("--drop_path", type=float, default=0.0)
    p.add_argument("--loss_on_unmasked", default=False, action="store_true")
    p.add_argument("--no_loss_on_unmasked", action="store_false", dest="loss_on_unmasked")
    p.add_argument("--opt", default="adamw", type=str)
    p.add_argument("--opt_eps", default=1e-8, type=float)
    p.add_argument("--opt_betas", default=[0.9, 0.95], type=float, nargs="+")
    p.add_argument("--clip_grad", type=float, default=None)
    p.add_argument("--skip_grad", type=float, default=None)
    p.add_argument("--weight_decay", type=float, default=0.05)
    p.add_argument("--weight_decay_end", type=float, default=None)
    p.add_argument("--blr", type=float, default=1e-4)
    p.add_argument("--min_lr", type=float, default=0.0)
    p.add_argument("--warmup_epochs", type=int, default=40)
    p.add_argument("--warmup_steps", type=int, default=-1)
    p.add_argument("--base_output_dir", default="./__output/pre", type=str)
    p.add_argument("--device", default="cuda")
    p.add_argument("--seed", default=0, type=int)
    p.add_argument("--resume", default="")
    p.add_argument("--auto_resume", action="store_true", default=True)
    p.add_argument("--no_auto_resume", action="store_false", dest="auto_resume")
    p.add_argument("--start_epoch", default=0, type=int)
    p.add_argument("--weights", default="", type=str, help="optional initial weights (.pth with a 'model' entry)")
    p.add_argument("--hflip", type=float, default=0.5, help="probability of a horizontal flip (run_pretraining.py:224)")
    p.add_argument("--intensity_shift", type=float, default=0.1, help="sigma of the random intensity shift (run_pretraining.py:228)")
    p.add_argument("--affine", action="store_true", help="random affine transformations (run_pretraining.py:240-244)")
    p.add_argument("--no_affine", action="store_false", dest="affine")
    p.set_defaults(affine=True)
    p.add_argument("--num_workers", default=10, type=int, help="reader threads of the asynchronous .npy feed (run_pretraining.py:256)")
    p.add_argument("--no_augment", action="store_true", help="feed the .npy samples unchanged (no flip / shift / affine)")
    p.add_argument("--data_path", required=True, type=str, help="'synthetic' or a folder with <domain>/*.npy")
    p.add_argument("--steps_per_epoch", default=None, type=int, help="synthetic data only")
    p.add_argument("--no_graph", action="store_true", help="run the step eagerly instead of replaying a hipGraph")
    p.add_argument("--no_debug_images", action="store_true", help="skip the picture of the first step of every epoch (<output_dir>/debug/)")
    p.add_argument("--print_freq", default=10, type=int)
    a_cfg, remaining = config_parser.parse_known_args(argv)
    if a_cfg.config:
        with open(a_cfg.config, "r") as f:
            p.set_defaults(**yaml.safe_load(f))
    args = p.parse_args(remaining)
    domains = args.in_domains.split("-")
    if isinstance(args.patch_size, int):
        args.patch_size = {d: ((8, 8) if d == "bscanlayermap" else (args.patch_size, args.patch_size)) for d in domains}
    if isinstance(args.input_size, int):
        args.input_size = {d: ((128, 128) if d == "bscanlayermap" else (args.input_size, args.input_size)) for d in domains}
    args.patch_size = {d: tuple(v) for d, v in args.patch_size.items()}
    args.input_size = {d: tuple(v) for d, v in args.input_size.items()}
    args.grid_sizes = {d: [args.input_size[d][i] // args.patch_size[d][i] for i in range(2)] for d in args.input_size}
    if args.perc_encoded_tokens is not None:
        total = sum(g[0] * g[1] for g in args.grid_sizes.values())
        args.num_encoded_tokens = int(total * args.perc_encoded_tokens)
    return args


class SyntheticData:
    """Device-resident synthetic batches: images U[0,1) fp32, layer maps integers 0..12."""

    def __init__(self, args, device, steps, seed):
        rng = np.random.default_rng(seed)
        self.steps = steps
        self.pool = []
        for _ in range(4):
            b = {}
            for d in args.all_domains:
                h, w = args.input_size[d]
                if d == "bscanlayermap":
                    b[d] = torch.from_numpy(rng.integers(0, 13, (args.batch_size, h, w), dtype=np.int64)).to(device)
                else:
                    b[d] = torch.from_numpy(rng.random((args.batch_size, 1, h, w), dtype=np.float32)).to(device)
            self.pool.append(b)

    def __len__(self):
        return self.steps

    def __iter__(self):
        for i in range(self.steps):
            yield self.pool[i % len(self.pool)]


def npy_folder_loader(args, device, rank, world, seed):
    """<root>/<domain>/*.npy, uint8 images / integer layer maps (mutils/datasets_pretrain.py:86-207, dataset_folder.py:32-51) through
    the asynchronous feed of mirage_amd.data.PrefetchingNpyLoader: reader threads -> pinned uint8 staging -> H2D + GPU augmentation
    (mirage_amd.data.DataAugmentationForMIRAGE, the reference's CPU transform of the same name, datasets_pretrain.py:18-83) on a
    copy stream, double-buffered against the training step (run_pretraining.py:478-487: shuffle, drop_last, prefetch 2, pinned)."""
    from .data import DataAugmentationForMIRAGE, PrefetchingNpyLoader
    augment = None if getattr(args, "no_augment", False) else DataAugmentationForMIRAGE(args, seed=seed * 1000 + rank)
    files = {d: sorted(glob.glob(os.path.join(args.data_path, d, "*.npy"))) for d in args.all_domains}
    assert min(len(v) for v in files.values()) > 0, f"no .npy files under {args.data_path}/<domain>/"
    return PrefetchingNpyLoader(files, args.batch_size, device, augment=augment, rank=rank, world=world, seed=seed,
                                num_workers=getattr(args, "num_workers", 8) or 8, prefetch=2)


def debug_image(preds, masks, targets, patch_sizes):
    """The picture `save_debug_images` of the reference composes at the first step of every epoch (run_pretraining.py:568-645): per task the
    input on top of the prediction with the UNMASKED patches replaced by the input (mask: 1 = predicted), both resampled to 128 x 128 (nearest),
    the tasks stacked vertically with an 8-pixel white line between them.  Layer maps: argmax of the class scores, both scaled by their maximum.
    Returns [B, C, rows, 128] fp32 on the CPU (or None without tasks).  Logging only: plain torch on the host side, off the step."""
    import torch.nn.functional as F
    out = None
    for task, pred in preds.items():
        if task not in targets or task not in masks:
            continue
        p, x, m = pred.detach().float().cpu(), targets[task].detach().cpu(), masks[task].detach().cpu()
        if x.dim() == 3:                                   # layer maps: class scores -> labels, both to 0..1 for the picture (:597-605)
            p = F.softmax(p, dim=1).argmax(dim=1, keepdim=True).float()
            x = x.unsqueeze(1).float()
            p = p / p.max().clamp_min(1e-12)
            x = x / x.max().clamp_min(1e-12)
        else:
            x = x.float()
        P = patch_sizes[task] if isinstance(patch_sizes, dict) else patch_sizes
        ph, pw = (P, P) if isinstance(P, int) else (P[0], P[1])
        c, h, w = x.shape[1], x.shape[2] // ph, x.shape[3] // pw
        mask = m.reshape(m.shape[0], c, h, w).float()      # 'b (c h w) -> b c h w' (:617)
        mask = F.interpolate(mask, size=tuple(x.shape[2:]), mode="nearest")
        p = p * mask + x * (1.0 - mask)
        pair = torch.cat([F.interpolate(x, (128, 128), mode="nearest"), F.interpolate(p, (128, 128), mode="nearest")], dim=2)
        out = pair if out is None else torch.cat([out, torch.ones(pair.shape[0], pair.shape[1], 8, pair.shape[3]), pair], dim=2)
    return out


def save_debug_images(preds, masks, targets, epoch, patch_sizes, output_dir):
    """<output_dir>/debug/<epoch>_<task-task-...>.jpg, laid out like torchvision.utils.save_image's default grid (8 per row, 2 pixels of
    padding, values clamped to [0, 1]) -- written with PIL, torchvision is not a dependency of this package."""
    img = debug_image(preds, masks, targets, patch_sizes)
    if img is None:
        return None
    from PIL import Image
    B, C, H, W = img.shape
    nrow = min(8, B)
    ncol = (B + nrow - 1) // nrow
    pad = 2
    grid = torch.zeros(3, ncol * (H + pad) + pad, nrow * (W + pad) + pad)
    for i in range(B):
        r, c = divmod(i, nrow)
        tile = img[i].expand(3, H, W) if C == 1 else img[i][:3]
        grid[:, pad + r * (H + pad): pad + r * (H + pad) + H, pad + c * (W + pad): pad + c * (W + pad) + W] = tile
    arr = (grid.clamp(0, 1) * 255 + 0.5).to(torch.uint8).permute(1, 2, 0).numpy()
    path = Path(output_dir) / "debug" / f"{epoch}_{'-'.join(t for t in preds if t in targets and t in masks)}.jpg"
    path.parent.mkdir(parents=True, exist_ok=True)
    Image.fromarray(arr).save(path)
    return path


def save_model(args, epoch, model, optimizer):
    """checkpoint-{epoch}.pth = {model, optimizer, epoch, scaler, args} (mutils/checkpoint.py:9-32)."""
    path = Path(args.output_dir) / f"checkpoint-{epoch}.pth"
    torch.save({"model": {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}, "optimizer": optimizer.state_dict(),
                "epoch": epoch, "scaler": {}, "args": args}, path)
    return path


def auto_load_model(args, model, optimizer):
    """Resume from the highest-numbered checkpoint (mutils/checkpoint.py:35-72)."""
    if args.auto_resume and not args.resume:
        latest = -1
        for ck in glob.glob(os.path.join(args.output_dir, "checkpoint-*.pth")):
            t = ck.split("-")[-1].split(".")[0]
            if t.isdigit():
                latest = max(latest, int(t))
        if latest >= 0:
            args.resume = os.path.join(args.output_dir, f"checkpoint-{latest}.pth")
    if args.resume:
        ck = torch.load(args.resume, map_location="cpu", weights_only=False)
        model.load_state_dict(ck["model"])
        if "optimizer" in ck and "epoch" in ck:
            optimizer.load_state_dict(ck["optimizer"])
            args.start_epoch = ck["epoch"] + 1
        print(f"Resumed from {args.resume} (start_epoch {args.start_epoch})")


def main(args):
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise RuntimeError("mirage_amd.run_pretraining needs an MI355X (HIP kernels); there is no CPU path")
    if os.environ.get("MRG_SHARE_GPU") == "1":        # test hook: several ranks on one GPU (with MRG_DIST_BACKEND=gloo), as in bench.py
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        backend = os.environ.get("MRG_DIST_BACKEND", "nccl")       # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
    torch.manual_seed(args.seed)                      # weight init: common to all ranks (and broadcast from rank 0 below)
    np.random.seed(args.seed)
    args.in_domains = args.in_domains.split("-")
    args.out_domains = args.out_domains.split("-")
    args.all_domains = list(dict.fromkeys(args.in_domains + args.out_domains))
    args.decoder_use_task_queries, args.decoder_use_xattn = True, True
    model = builder.get_model(args, args.model)
    args.output_dir = os.path.join(args.base_output_dir, args.model)
    if rank == 0:
        (Path(args.output_dir) / "debug").mkdir(exist_ok=True, parents=True)
        with open(Path(args.output_dir, "config.json"), "w") as f:
            json.dump({k: v for k, v in vars(args).items()}, f, indent=2, default=str)
    if args.weights:
        w = torch.load(args.weights, map_location="cpu", weights_only=False)["model"]
        if "_vit_large" in args.weights:
            w = {k.replace("blocks.", "encoder."): v for k, v in w.items()}
        else:
            w = {k: v for k, v in w.items() if "input_adapter" not in k and "output_adapter" not in k}
        print(model.load_state_dict(w, strict=False))
    model.to(device).train()
    # masks / stochastic depth: one RNG stream per rank, so that a global batch sees world x B mask draws
    torch.manual_seed(args.seed + 1 + rank)
    torch.cuda.manual_seed(args.seed + 1 + rank)
    loss_fns = builder.get_losses(args)

    if args.data_path == "synthetic":
        data = SyntheticData(args, device, args.steps_per_epoch or 100, 1234 + rank)
    else:
        data = npy_folder_loader(args, device, rank, world, args.seed)
    steps_per_epoch = len(data)
    total_batch = args.batch_size * world
    args.lr = args.blr * total_batch / 256
    if args.warmup_epochs > args.epochs:
        # the reference's scheduler (mutils/native_scaler.py:64-88) fails its length assertion here; say why
        raise ValueError(f"--warmup_epochs ({args.warmup_epochs}) exceeds --epochs ({args.epochs}): the per-iteration schedule would be "
                         f"longer than the run")
    lrs = optim.cosine_scheduler(args.lr, args.min_lr, args.epochs, steps_per_epoch, warmup_epochs=args.warmup_epochs,
                                 warmup_steps=args.warmup_steps)
    wd_end = args.weight_decay if args.weight_decay_end is None else args.weight_decay_end
    wds = optim.cosine_scheduler(args.weight_decay, wd_end, args.epochs, steps_per_epoch)
    optimizer = optim.FusedAdamW(model, lrs, wds, betas=tuple(args.opt_betas), eps=args.opt_eps, clip_grad=args.clip_grad,
                                 skip_grad=args.skip_grad)
    auto_load_model(args, model, optimizer)
    broadcast_params(optimizer.arena.flat_p)
    optimizer.arena.refresh_shadow(force=True)
    if args.start_epoch:
        optimizer.set_iteration(args.start_epoch * steps_per_epoch)
    reducer = GradReducer(optimizer.arena.flat_g) if world > 1 else None
    n_parameters = sum(p.numel() for p in model.parameters() if p.requires_grad)
    if rank == 0:
        print(f"Number of params: {n_parameters / 1e6} M\nLR = {args.lr:.8f}\nBatch size = {total_batch}\n"
              f"Number of training steps = {steps_per_epoch}")

    step = None
    start = time.time()
    for epoch in range(args.start_epoch, args.epochs):
        sums, n_log, t_ep = {}, 0, time.time()
        for it, batch in enumerate(data):
            if step is None:
                step = TrainStep(model, loss_fns, optimizer, batch, args.num_encoded_tokens, alphas=args.alphas,
                                 sample_tasks_uniformly=args.sample_tasks_uniformly, use_graph=not args.no_graph, reducer=reducer,
                                 loss_on_unmasked=args.loss_on_unmasked)
                step.load(batch)
                step.warmup_and_capture(warm_iters=1)
                optimizer.set_iteration(epoch * steps_per_epoch)
                if rank == 0:
                    # which GEMM variant table this run is on: shapes served from the committed table vs tuned during the warm-up step
                    # (a per-GPU batch the table was not measured at falls back to first-use autotuning)
                    from . import kernels as _K
                    tuned = [f"{k[2]}x{k[3]}x{k[4]}" for k, _, _ in _K.TUNE["log"]]
                    print(f"GEMM table {os.path.basename(_K._TUNE_FILE)}: {_K.TUNE['table_entries']} entries, {len(_K.TUNE['table_used'])} shapes of this run "
                          f"served from it, {len(tuned)} tuned at warm-up" + (f" ({', '.join(tuned[:12])}{' ...' if len(tuned) > 12 else ''})" if tuned else ""),
                          flush=True)
            step.step(batch)
            if it == 0 and rank == 0 and args.output_dir and not args.no_debug_images:
                # the reference's picture of the first step of every epoch (run_pretraining.py:709-712): one host sync per epoch
                path = save_debug_images(step.pred_images(), step.masks, batch, epoch, args.patch_size, args.output_dir)
                print(f"Saving debug images: {path}", flush=True)
            if it % args.print_freq == 0 or it == steps_per_epoch - 1:
                vals = {"loss": all_reduce_scalar_mean(step.loss.detach().clone().reshape(1))}
                for t, l in step.task_losses.items():
                    vals[f"{t}_loss"] = all_reduce_scalar_mean(l.detach().clone().reshape(1))
                vals["grad_norm"] = optimizer.grad_norm.clone()
                vals = {k: float(v) for k, v in vals.items()}           # the only host sync, once per print_freq steps
                if not math.isfinite(vals["loss"]):
                    print(f"Loss is {vals['loss']}, stopping training")
                    sys.exit(1)
                for k, v in vals.items():
                    sums[k] = sums.get(k, 0.0) + v
                n_log += 1
                if rank == 0:
                    gi = min(epoch * steps_per_epoch + it, len(lrs) - 1)
                    ips = total_batch * (it + 1) / max(time.time() - t_ep, 1e-9)
                    print(f"Epoch: [{epoch}] [{it}/{steps_per_epoch}] " + " ".join(f"{k}: {v:.4f}" for k, v in vals.items())
                          + f" lr: {lrs[gi]:.6f} images/sec: {ips:.1f}", flush=True)
        gi = min((epoch + 1) * steps_per_epoch - 1, len(lrs) - 1)
        stats = {f"[Epoch] {k}": v / max(n_log, 1) for k, v in sums.items()}
        stats.update({"[Epoch] lr": float(lrs[gi]), "[Epoch] min_lr": float(lrs[gi]), "[Epoch] weight_decay": float(wds[gi]),
                      "[Epoch] loss_scale": 1.0, "epoch": epoch, "n_parameters": n_parameters})
        saving = (epoch + 1) % args.save_ckpt_freq == 0 or epoch + 1 == args.epochs
        if saving and reducer is not None and reducer.update == "sharded":
            reducer.gather_moments(optimizer.exp_avg, optimizer.exp_avg_sq)      # collective: the moments of a shard live on its owner
        if rank == 0:
            if saving:
                save_model(args, epoch, model, optimizer)
            with open(os.path.join(args.output_dir, "log.txt"), mode="a", encoding="utf-8") as f:
                f.write(json.dumps(stats) + "\n")
    if rank == 0:
        with open(Path(args.output_dir) / "training_time.txt", "w") as f:
            f.write(str(int(time.time() - start)))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main(get_args())
