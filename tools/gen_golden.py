#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE implementation (imported from /root/reference).

Runs only in the build container (the reference never travels).  Outputs small .npz fixtures under
tests/golden/ that pin the oracle (oracle/mirage_oracle.py) and, through it, the HIP path.
Nothing from the reference's source text is written to the fixtures -- only numeric inputs/outputs.

    python tools/gen_golden.py            # regenerates every fixture (about a minute on 8 cores)
"""
import json
import os
import sys

sys.dont_write_bytecode = True
REF = os.environ.get("MIRAGE_REFERENCE", "/root/reference")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)

import argparse
import contextlib
import io
import types

import numpy as np
import torch
from torch.distributions.dirichlet import Dirichlet

from oracle import mirage_oracle as O  # config + procedural weights + synthetic data (NOT the math under test)

with contextlib.redirect_stdout(io.StringIO()):
    from mirage.model import MIRAGEModel
    from mirage.input_adapters import PatchedInputAdapter, SemSegInputAdapter
    from mirage.output_adapters import SpatialOutputAdapter
    from mirage.criterion import MaskedMSELoss, MaskedCrossEntropyLoss
    from mirage.utils import build_2d_sincos_posemb
    from mutils.native_scaler import cosine_scheduler, get_grad_norm_
    from mutils.optim_factory import create_optimizer, get_parameter_groups
    from mutils.pos_embed import interpolate_pos_embed

OUT = os.path.join(ROOT, "tests", "golden")


def build_reference(cfg, drop_path_rate=0.0, drop_rate=0.0, attn_drop_rate=0.0, decoder_attn_drop_rate=0.0):
    """Adapters exactly as run_pretraining.py:380-403 builds them, model as mirage/model.py:43."""
    args = types.SimpleNamespace()
    args.input_size = {d: (c["image"], c["image"]) for d, c in cfg["domains"].items()}
    args.grid_sizes = {d: [c["grid"], c["grid"]] for d, c in cfg["domains"].items()}
    ia, oa = {}, {}
    for d in cfg["in_domains"]:
        c = cfg["domains"][d]
        if c["kind"] == "semseg":
            ia[d] = SemSegInputAdapter(num_classes=13, dim_class_emb=64, interpolate_class_emb=False, stride_level=1,
                                       patch_size_full=(c["patch"], c["patch"]), image_size=(c["image"], c["image"]),
                                       learnable_pos_emb=bool(cfg.get("learnable_pos_emb", False)))
        else:
            ia[d] = PatchedInputAdapter(num_channels=1, stride_level=1, patch_size_full=(c["patch"], c["patch"]),
                                        image_size=(c["image"], c["image"]), learnable_pos_emb=bool(cfg.get("learnable_pos_emb", False)))
    for d in cfg["out_domains"]:
        c = cfg["domains"][d]
        oa[d] = SpatialOutputAdapter(num_channels=c["channels"], stride_level=1, patch_size_full=(c["patch"], c["patch"]),
                                     dim_tokens=cfg["decoder_dim"], depth=cfg["decoder_depth"],
                                     num_heads=cfg["decoder_heads"], use_task_queries=True, task=d,
                                     context_tasks=list(cfg["in_domains"]), use_xattn=True,
                                     image_size=(c["image"], c["image"]), attn_drop_rate=decoder_attn_drop_rate)
    model = MIRAGEModel(args, input_adapters=ia, output_adapters=oa if cfg["out_domains"] else None,
                        num_global_tokens=cfg["num_global_tokens"], dim_tokens=cfg["dim"], depth=cfg["depth"],
                        num_heads=cfg["heads"], drop_path_rate=drop_path_rate, drop_rate=drop_rate, attn_drop_rate=attn_drop_rate)
    return model


def reference_losses(cfg):
    fns = {}
    for d in cfg["out_domains"]:
        c = cfg["domains"][d]
        if c["kind"] == "semseg":
            fns[d] = MaskedCrossEntropyLoss(patch_size=(c["patch"], c["patch"]), stride=1, label_smoothing=0.0)
        else:
            fns[d] = MaskedMSELoss(patch_size=(c["patch"], c["patch"]), stride=1)
    return fns


def replay_mask_noise(cfg, B, seed, alphas=1.0):
    """Draw the three random inputs of generate_random_masks in the reference's RNG order
    (mirage/model.py:207,216,225) so they can be committed next to the reference's integer outputs."""
    layout, T = O.token_layout(cfg)
    torch.manual_seed(seed)
    n = len(cfg["in_domains"])
    p = Dirichlet(torch.Tensor([alphas] * n)).sample((B,))
    u = [torch.rand(B, e - s) for (s, e) in layout.values()]
    g = torch.rand(B, T)
    return p.numpy(), [x.numpy() for x in u], g.numpy()


def subsample(t, n=4096):
    f = t.detach().reshape(-1)
    step = max(1, f.numel() // n)
    return f[::step][:n].numpy().copy(), step


def run_case(name, cfg, B, K, data_seed, mask_seed, keep_intermediate=False, full_enc=True, with_grads=True):
    print(f"== {name}: B={B} K={K}")
    with contextlib.redirect_stdout(io.StringIO()):
        model = build_reference(cfg)
    sd = O.procedural_state_dict(cfg)
    msg = model.load_state_dict(sd, strict=True)   # proves oracle.param_shapes == the reference's state_dict
    model.train()
    x = O.synthetic_batch(cfg, B, seed=data_seed)
    x_in = {d: x[d] for d in cfg["in_domains"]}
    p, u, g = replay_mask_noise(cfg, B, mask_seed)
    fx = {"p": p, "g": g}
    for d, ud in zip(cfg["in_domains"], u):
        fx[f"u.{d}"] = ud

    inter = {}
    hooks = []
    if keep_intermediate:
        for d in cfg["in_domains"]:
            hooks.append(model.input_adapters[d].register_forward_hook(
                lambda m, i, o, d=d: inter.__setitem__(f"adapter.{d}", o.detach().numpy().copy())))
        for i, blk in enumerate(model.encoder):
            hooks.append(blk.register_forward_hook(
                lambda m, i_, o, i=i: inter.__setitem__(f"block.{i}", o.detach().numpy().copy())))
    captured = {}
    if cfg["out_domains"]:
        first = cfg["out_domains"][0]
        orig = model.output_adapters[first].forward

        def spy(encoder_tokens, input_info, ids_keep, ids_restore):
            captured["enc"] = encoder_tokens
            captured["ids_keep"] = ids_keep
            captured["ids_restore"] = ids_restore
            return orig(encoder_tokens=encoder_tokens, input_info=input_info, ids_keep=ids_keep, ids_restore=ids_restore)
        model.output_adapters[first].forward = spy

    torch.manual_seed(mask_seed)
    with contextlib.redirect_stdout(io.StringIO()):
        preds, task_masks = model(x_in, num_encoded_tokens=K, alphas=1.0, sample_tasks_uniformly=False)
    for h in hooks:
        h.remove()
    enc = captured["enc"]
    fx["ids_keep"] = captured["ids_keep"].numpy()
    fx["ids_restore"] = captured["ids_restore"].numpy()
    for d in cfg["in_domains"]:
        fx[f"task_mask.{d}"] = task_masks[d].numpy()
    if full_enc:
        fx["enc_tokens"] = enc.detach().numpy()
    else:
        fx["enc_tokens_sub"], fx["enc_tokens_step"] = subsample(enc, 65536)
    fx["enc_tokens_norm"] = np.float64(enc.detach().double().norm().item())
    loss_fns = reference_losses(cfg)
    task_losses = {d: loss_fns[d](preds[d].float(), x[d], mask=task_masks.get(d)) for d in preds}
    loss = sum(task_losses.values())
    for d, pr in preds.items():
        fx[f"pred_sub.{d}"], fx[f"pred_step.{d}"] = subsample(pr, 8192)
        fx[f"pred_norm.{d}"] = np.float64(pr.detach().double().norm().item())
        fx[f"loss.{d}"] = np.float64(task_losses[d].item())
    fx["loss"] = np.float64(loss.item())
    for k, v in inter.items():
        fx[k] = v
    if with_grads:
        loss.backward()
        names, norms = [], []
        for n_, p_ in model.named_parameters():
            if p_.grad is None:
                continue
            names.append(n_)
            norms.append(p_.grad.double().norm().item())
            if p_.grad.numel() <= 4096:
                fx[f"grad.{n_}"] = p_.grad.numpy().copy()
            else:
                fx[f"gradsub.{n_}"], _ = subsample(p_.grad, 2048)
        fx["grad_names"] = np.array(names)
        fx["grad_norms"] = np.array(norms, dtype=np.float64)
        fx["grad_total_norm"] = np.float64(get_grad_norm_(model.parameters()).item())
    fx["meta"] = np.array(json.dumps(dict(cfg=cfg, B=B, K=K, data_seed=data_seed, mask_seed=mask_seed)))
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **fx)
    print(f"   loss={fx['loss']:.6f}  enc_norm={fx['enc_tokens_norm']:.4f}")
    return model, x, (fx["ids_keep"], fx["ids_restore"])


def run_droppath_case(name, cfg, B, K, data_seed, mask_seed):
    """Stochastic depth (mirage/utils.py:103-134,249,259-262) in training mode: the reference draws torch.rand((B,1,1)) per
    active DropPath call, after the mask noise and in block order (attention branch, then MLP branch).  The draws are
    replayed from the same seed and committed, so that the oracle and the HIP engine can be fed the identical noise."""
    print(f"== {name}: B={B} K={K} drop_path_rate={cfg['drop_path_rate']}")
    with contextlib.redirect_stdout(io.StringIO()):
        model = build_reference(cfg, drop_path_rate=cfg["drop_path_rate"])
    model.load_state_dict(O.procedural_state_dict(cfg), strict=True)
    model.train()
    x = O.synthetic_batch(cfg, B, seed=data_seed)
    x_in = {d: x[d] for d in cfg["in_domains"]}
    p, u, g = replay_mask_noise(cfg, B, mask_seed)                    # leaves the generator where the encoder starts drawing
    rates = O.drop_path_rates(cfg)
    nact = sum(1 for r in rates if r > 0.0)
    dp = torch.stack([torch.stack([torch.rand((B, 1, 1)).reshape(B), torch.rand((B, 1, 1)).reshape(B)]) for _ in range(nact)])
    fx = {"p": p, "g": g, "dp_noise": dp.numpy()}
    for d, ud in zip(cfg["in_domains"], u):
        fx[f"u.{d}"] = ud
    captured = {}
    first = cfg["out_domains"][0]
    orig = model.output_adapters[first].forward

    def spy(encoder_tokens, input_info, ids_keep, ids_restore):
        captured["enc"] = encoder_tokens
        captured["ids_keep"], captured["ids_restore"] = ids_keep, ids_restore
        return orig(encoder_tokens=encoder_tokens, input_info=input_info, ids_keep=ids_keep, ids_restore=ids_restore)
    model.output_adapters[first].forward = spy
    torch.manual_seed(mask_seed)
    with contextlib.redirect_stdout(io.StringIO()):
        preds, task_masks = model(x_in, num_encoded_tokens=K, alphas=1.0, sample_tasks_uniformly=False)
    fx["ids_keep"], fx["ids_restore"] = captured["ids_keep"].numpy(), captured["ids_restore"].numpy()
    fx["enc_tokens"] = captured["enc"].detach().numpy()
    loss_fns = reference_losses(cfg)
    task_losses = {d: loss_fns[d](preds[d].float(), x[d], mask=task_masks.get(d)) for d in preds}
    loss = sum(task_losses.values())
    for d in preds:
        fx[f"loss.{d}"] = np.float64(task_losses[d].item())
    fx["loss"] = np.float64(loss.item())
    loss.backward()
    names, norms = [], []
    for n_, p_ in model.named_parameters():
        if p_.grad is not None:
            names.append(n_); norms.append(p_.grad.double().norm().item())
    fx["grad_names"] = np.array(names)
    fx["grad_norms"] = np.array(norms, dtype=np.float64)
    fx["grad_total_norm"] = np.float64(get_grad_norm_(model.parameters()).item())
    # the oracle must reproduce the reference from the replayed noise (this is what pins the replay order)
    ik, ir = torch.from_numpy(fx["ids_keep"]), torch.from_numpy(fx["ids_restore"])
    sd = {k: v.clone().requires_grad_(False) for k, v in O.procedural_state_dict(cfg).items()}
    with torch.no_grad():
        ol, _, oenc, _ = O.forward_loss(cfg, sd, x, ik, ir, dp_noise=dp)
    err = (oenc - captured["enc"].detach()).abs().max().item()
    print(f"   oracle vs reference: enc max abs err {err:.3e}, loss {float(ol):.6f} vs {loss.item():.6f}; dropped samples per draw:",
          [(torch.floor(1 - r + dp[j]) == 0).sum().item() for j, r in enumerate([r for r in rates if r > 0.0])])
    assert err < 1e-4 and abs(float(ol) - loss.item()) < 1e-5 * abs(loss.item())
    fx["meta"] = np.array(json.dumps(dict(cfg=cfg, B=B, K=K, data_seed=data_seed, mask_seed=mask_seed)))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **fx)


class ReplayedDropout(torch.nn.Module):
    """Stand-in for the reference's nn.Dropout modules (mirage/utils.py:152,172): same place in the graph, same semantics (zero with
    probability p, survivors scaled by 1 / (1 - p)), but the mask comes from the engine's counter-based definition
    (oracle.dropout_factor: Philox4x32-10 over (seed, step, stream, row, col)) instead of torch's unreplayable generator."""

    def __init__(self, p, seed, step, stream):
        super().__init__()
        self.p, self.seed, self.step, self.stream = p, seed, step, stream

    def forward(self, x):
        B, N, D = x.shape
        return x * O.dropout_factor(B * N, D, self.p, self.seed, self.step, self.stream).reshape(B, N, D).to(x.dtype)


def run_dropout_case(name, cfg, B, K, data_seed, mask_seed, drop_seed, drop_step):
    """nn.Dropout(drop_rate) after attn.proj and mlp.fc2 of every encoder block (mirage/model.py:54,87 -> mirage/utils.py:158,187) in
    training mode: the REFERENCE model, forward + losses + backward, with its Dropout modules replaced by ReplayedDropout."""
    p_drop = cfg["drop_rate"]
    print(f"== {name}: B={B} K={K} drop_rate={p_drop}")
    with contextlib.redirect_stdout(io.StringIO()):
        model = build_reference(cfg, drop_rate=p_drop)
    model.load_state_dict(O.procedural_state_dict(cfg), strict=True)
    model.train()
    n_swapped = 0
    for i, blk in enumerate(model.encoder):
        assert isinstance(blk.attn.proj_drop, torch.nn.Dropout) and blk.attn.proj_drop.p == p_drop
        assert isinstance(blk.mlp.drop, torch.nn.Dropout) and blk.mlp.drop.p == p_drop
        blk.attn.proj_drop = ReplayedDropout(p_drop, drop_seed, drop_step, 2 * i)
        blk.mlp.drop = ReplayedDropout(p_drop, drop_seed, drop_step, 2 * i + 1)
        n_swapped += 2
    left = [n for n, m in model.named_modules() if isinstance(m, torch.nn.Dropout) and m.p > 0]
    assert not left, left                                       # every active Dropout of the model is replayed
    x = O.synthetic_batch(cfg, B, seed=data_seed)
    x_in = {d: x[d] for d in cfg["in_domains"]}
    p, u, g = replay_mask_noise(cfg, B, mask_seed)
    fx = {"p": p, "g": g}
    for d, ud in zip(cfg["in_domains"], u):
        fx[f"u.{d}"] = ud
    captured = {}
    first = cfg["out_domains"][0]
    orig = model.output_adapters[first].forward

    def spy(encoder_tokens, input_info, ids_keep, ids_restore):
        captured["enc"] = encoder_tokens
        captured["ids_keep"], captured["ids_restore"] = ids_keep, ids_restore
        return orig(encoder_tokens=encoder_tokens, input_info=input_info, ids_keep=ids_keep, ids_restore=ids_restore)
    model.output_adapters[first].forward = spy
    torch.manual_seed(mask_seed)
    with contextlib.redirect_stdout(io.StringIO()):
        preds, task_masks = model(x_in, num_encoded_tokens=K, alphas=1.0, sample_tasks_uniformly=False)
    fx["ids_keep"], fx["ids_restore"] = captured["ids_keep"].numpy(), captured["ids_restore"].numpy()
    fx["enc_tokens"] = captured["enc"].detach().numpy()
    loss_fns = reference_losses(cfg)
    task_losses = {d: loss_fns[d](preds[d].float(), x[d], mask=task_masks.get(d)) for d in preds}
    loss = sum(task_losses.values())
    for d in preds:
        fx[f"loss.{d}"] = np.float64(task_losses[d].item())
    fx["loss"] = np.float64(loss.item())
    loss.backward()
    names, norms = [], []
    for n_, p_ in model.named_parameters():
        if p_.grad is not None:
            names.append(n_); norms.append(p_.grad.double().norm().item())
    fx["grad_names"] = np.array(names)
    fx["grad_norms"] = np.array(norms, dtype=np.float64)
    fx["grad_total_norm"] = np.float64(get_grad_norm_(model.parameters()).item())
    for n_ in ("encoder.0.attn.proj.weight", f"encoder.{cfg['depth'] - 1}.mlp.fc2.bias", "encoder.1.mlp.fc1.weight", "global_tokens"):
        fx[f"grad.{n_}"] = dict(model.named_parameters())[n_].grad.detach().numpy()
    # the same pass without dropout: how much the masks matter (a test that passes with dropout silently off would be worthless)
    with contextlib.redirect_stdout(io.StringIO()):
        ref0 = build_reference(cfg, drop_rate=0.0)
    ref0.load_state_dict(O.procedural_state_dict(cfg), strict=True)
    ref0.train()
    torch.manual_seed(mask_seed)
    with contextlib.redirect_stdout(io.StringIO()):
        preds0, tm0 = ref0(x_in, num_encoded_tokens=K, alphas=1.0, sample_tasks_uniformly=False)
    loss0 = sum(loss_fns[d](preds0[d].float(), x[d], mask=tm0.get(d)) for d in preds0)
    fx["loss_without_dropout"] = np.float64(loss0.item())
    # the oracle must reproduce the reference
    ik, ir = torch.from_numpy(fx["ids_keep"]), torch.from_numpy(fx["ids_restore"])
    sd = {k: v.clone().requires_grad_(False) for k, v in O.procedural_state_dict(cfg).items()}
    with torch.no_grad():
        ol, _, oenc, _ = O.forward_loss(cfg, sd, x, ik, ir, drop_state=(drop_seed, drop_step))
    err = (oenc - captured["enc"].detach()).abs().max().item()
    print(f"   oracle vs reference: enc max abs err {err:.3e}, loss {float(ol):.6f} vs {loss.item():.6f} (without dropout {loss0.item():.6f}); "
          f"{n_swapped} Dropout modules replayed")
    assert err < 1e-4 and abs(float(ol) - loss.item()) < 1e-5 * abs(loss.item())
    assert abs(loss0.item() - loss.item()) > 1e-3 * abs(loss.item())
    fx["meta"] = np.array(json.dumps(dict(cfg=cfg, B=B, K=K, data_seed=data_seed, mask_seed=mask_seed, drop_seed=drop_seed, drop_step=drop_step)))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **fx)


class ReplayedSDPA:
    """Stand-in for torch.nn.functional inside the reference's mirage/utils.py: scaled_dot_product_attention(q, k, v, scale, dropout_p)
    (utils.py:181-185, 216-220) computed as torch documents it -- softmax(scale q k^T), torch.dropout of the probabilities, @ v -- with
    the dropout mask taken from the engine's counter-based definition (oracle.attention_dropout_factor) instead of torch's generator.
    `streams`: the mask stream of every attention call of one forward pass, in call order.  Everything else is torch.nn.functional."""

    def __init__(self, seed, step, streams):
        self.seed, self.step, self.streams, self.calls = seed, step, list(streams), 0

    def __getattr__(self, name):
        return getattr(torch.nn.functional, name)

    def scaled_dot_product_attention(self, q, k, v, scale=None, dropout_p=0.0):
        B, H, Nq, hd = q.shape
        att = torch.softmax((q @ k.transpose(-1, -2)) * (scale if scale is not None else hd ** -0.5), dim=-1)
        stream = self.streams[self.calls % len(self.streams)]
        self.calls += 1
        if dropout_p > 0.0:
            att = att * O.attention_dropout_factor(B, H, Nq, k.shape[2], (dropout_p, self.seed, self.step, stream)).to(att.dtype)
        return att @ v


def run_attn_dropout_case(name, cfg, B, K, data_seed, mask_seed, drop_seed, drop_step):
    """attn_drop_rate > 0 in the encoder blocks (mirage/model.py:55,88) and in the output adapters (output_adapters.py:64): the REFERENCE
    model, forward + losses + backward, with torch.nn.functional.scaled_dot_product_attention replayed by ReplayedSDPA inside mirage.utils."""
    import mirage.utils as RU
    pe, pd = cfg["attn_drop_rate"], cfg["decoder_attn_drop_rate"]
    print(f"== {name}: B={B} K={K} attn_drop_rate={pe} decoder attn_drop_rate={pd}")
    with contextlib.redirect_stdout(io.StringIO()):
        model = build_reference(cfg, attn_drop_rate=pe, decoder_attn_drop_rate=pd)
    model.load_state_dict(O.procedural_state_dict(cfg), strict=True)
    model.train()
    streams = [0x10000 + i for i in range(cfg["depth"])]
    for a in range(len(cfg["out_domains"])):
        streams += [0x20000 + 256 * a] + [0x20000 + 256 * a + 1 + j for j in range(cfg["decoder_depth"])]
    x = O.synthetic_batch(cfg, B, seed=data_seed)
    x_in = {d: x[d] for d in cfg["in_domains"]}
    p, u, g = replay_mask_noise(cfg, B, mask_seed)
    fx = {"p": p, "g": g}
    for d, ud in zip(cfg["in_domains"], u):
        fx[f"u.{d}"] = ud
    captured = {}
    first = cfg["out_domains"][0]
    orig = model.output_adapters[first].forward

    def spy(encoder_tokens, input_info, ids_keep, ids_restore):
        captured["enc"] = encoder_tokens
        captured["ids_keep"], captured["ids_restore"] = ids_keep, ids_restore
        return orig(encoder_tokens=encoder_tokens, input_info=input_info, ids_keep=ids_keep, ids_restore=ids_restore)
    model.output_adapters[first].forward = spy
    shim = ReplayedSDPA(drop_seed, drop_step, streams)
    real_F, RU.F = RU.F, shim
    try:
        torch.manual_seed(mask_seed)
        with contextlib.redirect_stdout(io.StringIO()):
            preds, task_masks = model(x_in, num_encoded_tokens=K, alphas=1.0, sample_tasks_uniformly=False)
        assert shim.calls == len(streams), (shim.calls, len(streams))      # every attention of the model went through the replay
        fx["ids_keep"], fx["ids_restore"] = captured["ids_keep"].numpy(), captured["ids_restore"].numpy()
        fx["enc_tokens"] = captured["enc"].detach().numpy()
        loss_fns = reference_losses(cfg)
        task_losses = {d: loss_fns[d](preds[d].float(), x[d], mask=task_masks.get(d)) for d in preds}
        loss = sum(task_losses.values())
        for d in preds:
            fx[f"loss.{d}"] = np.float64(task_losses[d].item())
            fx[f"pred.{d}"] = preds[d].detach().numpy().astype(np.float32)
        fx["loss"] = np.float64(loss.item())
        loss.backward()
    finally:
        RU.F = real_F
    names, norms = [], []
    for n_, p_ in model.named_parameters():
        if p_.grad is not None:
            names.append(n_); norms.append(p_.grad.double().norm().item())
    fx["grad_names"] = np.array(names)
    fx["grad_norms"] = np.array(norms, dtype=np.float64)
    fx["grad_total_norm"] = np.float64(get_grad_norm_(model.parameters()).item())
    for n_ in ("encoder.0.attn.qkv.weight", f"encoder.{cfg['depth'] - 1}.attn.qkv.bias", f"output_adapters.{first}.decoder.kv.weight",
               f"output_adapters.{first}.decoder_transformer.0.attn.qkv.weight", "global_tokens"):
        fx[f"grad.{n_}"] = dict(model.named_parameters())[n_].grad.detach().numpy()
    # the same pass without the dropout: the masks must matter
    with contextlib.redirect_stdout(io.StringIO()):
        ref0 = build_reference(cfg)
    ref0.load_state_dict(O.procedural_state_dict(cfg), strict=True)
    ref0.train()
    torch.manual_seed(mask_seed)
    with contextlib.redirect_stdout(io.StringIO()):
        preds0, tm0 = ref0(x_in, num_encoded_tokens=K, alphas=1.0, sample_tasks_uniformly=False)
    loss0 = sum(loss_fns[d](preds0[d].float(), x[d], mask=tm0.get(d)) for d in preds0)
    fx["loss_without_dropout"] = np.float64(loss0.item())
    # the oracle must reproduce the reference
    ik, ir = torch.from_numpy(fx["ids_keep"]), torch.from_numpy(fx["ids_restore"])
    sd = {k: v.clone().requires_grad_(False) for k, v in O.procedural_state_dict(cfg).items()}
    with torch.no_grad():
        ol, _, oenc, opreds = O.forward_loss(cfg, sd, x, ik, ir, drop_state=(drop_seed, drop_step))
    err = (oenc - captured["enc"].detach()).abs().max().item()
    perr = max((opreds[d] - preds[d].detach()).abs().max().item() for d in preds)
    print(f"   oracle vs reference: enc max abs err {err:.3e}, pred max abs err {perr:.3e}, loss {float(ol):.6f} vs {loss.item():.6f} "
          f"(without dropout {loss0.item():.6f}); {shim.calls} attention calls replayed")
    assert err < 1e-4 and perr < 1e-3 and abs(float(ol) - loss.item()) < 1e-5 * abs(loss.item())
    assert abs(loss0.item() - loss.item()) > 1e-4 * abs(loss.item())
    fx["meta"] = np.array(json.dumps(dict(cfg=cfg, B=B, K=K, data_seed=data_seed, mask_seed=mask_seed, drop_seed=drop_seed, drop_step=drop_step)))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **fx)


def run_loss_smoothing_fixture(name="loss_smoothing"):
    """MaskedCrossEntropyLoss(label_smoothing = e) of the reference (mirage/criterion.py:11-51) on a small random problem: inputs, the
    loss and d loss / d logits for e in {0.1, 0.3}, with a token mask (one sample fully unmasked-out: the nanmean branch) and without."""
    print(f"== {name}")
    g = torch.Generator().manual_seed(77)
    B, C, H, P = 4, 13, 16, 4
    logits = torch.randn(B, C, H, H, generator=g) * 2.0
    target = torch.randint(0, C, (B, H, H), generator=g)
    mask = (torch.rand(B, (H // P) ** 2, generator=g) < 0.6).long()
    mask[2] = 0                                                   # a sample without a masked token: 0 / 0 -> dropped by nanmean
    fx = {"logits": logits.numpy(), "target": target.numpy(), "mask": mask.numpy(), "patch": np.int64(P)}
    for e in (0.1, 0.3):
        fn = MaskedCrossEntropyLoss(patch_size=(P, P), stride=1, label_smoothing=e)
        for tag, m in (("masked", mask), ("plain", None)):
            x = logits.clone().requires_grad_(True)
            l = fn(x, target, mask=m)
            l.backward()
            fx[f"loss.{e}.{tag}"] = np.float64(l.item())
            fx[f"dlogits.{e}.{tag}"] = x.grad.numpy().copy()
            o = O.masked_ce(logits, target, m, P, label_smoothing=e)
            assert abs(float(o) - l.item()) < 1e-6 * abs(l.item()), (e, tag, float(o), l.item())
    fx["meta"] = np.array(json.dumps(dict(B=B, C=C, H=H, P=P)))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **fx)
    print("   losses:", {k: float(v) for k, v in fx.items() if k.startswith("loss.")})


def run_norm_pix_fixture(name="loss_norm_pix"):
    """MaskedMSELoss(norm_pix=True) of the reference (mirage/criterion.py:70-117) on small random problems (1 and 3 channels, patch 8 / 4):
    inputs, loss and d loss / d pred, with a token mask (one sample without a masked token) and without."""
    print(f"== {name}")
    g = torch.Generator().manual_seed(78)
    fx = {}
    for tag, (B, C, H, P) in (("c1", (4, 1, 32, 8)), ("c3", (3, 3, 16, 4))):
        pred = torch.randn(B, C, H, H, generator=g)
        target = torch.rand(B, C, H, H, generator=g) * 3.0 + 0.5
        mask = (torch.rand(B, (H // P) ** 2, generator=g) < 0.6).long()
        mask[1] = 0
        fx[f"{tag}.pred"], fx[f"{tag}.target"], fx[f"{tag}.mask"], fx[f"{tag}.patch"] = pred.numpy(), target.numpy(), mask.numpy(), np.int64(P)
        fn = MaskedMSELoss(patch_size=P, stride=1, norm_pix=True)
        for mt, m in (("masked", mask), ("plain", None)):
            x = pred.clone().requires_grad_(True)
            l = fn(x, target, mask=m)
            l.backward()
            fx[f"{tag}.loss.{mt}"] = np.float64(l.item())
            fx[f"{tag}.dpred.{mt}"] = x.grad.numpy().copy()
            o = O.masked_mse(pred, target, m, P, norm_pix=True)
            assert abs(float(o) - l.item()) < 1e-6 * abs(l.item()), (tag, mt, float(o), l.item())
            plain = MaskedMSELoss(patch_size=P, stride=1)(pred, target, mask=m)
            assert abs(plain.item() - l.item()) > 1e-2 * abs(l.item())
    fx["meta"] = np.array(json.dumps(dict(cases=["c1", "c3"])))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **fx)
    print("   losses:", {k: float(v) for k, v in fx.items() if ".loss." in k})


def run_train_curve(name, cfg, B, K, steps=5):
    """5-step AdamW curve through the reference's optimizer factory + cosine schedules
    (run_pretraining.py:496-515,683-742), fp32, no AMP."""
    print(f"== {name}: train curve")
    with contextlib.redirect_stdout(io.StringIO()):
        model = build_reference(cfg)
    model.load_state_dict(O.procedural_state_dict(cfg), strict=True)
    model.train()
    args = types.SimpleNamespace(opt="adamw", weight_decay=0.05, lr=1e-3, opt_eps=1e-8, opt_betas=[0.9, 0.95], momentum=0.9)
    with contextlib.redirect_stdout(io.StringIO()):
        opt = create_optimizer(args, model)
        lrs = cosine_scheduler(args.lr, 0.0, epochs=steps, niter_per_ep=1, warmup_epochs=2)
        wds = cosine_scheduler(0.05, 0.05, epochs=steps, niter_per_ep=1)
    loss_fns = reference_losses(cfg)
    fx = {"lrs": lrs, "wds": wds}
    curve, norms = [], []
    for it in range(steps):
        for gr in opt.param_groups:
            gr["lr"] = lrs[it] * gr["lr_scale"]
            if gr["weight_decay"] > 0:
                gr["weight_decay"] = wds[it]
        x = O.synthetic_batch(cfg, B, seed=500 + it)
        p, u, g = replay_mask_noise(cfg, B, 700 + it)
        fx[f"p.{it}"] = p; fx[f"g.{it}"] = g
        for d, ud in zip(cfg["in_domains"], u):
            fx[f"u.{it}.{d}"] = ud
        torch.manual_seed(700 + it)
        with contextlib.redirect_stdout(io.StringIO()):
            preds, masks = model({d: x[d] for d in cfg["in_domains"]}, num_encoded_tokens=K, alphas=1.0)
        loss = sum(loss_fns[d](preds[d].float(), x[d], mask=masks.get(d)) for d in preds)
        opt.zero_grad()
        loss.backward()
        norms.append(get_grad_norm_(model.parameters()).item())
        opt.step()
        curve.append(loss.item())
    fx["curve"] = np.array(curve, dtype=np.float64)
    fx["grad_norms"] = np.array(norms, dtype=np.float64)
    names = [n for n, _ in model.named_parameters()]
    fx["final_names"] = np.array(names)
    fx["final_norms"] = np.array([p.detach().double().norm().item() for _, p in model.named_parameters()])
    fx["final_global_tokens"] = model.global_tokens.detach().numpy().copy()
    fx["meta"] = np.array(json.dumps(dict(cfg=cfg, B=B, K=K, steps=steps, lr=args.lr)))
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **fx)
    print("   curve", curve)


def run_optstate_case(name, cfg, B, K, steps=4):
    """Step boundary with the reference's own scaler (mutils/native_scaler.py:10-34, GradScaler disabled = the fp32 / bf16 path):
    (a) plain: optimizer.state_dict() after 2 steps -- the checkpoint['optimizer'] format (mutils/checkpoint.py:14-20);
    (b) clip_grad: 3 steps with clipping active; (c) skip_grad: `steps` steps with a threshold between the observed norms, so
    that some updates are skipped and AdamW's per-parameter `step` falls behind the iteration counter."""
    import warnings
    from mutils.native_scaler import NativeScalerWithGradNormCount
    print(f"== {name}: optimizer state / clip / skip")
    loss_fns = reference_losses(cfg)
    args = types.SimpleNamespace(opt="adamw", weight_decay=0.05, lr=1e-3, opt_eps=1e-8, opt_betas=[0.9, 0.95], momentum=0.9)
    with contextlib.redirect_stdout(io.StringIO()):
        lrs = cosine_scheduler(args.lr, 1e-4, epochs=steps, niter_per_ep=1, warmup_epochs=1, start_warmup_value=5e-4)
        wds = cosine_scheduler(0.05, 0.02, epochs=steps, niter_per_ep=1)
    fx = {"lrs": lrs, "wds": wds}
    for it in range(steps):
        p, u, g = replay_mask_noise(cfg, B, 900 + it)
        fx[f"p.{it}"] = p; fx[f"g.{it}"] = g
        for d, ud in zip(cfg["in_domains"], u):
            fx[f"u.{it}.{d}"] = ud

    def run(n_steps, clip_grad=None, skip_grad=None):
        with contextlib.redirect_stdout(io.StringIO()):
            model = build_reference(cfg)
        model.load_state_dict(O.procedural_state_dict(cfg), strict=True)
        model.train()
        with contextlib.redirect_stdout(io.StringIO()), warnings.catch_warnings():
            warnings.simplefilter("ignore")
            opt = create_optimizer(args, model)
            scaler = NativeScalerWithGradNormCount(enabled=False)
        norms, losses, adam_steps = [], [], []
        for it in range(n_steps):
            for gr in opt.param_groups:                      # run_pretraining.py:683-688
                gr["lr"] = lrs[it] * gr["lr_scale"]
                if gr["weight_decay"] > 0:
                    gr["weight_decay"] = wds[it]
            x = O.synthetic_batch(cfg, B, seed=800 + it)
            if it == 2:                                      # one out-of-range batch: a gradient-norm spike for the skip branch
                x = {d: (t * 4.0 if t.is_floating_point() else t) for d, t in x.items()}
            torch.manual_seed(900 + it)
            with contextlib.redirect_stdout(io.StringIO()):
                preds, masks = model({d: x[d] for d in cfg["in_domains"]}, num_encoded_tokens=K, alphas=1.0)
            loss = sum(loss_fns[d](preds[d].float(), x[d], mask=masks.get(d)) for d in preds)
            opt.zero_grad()
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                norm = scaler(loss, opt, clip_grad=clip_grad, skip_grad=skip_grad, parameters=model.parameters())
            norms.append(float(norm)); losses.append(float(loss))
            st = [float(v["step"]) for v in opt.state.values()]
            adam_steps.append(max(st) if st else 0.0)
        return model, opt, norms, losses, adam_steps

    # (a) plain, 2 steps: the state dict
    model, opt, norms, losses, adam_steps = run(2)
    sd = opt.state_dict()
    id2name = {id(p): n for n, p in model.named_parameters()}
    order = [id2name[id(p)] for gr in opt.param_groups for p in gr["params"]]
    fx["plain.order"] = np.array(order)                       # parameter index -> name, as torch numbers them
    fx["plain.groups"] = np.array(json.dumps([{k: (v if k != "params" else list(v)) for k, v in gr.items()} for gr in sd["param_groups"]],
                                             default=float))
    fx["plain.state_keys"] = np.array(sorted({k for v in sd["state"].values() for k in v}))
    fx["plain.step_dtype"] = np.array(str(next(iter(sd["state"].values()))["step"].dtype))
    fx["plain.step_shape"] = np.array(list(next(iter(sd["state"].values()))["step"].shape), dtype=np.int64)
    fx["plain.steps"] = np.array([float(sd["state"][i]["step"]) for i in range(len(order))])
    for i, n in enumerate(order):
        fx[f"plain.exp_avg.{i}"], _ = subsample(sd["state"][i]["exp_avg"], 512)
        fx[f"plain.exp_avg_sq.{i}"], _ = subsample(sd["state"][i]["exp_avg_sq"], 512)
    fx["plain.norms"] = np.array(norms); fx["plain.losses"] = np.array(losses)
    # (b) clipping active on every step
    clip = 0.25 * min(norms)
    model, opt, cn, cl, cs = run(3, clip_grad=clip, skip_grad=1e-3)          # skip_grad must be ignored when clip_grad is set
    fx["clip.value"] = np.array(clip); fx["clip.norms"] = np.array(cn); fx["clip.losses"] = np.array(cl); fx["clip.adam_steps"] = np.array(cs)
    fx["clip.final_norms"] = np.array([p.detach().double().norm().item() for _, p in model.named_parameters()])
    for n_ in ("encoder.0.mlp.fc1.weight", "encoder.1.attn.qkv.bias", "global_tokens"):
        fx[f"clip.final.{n_}"], _ = subsample(dict(model.named_parameters())[n_], 512)
    # (c) skipping: threshold from a dry run so that both branches are taken
    _, _, dry, _, _ = run(steps)
    srt = sorted(dry)
    thr = 0.5 * (srt[-1] + srt[-2])                          # the iteration with the largest norm is skipped, the ones before it are not
    model, opt, sn, sl, ss = run(steps, skip_grad=thr)
    taken = [b > a for a, b in zip([0.0] + ss[:-1], ss)]
    assert any(taken) and not all(taken), (thr, sn, ss)
    fx["skip.value"] = np.array(thr); fx["skip.norms"] = np.array(sn); fx["skip.losses"] = np.array(sl); fx["skip.adam_steps"] = np.array(ss)
    fx["skip.final_norms"] = np.array([p.detach().double().norm().item() for _, p in model.named_parameters()])
    for n_ in ("encoder.0.mlp.fc1.weight", "encoder.1.attn.qkv.bias", "global_tokens"):
        fx[f"skip.final.{n_}"], _ = subsample(dict(model.named_parameters())[n_], 512)
    fx["names"] = np.array([n for n, _ in model.named_parameters()])
    fx["meta"] = np.array(json.dumps(dict(cfg=cfg, B=B, K=K, steps=steps, lr=args.lr)))
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), **fx)
    print("   plain norms", norms, "clip", clip, cn, "skip thr", thr, sn, ss)


def run_harness_fixtures(cfg):
    """Schedules, param groups, posemb quirk, mask generator on its own."""
    fx = {}
    with contextlib.redirect_stdout(io.StringIO()):
        fx["sched.a"] = cosine_scheduler(1e-4, 0.0, 10, 7, warmup_epochs=2)
        fx["sched.b"] = cosine_scheduler(0.05, 0.05, 10, 7)
        fx["sched.c"] = cosine_scheduler(2e-4, 1e-6, 6, 5, warmup_epochs=1, warmup_steps=3)
        model = build_reference(cfg)
    # param groups by identity
    id2name = {id(p): n for n, p in model.named_parameters()}
    groups = get_parameter_groups(model, 0.05, model.no_weight_decay())
    for gr in groups:
        key = "decay" if gr["weight_decay"] > 0 else "no_decay"
        fx[f"group.{key}"] = np.array([id2name[id(p)] for p in gr["params"]])
    fx["no_weight_decay"] = np.array(sorted(model.no_weight_decay()))
    fx["state_dict_keys"] = np.array(list(model.state_dict().keys()))
    fx["state_dict_shapes"] = np.array([json.dumps(list(v.shape)) for v in model.state_dict().values()])
    fx["trainable"] = np.array([n for n, p in model.named_parameters() if p.requires_grad])
    # posemb (incl. the non-square axis quirk)
    fx["posemb.2x3x8"] = build_2d_sincos_posemb(2, 3, 8).numpy()
    fx["posemb.16x16x768"] = build_2d_sincos_posemb(16, 16, 768).numpy()
    fx["posemb.16x16x256"] = build_2d_sincos_posemb(16, 16, 256).numpy()
    # pos-embed checkpoint resampler (mutils/pos_embed.py): 8x8 -> 4x4? use model grid vs a 2x finer table
    D = cfg["dim"]; gsz = cfg["domains"][cfg["in_domains"][0]]["grid"]
    ck = {f"input_adapters.{cfg['in_domains'][0]}.pos_emb": build_2d_sincos_posemb(2 * gsz, 2 * gsz, D)}
    fx["posemb_resample.in"] = ck[f"input_adapters.{cfg['in_domains'][0]}.pos_emb"].numpy().copy()
    with contextlib.redirect_stdout(io.StringIO()):
        interpolate_pos_embed(model, ck)
    fx["posemb_resample.out"] = ck[f"input_adapters.{cfg['in_domains'][0]}.pos_emb"].numpy()
    # mask generator alone, several seeds, both sampling modes
    layout, T = O.token_layout(cfg)
    for seed in (0, 1, 2):
        B, K = 8, 24
        p, u, g = replay_mask_noise(cfg, B, seed)
        toks = {d: torch.zeros(B, e - s, 1) for d, (s, e) in layout.items()}
        torch.manual_seed(seed)
        with contextlib.redirect_stdout(io.StringIO()):
            tm, ik, ir = model.generate_random_masks(toks, K, alphas=1.0, sample_tasks_uniformly=False)
        fx[f"mask.{seed}.p"] = p; fx[f"mask.{seed}.g"] = g
        for d, ud in zip(cfg["in_domains"], u):
            fx[f"mask.{seed}.u.{d}"] = ud
            fx[f"mask.{seed}.tm.{d}"] = tm[d].numpy()
        fx[f"mask.{seed}.ids_keep"] = ik.numpy(); fx[f"mask.{seed}.ids_restore"] = ir.numpy()
    # uniform task pre-sampling (model.py:145-166,205): capture alphas -> Dirichlet sample
    torch.manual_seed(5)
    B, K = 8, 24
    al = model.sample_alphas(B, len(cfg["in_domains"]), alphas=[1.0] * len(cfg["in_domains"]))
    fx["mask.uni.alphas"] = al.numpy()
    p = Dirichlet(al).sample()
    u = [torch.rand(B, e - s) for (s, e) in layout.values()]
    g = torch.rand(B, T)
    torch.manual_seed(5)
    toks = {d: torch.zeros(B, e - s, 1) for d, (s, e) in layout.items()}
    with contextlib.redirect_stdout(io.StringIO()):
        tm, ik, ir = model.generate_random_masks(toks, K, alphas=1.0, sample_tasks_uniformly=True)
    fx["mask.uni.p"] = p.numpy(); fx["mask.uni.g"] = g.numpy()
    for d, ud in zip(cfg["in_domains"], u):
        fx[f"mask.uni.u.{d}"] = ud.numpy()
        fx[f"mask.uni.tm.{d}"] = tm[d].numpy()
    fx["mask.uni.ids_keep"] = ik.numpy(); fx["mask.uni.ids_restore"] = ir.numpy()
    # initialisation parity: reference model built under a fixed seed (mirage/model.py:95-121 init sequence)
    torch.manual_seed(1234)
    with contextlib.redirect_stdout(io.StringIO()):
        m2 = build_reference(cfg)
    sd2 = m2.state_dict()
    fx["init.names"] = np.array(list(sd2.keys()))
    fx["init.norms"] = np.array([float(v.double().norm()) for v in sd2.values()])
    fx["init.sums"] = np.array([float(v.double().sum()) for v in sd2.values()])
    fx["init.global_tokens"] = sd2["global_tokens"].numpy().copy()
    fx["init.enc0_qkv_w_head"] = sd2["encoder.0.attn.qkv.weight"][:4, :8].numpy().copy()
    fx["meta"] = np.array(json.dumps(dict(cfg=cfg)))
    np.savez_compressed(os.path.join(OUT, "harness_tiny.npz"), **fx)
    print("== harness fixtures written")


def run_cls_case(name, cfg, B):
    """Encoder-only feature path (mirage_wrapper.py:208-224 restated around the reference model):
    model(x, mask_inputs=False) -> LayerNorm -> mean over patch tokens.  Order-invariant pooled output."""
    print(f"== {name}: encoder-only pooled features")
    cfg = dict(cfg); cfg["out_domains"] = []
    with contextlib.redirect_stdout(io.StringIO()):
        model = build_reference(cfg)
    sd = O.procedural_state_dict(cfg)
    model.load_state_dict(sd, strict=True)
    model.eval()
    d = cfg["in_domains"][0]
    x = O.synthetic_batch(cfg, B, seed=4321)[d]
    D = cfg["dim"]
    nw = torch.from_numpy(1.0 + 0.2 * O.procedural_uniform("cls.norm.weight", D))
    nb = torch.from_numpy(0.04 * O.procedural_uniform("cls.norm.bias", D))
    torch.manual_seed(3)
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()):
        out, _ = model({d: x}, mask_inputs=False)
        out = torch.nn.functional.layer_norm(out, (D,), nw, nb, 1e-6)
        pooled = out[:, :-cfg["num_global_tokens"], :].mean(dim=1)
        glob = out[:, -cfg["num_global_tokens"]:, :].mean(dim=1)
    np.savez_compressed(os.path.join(OUT, f"{name}.npz"), pooled=pooled.numpy(), global_=glob.numpy(),
                        meta=np.array(json.dumps(dict(cfg=cfg, B=B, data_seed=4321))))


def tiny_cfg(in_domains, out_domains=None):
    return O.make_config(model="tiny", in_domains=in_domains, out_domains=out_domains, input_size=64, patch_size=8,
                         layermap_size=16, layermap_patch=2, decoder_dim=64, decoder_depth=2, decoder_heads=2,
                         dim=128, depth=2, heads=2)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    three = ("bscan", "slo", "bscanlayermap")
    jobs = {
        "tiny3": lambda: run_case("tiny3", tiny_cfg(three), B=3, K=24, data_seed=11, mask_seed=21, keep_intermediate=True),
        "tiny2": lambda: run_case("tiny2", tiny_cfg(("bscan", "slo")), B=2, K=32, data_seed=12, mask_seed=22, keep_intermediate=True),
        # learnable_pos_emb=True on the input adapters (mirage/input_adapters.py:72,185): the positional tables get gradients
        "tiny3_learnpos": lambda: run_case("tiny3_learnpos", dict(tiny_cfg(three), learnable_pos_emb=True), B=5, K=24, data_seed=14, mask_seed=24),
        "loss_smoothing": lambda: run_loss_smoothing_fixture(),
        "loss_norm_pix": lambda: run_norm_pix_fixture(),
        "tiny3_curve": lambda: run_train_curve("tiny3_curve", tiny_cfg(three), B=4, K=24, steps=5),
        "harness": lambda: run_harness_fixtures(tiny_cfg(three)),
        "tiny3_optstate": lambda: run_optstate_case("tiny3_optstate", tiny_cfg(three), B=4, K=24, steps=4),
        "tiny3_dropout": lambda: run_dropout_case("tiny3_dropout", dict(O.make_config(model="tiny", in_domains=three, input_size=64, patch_size=8,
                                                  layermap_size=16, layermap_patch=2, decoder_dim=64, decoder_depth=2, decoder_heads=2,
                                                  dim=128, depth=4, heads=2), drop_rate=0.1), B=6, K=24, data_seed=17, mask_seed=29,
                                                  drop_seed=0x1234ABCD5678, drop_step=3),
        "tiny3_attn_dropout": lambda: run_attn_dropout_case("tiny3_attn_dropout", dict(O.make_config(model="tiny", in_domains=three, input_size=64,
                                                            patch_size=8, layermap_size=16, layermap_patch=2, decoder_dim=64, decoder_depth=2,
                                                            decoder_heads=2, dim=128, depth=4, heads=2), attn_drop_rate=0.15,
                                                            decoder_attn_drop_rate=0.1), B=6, K=24, data_seed=19, mask_seed=31,
                                                            drop_seed=0x0BADC0FFEE12, drop_step=5),
        "tiny3_droppath": lambda: run_droppath_case("tiny3_droppath", dict(O.make_config(model="tiny", in_domains=three, input_size=64, patch_size=8,
                                                    layermap_size=16, layermap_patch=2, decoder_dim=64, decoder_depth=2, decoder_heads=2,
                                                    dim=128, depth=4, heads=2, drop_path_rate=0.4)), B=6, K=24, data_seed=13, mask_seed=23),
        "tiny_cls": lambda: run_cls_case("tiny_cls", tiny_cfg(("bscan",)), B=3),
        # BASELINE.json config 1: Base, 128/8, bscan only, B=2 (the reference's own CPU-runnable case), K=98 (run_pretraining.py:97)
        "c1_base": lambda: run_case("c1_base", O.make_config("base", ("bscan",), input_size=128, patch_size=8), B=2, K=98,
                                    data_seed=1234, mask_seed=30, full_enc=False),
        # BASELINE.json configs at their real shapes, small batch
        "c2_base": lambda: run_case("c2_base", O.make_config("base", ("bscan", "slo")), B=2, K=128, data_seed=1234, mask_seed=31),
        "c3_base": lambda: run_case("c3_base", O.make_config("base", three), B=2, K=98, data_seed=1234, mask_seed=32, full_enc=False),
        "c4_large": lambda: run_case("c4_large", O.make_config("large", three), B=1, K=98, data_seed=1234, mask_seed=33, full_enc=False),
        "c5_large_cls": lambda: run_cls_case("c5_large_cls", O.make_config("large", ("bscan",), out_domains=()), B=1),
    }
    for k, fn in jobs.items():
        if a.only and k not in a.only.split(","):
            continue
        fn()


if __name__ == "__main__":
    main()
