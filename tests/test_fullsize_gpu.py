"""Parity THROUGH THE PRODUCTION PATH at the BASELINE sizes (VERDICT r01 "what's weak" #1): the graph-replayed TrainStep -- side
streams, deferred split-K / column-sum reductions, XCD-banded split-K, epilogue column sums, early block-wise AdamW, the tuned
GEMM variants of gemm_tune_gfx950.json -- at per-GPU batch 64 (C2, C3) / 16 (C4) against the CPU oracle on the same batch and
the same mask noise; eager == graph bit-exactly at these shapes; and two data-parallel engine ranks sharing one GPU (gloo)
against one process on the concatenated batch.

Reference: mirage/model.py:305-431 (forward), mirage/criterion.py:31-117 (losses), run_pretraining.py:683-742 (step).
Tolerances: total loss <= 1e-3 relative to the fp32 oracle (BASELINE.json north_star) and <= 2e-4 to the oracle with bf16 weight matrices
(O.autocast_weights: what the reference's autocast multiplies with); per task <= 1.5e-3 / 4e-4; global gradient norm <= 1e-2; sampled
parameter gradients: norm within 2e-2, full-tensor rel-RMS within 2e-2 of the fp32 oracle (bf16 operands).  Every measured
error is printed and appended to gpurun_out/parity_measured.jsonl.
"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _sample_names(cfg):
    L = cfg["depth"]
    d0, dl = cfg["out_domains"][0], cfg["out_domains"][-1]
    i0 = cfg["in_domains"][0]
    return ["global_tokens", f"input_adapters.{i0}.proj.weight", "encoder.0.attn.qkv.weight", "encoder.0.mlp.fc2.bias",
            f"encoder.{L // 2}.mlp.fc1.weight", f"encoder.{L // 2}.norm1.weight", f"encoder.{L - 1}.attn.proj.weight",
            f"encoder.{L - 1}.mlp.fc2.weight", f"output_adapters.{d0}.decoder.kv.weight", f"output_adapters.{d0}.proj_context.weight",
            f"output_adapters.{dl}.out_proj.weight", f"output_adapters.{dl}.decoder_transformer.1.mlp.fc1.weight"]


def _run_engine(cfg, B, Kk, x, noise, use_graph):
    from mirage_amd import builder, optim
    from mirage_amd.trainer import TrainStep
    import test_model_gpu as T
    model, args, _ = T._build(cfg)
    model.train()
    loss_fns = builder.get_losses(args)
    opt = optim.FusedAdamW(model, [1e-4] * 8, [0.05] * 8, betas=(0.9, 0.95), eps=1e-8)
    xg = {d: t.to(DEV) for d, t in x.items()}
    ts = TrainStep(model, loss_fns, opt, xg, Kk, use_graph=use_graph, feed_mask_noise=True)
    if use_graph:
        # the warm-up iteration and the capture run on OTHER data than the measured replay: a mis-ordered kernel that reads what an
        # earlier iteration left behind cannot hide behind identical stale values (tests/test_train_gpu.py::test_graph_step_does_not_...)
        from oracle import mirage_oracle as O
        xw = {d: t.to(DEV) for d, t in O.synthetic_batch(cfg, B, seed=4321).items()}
        (pw, uw, gw), _, _, _ = O.synthetic_masks(cfg, B, Kk, seed=77)
        nw = (torch.from_numpy(pw).to(DEV), torch.cat([torch.from_numpy(a_) for a_ in uw], 1).to(DEV), torch.from_numpy(gw).to(DEV))
        ts.load(xw)
        for dst, src in zip(ts.static_noise, nw):
            dst.copy_(src)
        ts.warmup_and_capture(warm_iters=1)
        assert ts.g_fb is not None and ts._early is not None        # one graph, early block-wise AdamW on (the bench configuration)
    ts.step(xg, noise=noise)
    torch.cuda.synchronize()
    return model, opt, ts


def _fullsize_case(tag, kind, doms, B, Kk, check_eager=True, data_seed=1234, mask_seed=99, asserts=True):
    """-> the measured errors; asserts=False (tools/parity_seeds.py: the same case over other data / mask seeds) only measures"""
    from oracle import mirage_oracle as O
    import test_model_gpu as T
    cfg = O.make_config(kind, doms)
    x = O.synthetic_batch(cfg, B, seed=data_seed)
    (p, u, g), tm, ids_keep, ids_restore = O.synthetic_masks(cfg, B, Kk, seed=mask_seed)
    noise = (torch.from_numpy(p).to(DEV), torch.cat([torch.from_numpy(a) for a in u], 1).to(DEV), torch.from_numpy(g).to(DEV))
    model, opt, ts = _run_engine(cfg, B, Kk, x, noise, use_graph=True)
    loss = float(ts.loss)
    task = {d: float(v) for d, v in ts.task_losses.items()}
    gn = float(opt.grad_norm)
    params = dict(model.named_parameters())
    names = _sample_names(cfg)
    grads = {n: params[n].grad.detach().float().cpu().clone() for n in names}
    flat_p, flat_g = opt.arena.flat_p.clone(), opt.arena.flat_g.clone()
    assert opt.step_dev.tolist() == [1, 1]
    # ---- oracle: fp32 CPU restatement on the same batch / masks / weights
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    sdr = {k: v.clone().requires_grad_(O.param_shapes(cfg)[k][1]) for k, v in O.procedural_state_dict(cfg).items()}
    ref_loss, ref_ls, _, _ = O.forward_loss(cfg, sdr, x, ids_keep, ids_restore)
    ref_loss.backward()
    ref_gn = float(torch.sqrt(sum((t.grad.double() ** 2).sum() for t in sdr.values() if t.grad is not None)))
    meas = {"B": B, "loss_rel": abs(loss - float(ref_loss)) / abs(float(ref_loss)), "loss_tol": 1e-3,
            "grad_norm_rel": abs(gn - ref_gn) / ref_gn, "grad_norm_tol": 1e-2}
    for d in task:
        meas[f"loss_rel.{d}"] = abs(task[d] - float(ref_ls[d])) / max(1.0, abs(float(ref_ls[d])))
    # ... and against the oracle evaluated with the weight matrices in the format the reference's autocast region multiplies with
    # (O.autocast_weights: bf16-rounded, everything else fp32).  For a fixed set of weights that rounding shifts every task's loss by a
    # fixed amount whatever the batch (C2 slo -9.5e-4, C4 bscan -8.8e-4: tools/debug/parity_bias.py, profiles/r05_parity_measured.md),
    # which is most of the distance to the fp32 oracle; what is left is the engine's own arithmetic
    with torch.no_grad():
        w_total, w_ls, _, _ = O.forward_loss(cfg, O.autocast_weights(O.procedural_state_dict(cfg)), x, ids_keep, ids_restore)
    meas["loss_rel_w16"] = abs(loss - float(w_total)) / abs(float(w_total))
    for d in task:
        meas[f"loss_rel_w16.{d}"] = abs(task[d] - float(w_ls[d])) / max(1.0, abs(float(w_ls[d])))
    worst_n, worst_r = (0.0, ""), (0.0, "")
    for n in names:
        r = sdr[n].grad.double()
        mine = grads[n].double()
        scale = max(float(r.norm()), 1e-3 * ref_gn)
        en = abs(float(mine.norm()) - float(r.norm())) / scale
        er = float((mine - r).norm()) / scale
        worst_n = max(worst_n, (en, n)); worst_r = max(worst_r, (er, n))
    meas.update(grad_norm_worst=worst_n[0], grad_norm_worst_name=worst_n[1], grad_norm_worst_tol=2e-2,
                grad_rms_worst=worst_r[0], grad_rms_worst_name=worst_r[1], grad_rms_worst_tol=2e-2, data_seed=data_seed, mask_seed=mask_seed)
    # ---- the UPDATE (mutils/optim_factory.py:180 torch.optim.AdamW, first step: m_hat = g, sqrt(v_hat) = |g|).  The matrices leave the grouped
    # weight-gradient launches already updated (epi_kind 9), the rest is updated block by block: (i) every sampled parameter after the step
    # must be AdamW's first step applied to the engine's OWN gradient (in units of lr; fp32 roundoff of p is ~1e-3 lr), (ii) against the
    # oracle's step from the oracle's gradient the first step is lr * sign(g): the share of elements whose sign differs is recorded
    lr, wd, eps = 1e-4, 0.05, 1e-8
    sd0 = O.procedural_state_dict(cfg)
    upd_self, upd_sign = (0.0, ""), (0.0, "")
    for n in names:
        p0, p1 = sd0[n].double(), params[n].detach().double().cpu()
        dec = wd if n in opt.arena.names_decay else 0.0
        first = lambda g_: p0 * (1.0 - lr * dec) - lr * g_ / (g_.abs() + eps)
        upd_self = max(upd_self, (float((p1 - first(grads[n].double())).abs().max()) / lr, n))
        upd_sign = max(upd_sign, (float(((p1 - first(sdr[n].grad.double())).abs() > 0.5 * lr).double().mean()), n))
    meas.update(update_vs_own_gradient_in_lr=upd_self[0], update_vs_own_gradient_name=upd_self[1], update_sign_mismatch_share=upd_sign[0],
                update_sign_mismatch_name=upd_sign[1])
    del sdr
    if check_eager:
        model_e, opt_e, ts_e = _run_engine(cfg, B, Kk, x, noise, use_graph=False)
        meas["eager_equals_graph"] = bool(float(ts_e.loss) == loss and torch.equal(opt_e.arena.flat_g, flat_g)
                                          and torch.equal(opt_e.arena.flat_p, flat_p))
    T.report(tag, **meas)
    # masks the engine drew from the noise == the oracle's (bit-exact integer path at full size)
    tmg, ikg, irg = model.generate_random_masks_from_noise(*noise, Kk, list(doms))
    meas["masks_bit_exact"] = bool(torch.equal(ikg.cpu(), ids_keep) and torch.equal(irg.cpu(), ids_restore))
    if not asserts:
        return meas
    assert meas["masks_bit_exact"]
    assert meas["loss_rel"] <= 1e-3, meas                      # BASELINE.json north_star: the masked-reconstruction loss within 1e-3
    assert meas["loss_rel_w16"] <= 2e-4, meas                  # ... and within 2e-4 once the oracle multiplies with the same bf16 weight matrices
    for d in task:
        # per task: 4e-4 against the oracle with bf16 weight matrices (measured <= 7.7e-5 at C2, <= 1.8e-4 at C4 over 9 data x mask seeds);
        # 1.5e-3 against the fp32 oracle -- the weight format alone moves a task by up to 9.5e-4 (C2 slo) / 8.8e-4 (C4 bscan), the same for every
        # batch, and the reference's own autocast run sits 1.9e-3 from its fp32 run on C4's bscan (tests/golden/c4_large.npz floor.loss.bscan)
        assert meas[f"loss_rel_w16.{d}"] <= 4e-4, (d, meas)
        assert meas[f"loss_rel.{d}"] <= 1.5e-3, (d, meas)
    assert meas["grad_norm_rel"] <= 1e-2, meas
    assert worst_n[0] <= 2e-2, worst_n
    assert worst_r[0] <= 2e-2, worst_r            # 2e-2 (measured 8-9e-3)
    assert upd_self[0] <= 2e-2, upd_self          # the fused / block-wise updates ARE AdamW of the engine's gradient (within fp32 roundoff of p)
    assert upd_sign[0] <= 5e-2, upd_sign          # and lr * sign(g) agrees with the oracle's step on >= 95 % of every sampled tensor
    if check_eager:
        assert meas["eager_equals_graph"]
    return meas


@pytest.mark.parametrize("kind,doms,B,Kk,chains", [("base", ("bscan", "slo"), 64, 128, 1), ("base", ("bscan", "slo"), 64, 128, 2),
                                                     ("large", ("bscan", "slo", "bscanlayermap"), 16, 98, 1),
                                                     ("large", ("bscan", "slo", "bscanlayermap"), 16, 98, 2)])
def test_poisoned_workspaces_leave_every_gradient_unchanged(kind, doms, B, Kk, chains, monkeypatch):
    """Every float buffer the step allocates (activations, split-K slabs, LayerNorm / column-sum / attention partial rows, stream-K
    scratch) and the cached GEMM workspaces are filled with NaN before a kernel sees them: a reduction that sums a partial row nobody
    wrote, or a kernel that reads a workspace it did not fill, turns a gradient into NaN (the unexplained NaN of the
    reverted two-chain backward went through this shared machinery -- kernels.Deferred, reduce_batch, the colsum / LayerNorm
    partials).  The poisoned eager step must reproduce the clean step bit for bit, with one and with two forward chains."""
    from mirage_amd import builder, optim, engine as E, kernels as K
    from mirage_amd.trainer import TrainStep
    from oracle import mirage_oracle as O
    import test_model_gpu as T
    cfg = O.make_config(kind, doms)
    if kind == "large":
        cfg = O.make_config(kind, doms, depth=6)              # the shapes of C4 at B = 16; six blocks keep the four cases short
    x = {d: t.to(DEV) for d, t in O.synthetic_batch(cfg, B, seed=1234).items()}
    (p, u, g), _, _, _ = O.synthetic_masks(cfg, B, Kk, seed=99)
    noise = (torch.from_numpy(p).to(DEV), torch.cat([torch.from_numpy(a) for a in u], 1).to(DEV), torch.from_numpy(g).to(DEV))
    monkeypatch.setattr(E, "TWO_CHAINS", 2 if chains == 2 else 0)

    def run(poison):
        model, args, _ = T._build(cfg)
        model.train()
        opt = optim.FusedAdamW(model, [1e-4] * 8, [0.05] * 8, betas=(0.9, 0.95), eps=1e-8)
        ts = TrainStep(model, builder.get_losses(args), opt, x, Kk, use_graph=False, feed_mask_noise=True)
        ts.load(x)
        for dst, src in zip(ts.static_noise, noise):
            dst.copy_(src)
        ts._fwd_bwd(); E.join_side_stream()                   # first pass: autotuning, allocator warm-up
        torch.cuda.synchronize()
        real_empty, real_like = torch.empty, torch.empty_like
        if poison:
            def empty(*a, **kw):
                t = real_empty(*a, **kw)
                if t.is_cuda and t.is_floating_point():
                    t.fill_(float("nan"))
                return t

            def empty_like(*a, **kw):
                t = real_like(*a, **kw)
                if t.is_cuda and t.is_floating_point():
                    t.fill_(float("nan"))
                return t
            monkeypatch.setattr(torch, "empty", empty)
            monkeypatch.setattr(torch, "empty_like", empty_like)
            for ws in K._ws_cache.values():
                ws.fill_(float("nan"))
        try:
            ts._fwd_bwd(); E.join_side_stream()
            torch.cuda.synchronize()
        finally:
            monkeypatch.setattr(torch, "empty", real_empty)
            monkeypatch.setattr(torch, "empty_like", real_like)
        return float(ts.loss), opt.arena.flat_g[:opt.arena.n_trainable].clone()
    clean_loss, clean = run(False)
    loss, grads = run(True)
    bad = int((~torch.isfinite(grads)).sum())
    T.report(f"poison_{kind}_b{B}_chains{chains}", nonfinite_gradient_elements=bad, loss_equal=bool(loss == clean_loss),
             gradients_bit_equal=bool(torch.equal(grads, clean)))
    assert bad == 0, f"{bad} non-finite gradient elements with poisoned workspaces"
    assert loss == clean_loss and torch.equal(grads, clean)


def test_c2_base_b64_graph_step_vs_oracle():
    _fullsize_case("c2_b64_trainstep", "base", ("bscan", "slo"), 64, 128)


def test_c3_base_b64_graph_step_vs_oracle():
    _fullsize_case("c3_b64_trainstep", "base", ("bscan", "slo", "bscanlayermap"), 64, 98)


def test_c4_large_b16_graph_step_vs_oracle():
    _fullsize_case("c4_b16_trainstep", "large", ("bscan", "slo", "bscanlayermap"), 16, 98, check_eager=False)


def test_c4_large_b64_production_step_vs_oracle_on_8_samples():
    _c4_b64_case()


def test_c4_large_b64_graph_step_gradients_and_update_vs_oracle():
    """C4 at the PRODUCTION batch with every gradient check of the C2 / C3 cases (round-5 review, weak #1): M = 64 x 99 = 6336 rows -- the dgrad
    variants of gemm_tune_gfx950.json at that M, the seven grouped weight-gradient launches of the Large model and AdamW in their epilogue --
    against the CPU oracle's forward + backward on all 64 samples (about a minute on 32 host threads, ~25 GB of host memory)."""
    _fullsize_case("c4_b64_trainstep", "large", ("bscan", "slo", "bscanlayermap"), 64, 98, check_eager=False)


def test_c2_b64_five_step_curve_through_the_graph_replayed_trainstep():
    """SURVEY.md section 8(d)'s loss-curve check on the PRODUCTION schedule (round-5 review, weak #2: the 20-step curve of test_train_gpu runs
    B = 2 through eager autograd): five optimizer steps of the graph-replayed TrainStep at C2, B = 64 -- one hipGraph per step, side streams,
    grouped weight gradients with AdamW in the epilogue, device-side lr table -- against oracle.train_steps (run_pretraining.py:683-742 restated
    on CPU fp32) on the same five batches and mask noises.  |dloss| <= 1e-3 max(1, |loss|) on steps 1-4, gradient norms
    within 2e-2 on all five.  Step 5 is asserted at 4e-3, and that is a property of the schedule, not of the engine: lr ramps to 4e-4 (16x the
    reference's peak for this batch, blr 1e-4 x 64 / 256), AdamW's first updates are lr * sign(g), and the loss goes 4.02 -> 4.30 -> 2.67 -> 1.77
    -> 1.39 -- down 21 % in the last step alone -- so the 5-8e-4 of steps 1-4 (the weight-format offset of DESIGN section 7) is amplified by
    the next update (measured 2.9e-3 = 1 % of that step's decrease; the gradient norm is still within 7e-3): profiles/r06_parity_measured.md."""
    from oracle import mirage_oracle as O
    from mirage_amd import builder, optim
    from mirage_amd.trainer import TrainStep
    import test_model_gpu as T
    cfg = O.make_config("base", ("bscan", "slo"))
    B, Kk, steps = 64, 128, 5
    lrs, wds = [1e-4, 2e-4, 3e-4, 4e-4, 4e-4], [0.05] * steps
    batches = [O.synthetic_batch(cfg, B, seed=700 + it) for it in range(steps)]
    noises, masks = [], []
    for it in range(steps):
        noise, _, ik, ir = O.synthetic_masks(cfg, B, Kk, seed=500 + it)
        noises.append(noise); masks.append((ik, ir))
    model, args, _ = T._build(cfg)
    model.train()
    opt = optim.FusedAdamW(model, lrs, wds, betas=(0.9, 0.95), eps=1e-8)
    dev_noise = lambda n_: (torch.from_numpy(n_[0]).to(DEV), torch.cat([torch.from_numpy(a) for a in n_[1]], 1).to(DEV), torch.from_numpy(n_[2]).to(DEV))
    x0 = {d: t.to(DEV) for d, t in O.synthetic_batch(cfg, B, seed=4321).items()}          # warm-up / capture on other data
    ts = TrainStep(model, builder.get_losses(args), opt, x0, Kk, use_graph=True, feed_mask_noise=True)
    ts.load(x0)
    for dst, src in zip(ts.static_noise, dev_noise(O.synthetic_masks(cfg, B, Kk, seed=77)[0])):
        dst.copy_(src)
    ts.warmup_and_capture(warm_iters=1)
    assert ts.g_fb is not None
    curve, norms = [], []
    for it in range(steps):
        ts.step({d: t.to(DEV) for d, t in batches[it].items()}, noise=dev_noise(noises[it]))
        torch.cuda.synchronize()
        curve.append(float(ts.loss)); norms.append(float(opt.grad_norm))
    assert opt.step_dev.tolist() == [steps, steps]
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    ref_curve, ref_norms, _ = O.train_steps(cfg, O.procedural_state_dict(cfg), batches, masks, lrs, wds)
    ref_curve, curve = np.asarray(ref_curve), np.asarray(curve)
    err = np.abs(curve - ref_curve) / np.maximum(1.0, np.abs(ref_curve))
    nerr = np.abs(np.asarray(norms) - np.asarray(ref_norms)) / np.asarray(ref_norms)
    T.report("c2_b64_graph_curve_5_steps", loss_err_per_step=[float(e) for e in err], grad_norm_err_per_step=[float(e) for e in nerr],
             curve=[float(c) for c in curve], ref_curve=[float(c) for c in ref_curve], tol=1e-3)
    assert abs(ref_curve[0] - ref_curve[-1]) > 0.02 * abs(ref_curve[0]), "the schedule must move the loss for this check to mean anything"
    assert err[:4].max() <= 1e-3 and err.max() <= 4e-3, (err.tolist(), curve.tolist(), ref_curve.tolist())
    assert nerr.max() <= 2e-2, nerr.tolist()


def _c4_b64_case(data_seed=1234, mask_seed=99, asserts=True):
    """C4 at the PRODUCTION per-GPU batch (Large, 3 modalities, B = 64, M = 64 x 99 = 6336 rows: the shapes gemm_tune_gfx950.json is
    keyed on, grouped weight gradients, early AdamW) through the graph-replayed TrainStep; samples are independent in the forward
    pass, so the CPU oracle runs on 8 of the 64 samples: their encoder tokens (rel-RMS <= 1e-2, cosine >= 0.9999) and their masked
    task losses computed from the engine's own predictions (<= 1e-3).  Reference: mirage/model.py:305-431, criterion.py:31-117."""
    from mirage_amd import builder
    from oracle import mirage_oracle as O
    import test_model_gpu as T
    doms, B, Kk = ("bscan", "slo", "bscanlayermap"), 64, 98
    cfg = O.make_config("large", doms)
    x = O.synthetic_batch(cfg, B, seed=data_seed)
    (p, u, g), tm, ids_keep, ids_restore = O.synthetic_masks(cfg, B, Kk, seed=mask_seed)
    noise = (torch.from_numpy(p).to(DEV), torch.cat([torch.from_numpy(a) for a in u], 1).to(DEV), torch.from_numpy(g).to(DEV))
    model, opt, ts = _run_engine(cfg, B, Kk, x, noise, use_graph=True)
    assert opt.step_dev.tolist() == [1, 1] and np.isfinite(float(ts.loss)) and np.isfinite(float(opt.grad_norm))
    idx = torch.arange(0, B, 8) + torch.arange(8) % 8                  # 0, 9, 18, ..., 63: both halves of the batch, every residue
    idx_d = idx.to(DEV)
    loss_fns = builder.get_losses(builder.make_args(doms, doms))
    xg = {d: t.to(DEV) for d, t in x.items()}
    sub_losses = {d: float(loss_fns[d](ts.preds[d][idx_d].contiguous(), xg[d][idx_d].contiguous(), mask=ts.masks[d][idx_d].contiguous()))
                  for d in ts.preds}
    # encoder tokens of the same weights BEFORE the step's update are not retrievable afterwards: the weights moved by one AdamW step
    # of lr 1e-4, so the token comparison uses a second model instance with the initial weights (decoders off, training-mode forward)
    model2, _, _ = T._build(cfg)
    model2.train()
    model2.output_adapters = None
    with torch.no_grad():
        enc, _ = model2({d: xg[d] for d in doms}, num_encoded_tokens=Kk, mask_noise=noise)
    enc = enc.float()[idx_d].cpu()
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    sd = O.procedural_state_dict(cfg)
    with torch.no_grad():
        xs = {d: t[idx] for d, t in x.items()}
        ref_total, ref_ls, ref_enc, _ = O.forward_loss(cfg, sd, xs, ids_keep[idx], ids_restore[idx])
    meas = {"B": B, "oracle_samples": 8, "enc_rel_rms": T.rel_rms(enc, ref_enc), "enc_rel_rms_tol": 1e-2,
            "enc_cos": float(torch.nn.functional.cosine_similarity(enc.flatten().double(), ref_enc.flatten().double(), dim=0))}
    for d in sub_losses:
        meas[f"loss_rel.{d}"] = abs(sub_losses[d] - float(ref_ls[d])) / max(1.0, abs(float(ref_ls[d])))
    meas["loss_rel"] = abs(sum(sub_losses.values()) - float(ref_total)) / abs(float(ref_total))
    with torch.no_grad():                 # the same against the oracle with the weight matrices as autocast multiplies with them (see _fullsize_case)
        w_total, w_ls, _, _ = O.forward_loss(cfg, O.autocast_weights(sd), xs, ids_keep[idx], ids_restore[idx])
    for d in sub_losses:
        meas[f"loss_rel_w16.{d}"] = abs(sub_losses[d] - float(w_ls[d])) / max(1.0, abs(float(w_ls[d])))
    meas["loss_rel_w16"] = abs(sum(sub_losses.values()) - float(w_total)) / abs(float(w_total))
    meas.update(data_seed=data_seed, mask_seed=mask_seed)
    T.report("c4_b64_production_8samples", **meas)
    meas["masks_bit_exact"] = all(torch.equal(ts.masks[d].cpu(), torch.from_numpy(tm[list(doms).index(d)])) for d in doms)
    if not asserts:
        return meas
    assert meas["masks_bit_exact"]                                                                # mask indices bit-exact at full size
    assert meas["enc_rel_rms"] <= 1e-2 and meas["enc_cos"] >= 0.9999, meas
    assert meas["loss_rel"] <= 1e-3 and meas["loss_rel_w16"] <= 2e-4, meas
    for d in sub_losses:                       # tolerances: see _fullsize_case
        assert meas[f"loss_rel_w16.{d}"] <= 4e-4, (d, meas)
        assert meas[f"loss_rel.{d}"] <= 1.5e-3, (d, meas)


def test_c5_large_b256_two_chain_graph_forward_vs_oracle_on_8_samples():
    """C5 at its real size: Large encoder-only forward over all 256 bscan tokens, B = 256 (M = 65792 rows: gemm_p8_kernel, two
    half-batch chains of 32896 rows on two streams) captured in a hipGraph exactly as bench.py runs it; the CPU oracle on 8 of
    the samples: encoder tokens rel-RMS <= 1e-2, cosine >= 0.9999.  Reference: mirage_wrapper.py:208-224, mirage/model.py:305-431."""
    from mirage_amd import builder, engine
    from oracle import mirage_oracle as O
    import test_model_gpu as T
    B = 256
    cfg = O.make_config("large", ("bscan",), out_domains=())
    model, args, sd = T._build(cfg, with_decoders=False)
    model.eval()
    x = O.synthetic_batch(cfg, B, seed=4321)["bscan"]
    static = {"bscan": x.to(DEV)}
    assert engine.TWO_CHAINS >= 1
    # mask_inputs=False keeps every token but still in the mask generator's (random) order (mirage/model.py:340-346,384-391): the
    # order is pinned by feeding the generator's noise, and the oracle gets the same ids
    (p, u, g), _, ids_keep, ids_restore = O.synthetic_masks(cfg, B, 256, seed=77)
    noise = (torch.from_numpy(p).to(DEV), torch.cat([torch.from_numpy(a) for a in u], 1).to(DEV), torch.from_numpy(g).to(DEV))

    def fwd():
        with torch.no_grad():
            return model(static, mask_inputs=False, mask_noise=noise)[0]

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        eager = fwd().float().clone()
        fwd()                                       # second eager pass: nothing left to tune, the chains overlap
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out_static = fwd()
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    out = out_static.float()
    assert out.shape == (B, 257, 1024)
    assert torch.equal(out, eager), float((out - eager).abs().max())       # graph replay == eager, bit for bit, all 256 samples
    idx = torch.arange(0, B, 32) + torch.arange(8) * 4 + 1                 # 1, 37, 73, ...: samples of both half-batch chains
    assert int(idx.max()) < B and int((idx >= B // 2).sum()) == 4
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    with torch.no_grad():
        ref, _ = O.forward(cfg, sd, {"bscan": x[idx]}, ids_keep[idx], ids_restore[idx], decode=False)
    got = out[idx.to(DEV)].cpu()
    meas = {"B": B, "oracle_samples": 8, "enc_rel_rms": T.rel_rms(got, ref), "enc_rel_rms_tol": 1e-2,
            "enc_cos": float(torch.nn.functional.cosine_similarity(got.flatten().double(), ref.flatten().double(), dim=0)),
            "worst_sample_rel_rms": max(T.rel_rms(got[i], ref[i]) for i in range(8))}
    T.report("c5_b256_two_chain_graph_8samples", **meas)
    assert meas["enc_rel_rms"] <= 1e-2 and meas["worst_sample_rel_rms"] <= 1.2e-2 and meas["enc_cos"] >= 0.9999, meas


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _wait_ranks(procs, timeout):
    """wait for every child; on a hang abort them all (faulthandler prints where each one stood) and fail with their output instead
    of leaving stuck ranks on the GPU behind a bare TimeoutExpired"""
    import signal
    import time
    deadline = time.time() + timeout
    logs = [None] * len(procs)
    try:
        for i, p in enumerate(procs):
            logs[i] = p.communicate(timeout=max(1.0, deadline - time.time()))[0]
    except subprocess.TimeoutExpired:
        for p in procs:
            if p.poll() is None:
                p.send_signal(signal.SIGABRT)
        time.sleep(2.0)
        tails = []
        for i, p in enumerate(procs):
            if p.poll() is None:
                p.kill()
            try:
                out = p.communicate(timeout=20)[0]
            except Exception:
                out = b""
            tails.append((logs[i] or out or b"").decode(errors="replace")[-2500:])
        raise AssertionError(f"ranks did not finish within {timeout} s; output tails: {tails}")
    return [(lg or b"").decode(errors="replace") if isinstance(lg, bytes) else (lg or "") for lg in logs]


def _spawn(rank, world, port, out, *extra):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "MRG_DP_COMM", "MRG_DP_DTYPE", "MRG_DP_UPDATE"):
        env.pop(k, None)
    return subprocess.Popen([sys.executable, "-X", "faulthandler", os.path.join(ROOT, "tests", "_dp_engine_worker.py"), str(rank), str(world),
                             str(port), out, *extra], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)


@pytest.mark.parametrize("comm,wire,update", [("allreduce", "fp32", "replicated"), ("rs_ag", "bf16", "replicated"), ("allreduce", "fp32", "sharded")])
def test_two_engine_ranks_on_one_gpu(tmp_path, comm, wire, update):
    """Two data-parallel ranks of the HIP engine (child processes sharing GPU 0, gloo) through trainer.TrainStep's staged schedule
    (three backward graphs with the gradient exchange in between + the optimizer graph, AdamW of each stage behind its exchange)
    == one process on the concatenated batch: same losses, gradients and weights after two optimizer steps."""
    import test_model_gpu as T
    port = _free_port()
    outs = [str(tmp_path / f"r{r}.pt") for r in range(2)]
    single = str(tmp_path / "single.pt")
    procs = [_spawn(r, 2, port, outs[r], comm, wire, update) for r in range(2)]
    logs = _wait_ranks(procs, 300)
    for p, lg in zip(procs, logs):
        assert p.returncode == 0, lg[-3000:]
    ps = _spawn(0, 1, port, single)
    (lg,) = _wait_ranks([ps], 300)
    assert ps.returncode == 0, lg[-3000:]
    r0, r1, s = (torch.load(f, weights_only=False) for f in (outs[0], outs[1], single))
    assert r0["n_graphs"] == 4 and r0["cuts"] == [1, 2] and s["n_graphs"] == 1          # upper / middle / lower / optimizer
    assert r0["steps"] == [2, 2] and s["steps"] == [2, 2]
    assert r0["update"] == update
    assert torch.equal(r0["flat_p"], r1["flat_p"]) and torch.equal(r0["shadow"], r1["shadow"])     # replicas stay identical
    mean_losses = [(a + b) / 2 for a, b in zip(r0["losses"], r1["losses"])]
    if update == "sharded":
        # MRG_DP_UPDATE=sharded (reduce-scatter -> AdamW on the owned shard -> all-gather of the parameters): a rank's gradient arena holds
        # the average only inside its shards, so the averaged gradient is each element's owner's: after gather_moments both ranks hold
        # the same moments, and exp_avg = (1 - beta1) (beta1 g_1 + g_2) stands in for the gradient in the comparison below
        assert torch.equal(r0["exp_avg"], r1["exp_avg"]) and torch.equal(r0["exp_avg_sq"], r1["exp_avg_sq"])
        assert torch.equal(r0["shadow"].float(), r0["flat_p"].to(torch.bfloat16).float())          # the re-cast covers the gathered shards
        g_rel = float((r0["exp_avg"].double() - s["exp_avg"].double()).norm() / s["exp_avg"].double().norm())
    else:
        assert torch.equal(r0["flat_g"], r1["flat_g"])
        g_rel = float((r0["flat_g"].double() - s["flat_g"].double()).norm() / s["flat_g"].double().norm())
    dw = float((r0["flat_p"] - s["flat_p"]).abs().max())
    l_rel = [abs(a - b) / abs(b) for a, b in zip(mean_losses, s["losses"])]
    # The two runs use different GEMM tilings / split-K factors (M = 4 x N vs 8 x N rows), so bf16 intermediates round differently:
    # ~1e-3 relative on the gradients (a missing or doubled stage reduction would be O(1)); the first loss is computed from
    # identical weights and must agree to fp32 roundoff, the second one follows an Adam step taken on those gradients.
    gtol = 5e-3 if wire == "fp32" else 2e-2
    gn_rel = abs(r0["grad_norm"] - s["grad_norm"]) / s["grad_norm"]
    T.report(f"dp2_one_gpu_{comm}_{wire}_{update}", grad_rel=g_rel, grad_tol=gtol, weight_max_abs=dw, loss_rel_step0=l_rel[0], loss_rel_step1=l_rel[1],
             grad_norm_rel=gn_rel)
    # the logged norm (mutils/native_scaler.py:46-61 over the AVERAGED gradients) is assembled from the partial sums of three stage
    # updates issued eagerly between the graphs and of the rest inside the optimizer graph: all of them must be in it
    assert gn_rel <= (1e-2 if wire == "fp32" else 2e-2), (r0["grad_norm"], s["grad_norm"])
    assert r0["grad_norm"] == r1["grad_norm"]
    assert l_rel[0] <= 1e-5, (mean_losses, s["losses"])
    assert l_rel[1] <= 2e-3, (mean_losses, s["losses"])
    assert g_rel <= gtol, g_rel
    # weights: two AdamW steps of at most lr each; a gradient sign flip on a near-zero entry moves a weight by <= 2 lr per step
    assert dw <= 4 * 1e-3, dw
    moved = float((s["flat_p"] - r0["flat_p"]).abs().mean())
    assert moved <= 1e-4, moved


def _rank_env(**extra):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MRG_SHARE_GPU="1", MRG_DIST_BACKEND="gloo", **extra)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "MRG_DP_COMM", "MRG_DP_DTYPE", "MRG_DP_UPDATE", "MRG_FORCE_DP"):
        if k not in extra:
            env.pop(k, None)
    return env


def test_bench_self_launch_two_ranks():
    """`python bench.py --gpus 2` without a launcher spawns its two ranks itself (bench.py::_self_launch) and relays rank 0's single
    JSON line.  Both ranks share GPU 0 over gloo here (1-GPU box); the line must say two ranks joined and carry the roofline object."""
    import json
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "8", "--steps", "3", "--warmup", "1"],
                       env=_rank_env(), capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout[-2000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["ranks_seen"] == 2 and r["steps"] == 3 and r["warmup"] == 1
    assert r["config"]["global_batch"] == 16 and r["config"]["parallelism"] == "dp2" and r["scaling"] == "weak"
    assert r["value"] > 0 and r["roofline"]["frac"] > 0 and "cpu_baseline" not in r
    assert np.isfinite(r["final_loss"])
    # per-rank detail for the first real multi-GPU run: every rank's own step time, the time its compute stream was
    # blocked on the gradient exchange, and the exchange format that produced the numbers
    dp = r["dp"]
    assert len(dp["ranks_ms_per_step"]) == 2 and all(t > 0 for t in dp["ranks_ms_per_step"])
    assert len(dp["exposed_comm_ms_per_step"]) == 2 and all(0 <= t < r["ms_per_step"] * 1.05 for t in dp["exposed_comm_ms_per_step"])
    assert max(dp["ranks_ms_per_step"]) <= r["ms_per_step"] * 1.05
    assert dp["comm"] in ("allreduce", "rs_ag") and dp["wire_dtype"] in ("fp32", "bf16") and dp["backend"] == "gloo"
    assert dp["update"] in ("replicated", "sharded")
    assert dp["bucket_mib"] == 64 and dp["buckets"] >= 1 and dp["bytes_per_step"] > 0 and isinstance(dp["encoder_cuts"], list)


def test_two_ranks_over_rccl_on_one_gpu_or_recorded_refusal():
    """World size 2 over the nccl (= RCCL) backend with both ranks on GPU 0 (the boxes of this pool have one GPU): either the
    collectives and GradReducer's two exchange formats work, or RCCL refuses two ranks per device -- then the refusal is recorded
    (gpurun_out/parity_measured.jsonl) and the gloo tests above remain the multi-rank coverage."""
    import json
    port = _free_port()
    procs = []
    for r in range(2):
        env = _rank_env(RANK=str(r), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.pop("MRG_DIST_BACKEND", None)
        env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "debug", "rccl_two_ranks_one_gpu.py")], env=env, cwd=ROOT,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=240) + (p.returncode,))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            outs.append(("", "timeout: a rank never joined the RCCL communicator", 3))
    rcs = [o[2] for o in outs]
    why = [next((l for l in o[1].splitlines() if "RCCL refused" in l), o[1].strip()[-300:])[:500] for o in outs]
    rec = {"test": "rccl_two_ranks_one_gpu", "exit_codes": rcs, "stderr": why}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "parity_measured.jsonl"), "a") as f:
        f.write(json.dumps(rec) + "\n")
    if all(rc == 0 for rc in rcs):
        return
    if all(rc in (0, 3) for rc in rcs):
        pytest.skip("RCCL refuses two ranks on one device on this pool: " + " | ".join(w for w, o in zip(why, outs) if o[2] == 3))
    raise AssertionError(outs)


def test_run_pretraining_two_ranks_two_epochs(tmp_path):
    """mirage_amd.run_pretraining under two data-parallel ranks (sharing GPU 0, gloo) with hipGraphs: 2 epochs x 40 steps of the staged
    schedule -- more than the ~40 steps after which the host-side partial-sum cursor used to run off its buffer -- and the logged
    gradient norm must be the norm of the whole (averaged) gradient, not of the last stage (round-2 verdict, weak #1)."""
    import json
    import test_model_gpu as T
    port = _free_port()
    out = str(tmp_path / "out")
    argv = ["-m", "mirage_amd.run_pretraining", "--blr", "1e-2", "--data_path", "synthetic", "--in_domains", "bscan-slo", "--out_domains", "bscan-slo",
            "--input_size", "128", "--patch_size", "8", "--batch_size", "2", "--num_encoded_tokens", "98", "--epochs", "2",
            "--steps_per_epoch", "40", "--warmup_epochs", "1", "--save_ckpt_freq", "1", "--base_output_dir", out, "--print_freq", "1"]
    procs = []
    for r in range(2):
        env = _rank_env(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable] + argv, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = _wait_ranks(procs, 900)
    for p, lg in zip(procs, logs):
        assert p.returncode == 0, lg[-3000:]
    d = os.path.join(out, "miragepre_base")
    lines = [json.loads(l) for l in open(os.path.join(d, "log.txt"))]
    assert [l["epoch"] for l in lines] == [0, 1]
    # per-step log lines of rank 0: 80 steps, finite, the loss falls, and the norm is that of a full Base model's gradient
    steps = [l for l in logs[0].splitlines() if l.startswith("Epoch: [")]
    assert len(steps) == 80, len(steps)
    vals = [dict(zip(l.split()[3::2], l.split()[4::2])) for l in steps]
    losses = [float(v["loss:"]) for v in vals]; norms = [float(v["grad_norm:"]) for v in vals]
    assert all(np.isfinite(losses)) and all(np.isfinite(norms)) and min(norms) > 0
    assert np.mean(losses[-10:]) < np.mean(losses[:10])
    ck = torch.load(os.path.join(d, "checkpoint-1.pth"), map_location="cpu", weights_only=False)
    assert ck["optimizer"]["mrg_iteration"] == 80
    T.report("run_pretraining_dp2", loss_first=losses[0], loss_last=losses[-1], grad_norm_first=norms[0], grad_norm_last=norms[-1])


def test_run_pretraining_sharded_update_matches_replicated(tmp_path):
    """mirage_amd.run_pretraining under two ranks with MRG_DP_UPDATE=sharded (reduce-scatter -> AdamW on the owned shard -> all-gather of the
    parameters) against the replicated update: the same loss log line for line, the gradient norm to fp32 roundoff (the element updates are the same
    arithmetic on the same averaged gradients), and a checkpoint whose optimizer state is complete -- the moments of the OTHER rank's
    shards reach rank 0 through GradReducer.gather_moments before the file is written."""
    def run(update, out):
        port = _free_port()
        argv = ["-m", "mirage_amd.run_pretraining", "--blr", "1e-2", "--data_path", "synthetic", "--in_domains", "bscan-slo", "--out_domains", "bscan-slo",
                "--input_size", "128", "--patch_size", "8", "--batch_size", "2", "--num_encoded_tokens", "98", "--epochs", "1",
                "--steps_per_epoch", "12", "--warmup_epochs", "1", "--save_ckpt_freq", "1", "--base_output_dir", out, "--print_freq", "1"]
        procs = []
        for r in range(2):
            # MRG_TUNE=0: the shapes of this small model are not in the tuning table, and first-use autotuning is a timing -- two separate runs
            # may pick different kernels (different rounding); the heuristic choice is the same in every process
            env = _rank_env(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MRG_DP_UPDATE=update,
                            MRG_TUNE="0")
            procs.append(subprocess.Popen([sys.executable] + argv, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
        logs = _wait_ranks(procs, 600)
        for p, lg in zip(procs, logs):
            assert p.returncode == 0, lg[-3000:]
        steps = [l for l in logs[0].splitlines() if l.startswith("Epoch: [")]
        vals = [dict(zip(l.split()[3::2], l.split()[4::2])) for l in steps]
        ck = torch.load(os.path.join(out, "miragepre_base", "checkpoint-0.pth"), map_location="cpu", weights_only=False)
        return [(v["loss:"], v["grad_norm:"]) for v in vals], ck
    rep, ck_rep = run("replicated", str(tmp_path / "rep"))
    sh, ck_sh = run("sharded", str(tmp_path / "sh"))
    assert len(rep) == len(sh) == 12
    # the losses are the same computation; the logged norm is assembled from the per-rank partial sums in another order (root of the
    # all-reduced sums instead of one sum over the arena): equal to fp32 roundoff
    assert [a for a, _ in rep] == [a for a, _ in sh], list(zip(rep, sh))
    assert all(abs(float(a) - float(b)) <= 2e-5 * float(a) + 2e-4 for (_, a), (_, b) in zip(rep, sh)), list(zip(rep, sh))
    st_r, st_s = ck_rep["optimizer"]["state"], ck_sh["optimizer"]["state"]
    assert st_r.keys() == st_s.keys() and len(st_s) > 100
    for i in st_s:
        assert torch.equal(st_s[i]["exp_avg"], st_r[i]["exp_avg"]) and torch.equal(st_s[i]["exp_avg_sq"], st_r[i]["exp_avg_sq"]), i
    for k in ck_rep["model"]:
        assert torch.equal(ck_rep["model"][k], ck_sh["model"][k]), k
