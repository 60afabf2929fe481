"""Parity THROUGH THE PRODUCTION PATH at the BASELINE sizes (VERDICT r01 "what's weak" #1): the graph-replayed TrainStep -- side
streams, deferred split-K / column-sum reductions, XCD-banded split-K, epilogue column sums, early block-wise AdamW, the tuned
GEMM variants of gemm_tune_gfx950.json -- at per-GPU batch 64 (C2, C3) / 16 (C4) against the CPU oracle on the same batch and
the same mask noise; eager == graph bit-exactly at these shapes; and two data-parallel engine ranks sharing one GPU (gloo)
against one process on the concatenated batch.

Reference: mirage/model.py:305-431 (forward), mirage/criterion.py:31-117 (losses), run_pretraining.py:683-742 (step).
Tolerances: total and per-task loss <= 1e-3 relative (BASELINE.json north_star); global gradient norm <= 1e-2; sampled
parameter gradients: norm within 2e-2, full-tensor rel-RMS within 4e-2 of the fp32 oracle (bf16 operands).  Every measured
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
    ts.load(xg)
    for dst, src in zip(ts.static_noise, noise):
        dst.copy_(src)
    if use_graph:
        ts.warmup_and_capture(warm_iters=1)
        assert ts.g_fb is not None and ts._early is not None        # one graph, early block-wise AdamW on (the bench configuration)
    ts.step(xg, noise=noise)
    torch.cuda.synchronize()
    return model, opt, ts


def _fullsize_case(tag, kind, doms, B, Kk, check_eager=True):
    from oracle import mirage_oracle as O
    import test_model_gpu as T
    cfg = O.make_config(kind, doms)
    x = O.synthetic_batch(cfg, B, seed=1234)
    (p, u, g), tm, ids_keep, ids_restore = O.synthetic_masks(cfg, B, Kk, seed=99)
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
    worst_n, worst_r = (0.0, ""), (0.0, "")
    for n in names:
        r = sdr[n].grad.double()
        mine = grads[n].double()
        scale = max(float(r.norm()), 1e-3 * ref_gn)
        en = abs(float(mine.norm()) - float(r.norm())) / scale
        er = float((mine - r).norm()) / scale
        worst_n = max(worst_n, (en, n)); worst_r = max(worst_r, (er, n))
    meas.update(grad_norm_worst=worst_n[0], grad_norm_worst_name=worst_n[1], grad_norm_worst_tol=2e-2,
                grad_rms_worst=worst_r[0], grad_rms_worst_name=worst_r[1], grad_rms_worst_tol=4e-2)
    del sdr
    if check_eager:
        model_e, opt_e, ts_e = _run_engine(cfg, B, Kk, x, noise, use_graph=False)
        meas["eager_equals_graph"] = bool(float(ts_e.loss) == loss and torch.equal(opt_e.arena.flat_g, flat_g)
                                          and torch.equal(opt_e.arena.flat_p, flat_p))
    T.report(tag, **meas)
    # masks the engine drew from the noise == the oracle's (bit-exact integer path at full size)
    tmg, ikg, irg = model.generate_random_masks_from_noise(*noise, Kk, list(doms))
    assert torch.equal(ikg.cpu(), ids_keep) and torch.equal(irg.cpu(), ids_restore)
    assert meas["loss_rel"] <= 1e-3, meas
    for d in task:
        assert meas[f"loss_rel.{d}"] <= 1e-3, (d, meas)
    assert meas["grad_norm_rel"] <= 1e-2, meas
    assert worst_n[0] <= 2e-2, worst_n
    assert worst_r[0] <= 4e-2, worst_r
    if check_eager:
        assert meas["eager_equals_graph"]


def test_c2_base_b64_graph_step_vs_oracle():
    _fullsize_case("c2_b64_trainstep", "base", ("bscan", "slo"), 64, 128)


def test_c3_base_b64_graph_step_vs_oracle():
    _fullsize_case("c3_b64_trainstep", "base", ("bscan", "slo", "bscanlayermap"), 64, 98)


def test_c4_large_b16_graph_step_vs_oracle():
    _fullsize_case("c4_b16_trainstep", "large", ("bscan", "slo", "bscanlayermap"), 16, 98, check_eager=False)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _spawn(rank, world, port, out, *extra):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "MRG_DP_COMM", "MRG_DP_DTYPE"):
        env.pop(k, None)
    return subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_dp_engine_worker.py"), str(rank), str(world), str(port), out, *extra],
                            env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)


@pytest.mark.parametrize("comm,wire", [("allreduce", "fp32"), ("rs_ag", "bf16")])
def test_two_engine_ranks_on_one_gpu(tmp_path, comm, wire):
    """Two data-parallel ranks of the HIP engine (child processes sharing GPU 0, gloo) through trainer.TrainStep's staged schedule
    (three backward graphs with the gradient exchange in between + the optimizer graph, AdamW of each stage behind its exchange)
    == one process on the concatenated batch: same losses, gradients and weights after two optimizer steps."""
    import test_model_gpu as T
    port = _free_port()
    outs = [str(tmp_path / f"r{r}.pt") for r in range(2)]
    single = str(tmp_path / "single.pt")
    procs = [_spawn(r, 2, port, outs[r], comm, wire) for r in range(2)]
    logs = [p.communicate(timeout=600)[0].decode() for p in procs]
    for p, lg in zip(procs, logs):
        assert p.returncode == 0, lg[-3000:]
    ps = _spawn(0, 1, port, single)
    lg = ps.communicate(timeout=600)[0].decode()
    assert ps.returncode == 0, lg[-3000:]
    r0, r1, s = (torch.load(f, weights_only=False) for f in (outs[0], outs[1], single))
    assert r0["n_graphs"] == 4 and r0["cuts"] == [1, 2] and s["n_graphs"] == 1          # upper / middle / lower / optimizer
    assert r0["steps"] == [2, 2] and s["steps"] == [2, 2]
    assert torch.equal(r0["flat_p"], r1["flat_p"]) and torch.equal(r0["flat_g"], r1["flat_g"])     # replicas stay identical
    mean_losses = [(a + b) / 2 for a, b in zip(r0["losses"], r1["losses"])]
    g_rel = float((r0["flat_g"].double() - s["flat_g"].double()).norm() / s["flat_g"].double().norm())
    dw = float((r0["flat_p"] - s["flat_p"]).abs().max())
    l_rel = [abs(a - b) / abs(b) for a, b in zip(mean_losses, s["losses"])]
    # The two runs use different GEMM tilings / split-K factors (M = 4 x N vs 8 x N rows), so bf16 intermediates round differently:
    # ~1e-3 relative on the gradients (a missing or doubled stage reduction would be O(1)); the first loss is computed from
    # identical weights and must agree to fp32 roundoff, the second one follows an Adam step taken on those gradients.
    gtol = 5e-3 if wire == "fp32" else 2e-2
    T.report(f"dp2_one_gpu_{comm}_{wire}", grad_rel=g_rel, grad_tol=gtol, weight_max_abs=dw, loss_rel_step0=l_rel[0], loss_rel_step1=l_rel[1],
             grad_norm_rel=abs(r0["grad_norm"] - s["grad_norm"]) / s["grad_norm"])
    assert l_rel[0] <= 1e-5, (mean_losses, s["losses"])
    assert l_rel[1] <= 2e-3, (mean_losses, s["losses"])
    assert g_rel <= gtol, g_rel
    # weights: two AdamW steps of at most lr each; a gradient sign flip on a near-zero entry moves a weight by <= 2 lr per step
    assert dw <= 4 * 1e-3, dw
    moved = float((s["flat_p"] - r0["flat_p"]).abs().mean())
    assert moved <= 1e-4, moved
