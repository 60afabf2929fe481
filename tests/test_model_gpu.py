"""End-to-end parity of the HIP path against the CPU oracle and the reference-generated golden vectors. GPU only.

Tolerances (bf16 compute vs the fp32 reference):
  * mask indices / ids_keep / ids_restore: bit-exact;
  * loss (per task and summed): relative error <= 1e-3 (BASELINE.json north_star) -- except where the REFERENCE ITSELF,
    run under bf16 autocast on the same weights / inputs / masks, is not within 1e-3/3 of its own fp32 run: every fixture
    carries that measured noise floor (`floor.*`, tools/ref_bf16_floor.py) and the bound is max(1e-3, 3 x floor).
    (These fixtures use untrained hash-derived weights; rounding the weights to bf16 is a fixed perturbation of the
    model, so the floor is larger than for a trained model.  Measured floors: per-task 0.8e-4 .. 1.9e-3.)
  * encoder tokens: rel-RMS <= 1e-2 (floor 3.9e-3 .. 5.6e-3), cosine >= 0.9999;
  * gradients: every parameter within max(4e-2, 3 x floor) of the whole-gradient scale, total norm within
    max(1e-2, 3 x floor).
"""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _build(cfg, with_decoders=True):
    from mirage_amd import builder
    from oracle import mirage_oracle as O
    dom = cfg["domains"]
    img = next((d for d in cfg["in_domains"] if dom[d]["kind"] == "image"), None)
    lm = next((d for d in dom if dom[d]["kind"] == "semseg"), None)
    args = builder.make_args(cfg["in_domains"], cfg["out_domains"], input_size=dom[img]["image"] if img else 512,
                             patch_size=dom[img]["patch"] if img else 32, layermap_size=dom[lm]["image"] if lm else 128,
                             layermap_patch=dom[lm]["patch"] if lm else 8, decoder_dim=cfg["decoder_dim"],
                             decoder_depth=cfg["decoder_depth"], decoder_num_heads=cfg["decoder_heads"],
                             drop_path=cfg.get("drop_path_rate", 0.0))
    extra = {"drop_rate": cfg["drop_rate"]} if cfg.get("drop_rate", 0.0) > 0.0 else {}
    if cfg.get("attn_drop_rate", 0.0) > 0.0:
        extra["attn_drop_rate"] = cfg["attn_drop_rate"]
    model = builder.get_model(args, with_decoders=with_decoders and bool(cfg["out_domains"]), dim_tokens=cfg["dim"], depth=cfg["depth"],
                              num_heads=cfg["heads"], decoder_attn_drop_rate=cfg.get("decoder_attn_drop_rate", 0.0),
                              learnable_pos_emb=bool(cfg.get("learnable_pos_emb", False)), **extra)
    sd = O.procedural_state_dict(cfg)
    missing = model.load_state_dict(sd, strict=True)
    return model.to(DEV), args, sd


def report(case, **vals):
    """Measured errors next to their tolerances: printed (pytest -s / failure output) and appended to gpurun_out/parity_measured.jsonl
    so the slack actually used is on record (profiles/r02_parity_measured.md is made from it)."""
    import json, os
    line = json.dumps({"case": case, **{k: (round(v, 8) if isinstance(v, float) else v) for k, v in vals.items()}})
    print("PARITY " + line)
    try:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
        with open(os.path.join(root, "gpurun_out", "parity_measured.jsonl"), "a") as f:
            f.write(line + "\n")
    except OSError:
        pass


def _sub(t, n):
    f = t.detach().reshape(-1)
    step = max(1, f.numel() // n)
    return f[::step][:n]


def rel_rms(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _case(name, grads=True):
    from mirage_amd import builder
    from oracle import mirage_oracle as O
    fx, meta = load_golden(name)
    cfg, B, Kk = meta["cfg"], meta["B"], meta["K"]
    model, args, sd = _build(cfg)
    model.train()
    x = O.synthetic_batch(cfg, B, seed=meta["data_seed"])
    xg = {d: t.to(DEV) for d, t in x.items()}
    doms = cfg["in_domains"]
    p = torch.from_numpy(fx["p"]).to(DEV)
    u = torch.cat([torch.from_numpy(fx[f"u.{d}"]) for d in doms], 1).to(DEV)
    g = torch.from_numpy(fx["g"]).to(DEV)
    # masks: bit-exact from the committed noise, through the model's own generator
    tm, ids_keep, ids_restore = model.generate_random_masks_from_noise(p, u, g, Kk, doms)
    assert torch.equal(ids_keep.cpu(), torch.from_numpy(fx["ids_keep"]))
    assert torch.equal(ids_restore.cpu(), torch.from_numpy(fx["ids_restore"]))
    for d in doms:
        assert torch.equal(tm[d].cpu(), torch.from_numpy(fx[f"task_mask.{d}"]))
    # encoder tokens (decoders switched off) vs golden
    oa = model.output_adapters
    model.output_adapters = None
    with torch.no_grad():
        enc, _ = model({d: xg[d] for d in doms}, num_encoded_tokens=Kk, mask_noise=(p, u, g))
    model.output_adapters = oa
    enc = enc.float().cpu()
    # element-wise against the reference's tokens: the full tensor where the fixture holds it, else its 65536-element subsample
    got, ref = (enc, torch.from_numpy(fx["enc_tokens"])) if "enc_tokens" in fx else (_sub(enc, 65536), torch.from_numpy(fx["enc_tokens_sub"]))
    enc_rms = rel_rms(got, ref)
    cos = float(torch.nn.functional.cosine_similarity(got.flatten().double(), ref.flatten().double(), dim=0))
    meas = {"enc_rel_rms": enc_rms, "enc_rel_rms_tol": 1e-2, "enc_cos": cos}
    assert enc_rms <= 1e-2, enc_rms
    assert cos >= 0.9999, cos
    assert abs(float(enc.double().norm()) - float(fx["enc_tokens_norm"])) <= 5e-3 * float(fx["enc_tokens_norm"])
    # full forward + masked losses
    loss_fns = builder.get_losses(args)
    preds, masks = model({d: xg[d] for d in doms}, num_encoded_tokens=Kk, mask_noise=(p, u, g))
    losses = {d: loss_fns[d](preds[d], xg[d], mask=masks.get(d)) for d in preds}
    total = sum(losses.values())
    for d in preds:
        want = float(fx[f"loss.{d}"])
        tol = max(1e-3, 3.5 * float(fx[f"floor.loss.{d}"]))       # measured <= 2.9 x the reference's own bf16 deviation (tiny2 bscan, B = 2)
        meas[f"loss_rel.{d}"] = abs(float(losses[d]) - want) / max(1.0, abs(want)); meas[f"loss_tol.{d}"] = tol
        assert abs(float(losses[d]) - want) <= tol * max(1.0, abs(want)), (d, float(losses[d]), want, tol)
        assert abs(float(preds[d].float().double().norm()) - float(fx[f"pred_norm.{d}"])) <= 1e-2 * float(fx[f"pred_norm.{d}"])
    tol = max(1e-3, 3.5 * float(fx["floor.loss"]))
    meas["loss_rel"] = abs(float(total) - float(fx["loss"])) / abs(float(fx["loss"])); meas["loss_tol"] = tol
    assert abs(float(total) - float(fx["loss"])) <= tol * abs(float(fx["loss"])), (float(total), float(fx["loss"]), tol)
    if not grads:
        report(name, **meas)
        return
    total.backward()
    names = [str(n) for n in fx["grad_names"]]
    params = dict(model.named_parameters())
    tot2 = 0.0
    worst = (0.0, "")
    worst_t = (0.0, "")
    # ~3x the errors measured on MI355X (profiles/r02_parity_measured.md: per-parameter gradient norms within 9.1e-3 on the tiny
    # models and 4.0e-3 at full size, whole tensors within 9.4e-3 rel-RMS, total norm within 1.7e-3): a regression of that factor
    # fails.  (Round 2 used max(4e-2, 3 x the reference's own bf16 deviation) = up to 0.147, which a 30x regression passed.)
    gtol, gtol_t = 1.5e-2, 3e-2
    for n, gn in zip(names, fx["grad_norms"]):
        gr = params[n].grad
        assert gr is not None, n
        mine = float(gr.double().norm())
        tot2 += mine * mine
        # parameters whose true gradient is ~0 (softmax-shift-invariant k-bias) carry only noise: compare on the scale
        # of the whole gradient instead of their own
        scale = max(gn, 1e-3 * float(fx["grad_total_norm"]))
        err = abs(mine - gn) / scale
        if err > worst[0]:
            worst = (err, n)
        if f"grad.{n}" in fx:
            ref = torch.from_numpy(fx[f"grad.{n}"]).double()
            et = float((gr.cpu().double() - ref).norm()) / (scale + 1e-30)
            if et > worst_t[0]:
                worst_t = (et, n)
    ttol = 5e-3
    meas.update(grad_worst_rel=worst[0], grad_worst_name=worst[1], grad_worst_tol=gtol,
                grad_tensor_worst_rel=worst_t[0], grad_tensor_worst_name=worst_t[1], grad_tensor_worst_tol=gtol_t,
                grad_total_rel=abs(tot2 ** 0.5 - float(fx["grad_total_norm"])) / float(fx["grad_total_norm"]), grad_total_tol=ttol)
    report(name, **meas)
    assert worst[0] <= gtol, worst
    assert worst_t[0] <= gtol_t, worst_t
    assert abs(tot2 ** 0.5 - float(fx["grad_total_norm"])) <= ttol * float(fx["grad_total_norm"])


def test_tiny3_end_to_end():
    _case("tiny3")


def test_tiny2_end_to_end():
    _case("tiny2")


def test_learnable_positional_embeddings_end_to_end():
    """learnable_pos_emb=True on the input adapters (mirage/input_adapters.py:34,72 / :140,185): the tables are trainable parameters; the
    fixture is the reference's forward + losses + backward with them (tests/golden/tiny3_learnpos.npz), so `grad_names` holds the three
    pos_emb gradients (mrg_posemb_grad: per-token sums over the samples that kept the token).  An optimizer step then moves the tables and
    the next forward reads the moved values."""
    from mirage_amd import builder, optim
    from oracle import mirage_oracle as O
    fx, meta = load_golden("tiny3_learnpos")
    assert sum("pos_emb" in str(n) for n in fx["grad_names"]) == 3
    _case("tiny3_learnpos")
    cfg, B, Kk = meta["cfg"], meta["B"], meta["K"]
    model, args, _ = _build(cfg)
    model.train()
    x = {d: t.to(DEV) for d, t in O.synthetic_batch(cfg, B, seed=meta["data_seed"]).items()}
    doms = cfg["in_domains"]
    noise = (torch.from_numpy(fx["p"]).to(DEV), torch.cat([torch.from_numpy(fx[f"u.{d}"]) for d in doms], 1).to(DEV), torch.from_numpy(fx["g"]).to(DEV))
    opt = optim.FusedAdamW(model, [1e-2] * 4, [0.0] * 4)
    loss_fns = builder.get_losses(args)
    pe0 = {d: model.input_adapters[d].pos_emb.detach().clone() for d in doms}
    def loss_of():
        preds, masks = model({d: x[d] for d in doms}, num_encoded_tokens=Kk, mask_noise=noise)
        return sum(loss_fns[d](preds[d], x[d], mask=masks.get(d)) for d in preds)
    l0 = loss_of()
    opt.zero_grad(); l0.backward(); opt.step()
    for d in doms:
        pe = model.input_adapters[d].pos_emb
        assert pe.requires_grad and float((pe.detach() - pe0[d]).abs().max()) > 1e-3, d       # lr 1e-2: every touched entry moves by ~lr
    from mirage_amd import engine as E
    for d in doms:                         # the next forward reads the moved table (the fused optimizer updates the arena behind the tensor's back)
        ad = model.input_adapters[d]
        D, (h, wd) = ad.pos_emb.shape[1], ad.pos_emb.shape[-2:]
        assert torch.equal(E._pos_tokens(ad), ad.pos_emb.detach()[0].permute(1, 2, 0).reshape(h * wd, D).float())
    with torch.no_grad():
        assert torch.isfinite(loss_of())
    # the stand-alone adapter forward (every token of every sample) differentiates the table too: d pos_emb = sum over the batch of dy
    ad = model.input_adapters[doms[0]]
    img = x[doms[0]]
    model.zero_grad(set_to_none=False)
    y = ad(img)
    w = torch.randn_like(y)
    (y * w).sum().backward()
    D, (h, wd) = y.shape[-1], ad.pos_emb.shape[-2:]
    want = w.sum(0).t().reshape(1, D, h, wd)
    from mirage_amd.arena import grad_buffer
    torch.testing.assert_close(grad_buffer(ad.pos_emb).view(1, D, h, wd), want, rtol=1e-5, atol=1e-5)


def test_c2_base_end_to_end():
    _case("c2_base")


@pytest.mark.parametrize("name", ["tiny3", "c2_base"])
def test_training_forward_as_two_half_batch_chains(monkeypatch, name):
    # MRG_TWO_CHAINS=2: the training forward as two half-batch chains on two streams (engine._encoder_fwd_two_chains); the saved
    # activations are full-batch tensors assembled from the halves, so losses AND gradients must meet the same golden bounds
    from mirage_amd import engine
    monkeypatch.setattr(engine, "TWO_CHAINS", 2)
    _case(name)


def test_c3_base_end_to_end():
    _case("c3_base")


def test_c4_large_end_to_end():
    _case("c4_large")


def test_init_matches_reference_distribution_and_state_dict_layout():
    from oracle import mirage_oracle as O
    fx, meta = load_golden("harness_tiny")
    cfg = meta["cfg"]
    torch.manual_seed(0)
    model, args, _ = _build(cfg)
    keys = list(model.state_dict().keys())
    assert keys == [str(k) for k in fx["state_dict_keys"]]
    import json
    for k, s in zip(keys, fx["state_dict_shapes"]):
        assert list(model.state_dict()[k].shape) == json.loads(str(s))
    assert sorted(model.no_weight_decay()) == [str(s) for s in fx["no_weight_decay"]]
    assert sorted(n for n, p in model.named_parameters() if p.requires_grad) == sorted(str(s) for s in fx["trainable"])


def test_encoder_only_pooled_features():
    """Config-5 style path (mirage_wrapper.py:208-224): encoder over all tokens in random order -> LN -> mean pool."""
    from oracle import mirage_oracle as O
    for name in ("tiny_cls", "c5_large_cls"):
        fx, meta = load_golden(name)
        cfg, B = meta["cfg"], meta["B"]
        model, args, sd = _build(cfg, with_decoders=False)
        model.eval()
        d = cfg["in_domains"][0]
        x = O.synthetic_batch(cfg, B, seed=meta["data_seed"])[d].to(DEV)
        D = cfg["dim"]
        nw = torch.from_numpy(1.0 + 0.2 * O.procedural_uniform("cls.norm.weight", D)).to(DEV)
        nb = torch.from_numpy(0.04 * O.procedural_uniform("cls.norm.bias", D)).to(DEV)
        with torch.no_grad():
            out, _ = model({d: x}, mask_inputs=False)
            out = torch.nn.functional.layer_norm(out, (D,), nw, nb, 1e-6)
            pooled = out[:, :-1].mean(1)
        assert rel_rms(pooled.cpu(), torch.from_numpy(fx["pooled"])) <= 1e-2


def test_drop_path_training_parity():
    """A16: stochastic depth fused into the proj / fc2 epilogues (forward) and the scaled-gradient path (backward), fed the
    reference's own uniform draws (tests/golden/tiny3_droppath.npz; mirage/utils.py:103-134,259-262)."""
    from mirage_amd import builder
    fx, meta = load_golden("tiny3_droppath")
    cfg, B, Kk = meta["cfg"], meta["B"], meta["K"]
    model, args, sd = _build(cfg)
    model.train()
    assert [float(b.drop_path_prob) for b in model.encoder][0] == 0.0 and model.encoder[-1].drop_path_prob == pytest.approx(0.4)
    from oracle import mirage_oracle as O
    x = {d: t.to(DEV) for d, t in O.synthetic_batch(cfg, B, seed=meta["data_seed"]).items()}
    doms = cfg["in_domains"]
    noise = (torch.from_numpy(fx["p"]).to(DEV), torch.cat([torch.from_numpy(fx[f"u.{d}"]) for d in doms], 1).to(DEV),
             torch.from_numpy(fx["g"]).to(DEV))
    dpn = torch.from_numpy(fx["dp_noise"]).to(DEV)
    oa = model.output_adapters
    model.output_adapters = None
    with torch.no_grad():
        enc, _ = model({d: x[d] for d in doms}, num_encoded_tokens=Kk, mask_noise=noise, drop_path_noise=dpn)
    model.output_adapters = oa
    ref = torch.from_numpy(fx["enc_tokens"])
    assert rel_rms(enc.float().cpu(), ref) <= 1e-2
    loss_fns = builder.get_losses(args)
    preds, masks = model({d: x[d] for d in doms}, num_encoded_tokens=Kk, mask_noise=noise, drop_path_noise=dpn)
    total = sum(loss_fns[d](preds[d], x[d], mask=masks.get(d)) for d in preds)
    assert abs(float(total) - float(fx["loss"])) <= 3e-3 * abs(float(fx["loss"])), (float(total), float(fx["loss"]))
    total.backward()
    params = dict(model.named_parameters())
    tot = float(fx["grad_total_norm"])
    for n, gn in zip(fx["grad_names"], fx["grad_norms"]):
        mine = float(params[str(n)].grad.double().norm())
        assert abs(mine - gn) <= 4e-2 * max(gn, 1e-3 * tot), (str(n), mine, gn)
    # eval mode ignores the noise entirely (DropPath is the identity when not training)
    model.eval()
    with torch.no_grad():
        p_eval, _ = model({d: x[d] for d in doms}, num_encoded_tokens=Kk, mask_noise=noise)
        p_eval2, _ = model({d: x[d] for d in doms}, num_encoded_tokens=Kk, mask_noise=noise, drop_path_noise=dpn)
    for d in p_eval:
        assert torch.equal(p_eval[d], p_eval2[d])


def test_dropout_mask_is_the_oracles_philox_mask_bit_for_bit():
    """The fused epilogue (mrg_gemm drop_rng / drop_p / drop_stream) and the backward mask kernel (mrg_dropout_bwd_bf16) both apply
    exactly the oracle's keep pattern: zero operands + unit bias make the GEMM output the mask factor itself."""
    from mirage_amd import kernels as K
    from oracle import mirage_oracle as O
    seed, step = 0x1234ABCD5678, 7
    rng = torch.tensor([seed, step], dtype=torch.int64, device=DEV)
    for (M, N, Kd, p, sid) in ((774, 128, 64, 0.1, 0), (516, 768, 128, 0.25, 5), (130, 260, 64, 0.5, 1000003)):
        want = O.dropout_factor(M, N, p, seed, step, sid)
        A = torch.zeros(M, Kd, dtype=torch.bfloat16, device=DEV)
        W = torch.zeros(N, Kd, dtype=torch.bfloat16, device=DEV)
        out = torch.empty(M, N, dtype=torch.float32, device=DEV)
        K.gemm(A, W, M, N, Kd, out=out, bias=torch.ones(N, device=DEV), dropout=(rng, p, sid))
        assert torch.equal(out.cpu(), want), (M, N, p)
        # + residual, per-row factor and the bf16 twin, as the proj / fc2 epilogues of a Block use it
        res = torch.randn(M, N, device=DEV); rows = torch.rand(M, device=DEV) + 0.5
        out2 = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        K.gemm(A, W, M, N, Kd, out=out, out2=out2, bias=torch.ones(N, device=DEV), resid=res, row_scale=rows, dropout=(rng, p, sid))
        ref = res.cpu() + want * rows.cpu()[:, None]
        assert torch.allclose(out.cpu(), ref, rtol=1e-6, atol=1e-6) and torch.equal(out2.float(), out.bfloat16().float())
        g = torch.ones(M, N, device=DEV)
        back = K.dropout_bwd_bf16(g, (rng, p, sid)).float().cpu()
        assert torch.equal(back > 0, want > 0) and torch.allclose(back, want, rtol=4e-3)
        back2 = K.dropout_bwd_bf16(g, (rng, p, sid), row_scale=rows).float().cpu()
        assert torch.allclose(back2, want * rows.cpu()[:, None], rtol=8e-3)
    # another step / stream / seed -> another mask
    a = K.dropout_bwd_bf16(torch.ones(256, 256, device=DEV), (rng, 0.3, 0))
    rng2 = torch.tensor([seed, step + 1], dtype=torch.int64, device=DEV)
    assert not torch.equal(a, K.dropout_bwd_bf16(torch.ones(256, 256, device=DEV), (rng2, 0.3, 0)))
    assert not torch.equal(a, K.dropout_bwd_bf16(torch.ones(256, 256, device=DEV), (rng, 0.3, 1)))


def test_dropout_training_parity():
    """A16: nn.Dropout(drop_rate) after attn.proj / mlp.fc2 fused into the GEMM epilogues (forward) and regenerated for the backward
    pass, against the REFERENCE model run with the same Philox masks (tests/golden/tiny3_dropout.npz; mirage/utils.py:152,158,172,
    186-187, mirage/model.py:54,87).  Also: eval mode ignores it, every training forward draws new masks (device-side step counter,
    also under hipGraph replay), attn_drop_rate > 0 is refused loudly."""
    from mirage_amd import builder, engine as E
    from oracle import mirage_oracle as O
    fx, meta = load_golden("tiny3_dropout")
    cfg, B, Kk = meta["cfg"], meta["B"], meta["K"]
    model, args, sd = _build(cfg)
    model.train()
    assert all(b.drop_prob == pytest.approx(0.1) and b.drop_stream == 2 * i for i, b in enumerate(model.encoder))
    x = {d: t.to(DEV) for d, t in O.synthetic_batch(cfg, B, seed=meta["data_seed"]).items()}
    doms = cfg["in_domains"]
    noise = (torch.from_numpy(fx["p"]).to(DEV), torch.cat([torch.from_numpy(fx[f"u.{d}"]) for d in doms], 1).to(DEV),
             torch.from_numpy(fx["g"]).to(DEV))
    oa = model.output_adapters
    model.output_adapters = None
    E.DropoutRNG.set(DEV, meta["drop_seed"], meta["drop_step"] - 1)            # the next forward runs at the fixture's step
    with torch.no_grad():
        enc, _ = model({d: x[d] for d in doms}, num_encoded_tokens=Kk, mask_noise=noise)
    model.output_adapters = oa
    ref = torch.from_numpy(fx["enc_tokens"])
    meas = {"enc_rel_rms": rel_rms(enc.float().cpu(), ref)}
    assert meas["enc_rel_rms"] <= 1e-2
    loss_fns = builder.get_losses(args)
    E.DropoutRNG.set(DEV, meta["drop_seed"], meta["drop_step"] - 1)
    preds, masks = model({d: x[d] for d in doms}, num_encoded_tokens=Kk, mask_noise=noise)
    total = sum(loss_fns[d](preds[d], x[d], mask=masks.get(d)) for d in preds)
    meas["loss_rel"] = abs(float(total) - float(fx["loss"])) / abs(float(fx["loss"]))
    meas["loss_rel_if_dropout_were_off"] = abs(float(fx["loss_without_dropout"]) - float(fx["loss"])) / abs(float(fx["loss"]))
    assert meas["loss_rel"] <= 3e-3 and meas["loss_rel"] < 0.5 * meas["loss_rel_if_dropout_were_off"], meas
    total.backward()
    params = dict(model.named_parameters())
    tot = float(fx["grad_total_norm"])
    worst = 0.0
    for n, gn in zip(fx["grad_names"], fx["grad_norms"]):
        mine = float(params[str(n)].grad.double().norm())
        worst = max(worst, abs(mine - gn) / max(gn, 1e-3 * tot))
    worst_t = 0.0
    for key in fx:
        if key.startswith("grad."):
            r = torch.from_numpy(fx[key]).double()
            worst_t = max(worst_t, float((params[key[5:]].grad.cpu().double() - r).norm() / max(float(r.norm()), 1e-3 * tot)))
    meas.update(grad_worst_rel=worst, grad_tensor_worst_rel=worst_t)
    report("tiny3_dropout", **meas)
    assert worst <= 1.5e-2 and worst_t <= 3e-2, meas
    # every training forward advances the device-side step: new masks, different output; eval mode has no dropout at all
    with torch.no_grad():
        e1, _ = model({d: x[d] for d in doms}, num_encoded_tokens=Kk, mask_noise=noise)
        e2, _ = model({d: x[d] for d in doms}, num_encoded_tokens=Kk, mask_noise=noise)
        assert not torch.equal(e1[d0 := list(e1)[0]], e2[d0])
        model.eval()
        v1, _ = model({d: x[d] for d in doms}, num_encoded_tokens=Kk, mask_noise=noise)
        v2, _ = model({d: x[d] for d in doms}, num_encoded_tokens=Kk, mask_noise=noise)
        assert torch.equal(v1[d0], v2[d0])


def test_attention_probability_dropout_matches_reference_fixture():
    """attn_drop_rate > 0 in the encoder (0.15) and in the output adapters (0.1) (A16; mirage/model.py:55,88, output_adapters.py:64 ->
    utils.py:181-185, 216-220: SDPA's dropout_p): the HIP path against the REFERENCE model run with scaled_dot_product_attention replayed on the
    same Philox masks (tests/golden/tiny3_attn_dropout.npz, tools/gen_golden.py run_attn_dropout_case): encoder tokens, predictions, loss,
    every gradient norm, a few whole gradients.  Like the reference's SDPA call the dropout also applies in eval mode."""
    from mirage_amd import builder, engine as E
    from oracle import mirage_oracle as O
    fx, meta = load_golden("tiny3_attn_dropout")
    cfg, B, Kk = meta["cfg"], meta["B"], meta["K"]
    model, args, _ = _build(cfg)
    model.train()
    assert [a.attn_drop_stream for a in model._mrg_adrop] == ([0x10000 + i for i in range(cfg["depth"])]
            + [0x20000 + 256 * a + j for a in range(len(cfg["out_domains"])) for j in range(1 + cfg["decoder_depth"])])
    x = {d: t.to(DEV) for d, t in O.synthetic_batch(cfg, B, seed=meta["data_seed"]).items()}
    doms = cfg["in_domains"]
    noise = (torch.from_numpy(fx["p"]).to(DEV), torch.cat([torch.from_numpy(fx[f"u.{d}"]) for d in doms], 1).to(DEV),
             torch.from_numpy(fx["g"]).to(DEV))
    loss_fns = builder.get_losses(args)
    E.DropoutRNG.set(DEV, meta["drop_seed"], meta["drop_step"] - 1)            # the next forward runs at the fixture's step
    preds, masks = model({d: x[d] for d in doms}, num_encoded_tokens=Kk, mask_noise=noise)
    meas = {}
    for d in preds:
        meas[f"pred_rel_rms.{d}"] = rel_rms(preds[d].float().cpu(), torch.from_numpy(fx[f"pred.{d}"]))
        assert meas[f"pred_rel_rms.{d}"] <= 2e-2, meas
    total = sum(loss_fns[d](preds[d], x[d], mask=masks.get(d)) for d in preds)
    meas["loss_rel"] = abs(float(total) - float(fx["loss"])) / abs(float(fx["loss"]))
    meas["loss_rel_if_dropout_were_off"] = abs(float(fx["loss_without_dropout"]) - float(fx["loss"])) / abs(float(fx["loss"]))
    assert meas["loss_rel"] <= 3e-3 and meas["loss_rel"] < 0.5 * meas["loss_rel_if_dropout_were_off"], meas
    total.backward()
    params = dict(model.named_parameters())
    tot = float(fx["grad_total_norm"])
    worst = 0.0
    for n, gn in zip(fx["grad_names"], fx["grad_norms"]):
        mine = float(params[str(n)].grad.double().norm())
        worst = max(worst, abs(mine - gn) / max(gn, 1e-3 * tot))
    worst_t = 0.0
    for key in fx:
        if key.startswith("grad."):
            r = torch.from_numpy(fx[key]).double()
            worst_t = max(worst_t, float((params[key[5:]].grad.cpu().double() - r).norm() / max(float(r.norm()), 1e-3 * tot)))
    meas.update(grad_worst_rel=worst, grad_tensor_worst_rel=worst_t)
    report("tiny3_attn_dropout", **meas)
    assert worst <= 1.5e-2 and worst_t <= 3e-2, meas
    # encoder alone at the fixture's step; every forward advances the step (new masks) -- also in eval mode, as with the reference's SDPA call
    oa = model.output_adapters
    model.output_adapters = None
    with torch.no_grad():
        E.DropoutRNG.set(DEV, meta["drop_seed"], meta["drop_step"] - 1)
        enc, _ = model({d: x[d] for d in doms}, num_encoded_tokens=Kk, mask_noise=noise)
        assert rel_rms(enc.float().cpu(), torch.from_numpy(fx["enc_tokens"])) <= 1e-2
        model.eval()
        v1, _ = model({d: x[d] for d in doms}, num_encoded_tokens=Kk, mask_noise=noise)
        v2, _ = model({d: x[d] for d in doms}, num_encoded_tokens=Kk, mask_noise=noise)
        assert not torch.equal(v1, v2)
    model.output_adapters = oa


@pytest.mark.parametrize("fixture", ["tiny3_dropout", "tiny3_attn_dropout"])
def test_dropout_graph_replay_draws_new_masks_and_matches_eager(fixture):
    """TrainStep with drop_rate > 0 (or attn_drop_rate > 0): the Philox step lives on the device and is advanced inside the captured graph, so
    replays draw fresh masks, and a graph-replayed run equals the eager run step for step (same seed): losses bit-identical over 4 steps."""
    from mirage_amd import builder, optim, engine as E
    from mirage_amd.trainer import TrainStep
    from oracle import mirage_oracle as O
    fx, meta = load_golden(fixture)
    cfg, B, Kk = meta["cfg"], meta["B"], meta["K"]
    doms = cfg["in_domains"]
    x = {d: t.to(DEV) for d, t in O.synthetic_batch(cfg, B, seed=meta["data_seed"]).items()}
    noise = (torch.from_numpy(fx["p"]).to(DEV), torch.cat([torch.from_numpy(fx[f"u.{d}"]) for d in doms], 1).to(DEV),
             torch.from_numpy(fx["g"]).to(DEV))

    def run(use_graph):
        model, args, _ = _build(cfg)
        model.train()
        opt = optim.FusedAdamW(model, [1e-3] * 8, [0.05] * 8)
        ts = TrainStep(model, builder.get_losses(args), opt, x, Kk, use_graph=use_graph, feed_mask_noise=True)
        ts.load(x)
        for dst, src in zip(ts.static_noise, noise):
            dst.copy_(src)
        E.DropoutRNG.set(DEV, 4242, 0)
        ts.warmup_and_capture(warm_iters=1)
        losses = []
        for _ in range(4):
            ts.step(x, noise=noise)
            losses.append(float(ts.loss))
        return losses, int(E.DropoutRNG.state(DEV)[1])
    eager, st_e = run(False)
    graph, st_g = run(True)
    assert eager == graph, (eager, graph)
    assert st_e == st_g == 4                       # warm-up left no trace in the step counter; one step per optimizer step
    assert len(set(eager)) == 4


def test_modalities_follow_the_keys_of_x():
    """mirage/model.py:352-356 tokenises `x.items()`: the modalities present, in x's own order.  A 3-modal encoder fed {slo, bscan} (a subset, in
    another order than the adapters') must equal the oracle run on that order with the same weights; with output adapters an absent output task
    is the reference's KeyError (output_adapters.py:266)."""
    from mirage_amd import builder
    from oracle import mirage_oracle as O
    _, meta = load_golden("tiny3")
    cfg = meta["cfg"]
    model, args, sd = _build(dict(cfg, out_domains=[]))
    model.eval()
    sub = ["slo", "bscan"]
    assert list(model.input_adapters.keys()) != sub and len(model.input_adapters) == 3
    cfg_sub = dict(cfg, in_domains=sub, out_domains=[])
    B, Kk = 4, 20
    x = O.synthetic_batch(cfg, B, seed=91)
    rs = np.random.RandomState(5)
    T = [cfg["domains"][d]["grid"] ** 2 for d in sub]
    p = rs.gamma(1.0, size=(B, 2)).astype(np.float32); p /= p.sum(1, keepdims=True)
    u = [rs.rand(B, t).astype(np.float32) for t in T]
    g = rs.rand(B, sum(T)).astype(np.float32)
    tm, ids_keep, ids_restore = O.generate_masks_from_noise(p, u, g, Kk)
    with torch.no_grad():
        want, _ = O.forward(cfg_sub, sd, {d: x[d] for d in sub}, torch.from_numpy(ids_keep), torch.from_numpy(ids_restore), decode=False)
        noise = (torch.from_numpy(p).to(DEV), torch.cat([torch.from_numpy(a) for a in u], 1).to(DEV), torch.from_numpy(g).to(DEV))
        enc, masks = model({d: x[d].to(DEV) for d in sub}, num_encoded_tokens=Kk, mask_noise=noise)
        assert list(masks.keys()) == sub
        assert rel_rms(enc.float().cpu(), want) <= 1e-2
        enc2, masks2 = model({d: x[d].to(DEV) for d in reversed(sub)}, num_encoded_tokens=Kk, mask_noise=noise)
        assert list(masks2.keys()) == sub[::-1] and not torch.equal(enc, enc2)      # same noise, other token order: another selection
        one, _ = model({"bscan": x["bscan"].to(DEV)}, mask_inputs=False)
        assert one.shape[1] == cfg["domains"]["bscan"]["grid"] ** 2 + cfg["num_global_tokens"]
    full, _, _ = _build(cfg)
    with pytest.raises(KeyError):
        full({d: x[d].to(DEV) for d in sub}, num_encoded_tokens=Kk)
    with pytest.raises(ValueError, match="none of the input adapters"):
        model({"other": x["bscan"].to(DEV)})


def test_optimizer_refuses_to_step_an_adapter_that_was_not_in_the_batch():
    """A training call on a subset of the modalities: torch leaves the unused adapter's .grad at None and the reference's optimizer skips
    its parameters (mutils/optim_factory.py:180 -> torch.optim.AdamW skips p.grad is None).  One fused launch over the arena cannot skip, and
    stepping the cleared gradients would still apply weight decay and momentum decay: FusedAdamW.step must refuse (round-4 advisor item)."""
    from mirage_amd import optim
    from oracle import mirage_oracle as O
    _, meta = load_golden("tiny3")
    cfg = meta["cfg"]
    model, args, sd = _build(dict(cfg, out_domains=[]))
    model.train()
    opt = optim.FusedAdamW(model, [1e-3], [0.05])
    x = O.synthetic_batch(cfg, 4, seed=91)
    enc, _ = model({d: x[d].to(DEV) for d in ("slo", "bscan")}, num_encoded_tokens=20)
    opt.zero_grad()
    enc.float().square().mean().backward()
    assert model._mrg_unused_adapters == ["bscanlayermap"]
    with pytest.raises(RuntimeError, match="were not in the batch"):
        opt.step()
    enc, _ = model({d: x[d].to(DEV) for d in cfg["in_domains"]}, num_encoded_tokens=20)
    enc.float().square().mean().backward()
    assert model._mrg_unused_adapters == []
    opt.step()
    torch.cuda.synchronize()


def test_attention_dropout_snapshot_does_not_outlive_the_forward_pass():
    """model.forward hands its attention modules ONE Philox snapshot per pass (engine.attn_dropout_args); it must be gone afterwards, so
    that a stand-alone call of such a module draws its own masks instead of replaying the last pass's (round-4 advisor item)."""
    _, meta = load_golden("tiny3_attn_dropout")
    model, args, sd = _build(meta["cfg"])
    model.train()
    from oracle import mirage_oracle as O
    x = O.synthetic_batch(meta["cfg"], 2, seed=3)
    assert model._mrg_adrop
    model({d: x[d].to(DEV) for d in meta["cfg"]["in_domains"]}, num_encoded_tokens=meta["K"])
    assert all(getattr(a, "_mrg_adrop_rng", None) is None for a in model._mrg_adrop)
    torch.cuda.synchronize()


def test_mixed_token_grids_raise_like_the_reference():
    """The reference asserts that every context task has as many tokens as the decoding adapter's grid
    (mirage/output_adapters.py:176-179); the HIP path must fail the same way, not read out of bounds."""
    from mirage_amd import builder
    args = builder.make_args(("bscan", "bscanlayermap"), input_size=64, patch_size=8, layermap_size=128, layermap_patch=8,
                             decoder_dim=64, decoder_depth=1, decoder_num_heads=2)
    model = builder.get_model(args, dim_tokens=128, depth=1, num_heads=2).to(DEV)
    x = {"bscan": torch.rand(2, 1, 64, 64, device=DEV), "bscanlayermap": torch.randint(0, 13, (2, 128, 128), device=DEV)}
    with pytest.raises(AssertionError, match="# tokens"):
        model(x, num_encoded_tokens=32)
    torch.cuda.synchronize()


def test_loss_on_unmasked_matches_oracle():
    """run_pretraining.py:718-721 with --loss_on_unmasked: the criteria are called without a mask (criterion.py:49,115:
    plain mean over every pixel)."""
    from mirage_amd import builder
    from oracle import mirage_oracle as O
    fx, meta = load_golden("tiny3")
    cfg, B, Kk = meta["cfg"], meta["B"], meta["K"]
    model, args, sd = _build(cfg)
    model.train()
    x = O.synthetic_batch(cfg, B, seed=meta["data_seed"])
    xg = {d: t.to(DEV) for d, t in x.items()}
    doms = cfg["in_domains"]
    noise = (torch.from_numpy(fx["p"]).to(DEV), torch.cat([torch.from_numpy(fx[f"u.{d}"]) for d in doms], 1).to(DEV),
             torch.from_numpy(fx["g"]).to(DEV))
    loss_fns = builder.get_losses(args)
    preds, _ = model({d: xg[d] for d in doms}, num_encoded_tokens=Kk, mask_noise=noise)
    mine = {d: loss_fns[d](preds[d], xg[d]) for d in preds}
    ik, ir = torch.from_numpy(fx["ids_keep"]), torch.from_numpy(fx["ids_restore"])
    with torch.no_grad():
        _, opreds = O.forward(cfg, sd, x, ik, ir)
        ref = O.losses(cfg, opreds, x, {})
    for d in preds:
        assert abs(float(mine[d]) - float(ref[d])) <= 3e-3 * max(1.0, abs(float(ref[d]))), (d, float(mine[d]), float(ref[d]))
    sum(mine.values()).backward()
    assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)
