"""Host-side logic that needs no GPU: module surface / state_dict layout / init stream / schedules / parameter groups
against fixtures captured from the reference (tools/gen_golden.py)."""
import json
import os
import sys

import numpy as np
import pytest
import torch

from conftest import load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tiny_model(seed=None):
    from mirage_amd import builder
    fx, meta = load_golden("harness_tiny")
    cfg = meta["cfg"]
    args = builder.make_args(cfg["in_domains"], cfg["out_domains"], input_size=64, patch_size=8, layermap_size=16, layermap_patch=2,
                             decoder_dim=64, decoder_depth=2, decoder_num_heads=2)
    if seed is not None:
        torch.manual_seed(seed)
    return builder.get_model(args, dim_tokens=128, depth=2, num_heads=2), fx, cfg


def test_state_dict_layout_matches_reference():
    model, fx, cfg = _tiny_model()
    sd = model.state_dict()
    assert list(sd.keys()) == [str(k) for k in fx["state_dict_keys"]]
    for k, s in zip(sd.keys(), fx["state_dict_shapes"]):
        assert list(sd[k].shape) == json.loads(str(s)), k
    assert sorted(n for n, p in model.named_parameters() if p.requires_grad) == sorted(str(s) for s in fx["trainable"])
    assert sorted(model.no_weight_decay()) == [str(s) for s in fx["no_weight_decay"]]
    assert model.get_num_layers() == 2


def test_init_stream_matches_reference_bit_for_bit():
    """Same seed -> same weights as the reference constructor (mirage/model.py:95-121): module registration order and
    init calls consume the torch RNG identically."""
    model, fx, cfg = _tiny_model(seed=1234)
    sd = model.state_dict()
    names = [str(n) for n in fx["init.names"]]
    for n, nrm, sm in zip(names, fx["init.norms"], fx["init.sums"]):
        assert abs(float(sd[n].double().norm()) - nrm) <= 1e-12 + 1e-9 * abs(nrm), n
        assert abs(float(sd[n].double().sum()) - sm) <= 1e-9 + 1e-9 * abs(sm), n
    assert np.array_equal(sd["global_tokens"].numpy(), fx["init.global_tokens"])
    assert np.array_equal(sd["encoder.0.attn.qkv.weight"][:4, :8].numpy(), fx["init.enc0_qkv_w_head"])


def test_posemb_tables_and_quirk():
    from mirage_amd.utils import build_2d_sincos_posemb
    fx, _ = load_golden("harness_tiny")
    assert np.array_equal(build_2d_sincos_posemb(2, 3, 8).numpy(), fx["posemb.2x3x8"])
    np.testing.assert_allclose(build_2d_sincos_posemb(16, 16, 768).numpy(), fx["posemb.16x16x768"], atol=1e-6)


def test_schedules_and_parameter_groups():
    from mirage_amd import optim
    model, fx, cfg = _tiny_model()
    np.testing.assert_allclose(optim.cosine_scheduler(1e-4, 0.0, 10, 7, warmup_epochs=2), fx["sched.a"], rtol=1e-12)
    np.testing.assert_allclose(optim.cosine_scheduler(0.05, 0.05, 10, 7), fx["sched.b"], rtol=1e-12)
    np.testing.assert_allclose(optim.cosine_scheduler(2e-4, 1e-6, 6, 5, warmup_epochs=1, warmup_steps=3), fx["sched.c"], rtol=1e-12)
    groups = optim.get_parameter_groups(model, 0.05, model.no_weight_decay())
    by = {("decay" if g["weight_decay"] > 0 else "no_decay"): g["names"] for g in groups}
    assert by["decay"] == [str(s) for s in fx["group.decay"]]
    assert by["no_decay"] == [str(s) for s in fx["group.no_decay"]]


def test_factory_and_make_mask():
    from mirage_amd.model import MIRAGEModel, model_factory
    assert set(model_factory) >= {"miragepre_base", "miragepre_large"}
    m = MIRAGEModel.make_mask(4, 4, {"bscan": [(0, 1), (2, 3)], "slo": []}, full_tasks=["slo"], device="cpu")
    assert m["bscan"].shape == (1, 16) and float(m["bscan"].sum()) == 14 and float(m["slo"].sum()) == 0


def test_bucket_ranges_cover_exactly():
    from mirage_amd.dp import bucket_ranges
    for n, b in ((1000, 256), (64, 64), (98_000_000, 16 << 20), (5, 64)):
        r = bucket_ranges(n, b)
        assert r[0][0] == 0 and r[-1][1] == n and all(a[1] == c[0] for a, c in zip(r, r[1:]))


def test_hf_directory_layout_roundtrip(tmp_path):
    """F4: save_pretrained / from_pretrained write and read the PyTorchModelHubMixin layout the reference uses
    (hf/helper_hf.py:9-58): config.json + model.safetensors with `model.`-prefixed keys."""
    import json, os
    import torch
    from safetensors.torch import load_file
    from mirage_amd.hf import MIRAGEhf
    torch.manual_seed(0)
    m = MIRAGEhf(input_size=64, patch_size=8, modalities="bscan-slo", size="tiny", dim_tokens=64, depth=2, num_heads=2)
    d = m.save_pretrained(str(tmp_path / "MIRAGE-tiny"))
    cfg = json.load(open(os.path.join(d, "config.json")))
    assert cfg == {"input_size": 64, "patch_size": 8, "modalities": "bscan-slo", "size": "tiny"}
    sd = load_file(os.path.join(d, "model.safetensors"))
    assert set(sd) == set(m.state_dict()) and all(k.startswith("model.") for k in sd)
    assert "model.global_tokens" in sd and "model.input_adapters.slo.proj.weight" in sd and "model.encoder.1.mlp.fc2.bias" in sd
    m2 = MIRAGEhf.from_pretrained(d, device="cpu", dim_tokens=64, depth=2, num_heads=2)
    for k, v in m.state_dict().items():
        assert torch.equal(v, m2.state_dict()[k]), k
    # a bare (un-prefixed) model state_dict loads too, as in hf/helper_hf.py:50-54
    m3 = MIRAGEhf(input_size=64, patch_size=8, modalities="bscan-slo", size="tiny", dim_tokens=64, depth=2, num_heads=2)
    msg = m3.load_state_dict(m.model.state_dict(), strict=False)
    assert len(msg.missing_keys) == 0


def test_half_batch_placement_and_weight_gradient_collector_rules():
    # kernels.Halves: chain A allocates full-batch tensors and works on the first halves, chain B receives the second halves of the
    # SAME tensors in the same order; full_of() recovers the full tensor from any view of a first half (no GPU needed: placement only)
    import pytest
    from mirage_amd import kernels as K, _lib as L
    K.Halves.begin(1)
    a1 = K.new((4, 3, 8), torch.float32, "cpu"); a2 = K.new((12,), torch.float32, "cpu")
    K.Halves.end()
    K.Halves.begin(2)
    b1 = K.new((4, 3, 8), torch.float32, "cpu"); b2 = K.new((12,), torch.float32, "cpu")
    K.Halves.end()
    a1.fill_(1.0); b1.fill_(2.0); a2.fill_(3.0); b2.fill_(4.0)
    f1 = K.Halves.full_of(a1.view(12, 8)); f2 = K.Halves.full_of(a2)
    assert f1.shape == (24, 8) and torch.equal(f1[:12], torch.ones(12, 8)) and torch.equal(f1[12:], torch.full((12, 8), 2.0))
    assert f2.shape == (24,) and torch.equal(f2[12:], torch.full((12,), 4.0))
    assert K.Halves.full_of(None) is None and K.new((2,), torch.float32, "cpu").shape == (2,)        # mode 0: plain torch.empty
    K.Halves.begin(2)
    with pytest.raises(RuntimeError, match="diverged"):
        K.new((4, 3, 9), torch.float32, "cpu")
    K.Halves.end()
    # kernels.WgradGroup.eligible: only plain weight gradients (fp32 overwrite, nothing fused, aligned, whole 64-token k-tiles)
    A = torch.zeros(128, 512, dtype=torch.bfloat16); B = torch.zeros(128, 256, dtype=torch.bfloat16); out = torch.zeros(512, 256)
    none = dict(out2=None, bias=None, resid=None, accumulate=False, act=L.MRG_ACT_NONE, aux_out=None, aux_in=None, row_map=None, add_tab=None,
                a_map=None, b_map=None, m_dev=None, k_dev=None, row_scale=None, colsum=None)
    ok = lambda **kw: K.WgradGroup.eligible(A, B, kw.pop("M", 512), kw.pop("N", 256), kw.pop("K", 128), None, None, kw.pop("out", out), None,
                                            **{**none, **kw})
    assert ok()
    assert not ok(accumulate=True) and not ok(bias=torch.zeros(256)) and not ok(a_map=torch.zeros(4, dtype=torch.int32))
    assert not ok(K=100) and not ok(M=128) and not ok(out=out.bfloat16()) and not ok(K=K.WgradGroup.max_k + 64)


def test_bench_refuses_more_ranks_than_gpus():
    """`python bench.py --gpus 2` on a box without two GPUs must refuse (exit code 2, message on stderr, nothing on stdout) instead of
    running fewer ranks or hanging in a rendezvous (bench.py::_self_launch)."""
    import subprocess, sys
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MRG_SHARE_GPU"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=300)
    if torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs visible")
    assert p.returncode == 2 and "only" in p.stderr and p.stdout.strip() == ""


def test_wrapper_inference_geometry():
    """mirage_wrapper._inference_args: modalities -> per-modality image / patch / grid tables (layer maps keep 128 / 8 whatever the image
    size; `grid_size` is the image grid for every modality, as mirage_wrapper.py:66-82 builds it), the pooling table and the registry."""
    import types
    from mirage_amd import mirage_wrapper as W
    ck = types.SimpleNamespace(model="miragepre_base", out_domains="bscan-slo", decoder_dim=256, decoder_depth=2, decoder_num_heads=8)
    a = W._inference_args(ck, "bscan-bscanlayermap", 512, 32)
    assert a.in_domains == ["bscan", "bscanlayermap"] and a.out_domains == ["bscan", "slo"]
    assert a.input_size == {"bscan": (512, 512), "bscanlayermap": (128, 128)} and a.patch_size == {"bscan": (32, 32), "bscanlayermap": (8, 8)}
    assert a.grid_size == {"bscan": [16, 16], "bscanlayermap": [16, 16]} and a.grid_sizes == {"bscan": [16, 16], "bscanlayermap": [16, 16]}
    assert a.num_global_tokens == 1 and a.drop_path == 0.0 and ck.out_domains == "bscan-slo"          # the checkpoint's namespace is not mutated
    b = W._inference_args(ck, "slo", (256, 128), 16)
    assert b.input_size["slo"] == (256, 128) and b.grid_size["slo"] == [16, 8]
    assert set(W.miragecls_factory) == {"global", "cls", "token_mix"}
    assert W.miragecls_factory["token_mix"] is W.MIRAGEClsTokenMix and issubclass(W.MIRAGEClsCLS, W.MIRAGEClsGlobal)
    x = torch.arange(2 * 5 * 4, dtype=torch.float32).reshape(2, 5, 4)
    g = W._token_groups(x, 1)
    assert g["patch"].shape == (2, 4, 4) and g["global"].shape == (2, 1, 4) and torch.equal(g["global"][:, 0], x[:, -1])


def test_partial_sum_slots_are_replay_safe():
    """optim.PartialSumSlots (the bookkeeping behind the fused gradient norm; round-2 defect: a host cursor that a captured step() never
    reset): a range keeps its slice for ever, repeated steps hand out nothing new, a repartition retires what it overlaps, and only a
    gap-free tiling of the arena is accepted."""
    from mirage_amd.optim import PartialSumSlots
    sl = PartialSumSlots(capacity=100)
    step1 = [sl.get(s, e, n)[:2] for s, e, n in ((512, 1024, 7), (256, 512, 5), (0, 256, 3))]          # backward order: last stage first
    assert step1 == [(0, 7), (7, 5), (12, 3)] and sl.used == 15 and sl.is_partition(1000)
    for _ in range(1000):                                                                            # 1000 further steps: same slices, no growth
        assert [sl.get(s, e, n)[:2] for s, e, n in ((512, 1024, 7), (256, 512, 5), (0, 256, 3))] == step1
    assert sl.used == 15
    # a different partition of the same arena (single launch) retires all three and gets a fresh slice; the old ones are reported for zeroing
    off, n, retired = sl.get(0, 1024, 9)
    assert (off, n) == (15, 9) and sorted(retired) == [(0, 7), (7, 5), (12, 3)] and sl.is_partition(1000) and sl.used == 24
    # back to stages: the single range is retired again, slices are NEW ones (captured graphs may still hold the old addresses)
    a = sl.get(512, 1024, 7)
    assert a[0] == 24 and a[2] == [(15, 9)] and not sl.is_partition(1000)                            # [0, 512) not covered yet
    sl.get(0, 512, 8)
    assert sl.is_partition(1000) and not sl.is_partition(2000)
    with pytest.raises(RuntimeError, match="exhausted"):
        PartialSumSlots(capacity=4).get(0, 64, 5)


def test_gelu_cdf_rational_coefficients():
    """The GELU-only epilogue (csrc/common.h gelu_cdf_rational; replaces nn.GELU() of mirage/utils.py:142,156 in the inference forward)
    evaluates Phi(x) - 0.5 = x P(x^2) / Q(x^2), clamped at |x| = 5, in fp32 Horner form.  The coefficients are read from the header and
    re-evaluated here exactly as the kernel does (numpy float32) against erf: |error| of Phi <= 1e-6 everywhere, of x Phi(x) <= 1e-5 (|x| <= 9: the absolute error of Phi times |x|)."""
    import os, re
    from scipy.special import erf
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mirage_amd", "csrc", "common.h")).read()
    co = {m.group(1): np.float32(float(m.group(2))) for m in re.finditer(r"#define MRG_GELU_([PQ]\d)\s+\(?(-?[0-9.e+-]+)f\)?", src)}
    assert sorted(co) == ["P0", "P1", "P2", "P3", "P4", "Q0", "Q1", "Q2", "Q3"]
    x = np.linspace(-9, 9, 1_800_001).astype(np.float32)
    xc = np.clip(x, np.float32(-5), np.float32(5))
    u = (xc * xc).astype(np.float32)
    p = co["P4"]
    for k in ("P3", "P2", "P1", "P0"):
        p = (p * u + co[k]).astype(np.float32)
    q = co["Q3"]
    for k in ("Q2", "Q1", "Q0"):
        q = (q * u + co[k]).astype(np.float32)
    q = (q * u + np.float32(1)).astype(np.float32)
    phi = ((xc * p).astype(np.float32) * (np.float32(1) / q).astype(np.float32) + np.float32(0.5)).astype(np.float32)
    ref = 0.5 * (1 + erf(x.astype(np.float64) / np.sqrt(2)))
    assert float(np.abs(phi - ref).max()) <= 1e-6
    assert float(np.abs(x.astype(np.float64) * phi - x.astype(np.float64) * ref).max()) <= 1e-5
    assert float(phi.min()) >= -1e-6 and float(phi.max()) <= 1.0 + 1e-6


def test_task_mask_columns_are_read_in_place_and_masked_pixel_coefficients():
    """engine.loss_mask hands the loss kernels a task's columns of the step's [B, all tokens] mask without a copy (mrg_loss_desc.mask_bs) and
    copies only what they cannot address; the backward coefficients the fused loss launch forms are 1 / (masked pixels x samples with any)
    (criterion.py:36-51, 103-117: per-sample mean over the masked pixels, nanmean over the samples) -- restated here against torch autograd
    of the oracle's masked MSE, which is what mrg_loss_coef / mrg_loss_desc.nvalid are tested against on the GPU."""
    from mirage_amd import engine as E
    from oracle import mirage_oracle as O
    m = (torch.rand(4, 48) < 0.6).long()
    v = E.loss_mask(m[:, 16:32])
    assert v.data_ptr() == m[:, 16:32].data_ptr() and v.stride() == (48, 1)
    assert E.loss_mask(None) is None
    assert E.loss_mask(m.t()[:16].t()[:, ::2]).is_contiguous()                      # strided tokens: copied
    assert E.loss_mask(m[:, :16].bool()).dtype == torch.int64
    # coefficients: d loss / d pred = 2 (pred - target) mask coef[b] / C for the MSE
    torch.manual_seed(3)
    B, g, P = 4, 4, 2
    pred = torch.randn(B, 1, g * P, g * P, requires_grad=True)
    tgt = torch.rand(B, 1, g * P, g * P)
    mask = (torch.rand(B, g * g) < 0.5).long()
    mask[2] = 0
    O.masked_mse(pred, tgt, mask, P).backward()
    n = mask.sum(1).float() * P * P
    nvalid = float((n > 0).sum())
    coef = torch.where(n > 0, 1.0 / (n * nvalid), torch.zeros_like(n))
    up = mask.view(B, 1, g, g).repeat_interleave(P, 2).repeat_interleave(P, 3).float()
    want = 2.0 * (pred.detach() - tgt) * up * coef.view(B, 1, 1, 1)
    keep = [0, 1, 3]                                                                # (the reference back-propagates NaN into sample 2)
    torch.testing.assert_close(pred.grad[keep], want[keep], rtol=1e-5, atol=1e-7)


def test_bench_step_floors_are_consistent():
    """bench.py's `roofline.hbm` (round 6): the whole step against its HBM floor (algorithmic bytes / 6.3 TB/s) and its MFMA floor (algorithmic
    flops / 2.5 PF).  The byte model must reproduce the documented C2 figure (19.0 GB of encoder dataflow + 4.0 GB = 23.0 GB),
    scale per GPU batch, and name the binding floor: HBM for the training configurations, MFMA (barely) for the encoder-only forward."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        spec.loader.exec_module(b)
    finally:
        sys.argv = argv
    by = {name: b.algorithmic_bytes(m, doms, B, K, train) for name, (m, doms, B, K, train) in b.CONFIGS.items()}
    assert abs(by["c2"] / 1e9 - 23.0) < 0.3
    m, doms, B, K, train = b.CONFIGS["c4"]
    assert abs(b.algorithmic_bytes(m, doms, 2 * B, K, train) / by["c4"] - 2.0) < 0.2          # activations dominate: ~2x at twice the batch
    for name, ms in (("c2", 8.77), ("c4", 19.52), ("c5", 47.9)):
        m, doms, B, K, train = b.CONFIGS[name]
        fpi = b.flops_per_image(m, doms, K, train)
        h = b.step_hbm(m, doms, B, K, train, ms, fpi)
        assert h["binding_floor"] == ("mfma" if name == "c5" else "hbm")
        floor = max(h["hbm_floor_ms"], h["mfma_floor_ms"])
        assert abs(h["frac_of_binding_floor"] - floor / ms) < 1e-3 and 0.3 < h["frac_of_binding_floor"] < 0.5
        assert abs(h["flop_per_byte"] - B * fpi / by[name]) < 0.2 and h["ridge_flop_per_byte"] > 390
        assert abs(h["achieved_tbps"] - by[name] / (ms * 1e-3) / 1e12) < 2e-3


def test_debug_image_composition_matches_the_reference_recipe():
    """run_pretraining.save_debug_images (reference run_pretraining.py:568-645): unmasked patches of the prediction replaced by the input, input
    over prediction at 128 x 128 (nearest), tasks separated by an 8-pixel white line, layer maps as argmax / max; the file is the 8-per-row
    grid torchvision.utils.save_image would write.  Restated independently here with einops, as the reference writes it."""
    from einops import rearrange
    import torch.nn.functional as F
    from mirage_amd import run_pretraining as R
    g = torch.Generator().manual_seed(5)
    B = 3
    x = {"bscan": torch.rand(B, 1, 64, 64, generator=g), "bscanlayermap": torch.randint(0, 13, (B, 16, 16), generator=g)}
    preds = {"bscan": torch.rand(B, 1, 64, 64, generator=g).bfloat16(), "bscanlayermap": torch.randn(B, 13, 16, 16, generator=g).bfloat16()}
    masks = {"bscan": torch.randint(0, 2, (B, 64), generator=g), "bscanlayermap": torch.randint(0, 2, (B, 64), generator=g)}
    patch = {"bscan": (8, 8), "bscanlayermap": (2, 2)}
    img = R.debug_image(preds, masks, x, patch)
    rows = []
    for t in preds:
        p, inp = preds[t].float(), x[t]
        if t == "bscanlayermap":
            p = F.softmax(p, dim=1).argmax(dim=1, keepdim=True).float()
            inp = inp.unsqueeze(1)
            p = p / p.max(); inp = inp / inp.max()
        c, h, w = inp.shape[1], inp.shape[2] // patch[t][0], inp.shape[3] // patch[t][1]
        m = F.interpolate(rearrange(masks[t], "b (c h w) -> b c h w", c=c, h=h, w=w).float(), size=inp.shape[2:], mode="nearest")
        p = p * m + inp * (1 - m)
        rows += [F.interpolate(inp.float(), (128, 128), mode="nearest"), F.interpolate(p, (128, 128), mode="nearest"), torch.ones(B, 1, 8, 128)]
    want = torch.cat(rows[:-1], dim=2)
    assert img.shape == (B, 1, 2 * 256 + 8, 128) and torch.equal(img, want)
    import tempfile
    from PIL import Image
    with tempfile.TemporaryDirectory() as d:
        path = R.save_debug_images(preds, masks, x, 7, patch, d)
        assert path.name == "7_bscan-bscanlayermap.jpg" and path.parent.name == "debug"
        im = Image.open(path)
        assert im.size == (3 * 130 + 2, 522 + 2) and im.mode == "RGB"
