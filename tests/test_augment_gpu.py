"""F2 data path on the GPU: mrg_augment_image / mrg_augment_label (through the C-ABI) against the CPU oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _oracle_batch(A, D, u8, task, flips, affs, shifts, do_affine=True):
    outs = []
    for b in range(u8.shape[0]):
        arr = u8[b].astype(np.int64) if task in A.LABEL_TASKS else u8[b].astype(np.float32) / 255.0
        outs.append(A.augment_sample({task: arr}, flips[b], {task: float(shifts[task][b])}, affs[b], do_affine)[task])
    return torch.stack(outs)


@pytest.mark.parametrize("task,H,W", [("bscan", 512, 512), ("slo", 512, 512), ("bscan", 96, 160)])
def test_image_augmentation_matches_oracle(task, H, W):
    from mirage_amd import data as D
    from oracle import augment_oracle as A
    import types
    rng = np.random.default_rng(11)
    B = 6
    u8 = rng.integers(0, 256, (B, H, W), dtype=np.uint8)
    aug = D.DataAugmentationForMIRAGE(types.SimpleNamespace(hflip=0.5, intensity_shift=0.1, affine=True), seed=3)
    flips, affs, shifts = aug.draw([task], B)
    assert any(flips) and not all(flips)
    got = aug.apply({task: torch.from_numpy(u8).to(DEV)}, flips, affs, shifts)[task]
    assert got.shape == (B, 1, H, W) and got.dtype == torch.float32
    ref = _oracle_batch(A, D, u8, task, flips, affs, shifts)
    err = (got.cpu() - ref).abs()
    # bilinear weights see ~1e-5 px of coordinate rounding between the two float32 evaluations; pixel values are in [0, 1]
    assert float(err.max()) <= 2e-4 and float(err.mean()) <= 2e-6, (float(err.max()), float(err.mean()))
    # affine off: only flip + shift + clip, bit-exact
    aug.do_affine = False
    got0 = aug.apply({task: torch.from_numpy(u8).to(DEV)}, flips, affs, shifts)[task]
    ref0 = _oracle_batch(A, D, u8, task, flips, affs, shifts, do_affine=False)
    assert torch.equal(got0.cpu(), ref0)


def test_label_augmentation_matches_oracle():
    from mirage_amd import data as D
    from oracle import augment_oracle as A
    import types
    rng = np.random.default_rng(12)
    B, H, W = 6, 128, 128
    u8 = rng.integers(0, 13, (B, H, W), dtype=np.uint8)
    # piecewise-constant maps (as real layer maps are): constant rows per band
    u8 = np.sort(u8, axis=1)
    aug = D.DataAugmentationForMIRAGE(types.SimpleNamespace(hflip=0.5, intensity_shift=0.1, affine=True), seed=4)
    flips, affs, shifts = aug.draw(["bscanlayermap"], B)
    got = aug.apply({"bscanlayermap": torch.from_numpy(u8).to(DEV)}, flips, affs, shifts)["bscanlayermap"]
    assert got.shape == (B, H, W) and got.dtype == torch.int64
    ref = _oracle_batch(A, D, u8, "bscanlayermap", flips, affs, shifts)
    diff = (got.cpu() != ref)
    # integer outputs can differ only where the interpolated value sits on a rounding tie (x.5 +- float noise)
    assert float(diff.float().mean()) <= 1e-3, float(diff.float().mean())
    assert int((got.cpu() - ref).abs().max()) <= 1
    assert int(got.min()) >= 0 and int(got.max()) <= 12


def test_whole_batch_dict_and_errors():
    from mirage_amd import data as D
    import types
    args = types.SimpleNamespace(hflip=0.5, intensity_shift=0.1, affine=True, input_size={"bscan": (64, 64), "slo": (64, 64), "bscanlayermap": (16, 16)})
    aug = D.DataAugmentationForMIRAGE(args, seed=1)
    g = torch.Generator().manual_seed(0)
    batch = {"bscan": torch.randint(0, 256, (4, 64, 64), generator=g, dtype=torch.uint8).to(DEV),
             "slo": torch.randint(0, 256, (4, 64, 64), generator=g, dtype=torch.uint8).to(DEV),
             "bscanlayermap": torch.randint(0, 13, (4, 16, 16), generator=g, dtype=torch.uint8).to(DEV)}
    out = aug(batch)
    assert out["bscan"].shape == (4, 1, 64, 64) and out["slo"].shape == (4, 1, 64, 64) and out["bscanlayermap"].shape == (4, 16, 16)
    assert 0.0 <= float(out["bscan"].min()) and float(out["bscan"].max()) <= 1.0
    with pytest.raises(NotImplementedError):
        aug({"bscan": torch.zeros(2, 32, 32, dtype=torch.uint8, device=DEV)})
    with pytest.raises(ValueError):
        aug({"bscan": torch.zeros(2, 64, 64, dtype=torch.float32, device=DEV)})


def test_driver_trains_from_npy_folder_with_gpu_augmentation(tmp_path):
    """run_pretraining over <root>/<domain>/*.npy (uint8 images, integer layer maps; docs/pretraining.md layout) with the
    reference's augmentation flags: bytes go to the GPU, DataAugmentationForMIRAGE runs there, the step trains."""
    import json, os
    from mirage_amd import run_pretraining as R
    rng = np.random.default_rng(0)
    root = tmp_path / "data"
    # 128/8 images and the (fixed) 128/8 layer map: every task has a 16x16 token grid, as the decoders require
    for d, (hw, hi) in {"bscan": (128, 256), "slo": (128, 256), "bscanlayermap": (128, 13)}.items():
        os.makedirs(root / d)
        for i in range(8):
            np.save(root / d / f"s{i:03d}.npy", rng.integers(0, hi, (hw, hw), dtype=np.uint8))
    out = str(tmp_path / "out")
    argv = ["--data_path", str(root), "--in_domains", "bscan-slo-bscanlayermap", "--out_domains", "bscan-slo-bscanlayermap",
            "--input_size", "128", "--patch_size", "8", "--batch_size", "4", "--num_encoded_tokens", "98", "--epochs", "2",
            "--warmup_epochs", "1", "--save_ckpt_freq", "1", "--base_output_dir", out, "--print_freq", "1", "--no_graph",
            "--hflip", "0.5", "--intensity_shift", "0.1", "--affine"]
    args = R.get_args(argv)
    assert args.affine and args.hflip == 0.5 and args.intensity_shift == 0.1
    R.main(args)
    lines = [json.loads(l) for l in open(os.path.join(out, "miragepre_base", "log.txt"))]
    assert len(lines) == 2 and all(np.isfinite(l["[Epoch] loss"]) for l in lines)
