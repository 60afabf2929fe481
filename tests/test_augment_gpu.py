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
    small = aug({"bscan": torch.zeros(2, 32, 32, dtype=torch.uint8, device=DEV)})       # raw size != input_size: the TF.resize branch
    assert small["bscan"].shape == (2, 1, 64, 64)
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


def test_graph_training_from_the_loader_reads_batches_in_place(tmp_path, monkeypatch):
    """The captured step consumes the loader's batches where its augmentation kernel wrote them (input slots; the tensors are
    allocated on the loader's copy stream and handed over with record_stream): 3 epochs x 4 steps from a .npy folder with hipGraphs give
    exactly the log of the same run through the staging copy (MRG_INPUT_SLOTS=0)."""
    import json, os
    from mirage_amd import run_pretraining as R, trainer
    rng = np.random.default_rng(1)
    root = tmp_path / "data"
    for d, (hw, hi) in {"bscan": (128, 256), "slo": (128, 256), "bscanlayermap": (128, 13)}.items():
        os.makedirs(root / d)
        for i in range(16):
            np.save(root / d / f"s{i:03d}.npy", rng.integers(0, hi, (hw, hw), dtype=np.uint8))

    def run(tag, in_place):
        monkeypatch.setattr(trainer, "IN_PLACE_INPUTS", in_place)
        out = str(tmp_path / tag)
        argv = ["--data_path", str(root), "--in_domains", "bscan-slo-bscanlayermap", "--out_domains", "bscan-slo-bscanlayermap",
                "--input_size", "128", "--patch_size", "8", "--batch_size", "4", "--num_encoded_tokens", "98", "--epochs", "3",
                "--warmup_epochs", "1", "--save_ckpt_freq", "3", "--base_output_dir", out, "--print_freq", "1", "--seed", "5",
                "--hflip", "0.5", "--intensity_shift", "0.1", "--affine"]
        R.main(R.get_args(argv))
        return [json.loads(l) for l in open(os.path.join(out, "miragepre_base", "log.txt"))]
    a, b = run("slots", True), run("copy", False)
    assert len(a) == 3 and all(np.isfinite(l["[Epoch] loss"]) for l in a)
    keys = [k for k in a[0] if "loss" in k or "grad_norm" in k]
    assert keys and all(la[k] == lb[k] for la, lb in zip(a, b) for k in keys), (a, b)
    assert a[-1]["[Epoch] loss"] < a[0]["[Epoch] loss"]


def test_resize_branch_matches_aten():
    """TF.resize branch (mutils/datasets_pretrain.py:66-81): torchvision resizes tensors through aten's interpolate -- bilinear with
    antialias=True for images, nearest for label maps.  The HIP kernels against torch's CPU kernels of the same ops, down- and
    up-scaling, non-integer ratios."""
    from mirage_amd import kernels as K
    from oracle import augment_oracle as A
    g = torch.Generator().manual_seed(5)
    for (hi, wi), (ho, wo) in (((512, 512), (128, 128)), ((96, 160), (128, 128)), ((300, 200), (224, 224)), ((64, 64), (512, 512))):
        x = torch.rand(3, hi, wi, generator=g)
        want = torch.stack([A.resize_sample(x[b], (ho, wo), label=False) for b in range(3)])
        got = K.resize_image(x.to(DEV), (ho, wo)).cpu()
        torch.testing.assert_close(got, want, rtol=0, atol=2e-6)
        lab = torch.randint(0, 13, (3, hi, wi), generator=g)
        wantl = torch.stack([A.resize_sample(lab[b], (ho, wo), label=True) for b in range(3)])
        gotl = K.resize_label(lab.to(DEV), (ho, wo)).cpu()
        assert torch.equal(gotl, wantl)


def test_prefetching_loader_matches_synchronous_reads(tmp_path):
    """Asynchronous feed (reader threads -> pinned staging -> copy stream, run_pretraining.py:478-487 semantics): without
    augmentation and shuffling every batch equals the directly loaded files (drop_last, rank sharding, two epochs, ring reuse);
    with raw samples of another size the resize branch brings them to args.input_size; shuffled epochs are permutations."""
    import types
    from mirage_amd.data import PrefetchingNpyLoader, DataAugmentationForMIRAGE
    rng = np.random.default_rng(3)
    root = tmp_path / "d"
    n, B = 23, 4
    files = {}
    for d, (hw, hi) in {"bscan": (64, 256), "bscanlayermap": (32, 13)}.items():
        (root / d).mkdir(parents=True)
        for i in range(n):
            np.save(root / d / f"s{i:03d}.npy", rng.integers(0, hi, (hw, hw), dtype=np.uint8))
        files[d] = sorted(str(p) for p in (root / d).glob("*.npy"))
    for world in (1, 2):
        for rank in range(world):
            ld = PrefetchingNpyLoader(files, B, DEV, augment=None, rank=rank, world=world, seed=0, num_workers=4, prefetch=2, shuffle=False)
            assert len(ld) == n // (B * world)
            for epoch in range(2):
                got = list(ld)
                assert len(got) == len(ld)
                for s, b in enumerate(got):
                    ids = range((s * world + rank) * B, (s * world + rank + 1) * B)
                    img = np.stack([np.load(files["bscan"][i]) for i in ids]).astype(np.float32) / 255.0
                    lab = np.stack([np.load(files["bscanlayermap"][i]) for i in ids]).astype(np.int64)
                    assert b["bscan"].shape == (B, 1, 64, 64) and b["bscan"].dtype == torch.float32
                    np.testing.assert_allclose(b["bscan"][:, 0].cpu().numpy(), img, rtol=0, atol=1e-7)
                    assert np.array_equal(b["bscanlayermap"].cpu().numpy(), lab)
            ld.close()
    # shuffled: every epoch is a permutation of the first steps * B indices of a random order, different between epochs
    ld = PrefetchingNpyLoader(files, B, DEV, augment=None, seed=1, num_workers=4, shuffle=True)
    sums = []
    for epoch in range(2):
        vals = torch.cat([b["bscan"].flatten(1).sum(1) for b in ld]).cpu().numpy()
        assert len(vals) == (n // B) * B and len(np.unique(np.round(vals, 3))) == len(vals)
        sums.append(vals)
    assert not np.array_equal(sums[0], sums[1])
    ld.close()
    # augmentation on the copy stream + resize branch: raw 64x64 / 32x32 samples, model input 128x128 / 16x16
    args = types.SimpleNamespace(hflip=0.5, intensity_shift=0.1, affine=True, input_size={"bscan": (128, 128), "bscanlayermap": (16, 16)})
    ld = PrefetchingNpyLoader(files, B, DEV, augment=DataAugmentationForMIRAGE(args, seed=7), seed=0, num_workers=4)
    b = next(iter(ld))
    assert b["bscan"].shape == (B, 1, 128, 128) and b["bscanlayermap"].shape == (B, 16, 16)
    assert float(b["bscan"].min()) >= 0.0 and float(b["bscan"].max()) <= 1.0 + 1e-6
    assert int(b["bscanlayermap"].min()) >= 0 and int(b["bscanlayermap"].max()) <= 12
    ld.close()
