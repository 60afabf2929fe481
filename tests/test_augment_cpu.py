"""F2 data path, CPU side: the augmentation oracle (oracle/augment_oracle.py) against closed-form cases and an independent
float64 restatement, and the host-side parameter code of mirage_amd.data against the oracle.  torchvision (the reference's
dependency that owns this algorithm) is not installed, so there is no golden vector from it: parity is unpinned against
torchvision itself and pinned against aten grid_sample + the properties below."""
import numpy as np
import pytest
import torch

from oracle import augment_oracle as A


def _img(H, W, seed=0):
    return (np.random.default_rng(seed).integers(0, 256, (H, W)).astype(np.float32) / 255.0)


def test_inverse_matrix_known_answers():
    assert A.inverse_affine_matrix(0.0, (0, 0), 1.0, (0.0, 0.0)) == pytest.approx((1, 0, 0, 0, 1, 0))
    # pure translation by (+3, -2): output (x, y) samples input (x - 3, y + 2)
    assert A.inverse_affine_matrix(0.0, (3, -2), 1.0, (0.0, 0.0)) == pytest.approx((1, 0, -3, 0, 1, 2))
    # scale 2 about the centre: samples input at half the offset
    assert A.inverse_affine_matrix(0.0, (0, 0), 2.0, (0.0, 0.0)) == pytest.approx((0.5, 0, 0, 0, 0.5, 0))
    # 90 degrees: inverse rotation
    m = A.inverse_affine_matrix(90.0, (0, 0), 1.0, (0.0, 0.0))
    assert m == pytest.approx((0, 1, 0, -1, 0, 0), abs=1e-12)
    # forward * inverse = identity for a general draw
    ang, t, s, sh = 7.3, (11, -4), 1.07, (3.1, 0.0)
    inv = np.array(A.inverse_affine_matrix(ang, t, s, sh) + (0, 0, 1)).reshape(3, 3)
    import math
    rot, sx = math.radians(ang), math.radians(sh[0])
    fwd_rss = s * np.array([[math.cos(rot), -math.cos(rot) * math.tan(sx) - math.sin(rot)], [math.sin(rot), -math.sin(rot) * math.tan(sx) + math.cos(rot)]])
    fwd = np.eye(3); fwd[:2, :2] = fwd_rss; fwd[:2, 2] = t
    np.testing.assert_allclose(fwd @ inv, np.eye(3), atol=1e-12)


def test_identity_and_integer_translation_are_exact():
    img = _img(16, 24, 1)
    t = torch.from_numpy(img).unsqueeze(0)
    out = A.apply_grid(t, A.affine_grid(A.inverse_affine_matrix(0.0, (0, 0), 1.0, (0.0, 0.0)), 24, 16))
    np.testing.assert_allclose(out[0].numpy(), img, atol=1e-6)
    out = A.apply_grid(t, A.affine_grid(A.inverse_affine_matrix(0.0, (3, -2), 1.0, (0.0, 0.0)), 24, 16))[0].numpy()
    want = np.zeros_like(img)
    want[:-2, 3:] = img[2:, :-3]
    np.testing.assert_allclose(out, want, atol=1e-5)


def test_grid_sample_path_matches_float64_restatement():
    rng = np.random.default_rng(3)
    img = _img(20, 28, 2)
    for _ in range(4):
        p = A.sample_affine_params(rng, img_size=(28, 20))
        m = A.inverse_affine_matrix(*p)
        got = A.apply_grid(torch.from_numpy(img).unsqueeze(0), A.affine_grid(m, 28, 20))[0].numpy()
        np.testing.assert_allclose(got, A.affine_numpy(img, m), atol=2e-5)


def test_label_maps_are_bilinear_then_rounded_and_slo_only_translates_in_x():
    rng = np.random.default_rng(5)
    lab = rng.integers(0, 13, (16, 16)).astype(np.int64)
    params = (8.0, (2, 1), 1.05, (4.0, 0.0))
    out = A.augment_sample({"bscanlayermap": lab, "slo": _img(16, 16, 4), "bscan": _img(16, 16, 5)}, flip=True,
                           shifts={"slo": 0.07, "bscan": -0.2}, affine_params=params)
    assert out["bscanlayermap"].dtype == torch.int64 and out["bscanlayermap"].shape == (16, 16)
    m = A.inverse_affine_matrix(*params)
    want = np.rint(A.affine_numpy(np.flip(lab, -1).astype(np.float64), m))
    ties = np.abs(A.affine_numpy(np.flip(lab, -1).astype(np.float64), m) % 1.0 - 0.5) < 1e-4
    assert np.array_equal(out["bscanlayermap"].numpy()[~ties], want[~ties].astype(np.int64))
    # slo: angle 0, shear 0, translate (tx, 0), same scale
    assert A.task_affine_params("slo", params) == (0.0, (2, 0), 1.05, (0.0, 0.0))
    slo = np.clip(np.flip(_img(16, 16, 4), -1) + np.float32(0.07), 0, 1)
    np.testing.assert_allclose(out["slo"][0].numpy(), A.affine_numpy(slo, A.inverse_affine_matrix(0.0, (2, 0), 1.05, (0.0, 0.0))), atol=2e-5)
    assert out["bscan"].shape == (1, 16, 16) and float(out["bscan"].min()) >= 0.0 and float(out["bscan"].max()) <= 1.0


def test_host_parameter_code_matches_oracle():
    from mirage_amd import data as D
    rng1, rng2 = np.random.default_rng(9), np.random.default_rng(9)
    for _ in range(5):
        p1, p2 = D.RandomAffineParams().sample(rng1), A.sample_affine_params(rng2)
        assert p1 == p2
        assert -10 <= p1[0] <= 10 and abs(p1[1][0]) <= 52 and abs(p1[1][1]) <= 52 and 0.9 <= p1[2] <= 1.1 and abs(p1[3][0]) <= 5 and p1[3][1] == 0.0
        for task, (H, W) in (("bscan", (512, 512)), ("slo", (512, 512)), ("bscanlayermap", (128, 128))):
            assert D.inverse_affine_matrix(*D.task_affine_params(task, p1)) == pytest.approx(A.inverse_affine_matrix(*A.task_affine_params(task, p1)))
            kp = D.kernel_params(task, H, W, True, 0.05, p1)
            np.testing.assert_array_equal(kp[:6].reshape(2, 3), A.rescaled_theta(A.inverse_affine_matrix(*A.task_affine_params(task, p1)), W, H).numpy())
            assert kp[7] == 1.0 and kp[6] == (0.0 if task == "bscanlayermap" else np.float32(0.05))
