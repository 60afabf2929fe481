"""TEST INFRASTRUCTURE ONLY -- CPU oracle of the pretraining augmentation.

Restates DataAugmentationForMIRAGE.__call__ (reference mutils/datasets_pretrain.py:33-81) for one sample with the random
draws made explicit.  The geometric part of the reference lives in a third-party dependency that is NOT in
/root/reference and NOT installed here: torchvision==0.20.1 (reference requirements.txt:56).  Its published tensor-path
algorithm is restated below, function by function:
  * transforms.RandomAffine.get_params           -> `sample_affine_params`
  * transforms.functional._get_inverse_affine_matrix, functional.affine (tensor branch: centre-relative coordinates)
                                                  -> `inverse_affine_matrix`
  * _functional_tensor._gen_affine_grid           -> `affine_grid`
  * _functional_tensor._apply_grid_transform (ones-channel fill blend), _cast_squeeze_in/_out (round for integer dtypes)
                                                  -> `apply_grid`
The sampling step itself is torch.nn.functional.grid_sample (aten, present here), i.e. the same code torchvision calls.
PARITY UNPINNED against torchvision itself (absent); pinned against grid_sample, against a float64 numpy bilinear
restatement (`affine_numpy`, small cases) and against closed-form cases (identity, integer translation, flip) in
tests/test_augment_cpu.py.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this module."""
import math
from typing import Dict, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

LABEL_TASKS = ("layermaps", "bscanlayermap")          # datasets_pretrain.py:43,65
FULL_AFFINE_TASKS = ("bscan", "bscanlayermap")        # datasets_pretrain.py:51


def sample_affine_params(rng: np.random.Generator, degrees=10.0, translate=(0.1, 0.1), scale=(0.9, 1.1), shear=5.0,
                         img_size=(512, 512)):
    """RandomAffine.get_params with RandomAffine(degrees=10, translate=(.1,.1), scale=(.9,1.1), shear=5) as constructed at
    datasets_pretrain.py:24-31 and called with img_size [512, 512] for EVERY task (:37-40, a reference quirk: the 128x128
    layer map gets the B-scan's pixel translation).  shear=5 means x-shear in (-5, 5), y-shear 0."""
    angle = float(rng.uniform(-degrees, degrees))
    max_dx, max_dy = float(translate[0] * img_size[0]), float(translate[1] * img_size[1])
    tx = int(round(float(rng.uniform(-max_dx, max_dx))))
    ty = int(round(float(rng.uniform(-max_dy, max_dy))))
    sc = float(rng.uniform(scale[0], scale[1]))
    shear_x = float(rng.uniform(-shear, shear))
    return angle, (tx, ty), sc, (shear_x, 0.0)


def task_affine_params(task: str, params):
    """datasets_pretrain.py:51-56: B-scan and its layer map get the full transform, every other task only the x translation
    and the scale."""
    angle, translate, sc, shear = params
    if task in FULL_AFFINE_TASKS:
        return angle, translate, sc, shear
    return 0.0, (translate[0], 0), sc, (0.0, 0.0)


def inverse_affine_matrix(angle: float, translate: Sequence[float], scale: float, shear: Sequence[float],
                          center=(0.0, 0.0)) -> Tuple[float, ...]:
    """torchvision.transforms.functional._get_inverse_affine_matrix(inverted=True); for tensors functional.affine passes
    centre-relative coordinates, i.e. center = (0, 0) when the user gives none (datasets_pretrain.py:63 passes
    affine.center = None).  Returns the 6 entries of the output->input matrix."""
    rot = math.radians(angle)
    sx, sy = math.radians(shear[0]), math.radians(shear[1])
    cx, cy = center
    tx, ty = float(translate[0]), float(translate[1])
    a = math.cos(rot - sy) / math.cos(sy)
    b = -math.cos(rot - sy) * math.tan(sx) / math.cos(sy) - math.sin(rot)
    c = math.sin(rot - sy) / math.cos(sy)
    d = -math.sin(rot - sy) * math.tan(sx) / math.cos(sy) + math.cos(rot)
    m = [d, -b, 0.0, -c, a, 0.0]
    m = [v / scale for v in m]
    m[2] += m[0] * (-cx - tx) + m[1] * (-cy - ty)
    m[5] += m[3] * (-cx - tx) + m[4] * (-cy - ty)
    m[2] += cx
    m[5] += cy
    return tuple(m)


def rescaled_theta(matrix: Sequence[float], W: int, H: int) -> torch.Tensor:
    """theta.transpose(1, 2) / [0.5 w, 0.5 h] of _gen_affine_grid, returned as the fp32 [2, 3] the kernel consumes."""
    theta = torch.tensor(matrix, dtype=torch.float32).reshape(1, 2, 3)
    r = theta.transpose(1, 2) / torch.tensor([0.5 * W, 0.5 * H], dtype=torch.float32)        # [1, 3, 2]
    return r[0].t().contiguous()


def affine_grid(matrix: Sequence[float], W: int, H: int) -> torch.Tensor:
    """_gen_affine_grid(theta, w, h, ow=w, oh=h) -> [1, H, W, 2] normalised sampling grid."""
    d = 0.5
    base = torch.empty(1, H, W, 3, dtype=torch.float32)
    base[..., 0].copy_(torch.linspace(-W * 0.5 + d, W * 0.5 + d - 1, steps=W))
    base[..., 1].copy_(torch.linspace(-H * 0.5 + d, H * 0.5 + d - 1, steps=H).unsqueeze_(-1))
    base[..., 2].fill_(1)
    theta = torch.tensor(matrix, dtype=torch.float32).reshape(1, 2, 3)
    r = theta.transpose(1, 2) / torch.tensor([0.5 * W, 0.5 * H], dtype=torch.float32)
    return base.view(1, H * W, 3).bmm(r).view(1, H, W, 2)


def apply_grid(img: torch.Tensor, grid: torch.Tensor, fill: float = 0.0) -> torch.Tensor:
    """_apply_grid_transform(img [1, H, W], grid, 'bilinear', fill) including the integer round-trip of _cast_squeeze_*."""
    out_dtype = img.dtype
    x = img.to(torch.float32).unsqueeze(0)                                   # [1, 1, H, W]
    x = torch.cat((x, torch.ones_like(x)), dim=1)
    x = F.grid_sample(x, grid, mode="bilinear", padding_mode="zeros", align_corners=False)
    mask = x[:, -1:, :, :]
    x = x[:, :-1, :, :]
    x = x * mask + (1.0 - mask) * fill
    x = x.squeeze(0)
    if out_dtype in (torch.uint8, torch.int8, torch.int16, torch.int32, torch.int64):
        x = torch.round(x)
    return x.to(out_dtype)


def augment_sample(sample: Dict[str, np.ndarray], flip: bool, shifts: Dict[str, float], affine_params, do_affine: bool = True
                   ) -> Dict[str, torch.Tensor]:
    """One __call__ of DataAugmentationForMIRAGE on already-loaded arrays (image tasks float32 in [0,1], label tasks int64),
    samples already at args.input_size (the TF.resize branch :70-75 is not taken).  `shifts[task]` is the signed shift
    (:45-47), 0 for none.  Returns image tasks as [1, H, W] fp32, label tasks as [H, W] int64 (:76-78)."""
    out = {}
    for task, arr in sample.items():
        if flip:
            arr = np.flip(arr, axis=-1)
        if task not in LABEL_TASKS and shifts.get(task, 0.0) != 0.0:
            arr = np.clip(arr + np.asarray([shifts[task]], dtype=np.float32), 0, 1)
        img = torch.from_numpy(arr.copy()).contiguous().unsqueeze(0)
        if do_affine:
            H, W = img.shape[-2:]
            a, t, s, sh = task_affine_params(task, affine_params)
            img = apply_grid(img, affine_grid(inverse_affine_matrix(a, t, s, sh), W, H))
        if task in LABEL_TASKS:
            img = img.squeeze(0)
        out[task] = img
    return out


def affine_numpy(img: np.ndarray, matrix: Sequence[float]) -> np.ndarray:
    """Independent float64 restatement of the same resampling (centre-relative inverse map, bilinear, zero padding, mask
    blend with fill 0) with explicit loops: small cases only."""
    H, W = img.shape
    out = np.zeros((H, W), dtype=np.float64)
    m = [float(v) for v in matrix]
    for i in range(H):
        for j in range(W):
            x, y = j + 0.5 - W / 2, i + 0.5 - H / 2
            ix = m[0] * x + m[1] * y + m[2] + W / 2 - 0.5
            iy = m[3] * x + m[4] * y + m[5] + H / 2 - 0.5
            x0, y0 = math.floor(ix), math.floor(iy)
            val = mask = 0.0
            for (yy, xx, w) in ((y0, x0, (x0 + 1 - ix) * (y0 + 1 - iy)), (y0, x0 + 1, (ix - x0) * (y0 + 1 - iy)),
                                (y0 + 1, x0, (x0 + 1 - ix) * (iy - y0)), (y0 + 1, x0 + 1, (ix - x0) * (iy - y0))):
                if 0 <= yy < H and 0 <= xx < W:
                    val += w * float(img[yy, xx]); mask += w
            out[i, j] = val * mask
    return out


def resize_sample(img: torch.Tensor, size, label: bool) -> torch.Tensor:
    """TF.resize branch (mutils/datasets_pretrain.py:66-81): after the affine step a sample whose size differs from
    args.input_size[task] is resized -- label maps with InterpolationMode.NEAREST, images with BILINEAR.  For tensors torchvision
    0.20.1 (_functional_tensor.resize) calls torch.nn.functional.interpolate(size=..., mode=..., align_corners=False [bilinear],
    antialias=True [bilinear; the default since 0.17]); integer inputs are cast to float for the call and back.  img: [H, W]."""
    x = img[None, None].float()
    if label:
        return torch.nn.functional.interpolate(x, size=tuple(size), mode="nearest")[0, 0].to(img.dtype)
    return torch.nn.functional.interpolate(x, size=tuple(size), mode="bilinear", align_corners=False, antialias=True)[0, 0]
