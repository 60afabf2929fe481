"""Pretraining data path: uint8 .npy samples go to the GPU as bytes (4x less PCIe than fp32) and the
whole of DataAugmentationForMIRAGE.__call__ (reference mutils/datasets_pretrain.py:18-83) -- horizontal flip, intensity
shift + clip, the shared random affine (full for bscan / bscanlayermap, x-translation + scale for the other tasks), bilinear
resampling with zero fill, label rounding -- runs as one fused HIP pass per task (csrc/augment.hip).

The random parameters are drawn on the host per sample in the reference's order of decisions (flip, then per image task the
shift magnitude and its sign, then the affine parameters) from one numpy Generator per rank; the reference's own streams
(`random`, `np.random`, torch, one per DataLoader worker) are not reproducible across worker layouts, so only the
distributions and the parameter -> pixels map are part of the contract."""
import math
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch

from . import kernels as K

LABEL_TASKS = ("layermaps", "bscanlayermap")          # datasets_pretrain.py:43,65
FULL_AFFINE_TASKS = ("bscan", "bscanlayermap")        # datasets_pretrain.py:51


def inverse_affine_matrix(angle: float, translate: Sequence[float], scale: float, shear: Sequence[float]) -> Tuple[float, ...]:
    """Output->input matrix of TF.affine for tensors with center=None (datasets_pretrain.py:57-64): torchvision 0.20.1
    transforms.functional._get_inverse_affine_matrix in centre-relative pixel coordinates."""
    rot = math.radians(angle)
    sx, sy = math.radians(shear[0]), math.radians(shear[1])
    tx, ty = float(translate[0]), float(translate[1])
    a = math.cos(rot - sy) / math.cos(sy)
    b = -math.cos(rot - sy) * math.tan(sx) / math.cos(sy) - math.sin(rot)
    c = math.sin(rot - sy) / math.cos(sy)
    d = -math.sin(rot - sy) * math.tan(sx) / math.cos(sy) + math.cos(rot)
    m = [d / scale, -b / scale, 0.0, -c / scale, a / scale, 0.0]
    m[2] += m[0] * (-tx) + m[1] * (-ty)
    m[5] += m[3] * (-tx) + m[4] * (-ty)
    return tuple(m)


class RandomAffineParams:
    """transforms.RandomAffine(degrees=10, translate=(0.1, 0.1), scale=(0.9, 1.1), shear=5).get_params(..., [512, 512])
    (datasets_pretrain.py:24-31,37-40): angle U(-10,10) deg; tx, ty = round(U(-51.2, 51.2)) pixels -- always relative to
    512, also for the 128x128 layer map; scale U(0.9,1.1); x-shear U(-5,5) deg, y-shear 0."""

    def __init__(self, degrees=10.0, translate=(0.1, 0.1), scale=(0.9, 1.1), shear=5.0, img_size=(512, 512)):
        self.degrees, self.translate, self.scale, self.shear, self.img_size = degrees, translate, scale, shear, img_size

    def sample(self, rng: np.random.Generator):
        angle = float(rng.uniform(-self.degrees, self.degrees))
        max_dx, max_dy = float(self.translate[0] * self.img_size[0]), float(self.translate[1] * self.img_size[1])
        tx = int(round(float(rng.uniform(-max_dx, max_dx))))
        ty = int(round(float(rng.uniform(-max_dy, max_dy))))
        sc = float(rng.uniform(self.scale[0], self.scale[1]))
        return angle, (tx, ty), sc, (float(rng.uniform(-self.shear, self.shear)), 0.0)


def task_affine_params(task: str, params):
    """datasets_pretrain.py:51-56."""
    angle, translate, sc, shear = params
    if task in FULL_AFFINE_TASKS:
        return angle, translate, sc, shear
    return 0.0, (translate[0], 0), sc, (0.0, 0.0)


def kernel_params(task: str, H: int, W: int, flip: bool, shift: float, affine_params) -> np.ndarray:
    """The 8 floats mrg_augment_* takes per sample: theta / (0.5 W, 0.5 H) in fp32 as _gen_affine_grid computes it
    (fp32 matrix, fp32 division), shift, flip."""
    m = np.asarray(inverse_affine_matrix(*task_affine_params(task, affine_params)), dtype=np.float32)
    half = np.asarray([0.5 * W, 0.5 * H], dtype=np.float32)
    out = np.empty(8, dtype=np.float32)
    out[0:3] = m[0:3] / half[0]
    out[3:6] = m[3:6] / half[1]
    out[6] = 0.0 if task in LABEL_TASKS else shift
    out[7] = 1.0 if flip else 0.0
    return out


class DataAugmentationForMIRAGE:
    """Batch-level GPU counterpart of the reference class of the same name (datasets_pretrain.py:18-83).  args fields used:
    hflip (probability), intensity_shift (sigma), affine (bool), input_size (per-task target size; raw samples of another size
    are resized after the affine step, datasets_pretrain.py:66-81)."""

    def __init__(self, args, seed: int = 0):
        self.args = args
        self.hflip = float(getattr(args, "hflip", 0.5))
        self.intensity_shift = float(getattr(args, "intensity_shift", 0.1))
        self.do_affine = bool(getattr(args, "affine", True))
        self.affine = RandomAffineParams()
        self.rng = np.random.default_rng(seed)

    def draw(self, tasks: Sequence[str], B: int):
        """Per-sample decisions in the reference's order (:34-48): flip, affine parameters, then per image task the shift
        magnitude |N(0, sigma)| and its sign."""
        flips, affs, shifts = [], [], {t: np.zeros(B, dtype=np.float32) for t in tasks}
        for b in range(B):
            flips.append(bool(self.rng.random() < self.hflip))
            affs.append(self.affine.sample(self.rng))
            for t in tasks:
                if self.intensity_shift > 0 and t not in LABEL_TASKS:
                    s = np.float32(self.rng.normal(0.0, self.intensity_shift))
                    if self.rng.random() < 0.5:
                        s = -s
                    shifts[t][b] = s
        return flips, affs, shifts

    def apply(self, batch_u8: Dict[str, torch.Tensor], flips, affs, shifts) -> Dict[str, torch.Tensor]:
        out = {}
        for t, src in batch_u8.items():
            if src.dtype != torch.uint8 or src.dim() != 3:
                raise ValueError(f"mirage_amd.data: task {t!r} must be a uint8 [B, H, W] tensor, got {src.dtype} {tuple(src.shape)}")
            B, H, W = src.shape
            want = getattr(self.args, "input_size", {}).get(t) if isinstance(getattr(self.args, "input_size", None), dict) else None
            prm = np.stack([kernel_params(t, H, W, flips[b], float(shifts[t][b]), affs[b]) for b in range(B)])
            prm = torch.from_numpy(prm).to(src.device, non_blocking=True)
            if t in LABEL_TASKS:
                y = K.augment_label(src, prm, self.do_affine)
                if want is not None and tuple(want) != (H, W):          # datasets_pretrain.py:66-81: NEAREST for label maps
                    y = K.resize_label(y, want)
                out[t] = y
            else:
                y = K.augment_image(src, prm, self.do_affine)
                if want is not None and tuple(want) != (H, W):          # BILINEAR (antialiased, as torchvision resizes tensors)
                    y = K.resize_image(y, want)
                out[t] = y.unsqueeze(1)
        return out

    def __call__(self, batch_u8: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        B = next(iter(batch_u8.values())).shape[0]
        return self.apply(batch_u8, *self.draw(list(batch_u8.keys()), B))


class PrefetchingNpyLoader:
    """Asynchronous feed of <root>/<task>/*.npy samples (the reference's DataLoader settings,
    run_pretraining.py:478-487: shuffle, drop_last, prefetch_factor 2, pinned memory, persistent workers).

    Pipeline per batch (all stages overlap the training step of the previous batch):
      reader threads (np.load releases the GIL; `num_workers` of them, created before anything touches the GPU is not required --
      they never call into HIP) fill one slot of a ring of PINNED uint8 staging buffers [B, H, W] per task
      -> the consumer enqueues the H2D copies (bytes: 4x less PCIe than fp32) and the augmentation kernels on a dedicated copy
         stream into fresh device tensors and records an event
      -> the training stream waits on that event only when it takes the batch (one step later).
    The host never blocks on the GPU except when a staging slot is about to be refilled while its copy is still in flight.
    Samples are matched across tasks by sorted file order (mutils/dataset_folder.py:32-51); rank r of `world` takes every
    world-th batch of the epoch's permutation (the reference is single-process)."""

    def __init__(self, files: Dict[str, Sequence[str]], batch_size: int, device, augment: Optional[DataAugmentationForMIRAGE] = None,
                 rank: int = 0, world: int = 1, seed: int = 0, num_workers: int = 8, prefetch: int = 2, shuffle: bool = True):
        import concurrent.futures
        self.files = {t: list(v) for t, v in files.items()}
        self.tasks = list(self.files.keys())
        n = min(len(v) for v in self.files.values())
        if n <= 0:
            raise ValueError("PrefetchingNpyLoader: no samples")
        self.n, self.B, self.device, self.augment = n, int(batch_size), torch.device(device), augment
        self.rank, self.world, self.shuffle = rank, world, shuffle
        self.rng = np.random.default_rng(seed)
        self.steps = n // (self.B * world)                               # drop_last
        self.prefetch = max(1, int(prefetch))
        self.pool = concurrent.futures.ThreadPoolExecutor(max_workers=max(1, int(num_workers)), thread_name_prefix="mrg-npy")
        # one assembler per batch in preparation: it fans the B x n_tasks reads out over the reader pool and waits for them
        self.assemblers = concurrent.futures.ThreadPoolExecutor(max_workers=self.prefetch + 1, thread_name_prefix="mrg-batch")
        self.copy_stream = torch.cuda.Stream(device=self.device)
        first = {t: np.load(self.files[t][0]) for t in self.tasks}
        self.shapes = {t: tuple(a.shape[-2:]) for t, a in first.items()}
        # ring of pinned staging slots: prefetch batches being filled / queued + one being copied
        self.slots = [{t: torch.empty((self.B,) + self.shapes[t], dtype=torch.uint8).pin_memory() for t in self.tasks}
                      for _ in range(self.prefetch + 1)]
        self.slot_free = [None] * len(self.slots)                         # event: the H2D copy out of this slot has completed

    def __len__(self):
        return self.steps

    def _fill(self, slot: int, ids) -> int:
        """Reader side: B x n_tasks np.load calls spread over the pool, straight into the pinned slot."""
        ev = self.slot_free[slot]
        if ev is not None:
            ev.synchronize()                                              # the previous copy out of this slot must be done
        bufs = {t: self.slots[slot][t].numpy() for t in self.tasks}

        def one(t, b, i):
            a = np.load(self.files[t][i])
            if a.shape[-2:] != self.shapes[t] or a.size != self.shapes[t][0] * self.shapes[t][1]:
                raise ValueError(f"{self.files[t][i]}: shape {a.shape} differs from the first sample's {self.shapes[t]} (one [H, W] plane per file)")
            # the staging slots are uint8 (docs/pretraining.md: 8-bit images, layer ids 0..12): anything else would wrap silently
            if a.dtype != np.uint8:
                if not np.issubdtype(a.dtype, np.integer):
                    raise ValueError(f"{self.files[t][i]}: dtype {a.dtype}; the asynchronous feed stages 8-bit integers (uint8 images / label maps)")
                if a.size and (int(a.min()) < 0 or int(a.max()) > 255):
                    raise ValueError(f"{self.files[t][i]}: values outside 0..255 cannot be staged as uint8")
            bufs[t][b] = a.reshape(self.shapes[t]).astype(np.uint8, copy=False)

        futs = [self.pool.submit(one, t, b, int(i)) for t in self.tasks for b, i in enumerate(ids)]
        for f in futs:
            f.result()
        return slot

    def _upload(self, slot: int):
        """Consumer side: H2D + augmentation on the copy stream; returns (batch, ready event)."""
        with torch.cuda.stream(self.copy_stream):
            dev = {t: self.slots[slot][t].to(self.device, non_blocking=True) for t in self.tasks}
            done = torch.cuda.Event()
            done.record(self.copy_stream)
            self.slot_free[slot] = done
            if self.augment is not None:
                batch = self.augment(dev)
            else:
                batch = {t: (v.long() if t in LABEL_TASKS else (v.float() / 255.0).unsqueeze(1)) for t, v in dev.items()}
            ready = torch.cuda.Event()
            ready.record(self.copy_stream)
        return batch, ready

    def __iter__(self):
        import collections
        perm = self.rng.permutation(self.n) if self.shuffle else np.arange(self.n)
        B, W, R = self.B, self.world, self.rank
        batches = [perm[(s * W + R) * B:(s * W + R + 1) * B] for s in range(self.steps)]
        inflight = collections.deque()                                   # futures of slots being filled, in batch order
        nxt = 0

        def submit():
            nonlocal nxt
            if nxt < len(batches):
                slot = nxt % len(self.slots)
                inflight.append(self.assemblers.submit(self._fill, slot, batches[nxt]))
                nxt += 1

        for _ in range(min(self.prefetch, len(batches))):
            submit()
        pending = None
        for _ in range(len(batches)):
            slot = inflight.popleft().result()
            cur = self._upload(slot)                                      # enqueued before the previous batch's step is launched
            submit()
            if pending is not None:
                yield self._take(pending)
            pending = cur
        if pending is not None:
            yield self._take(pending)

    def _take(self, item):
        batch, ready = item
        main = torch.cuda.current_stream(self.device)
        main.wait_event(ready)
        for v in batch.values():
            v.record_stream(main)                                         # allocated on the copy stream, consumed on this one
        return batch

    def close(self):
        self.assemblers.shutdown(wait=False, cancel_futures=True)
        self.pool.shutdown(wait=False, cancel_futures=True)
