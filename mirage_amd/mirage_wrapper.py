"""Inference / linear-probe front ends on the MI355X engine.

Public surface = the reference's (`/root/reference/mirage_wrapper.py`): `MIRAGEWrapper(input_size, patch_size, modalities, weights,
device)` (:47-188) with `.args`, `.model`, `.device`, `get_model()`, `get_output_adapters()` and a dict-in / dict-out `forward`;
the three classification variants `MIRAGEClsGlobal` / `MIRAGEClsCLS` / `MIRAGEClsTokenMix` (:191-244) with `.norm`, `.head`,
`.embed_dim`, `.num_classes`, `pool()`, `build_head()`, registered in `miragecls_factory` under "global" / "cls" / "token_mix".
Everything behind that surface is this package's own: the model is assembled by `builder.get_model` (one constructor path for
training and inference), the three pooling rules are rows of one table, and LayerNorm / Linear of the head run on the HIP kernels.

`weights`: a reference-format checkpoint `{'model': state_dict, 'args': Namespace}` (mutils/checkpoint.py:14-20), written by the
reference or by `mirage_amd.run_pretraining`; `state=` passes the same dict in memory.
"""
import copy
from typing import Callable, Dict, Optional, Tuple, Union

import torch
import torch.nn as nn

from . import builder
from .factory import get_factory_adder
from .utils import LayerNorm, Linear, pair

_LAYERMAP = "bscanlayermap"
_LAYERMAP_GEOMETRY = ((128, 128), (8, 8))            # (image, patch) of the layer-map modality: fixed, whatever the image size
_MODEL_DIMS = {"large": dict(dim_tokens=1024, depth=24, num_heads=16), "base": {}}


def _inference_args(ckpt_args, modalities: str, input_size, patch_size):
    """The checkpoint's training namespace re-targeted at an inference geometry: input modalities from `modalities`, image / patch
    sizes from the caller (layer maps keep theirs), and the per-modality grids both spellings of which the callers read
    (`grid_size`: the wrapper's user-mask shape, image grid for every modality as in the reference; `grid_sizes`: the model's)."""
    a = copy.copy(ckpt_args)
    a.in_domains = [m for m in modalities.split("-") if m]
    if isinstance(getattr(a, "out_domains", None), str):
        a.out_domains = a.out_domains.split("-")
    img, pat = pair(input_size), pair(patch_size)
    if img is None or pat is None:
        raise ValueError("MIRAGEWrapper: input_size and patch_size are required")
    geometry = {m: (_LAYERMAP_GEOMETRY if m == _LAYERMAP else (img, pat)) for m in a.in_domains}
    a.input_size = {m: g[0] for m, g in geometry.items()}
    a.patch_size = {m: g[1] for m, g in geometry.items()}
    a.grid_size = {m: [s // p for s, p in zip(img, pat)] for m in a.in_domains}
    a.grid_sizes = {m: [s // p for s, p in zip(*geometry[m])] for m in a.in_domains}
    for name, default in (("num_global_tokens", 1), ("drop_path", 0.0), ("decoder_use_task_queries", True), ("decoder_use_xattn", True)):
        if not hasattr(a, name):
            setattr(a, name, default)
    return a


class MIRAGEWrapper(nn.Module):
    """Reconstruction front end: any subset of the trained modalities in, every output modality out."""

    def __init__(self, input_size=512, patch_size=32, modalities="bscan-slo-bscanlayermap", weights=None, device="cuda",
                 state: Optional[dict] = None):
        super().__init__()
        if weights is None and state is None:
            raise ValueError("MIRAGEWrapper: `weights` (checkpoint path) or `state` (loaded checkpoint) is required")
        ckpt = state if state is not None else torch.load(weights, map_location="cpu", weights_only=False)
        self.args = _inference_args(ckpt["args"], modalities, input_size, patch_size)
        self.model = self.get_model()
        report = self.model.load_state_dict(ckpt["model"], strict=False)
        if report.missing_keys:
            raise RuntimeError(f"MIRAGEWrapper: checkpoint lacks {len(report.missing_keys)} parameters, e.g. {report.missing_keys[:3]}")
        self.to(device)

    # -- construction hooks (subclasses switch the decoders off) ---------------------------------------
    def _wants_decoders(self) -> bool:
        return True

    def get_output_adapters(self) -> Union[None, dict]:
        return self.model.output_adapters if hasattr(self, "model") else None

    def get_model(self):
        name = str(getattr(self.args, "model", ""))
        dims = next((d for key, d in _MODEL_DIMS.items() if key in name), None)
        if dims is None:                                   # reduced-size models of the parity fixtures store their dims explicitly
            if not all(hasattr(self.args, k) for k in ("dim_tokens", "depth", "num_heads")):
                raise ValueError(f"Unknown model size: {name!r}")
            dims = dict(dim_tokens=self.args.dim_tokens, depth=self.args.depth, num_heads=self.args.num_heads)
        build_args = copy.copy(self.args)
        if not self._wants_decoders():
            build_args.out_domains = []
        # builder.get_model routes explicit dims through MIRAGEModel(...) and names through model_factory; Base passes its defaults
        return builder.get_model(build_args, with_decoders=self._wants_decoders(),
                                 **(dims or dict(dim_tokens=768, depth=12, num_heads=12)))

    # -- inference --------------------------------------------------------------------------------------
    def _blank(self, modality: str) -> torch.Tensor:
        h, w = self.args.input_size[modality]
        return torch.zeros((1, h, w), dtype=torch.long) if modality == _LAYERMAP else torch.zeros((1, 1, h, w))

    def forward(self, x: dict):
        """{modality: [1, C, H, W] in [0, 1] (layer maps: [1, H, W] integer ids)} -> {modality: reconstruction}.  A modality that is
        absent from `x` enters as an all-masked input (mirage_wrapper.py:150-179); like the reference's user-mask path
        (mirage/model.py:378-382, one visible-token count for the batch) this is meaningful for B = 1."""
        dev = self.device
        feed, masks = {}, {}
        for m in self.args.in_domains:
            present = m in x
            feed[m] = (x[m] if present else self._blank(m)).to(dev)
            gh, gw = self.args.grid_size[m]
            masks[m] = torch.full((1, gh * gw), 0 if present else 1, dtype=torch.long, device=dev)
        x.update(feed)                                     # the reference leaves the moved / filled tensors in the caller's dict
        preds, _ = self.model(feed, mask_inputs=False, task_masks=masks)
        return preds

    @property
    def device(self):
        return next(self.parameters()).device


add_miragecls, miragecls_factory = get_factory_adder()

# pooling rule -> (token groups that are averaged and concatenated, in order)
_POOLING: Dict[str, Tuple[str, ...]] = {"global": ("patch",), "cls": ("global",), "token_mix": ("patch", "global")}


def _token_groups(tokens: torch.Tensor, n_global: int) -> Dict[str, torch.Tensor]:
    """encoder output [B, T + n_global, D] (global tokens last, mirage/model.py:391) -> its two groups"""
    return {"patch": tokens[:, :tokens.shape[1] - n_global], "global": tokens[:, tokens.shape[1] - n_global:]}


class MIRAGEClsGlobal(MIRAGEWrapper):
    """Encoder over every token of ONE modality -> LayerNorm -> pooled feature -> Linear head (mirage_wrapper.py:191-228).
    `pooling` names a row of the table above; the subclasses only pick a different row."""
    pooling = "global"

    def __init__(self, num_classes=0, *args, **kwargs):
        super().__init__(*args, **kwargs)
        if num_classes <= 0:
            raise ValueError("num_classes must be positive")
        if len(self.args.in_domains) != 1:
            raise ValueError("the classification wrappers take exactly one input modality")
        self.num_classes = num_classes
        self.embed_dim = int(self.model.encoder[0].norm1.normalized_shape[0])
        self.norm = LayerNorm(self.embed_dim, eps=1e-06, elementwise_affine=True)
        self.build_head()
        self.to(self.device)

    def _wants_decoders(self) -> bool:
        return False

    def get_output_adapters(self):
        return None

    def build_head(self, factor: Optional[int] = None):
        width = len(_POOLING[self.pooling]) if factor is None else factor
        self.head = Linear(self.embed_dim * width, self.num_classes)

    def pool(self, x: torch.Tensor) -> torch.Tensor:
        groups = _token_groups(x, self.args.num_global_tokens)
        feats = [groups[g].mean(dim=1) for g in _POOLING[self.pooling]]
        return feats[0] if len(feats) == 1 else torch.cat(feats, dim=1)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        tokens, _ = self.model({self.args.in_domains[0]: x}, mask_inputs=False)
        return self.head(self.pool(self.norm(tokens)))


class MIRAGEClsCLS(MIRAGEClsGlobal):
    pooling = "cls"


class MIRAGEClsTokenMix(MIRAGEClsGlobal):
    pooling = "token_mix"


for _key, _cls in (("global", MIRAGEClsGlobal), ("cls", MIRAGEClsCLS), ("token_mix", MIRAGEClsTokenMix)):
    add_miragecls(_key)(_cls)
