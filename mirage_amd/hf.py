"""Hugging-Face style packaging of the encoder (reference hf/mirage_hf.py:582-690 `MIRAGEWrapper`,
hf/helper_hf.py:9-23 `MIRAGEhf`): an encoder-only feature extractor -- every modality token unmasked, global token last,
no output adapters -- whose directory layout is what `PyTorchModelHubMixin.save_pretrained` writes: `config.json`
({input_size, patch_size, modalities, size}) + `model.safetensors` with the wrapper's state_dict (keys prefixed `model.`).
Released `MIRAGE-Base/Large` exports therefore load here unchanged, and exports written here load in the reference class."""
import json
import os
import types
from typing import Dict, Mapping, Optional, Union

import torch
import torch.nn as nn

from .input_adapters import PatchedInputAdapter
from .model import MIRAGEModel
from .utils import pair

_SIZES = {"base": dict(dim_tokens=768, depth=12, num_heads=12), "large": dict(dim_tokens=1024, depth=24, num_heads=16)}


class MIRAGEhf(nn.Module):
    def __init__(self, input_size=512, patch_size=32, modalities="bscan-slo", size="base", **dims):
        """`dims` (dim_tokens / depth / num_heads) overrides the size preset; used by the reduced-size tests only."""
        super().__init__()
        if size not in _SIZES and not dims:
            raise ValueError(f"Unknown model size: {size}")
        self.config = dict(input_size=input_size, patch_size=patch_size, modalities=modalities, size=size)
        self.size = size
        args = types.SimpleNamespace(num_global_tokens=1, drop_path=0.0, in_domains=modalities.split("-"))
        isz, psz = pair(input_size), pair(patch_size)
        args.patch_size = {d: psz for d in args.in_domains}
        args.input_size = {d: isz for d in args.in_domains}
        args.grid_sizes = {d: [isz[i] // psz[i] for i in range(2)] for d in args.in_domains}
        self.args = args
        input_adapters = {d: PatchedInputAdapter(num_channels=1, stride_level=1, patch_size_full=tuple(psz), image_size=isz)
                          for d in args.in_domains}
        kw = dict(_SIZES.get(size, {}))
        kw.update(dims)
        self.model = MIRAGEModel(args, input_adapters=input_adapters, output_adapters=None, num_global_tokens=1, drop_path_rate=0.0, **kw)

    def forward(self, x: Union[Dict[str, torch.Tensor], torch.Tensor], return_all_layers: bool = False):
        """Dict of [B, 1, H, W] tensors in [0, 1] -> encoder tokens [B, T + 1, D] (hf/mirage_hf.py:540-565)."""
        x = {"bscan": x} if isinstance(x, torch.Tensor) else x
        # MIRAGELight (hf/mirage_hf.py:510-538) concatenates every token in natural order; the pretraining model reaches the same
        # through all-visible task masks (a stable sort of zeros keeps the order)
        first = next(iter(x.values()))
        B = first.shape[0]
        masks = {d: torch.zeros((B, self.args.grid_sizes[d][0] * self.args.grid_sizes[d][1]), dtype=torch.long, device=first.device)
                 for d in self.args.in_domains if d in x}
        out = self.model(x, mask_inputs=False, task_masks=masks, return_all_layers=return_all_layers)
        return out if return_all_layers else out[0]

    def load_state_dict(self, state_dict: Mapping[str, torch.Tensor], strict: bool = True, assign: bool = False):
        """hf/mirage_hf.py:682-688: a bare model state_dict goes to the inner model; keys prefixed `model.` (the safetensors
        layout) are accepted too."""
        if state_dict and all(k.startswith("model.") for k in state_dict):
            state_dict = {k[len("model."):]: v for k, v in state_dict.items()}
        return self.model.load_state_dict(state_dict, strict)

    @property
    def device(self):
        return next(self.parameters()).device

    # -- PyTorchModelHubMixin-compatible directory layout ------------------------------------------------------------
    def save_pretrained(self, save_directory: str, config: Optional[dict] = None) -> str:
        from safetensors.torch import save_file
        os.makedirs(save_directory, exist_ok=True)
        with open(os.path.join(save_directory, "config.json"), "w") as f:
            json.dump(config if config is not None else self.config, f, indent=2, sort_keys=True)
        sd = {k: v.detach().cpu().contiguous() for k, v in self.state_dict().items()}
        save_file(sd, os.path.join(save_directory, "model.safetensors"), metadata={"format": "pt"})
        return save_directory

    @classmethod
    def from_pretrained(cls, directory: str, device: Union[str, torch.device] = "cuda", **dims) -> "MIRAGEhf":
        from safetensors.torch import load_file
        with open(os.path.join(directory, "config.json")) as f:
            cfg = json.load(f)
        m = cls(**{k: cfg[k] for k in ("input_size", "patch_size", "modalities", "size") if k in cfg}, **dims)
        msg = m.load_state_dict(load_file(os.path.join(directory, "model.safetensors")), strict=False)
        assert len(msg.missing_keys) == 0, msg.missing_keys
        return m.to(device)
