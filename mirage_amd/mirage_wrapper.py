"""Inference / classification wrappers with the reference's names and behaviour (mirage_wrapper.py: MIRAGEWrapper :47-188,
MIRAGEClsGlobal :191-228, MIRAGEClsCLS :231-234, MIRAGEClsTokenMix :237-244) on the MI355X engine.

`weights` is a reference-format checkpoint: `{'model': state_dict, 'args': Namespace}` (mutils/checkpoint.py:14-20) as written
by the reference or by mirage_amd.run_pretraining.  The heads (LayerNorm + pooling + Linear) run on the HIP kernels too.
"""
from typing import Optional, Union

import numpy as np
import torch
import torch.nn as nn

from . import builder
from .factory import get_factory_adder
from .model import MIRAGEModel
from .utils import LayerNorm, Linear, pair


class MIRAGEWrapper(nn.Module):
    def __init__(self, input_size=512, patch_size=32, modalities="bscan-slo-bscanlayermap", weights=None, device="cuda",
                 state: Optional[dict] = None):
        super().__init__()
        assert weights is not None or state is not None
        state_dict = state if state is not None else torch.load(weights, map_location="cpu", weights_only=False)
        model_state_dict = state_dict["model"]
        args = state_dict["args"]
        args.in_domains = modalities.split("-")
        input_size, patch_size = pair(input_size), pair(patch_size)
        args.patch_size, args.input_size, args.grid_size = {}, {}, {}
        for domain in args.in_domains:
            if domain != "bscanlayermap":
                args.patch_size[domain], args.input_size[domain] = patch_size, input_size
            else:
                args.patch_size[domain], args.input_size[domain] = (8, 8), (128, 128)
            args.grid_size[domain] = [input_size[i] // patch_size[i] for i in range(len(input_size))]
        args.grid_sizes = {d: [args.input_size[d][i] // args.patch_size[d][i] for i in range(2)] for d in args.in_domains}
        if isinstance(getattr(args, "out_domains", None), str):
            args.out_domains = args.out_domains.split("-")
        self.args = args
        self.model = self.get_model()
        msg = self.model.load_state_dict(model_state_dict, strict=False)
        assert len(msg.missing_keys) == 0, msg.missing_keys
        self.to(device)

    def get_output_adapters(self) -> Union[None, dict]:
        a = self.args
        return {d: builder.DOMAIN_CONF[d]["output_adapter"](
            stride_level=builder.DOMAIN_CONF[d]["stride_level"], patch_size_full=tuple(a.patch_size[d]), dim_tokens=a.decoder_dim,
            depth=a.decoder_depth, num_heads=a.decoder_num_heads, use_task_queries=a.decoder_use_task_queries, task=d,
            context_tasks=list(a.in_domains), use_xattn=a.decoder_use_xattn, image_size=a.input_size[d]) for d in a.out_domains}

    def get_model(self):
        a = self.args
        input_adapters = {d: builder.DOMAIN_CONF[d]["input_adapter"](
            stride_level=builder.DOMAIN_CONF[d]["stride_level"], patch_size_full=tuple(a.patch_size[d]), image_size=a.input_size[d])
            for d in a.in_domains}
        output_adapters = self.get_output_adapters()
        kw = dict(args=a, input_adapters=input_adapters, output_adapters=output_adapters, num_global_tokens=a.num_global_tokens,
                  drop_path_rate=getattr(a, "drop_path", 0.0))
        if "large" in a.model:
            return MIRAGEModel(dim_tokens=1024, depth=24, num_heads=16, **kw)
        if "base" in a.model:
            return MIRAGEModel(**kw)
        # reduced-size models (tests): explicit dims stored next to the model name
        return MIRAGEModel(dim_tokens=a.dim_tokens, depth=a.depth, num_heads=a.num_heads, **kw)

    def forward(self, x: dict):
        """dict of [1, C, H, W] tensors in [0,1] (missing modalities are treated as fully masked) -> dict of reconstructions.
        As in the reference (mirage_wrapper.py:150-179 -> mirage/model.py:378-382) this path is meaningful for B = 1."""
        masks = {}
        for k in self.args.in_domains:
            if k not in x:
                x[k] = torch.zeros((1, *self.args.input_size[k])).long() if k == "bscanlayermap" else torch.zeros((1, 1, *self.args.input_size[k]))
                fill_v = 1
            else:
                fill_v = 0
            masks[k] = torch.from_numpy(np.full(self.args.grid_size[k], fill_v)).long().flatten()[None].to(self.device)
            x[k] = x[k].to(self.device)
        preds, _masks = self.model(x, mask_inputs=False, task_masks=masks)
        return preds

    @property
    def device(self):
        return next(self.parameters()).device


add_miragecls, miragecls_factory = get_factory_adder()


@add_miragecls("global")
class MIRAGEClsGlobal(MIRAGEWrapper):
    """Encoder + LayerNorm + mean over patch tokens + Linear head (mirage_wrapper.py:191-228)."""

    def __init__(self, num_classes=0, *args, **kwargs):
        super().__init__(*args, **kwargs)
        assert num_classes > 0
        assert len(self.args.in_domains) == 1
        self.num_classes = num_classes
        self.model.output_adapters = None
        self.embed_dim = self.model.encoder[0].norm1.normalized_shape[0]
        self.norm = LayerNorm(self.embed_dim, eps=1e-06, elementwise_affine=True)
        self.build_head()
        self.to(self.device)

    def build_head(self, factor=1):
        self.head = Linear(self.embed_dim * factor, self.num_classes)

    def forward(self, x):
        x_d = {self.args.in_domains[0]: x}
        out, _masks = self.model(x_d, mask_inputs=False)
        out = self.norm(out)
        out = self.pool(out)
        return self.head(out)

    def pool(self, x):
        return x[:, :-self.args.num_global_tokens, :].mean(dim=1)

    def get_output_adapters(self):
        return None


@add_miragecls("cls")
class MIRAGEClsCLS(MIRAGEClsGlobal):
    def pool(self, x):
        return x[:, -self.args.num_global_tokens:, :].mean(dim=1)


@add_miragecls("token_mix")
class MIRAGEClsTokenMix(MIRAGEClsGlobal):
    def build_head(self, factor=2):
        super().build_head(factor)

    def pool(self, x):
        patch = x[:, :-self.args.num_global_tokens, :].mean(dim=1)
        global_ = x[:, -self.args.num_global_tokens:, :].mean(dim=1)
        return torch.cat([patch, global_], dim=1)
