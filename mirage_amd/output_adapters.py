"""SpatialOutputAdapter with the reference's constructor, init(dim_tokens_enc) protocol, forward signature and
state_dict layout (mirage/output_adapters.py:22-296).  Segmentation heads of the reference's fine-tuning scripts
(output_adapters.py:325-762) are outside the pretraining path and not provided.
"""
from functools import partial
from typing import Callable, Dict, Optional, Tuple, Union

import torch
import torch.nn as nn

from . import engine as E
from .utils import Block, CrossAttention, LayerNorm, Linear, Mlp, build_2d_sincos_posemb, pair, trunc_normal_


class SpatialOutputAdapter(nn.Module):
    def __init__(self, num_channels: int, stride_level: int, patch_size_full: Union[int, Tuple[int, int]],
                 dim_tokens_enc: Optional[int] = None, dim_tokens: int = 256, depth: int = 0, learnable_pos_emb: int = False,
                 image_size: Union[int, Tuple[int, int]] = 224, mlp_ratio: int = 4, num_heads: int = 8, qkv_bias: bool = True,
                 drop_rate: float = 0.0, attn_drop_rate: float = 0.0, drop_path_rate: float = 0.0,
                 norm_layer: Callable = partial(LayerNorm, eps=1e-6), use_task_queries: bool = True, task: Optional[str] = None,
                 context_tasks: Optional[list] = None, use_xattn: bool = True):
        super().__init__()
        self.num_channels = num_channels
        self.stride_level = stride_level
        self.patch_size_full = pair(patch_size_full)
        self.dim_tokens_enc = dim_tokens_enc
        self.dim_tokens = dim_tokens
        self.learnable_pos_emb = learnable_pos_emb
        self.image_size = pair(image_size)
        self.use_task_queries = use_task_queries
        self.task = task
        self.use_xattn = use_xattn
        assert self.patch_size_full is not None
        assert self.image_size is not None
        if learnable_pos_emb:
            # the reference allocates this table channels-LAST ([1, h, w, D], mirage/output_adapters.py:107) and then resamples it with
            # F.interpolate as if it were [1, D, h, w] (:176, :221): its own forward fails on the shape mismatch, so there is no behaviour
            # to reproduce
            raise NotImplementedError("mirage_amd.SpatialOutputAdapter: learnable_pos_emb=True cannot run in the reference either "
                                      "(mirage/output_adapters.py:107 vs :176)")
        if not use_xattn or not qkv_bias:
            raise NotImplementedError("mirage_amd.SpatialOutputAdapter: use_xattn=False / qkv_bias=False are "
                                      "not on the pretraining path (run_pretraining.py:126-148 defaults)")
        if drop_path_rate and drop_path_rate > 0.0 and depth > 0:
            # the engine's decoder pass (engine.decoder_fwd) runs its blocks without stochastic depth: refuse instead of silently dropping it
            raise NotImplementedError("mirage_amd.SpatialOutputAdapter: drop_path_rate > 0 is not implemented for the decoder blocks "
                                      "(run_pretraining.py builds the output adapters without it)")
        self.P_H = max(1, self.patch_size_full[0] // stride_level)
        self.P_W = max(1, self.patch_size_full[1] // stride_level)
        if self.P_H != self.P_W:
            raise NotImplementedError("mirage_amd: square patches only")

        self.task_embeddings = None
        if context_tasks is not None:
            # a plain dict makes nn.ParameterDict sort its keys, exactly as in the reference
            self.task_embeddings = nn.ParameterDict({t: nn.Parameter(torch.zeros(1, 1, self.dim_tokens)) for t in context_tasks})
            for embedding in self.task_embeddings.values():
                trunc_normal_(embedding, std=0.02)
        self.mask_token = nn.Parameter(torch.zeros(1, 1, self.dim_tokens))
        h_posemb = self.image_size[0] // (self.stride_level * self.P_H)
        w_posemb = self.image_size[1] // (self.stride_level * self.P_W)
        self.pos_emb = nn.Parameter(build_2d_sincos_posemb(h=h_posemb, w=w_posemb, embed_dim=self.dim_tokens), requires_grad=False)

        self.decoder = CrossAttention(dim=self.dim_tokens, num_heads=num_heads, qkv_bias=qkv_bias, attn_drop=attn_drop_rate,
                                      proj_drop=drop_rate)
        self.context_norm = norm_layer(self.dim_tokens)
        self.query_norm = norm_layer(self.dim_tokens)
        self.out_norm = norm_layer(self.dim_tokens)
        self.mlp = Mlp(in_features=self.dim_tokens, hidden_features=int(self.dim_tokens * mlp_ratio))
        if depth > 0:
            dpr = [x.item() for x in torch.linspace(0, drop_path_rate, depth)]
            self.decoder_transformer = nn.Sequential(*[
                Block(dim=self.dim_tokens, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, drop=drop_rate,
                      attn_drop=attn_drop_rate, drop_path=dpr[i], norm_layer=norm_layer) for i in range(depth)])
        else:
            self.decoder_transformer = nn.Identity()
        self.dim_patch = self.num_channels * self.P_H * self.P_W
        self.out_proj = Linear(self.dim_tokens, self.dim_patch)
        if self.dim_tokens_enc is not None:
            self.init(dim_tokens_enc=self.dim_tokens_enc)

    def init(self, dim_tokens_enc: int = 768):
        self.dim_tokens_enc = dim_tokens_enc
        self.proj_context = Linear(self.dim_tokens_enc, self.dim_tokens)

    @torch.jit.ignore
    def no_weight_decay(self):
        return {"pos_emb", "mask_token", "task_embeddings"}

    def forward(self, encoder_tokens: torch.Tensor, input_info: Dict, ids_keep: torch.Tensor, ids_restore: torch.Tensor):
        """Same contract as mirage/output_adapters.py:248-296: encoder tokens [B, K+ng, D] (global tokens last) ->
        reconstruction [B, C, H, W]."""
        assert self.dim_tokens_enc is not None, "Need to call init(dim_tokens_enc) function first"
        names = list(input_info["tasks"].keys())
        dom_start = [input_info["tasks"][n]["start_idx"] for n in names] + [input_info["num_task_tokens"]]
        ng = input_info.get("num_global_tokens", 0)
        if ng < 1:
            raise NotImplementedError("mirage_amd: decoding without global tokens is not implemented")
        (img,) = E.DecodersFn.apply(encoder_tokens, None, [self], names, dom_start, ng, ids_keep.contiguous(), ids_restore.contiguous(),
                                    torch.is_grad_enabled(), False, None, *[p for p in self.parameters()])
        return img
