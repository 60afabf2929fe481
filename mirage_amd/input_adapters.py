"""Input adapters with the reference's constructor / init(dim_tokens) protocol and state_dict layout
(mirage/input_adapters.py: PatchedInputAdapter :12, SemSegInputAdapter :113).

Inside MIRAGEModel the adapters are not called token-by-token: the model gathers the kept patches first and runs
one im2col + MFMA GEMM per modality (engine.embed_fwd).  `forward(x)` here embeds ALL tokens (the reference's
behaviour) through the same kernels, for callers that use an adapter on its own.
"""
from typing import Optional, Tuple, Union

import torch
import torch.nn as nn

from . import engine as E
from . import kernels as K
from .arena import grad_buffer, shadow
from .utils import build_2d_sincos_posemb, pair, trunc_normal_


class _AllTokensFn(torch.autograd.Function):
    """Embeds every patch of one modality: ids = arange(T) per sample."""

    @staticmethod
    def forward(ctx, x, adapter, *params):
        B = x.shape[0]
        g_h = x.shape[-2] // adapter.P_H
        g_w = x.shape[-1] // adapter.P_W
        T = g_h * g_w
        dev = x.device
        D = adapter.dim_tokens
        ids = torch.arange(T, device=dev, dtype=torch.int64)[None].expand(B, -1).contiguous()
        rows, toks, srcb, counts = K.embed_plan(ids, T, [0, T])
        cap = B * T
        if adapter.kind == "semseg":
            A = K.im2col_layermap(x.contiguous(), adapter.class_emb.weight, srcb[0], toks[0], counts[0:1], cap, adapter.P_H)
        else:
            A = K.im2col_image(x.contiguous().float(), srcb[0], toks[0], counts[0:1], cap, adapter.P_H)
        out = torch.empty((B, T, D), dtype=torch.float32, device=dev)
        Kp = A.shape[1]
        K.gemm(A, shadow(adapter.proj.weight), cap, D, Kp, lda=Kp, ldb=Kp, out=out, ldc=D, bias=adapter.proj.bias, row_map=rows[0],
               add_tab=E._pos_tokens(adapter), add_row=toks[0], m_dev=counts[0:1])
        ctx.adapter, ctx.keep = adapter, (A, rows, toks, srcb, counts, x)
        return out

    @staticmethod
    def backward(ctx, dy):
        ad = ctx.adapter
        A, rows, toks, srcb, counts, x = ctx.keep
        B, T, D = dy.shape
        cap, Kp = A.shape
        dy32 = dy.contiguous().float()
        dy16 = E.to_bf16(dy32)
        KS = E.KS
        K.gemm(dy16, A, D, Kp, cap, a_mode=KS, b_mode=KS, lda=D, ldb=Kp, out=grad_buffer(ad.proj.weight).view(D, Kp), ldc=Kp,
               splitk=K.pick_splitk(D, Kp, cap))
        K.colsum(dy32.view(cap, D), grad_buffer(ad.proj.bias))
        if ad.kind == "semseg":
            dA = torch.empty((cap, Kp), dtype=torch.bfloat16, device=dy.device)
            K.gemm(dy16, shadow(ad.proj.weight), cap, Kp, D, b_mode=KS, lda=D, ldb=Kp, out=dA, ldc=Kp)
            ncls, Ee = ad.class_emb.weight.shape
            K.class_emb_grad(dA, x.contiguous(), srcb[0], toks[0], counts[0:1], ncls, Ee, ad.P_H, grad_buffer(ad.class_emb.weight))
        if ad.pos_emb.requires_grad:                 # learnable_pos_emb (mirage/input_adapters.py:72,185): every sample holds every token
            K.posemb_grad(dy32.view(cap, D), grad_buffer(ad.pos_emb), B, T)
        return (None,) * len(ctx.needs_input_grad)


class PatchedInputAdapter(nn.Module):
    """Patch projection Conv2d(k = s = P) + fixed sin-cos positional embedding (mirage/input_adapters.py:12-110)."""
    kind = "image"

    def __init__(self, num_channels: int, stride_level: int, patch_size_full: Union[int, Tuple[int, int]],
                 dim_tokens: Optional[int] = None, sincos_pos_emb: bool = True, learnable_pos_emb: bool = False,
                 image_size: Union[int, Tuple[int]] = 224):
        super().__init__()
        self.num_channels = num_channels
        self.stride_level = stride_level
        self.patch_size_full = pair(patch_size_full)
        self.dim_tokens = dim_tokens
        self.sincos_pos_emb = sincos_pos_emb
        self.learnable_pos_emb = learnable_pos_emb
        self.image_size = pair(image_size)
        self.num_patches = (self.image_size[0] // self.patch_size_full[0]) * (self.image_size[1] // self.patch_size_full[1])
        self.P_H = max(1, self.patch_size_full[0] // stride_level)
        self.P_W = max(1, self.patch_size_full[1] // stride_level)
        if self.P_H != self.P_W:
            raise NotImplementedError("mirage_amd: square patches only")
        if self.dim_tokens is not None:
            self.init(dim_tokens=dim_tokens)

    def init(self, dim_tokens: int = 768):
        self.dim_tokens = dim_tokens
        h_posemb = self.image_size[0] // (self.stride_level * self.P_H)
        w_posemb = self.image_size[1] // (self.stride_level * self.P_W)
        if self.sincos_pos_emb:
            self.pos_emb = nn.Parameter(build_2d_sincos_posemb(h=h_posemb, w=w_posemb, embed_dim=self.dim_tokens),
                                        requires_grad=self.learnable_pos_emb)
        else:
            self.pos_emb = nn.Parameter(torch.zeros(1, self.dim_tokens, h_posemb, w_posemb))
            trunc_normal_(self.pos_emb, std=0.02)
        self.proj = nn.Conv2d(in_channels=self.num_channels, out_channels=self.dim_tokens, kernel_size=(self.P_H, self.P_W),
                              stride=(self.P_H, self.P_W))

    @torch.jit.ignore
    def no_weight_decay(self):
        return {"pos_emb"}

    def forward(self, x):
        _B, _C, H, W = x.shape
        assert self.dim_tokens is not None, "Need to call init(dim_tokens) function first"
        assert (H % self.P_H == 0) and (W % self.P_W == 0), f"Image sizes {H}x{W} must be divisible by patch sizes {self.P_H}x{self.P_W}"
        if (H // self.P_H, W // self.P_W) != tuple(self.pos_emb.shape[-2:]):
            raise NotImplementedError("mirage_amd: positional-embedding resampling at forward time is not implemented; "
                                      "resample the checkpoint with mirage_amd.checkpoint.interpolate_pos_embed")
        return _AllTokensFn.apply(x, self, self.proj.weight, self.proj.bias, self.pos_emb)


class SemSegInputAdapter(nn.Module):
    """Class embedding + patch projection for label maps (mirage/input_adapters.py:113-238)."""
    kind = "semseg"

    def __init__(self, num_classes: int, stride_level: int, patch_size_full: Union[int, Tuple[int, int]],
                 dim_tokens: Optional[int] = None, sincos_pos_emb: int = True, learnable_pos_emb: int = False,
                 image_size: Union[int, Tuple[int]] = 224, dim_class_emb: int = 64, interpolate_class_emb: bool = False,
                 emb_padding_idx: int = None):
        super().__init__()
        self.num_classes = num_classes
        self.stride_level = stride_level
        self.patch_size_full = pair(patch_size_full)
        self.dim_tokens = dim_tokens
        self.sincos_pos_emb = sincos_pos_emb
        self.learnable_pos_emb = learnable_pos_emb
        self.image_size = pair(image_size)
        self.dim_class_emb = dim_class_emb
        self.interpolate_class_emb = interpolate_class_emb
        self.emb_padding_idx = emb_padding_idx
        if self.emb_padding_idx is not None:
            self.num_classes += 1
        if interpolate_class_emb or emb_padding_idx is not None:
            raise NotImplementedError("mirage_amd: interpolate_class_emb / emb_padding_idx are not on the pretraining path")
        self.P_H = max(1, self.patch_size_full[0] // stride_level)
        self.P_W = max(1, self.patch_size_full[1] // stride_level)
        if self.P_H != self.P_W:
            raise NotImplementedError("mirage_amd: square patches only")
        if self.dim_tokens is not None:
            self.init(dim_tokens=dim_tokens)

    def init(self, dim_tokens: int = 768):
        self.dim_tokens = dim_tokens
        h_posemb = self.image_size[0] // (self.stride_level * self.P_H)
        w_posemb = self.image_size[1] // (self.stride_level * self.P_W)
        if self.sincos_pos_emb:
            self.pos_emb = nn.Parameter(build_2d_sincos_posemb(h=h_posemb, w=w_posemb, embed_dim=self.dim_tokens),
                                        requires_grad=bool(self.learnable_pos_emb))
        else:
            self.pos_emb = nn.Parameter(torch.zeros(1, self.dim_tokens, h_posemb, w_posemb))
            trunc_normal_(self.pos_emb, std=0.02)
        self.class_emb = nn.Embedding(num_embeddings=self.num_classes, embedding_dim=self.dim_class_emb, padding_idx=self.emb_padding_idx)
        trunc_normal_(self.class_emb.weight, std=0.02)
        self.proj = nn.Conv2d(in_channels=self.dim_class_emb, out_channels=self.dim_tokens, kernel_size=(self.P_H, self.P_W),
                              stride=(self.P_H, self.P_W))

    @torch.jit.ignore
    def no_weight_decay(self):
        return {"pos_emb", "class_emb"}

    def forward(self, x):
        _B, H, W = x.shape
        assert self.dim_tokens is not None, "Need to call init(dim_tokens) function first"
        assert (H % self.P_H == 0) and (W % self.P_W == 0), f"Image sizes {H}x{W} must be divisible by patch sizes {self.P_H}x{self.P_W}"
        if (H // self.P_H, W // self.P_W) != tuple(self.pos_emb.shape[-2:]):
            raise NotImplementedError("mirage_amd: positional-embedding resampling at forward time is not implemented")
        return _AllTokensFn.apply(x, self, self.proj.weight, self.proj.bias, self.class_emb.weight, self.pos_emb)
