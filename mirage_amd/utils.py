"""Transformer building blocks with the reference's names, constructor arguments and state_dict layout
(mirage/utils.py: pair :12, build_2d_sincos_posemb :24, trunc_normal_ :85, DropPath :124, Mlp :137,
Attention :162, CrossAttention :191, Block :226).  Parameters are ordinary fp32 nn.Parameters; every forward runs
on the HIP kernels (no torch math), so inputs and parameters must be on the GPU.
"""
import math
import warnings
from typing import Callable, Optional

import torch
import torch.nn as nn

from . import _lib as L
from . import engine as E
from . import kernels as K
from .arena import grad_buffer, shadow

BF16, F32 = torch.bfloat16, torch.float32


def pair(t):
    if t is None:
        return None
    if isinstance(t, tuple):
        return t
    if isinstance(t, list):
        return tuple(t)
    return (t, t)


def build_2d_sincos_posemb(h, w, embed_dim=1024, temperature=10000.0):
    """Fixed 2-D sin-cos table [1, D, h, w] (mirage/utils.py:24-41), including the reference's axis convention: the
    token at flat index m = row*w + col is given "w-coordinate" m // h and "h-coordinate" m % h."""
    assert embed_dim % 4 == 0, "Embed dimension must be divisible by 4 for 2D sin-cos position embedding"
    pos_dim = embed_dim // 4
    omega = torch.arange(pos_dim, dtype=torch.float32) / pos_dim
    omega = 1.0 / (temperature ** omega)
    m = torch.arange(h * w, dtype=torch.float32)
    cw = torch.div(m, h, rounding_mode="floor")
    ch = m - cw * h
    out_w = cw[:, None] * omega[None, :]
    out_h = ch[:, None] * omega[None, :]
    tab = torch.cat([torch.sin(out_w), torch.cos(out_w), torch.sin(out_h), torch.cos(out_h)], dim=1)
    return tab.reshape(1, h, w, embed_dim).permute(0, 3, 1, 2).contiguous()


def trunc_normal_(tensor, mean=0.0, std=1.0, a=-2.0, b=2.0):
    """Truncated normal init by inverse-CDF sampling (same algorithm and RNG consumption as mirage/utils.py:44-100)."""
    def norm_cdf(x):
        return (1.0 + math.erf(x / math.sqrt(2.0))) / 2.0
    if (mean < a - 2 * std) or (mean > b + 2 * std):
        warnings.warn("mean is more than 2 std from [a, b] in trunc_normal_.", stacklevel=2)
    with torch.no_grad():
        lo, up = norm_cdf((a - mean) / std), norm_cdf((b - mean) / std)
        tensor.uniform_(2 * lo - 1, 2 * up - 1)
        tensor.erfinv_()
        tensor.mul_(std * math.sqrt(2.0))
        tensor.add_(mean)
        tensor.clamp_(min=a, max=b)
    return tensor


def drop_path_factor(u: torch.Tensor, drop_prob: float) -> torch.Tensor:
    """Per-sample factor of drop_path (mirage/utils.py:103-118): floor(keep_prob + u) / keep_prob for uniform noise u."""
    keep = 1.0 - drop_prob
    return torch.floor(keep + u.float()) / keep


class DropPath(nn.Module):
    """Stochastic depth (mirage/utils.py:124-134).  Every configuration on the pretraining path uses p = 0
    (run_pretraining.py:150), where the reference installs nn.Identity.  Inside a Block the per-sample factor is fused into
    the proj / fc2 GEMM epilogues (engine.block_fwd); called stand-alone it is the same elementwise rule as the reference."""

    def __init__(self, drop_prob=0.0):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        shape = (x.shape[0],) + (1,) * (x.ndim - 1)
        return x * drop_path_factor(torch.rand(shape, dtype=torch.float32, device=x.device), self.drop_prob).to(x.dtype)

    def extra_repr(self):
        return "p={}".format(self.drop_prob)


def _check_no_dropout(p, what):
    if p and p > 0.0:
        raise NotImplementedError(f"mirage_amd: {what} > 0 is not implemented on the HIP path (the pretraining configs use 0)")


def _check_prob(p, what):
    if p is None or not (0.0 <= float(p) < 1.0):
        raise ValueError(f"mirage_amd: {what} = {p!r} must be in [0, 1)")


# ------------------------------------------------------------------------- generic stand-alone ops (autograd)
class _LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, lin, gelu, weight, bias):
        x16 = x if x.dtype == BF16 else E.to_bf16(x.contiguous().float())
        save = torch.is_grad_enabled() or True
        y, _, pre = E.lin_fwd(x16.contiguous(), lin, out_dtype=F32 if x.dtype != BF16 else BF16, gelu=gelu, save_pre=gelu)
        ctx.lin, ctx.x16, ctx.pre, ctx.in_dtype = lin, x16, pre, x.dtype
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        lin = ctx.lin
        if ctx.pre is not None:
            # dy is w.r.t. gelu(pre): fold gelu' in with an elementwise pass through the GEMM-free path
            raise NotImplementedError
        dy16 = dy if dy.dtype == BF16 else E.to_bf16(dy.float())
        dx, _ = E.lin_bwd(dy16.view(-1, dy16.shape[-1]), ctx.x16.view(-1, ctx.x16.shape[-1]), lin,
                          dx_dtype=F32 if ctx.in_dtype != BF16 else BF16)
        E.join_side_stream()
        return dx.view(ctx.x16.shape), None, None, None, None


class Linear(nn.Linear):
    """nn.Linear whose forward/backward run on the bf16 MFMA GEMM (fp32 master weights, bf16 operands)."""

    def forward(self, x):
        return _LinearFn.apply(x, self, False, self.weight, self.bias)


class _LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, ln, weight, bias):
        x32 = x.contiguous().float()
        y, mean, rstd = K.ln_fwd(x32, ln.weight, ln.bias, ln.eps, out_dtype=F32)
        ctx.ln, ctx.keep = ln, (x32, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x32, mean, rstd = ctx.keep
        ln = ctx.ln
        dx, _ = K.ln_bwd(dy.contiguous().float(), x32, mean, rstd, ln.weight, None, grad_buffer(ln.weight), grad_buffer(ln.bias),
                         None, want_bf16=False)
        return dx, None, None, None


class LayerNorm(nn.LayerNorm):
    """nn.LayerNorm on the wave-per-row HIP kernel (fp32 statistics)."""

    def forward(self, x):
        return _LayerNormFn.apply(x, self, self.weight, self.bias)


class _MlpFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mlp, *params):
        x16 = E.to_bf16(x.contiguous().float())
        h_act, _, h_pre = E.lin_fwd(x16, mlp.fc1, gelu=True, save_pre=True)
        y, _, _ = E.lin_fwd(h_act, mlp.fc2, out_dtype=F32)
        ctx.mlp, ctx.keep = mlp, (x16, h_pre, h_act)
        return y

    @staticmethod
    def backward(ctx, dy):
        x16, h_pre, h_act = ctx.keep
        mlp = ctx.mlp
        dy16 = E.to_bf16(dy.contiguous().float())
        d_pre, _ = E.lin_bwd(dy16, h_act, mlp.fc2, gelu_pre=h_pre)
        dx, _ = E.lin_bwd(d_pre, x16, mlp.fc1, dx_dtype=F32)
        E.join_side_stream()
        return (dx, None) + (None,) * 4


class Mlp(nn.Module):
    """fc1 -> exact-erf GELU -> fc2 (mirage/utils.py:137-159); GELU is fused into the fc1 epilogue."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0, _in_block=False):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        if act_layer is not nn.GELU:
            raise NotImplementedError("mirage_amd.Mlp: only the exact-erf nn.GELU of the reference is fused")
        # dropout after fc2 is fused into the fc2 epilogue of a Block (engine.block_fwd); a stand-alone Mlp has no such epilogue
        if _in_block:
            _check_prob(drop, "Mlp drop")
        else:
            _check_no_dropout(drop, "Mlp drop (outside a Block)")
        self.fc1 = Linear(in_features, hidden_features)
        self.act = act_layer()
        self.fc2 = Linear(hidden_features, out_features)
        self.drop = nn.Dropout(drop)

    def forward(self, x):
        return _MlpFn.apply(x, self, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias)


class _AttentionFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, att, *params):
        B, N, C = x.shape
        H = att.num_heads
        hd = C // H
        x16 = E.to_bf16(x.contiguous().float())
        qkv, _, _ = E.lin_fwd(x16, att.qkv)
        st = (N * 3 * C, 3 * C)
        adrop = E.attn_dropout_args(att)
        o, lse = K.attn_fwd(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], B, H, N, N, hd, att.scale, st, st, st, drop=adrop)
        y, _, _ = E.lin_fwd(o, att.proj, out_dtype=F32)
        ctx.att, ctx.keep = att, (x16, qkv, o, lse, adrop)
        return y

    @staticmethod
    def backward(ctx, dy):
        x16, qkv, o, lse, adrop = ctx.keep
        att = ctx.att
        B, N, C = x16.shape
        H = att.num_heads
        hd = C // H
        dy16 = E.to_bf16(dy.contiguous().float())
        d_o, _ = E.lin_bwd(dy16, o, att.proj)
        d_qkv = torch.empty_like(qkv)
        st = (N * 3 * C, 3 * C)
        K.attn_bwd(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], o, lse, d_o, d_qkv[..., :C], d_qkv[..., C:2 * C], d_qkv[..., 2 * C:],
                   B, H, N, N, hd, att.scale, st, st, st, st, st, st, drop=adrop)
        dx, _ = E.lin_bwd(d_qkv, x16, att.qkv, dx_dtype=F32)
        E.join_side_stream()
        return (dx, None) + (None,) * 4


class Attention(nn.Module):
    """Fused-QKV self attention (mirage/utils.py:162-188)."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, attn_drop=0.0, proj_drop=0.0, _in_block=False):
        super().__init__()
        # dropout on the attention PROBABILITIES (F.scaled_dot_product_attention(dropout_p=attn_drop), mirage/utils.py:181-185; applied in
        # eval mode too, like the reference): the plain path of csrc/attention_drop.hip, see engine.attn_dropout_args
        _check_prob(attn_drop, "attn_drop")
        if _in_block:
            _check_prob(proj_drop, "proj_drop")
        else:
            _check_no_dropout(proj_drop, "proj_drop (outside a Block)")
        self.num_heads = num_heads
        head_dim = dim // num_heads
        if head_dim not in (32, 64):
            raise NotImplementedError(f"mirage_amd.Attention: head_dim {head_dim} not in (32, 64)")
        self.scale = head_dim ** -0.5
        self.qkv = Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = attn_drop
        self.proj = Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)

    def forward(self, x):
        return _AttentionFn.apply(x, self, self.qkv.weight, self.qkv.bias, self.proj.weight, self.proj.bias)


class _CrossAttentionFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, context, att, *params):
        B, N, C = x.shape
        M = context.shape[1]
        H = att.num_heads
        hd = C // H
        x16 = E.to_bf16(x.contiguous().float())
        c16 = E.to_bf16(context.contiguous().float())
        q, _, _ = E.lin_fwd(x16, att.q)
        kv, _, _ = E.lin_fwd(c16, att.kv)
        sq, skv = (N * C, C), (M * 2 * C, 2 * C)
        adrop = E.attn_dropout_args(att)
        o, lse = K.attn_fwd(q, kv[..., :C], kv[..., C:], B, H, N, M, hd, att.scale, sq, skv, skv, drop=adrop)
        y, _, _ = E.lin_fwd(o, att.proj, out_dtype=F32)
        ctx.att, ctx.keep = att, (x16, c16, q, kv, o, lse, adrop)
        return y

    @staticmethod
    def backward(ctx, dy):
        x16, c16, q, kv, o, lse, adrop = ctx.keep
        att = ctx.att
        B, N, C = x16.shape
        M = c16.shape[1]
        H = att.num_heads
        hd = C // H
        dy16 = E.to_bf16(dy.contiguous().float())
        d_o, _ = E.lin_bwd(dy16, o, att.proj)
        dq = torch.empty_like(q)
        dkv = torch.empty_like(kv)
        sq, skv = (N * C, C), (M * 2 * C, 2 * C)
        K.attn_bwd(q, kv[..., :C], kv[..., C:], o, lse, d_o, dq, dkv[..., :C], dkv[..., C:], B, H, N, M, hd, att.scale, sq, skv, skv,
                   sq, skv, skv, drop=adrop)
        dx, _ = E.lin_bwd(dq, x16, att.q, dx_dtype=F32)
        dc, _ = E.lin_bwd(dkv, c16, att.kv, dx_dtype=F32)
        E.join_side_stream()
        return (dx, dc, None) + (None,) * 6


class CrossAttention(nn.Module):
    """Query / context attention (mirage/utils.py:191-223)."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, attn_drop=0.0, proj_drop=0.0):
        super().__init__()
        _check_prob(attn_drop, "attn_drop")                # dropout on the probabilities: see Attention
        _check_no_dropout(proj_drop, "proj_drop")
        self.num_heads = num_heads
        head_dim = dim // num_heads
        if head_dim not in (32, 64):
            raise NotImplementedError(f"mirage_amd.CrossAttention: head_dim {head_dim} not in (32, 64)")
        self.scale = head_dim ** -0.5
        self.q = Linear(dim, dim, bias=qkv_bias)
        self.kv = Linear(dim, dim * 2, bias=qkv_bias)
        self.attn_drop = attn_drop
        self.proj = Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)

    def forward(self, x, context):
        return _CrossAttentionFn.apply(x, context, self, self.q.weight, self.q.bias, self.kv.weight, self.kv.bias, self.proj.weight,
                                       self.proj.bias)


class Block(nn.Module):
    """Pre-LN transformer block (mirage/utils.py:226-262)."""

    def __init__(self, dim: int, num_heads: int, mlp_ratio: float = 4.0, qkv_bias: bool = False, drop: float = 0.0,
                 attn_drop: float = 0.0, drop_path: float = 0.0, act_layer: Callable = nn.GELU, norm_layer: Callable = LayerNorm):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, attn_drop=attn_drop, proj_drop=drop, _in_block=True)
        self.drop_path = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop, _in_block=True)
        if not (qkv_bias and isinstance(self.norm1, nn.LayerNorm)):
            raise NotImplementedError("mirage_amd.Block: needs qkv_bias=True and a LayerNorm norm_layer (the reference's defaults)")
        self.drop_path_prob = float(drop_path)
        # nn.Dropout(drop) after attn.proj and after mlp.fc2 (mirage/utils.py:158,187): Philox masks fused into those GEMMs' epilogues.
        # drop_stream: even id that tells this block's two mask streams (id, id + 1) from every other block's within one step; the
        # owning model numbers its blocks (MIRAGEModel: 2 * index), a stand-alone Block keeps 0
        self.drop_prob = float(drop)
        self.drop_stream = 0

    def _params(self):
        return (self.norm1.weight, self.norm1.bias, self.attn.qkv.weight, self.attn.qkv.bias, self.attn.proj.weight, self.attn.proj.bias,
                self.norm2.weight, self.norm2.bias, self.mlp.fc1.weight, self.mlp.fc1.bias, self.mlp.fc2.weight, self.mlp.fc2.bias)

    def drop_path_rows(self, B: int, N: int, device, noise: Optional[torch.Tensor] = None):
        """Per-row factors (attention branch, MLP branch) for one forward, or None when drop_path is inactive.  `noise`:
        uniform [2, B] (tests feed the reference's draws); default = torch.rand on `device`, one draw per branch per sample in
        the reference's order (attention first, mirage/utils.py:260-261)."""
        if self.drop_path_prob == 0.0 or not self.training:
            return None
        u = noise if noise is not None else torch.rand((2, B), dtype=torch.float32, device=device)
        f = drop_path_factor(u.to(device), self.drop_path_prob)
        return (f[0].repeat_interleave(N).contiguous(), f[1].repeat_interleave(N).contiguous())

    def dropout_args(self, rng_snapshot: Optional[torch.Tensor]):
        """(rng snapshot, p, stream id) for engine.block_fwd, or None when dropout is inactive (p = 0 or eval mode)"""
        if self.drop_prob == 0.0 or not self.training or rng_snapshot is None:
            return None
        return (rng_snapshot, self.drop_prob, self.drop_stream)

    def forward(self, x, drop_path_noise: Optional[torch.Tensor] = None):
        dp = self.drop_path_rows(x.shape[0], x.shape[1], x.device, drop_path_noise)
        if self.drop_prob > 0.0 and self.training:
            drop = self.dropout_args(E.DropoutRNG.begin_forward(x.device))
            dp = (dp[0], dp[1], drop) if dp is not None else (None, None, drop)
        return E.BlockFn.apply(x, self, torch.is_grad_enabled(), dp, *self._params())
