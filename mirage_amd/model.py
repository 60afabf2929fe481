"""MIRAGEModel with the reference's constructor, factory names, forward() signature and state_dict layout
(mirage/model.py: MIRAGEModel :22-431, miragepre_base :435, miragepre_large :457), executing on the HIP kernels.

Differences a caller can observe (all documented in DESIGN.md):
  * inputs / parameters must be on the GPU; predictions are bf16 (the reference returns fp16 under its autocast);
  * masks are drawn with device-side RNG (the reference draws the Dirichlet shares on the CPU), so the integer
    masks differ from the reference's for a given seed -- `generate_random_masks_from_noise` is the bit-exact,
    noise-in / indices-out form used by the parity tests;
  * only the kept tokens are embedded (the reference embeds all tokens and gathers).
"""
import itertools
import math
from collections import OrderedDict
from functools import partial
from typing import Dict, List, Optional, Sequence, Union

import torch
from torch import Tensor, nn

from . import engine as E
from . import kernels as K
from .arena import build_arena
from .factory import get_factory_adder
from .utils import Block, LayerNorm, trunc_normal_

add_model, model_factory = get_factory_adder()


class MIRAGEModel(nn.Module):
    """MultiMAE-style masked autoencoder: input adapters -> random token masking -> ViT encoder over kept tokens +
    global tokens -> per-task cross-attention output adapters.  Args as in mirage/model.py:29-58."""

    def __init__(self, args, input_adapters: Dict[str, nn.Module], output_adapters: Optional[Dict[str, nn.Module]],
                 num_global_tokens: int = 1, dim_tokens: int = 768, depth: int = 12, num_heads: int = 12, mlp_ratio: float = 4.0,
                 qkv_bias: bool = True, drop_rate: float = 0.0, attn_drop_rate: float = 0.0, drop_path_rate: float = 0.0,
                 norm_layer=partial(LayerNorm, eps=1e-6)):
        super().__init__()
        self.args = args
        for adapter in input_adapters.values():
            adapter.init(dim_tokens=dim_tokens)
        self.input_adapters = nn.ModuleDict(input_adapters)
        if output_adapters is not None:
            for adapter in output_adapters.values():
                adapter.init(dim_tokens_enc=dim_tokens)
            self.output_adapters = nn.ModuleDict(output_adapters)
        else:
            self.output_adapters = None
        self.num_global_tokens = num_global_tokens
        self.global_tokens = nn.Parameter(torch.zeros(1, num_global_tokens, dim_tokens))
        trunc_normal_(self.global_tokens, std=0.02)
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, depth)]
        self.encoder = nn.Sequential(*[
            Block(dim=dim_tokens, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, drop=drop_rate,
                  attn_drop=attn_drop_rate, drop_path=dpr[i], norm_layer=norm_layer) for i in range(depth)])
        for i, blk in enumerate(self.encoder):
            blk.drop_stream = 2 * i          # Philox mask streams of block i: 2i (after attn.proj), 2i + 1 (after mlp.fc2)
            blk.attn.attn_drop_stream = E.ATTN_DROP_STREAM0 + i       # dropout on the attention probabilities (attn_drop_rate)
        # modules with dropout on their attention probabilities: every forward pass hands them ONE Philox snapshot (engine.attn_dropout_args)
        self._mrg_adrop = [b.attn for b in self.encoder if b.attn.attn_drop > 0.0]
        for a, ad in enumerate((self.output_adapters or {}).values()):
            dec = getattr(ad, "decoder", None)
            if dec is not None:
                dec.attn_drop_stream = 0x20000 + 256 * a
                if dec.attn_drop > 0.0:
                    self._mrg_adrop.append(dec)
            for j, blk in enumerate(ad.decoder_transformer if isinstance(ad.decoder_transformer, nn.Sequential) else []):
                blk.attn.attn_drop_stream = 0x20000 + 256 * a + 1 + j
                if blk.attn.attn_drop > 0.0:
                    self._mrg_adrop.append(blk.attn)
        # initialisation: same calls in the same order as mirage/model.py:95-121 (same RNG stream -> same weights)
        self.apply(self._init_weights)
        for name, m in self.named_modules():
            if isinstance(m, nn.Linear):
                if "qkv" in name:
                    val = math.sqrt(6.0 / float(m.weight.shape[0] // 3 + m.weight.shape[1]))
                    nn.init.uniform_(m.weight, -val, val)
                elif "kv" in name:
                    val = math.sqrt(6.0 / float(m.weight.shape[0] // 2 + m.weight.shape[1]))
                    nn.init.uniform_(m.weight, -val, val)
            elif isinstance(m, nn.Conv2d):
                if ".proj" in name:
                    w = m.weight.data
                    nn.init.xavier_uniform_(w.view([w.shape[0], -1]))
        self.input_info = None
        self.token_dist = None
        self._mrg_arena = None
        self._mrg_cache = {}

    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            nn.init.xavier_uniform_(m.weight)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def get_num_layers(self):
        return len(self.encoder)

    @torch.jit.ignore
    def no_weight_decay(self):
        no_wd_set = {"global_tokens"}
        for task, adapter in self.input_adapters.items():
            if hasattr(adapter, "no_weight_decay"):
                no_wd_set |= {f"input_adapters.{task}.{name}" for name in adapter.no_weight_decay()}
        if self.output_adapters is not None:
            for task, adapter in self.output_adapters.items():
                if hasattr(adapter, "no_weight_decay"):
                    no_wd_set |= {f"output_adapters.{task}.{name}" for name in adapter.no_weight_decay()}
        return no_wd_set

    # ------------------------------------------------------------------------------------ arena / caches
    def _ensure_arena(self, device: torch.device):
        a = self._mrg_arena
        if a is None or a.device != device or not a.owns(self):
            if device.type != "cuda":
                raise RuntimeError("mirage_amd.MIRAGEModel runs on the GPU only (HIP kernels); move the model and inputs to cuda")
            a = build_arena(self, device)
            self._mrg_cache = {}
        a.refresh_shadow()
        return a

    def _apply(self, fn, *a, **kw):
        out = super()._apply(fn, *a, **kw)
        self._mrg_arena = None
        return out

    def _dom_start(self, doms: Sequence[str]) -> List[int]:
        key = ("dom_start", tuple(doms))
        v = self._mrg_cache.get(key)
        if v is None:
            v, i = [0], 0
            for d in doms:
                i += self.input_adapters[d].pos_emb.shape[-2] * self.input_adapters[d].pos_emb.shape[-1]
                v.append(i)
            self._mrg_cache[key] = v
        return v

    def _global_rows(self, B: int, N: int, row: int, device) -> Tensor:
        key = ("grows", B, N, row, str(device))
        v = self._mrg_cache.get(key)
        if v is None:
            v = (torch.arange(B, device=device, dtype=torch.int32) * N + row).contiguous()
            self._mrg_cache[key] = v
        return v

    # ------------------------------------------------------------------------------------ masking
    def _const(self, key, build):
        """Device constants are created once (no host->device copies inside a captured step)."""
        v = self._mrg_cache.get(key)
        if v is None:
            v = build()
            self._mrg_cache[key] = v
        return v

    def sample_alphas(self, B: int, n_tasks: int, alphas: Union[float, List[float], Tensor] = 1.0, eps: float = 1e-5, device=None):
        """Uniformly choose a non-empty subset of tasks per sample, alphas = choice * alpha + eps (mirage/model.py:145-166)."""
        al = tuple(float(a) for a in (alphas if isinstance(alphas, (list, tuple)) else [alphas] * n_tasks))
        valid = self._const(("valid_choices", n_tasks, str(device)), lambda: torch.tensor(
            [list(i) for i in itertools.product([0, 1], repeat=n_tasks)][1:], dtype=torch.float32, device=device))
        alt = self._const(("alphas", al, str(device)), lambda: torch.tensor(al, dtype=torch.float32, device=device))
        choice = torch.randint(0, len(valid), (B,), device=device)
        return torch.index_select(valid, 0, choice) * alt + eps

    def generate_random_masks_from_noise(self, p: Tensor, u: Tensor, g: Tensor, num_encoded_tokens: int, doms: Sequence[str]):
        """Pure function of the three random inputs: Dirichlet shares p [B, n], per-task noise
        u [B, T] (tasks concatenated), tie-break noise g [B, T] -> (task_masks, ids_keep, ids_restore), all int64."""
        dom_start = self._dom_start(doms)
        tm, ids_keep, ids_restore = K.maskgen(p.contiguous().float(), u.contiguous().float(), g.contiguous().float(),
                                              num_encoded_tokens, dom_start)
        task_masks = {d: tm[:, dom_start[i]:dom_start[i + 1]] for i, d in enumerate(doms)}
        return task_masks, ids_keep, ids_restore

    def generate_random_masks(self, input_tokens: Dict[str, Tensor], num_encoded_tokens: int,
                              alphas: Union[float, List[float], Tensor] = 1.0, sample_tasks_uniformly: bool = False):
        """Dirichlet split of num_encoded_tokens over the tasks + random subsets (mirage/model.py:168-239).
        `input_tokens` only supplies batch size, token counts and device (any tensors [B, T_d, ...])."""
        doms = list(input_tokens.keys())
        first = next(iter(input_tokens.values()))
        B, device = first.shape[0], first.device
        n = len(doms)
        T = sum(t.shape[1] for t in input_tokens.values())
        alphas = [alphas] * n if isinstance(alphas, float) else alphas
        if sample_tasks_uniformly:
            conc = self.sample_alphas(B, n, alphas=alphas, device=device)
        else:
            al = tuple(float(a) for a in alphas)
            conc = self._const(("conc", al, B, str(device)),
                               lambda: torch.tensor(al, dtype=torch.float32, device=device).expand(B, n).contiguous())
        # the Dirichlet shares (model.py:205-207) and the two uniform draws (model.py:216,225) are drawn inside the mask launch
        dom_start = self._dom_start(doms)
        tm, ids_keep, ids_restore = K.maskgen_rng(conc, self.mask_rng(device), num_encoded_tokens, T, dom_start)
        return {d: tm[:, dom_start[i]:dom_start[i + 1]] for i, d in enumerate(doms)}, ids_keep, ids_restore

    def mask_rng(self, device) -> Tensor:
        """Device state {seed, step, 0} (int64[3]) of the mask generator's Philox streams; the mask launch advances `step` itself, so a
        captured step draws new masks at every replay.  Seeded from `torch.initial_seed()` on first use and again whenever that seed
        has changed since (i.e. after `torch.manual_seed(other)`; not inside a stream capture); `seed_masks` pins it explicitly."""
        device = torch.device(device)
        states = self.__dict__.setdefault("_mask_rng_states", {})
        src = torch.initial_seed()
        st = states.get(device)
        if st is None:
            st = states[device] = [torch.tensor([src & 0x7FFFFFFFFFFFFFFF, 0, 0], dtype=torch.int64, device=device), src]
        elif st[1] != src and not (device.type == "cuda" and torch.cuda.is_current_stream_capturing()):
            st[0].copy_(torch.tensor([src & 0x7FFFFFFFFFFFFFFF, 0, 0], dtype=torch.int64))
            st[1] = src
        return st[0]

    def seed_masks(self, seed: int, step: int = 0, device=None):
        """Pin the mask stream (tests, resume): the next forward pass draws the masks of (seed, step)."""
        device = device if device is not None else next(self.parameters()).device
        st = self.mask_rng(device)
        st.copy_(torch.tensor([int(seed) & 0x7FFFFFFFFFFFFFFF, int(step), 0], dtype=torch.int64))

    @staticmethod
    def make_mask(N_H, N_W, xy_idxs, full_tasks=[], indicate_visible=True, flatten=True, device="cuda"):
        """Masks from lists of visible (x, y) patch coordinates (mirage/model.py:241-277)."""
        xy_idxs = {k: torch.LongTensor(v) for k, v in xy_idxs.items()}
        task_masks = {k: torch.ones(N_H, N_W).to(device) for k in xy_idxs.keys()}
        for k in xy_idxs.keys():
            if len(xy_idxs[k]) > 0:
                task_masks[k][xy_idxs[k][:, 1], xy_idxs[k][:, 0]] = 0
        for task in full_tasks:
            task_masks[task][:] = 0
        if not indicate_visible:
            task_masks = {k: 1 - v for k, v in task_masks.items()}
        if flatten:
            task_masks = {k: v.flatten().unsqueeze(0) for k, v in task_masks.items()}
        return task_masks

    def generate_input_info(self, input_task_tokens, image_size):
        """Token bookkeeping (mirage/model.py:279-303); input_task_tokens: domain -> number of tokens (or a tensor)."""
        input_info = OrderedDict()
        i = 0
        input_info["tasks"] = {}
        for domain, t in input_task_tokens.items():
            num_tokens = t if isinstance(t, int) else t.shape[1]
            d = {"num_tokens": num_tokens, "has_posemb": True, "start_idx": i, "end_idx": i + num_tokens}
            if isinstance(image_size, dict):
                d["image_size"] = image_size[domain]
            if getattr(self.args, "grid_sizes", None) is not None:
                d["grid_size"] = self.args.grid_sizes[domain]
            i += num_tokens
            input_info["tasks"][domain] = d
        if isinstance(image_size, int):
            input_info["image_size"] = image_size
        input_info["num_task_tokens"] = i
        input_info["num_global_tokens"] = self.num_global_tokens
        return input_info

    # ------------------------------------------------------------------------------------ forward
    def forward(self, x: Union[Dict[str, Tensor], Tensor], mask_inputs: bool = True, task_masks: Optional[Dict[str, Tensor]] = None,
                num_encoded_tokens: int = 128, alphas: Union[float, List[float]] = 1.0, sample_tasks_uniformly: bool = False,
                return_all_layers: bool = False, reshape: bool = False, mask_noise=None, drop_path_noise=None):
        """Same contract as mirage/model.py:305-431.  Extra keywords: `mask_noise=(p, u, g)` feeds explicit noise to the
        mask generator (bit-exact parity tests); `drop_path_noise` [blocks with drop_path > 0, 2, B] feeds the uniform
        draws of stochastic depth (default: torch.rand on the device)."""
        try:
            return self._forward(x, mask_inputs, task_masks, num_encoded_tokens, alphas, sample_tasks_uniformly, return_all_layers, reshape,
                                 mask_noise, drop_path_noise)
        finally:
            # the Philox snapshot handed to the attention modules belongs to THIS pass (their saved tuples keep it for the backward pass):
            # a later stand-alone call of such a module must draw its own (engine.attn_dropout_args), not replay this pass's masks
            for att in self._mrg_adrop:
                att._mrg_adrop_rng = None

    def _forward(self, x, mask_inputs, task_masks, num_encoded_tokens, alphas, sample_tasks_uniformly, return_all_layers, reshape, mask_noise,
                 drop_path_noise):
        x = {"bscan": x} if isinstance(x, Tensor) else x
        # the reference tokenises x.items() in x's OWN order and only the modalities it finds there (mirage/model.py:352-356): a subset of the
        # input adapters (features of one modality from a multi-modal checkpoint) and any key order are valid calls
        x = {d: t for d, t in x.items() if d in self.input_adapters}
        doms = list(x.keys())
        if not doms:
            raise ValueError(f"MIRAGEModel.forward: x holds none of the input adapters' modalities {list(self.input_adapters.keys())}")
        if self.output_adapters is not None:
            for task in self.output_adapters.keys():
                if task not in x:
                    raise KeyError(task)             # input_info['tasks'][self.task] (mirage/output_adapters.py:266): an output task must be an input
        first = x[doms[0]]
        B, device = first.shape[0], first.device
        self._ensure_arena(device)
        dom_start = self._dom_start(doms)
        sizes = {d: dom_start[i + 1] - dom_start[i] for i, d in enumerate(doms)}
        for d in doms:
            ad = self.input_adapters[d]
            H, W = x[d].shape[-2:]
            assert (H % ad.P_H == 0) and (W % ad.P_W == 0), f"Image sizes {H}x{W} must be divisible by patch sizes {ad.P_H}x{ad.P_W}"
            if (H // ad.P_H) * (W // ad.P_W) != sizes[d]:
                raise NotImplementedError("mirage_amd: input size differs from the adapter's positional-embedding grid")
        if self.input_info is None:
            self.input_info = self.generate_input_info(sizes, image_size=getattr(self.args, "input_size", None))
        input_info = self.input_info
        T = dom_start[-1]
        if mask_inputs:
            num_encoded_tokens = num_encoded_tokens if num_encoded_tokens is not None else self.num_encoded_tokens
        else:
            num_encoded_tokens = T

        if mask_noise is not None:
            task_masks, ids_keep, ids_restore = self.generate_random_masks_from_noise(*mask_noise, num_encoded_tokens, doms)
        elif task_masks is None:
            shapes = {d: first.new_empty((B, sizes[d], 0)) for d in doms}
            task_masks, ids_keep, ids_restore = self.generate_random_masks(shapes, num_encoded_tokens, alphas=alphas,
                                                                           sample_tasks_uniformly=sample_tasks_uniformly)
        else:
            # user masks (mirage/model.py:378-382): batch-wide visible count, i.e. only meaningful for B = 1
            mask_all = torch.cat([task_masks[t].to(device) for t in doms], dim=1)
            ids_shuffle = torch.argsort(mask_all, dim=1, stable=True)
            ids_restore = torch.argsort(ids_shuffle, dim=1)
            ids_keep = ids_shuffle[:, :int((mask_all == 0).sum())].contiguous()

        save = torch.is_grad_enabled()
        snap_a = None
        if self._mrg_adrop:
            # attn_drop_rate > 0 (mirage/model.py:88 -> utils.py:184, SDPA's dropout_p: also in eval mode): one Philox step per forward pass,
            # shared with the proj / fc2 dropout below
            snap_a = E.DropoutRNG.begin_forward(device)
            for att in self._mrg_adrop:
                att._mrg_adrop_rng = snap_a
        x0 = E.EmbedFn.apply(self, x, ids_keep.contiguous(), save, *self._embed_params())
        blocks = list(self.encoder)

        if return_all_layers:
            features = OrderedDict()
            gs = self.args.grid_sizes["bscan"]
            cur = x0
            for n, blk in enumerate(blocks):
                cur = blk(cur)
                f = cur[:, :-self.num_global_tokens]
                features[f"layer_{n}"] = f.reshape(B, gs[0], gs[1], -1).permute(0, 3, 1, 2)
            return features

        want16 = self.output_adapters is not None
        dps = None
        if self.training and any(getattr(b, "drop_path_prob", 0.0) > 0.0 for b in blocks):
            # one uniform draw per (block with drop_path > 0, branch, sample), in the reference's call order
            # (mirage/utils.py:259-262: attention branch, then MLP branch, block by block)
            act = [i for i, b in enumerate(blocks) if b.drop_path_prob > 0.0]
            u = drop_path_noise if drop_path_noise is not None else torch.rand((len(act), 2, B), dtype=torch.float32, device=device)
            dps = [None] * len(blocks)
            for j, i in enumerate(act):
                dps[i] = blocks[i].drop_path_rows(B, x0.shape[1], device, u[j])
        if self.training and any(getattr(b, "drop_prob", 0.0) > 0.0 for b in blocks):
            # nn.Dropout after proj / fc2 (drop_rate, mirage/model.py:54,87): one Philox step per forward pass, one mask stream per
            # (block, branch); the snapshot keeps forward and backward of THIS pass on the same masks
            snap = snap_a if snap_a is not None else E.DropoutRNG.begin_forward(device)
            dps = dps if dps is not None else [None] * len(blocks)
            for i, b in enumerate(blocks):
                drop = b.dropout_args(snap)
                if drop is not None:
                    dps[i] = (dps[i][0], dps[i][1], drop) if dps[i] is not None else (None, None, drop)
        split = getattr(self, "_mrg_split_at", None)
        cuts = [split] if isinstance(split, int) else list(split or [])
        cuts = [c for c in cuts if 0 < c < len(blocks)]
        if cuts and save:
            # data-parallel overlap (trainer.TrainStep): the encoder is several autograd nodes so that the backward pass can be run
            # down to the activation at a cut, the gradients of everything above it all-reduced, and the rest run meanwhile
            bounds = [0] + sorted(cuts) + [len(blocks)]
            h, hs = x0, []
            enc32 = enc16 = None
            for j in range(len(bounds) - 1):
                a, b = bounds[j], bounds[j + 1]
                last = j == len(bounds) - 2
                h32, h16 = E.EncoderFn.apply(h, blocks[a:b], save, want16 if last else False, True, dps[a:b] if dps else None,
                                             *self._encoder_params())
                if last:
                    enc32, enc16 = h32, h16
                else:
                    hs.append(h32)
                    h = h32
            self._mrg_h_cuts = hs
            self._mrg_h_mid = hs[-1]
        else:
            enc32, enc16 = E.EncoderFn.apply(x0, blocks, save, want16, True, dps, *self._encoder_params())
        if self.output_adapters is None:
            if reshape:
                gs = self.args.grid_sizes["bscan"]
                enc32 = enc32[:, :-self.num_global_tokens].reshape(B, gs[0], gs[1], -1).permute(0, 3, 1, 2)
            return enc32, task_masks

        ads = list(self.output_adapters.values())
        loss_spec = None
        if E.LaneLosses.spec is not None and save:
            # trainer.TrainStep: the masked losses run on the adapters' lanes; what comes back under each task's name is its LOSS
            loss_spec = []
            for name in self.output_adapters.keys():
                kind, patch, target, use_mask, *rest = E.LaneLosses.spec[name]
                m = task_masks.get(name) if use_mask else None
                loss_spec.append((kind, patch, target.contiguous() if kind != E.L.MRG_LOSS_MSE else target.contiguous().float(),
                                  E.loss_mask(m), float(rest[0]) if rest else 0.0,
                                  bool(rest[1]) if len(rest) > 1 else False,
                                  True))       # the masks are this call's own (task_masks and ids_keep come from one mask-generator launch)
        imgs = E.DecodersFn.apply(enc32, enc16, ads, doms, dom_start, self.num_global_tokens, ids_keep.contiguous(),
                                  ids_restore.contiguous(), save, True, loss_spec, *self._decoder_params())
        preds = {name: img for name, img in zip(self.output_adapters.keys(), imgs)}
        return preds, task_masks

    def _embed_params(self):
        v = self._mrg_cache.get("embed_params")
        if v is None:
            v = [self.global_tokens] + [p for a in self.input_adapters.values() for p in a.parameters() if p.requires_grad]
            self._mrg_cache["embed_params"] = v
        return v

    def _encoder_params(self):
        v = self._mrg_cache.get("encoder_params")
        if v is None:
            v = list(self.encoder.parameters())
            self._mrg_cache["encoder_params"] = v
        return v

    def _decoder_params(self):
        v = self._mrg_cache.get("decoder_params")
        if v is None:
            v = [p for p in self.output_adapters.parameters() if p.requires_grad]
            self._mrg_cache["decoder_params"] = v
        return v


@add_model
def miragepre_base(input_adapters: Dict[str, nn.Module], output_adapters: Optional[Dict[str, nn.Module]], args, **kwargs):
    return MIRAGEModel(args, input_adapters=input_adapters, output_adapters=output_adapters, dim_tokens=768, depth=12, num_heads=12,
                       mlp_ratio=4, qkv_bias=True, norm_layer=partial(LayerNorm, eps=1e-6), **kwargs)


@add_model
def miragepre_large(input_adapters: Dict[str, nn.Module], output_adapters: Optional[Dict[str, nn.Module]], args, **kwargs):
    return MIRAGEModel(args, input_adapters=input_adapters, output_adapters=output_adapters, dim_tokens=1024, depth=24, num_heads=16,
                       mlp_ratio=4, qkv_bias=True, norm_layer=partial(LayerNorm, eps=1e-6), **kwargs)
