"""Model / loss construction exactly as run_pretraining.py:38-57 (DOMAIN_CONF) and :373-419 (get_model) do it."""
import copy
import types
from functools import partial
from typing import Dict, Optional, Sequence

from .criterion import MaskedCrossEntropyLoss, MaskedMSELoss
from .input_adapters import PatchedInputAdapter, SemSegInputAdapter
from .model import model_factory
from .output_adapters import SpatialOutputAdapter

DEFAULT_CONF = {
    "channels": 1,
    "stride_level": 1,
    "input_adapter": partial(PatchedInputAdapter, num_channels=1),
    "output_adapter": partial(SpatialOutputAdapter, num_channels=1),
    "loss": MaskedMSELoss,
}

DOMAIN_CONF = {
    "bscan": copy.deepcopy(DEFAULT_CONF),
    "slo": copy.deepcopy(DEFAULT_CONF),
    "bscanlayermap": {
        "num_classes": 13,
        "stride_level": 1,
        "input_adapter": partial(SemSegInputAdapter, num_classes=13, dim_class_emb=64, interpolate_class_emb=False),
        "output_adapter": partial(SpatialOutputAdapter, num_channels=13),
        "loss": partial(MaskedCrossEntropyLoss, label_smoothing=0.0),
    },
}


def make_args(in_domains: Sequence[str], out_domains: Optional[Sequence[str]] = None, input_size=512, patch_size=32,
              layermap_size=128, layermap_patch=8, decoder_dim=256, decoder_depth=2, decoder_num_heads=8, num_global_tokens=1,
              drop_path=0.0, **extra):
    """Namespace carrying what get_args() derives (run_pretraining.py:317-339): per-domain patch / input / grid sizes."""
    a = types.SimpleNamespace(**extra)
    a.in_domains = list(in_domains)
    a.out_domains = list(out_domains if out_domains is not None else in_domains)
    a.all_domains = list(dict.fromkeys(a.in_domains + a.out_domains))
    a.patch_size, a.input_size, a.grid_sizes = {}, {}, {}
    for d in a.all_domains:
        if d == "bscanlayermap":
            a.patch_size[d], a.input_size[d] = (layermap_patch, layermap_patch), (layermap_size, layermap_size)
        else:
            a.patch_size[d], a.input_size[d] = (patch_size, patch_size), (input_size, input_size)
        a.grid_sizes[d] = [a.input_size[d][i] // a.patch_size[d][i] for i in range(2)]
    a.decoder_dim, a.decoder_depth, a.decoder_num_heads = decoder_dim, decoder_depth, decoder_num_heads
    a.decoder_use_task_queries, a.decoder_use_xattn = True, True
    a.num_global_tokens, a.drop_path = num_global_tokens, drop_path
    return a


def get_model(args, model_name: str = "miragepre_base", with_decoders: bool = True, decoder_attn_drop_rate: float = 0.0,
              learnable_pos_emb: bool = False, **model_kwargs):
    """Adapters + model (run_pretraining.py:373-419).  model_kwargs (dim_tokens / depth / num_heads) allow the
    reduced-size models used by the parity fixtures; decoder_attn_drop_rate is SpatialOutputAdapter's attn_drop_rate
    (mirage/output_adapters.py:64; run_pretraining.py leaves it 0); learnable_pos_emb is the INPUT adapters' option
    (mirage/input_adapters.py:34,140; run_pretraining.py leaves it False)."""
    input_adapters = {
        d: DOMAIN_CONF[d]["input_adapter"](stride_level=DOMAIN_CONF[d]["stride_level"], patch_size_full=tuple(args.patch_size[d]),
                                           image_size=args.input_size[d], learnable_pos_emb=learnable_pos_emb)
        for d in args.in_domains
    }
    output_adapters = None
    if with_decoders and args.out_domains:
        output_adapters = {
            d: DOMAIN_CONF[d]["output_adapter"](stride_level=DOMAIN_CONF[d]["stride_level"], patch_size_full=tuple(args.patch_size[d]),
                                                dim_tokens=args.decoder_dim, depth=args.decoder_depth, num_heads=args.decoder_num_heads,
                                                use_task_queries=args.decoder_use_task_queries, task=d,
                                                context_tasks=list(args.in_domains), use_xattn=args.decoder_use_xattn,
                                                image_size=args.input_size[d], attn_drop_rate=decoder_attn_drop_rate)
            for d in args.out_domains
        }
    if model_kwargs:
        from .model import MIRAGEModel
        return MIRAGEModel(args, input_adapters=input_adapters, output_adapters=output_adapters,
                           num_global_tokens=args.num_global_tokens, drop_path_rate=args.drop_path, **model_kwargs)
    return model_factory[model_name](input_adapters=input_adapters, output_adapters=output_adapters,
                                     num_global_tokens=args.num_global_tokens, drop_path_rate=args.drop_path, args=args)


def get_losses(args) -> Dict[str, object]:
    """tasks_loss_fn (run_pretraining.py:465-471)."""
    return {d: DOMAIN_CONF[d]["loss"](patch_size=tuple(args.patch_size[d]), stride=DOMAIN_CONF[d]["stride_level"]) for d in args.out_domains}
