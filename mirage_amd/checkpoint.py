"""Checkpoint compatibility: reference-format files and weight import.

  * save_model / auto_load_model: `checkpoint-{epoch}.pth` = {model, optimizer, epoch, scaler, args}, latest-epoch resume
    (mutils/checkpoint.py:9-72) -- implemented in mirage_amd.run_pretraining, re-exported here.
  * interpolate_pos_embed: bicubic resampling of `input_adapters.<d>.pos_emb` when a checkpoint was trained on another
    token grid (mutils/pos_embed.py:7-22).  Load-time only; plain torch on the host side.
  * load_pretrained_encoder: MultiMAE-B style (drop adapters) and MAE ViT-L style ('blocks.' -> 'encoder.') initial weights
    (run_pretraining.py:429-446).
"""
import re

import torch

from .run_pretraining import auto_load_model, save_model  # noqa: F401


def interpolate_pos_embed(model, checkpoint_model: dict) -> None:
    pattern = r"input_adapters\.(.*)\.pos_emb"
    for key in [k for k in checkpoint_model if re.match(pattern, k)]:
        domain = re.match(pattern, key).group(1)
        if getattr(model.input_adapters, domain, None) is None:
            continue
        pe = checkpoint_model[key]
        _, _, oh, ow = pe.shape
        _, _, nh, nw = getattr(model.input_adapters, domain).pos_emb.shape
        if (oh, ow) != (nh, nw):
            checkpoint_model[key] = torch.nn.functional.interpolate(pe.float().cpu(), size=(nh, nw), mode="bicubic", align_corners=False)


def load_pretrained_encoder(model, weights_path: str, map_location="cpu"):
    weights = torch.load(weights_path, map_location=map_location, weights_only=False)["model"]
    if "_vit_large" in weights_path:
        weights = {k.replace("blocks.", "encoder."): v for k, v in weights.items()}
    else:
        weights = {k: v for k, v in weights.items() if "input_adapter" not in k and "output_adapter" not in k}
    return model.load_state_dict(weights, strict=False)
