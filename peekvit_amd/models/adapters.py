"""State-dict key adapters for LOCAL torchvision / timm ViT checkpoints (reference models/adapters.py:75-166).

Pure dictionary renaming onto the reference's key contract.  The reference's download
branches need network access and are not reproduced; `load_weights` only accepts local files.
"""
from __future__ import annotations

import re
from typing import Dict

import torch

_TORCHVISION = [
    (r"^class_token$", "class_tokens"),
    (r"^encoder\.layers\.encoder_layer_(\d+)\.self_attention\.", r"encoder.layers.\1.self_attention.self_attention."),
    (r"^encoder\.layers\.encoder_layer_(\d+)\.mlp\.linear_1\.", r"encoder.layers.\1.mlp.fc1."),
    (r"^encoder\.layers\.encoder_layer_(\d+)\.mlp\.linear_2\.", r"encoder.layers.\1.mlp.fc2."),
    (r"^encoder\.layers\.encoder_layer_(\d+)\.mlp\.0\.", r"encoder.layers.\1.mlp.fc1."),
    (r"^encoder\.layers\.encoder_layer_(\d+)\.mlp\.3\.", r"encoder.layers.\1.mlp.fc2."),
    (r"^encoder\.layers\.encoder_layer_(\d+)\.", r"encoder.layers.\1."),
    (r"^heads\.head\.", "head."),
]

_TIMM = [
    (r"^cls_token$", "class_tokens"),
    (r"^pos_embed$", "encoder.pos_embedding"),
    (r"^patch_embed\.proj\.", "conv_proj."),
    (r"^blocks\.(\d+)\.norm1\.", r"encoder.layers.\1.ln_1."),
    (r"^blocks\.(\d+)\.norm2\.", r"encoder.layers.\1.ln_2."),
    (r"^blocks\.(\d+)\.attn\.qkv\.weight$", r"encoder.layers.\1.self_attention.self_attention.in_proj_weight"),
    (r"^blocks\.(\d+)\.attn\.qkv\.bias$", r"encoder.layers.\1.self_attention.self_attention.in_proj_bias"),
    (r"^blocks\.(\d+)\.attn\.proj\.", r"encoder.layers.\1.self_attention.self_attention.out_proj."),
    (r"^blocks\.(\d+)\.mlp\.fc1\.", r"encoder.layers.\1.mlp.fc1."),
    (r"^blocks\.(\d+)\.mlp\.fc2\.", r"encoder.layers.\1.mlp.fc2."),
    (r"^norm\.", "encoder.ln."),
]


def _rename(sd: Dict[str, torch.Tensor], rules, num_classes: int) -> Dict[str, torch.Tensor]:
    out = {}
    for key, val in sd.items():
        new = key
        for pat, rep in rules:
            if re.search(pat, new):
                new = re.sub(pat, rep, new)
                break
        out[new] = val
    # a classifier trained for another label set is replaced by a zero head of the requested size (reference adapters.py:109-113,
    # 159-164: `torch.zeros`, to be fine-tuned); a checkpoint without a head raises KeyError as the reference does
    width = out["head.weight"].shape[1]
    if out["head.weight"].shape[0] != num_classes:
        print("Loading weights for a different number of classes. Replacing head with random weights. You should fine-tune the model.")
        out["head.weight"] = torch.zeros((num_classes, width))
        out["head.bias"] = torch.zeros(num_classes)
    return out


def adapt_torch_state_dict(sd, num_classes: int):
    """torchvision `vit_*` checkpoint keys -> reference keys (reference models/adapters.py:75-115)."""
    return _rename(sd, _TORCHVISION, num_classes)


def adapt_timm_state_dict(sd, num_classes: int):
    """timm / DeiT checkpoint keys -> reference keys (reference models/adapters.py:118-166)."""
    return _rename(sd, _TIMM, num_classes)
