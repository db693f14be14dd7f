"""`add_noise`: splice a NoiseBlock into a model's encoder (reference utils/utils.py:162-191)."""
from __future__ import annotations

from collections import OrderedDict

import torch

from ..models.blocks import NoiseBlock


def add_noise(model, layer: int, noise_type: str, std: float = None, snr: float = None, prob: float = None, **kwargs) -> NoiseBlock:
    """Insert a NoiseBlock in front of encoder block `layer` and return it.  `model.encoder.layers` is rebuilt as a new nn.Sequential
    around the SAME block objects - positional for a plain Sequential, under the name 'noise' when the layers carry names (the
    reference's OrderedDict case) - so the blocks on both sides still dispatch to the MI355X kernels, and the harness can later call
    `.set_value()` on the returned module."""
    module = NoiseBlock(noise_type=noise_type, std=std, snr=snr, prob=prob)
    old = model.encoder.layers
    named = list(old.named_children())
    if any(not name.isdigit() for name, _ in named):
        named.insert(layer, ("noise", module))
        model.encoder.layers = torch.nn.Sequential(OrderedDict(named))
    else:
        mods = [m for _, m in named]
        mods.insert(layer, module)
        model.encoder.layers = torch.nn.Sequential(*mods)
    return module
