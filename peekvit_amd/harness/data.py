"""Synthetic stand-in for the reference's dataset wrappers (data/*.py expose .train_dataset / .val_dataset; the real ones
download over HTTP, which is unavailable): N(0,1) images of the configured size, uniform labels, seeded."""
from __future__ import annotations

import torch
from torch.utils.data import Dataset


class _Synth(Dataset):
    def __init__(self, n: int, image_size: int, num_classes: int, seed: int):
        g = torch.Generator().manual_seed(seed)
        self.x = torch.randn(n, 3, image_size, image_size, generator=g)
        self.y = torch.randint(0, num_classes, (n,), generator=g)

    def __len__(self):
        return self.x.shape[0]

    def __getitem__(self, i):
        return self.x[i], self.y[i]


class SyntheticImages:
    def __init__(self, image_size: int, num_classes: int, train_size: int = 64, val_size: int = 64, seed: int = 0, **_):
        self.image_size, self.num_classes = image_size, num_classes
        self.train_dataset = _Synth(train_size, image_size, num_classes, seed)
        self.val_dataset = _Synth(val_size, image_size, num_classes, seed + 1)
        self.denormalize_transform = lambda t: t
