"""Checkpoints in the reference's on-disk format (utils/utils.py:198-256): a dict with model_class, noise_args, model_args,
state_dict, optimizer, epoch at <dir>/checkpoints/epoch_NNN.pth - loadable by a stock peekvit checkout and vice versa
(state-dict keys/shapes are identical, tests/test_host_contract.py)."""
from __future__ import annotations

import os
from typing import Optional

import torch


def save_state(path: str, model: torch.nn.Module, model_args: dict, noise_args=None, optimizer=None, epoch: int = 0) -> str:
    ckpt_dir = os.path.join(path, "checkpoints")
    os.makedirs(ckpt_dir, exist_ok=True)
    file = os.path.join(ckpt_dir, f"epoch_{epoch:03d}.pth")
    torch.save({"model_class": type(model).__name__, "noise_args": noise_args, "model_args": dict(model_args),
                "state_dict": model.state_dict(), "optimizer": optimizer.state_dict() if optimizer is not None else None,
                "epoch": epoch}, file)
    return file


def get_checkpoint_path(experiment_dir: str) -> Optional[str]:
    """Lexicographically last checkpoint (the reference's rule, utils/utils.py:260-285)."""
    d = os.path.join(experiment_dir, "checkpoints")
    files = sorted(f for f in os.listdir(d) if f.endswith(".pth")) if os.path.isdir(d) else []
    return os.path.join(d, files[-1]) if files else None


def load_state(file: str, model: Optional[torch.nn.Module] = None, strict: bool = True):
    """Returns (model, state): rebuilds the model from model_class/model_args when none is given."""
    state = torch.load(file, map_location="cpu", weights_only=False)
    if model is None:
        from peekvit_amd.models import rankvit, residualvit, vit
        classes = {c.__name__: c for c in (vit.VisionTransformer, rankvit.RankVisionTransformer, residualvit.ResidualVisionTransformer)}
        model = classes[state["model_class"]](**state["model_args"])
    model.load_state_dict(state["state_dict"], strict=strict)
    return model, state
