"""Checkpoints in the reference's on-disk format (utils/utils.py:198-256): a dict with model_class, noise_args, model_args,
state_dict, optimizer, epoch at <dir>/checkpoints/epoch_NNN.pth - loadable by a stock peekvit checkout and vice versa
(state-dict keys/shapes are identical, tests/test_host_contract.py)."""
from __future__ import annotations

import os
from typing import Optional

import torch


def save_state(path: str, model: torch.nn.Module, model_args: dict, noise_args=None, optimizer=None, epoch: int = 0,
               skip_optimizer: bool = False) -> str:
    """The reference's dict (utils/utils.py:198-215).  `model_args` is stored as given - the reference stores `dict(cfg.model)`,
    `_target_` and the pretrained-weight keys included, and strips them when loading.  `skip_optimizer=True` is the reference's
    default (no optimizer state on disk); here the optimizer is kept when one is passed, so a run can resume."""
    ckpt_dir = os.path.join(path, "checkpoints")
    os.makedirs(ckpt_dir, exist_ok=True)
    file = os.path.join(ckpt_dir, f"epoch_{epoch:03d}.pth")
    torch.save({"model_class": type(model).__name__, "noise_args": dict(noise_args) if noise_args else None,
                "model_args": _plain(model_args) if model_args else None, "state_dict": model.state_dict(),
                "optimizer": optimizer.state_dict() if optimizer is not None and not skip_optimizer else None, "epoch": epoch}, file)
    return file


def _plain(obj):
    """dict / list containers (incl. OmegaConf DictConfig / ListConfig of a checkpoint written through Hydra) -> builtin types."""
    if hasattr(obj, "items"):
        return {str(k): _plain(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)) or type(obj).__name__ == "ListConfig":
        return [_plain(v) for v in obj]
    return obj


# keys the reference pops from model_args before it rebuilds the model (utils/utils.py:236-238): Hydra's class path and the
# pretrained-weight sources (the state dict in the checkpoint supersedes them, and they may name files that are not there)
_NOT_CONSTRUCTOR_ARGS = ("_target_", "torch_pretrained_weights", "timm_pretrained_weights")


def get_checkpoint_path(experiment_dir: str) -> Optional[str]:
    """Lexicographically last checkpoint (the reference's rule, utils/utils.py:260-285)."""
    d = os.path.join(experiment_dir, "checkpoints")
    files = sorted(f for f in os.listdir(d) if f.endswith(".pth")) if os.path.isdir(d) else []
    return os.path.join(d, files[-1]) if files else None


def load_state(file: str, model: Optional[torch.nn.Module] = None, optimizer=None, strict: bool = False):
    """Returns (model, state): rebuilds the model from model_class / model_args when none is given, exactly as the reference does
    (utils/utils.py:218-256): `_target_` and the pretrained-weight keys are dropped from model_args first, the state dict is loaded
    with strict=False by default, and a given optimizer receives the stored optimizer state.  Accepts checkpoints written by a stock
    peekvit checkout (model_args = dict(cfg.model))."""
    state = torch.load(file, map_location="cpu", weights_only=False)
    if model is None:
        from peekvit_amd.models import rankvit, residualvit, vit
        classes = {c.__name__: c for c in (vit.VisionTransformer, rankvit.RankVisionTransformer, residualvit.ResidualVisionTransformer)}
        if state["model_class"] not in classes:
            raise ValueError(f"checkpoint of class {state['model_class']!r}: only {sorted(classes)} are built here")
        args = {k: v for k, v in _plain(state["model_args"] or {}).items() if k not in _NOT_CONSTRUCTOR_ARGS}
        model = classes[state["model_class"]](**args)
    try:
        res = model.load_state_dict(state["state_dict"], strict=strict)
        if len(res[0]) > 0:
            print("Some parameters are not present in the checkpoint and will be randomly initialized: ", res[0])
    except RuntimeError as e:
        if strict:
            raise
        # the reference reports a checkpoint of another architecture and carries on (utils/utils.py:241-250)
        print(e)
        print("The model state dict could not be loaded. This is probably because the checkpoint has a different architecture.")
        print("Checkpoint class: ", state["model_class"], " Model class: ", type(model).__name__, " Checkpoint args: ", state["model_args"])
    if optimizer is not None and state.get("optimizer") is not None:
        optimizer.load_state_dict(state["optimizer"])
    return model, state
