"""A ~100-line stand-in for the part of Hydra/OmegaConf the reference's scripts use (hydra-core is not installed here):
root YAML with a `defaults` list, config groups, `_target_` instantiation, `${a.b}` interpolation, CLI overrides
(`group=name` swaps a group file, `a.b.c=value` sets a leaf) - the syntax of README.md:63-70 of the reference."""
from __future__ import annotations

import copy
import importlib
import os
import re
from typing import Any, Dict, List, Sequence

import yaml

CONFIG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "configs")
_INTERP = re.compile(r"\$\{([^}]+)\}")


def _load_yaml(path: str) -> dict:
    with open(path) as f:
        return yaml.safe_load(f) or {}


def _set(cfg: dict, dotted: str, value: Any):
    keys = dotted.split(".")
    for k in keys[:-1]:
        cfg = cfg.setdefault(k, {})
    cfg[keys[-1]] = value


def _get(cfg: dict, dotted: str) -> Any:
    for k in dotted.split("."):
        cfg = cfg[k]
    return cfg


def _resolve(node: Any, root: dict, depth: int = 0) -> Any:
    if depth > 16:
        raise ValueError("interpolation cycle")
    if isinstance(node, dict):
        return {k: _resolve(v, root, depth) for k, v in node.items()}
    if isinstance(node, list):
        return [_resolve(v, root, depth) for v in node]
    if isinstance(node, str):
        m = _INTERP.fullmatch(node)
        if m:                                         # whole-value reference keeps the referenced type
            return _resolve(_get(root, m.group(1)), root, depth + 1)
        return _INTERP.sub(lambda mm: str(_resolve(_get(root, mm.group(1)), root, depth + 1)), node)
    return node


def load_config(config_name: str, overrides: Sequence[str] = (), config_dir: str = CONFIG_DIR) -> Dict[str, Any]:
    root = _load_yaml(os.path.join(config_dir, config_name + ".yaml"))
    defaults: List[Any] = root.pop("defaults", [])
    groups = {}
    for d in defaults:
        if isinstance(d, dict):
            groups.update(d)
    leaf_overrides = []
    for ov in overrides:
        key, _, val = ov.partition("=")
        if "." not in key and os.path.isdir(os.path.join(config_dir, key)):
            groups[key] = val                          # group override: model=vit_b_16
        else:
            leaf_overrides.append((key, yaml.safe_load(val)))
    cfg: Dict[str, Any] = {}
    for group, name in groups.items():
        if name in (None, "null"):
            continue
        cfg[group] = _load_yaml(os.path.join(config_dir, group, f"{name}.yaml"))
    for k, v in root.items():                          # `_self_` last: the root file overrides group content
        if isinstance(v, dict) and isinstance(cfg.get(k), dict):
            cfg[k].update(v)
        else:
            cfg[k] = v
    for key, val in leaf_overrides:
        _set(cfg, key, val)
    return _resolve(cfg, cfg)


def instantiate(node: dict, **kwargs) -> Any:
    """`_target_`-style construction (hydra.utils.instantiate for the flat cases the reference uses)."""
    node = copy.deepcopy(node)
    target = node.pop("_target_")
    module, _, attr = target.rpartition(".")
    node.update(kwargs)
    return getattr(importlib.import_module(module), attr)(**node)
