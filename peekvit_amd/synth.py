"""Deterministic synthetic weights and images for the ViT encoder hot path.

There is no dataset and no pretrained checkpoint reachable (no network), and the
GPU box never sees /root/reference, so every test, the golden-vector script and
bench.py need tensors that can be REGENERATED bit-identically anywhere from a
name and a seed.  This module owns that generator:

    value(name, flat_index) = f(splitmix64(fnv1a(name) ^ seed) + flat_index)

All values are rounded to bf16-representable fp32, so the MI355X path (bf16 MFMA
operands) and the fp32 oracle see *identical* weights/inputs and differences are
due to arithmetic only (SURVEY.md section 7 H1, section 8c golden vectors (2)).

Parameter names/shapes follow the reference state-dict contract (SURVEY.md
section 8b; reference models/vit.py:104-199, models/residualvit.py:390-500).
The reference zero-initialises `head` and `class_tokens` (vit.py:165,186-188)
which makes logits identically zero; like SURVEY appendix A.1 prescribes, the
synthetic state dict re-draws them N(0, 0.02).
"""
from __future__ import annotations

import math
from typing import Dict, Iterable, Optional, Sequence, Tuple

import numpy as np

_MASK64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _fnv1a64(name: str) -> int:
    h = 0xCBF29CE484222325
    for b in name.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _splitmix64(x: np.ndarray) -> np.ndarray:
    """Vectorised splitmix64 finaliser on uint64 arrays (wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def hash_uniform(name: str, n: int, seed: int = 0, stream: int = 0) -> np.ndarray:
    """n float64 values in [0, 1), a pure function of (name, seed, stream, index)."""
    key = (_fnv1a64(name) ^ (seed * 0x9E3779B97F4A7C15) ^ (stream * 0xD1B54A32D192ED03)) & 0xFFFFFFFFFFFFFFFF
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) + np.uint64(key)
    bits = _splitmix64(_splitmix64(idx))
    return (bits >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def hash_normal(name: str, n: int, seed: int = 0) -> np.ndarray:
    """n float64 standard normals (Box-Muller on two hashed uniform streams)."""
    u1 = hash_uniform(name, n, seed, stream=1)
    u2 = hash_uniform(name, n, seed, stream=2)
    u1 = np.maximum(u1, 2.0 ** -53)
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * math.pi * u2)


def round_to_bf16(a: np.ndarray) -> np.ndarray:
    """Round fp32 values to the nearest bf16 (ties to even), returned as fp32."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    u = a.view(np.uint32)
    with np.errstate(over="ignore"):
        r = (u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000)
    return r.view(np.float32)


def tensor(name: str, shape: Sequence[int], kind: str, scale: float = 1.0, shift: float = 0.0,
           seed: int = 0, bf16: bool = True) -> np.ndarray:
    n = int(np.prod(shape)) if len(shape) else 1
    if kind == "normal":
        v = hash_normal(name, n, seed) * scale + shift
    elif kind == "uniform":  # U(-scale, scale) + shift
        v = (hash_uniform(name, n, seed) * 2.0 - 1.0) * scale + shift
    else:
        raise ValueError(kind)
    v = v.astype(np.float32).reshape(tuple(shape))
    return round_to_bf16(v) if bf16 else v


# ----------------------------------------------------------------------------------------------
# model configurations named by BASELINE.json (reference configs/model/*.yaml)
# ----------------------------------------------------------------------------------------------
MODEL_CONFIGS: Dict[str, dict] = {
    # micro model for per-op full-tensor fixtures (widened so D is MFMA friendly)
    "vit_micro": dict(image_size=32, patch_size=8, num_layers=2, num_heads=2, hidden_dim=128, mlp_dim=256,
                      num_classes=10),
    # reference configs/model/vit_tiny.yaml:1-8 on imagenette (configs/dataset/imagenette.yaml: 160, 10)
    "vit_tiny": dict(image_size=160, patch_size=8, num_layers=4, num_heads=8, hidden_dim=256, mlp_dim=768,
                     num_classes=10),
    # reference configs/model/vit_small.yaml:1-8 at 224 / 1000 classes (BASELINE config 2)
    "vit_small": dict(image_size=224, patch_size=16, num_layers=8, num_heads=8, hidden_dim=384, mlp_dim=1536,
                      num_classes=1000),
    # reference configs/model/vit_b_16.yaml:1-9 (BASELINE config 3, the headline)
    "vit_b_16": dict(image_size=224, patch_size=16, num_layers=12, num_heads=12, hidden_dim=768, mlp_dim=3072,
                     num_classes=1000),
}


def seq_length(cfg: dict) -> int:
    return (cfg["image_size"] // cfg["patch_size"]) ** 2 + cfg.get("num_class_tokens", 1) + cfg.get("num_registers", 0)


def state_dict_spec(cfg: dict, variant: str = "vit") -> Iterable[Tuple[str, Tuple[int, ...], str, float, float]]:
    """Yield (name, shape, kind, scale, shift) for every parameter of the reference module tree.

    Scales mimic PyTorch's default initialisers closely enough to give realistic activation
    statistics; exact init parity is irrelevant because tests load these tensors into both sides.
    """
    D, M, L, P = cfg["hidden_dim"], cfg["mlp_dim"], cfg["num_layers"], cfg["patch_size"]
    C = cfg["num_classes"]
    nc, nr = cfg.get("num_class_tokens", 1), cfg.get("num_registers", 0)
    S = seq_length(cfg)
    fan = 3 * P * P
    yield "class_tokens", (1, nc, D), "normal", 0.02, 0.0
    if nr > 0:
        yield "register_tokens", (1, nr, D), "normal", 0.02, 0.0
    if variant == "residualvit" and cfg.get("add_budget_token") == "learnable":
        yield "learnable_budget_token_1", (1, 1, D), "normal", 1.0, 0.0
    yield "conv_proj.weight", (D, 3, P, P), "normal", math.sqrt(1.0 / fan), 0.0
    yield "conv_proj.bias", (D,), "uniform", 0.02, 0.0
    yield "encoder.pos_embedding", (1, S, D), "normal", 0.02, 0.0
    for i in range(L):
        p = f"encoder.layers.{i}."
        if variant == "residualvit":
            skip = (cfg.get("residual_layers") or ["attention+mlp"] * L)[i]
            if skip in ("attention", "mlp", "attention+mlp"):
                yield p + "residual_gate.projection.weight", (1, D), "uniform", 1.0 / math.sqrt(D), 0.0
                yield p + "residual_gate.projection.bias", (1,), "uniform", 1.0 / math.sqrt(D), 0.0
        yield p + "ln_1.weight", (D,), "uniform", 0.1, 1.0
        yield p + "ln_1.bias", (D,), "uniform", 0.05, 0.0
        yield p + "self_attention.self_attention.in_proj_weight", (3 * D, D), "uniform", math.sqrt(6.0 / (4 * D)), 0.0
        yield p + "self_attention.self_attention.in_proj_bias", (3 * D,), "uniform", 0.02, 0.0
        yield p + "self_attention.self_attention.out_proj.weight", (D, D), "uniform", 1.0 / math.sqrt(D), 0.0
        yield p + "self_attention.self_attention.out_proj.bias", (D,), "uniform", 0.02, 0.0
        yield p + "ln_2.weight", (D,), "uniform", 0.1, 1.0
        yield p + "ln_2.bias", (D,), "uniform", 0.05, 0.0
        yield p + "mlp.fc1.weight", (M, D), "uniform", 1.0 / math.sqrt(D), 0.0
        yield p + "mlp.fc1.bias", (M,), "uniform", 1.0 / math.sqrt(D), 0.0
        yield p + "mlp.fc2.weight", (D, M), "uniform", 1.0 / math.sqrt(M), 0.0
        yield p + "mlp.fc2.bias", (D,), "uniform", 1.0 / math.sqrt(M), 0.0
        if variant == "residualvit" and cfg.get("add_budget_token") == "learnable":
            yield p + "budget_token_gate.weight", (1, D), "uniform", 1.0 / math.sqrt(D), 0.0
            yield p + "budget_token_gate.bias", (1,), "uniform", 1.0 / math.sqrt(D), 0.0
    yield "encoder.ln.weight", (D,), "uniform", 0.1, 1.0
    yield "encoder.ln.bias", (D,), "uniform", 0.05, 0.0
    yield "head.weight", (C, D), "normal", 0.02, 0.0
    yield "head.bias", (C,), "uniform", 0.02, 0.0


def synth_state_dict(cfg: dict, variant: str = "vit", seed: int = 0) -> Dict[str, np.ndarray]:
    """bf16-representable fp32 numpy arrays keyed by the reference state-dict names."""
    return {name: tensor(name, shape, kind, scale, shift, seed)
            for name, shape, kind, scale, shift in state_dict_spec(cfg, variant)}


# E[10^(-12 u)], u ~ U[0, 1): mean square of a log-uniform magnitude over six decades
_LOGU6_MS = (1.0 - 1e-12) / (12.0 * math.log(10.0))


def hostile_state_dict(cfg: dict, seed: int = 0) -> Dict[str, np.ndarray]:
    """The plain-ViT synthetic state dict made numerically HOSTILE to 16-bit operands (round 3, tests/golden/hostile.npz), still a pure
    function of (name, seed) and bf16-representable:
      * every weight MATRIX (conv_proj, in-proj, out-proj, fc1, fc2, head) keeps its signs and its overall RMS but draws its
        magnitudes log-uniformly over six decades, |w| = c * 10^(-6u), u ~ U[0,1): a third of the elements sit below fp16's smallest
        normal number relative to the largest ones;
      * three channels of every LayerNorm gain and four rows of every fc1 are multiplied by 100 (outlier channels);
      * one token is massive: positional-embedding row 5 is multiplied by 5e4 (values ~ N(0, 1e3)), so that token's residual stream
        is three orders of magnitude above its neighbours' through the whole encoder."""
    sd = synth_state_dict(cfg, "vit", seed)
    D = cfg["hidden_dim"]
    out = {}
    for name, w in sd.items():
        v = w.astype(np.float64)
        if w.ndim >= 2 and name not in ("class_tokens", "register_tokens", "encoder.pos_embedding"):
            u = hash_uniform("hostile/" + name, w.size, seed).reshape(w.shape)
            rms = math.sqrt(float(np.mean(v * v)))
            v = np.sign(v) * (10.0 ** (-6.0 * u)) * (rms / math.sqrt(_LOGU6_MS))
            if name.endswith("mlp.fc1.weight"):
                v[[1, 50, 101, v.shape[0] - 3]] *= 100.0
        elif name.endswith(("ln_1.weight", "ln_2.weight", "encoder.ln.weight")):
            v[[3, 77 % D, D - 5]] *= 100.0
        elif name == "encoder.pos_embedding":
            v[0, 5] *= 5e4
        out[name] = round_to_bf16(v.astype(np.float32))
    return out


def hostile_variants(cfg: dict, seed: int = 0) -> Dict[str, Dict[str, np.ndarray]]:
    """name -> state dict: the full hostile set and its ingredients one at a time (tests/golden/hostile.npz holds the reference's outputs)."""
    base = synth_state_dict(cfg, "vit", seed)
    host = hostile_state_dict(cfg, seed)
    is_mat = lambda k, v: v.ndim >= 2 and k not in ("class_tokens", "register_tokens", "encoder.pos_embedding")
    logu = {k: (host[k].copy() if is_mat(k, v) else v) for k, v in base.items()}
    for k in logu:
        if k.endswith("mlp.fc1.weight"):
            rows = [1, 50, 101, logu[k].shape[0] - 3]
            logu[k][rows] = round_to_bf16(logu[k][rows] / 100.0)                 # the magnitudes alone, without the x100 rows
    gains = {k: (host[k] if k.endswith(("ln_1.weight", "ln_2.weight", "encoder.ln.weight")) else v) for k, v in base.items()}
    massive = dict(base, **{"encoder.pos_embedding": host["encoder.pos_embedding"]})
    return {"hostile": host, "loguniform": logu, "ln_gain": gains, "massive_token": massive, "trained_like": trained_like_state_dict(cfg, seed)}


def trained_like_state_dict(cfg: dict, seed: int = 0) -> Dict[str, np.ndarray]:
    """What TRAINED ViTs look like to a 16-bit path, without the x100 gains of the hostile set (round 5): in every third encoder
    layer the q and k rows of the in-projection are scaled up so that the attention logits reach ~40 - 80 (sharp heads), and two channels of the
    residual stream carry massive activations (a constant ~40 x the typical magnitude on every token, through the positional embedding - the
    'massive activation' channels of pretrained ViTs).  Everything else is the benign synthetic model.  bf16-representable like the rest."""
    sd = {k: v.copy() for k, v in synth_state_dict(cfg, "vit", seed).items()}
    D, L = cfg["hidden_dim"], cfg["num_layers"]
    for i in range(L):
        if i % 3 == 1:
            w = sd[f"encoder.layers.{i}.self_attention.self_attention.in_proj_weight"]
            b = sd[f"encoder.layers.{i}.self_attention.self_attention.in_proj_bias"]
            w[:2 * D] = round_to_bf16(w[:2 * D] * 5.0)
            b[:2 * D] = round_to_bf16(b[:2 * D] * 5.0)
    pos = sd["encoder.pos_embedding"]
    for c in (5 % D, (D // 2 + 13) % D):
        pos[:, :, c] = round_to_bf16(pos[:, :, c] + 40.0)               # (the tokens themselves are ~1)
    return sd


def synth_images(batch: int, image_size: int, seed: int = 0, name: str = "images") -> np.ndarray:
    """[B,3,R,R] fp32 ~ N(0,1), bf16-representable (ImageNet-normalised images are ~zero-mean/unit-var)."""
    return tensor(f"{name}/{image_size}", (batch, 3, image_size, image_size), "normal", 1.0, 0.0, seed)


def load_synth_weights(model, cfg: dict, variant: str = "vit", seed: int = 0, strict: bool = True):
    """Load the synthetic state dict into a torch module that follows the reference key contract."""
    import torch
    sd = {k: torch.from_numpy(v.copy()) for k, v in synth_state_dict(cfg, variant, seed).items()}
    return model.load_state_dict(sd, strict=strict)


def fwd_flops_per_image(cfg: dict, seq_per_layer: Optional[Sequence[int]] = None) -> float:
    """GEMM-only forward FLOPs per image: SURVEY.md section 8d / BASELINE.md section 3 formula."""
    D, M, L, P, C = cfg["hidden_dim"], cfg["mlp_dim"], cfg["num_layers"], cfg["patch_size"], cfg["num_classes"]
    Np = (cfg["image_size"] // P) ** 2
    S0 = seq_length(cfg)
    seqs = list(seq_per_layer) if seq_per_layer is not None else [S0] * L
    f = Np * 3 * P * P * D * 2.0
    for S in seqs:
        f += S * D * 3 * D * 2.0 + 2 * S * S * D * 2.0 + S * D * D * 2.0 + 2 * S * D * M * 2.0
    f += D * C * 2.0
    return f
