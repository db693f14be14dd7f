"""Generate tests/golden/adapters.json and tests/golden/flops_hooks.json from the REAL reference.

    python oracle/make_golden_aux.py          (build container only; same import recipe as make_golden.py)

adapters.json   `adapt_torch_state_dict` / `adapt_timm_state_dict` (reference models/adapters.py:75-166) run on synthetic key sets shaped
                like torchvision's `vit_b_16` (both the `mlp.linear_1` and the older `mlp.0` naming) and timm's `vit_*`/DeiT checkpoints:
                the old-key -> new-key map, the output shapes, and whether the head was replaced by zeros (num_classes mismatch).
flops_hooks.json  the reference's two custom counter hooks (utils/flops_count.py:27-145) attached as forward hooks to every module whose
                exact type is nn.Linear / nn.MultiheadAttention (the type test ptflops applies to `custom_modules_hooks`) of the
                reference's own models, one forward, `__flops__` (MACs) summed per module type.  `ptflops` itself is absent from the image:
                a placeholder module satisfies `from ptflops import get_model_complexity_info` (flops_count.py:3), which only
                `compute_flops` would call; the conv / LayerNorm counts that come from ptflops' built-in hooks are NOT captured here.
"""
from __future__ import annotations

import json
import os
import sys
import types

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "oracle"))

import numpy as np
import torch

from make_golden import GOLD, import_reference, reference_sha256
from peekvit_amd import synth


def torchvision_keys(L, D, M, C, S, P, old_mlp_names=False):
    fc1, fc2 = ("mlp.0", "mlp.3") if old_mlp_names else ("mlp.linear_1", "mlp.linear_2")
    sd = {"class_token": (1, 1, D), "conv_proj.weight": (D, 3, P, P), "conv_proj.bias": (D,), "encoder.pos_embedding": (1, S, D)}
    for i in range(L):
        p = f"encoder.layers.encoder_layer_{i}."
        sd.update({p + "ln_1.weight": (D,), p + "ln_1.bias": (D,), p + "self_attention.in_proj_weight": (3 * D, D),
                   p + "self_attention.in_proj_bias": (3 * D,), p + "self_attention.out_proj.weight": (D, D),
                   p + "self_attention.out_proj.bias": (D,), p + "ln_2.weight": (D,), p + "ln_2.bias": (D,),
                   p + fc1 + ".weight": (M, D), p + fc1 + ".bias": (M,), p + fc2 + ".weight": (D, M), p + fc2 + ".bias": (D,)})
    sd.update({"encoder.ln.weight": (D,), "encoder.ln.bias": (D,), "heads.head.weight": (C, D), "heads.head.bias": (C,)})
    return sd


def timm_keys(L, D, M, C, S, P):
    sd = {"cls_token": (1, 1, D), "pos_embed": (1, S, D), "patch_embed.proj.weight": (D, 3, P, P), "patch_embed.proj.bias": (D,)}
    for i in range(L):
        p = f"blocks.{i}."
        sd.update({p + "norm1.weight": (D,), p + "norm1.bias": (D,), p + "attn.qkv.weight": (3 * D, D), p + "attn.qkv.bias": (3 * D,),
                   p + "attn.proj.weight": (D, D), p + "attn.proj.bias": (D,), p + "norm2.weight": (D,), p + "norm2.bias": (D,),
                   p + "mlp.fc1.weight": (M, D), p + "mlp.fc1.bias": (M,), p + "mlp.fc2.weight": (D, M), p + "mlp.fc2.bias": (D,)})
    sd.update({"norm.weight": (D,), "norm.bias": (D,), "head.weight": (C, D), "head.bias": (C,)})
    return sd


def run_adapter(fn, keys, num_classes):
    # every tensor is filled with its own ordinal so the old-key -> new-key map can be read back from the OUTPUT values
    sd = {k: torch.full(shape, float(i + 1)) for i, (k, shape) in enumerate(keys.items())}
    order = list(keys)
    out = fn(sd, num_classes)
    mapping, zero_head = {}, []
    for nk, v in out.items():
        tag = float(v.flatten()[0])
        if tag == 0.0 and float(v.abs().sum()) == 0.0:
            zero_head.append(nk)
        else:
            mapping[order[int(tag) - 1]] = nk
    return {"map": mapping, "shapes": {k: list(v.shape) for k, v in out.items()}, "zeroed": sorted(zero_head),
            "order": list(out.keys())}


def adapters_golden():
    from peekvit.models.adapters import adapt_timm_state_dict, adapt_torch_state_dict
    L, D, M, S, P = 12, 32, 64, 5, 16         # 12 layers so the two-digit `encoder_layer_1x` names are covered
    out = {}
    for tag, keys, fn in (("torch", torchvision_keys(L, D, M, 1000, S, P), adapt_torch_state_dict),
                          ("torch_old_mlp", torchvision_keys(L, D, M, 1000, S, P, old_mlp_names=True), adapt_torch_state_dict),
                          ("timm", timm_keys(L, D, M, 1000, S, P), adapt_timm_state_dict)):
        for C in (1000, 10):
            out[f"{tag}_C{C}"] = dict(run_adapter(fn, keys, C), input_shapes=[[k, list(v)] for k, v in keys.items()], num_classes=C)
    return out


def hook_counts(model, x, lin_hook, mha_hook):
    hs, mods = [], []
    for mod in model.modules():
        if type(mod) is torch.nn.Linear:
            mod.__flops__ = 0
            mods.append(("linear", mod))
            hs.append(mod.register_forward_hook(lin_hook))
        elif type(mod) is torch.nn.MultiheadAttention:
            mod.__flops__ = 0
            mods.append(("mha", mod))
            hs.append(mod.register_forward_hook(mha_hook))
    with torch.no_grad():
        model(x)
    for h in hs:
        h.remove()
    names = {id(m): n for n, m in model.named_modules()}
    per = {names[id(m)]: int(m.__flops__) for _, m in mods}
    tot = {"linear": sum(int(m.__flops__) for k, m in mods if k == "linear"), "mha": sum(int(m.__flops__) for k, m in mods if k == "mha")}
    for _, m in mods:
        del m.__flops__
        if hasattr(m, "avg_sparsity"):
            del m.avg_sparsity
    return {"per_module_macs": per, "total_macs": tot}


def flops_golden(VT, RVT, ResVT):
    pt = types.ModuleType("ptflops")
    pt.get_model_complexity_info = None
    sys.modules["ptflops"] = pt
    from peekvit.utils.flops_count import res_linear_flops_counter_hook as lin, res_multihead_attention_counter_hook as mha
    out = {}
    res_extra = dict(gate_temp=1, add_input=False, gate_type="sigmoid", gate_threshold=0.5, add_budget_token="learnable")
    for tag, cname, cls, extra, budget, batch in (
            ("vit_micro", "vit_micro", VT, {}, None, 2), ("vit_tiny", "vit_tiny", VT, {}, None, 1), ("vit_b_16", "vit_b_16", VT, {}, None, 1),
            ("rankvit_micro_b0.5", "vit_micro", RVT, {"rankvit_layers": [0, 1]}, 0.5, 2),
            ("rankvit_b_16_b0.5", "vit_b_16", RVT, {"rankvit_layers": [3, 6, 9]}, 0.5, 1),
            ("residualvit_micro_gb10_b0.5", "vit_micro", ResVT, dict(res_extra, gate_bias=10), 0.5, 2),
            ("residualvit_micro_gb0_b0.2", "vit_micro", ResVT, dict(res_extra, gate_bias=0), 0.2, 2),
            ("residualvit_micro_gb0_b0.5", "vit_micro", ResVT, dict(res_extra, gate_bias=0), 0.5, 2)):
        cfg = dict(synth.MODEL_CONFIGS[cname])
        if cls is ResVT:
            extra = dict(extra, residual_layers=["attention+mlp"] * cfg["num_layers"])
        torch.manual_seed(0)
        m = cls(**cfg, **extra).eval()
        synth.load_synth_weights(m, dict(cfg, **extra) if cls is ResVT else cfg, "residualvit" if cls is ResVT else "vit", seed=0)
        if budget is not None:
            m.set_budget(budget)
        x = torch.from_numpy(synth.synth_images(batch, cfg["image_size"], seed=0))
        r = hook_counts(m, x, lin, mha)
        r.update(batch=batch, budget=budget, config=cname, kind=cls.__name__)
        if cls is ResVT:
            r["zero_rows_per_block"] = [int((blk.mask == 0).sum()) for blk in m.encoder.layers]
        out[tag] = r
        print(f"  flops {tag}: linear {r['total_macs']['linear']} mha {r['total_macs']['mha']} MACs (batch {batch})")
    return out


def main():
    VT, RVT, ResVT = import_reference()
    sha = reference_sha256()
    with open(os.path.join(GOLD, "adapters.json"), "w") as f:
        json.dump({"reference_sha256": sha, "cases": adapters_golden()}, f, sort_keys=True, separators=(",", ":"))
    with open(os.path.join(GOLD, "flops_hooks.json"), "w") as f:
        json.dump({"reference_sha256": sha, "cases": flops_golden(VT, RVT, ResVT)}, f, indent=1, sort_keys=True)
    print("wrote adapters.json, flops_hooks.json")


if __name__ == "__main__":
    main()
