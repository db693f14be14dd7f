"""What the end-to-end parity tests of the pruned / gated models OBSERVE on an MI355X, written down (the tests used to pick
their tolerance from a branch - keep sets equal to the reference's or not - that only a dropped `print` recorded).

    python scripts/parity_observed.py gpurun_out/parity_observed.json      # on the GPU box; copy to profiles/r03_parity_observed.json

For every RankViT case of tests/test_hip_models.py::test_rankvit_parity (default precision mode) and the explicit-f16 case of
tests/test_hip_precision.py: per ranked layer whether the kept SET equals the REAL reference's (tests/golden/rankvit.npz), the fraction of
kept indices in common, and the logits' relative L2 error against the reference; for the ResidualViT cases the logits / mask errors per
budget.  The tests assert these outcomes as constants."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

from conftest import GOLDEN, rel_l2
from peekvit_amd import engine, synth
from peekvit_amd.models.rankvit import RankVisionTransformer
from peekvit_amd.models.residualvit import ResidualVisionTransformer

DEV = "cuda:0"
out = {"rankvit": {}, "residualvit": {}}
g = np.load(os.path.join(GOLDEN, "rankvit.npz"))
for mode in ("auto", "f16"):
    for name, layers, b in [("vit_micro", [0, 1], 0.5), ("vit_micro", [0, 1], 0.25), ("vit_tiny", [1, 2, 3], 0.5), ("vit_b_16", [3, 6, 9], 0.5)]:
        cfg = synth.MODEL_CONFIGS[name]
        m = RankVisionTransformer(**cfg, rankvit_layers=layers)
        synth.load_synth_weights(m, dict(cfg, rankvit_layers=layers), "vit", seed=0)
        m = m.eval().to(DEV)
        m.set_budget(b)
        x = torch.from_numpy(synth.synth_images(2, cfg["image_size"], seed=0)).to(DEV)
        with torch.no_grad(), engine.precision(mode):
            logits = m(x).cpu().numpy()
        per = {}
        for li in layers:
            got = np.sort(m.encoder.layers[li].last_keep.cpu().numpy().astype(np.int64), axis=1)
            ref = np.sort(g[f"{name}_b{b}_keep{li}"], axis=1)
            common = np.mean([len(np.intersect1d(a, r)) / len(r) for a, r in zip(got, ref)])
            per[str(li)] = {"set_equal": bool(np.array_equal(got, ref)), "fraction_in_common": float(common)}
        out["rankvit"][f"{mode}/{name}/{layers}/{b}"] = {"layers": per, "logits_rel_l2": rel_l2(logits, g[f"{name}_b{b}_logits"])}
        print(mode, name, layers, b, out["rankvit"][f"{mode}/{name}/{layers}/{b}"], flush=True)
gr = np.load(os.path.join(GOLDEN, "residualvit.npz"))
for tag, name, gb in [("vit_micro", "vit_micro", 10), ("vit_micro_gb0", "vit_micro", 0), ("vit_b_16", "vit_b_16", 10)]:
    extra = dict(gate_type="sigmoid", gate_temp=1, gate_bias=gb, add_budget_token="learnable", gate_threshold=0.5)
    cfg = synth.MODEL_CONFIGS[name]
    m = ResidualVisionTransformer(**cfg, **extra)
    synth.load_synth_weights(m, dict(cfg, **extra), "residualvit", seed=0)
    m = m.eval().to(DEV)
    x = torch.from_numpy(synth.synth_images(2, cfg["image_size"], seed=0)).to(DEV)
    for b in (0.2, 0.5, 1.0):
        m.set_budget(b)
        with torch.no_grad():
            logits = m(x).cpu().numpy()
        masks = torch.stack([blk.mask.cpu() for blk in m.encoder.layers]).numpy()
        ref = gr[f"{tag}_b{b}_logits"]
        out["residualvit"][f"{tag}/{b}"] = {"logits_rel_l2": rel_l2(logits, ref) if np.linalg.norm(ref) > 0 else None,
                                            "max_mask_error": float(np.abs(masks - gr[f"{tag}_b{b}_masks"]).max())}
        print(tag, b, out["residualvit"][f"{tag}/{b}"], flush=True)
# hostile weights (tests/golden/hostile.npz): mode auto vs unguarded fp16 vs the fallback mode, and which guard bits the forward raised
import warnings
from peekvit_amd import ops
from peekvit_amd.models.vit import VisionTransformer
gh = np.load(os.path.join(GOLDEN, "hostile.npz"))
out["hostile"] = {}
for name, variant in [("vit_tiny", "loguniform"), ("vit_tiny", "massive_token"), ("vit_tiny", "ln_gain"), ("vit_tiny", "hostile"),
                      ("vit_b_16", "loguniform"), ("vit_b_16", "hostile")]:
    cfg = synth.MODEL_CONFIGS[name]
    m = VisionTransformer(**cfg)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.hostile_variants(cfg)[variant].items()})
    m = m.eval().to(DEV)
    x = torch.from_numpy(synth.synth_images(2, cfg["image_size"], seed=0)).to(DEV)
    ref = gh[f"{name}/{variant}/logits"]
    row = {}
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        n0 = engine.fallback_count
        row["auto_rel_l2"] = rel_l2(m(x).cpu().numpy(), ref)
        row["auto_repeated_in_fallback_mode"] = engine.fallback_count - n0
        flag = torch.zeros(1, dtype=torch.int32, device=DEV)
        ops.set_range_flag(flag)
        try:
            with engine.precision("f16"):
                row["unguarded_f16_rel_l2"] = rel_l2(m(x).cpu().numpy(), ref)
        finally:
            ops.set_range_flag(None)
        row["guard_bits_raised_by_the_f16_forward"] = int(flag.item())
        for mode in ("bf16", "bf16x3"):
            with engine.precision(mode):
                row[mode + "_rel_l2"] = rel_l2(m(x).cpu().numpy(), ref)
    out["hostile"][f"{name}/{variant}"] = row
    print(name, variant, row, flush=True)
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "parity_observed.json")
os.makedirs(os.path.dirname(path), exist_ok=True)
json.dump(out, open(path, "w"), indent=1)
