"""tests/golden/hostile.npz: the REAL reference (/root/reference, CPU fp32) on numerically hostile weights (round 3).

    python oracle/make_golden_hostile.py            # build container only: the reference never travels to the GPU box

Round 2 demonstrated the 1e-3 contract of the fp16-operand default only on benign synthetic weights (N(0, 0.02)-class).  These
fixtures hold the reference's logits / per-block class-token rows for vit_tiny and vit_b_16 (B = 2) with peekvit_amd.synth.hostile_state_dict:
log-uniform weight magnitudes over six decades, outlier channels (x100 LayerNorm gains, x100 fc1 rows), one massive token (positional
row x 5e4) - and for three single-ingredient variants of vit_tiny, so that a test can tell WHICH property a path cannot carry.  The
x100 LayerNorm gains drive the attention scores to |s| ~ 1e3: no 16-bit rounding of q and k survives that softmax, which is what the
attention-score guard (include/peekvit_hip.h, PV_SCORE_LIMIT) exists for.
Same import recipe and source-path assertion as oracle/make_golden.py (import_reference)."""
from __future__ import annotations

import os
import sys

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import numpy as np
import torch

from oracle.make_golden import GOLD, import_reference, reference_sha256
from peekvit_amd import synth


def run(VT, name, sd, batch=2):
    cfg = synth.MODEL_CONFIGS[name]
    m = VT(**cfg).eval()
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()}, strict=True)
    x = torch.from_numpy(synth.synth_images(batch, cfg["image_size"], seed=0))
    rows, scores = [], []
    hooks = [blk.register_forward_hook(lambda mod, i, o: rows.append(o[:, 0].detach().clone())) for blk in m.encoder.layers]
    with torch.no_grad():
        logits = m(x)
    for h in hooks:
        h.remove()
    return logits.numpy(), torch.stack(rows).numpy()


def main():
    VT, _, _ = import_reference()
    torch.set_num_threads(8)
    out = {}
    path = os.path.join(GOLD, "hostile.npz")
    only = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--only=")]
    if only and os.path.exists(path):
        # add variants to the fixture file without touching the entries it already holds (round 5: --only=trained_like)
        out = {k: v for k, v in np.load(path).items()}
    for name, which in (("vit_tiny", ("hostile", "loguniform", "ln_gain", "massive_token", "trained_like")), ("vit_b_16", ("hostile", "loguniform", "trained_like"))):
        vs = synth.hostile_variants(synth.MODEL_CONFIGS[name])
        for v in which:
            if only and v not in only:
                continue
            logits, rows = run(VT, name, vs[v])
            out[f"{name}/{v}/logits"], out[f"{name}/{v}/block_cls"] = logits, rows
            print(f"  {name} {v}: |logits| mean {np.abs(logits).mean():.4f}  max |class row| {np.abs(rows).max():.3g}")
    np.savez_compressed(path, **out)
    import json
    with open(os.path.join(GOLD, "hostile_meta.json"), "w") as f:
        json.dump({"reference_sha256": reference_sha256(), "torch": torch.__version__}, f, indent=1, sort_keys=True)
    print("wrote", os.path.join(GOLD, "hostile.npz"))


if __name__ == "__main__":
    main()
