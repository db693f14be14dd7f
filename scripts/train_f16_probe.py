"""Round 5 diagnostics: the fp16-operand, loss-scaled training step against the reference's golden training step (tests/golden/train_step.npz)
next to the bf16-operand one: forward logits, loss, gradient errors, the loss scale chosen and how much of the fp16 range the internal gradients use.
    python scripts/train_f16_probe.py [model ...]"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from peekvit_amd import synth, train_engine  # noqa: E402
from peekvit_amd.models.rankvit import RankVisionTransformer  # noqa: E402
from peekvit_amd.models.vit import VisionTransformer  # noqa: E402


def rel(a, b):
    a = a.detach().double().cpu().numpy() if torch.is_tensor(a) else np.asarray(a, dtype=np.float64)
    b = b.detach().double().cpu().numpy() if torch.is_tensor(b) else np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def run(name, batch, operand, g):
    train_engine._TRAIN_OPERAND = operand
    train_engine._DX1_16 = os.environ.get("PEEKVIT_AMD_TRAIN_DX1", "16") == "16"
    if name.startswith("rankvit"):
        cfg = synth.MODEL_CONFIGS[name.replace("rankvit", "vit")]
        m = RankVisionTransformer(**cfg, rankvit_layers=[3, 6, 9] if name.endswith("b_16") else [1])
        m.set_budget(0.5)
    else:
        cfg = synth.MODEL_CONFIGS[name]
        m = VisionTransformer(**cfg)
    synth.load_synth_weights(m, cfg)
    m = m.cuda().train()
    x = torch.from_numpy(synth.synth_images(batch, cfg["image_size"], seed=0)).cuda()
    y = (torch.arange(batch) % cfg["num_classes"]).cuda()
    train_engine.debug_amax = []
    logits = m(x)
    loss = torch.nn.functional.cross_entropy(logits, y)
    loss.backward()
    st = train_engine.train_state(m)
    named = dict(m.named_parameters())
    names = [str(n) for n in g[f"{name}/names"]]
    ref = g[f"{name}/grad_norms"]
    total_ref = float(g[f"{name}/total_norm"])
    out = {"model": name, "operand": operand, "used": train_engine.pass_operand(m), "scale": st.scale, "skipped": st.skipped,
           "logits_rel_l2": rel(logits, g[f"{name}/logits"]), "loss_rel": abs(loss.item() - float(g[f"{name}/loss"])) / float(g[f"{name}/loss"])}
    if named[names[0]].grad is not None:
        gn = np.array([float(named[n].grad.norm()) for n in names])
        out["grad_norm_worst_rel"] = float(np.max(np.abs(gn - ref) / (ref + 1e-2 * total_ref)))
        out["grad_norm_worst_rel_big"] = float(np.max((np.abs(gn - ref) / ref)[ref > 1e-3 * total_ref]))
        torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)           # the stored complete gradients are post-clip (max_norm 1.0)
        out["grad_full_worst_rel_l2"] = max(rel(named[k.split("/grad/")[1]].grad, g[k]) for k in g.files if k.startswith(f"{name}/grad/"))
        out["grad_full"] = {k.split("/grad/")[1]: float(f"{rel(named[k.split('/grad/')[1]].grad, g[k]):.2e}") for k in g.files if k.startswith(f"{name}/grad/")}
    am = train_engine.debug_amax
    if am:
        out["amax_scaled"] = {k: float(f"{max(a[k] for a in am):.3g}") for k in am[0]}
        out["amax_min_over_blocks"] = {k: float(f"{min(a[k] for a in am):.3g}") for k in am[0]}
    train_engine.debug_amax = None
    return out


if __name__ == "__main__":
    g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "train_step.npz"))
    cases = [("vit_micro", 6), ("vit_tiny", 3), ("rankvit_micro", 6), ("vit_b_16", 2), ("rankvit_b_16", 2)]
    want = sys.argv[1:]
    for name, batch in cases:
        if want and name not in want:
            continue
        for operand in ("bf16", "f16"):
            print(json.dumps(run(name, batch, operand, g)), flush=True)
