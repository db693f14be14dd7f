"""Generate tests/golden/train_step.npz: ONE training step of the REAL reference model.

    python oracle/make_golden_train.py        (build container only; same import recipe as make_golden.py)

The reference's step (train/train.py:112-121) is `out = model(batch); loss = CrossEntropyLoss()(out, labels); loss.backward();
clip_grad_norm_(model.parameters(), 1.0); optimizer.step()` with Adam(lr 1e-3) (configs/optimizer/adam.yaml).  train.py itself needs
hydra / torchmetrics (absent), so the step is driven here around the reference's own module class, in train mode.  Stored per
model: logits, loss, per-parameter gradient L2 norms, the total norm, a few complete gradients, and per-parameter checksums of
the weights after the clipped Adam step.  Inputs / weights / labels are pure functions (peekvit_amd.synth, labels = i mod C).
"""
from __future__ import annotations

import os
import sys

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "oracle"))

import numpy as np
import torch

from make_golden import GOLD, import_reference
from peekvit_amd import synth

FULL_GRADS_BIG = ("head.bias", "class_tokens", "conv_proj.bias", "encoder.layers.0.ln_1.weight", "encoder.layers.0.ln_1.bias",
                  "encoder.layers.0.self_attention.self_attention.in_proj_bias", "encoder.layers.1.mlp.fc2.bias", "encoder.ln.weight",
                  "encoder.layers.11.mlp.fc1.bias", "encoder.layers.6.self_attention.self_attention.out_proj.bias",
                  "encoder.layers.3.ln_2.weight")          # ViT-B/16-sized steps: vectors only (head.weight alone would be 3 MB)
FULL_GRADS = ("head.weight", "head.bias", "class_tokens", "conv_proj.bias", "encoder.layers.0.ln_1.weight", "encoder.layers.0.ln_1.bias",
              "encoder.layers.0.self_attention.self_attention.in_proj_bias", "encoder.layers.1.mlp.fc2.bias", "encoder.ln.weight")


def main():
    VT, RVT, ResVT = import_reference()
    res_extra = dict(residual_layers=["attention+mlp"] * 2, gate_temp=1, add_input=False, gate_type="sigmoid", gate_threshold=0.5,
                     gate_bias=10, add_budget_token="learnable")
    out = {}
    # (tag, config, batch, class, extra kwargs, budget): the last row is RankViT (rankvit.py:55-101) pruning to half the tokens in layer 1
    # vit_b_16 / rankvit_b_16 ([3,6,9] @ 0.5): BASELINE configs[2] / [3] as a whole fwd+bwd step at ViT-B/16 size (D 768, dh 64, S 197)
    for name, cname, batch, cls, extra, budget in (("vit_micro", "vit_micro", 6, VT, {}, None), ("vit_tiny", "vit_tiny", 3, VT, {}, None),
                                                   ("rankvit_micro", "vit_micro", 6, RVT, {"rankvit_layers": [1]}, 0.5),
                                                   ("residualvit_micro", "vit_micro", 6, ResVT, res_extra, 0.5),
                                                   ("vit_b_16", "vit_b_16", 2, VT, {}, None),
                                                   ("rankvit_b_16", "vit_b_16", 2, RVT, {"rankvit_layers": [3, 6, 9]}, 0.5)):
        cfg = synth.MODEL_CONFIGS[cname]
        torch.manual_seed(0)
        m = cls(**cfg, **extra)
        synth.load_synth_weights(m, dict(cfg, **extra) if cls is ResVT else cfg, "residualvit" if cls is ResVT else "vit", seed=0)
        if budget is not None and cls is not ResVT:
            m.set_budget(budget)
        m.train()
        torch.manual_seed(7)        # ResidualViT draws one budget per sample with torch.rand(n) in training (residualvit.py:541-549,565-567)
        x = torch.from_numpy(synth.synth_images(batch, cfg["image_size"], seed=0))
        y = torch.arange(batch) % cfg["num_classes"]
        opt = torch.optim.Adam(m.parameters(), lr=1e-3)
        opt.zero_grad()
        logits = m(x)
        loss = torch.nn.CrossEntropyLoss()(logits, y)
        loss.backward()
        names = [n for n, _ in m.named_parameters()]
        gn = np.array([float(p.grad.norm()) for _, p in m.named_parameters()])
        total = float(torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0))
        for n, p in m.named_parameters():
            if n in (FULL_GRADS_BIG if cname == "vit_b_16" else FULL_GRADS):
                out[f"{name}/grad/{n}"] = p.grad.detach().numpy().copy()      # AFTER the clip (scaled by 1/total when total > 1)
        opt.step()
        out[f"{name}/logits"] = logits.detach().numpy()
        out[f"{name}/loss"] = np.array(float(loss))
        out[f"{name}/grad_norms"] = gn                                        # before the clip, named_parameters() order
        out[f"{name}/total_norm"] = np.array(total)
        out[f"{name}/post_step_sum"] = np.array([float(p.detach().double().sum()) for _, p in m.named_parameters()])
        out[f"{name}/post_step_abs"] = np.array([float(p.detach().double().abs().sum()) for _, p in m.named_parameters()])
        out[f"{name}/names"] = np.array(names)
        print(f"{name}: loss {float(loss):.6f} total grad norm {total:.5f} params {len(names)}")
    np.savez_compressed(os.path.join(GOLD, "train_step.npz"), **out)


if __name__ == "__main__":
    main()
