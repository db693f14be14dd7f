"""Round 4 (CPU only, the oracle): where the golden ResidualViT toy's fp16 error comes from.  (1) per model / budget: logits error of the oracle's
"f16" mode (the HIP path's rounding points) against fp32, and how close the gate masks sit to zero; (2) for vit_micro @ 0.2 / 0.5: the error with ONE
rounding site enabled at a time.  Finding: masks are 0.05 .. 0.44 (nothing near the threshold); thirteen activation-rounding sites of 1 - 4.6e-4 each add
up to 1.2e-3 on a 2-layer, 18-token, width-128 model.   python scripts/residual_toy_ablation.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import vit_oracle as O
from peekvit_amd import synth
torch.set_num_threads(8)
rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
for name, gb in (("vit_micro", 10), ("vit_micro", 0), ("vit_tiny", 10)):
    cfg = dict(synth.MODEL_CONFIGS[name])
    c = dict(cfg, gate_type="sigmoid", gate_temp=1, gate_bias=gb, add_budget_token="learnable", gate_threshold=0.5)
    sd = synth.synth_state_dict(c, "residualvit")
    x = torch.from_numpy(synth.synth_images(2, cfg["image_size"]))
    for b in (0.2, 0.5, 1.0):
        t32, t16 = {}, {}
        l32 = O.residualvit_forward(x, sd, c, b, "fp32", trace=t32).numpy()
        l16 = O.residualvit_forward(x, sd, c, b, "f16", trace=t16).numpy()
        m32, m16 = torch.stack(t32["masks"]).numpy(), torch.stack(t16["masks"]).numpy()
        print(f"{name} gate_bias {gb} budget {b}: f16-mode logits error {rel(l16, l32):.2e}  mask max diff {np.abs(m32 - m16).max():.1e}  masks: {m32.size} values, "
              f"{int((m32 == 0).sum())} zero, {int(((m32 > 0) & (m32 < 1e-2)).sum())} in (0, 1e-2), max {m32.max():.3f}")
name, gb = "vit_micro", 10
cfg = dict(synth.MODEL_CONFIGS[name])
c = dict(cfg, gate_type="sigmoid", gate_temp=1, gate_bias=gb, add_budget_token="learnable", gate_threshold=0.5)
sd = synth.synth_state_dict(c, "residualvit")
x = torch.from_numpy(synth.synth_images(2, cfg["image_size"]))
orig = O.rb
sites = ["patch x", "patch w"] + [f"block {i} {s}" for i in range(cfg["num_layers"]) for s in
                                  ("qkv in", "qkv w", "q", "k", "v", "P", "out-proj in", "out-proj w", "fc1 in", "fc1 w", "fc2 in", "fc2 w")]
for b in (0.2, 0.5):
    l32 = O.residualvit_forward(x, sd, c, b, "fp32").numpy()
    print(f"budget {b}: all sites rounded {rel(O.residualvit_forward(x, sd, c, b, 'f16').numpy(), l32):.2e}; one site at a time:")
    for only in range(len(sites)):
        k = [0]
        def sel(t, mode, only=only):
            i = k[0]; k[0] += 1
            return orig(t, mode) if i == only else t
        O.rb = sel
        e = rel(O.residualvit_forward(x, sd, c, b, "f16").numpy(), l32)
        O.rb = orig
        if e > 5e-6:
            print(f"    {sites[only]:22s} {e:.2e}")
