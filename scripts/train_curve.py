"""Loss curves of the HIP training path and the stock-op fp32 composite on the same data, same init, same optimizer (Adam 1e-3, clip 1.0):
a learnable synthetic task (labels = argmax of a fixed random linear map of the mean patch colour).  python scripts/train_curve.py [model] [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from peekvit_amd import synth
from peekvit_amd.models.vit import VisionTransformer
name = sys.argv[1] if len(sys.argv) > 1 else "vit_tiny"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
cfg = synth.MODEL_CONFIGS[name]
g = torch.Generator().manual_seed(0)
N, B = 512, 64
x = torch.randn(N, 3, cfg["image_size"], cfg["image_size"], generator=g)
teacher = torch.randn(3, cfg["num_classes"], generator=g)
x = x + 2.0 * torch.randn(N, 3, 1, 1, generator=g)                       # a per-image colour cast the label depends on
y = (x.mean(dim=(2, 3)) @ teacher).argmax(1)
curves = {}
for mode in ("hip", "torch"):
    os.environ["PEEKVIT_AMD_TRAIN"] = mode
    torch.manual_seed(0)
    m = VisionTransformer(**cfg); synth.load_synth_weights(m, cfg); m = m.cuda().train()
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    losses = []
    for s in range(steps):
        idx = torch.arange(s * B, (s + 1) * B) % N
        xb, yb = x[idx].cuda(), y[idx].cuda()
        opt.zero_grad()
        loss = torch.nn.functional.cross_entropy(m(xb), yb)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
        opt.step()
        losses.append(float(loss))
    curves[mode] = losses
    print(mode, " ".join(f"{v:.3f}" for v in losses[::max(1, steps // 12)]), "final", f"{sum(losses[-5:]) / 5:.4f}")
d = max(abs(a - b) for a, b in zip(curves["hip"], curves["torch"]))
print(f"max |loss_hip - loss_torch| over {steps} steps: {d:.4f}; first-5 mean {sum(curves['hip'][:5])/5:.3f} -> last-5 mean {sum(curves['hip'][-5:])/5:.3f}")
