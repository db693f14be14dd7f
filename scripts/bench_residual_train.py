"""One-off: ResidualViT-B/16 (sigmoid gates, learnable budget token) fwd+bwd step time on the HIP masked-block path vs the stock-op
composite.  Usage: python scripts/bench_residual_train.py [batch]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from peekvit_amd import synth, ops
from peekvit_amd.models.residualvit import ResidualVisionTransformer
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
cfg = synth.MODEL_CONFIGS["vit_b_16"]
extra = dict(residual_layers=["attention+mlp"] * cfg["num_layers"], gate_temp=1, add_input=False, gate_type="sigmoid", gate_threshold=0.5,
             gate_bias=10, add_budget_token="learnable")
m = ResidualVisionTransformer(**cfg, **extra)
synth.load_synth_weights(m, dict(cfg, **extra), "residualvit", seed=0)
m = m.cuda().train()
x = torch.randn(B, 3, 224, 224, device="cuda"); y = torch.randint(0, 1000, (B,), device="cuda")
def step():
    for p in m.parameters(): p.grad = None
    torch.nn.functional.cross_entropy(m(x), y).backward()
for mode in ("hip", "torch"):
    os.environ["PEEKVIT_AMD_TRAIN"] = mode
    for _ in range(2): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    assert all(torch.isfinite(p.grad).all() for p in m.parameters())
    print(f"ResidualViT-B/16 fwd+bwd batch {B} [{mode}]: {dt*1e3:.1f} ms/step  {B/dt:.0f} img/s")
