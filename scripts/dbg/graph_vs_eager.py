"""ViT-B/16 at batch 2048: eager forward vs one hipGraph replay (how much of the step is launch gaps?)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from peekvit_amd import synth
from peekvit_amd.graph import GraphedForward
from peekvit_amd.models.vit import VisionTransformer
name = sys.argv[1] if len(sys.argv) > 1 else "vit_b_16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
cfg = synth.MODEL_CONFIGS[name]
m = VisionTransformer(**cfg); synth.load_synth_weights(m, cfg); m = m.eval().to("cuda:0")
x = torch.randn(B, 3, cfg["image_size"], cfg["image_size"], device="cuda:0")
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
with torch.no_grad():
    e = [t(lambda: m(x)) for _ in range(2)]
    g = GraphedForward(m, x)
    r = [t(lambda: g(x)) for _ in range(2)]
    e2 = t(lambda: m(x))
print(f"{name} batch {B}: eager {e[0]:.3f} {e[1]:.3f} {e2:.3f} ms   graph replay {r[0]:.3f} {r[1]:.3f} ms")
