"""Launch-bound regime (BASELINE config 1: vit_tiny 160x160, batch 32): eager ctypes launches vs one hipGraph replay
(peekvit_amd.graph.GraphedForward).  Usage: python scripts/bench_small_batch.py [model] [batch]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from peekvit_amd import synth
from peekvit_amd.graph import GraphedForward
from peekvit_amd.models.vit import VisionTransformer
name = sys.argv[1] if len(sys.argv) > 1 else "vit_tiny"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
cfg = synth.MODEL_CONFIGS[name]
m = VisionTransformer(**cfg); synth.load_synth_weights(m, cfg); m = m.eval().cuda()
x = torch.randn(B, 3, cfg["image_size"], cfg["image_size"], device="cuda")
def timeit(f, n=200):
    for _ in range(20): f(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f(x)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
with torch.no_grad():
    te = timeit(m)
    g = GraphedForward(m, x)
    tg = timeit(g)
    assert torch.equal(g(x), m(x))
print(f"{name} batch {B}: eager {te*1e3:.3f} ms ({B/te:.0f} img/s)   hipGraph replay {tg*1e3:.3f} ms ({B/tg:.0f} img/s), bit-identical")
