"""Host-side profile of a small-batch eager forward (where does the launch path spend its time?): python scripts/dbg/prof_small.py [model] [batch]"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from peekvit_amd import synth
from peekvit_amd.models.vit import VisionTransformer
name = sys.argv[1] if len(sys.argv) > 1 else "vit_tiny"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
cfg = synth.MODEL_CONFIGS[name]
model = VisionTransformer(**cfg)
synth.load_synth_weights(model, cfg)
model = model.eval().to("cuda:0")
x = torch.randn(B, 3, cfg["image_size"], cfg["image_size"], device="cuda:0")
with torch.no_grad():
    for _ in range(5):
        model(x)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(50):
        model(x)
    torch.cuda.synchronize()
    print(f"{name} batch {B}: {(time.perf_counter() - t) / 50 * 1e3:.3f} ms per forward")
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(50):
        model(x)
    torch.cuda.synchronize()
    pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
