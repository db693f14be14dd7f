"""LayerNorm folding (opt-in PEEKVIT_AMD_FOLD_LN=1): accuracy vs the reference's golden logits and throughput, per operand type.
Usage: python scripts/fold_ln_report.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from peekvit_amd import synth, engine
from peekvit_amd.models.vit import VisionTransformer
gold = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
rel = lambda a, b: float(np.linalg.norm(a.astype(np.float64) - b) / np.linalg.norm(b.astype(np.float64)))
cfg = synth.MODEL_CONFIGS["vit_b_16"]
m = VisionTransformer(**cfg); synth.load_synth_weights(m, cfg); m = m.eval().cuda()
x2 = torch.from_numpy(synth.synth_images(2, cfg["image_size"])).cuda()
xb = torch.cat([x2, torch.randn(14, 3, 224, 224, device="cuda")])             # 16 images = 3152 rows: enough for the 256-row tile path
g = np.load(os.path.join(gold, "vit_b_16.npz"))["logits"]
x = torch.randn(2048, 3, 224, 224, device="cuda")
for mode in ("bf16", "f16"):
    for fold in (False, True):
        engine._FOLD_LN = fold
        with torch.no_grad(), engine.precision(mode):
            err = rel(m(xb)[:2].cpu().numpy(), g)
            for _ in range(2): m(x)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(5): m(x)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        print(f"{mode:5s} fold_ln={int(fold)}: logits rel-L2 vs reference {err:.2e}   {dt*1e3:.1f} ms/forward  {2048/dt:.0f} img/s")
