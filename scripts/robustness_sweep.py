"""Crash / consistency sweep over odd batch sizes (ragged GEMM tiles, TN tail rows, small-M kernel choice): inference logits of
image 0 must not depend on the batch it travels in; a training step must produce finite gradients.  Usage: python scripts/robustness_sweep.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from peekvit_amd import synth
from peekvit_amd.models.vit import VisionTransformer
for name, batches in (("vit_small", (1, 7, 11, 100, 333, 1000)), ("vit_b_16", (1, 3, 10, 65, 130))):
    cfg = synth.MODEL_CONFIGS[name]
    m = VisionTransformer(**cfg); synth.load_synth_weights(m, cfg); m = m.cuda()
    x = torch.randn(max(batches), 3, cfg["image_size"], cfg["image_size"], device="cuda")
    ref = None
    for B in batches:
        with torch.no_grad():
            y = m.eval()(x[:B])
        assert torch.isfinite(y).all()
        ref = y[0] if ref is None else ref
        same = torch.equal(y[0], ref)
        m.train()
        for p in m.parameters(): p.grad = None
        torch.nn.functional.cross_entropy(m(x[:B]), torch.arange(B, device="cuda") % cfg["num_classes"]).backward()
        ok = all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())
        print(f"{name} B={B:5d}: logits[0] batch-invariant {same}   training grads finite {ok}")
        assert same and ok
print("sweep ok")
