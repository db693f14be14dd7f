"""Diagnostic (round 6): where the hostile-weights ViT-B/16 loses its precision on noise images - per-image and total error against bf16x3 of
(a) all layers' attention half hybrid, (b) + all MLP halves split, and a per-layer bisection of what is left."""
import os, sys, warnings
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from peekvit_amd import engine, synth
from peekvit_amd.models.vit import VisionTransformer
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
cfg = synth.MODEL_CONFIGS["vit_b_16"]
m = VisionTransformer(**cfg)
m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.hostile_variants(cfg)["hostile"].items()})
m = m.eval().cuda()
x = torch.randn(B, 3, 224, 224, generator=torch.Generator(device="cuda").manual_seed(4321), device="cuda").to(torch.bfloat16).float()
warnings.simplefilter("ignore")
rel = lambda a, b: float((a - b).norm() / b.norm())
engine.SELFCHECK_IMAGES = 0
with torch.no_grad():
    with engine.precision("bf16x3"):
        ref = m(x).clone()
    engine.reset_guard(m)
    for _ in range(3):
        y = m(x)
    st = engine.guard_state(m)
    print("attention halves hybrid:", sorted(st.hybrid), "error", rel(y, ref), "per image max", float(((y - ref).norm(dim=1) / ref.norm(dim=1)).max()))
    st.mlp_hybrid = True
    y2 = m(x)
    print("+ MLP halves split:        error", rel(y2, ref), "per image max", float(((y2 - ref).norm(dim=1) / ref.norm(dim=1)).max()))
    # what is left: out-projection on fp16 operands, probabilities and P.V in 16 bits, q|k|v (fp32 in the hybrid layers).  Block by block: the tokens after
    # each block against the bf16x3 run's
    feats = {}
    def hook(tag):
        def f(mod, i, o):
            feats.setdefault(tag, []).append(o.detach().float().clone())
        return f
    hs = [blk.register_forward_hook(hook(i)) for i, blk in enumerate(m.encoder.layers)]
    with engine.precision("bf16x3"):
        m(x)
    y3 = m(x)
    for h in hs: h.remove()
    for i in range(len(m.encoder.layers)):
        a, b = feats[i][1], feats[i][0]
        if a.shape == b.shape:
            print("after block", i, "tokens rel L2 vs bf16x3", rel(a, b), " class rows", rel(a[:, 0], b[:, 0]))
