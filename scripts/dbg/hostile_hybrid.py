"""Diagnostics: what mode auto does on the hostile-weights ViT-B/16 fixture at a given batch (local fallback, self-check verdicts, flag words)."""
import os, sys, warnings
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from peekvit_amd import engine, synth
from peekvit_amd.models.vit import VisionTransformer
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cfg = synth.MODEL_CONFIGS["vit_b_16"]
m = VisionTransformer(**cfg)
m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.hostile_variants(cfg)[sys.argv[2] if len(sys.argv) > 2 else "hostile"].items()})
m = m.eval().cuda()
x = torch.randn(B, 3, 224, 224, generator=torch.Generator(device="cuda").manual_seed(4321), device="cuda").to(torch.bfloat16).float()
warnings.simplefilter("always")
with torch.no_grad():
    for i in range(3):
        f0, h0 = engine.fallback_count, engine.hybrid_fallback_count
        y = m(x)
        st = engine.guard_state(m)
        print(i, "fallbacks", engine.fallback_count - f0, "local", engine.hybrid_fallback_count - h0, "hybrid", sorted(st.hybrid), "no_fold", st.no_fold, "unsafe", st.unsafe,
              "verdicts", {k[1:3] + (len(k[5]),): v for k, v in st.verdicts.items()}, "selfcheck_last", engine.selfcheck_last, "flag words", engine.range_flag_for(x.device).tolist()[:14])
    with engine.precision("bf16x3"):
        ref = m(x)
    print("auto vs bf16x3:", float((y - ref).norm() / ref.norm()))
    # the hybrid forward alone, without the self-check
    engine.SELFCHECK_IMAGES = 0
    engine.reset_guard(m)
    for i in range(2):
        f0 = engine.fallback_count
        y = m(x)
        print("no self-check", i, "fallbacks", engine.fallback_count - f0, "hybrid", sorted(engine.guard_state(m).hybrid), "err vs bf16x3", float((y - ref).norm() / ref.norm()),
              "first 8 images", float((y[:8] - ref[:8]).norm() / ref[:8].norm()))
    import time
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        m(x)
    torch.cuda.synchronize(); print("hybrid forward img/s", B * 10 / (time.perf_counter() - t0))
