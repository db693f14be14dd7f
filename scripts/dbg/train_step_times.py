import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from peekvit_amd import synth, train_engine
from peekvit_amd.models.vit import VisionTransformer
cfg = synth.MODEL_CONFIGS["vit_small"]; B = 512
m = VisionTransformer(**cfg); synth.load_synth_weights(m, cfg); m = m.cuda().train()
x = torch.randn(B, 3, 224, 224, device="cuda"); y = torch.randint(0, 1000, (B,), device="cuda")
params = list(m.parameters()); opt = torch.optim.Adam(params, lr=1e-3, fused=True)
def step():
    for p in params: p.grad = None
    torch.nn.functional.cross_entropy(m(x), y).backward()
    torch.nn.utils.clip_grad_norm_(params, 1.0, foreach=True)
    opt.step()
import gc
if len(sys.argv) > 1 and sys.argv[1] == "nogc": gc.disable()
if len(sys.argv) > 1 and sys.argv[1] == "bf16": train_engine._TRAIN_OPERAND = "bf16"
ts = []; segs = []
for i in range(40):
    torch.cuda.synchronize(); t0 = time.perf_counter(); step(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    ms = torch.cuda.memory_stats(); segs.append((ms["segment.all.allocated"], ms["num_alloc_retries"], round(ms["reserved_bytes.all.current"] / 1e9, 2), gc.get_count()))
print("segments/retries/reserved/gc:", segs[5:16])
print("per-step ms (synchronised each step):", [round(t, 1) for t in ts])
st = train_engine.train_state(m)
print("scale", st.scale, "amax", st.amax, "steps", st.steps, "skipped", st.skipped)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): step()
torch.cuda.synchronize(); print("20 free-running steps ms/step", (time.perf_counter() - t0) / 20 * 1e3)
