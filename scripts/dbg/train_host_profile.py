"""Host-side profile of the training step of a small model (is the step host-bound, and by what?)."""
import cProfile, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from peekvit_amd import synth, train_engine
from peekvit_amd.models.vit import VisionTransformer
name, B = (sys.argv[1] if len(sys.argv) > 1 else "vit_small"), int(sys.argv[2]) if len(sys.argv) > 2 else 512
cfg = synth.MODEL_CONFIGS[name]
m = VisionTransformer(**cfg); synth.load_synth_weights(m, cfg); m = m.cuda().train()
x = torch.randn(B, 3, cfg["image_size"], cfg["image_size"], device="cuda"); y = torch.randint(0, cfg["num_classes"], (B,), device="cuda")
params = list(m.parameters()); opt = torch.optim.Adam(params, lr=1e-3, fused=True)
def step():
    for p in params: p.grad = None
    torch.nn.functional.cross_entropy(m(x), y).backward()
    torch.nn.utils.clip_grad_norm_(params, 1.0, foreach=True)
    opt.step()
for operand in ("bf16", "f16", "bf16", "f16", "f16"):
    train_engine._TRAIN_OPERAND = operand
    for _ in range(5): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): step()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize(); t = time.perf_counter() - t0
    print(operand, "ms/step", t / 20 * 1e3, "host-only ms/step", t_host / 20 * 1e3)
train_engine._TRAIN_OPERAND = "f16"
pr = cProfile.Profile(); pr.enable()
for _ in range(10): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
# where the f16 backward's extra host time goes: wall time inside the loss-scale machinery's pieces, per step
import collections
acc = collections.Counter()
def wrap(obj, name):
    f = getattr(obj, name)
    def g(*a, **k):
        t = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            acc[name] += time.perf_counter() - t
    setattr(obj, name, g)
for n in ("_finish", "resolve", "begin_backward", "note"):
    wrap(train_engine.TrainPass, n)
wrap(train_engine, "_unscale"); wrap(train_engine, "_optimizer_pre_hook")
for _ in range(3): step()
acc.clear(); torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): step()
torch.cuda.synchronize(); t = time.perf_counter() - t0
print("f16 instrumented ms/step", t / 20 * 1e3, {k: round(v / 20 * 1e3, 3) for k, v in acc.items()})
