"""ResidualViT-B/16 (sigmoid gates, learnable budget token, configs/model/residualvit_b_16.yaml settings): inference img/s at batch 2048 and
the training step at batch 512/1024 on the HIP path, with the per-kernel table.  python scripts/bench_residual.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from peekvit_amd import synth, ops
from peekvit_amd.models.residualvit import ResidualVisionTransformer
cfg = synth.MODEL_CONFIGS["vit_b_16"]
extra = dict(residual_layers=["attention+mlp"] * cfg["num_layers"], gate_temp=1, add_input=False, gate_type="sigmoid", gate_threshold=0.5,
             gate_bias=10, add_budget_token="learnable")
m = ResidualVisionTransformer(**cfg, **extra)
synth.load_synth_weights(m, dict(cfg, **extra), "residualvit", seed=0)
m = m.cuda().eval()
m.set_budget(0.5)
out = {}
x = torch.randn(2048, 3, 224, 224, device="cuda")
with torch.no_grad():
    for _ in range(3): m(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): m(x)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    with ops.KernelTimer() as kt:
        m(x)
    torch.cuda.synchronize()
out["inference B=2048"] = {"img_per_s": round(2048 / dt, 1), "ms": round(dt * 1e3, 2),
                           "kernels_ms": {k: round(v["ms"], 3) for k, v in sorted(kt.summary().items(), key=lambda kv: -kv[1]["ms"])}}
m.train()
for B in (512, 1024):
    xb, y = x[:B], torch.randint(0, 1000, (B,), device="cuda")
    def step():
        for p in m.parameters(): p.grad = None
        torch.nn.functional.cross_entropy(m(xb), y).backward()
    for _ in range(2): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    with ops.KernelTimer() as kt:
        step()
    torch.cuda.synchronize()
    pv = sum(v["ms"] for v in kt.summary().values())
    out[f"train step B={B}"] = {"img_per_s": round(B / dt, 1), "ms": round(dt * 1e3, 2), "ms_in_pv_kernels": round(pv, 2),
                                "kernels_ms": {k: round(v["ms"], 3) for k, v in sorted(kt.summary().items(), key=lambda kv: -kv[1]["ms"])}}
# the reference's finetuning (train/train.py:100): only parameters named gate / class / head / threshold / budget are trained
for n, p in m.named_parameters():
    p.requires_grad_(any(k in n for k in ("gate", "class", "head", "threshold", "budget")))
B = 1024
xb, y = x[:B], torch.randint(0, 1000, (B,), device="cuda")
def step():
    for p in m.parameters(): p.grad = None
    torch.nn.functional.cross_entropy(m(xb), y).backward()
for _ in range(2): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3): step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
with ops.KernelTimer() as kt:
    step()
torch.cuda.synchronize()
out[f"finetune step (gates / class tokens / head only) B={B}"] = {"img_per_s": round(B / dt, 1), "ms": round(dt * 1e3, 2),
    "kernels_ms": {k: round(v["ms"], 3) for k, v in sorted(kt.summary().items(), key=lambda kv: -kv[1]["ms"])}}
print(json.dumps(out, indent=1))
