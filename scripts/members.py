"""Per-member GEMM table of one model forward (HIP events on the launch stream, ops.KernelTimer.members): python scripts/members.py <model> <batch> [steps]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from peekvit_amd import ops, synth
from peekvit_amd.models.vit import VisionTransformer
name, batch = sys.argv[1], int(sys.argv[2]); steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
cfg = synth.MODEL_CONFIGS[name]
m = VisionTransformer(**cfg); synth.load_synth_weights(m, cfg); m = m.eval().cuda()
x = torch.randn(batch, 3, cfg["image_size"], cfg["image_size"], generator=torch.Generator(device="cuda").manual_seed(1), device="cuda").to(torch.bfloat16).float()
with torch.no_grad():
    for _ in range(5): m(x)
    torch.cuda.synchronize()
    with ops.KernelTimer() as kt:
        for _ in range(steps): m(x)
        torch.cuda.synchronize()
ks = kt.summary()
print(f"{name} batch {batch}: sum of kernel times {sum(v['ms'] for v in ks.values()) / steps:.3f} ms/step")
for k, v in sorted(ks.items(), key=lambda kv: -kv[1]["ms"]):
    print(f"  {k:28s} {v['launches'] // steps:3d} x {v['ms'] / v['launches'] * 1e3:8.1f} us = {v['ms'] / steps:7.3f} ms   {v['flops'] / v['ms'] / 1e9 if v['flops'] else 0:7.1f} TF/s  {v['bytes'] / v['ms'] / 1e6 if v['bytes'] else 0:7.0f} GB/s")
for (N, K, epi), v in sorted(kt.members("pv_gemm_bf16").items(), key=lambda kv: -kv[1]["ms"]):
    print(f"    gemm N={N:5d} K={K:5d} epi={epi}: {v['launches'] // steps:3d} x {v['ms'] / v['launches'] * 1e3:8.1f} us   {v['flops'] / v['ms'] / 1e9:7.1f} TF/s  {v['bytes'] / v['ms'] / 1e6:7.0f} GB/s (algorithmic)")
