"""Board power and sclk while ONE kernel of the forward runs back to back (1.5 s each), ViT-B/16 shapes at batch 2048:
where the step's energy goes.  Writes gpurun_out/power_by_kernel.json.  Read-only sysfs sampling (scripts/raster_ab.py PowerSampler)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import torch
from raster_ab import PowerSampler
from peekvit_amd import ops
from peekvit_amd._lib import PV_EPI_BIAS_BF16, PV_EPI_BIAS_GELU_BF16, PV_EPI_BIAS_RES_F32

dev = torch.device("cuda:0")
pr = torch.cuda.get_device_properties(0)
pci = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0" if hasattr(pr, "pci_bus_id") else None
sm = PowerSampler(pci)
if not sm.cards:
    sm = PowerSampler(None)
sm.start()
B, S, D, H, Mh = 2048, 197, 768, 12, 3072
R = B * S
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(R, D, generator=g, device=dev)
gamma, beta = torch.ones(D, device=dev), torch.zeros(D, device=dev)
h = torch.empty(R, D, dtype=torch.bfloat16, device=dev)
qkv = (torch.randn(R, 3 * D, generator=g, device=dev) * 0.5).to(torch.bfloat16)
att = torch.empty(R, D, dtype=torch.bfloat16, device=dev)
a768 = torch.randn(R, D, generator=g, device=dev).to(torch.bfloat16)
a3072 = torch.randn(R, Mh, generator=g, device=dev).to(torch.bfloat16)
w = {n: (torch.randn(n[0], n[1], generator=g, device=dev) * n[1] ** -0.5).to(torch.bfloat16) for n in ((3 * D, D), (D, D), (Mh, D), (D, Mh))}
bias = {n: torch.randn(n, generator=g, device=dev) for n in (3 * D, D, Mh)}
o16 = torch.empty(R, Mh, dtype=torch.bfloat16, device=dev)
o_qkv = torch.empty(R, 3 * D, dtype=torch.bfloat16, device=dev)
o32, res = torch.empty(R, D, device=dev), torch.randn(R, D, generator=g, device=dev)
copy_src, copy_dst = torch.empty(1 << 28, device=dev), torch.empty(1 << 28, device=dev)      # 1 GiB fp32 each
cases = {
    "idle": (lambda: time.sleep(0.002), 0, 0),
    "hbm_copy_2GiB": (lambda: copy_dst.copy_(copy_src), 0.0, 2.0 * copy_src.numel() * 4),
    "layernorm": (lambda: ops.layernorm_bf16(x, gamma, beta, 1e-5, h), 0.0, 6.0 * R * D),
    "attention": (lambda: ops.attention(qkv, att, B, S, H, D // H), 4.0 * B * H * S * S * (D // H), 8.0 * R * D),
    "gemm_qkv": (lambda: ops.gemm(a768, w[(3 * D, D)], bias[3 * D], o_qkv, PV_EPI_BIAS_BF16), 2.0 * R * 3 * D * D, 0),
    "gemm_out_res": (lambda: ops.gemm(a768, w[(D, D)], bias[D], o32, PV_EPI_BIAS_RES_F32, res=res), 2.0 * R * D * D, 0),
    "gemm_fc1_gelu": (lambda: ops.gemm(a768, w[(Mh, D)], bias[Mh], o16, PV_EPI_BIAS_GELU_BF16), 2.0 * R * Mh * D, 0),
    "gemm_fc2_res": (lambda: ops.gemm(a3072, w[(D, Mh)], bias[D], o32, PV_EPI_BIAS_RES_F32, res=res), 2.0 * R * Mh * D, 0),
}
out = {}
for name, (fn, flops, nbytes) in cases.items():
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    n = max(4, int(1500.0 / max(e0.elapsed_time(e1), 0.05)))
    t0 = time.perf_counter()
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    t1 = time.perf_counter()
    ms = e0.elapsed_time(e1) / n
    r = {"ms": round(ms, 4), **sm.mean(t0 + 0.35 * (t1 - t0), t1)}
    if flops: r["tflops"] = round(flops / ms / 1e9, 1)
    if nbytes: r["algorithmic_gbs"] = round(nbytes / ms / 1e6, 1)
    if "power_w" in r: r["joule_per_launch"] = round(r["power_w"] * ms * 1e-3, 3)
    out[name] = r
    print(name, r, flush=True)
sm.stop = True
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "power_by_kernel.json"), "w"), indent=1)
