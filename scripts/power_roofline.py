"""Sweep scripts/bin/power_roofline (the dependency-free MFMA + LDS + L2 + HBM mix, see scripts/power_roofline.hip) and attach board power and
sclk to every point: the EMPIRICAL MFMA roofline of this MI355X under its 1400 W cap as a function of bytes per FLOP.  The real token GEMMs
(ViT-B/16, batch 2048) are measured in the same process afterwards, so each can be placed against the synthetic point with its traffic.
Writes gpurun_out/r02_power_roofline.json.    Build first (here or on the box):
    hipcc --offload-arch=gfx950 -O3 -w scripts/power_roofline.hip -o scripts/bin/power_roofline"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import torch
from raster_ab import PowerSampler

pr = torch.cuda.get_device_properties(0)
pci = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0" if hasattr(pr, "pci_bus_id") else None
sm = PowerSampler(pci)
if not sm.cards:
    sm = PowerSampler(None)
sm.start()
exe = os.path.join(ROOT, "scripts", "bin", "power_roofline")
src = os.path.join(ROOT, "scripts", "power_roofline.hip")
if not os.path.exists(exe) or os.path.getmtime(exe) < os.path.getmtime(src):
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-w", src, "-o", exe], check=True)
mixes = [(0, 0, 0), (6, 0, 0), (6, 2, 0), (6, 2, 2), (6, 2, 4), (6, 2, 6), (6, 2, 8), (6, 2, 10), (6, 2, 12), (6, 2, 16), (6, 1, 4), (6, 1, 8), (0, 0, 8), (0, 0, 16), (6, 0, 8), (3, 1, 8)]
out = {"synthetic": [], "note": "per wave and trip: 16 MFMA 16x16x32 bf16 + l x ds_read_b128 + c x 1 KiB L2-window loads; m x 1 KiB HBM-stream loads per 8 trips; "
                                "one 8-wave workgroup per CU; loads consumed one period after issue"}
for l, c, m in mixes:
    t0 = time.perf_counter()
    r = subprocess.run([exe, "-l", str(l), "-c", str(c), "-m", str(m), "-s", "2.5"], capture_output=True, text=True, timeout=120)
    t1 = time.perf_counter()
    line = [x for x in r.stdout.splitlines() if x.startswith("{")]
    if r.returncode != 0 or not line:
        print("failed:", l, c, m, r.stdout[-200:], r.stderr[-200:], flush=True)
        continue
    rec = json.loads(line[-1])
    rec.update(sm.mean(t0 + 0.5 * (t1 - t0), t1 - 0.1))          # the second half of the run: hipMalloc / fill are over, DVFS has settled
    out["synthetic"].append(rec)
    print(rec, flush=True)

# the real GEMMs, same box, same minute
from peekvit_amd import ops
from peekvit_amd._lib import PV_EPI_BIAS_BF16, PV_EPI_BIAS_GELU_BF16, PV_EPI_BIAS_RES_F32
dev = torch.device("cuda:0")
B, S, D, Mh = 2048, 197, 768, 3072
R = B * S
g = torch.Generator(device=dev).manual_seed(0)
a768 = torch.randn(R, D, generator=g, device=dev).to(torch.bfloat16)
a3072 = torch.randn(R, Mh, generator=g, device=dev).to(torch.bfloat16)
w = {n: (torch.randn(n[0], n[1], generator=g, device=dev) * n[1] ** -0.5).to(torch.bfloat16) for n in ((3 * D, D), (D, D), (Mh, D), (D, Mh))}
bias = {n: torch.randn(n, generator=g, device=dev) for n in (3 * D, D, Mh)}
o16 = torch.empty(R, Mh, dtype=torch.bfloat16, device=dev)
o_qkv = torch.empty(R, 3 * D, dtype=torch.bfloat16, device=dev)
o32, res = torch.empty(R, D, device=dev), torch.randn(R, D, generator=g, device=dev)
# algorithmic HBM bytes per launch (operands once + outputs once)
cases = {
    "gemm_qkv": (lambda: ops.gemm(a768, w[(3 * D, D)], bias[3 * D], o_qkv, PV_EPI_BIAS_BF16), 2.0 * R * 3 * D * D, 2.0 * R * D + 2.0 * R * 3 * D),
    "gemm_out_res": (lambda: ops.gemm(a768, w[(D, D)], bias[D], o32, PV_EPI_BIAS_RES_F32, res=res), 2.0 * R * D * D, 2.0 * R * D + 8.0 * R * D),
    "gemm_fc1_gelu": (lambda: ops.gemm(a768, w[(Mh, D)], bias[Mh], o16, PV_EPI_BIAS_GELU_BF16), 2.0 * R * Mh * D, 2.0 * R * D + 2.0 * R * Mh),
    "gemm_fc2_res": (lambda: ops.gemm(a3072, w[(D, Mh)], bias[D], o32, PV_EPI_BIAS_RES_F32, res=res), 2.0 * R * Mh * D, 2.0 * R * Mh + 8.0 * R * D),
}
out["gemms"] = {}
for name, (fn, flops, nbytes) in cases.items():
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    n = max(4, int(2000.0 / max(e0.elapsed_time(e1), 0.05)))
    t0 = time.perf_counter()
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    t1 = time.perf_counter()
    ms = e0.elapsed_time(e1) / n
    out["gemms"][name] = {"ms": round(ms, 4), "tflops": round(flops / ms / 1e9, 1), "algorithmic_hbm_GBps": round(nbytes / ms / 1e6, 1),
                          "algorithmic_bytes_per_kflop": round(nbytes / flops * 1e3, 3), **sm.mean(t0 + 0.35 * (t1 - t0), t1)}
    print(name, out["gemms"][name], flush=True)
sm.stop = True
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r02_power_roofline.json"), "w"), indent=1)
