"""Per-shape GEMM microbenchmark (HIP events, random data). Usage: python scripts/bench_gemm.py [iters]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from peekvit_amd import ops
from peekvit_amd._lib import PV_EPI_BIAS_BF16, PV_EPI_BIAS_GELU_BF16, PV_EPI_BIAS_RES_F32

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
M = int(os.environ.get("M", 403456))
dev = "cuda:0"
shapes = [("qkv", 2304, 768, PV_EPI_BIAS_BF16), ("out", 768, 768, PV_EPI_BIAS_RES_F32), ("fc1", 3072, 768, PV_EPI_BIAS_GELU_BF16),
          ("fc2", 768, 3072, PV_EPI_BIAS_RES_F32), ("fc1-nogelu", 3072, 768, PV_EPI_BIAS_BF16)]
g = torch.Generator(device=dev).manual_seed(0)
tot_ms = 0.0; tot_fl = 0.0
for name, N, K, epi in shapes:
    a = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g, device=dev) * K ** -0.5).to(torch.bfloat16)
    bias = torch.randn(N, generator=g, device=dev)
    out = torch.empty((M, N), dtype=torch.float32 if epi == PV_EPI_BIAS_RES_F32 else torch.bfloat16, device=dev)
    res = torch.randn(M, N, generator=g, device=dev) if epi == PV_EPI_BIAS_RES_F32 else None
    for _ in range(3):
        ops.gemm(a, w, bias, out, epi, res=res)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.gemm(a, w, bias, out, epi, res=res)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    fl = 2.0 * M * N * K
    if name != "fc1-nogelu":
        tot_ms += ms; tot_fl += fl
    print(f"{name:11s} N={N:5d} K={K:5d}: {ms:7.3f} ms  {fl / ms / 1e9:8.1f} TF/s")
    del a, w, out, res
print(f"layer total {tot_ms:.3f} ms  {tot_fl / tot_ms / 1e9:.1f} TF/s  (x12 = {12 * tot_ms:.1f} ms)")
