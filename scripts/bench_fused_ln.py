import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from peekvit_amd import ops
from peekvit_amd._lib import PV_EPI_BIAS_RES_F32
dev = "cuda:0"; M = 403456
g = torch.Generator(device=dev).manual_seed(0)
def timeit(f, n=10):
    for _ in range(2): f()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
for name, N, K in (("out", 768, 768), ("fc2", 768, 3072)):
    a = torch.randn(M, K, generator=g, device=dev).bfloat16(); w = (torch.randn(N, K, generator=g, device=dev) * K ** -0.5).bfloat16()
    bias = torch.randn(N, generator=g, device=dev); res = torch.randn(M, N, generator=g, device=dev)
    gam = torch.randn(N, generator=g, device=dev); bet = torch.randn(N, generator=g, device=dev)
    out = torch.empty(M, N, device=dev); h = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    t0 = timeit(lambda: ops.gemm(a, w, bias, out, PV_EPI_BIAS_RES_F32, res=res))
    t1 = timeit(lambda: ops.layernorm_bf16(out, gam, bet, 1e-5, h))
    t2 = timeit(lambda: ops.gemm(a, w, bias, out, PV_EPI_BIAS_RES_F32, res=res, ln=(gam, bet, 1e-5, h, None)))
    print(f"{name}: gemm {t0:.3f} ms + LN {t1:.3f} ms = {t0+t1:.3f}   fused {t2:.3f} ms")
