"""The full-row GEMM launches of vit_small at batch 512, alone (for rocprofv3 --pmc passes): python3 scripts/fullrow_only.py [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from peekvit_amd import engine, ops
from peekvit_amd._lib import PV_EPI_BIAS_RES_F32
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = "cuda:0"
M, N = 512 * 197, 384
g = torch.Generator(device=dev).manual_seed(0)
with engine.precision("f16"):
    for K in (384, 1536):
        a = torch.randn(M, K, generator=g, device=dev).to(torch.float16)
        w = (torch.randn(N, K, generator=g, device=dev) * K ** -0.5).to(torch.float16)
        bias, res = torch.randn(N, generator=g, device=dev), torch.randn(M, N, generator=g, device=dev)
        gam, bet = torch.rand(N, generator=g, device=dev) + 0.5, torch.randn(N, generator=g, device=dev) * 0.1
        out, h = torch.empty(M, N, device=dev), torch.empty(M, N, dtype=torch.float16, device=dev)
        for _ in range(iters):
            ops.gemm(a, w, bias, out, PV_EPI_BIAS_RES_F32, res=res, ln=(gam, bet, 1e-5, h, None))
        torch.cuda.synchronize()
print("done")
