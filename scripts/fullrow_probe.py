"""Round 4: where the full-row GEMM (128 x N tiles, narrow hidden dims) spends a tile - its launch time against K (K = 64: prologue + epilogue
only) with and without the fused LayerNorm, at vit_small's batch-512 row count.  python scripts/fullrow_probe.py"""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from peekvit_amd import ops
from peekvit_amd._lib import PV_EPI_BIAS_RES_F32
dev = "cuda:0"
M, N = 512 * 197, 384
g = torch.Generator(device=dev).manual_seed(0)
res = torch.randn(M, N, generator=g, device=dev)
gam, bet = torch.rand(N, generator=g, device=dev) + 0.5, torch.randn(N, generator=g, device=dev) * 0.1
bias = torch.randn(N, generator=g, device=dev) * 0.1
for K in (64, 128, 384, 768, 1536):
    a = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g, device=dev) * K ** -0.5).to(torch.bfloat16)
    for with_ln in (False, True):
        out = torch.empty((M, N), dtype=torch.float32, device=dev)
        h = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
        ln = (gam, bet, 1e-5, h, None) if with_ln else None
        for _ in range(3):
            ops.gemm(a, w, bias, out, PV_EPI_BIAS_RES_F32, res=res, ln=ln)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.gemm(a, w, bias, out, PV_EPI_BIAS_RES_F32, res=res, ln=ln)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        nbytes = 2.0 * M * K + 2.0 * N * K + 8.0 * M * N + (2.0 * M * N if with_ln else 0)
        print(f"K={K:5d} ln={int(with_ln)}: {us:7.1f} us   {2.0 * M * N * K / us / 1e6:7.1f} TF/s   algorithmic {nbytes / us / 1e3:6.0f} GB/s", flush=True)
