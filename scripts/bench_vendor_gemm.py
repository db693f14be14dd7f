"""Calibration only (not part of the product path): the vendor GEMM (torch.nn.functional.linear -> hipBLASLt/rocBLAS, bf16, no
epilogue beyond the bias) on the four ViT-B/16 token GEMM shapes, to place pv_gemm_bf16's TF/s next to the best library number
for the same shape on the same (power-capped) box.  Usage: python scripts/bench_vendor_gemm.py [iters]"""
import sys, os
import torch
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
M = int(os.environ.get("M", 403456))
for name, N, K in [("qkv", 2304, 768), ("out", 768, 768), ("fc1", 3072, 768), ("fc2", 768, 3072)]:
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda").to(torch.bfloat16)
    for _ in range(3):
        torch.nn.functional.linear(a, w, b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        torch.nn.functional.linear(a, w, b)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"vendor {name:4s} N={N:5d} K={K:5d}: {ms:7.3f} ms  {2.0 * M * N * K / ms / 1e9:8.1f} TF/s (bf16 out, bias only)")
