"""Calibration (not part of the product path): the four ViT-B/16 token GEMMs at M = 403456 as
  (a) the vendor bf16 GEMM (torch.nn.functional.linear -> hipBLASLt, bias only, bf16 out)
  (b) (a) + the separate passes the reference's op sequence needs behind it: exact GELU (fc1), fp32 residual add (out-proj, fc2)
  (c) pv_gemm_bf16 with the SAME epilogue as (a)       (d) pv_gemm_bf16 with its fused epilogue (what the engine launches)
interleaved rounds in one process, bf16 operands for all four.  gpurun_out/vendor_vs_fused.json"""
import json, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from peekvit_amd import ops
from peekvit_amd._lib import PV_EPI_BIAS_BF16, PV_EPI_BIAS_GELU_BF16, PV_EPI_BIAS_RES_F32
dev = "cuda:0"
M = int(os.environ.get("M", 403456))
g = torch.Generator(device=dev).manual_seed(0)
out = {}
for name, N, K, fused in [("qkv", 2304, 768, PV_EPI_BIAS_BF16), ("out", 768, 768, PV_EPI_BIAS_RES_F32), ("fc1", 3072, 768, PV_EPI_BIAS_GELU_BF16),
                          ("fc2", 768, 3072, PV_EPI_BIAS_RES_F32)]:
    a = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g, device=dev) * K ** -0.5).to(torch.bfloat16)
    b32 = torch.randn(N, generator=g, device=dev)
    b16 = b32.to(torch.bfloat16)
    res = torch.randn(M, N, generator=g, device=dev) if fused == PV_EPI_BIAS_RES_F32 else None
    o16 = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    o32 = torch.empty((M, N), dtype=torch.float32, device=dev) if res is not None else None

    def vendor():
        return F.linear(a, w, b16)

    def vendor_plus():
        y = F.linear(a, w, b16)
        if fused == PV_EPI_BIAS_GELU_BF16:
            return F.gelu(y)
        if fused == PV_EPI_BIAS_RES_F32:
            return torch.add(res, y, out=o32)          # fp32 residual stream: read res + y, write fp32
        return y

    cases = {"vendor_bias": vendor, "vendor_plus_passes": vendor_plus,
             "pv_same_epilogue": lambda: ops.gemm(a, w, b32, o16, PV_EPI_BIAS_BF16),
             "pv_fused": lambda: ops.gemm(a, w, b32, o32 if res is not None else o16, fused, res=res)}
    times = {k: [] for k in cases}
    for k, fn in cases.items():
        for _ in range(2):
            fn()
    torch.cuda.synchronize()
    for _ in range(5):
        for k, fn in cases.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record(); torch.cuda.synchronize()
            times[k].append(e0.elapsed_time(e1) / 10)
    fl = 2.0 * M * N * K
    r = {k: {"ms": round(statistics.median(v), 4), "tflops": round(fl / statistics.median(v) / 1e9, 1)} for k, v in times.items()}
    out[name] = r
    print(name, r, flush=True)
    del a, w, res, o16, o32
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "vendor_vs_fused.json"), "w"), indent=1)
