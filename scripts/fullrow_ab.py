"""Full-row GEMM (+ fused LayerNorm) against the tile kernels + standalone LayerNorm, per residual-GEMM shape of the narrow models,
interleaved rounds in one process.  gpurun_out/fullrow_ab.json"""
import ctypes as C, json, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from peekvit_amd import _lib, ops
from peekvit_amd._lib import PV_EPI_BIAS_RES_F32
dev = "cuda:0"
lib = _lib.load()
lib.pv_debug_set_gemm_fullrow.restype, lib.pv_debug_set_gemm_fullrow.argtypes = None, [C.c_int]
g = torch.Generator(device=dev).manual_seed(0)
out = {}
for name, M, N, K in [("vit_small out", 512 * 197, 384, 384), ("vit_small fc2", 512 * 197, 384, 1536), ("vit_tiny out B32", 32 * 401, 256, 256),
                      ("vit_tiny fc2 B32", 32 * 401, 256, 768), ("vit_tiny out B512", 512 * 401, 256, 256), ("vit_tiny fc2 B512", 512 * 401, 256, 768),
                      ("D512 out", 65536, 512, 512), ("D512 fc2", 65536, 512, 2048)]:
    a = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g, device=dev) * K ** -0.5).to(torch.bfloat16)
    bias, res = torch.randn(N, generator=g, device=dev), torch.randn(M, N, generator=g, device=dev)
    gam, bet = torch.ones(N, device=dev), torch.zeros(N, device=dev)
    o, h = torch.empty(M, N, device=dev), torch.empty(M, N, dtype=torch.bfloat16, device=dev)

    def fused():
        lib.pv_debug_set_gemm_fullrow(1)
        ops.gemm(a, w, bias, o, PV_EPI_BIAS_RES_F32, res=res, ln=(gam, bet, 1e-5, h, None))

    def separate():
        lib.pv_debug_set_gemm_fullrow(0)
        ops.gemm(a, w, bias, o, PV_EPI_BIAS_RES_F32, res=res)
        ops.layernorm_bf16(o, gam, bet, 1e-5, h)

    def fullrow_noln():
        lib.pv_debug_set_gemm_fullrow(1)
        ops.gemm(a, w, bias, o, PV_EPI_BIAS_RES_F32, res=res)

    def tile_noln():
        lib.pv_debug_set_gemm_fullrow(0)
        ops.gemm(a, w, bias, o, PV_EPI_BIAS_RES_F32, res=res)

    cases = {"fused_fullrow+LN": fused, "tile+LN": separate, "fullrow": fullrow_noln, "tile": tile_noln}
    times = {k: [] for k in cases}
    for fn in cases.values():
        fn(); fn()
    torch.cuda.synchronize()
    iters = max(5, min(100, int(3e3 / (2.0 * M * N * K / 3e14 * 1e3 + 0.05))))
    for _ in range(5):
        for k, fn in cases.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                fn()
            e1.record(); torch.cuda.synchronize()
            times[k].append(e0.elapsed_time(e1) / iters * 1e3)
    r = {k: round(statistics.median(v), 1) for k, v in times.items()}
    out[name] = {"M": M, "N": N, "K": K, **r}
    print(f"{name:20s} M={M:6d} N={N} K={K:4d} " + "  ".join(f"{k} {v:7.1f} us" for k, v in r.items()), flush=True)
lib.pv_debug_set_gemm_fullrow(-1)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "fullrow_ab.json"), "w"), indent=1)
