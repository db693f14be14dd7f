"""Raster sweep of the 256^2 token GEMMs with the shipped library (pv_debug_set_gemm_raster): python scripts/raster_sweep.py [--rounds 6]
Round 3 re-run of round 2's question (scripts/raster_ab.py) after the output stores became non-temporal - do larger groups pay now that the
outputs no longer pass through L2?  Interleaved rounds in one process, median ms per (shape, gm, gc)."""
import argparse, ctypes as C, json, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from peekvit_amd import _lib
from peekvit_amd._lib import GemmArgs

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=6)
ap.add_argument("--iters", type=int, default=8)
ap.add_argument("--M", type=int, default=403456)
a = ap.parse_args()
lib = _lib.load("f16")
lib.pv_debug_set_gemm_raster.argtypes = [C.c_int, C.c_int]
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
M = a.M
out_json = {}
for name, N, K, epi, fold, rasters in (
        ("qkv_fold", 2304, 768, 0, True, [(8, 0), (4, 0), (12, 0), (16, 0), (8, 5), (16, 5), (11, 3), (32, 3), (197, 0), (64, 0)]),
        ("fc1_fold", 3072, 768, 1, True, [(6, 6), (8, 6), (12, 6), (16, 6), (8, 4), (16, 4), (32, 4), (11, 3), (32, 3), (6, 0), (64, 6)]),
        ("fc2", 768, 3072, 2, False, [(1, 0), (2, 0), (4, 0), (8, 0), (16, 0)]),
        ("out", 768, 768, 2, False, [(1, 0), (2, 0), (4, 0), (8, 0), (16, 0)])):
    A = torch.randn(M, K, generator=g, device=dev).to(torch.float16)
    W = (torch.randn(N, K, generator=g, device=dev) * K ** -0.5).to(torch.float16)
    bias = torch.randn(N, generator=g, device=dev) * 0.1
    stat = torch.stack([torch.randn(M, generator=g, device=dev) * 0.05, 1.0 + 0.1 * torch.rand(M, generator=g, device=dev)], 1).contiguous()
    c1 = W.float().sum(1).contiguous()
    out = torch.empty((M, N), dtype=torch.float32 if epi == 2 else torch.float16, device=dev)
    res = torch.randn(M, N, generator=g, device=dev) if epi == 2 else None
    x16 = torch.empty((M, N), dtype=torch.float16, device=dev) if epi == 2 else None
    rstat = torch.empty((N // 256, M, 2), dtype=torch.float32, device=dev) if epi == 2 else None
    ga = GemmArgs(A=A.data_ptr(), W=W.data_ptr(), out=out.data_ptr(), M=M, N=N, K=K, lda=K, ldw=K, ldo=N, epilogue=epi)
    if fold:
        ga.fold_stat, ga.fold_c1, ga.fold_c2 = stat.data_ptr(), c1.data_ptr(), bias.data_ptr()
    else:
        ga.bias = bias.data_ptr()
    if epi == 0:
        ga.qcols, ga.qscale = 768, 0.125
    if res is not None:
        ga.res, ga.ldr, ga.x16_out, ga.rowstat_out = res.data_ptr(), N, x16.data_ptr(), rstat.data_ptr()
    times = {r: [] for r in rasters}
    for rnd in range(a.rounds + 1):
        for r in rasters:
            lib.pv_debug_set_gemm_raster(r[0], r[1])
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                assert lib.pv_gemm_bf16(C.byref(ga), st) == 0
            e1.record(); torch.cuda.synchronize()
            if rnd:
                times[r].append(e0.elapsed_time(e1) / a.iters)
    lib.pv_debug_set_gemm_raster(0, 0)
    row = {f"gm{r[0]}_gc{r[1]}": round(statistics.median(t), 4) for r, t in times.items()}
    out_json[name] = row
    print(name, " ".join(f"{k}={v:.4f}" for k, v in sorted(row.items(), key=lambda kv: kv[1])), flush=True)
    del A, W, out, res, x16, rstat
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out_json, open("gpurun_out/raster_sweep.json", "w"), indent=1)
