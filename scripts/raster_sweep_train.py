"""Round 5: raster sweep of the two TRAINING GEMMs that move the most bytes per FLOP - fc1 with the pair epilogue (gelu | pre: 4.96 GB written) and the
fc2 data gradient with the gelu' epilogue (2.48 GB of saved pre-activations read) - whose tile raster had only ever been inherited from the forward's fc1.
    python scripts/raster_sweep_train.py [--rounds 5]"""
import argparse, ctypes as C, json, os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from peekvit_amd import _lib
from peekvit_amd._lib import GemmArgs
ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=5); ap.add_argument("--iters", type=int, default=6); ap.add_argument("--M", type=int, default=403456)
a = ap.parse_args()
lib = _lib.load("f16")
lib.pv_debug_set_gemm_raster.argtypes = [C.c_int, C.c_int]
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
M, N, K = a.M, 3072, 768
A = torch.randn(M, K, generator=g, device=dev).to(torch.float16)
W = (torch.randn(N, K, generator=g, device=dev) * K ** -0.5).to(torch.float16)
bias = torch.randn(N, generator=g, device=dev) * 0.1
pre = torch.randn(M, N, generator=g, device=dev).to(torch.float16)
rasters = [(6, 6), (4, 6), (3, 6), (2, 6), (8, 6), (12, 6), (4, 4), (8, 4), (6, 3), (11, 3), (3, 12), (4, 12), (6, 12), (2, 12), (16, 2), (32, 2), (32, 1), (64, 1)]
out_json = {}
for name, epi in (("fc1_pair", 6), ("fc2_dgrad_gelu_grad", 7)):
    out = torch.empty((M, 2 * N if epi == 6 else N), dtype=torch.float16, device=dev)
    ga = GemmArgs(A=A.data_ptr(), W=W.data_ptr(), out=out.data_ptr(), M=M, N=N, K=K, lda=K, ldw=K, ldo=out.shape[1], epilogue=epi)
    if epi == 6:
        ga.bias = bias.data_ptr()
    else:
        ga.res, ga.ldr = pre.data_ptr(), N
    times = {r: [] for r in rasters}
    for rnd in range(a.rounds + 1):
        for r in rasters:
            lib.pv_debug_set_gemm_raster(r[0], r[1])
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                rc = lib.pv_gemm_bf16(C.byref(ga), st)
                assert rc == 0, rc
            e1.record(); torch.cuda.synchronize()
            if rnd:
                times[r].append(e0.elapsed_time(e1) / a.iters)
    lib.pv_debug_set_gemm_raster(0, 0)
    row = {f"gm{r[0]}_gc{r[1]}": round(statistics.median(t), 4) for r, t in times.items()}
    out_json[name] = row
    print(name, " ".join(f"{k}={v:.4f}" for k, v in sorted(row.items(), key=lambda kv: kv[1])), flush=True)
    del out
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out_json, open("gpurun_out/r05_raster_sweep_train.json", "w"), indent=1)
