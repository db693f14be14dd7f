"""Round 4: full-row GEMM (+ fused LayerNorm) variants, interleaved, with a bit-identity check of the outputs: split remainder on / off, the plain
or the deep-pipelined K loop.  argv: mode numbers.  gpurun_out/fullrow_ab4.json"""
import ctypes as C, json, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from peekvit_amd import _lib, ops
from peekvit_amd._lib import PV_EPI_BIAS_RES_F32
dev = "cuda:0"
lib = _lib.load()
lib.pv_debug_set_fullrow_split.restype, lib.pv_debug_set_fullrow_split.argtypes = None, [C.c_int]
lib.pv_debug_set_fullrow_dp.restype, lib.pv_debug_set_fullrow_dp.argtypes = None, [C.c_int]
# mode = 10 * dp + split: dp 0 = the plain K loop (one vmcnt(0) + barrier per K-tile), 1 = the deep-pipelined one
modes = [int(x) for x in sys.argv[1:]] or [0, 1, 10, 11]
g = torch.Generator(device=dev).manual_seed(0)
out = {}
for name, M, N, K in [("vit_small out", 512 * 197, 384, 384), ("vit_small fc2", 512 * 197, 384, 1536), ("vit_tiny out B512", 512 * 401, 256, 256),
                      ("vit_tiny fc2 B512", 512 * 401, 256, 768), ("D512 out", 65536 + 77, 512, 512), ("D512 fc2", 65536 + 77, 512, 2048), ("small M", 1000, 384, 256)]:
    a = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g, device=dev) * K ** -0.5).to(torch.bfloat16)
    bias, res = torch.randn(N, generator=g, device=dev), torch.randn(M, N, generator=g, device=dev)
    gam, bet = torch.rand(N, generator=g, device=dev) + 0.5, torch.randn(N, generator=g, device=dev) * 0.1
    outs = {}
    times = {m: [] for m in modes}
    bufs = {m: (torch.empty(M, N, device=dev), torch.empty(M, N, dtype=torch.bfloat16, device=dev)) for m in modes}

    def run(m):
        lib.pv_debug_set_fullrow_split(m % 10)
        lib.pv_debug_set_fullrow_dp(m // 10 % 10)
        ops.gemm(a, w, bias, bufs[m][0], PV_EPI_BIAS_RES_F32, res=res, ln=(gam, bet, 1e-5, bufs[m][1], None))

    for m in modes:
        run(m); run(m)
    torch.cuda.synchronize()
    same = all(torch.equal(bufs[m][0], bufs[modes[0]][0]) and torch.equal(bufs[m][1].view(torch.int16), bufs[modes[0]][1].view(torch.int16)) for m in modes)
    iters = 50
    for _ in range(5):
        for m in modes:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                run(m)
            e1.record(); torch.cuda.synchronize()
            times[m].append(e0.elapsed_time(e1) / iters * 1e3)
    r = {f"mode{m}": round(statistics.median(v), 1) for m, v in times.items()}
    nbytes = 2.0 * M * K + 2.0 * N * K + 10.0 * M * N
    out[name] = {"M": M, "N": N, "K": K, "bit_identical": same, **r, "hbm_floor_us_at_6.29TBps": round(nbytes / 6.29e6, 1)}
    print(f"{name:20s} M={M:6d} N={N} K={K:4d} identical={same} " + "  ".join(f"{k} {v:7.1f} us" for k, v in r.items()) + f"   floor {nbytes / 6.29e6:6.1f} us", flush=True)
lib.pv_debug_set_fullrow_split(-1)
lib.pv_debug_set_fullrow_dp(-1)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "fullrow_ab4.json"), "w"), indent=1)
