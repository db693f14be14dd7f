"""A/B of the 128x128 GEMM kernel (forced with PV_GEMM_TILE=128) between two builds of the library, per shape, interleaved rounds in
one process: python scripts/gemm128_ab.py <lib_a.so> <lib_b.so>.  Also times the default dispatch (256^2 where eligible) of lib a."""
import ctypes as C, json, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("PV_GEMM_TILE", "128")
import torch
from peekvit_amd._lib import GemmArgs, PV_EPI_BIAS_BF16, PV_EPI_BIAS_GELU_BF16, PV_EPI_BIAS_RES_F32
dev = torch.device("cuda:0")
libs = {}
for path in sys.argv[1:3]:
    lib = C.CDLL(path)
    lib.pv_gemm_bf16.restype, lib.pv_gemm_bf16.argtypes = C.c_int, [C.POINTER(GemmArgs), C.c_void_p]
    libs[os.path.basename(path)] = lib
shapes = []
for tag, M, D, Mh in (("vit_small B512", 512 * 197, 384, 1536), ("vit_tiny B32", 32 * 401, 256, 768), ("vit_b16 B8", 8 * 197, 768, 3072),
                      ("rank S26 B2048", 2048 * 26, 768, 3072), ("vit_b16 B64", 64 * 197, 768, 3072)):
    shapes += [(tag + " qkv", M, 3 * D, D, PV_EPI_BIAS_BF16), (tag + " out", M, D, D, PV_EPI_BIAS_RES_F32),
               (tag + " fc1", M, Mh, D, PV_EPI_BIAS_GELU_BF16), (tag + " fc2", M, D, Mh, PV_EPI_BIAS_RES_F32)]
g = torch.Generator(device=dev).manual_seed(0)
stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
out_all = {}
for name, M, N, K, epi in shapes:
    a = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g, device=dev) * K ** -0.5).to(torch.bfloat16)
    bias = torch.randn(N, generator=g, device=dev)
    res = torch.randn(M, N, generator=g, device=dev) if epi == PV_EPI_BIAS_RES_F32 else None
    outs = {k: torch.empty((M, N), dtype=torch.float32 if res is not None else torch.bfloat16, device=dev) for k in libs}
    def args(o):
        return GemmArgs(A=a.data_ptr(), W=w.data_ptr(), bias=bias.data_ptr(), out=o.data_ptr(), res=res.data_ptr() if res is not None else 0,
                        M=M, N=N, K=K, lda=K, ldw=K, ldo=N, ldr=N, qscale=1.0, epilogue=epi)
    ga = {k: args(outs[k]) for k in libs}
    iters = max(5, min(200, int(2e3 / (2.0 * M * N * K / 4e14 * 1e3 + 0.02))))
    times = {k: [] for k in libs}
    for k, lib in libs.items():
        assert lib.pv_gemm_bf16(C.byref(ga[k]), stream) == 0
    torch.cuda.synchronize()
    ks = list(libs)
    same = torch.equal(outs[ks[0]], outs[ks[1]]) if len(ks) == 2 else None
    for _ in range(5):
        for k, lib in libs.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                lib.pv_gemm_bf16(C.byref(ga[k]), stream)
            e1.record(); torch.cuda.synchronize()
            times[k].append(e0.elapsed_time(e1) / iters * 1e3)
    r = {k: round(statistics.median(v), 2) for k, v in times.items()}
    fl = 2.0 * M * N * K
    print(f"{name:22s} M={M:6d} N={N:5d} K={K:5d}  " + "  ".join(f"{k}: {v:8.2f} us {fl / v / 1e6:7.1f} TF" for k, v in r.items()) + f"  bit-identical {same}", flush=True)
    out_all[name] = {"M": M, "N": N, "K": K, **r, "identical": same}
json.dump(out_all, open(os.path.join(ROOT, "gpurun_out", "gemm128_ab.json"), "w"), indent=1)
