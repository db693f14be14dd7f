"""count rare wrong GELU outputs of the 128^2 kernel (global-memory table) per library variant"""
import os, sys, math, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["PV_GEMM_TILE"] = "128"
import torch
from peekvit_amd._lib import GemmArgs, PV_EPI_BIAS_GELU_BF16
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
M, N, K = 2560, 3072, 768
a = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16)
w = ((torch.rand(N, K, generator=g, device=dev) * 2 - 1) / math.sqrt(K)).to(torch.bfloat16)
bias = (torch.rand(N, generator=g, device=dev) * 2 - 1) * 0.1
pre = a.double() @ w.double().t() + bias.double()
ref = torch.nn.functional.gelu(pre)
st = torch.cuda.current_stream().cuda_stream
for tag in sys.argv[1:]:
    lib = C.CDLL(os.path.join(ROOT, "peekvit_amd", f"libpeekvit_hip{tag}.so"))
    lib.pv_gemm_bf16.argtypes = [C.c_void_p, C.c_void_p]
    bad_runs, bad_total = 0, 0
    for it in range(40):
        out = torch.zeros((M, N), dtype=torch.bfloat16, device=dev)
        ga = GemmArgs(A=a.data_ptr(), W=w.data_ptr(), bias=bias.data_ptr(), out=out.data_ptr(), M=M, N=N, K=K, lda=K, ldw=K, ldo=N, epilogue=PV_EPI_BIAS_GELU_BF16)
        assert lib.pv_gemm_bf16(C.byref(ga), st) == 0
        err = (out.double() - ref).abs()
        badm = err > 0.02 * ref.abs() + 1e-3
        bad = int(badm.sum())
        bad_runs += bad > 0; bad_total += bad
        if bad:
            cols = torch.cat([cols, badm.nonzero()[:, 1].cpu()]) if "cols" in dir() else badm.nonzero()[:, 1].cpu()
    print(f"variant '{tag}': runs with glitches {bad_runs}/40, wrong elements {bad_total}", flush=True)
    if bad_total:
        # which of a lane's four consecutive columns (c & 3: the two pv_gelu_lut2 pairs are (0, 1) and (2, 3)) and which 16-lane group (c >> 2 & 3)
        print("   wrong elements by (column & 3):", torch.bincount(cols & 3, minlength=4).tolist(), " by lane group ((column >> 2) & 3):", torch.bincount((cols >> 2) & 3, minlength=4).tolist(), flush=True)
        del cols
