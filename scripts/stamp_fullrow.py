"""Diagnostic: -DPV_STAMPS build of the GEMM; where a full-row tile's cycles go (prologue / K loop / epilogue passes) and the clock it ran at."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from peekvit_amd._lib import GemmArgs
so = os.path.join(ROOT, "gpurun_out", "libpv_stamps.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DPV_STAMPS", "-DPV_OPERAND_F16",
                       os.path.join(ROOT, "peekvit_amd/csrc/pv_gemm.hip"), "-o", so])
lib = C.CDLL(so)
lib.pv_gemm_bf16.argtypes = [C.POINTER(GemmArgs), C.c_void_p]; lib.pv_debug_set_stamp_buffer.argtypes = [C.c_void_p]
lib.pv_debug_set_fullrow_dp.argtypes = [C.c_int]; lib.pv_debug_set_fullrow_split.argtypes = [C.c_int]
dev = "cuda:0"; M = int(os.environ.get("M", 512 * 197))
g = torch.Generator(device=dev).manual_seed(0)
for name, N, K in [("out", 384, 384), ("fc2", 384, 1536)]:
    a = torch.randn(M, K, generator=g, device=dev).to(torch.float16)
    w = (torch.randn(N, K, generator=g, device=dev) * K ** -0.5).to(torch.float16)
    bias, res = torch.randn(N, generator=g, device=dev), torch.randn(M, N, generator=g, device=dev)
    gam, bet = torch.rand(N, generator=g, device=dev) + 0.5, torch.randn(N, generator=g, device=dev) * 0.1
    out, h = torch.empty(M, N, device=dev), torch.empty(M, N, dtype=torch.float16, device=dev)
    for dp in (1,):
        lib.pv_debug_set_fullrow_dp(dp)
        nblk = 2048
        dbg = torch.zeros(nblk * 16, dtype=torch.int64, device=dev)
        lib.pv_debug_set_stamp_buffer(dbg.data_ptr())
        args = GemmArgs(A=a.data_ptr(), W=w.data_ptr(), bias=bias.data_ptr(), out=out.data_ptr(), res=res.data_ptr(), row_scale=0, pos=0, M=M, N=N, K=K,
                        lda=K, ldw=K, ldo=N, ldr=N, rows_per_img_in=0, rows_per_img_out=0, row_off=0, qcols=0, qscale=1.0, epilogue=2)
        args.ln_gamma, args.ln_beta, args.ln_out, args.ln_eps = gam.data_ptr(), bet.data_ptr(), h.data_ptr(), 1e-5
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        for _ in range(20):
            assert lib.pv_gemm_bf16(C.byref(args), st) == 0
        torch.cuda.synchronize()
        d = dbg.view(nblk, 16).cpu()
        used = d[:, 0] != 0
        for kind in (0, 1, 2):
            sel = used & (d[:, 7] == kind)
            if sel.sum() == 0: continue
            x = d[sel].double()
            seg = lambda i, j: (x[:, j] - x[:, i]).median().item()
            last = {0: 4, 1: 3, 2: 13}[kind]
            clk = ((x[:, last] - x[:, 0]) / ((x[:, 6] - x[:, 5]) * 10e-9)).median().item() / 1e9
            print(f"{name} K={K} dp={dp} {('full', 'half', 'big (160 rows)')[kind]} tiles {int(sel.sum())}: prologue {seg(0,1):.0f}  kloop {seg(1,2):.0f} ({seg(1,2) / (K // 64):.0f}/ktile)  "
                  f"pass0 {seg(2,3):.0f} [K-loop end -> barrier {seg(2,8):.0f} | image + barrier {seg(8,10):.0f} | batch 0: image rows + residual -> fp32 stores {seg(10,11):.0f} | LayerNorm + 16-bit stores {seg(11,12):.0f} | batch 1 {seg(12,3):.0f}]  pass1 {seg(3,4) if kind != 1 else 0:.0f}  pass2 {seg(4,13) if kind == 2 else 0:.0f}  total {seg(0,last):.0f} ticks  clock {clk:.2f} GHz", flush=True)
