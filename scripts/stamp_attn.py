"""Diagnostic: -DPV_STAMPS build of the attention kernel; per-wave cycle shares (issue | DMA wait | compute)."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
so = os.path.join(ROOT, "gpurun_out", "libpv_astamps.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
src = os.path.join(ROOT, "peekvit_amd/csrc/pv_attention.hip")
code = open(src).read().replace("#else\n#define PV_ASTAMP(i)", "#else\n#define PV_ASTAMP(i)")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DPV_STAMPS", src, "-o", so])
lib = C.CDLL(so)
B, S, H, dh = 2048, 197, 12, 64
dev = "cuda:0"
qkv = (torch.randn(B, S, 3 * H * dh, device=dev) * 0.5).to(torch.bfloat16)
out = torch.empty(B, S, H * dh, dtype=torch.bfloat16, device=dev)
dbg = torch.zeros(B * H * 4 * 8, dtype=torch.int64, device=dev)
# set the __device__ pointer via hipMemcpyToSymbol equivalent: use the module's symbol through hip runtime
hip = C.CDLL("libamdhip64.so")
sym = C.c_void_p(); size = C.c_size_t()
# simpler: the kernel reads d_pv_adbg; write it with hipMemcpyToSymbol through the library's own symbol address
addr = C.c_void_p.in_dll(lib, "d_pv_adbg") if False else None
lib.pv_attention_bf16.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]
lib.pv_debug_set_attn_stamp_buffer.argtypes = [C.c_void_p]
lib.pv_debug_set_attn_stamp_buffer(dbg.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(3):
    assert lib.pv_attention_bf16(qkv.data_ptr(), out.data_ptr(), B, S, H, dh, None, st) == 0
torch.cuda.synchronize()
d = dbg.view(B * H, 4, 8).cpu().double()
for w in range(4):
    x = d[:, w]
    print(f"wave {w}: issue {(x[:,1]-x[:,0]).median():.0f}  dma-wait {(x[:,2]-x[:,1]).median():.0f}  compute {(x[:,3]-x[:,2]).median():.0f}  total {(x[:,3]-x[:,0]).median():.0f}")
