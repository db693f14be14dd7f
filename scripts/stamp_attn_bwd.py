"""Diagnostic: -DPV_STAMPS build of pv_attn_bwd4_kernel; per-wave time shares of its phases (s_memtime, 100 MHz ticks or shader cycles).
  python scripts/stamp_attn_bwd.py --build (here)     python scripts/stamp_attn_bwd.py (GPU box)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from peekvit_amd import _build
if "--build" in sys.argv:
    print(_build.build_variant("abw4_stamps", ["-DPV_STAMPS", "-DPV_OPERAND_F16"]))
    sys.exit(0)
import torch
lib = C.CDLL(os.path.join(_build.HERE, "libpeekvit_hip_abw4_stamps.so"))
B, S, H, dh = 2048, 197, 12, 64
D = H * dh
dev = "cuda:0"
P, I, F = C.c_void_p, C.c_int64, C.c_float
lib.pv_attention_lse_bf16.argtypes = [P, P, P] + [I] * 4 + [P, P]
lib.pv_attention_bwd_lse_bf16.argtypes = [P] * 6 + [I] * 4 + [F, P]
lib.pv_debug_set_attn_stamp_buffer.argtypes = [P]
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
g = torch.Generator(device=dev).manual_seed(0)
qkv = (torch.randn(B, S, 3 * D, generator=g, device=dev) * 0.7).to(torch.float16)
dout = (torch.randn(B, S, D, generator=g, device=dev) * 0.1).to(torch.float16)
att = torch.empty(B, S, D, dtype=torch.float16, device=dev)
lse = torch.empty(B, H, S, dtype=torch.float32, device=dev)
dqkv = torch.empty_like(qkv)
dbp = torch.empty(B, 3 * D, device=dev)
assert lib.pv_attention_lse_bf16(qkv.data_ptr(), att.data_ptr(), lse.data_ptr(), B, S, H, dh, None, st) == 0
dbg = torch.zeros(B * H * 8 * 8, dtype=torch.int64, device=dev)
torch.cuda.synchronize()
for it in range(3):
    if it == 2:
        lib.pv_debug_set_attn_stamp_buffer(dbg.data_ptr())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    assert lib.pv_attention_bwd_lse_bf16(qkv.data_ptr(), dout.data_ptr(), att.data_ptr(), lse.data_ptr(), dqkv.data_ptr(), dbp.data_ptr(), B, S, H, dh, dh ** -0.5, st) == 0
    e1.record(); torch.cuda.synchronize()
print(f"launch {e0.elapsed_time(e1):.3f} ms")
d = dbg.view(B * H, 8, 8).cpu().double()
span = d[:, :, 7].max() - d[:, :, 0].min()
print(f"whole kernel: {span:.0f} ticks -> {span / (e0.elapsed_time(e1) * 1e3):.1f} ticks per us")
names = ["issue+prologue", "wait+barrier", "pass 1", "barrier", "stage 2+wait", "barrier", "pass 2"]
for w in range(8):
    x = d[:, w]
    print(f"wave {w}: " + "  ".join(f"{n} {(x[:, i + 1] - x[:, i]).median():.0f}" for i, n in enumerate(names[:7])) + f"  total {(x[:, 7] - x[:, 0]).median():.0f}")
wg = d[:, :, 7].max(1).values - d[:, :, 0].min(1).values
print(f"workgroup lifetime: median {wg.median():.0f}  p10 {wg.quantile(0.1):.0f}  p90 {wg.quantile(0.9):.0f}")
