"""Diagnostic: -DPV_STAMPS build of pv_attn_bwd5_kernel; per-wave phase times (s_memtime ticks) of the LAST item of every workgroup.
  python scripts/stamp_attn_bwd5.py --build (here)     python scripts/stamp_attn_bwd5.py (GPU box)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from peekvit_amd import _build
if "--build" in sys.argv:
    print(_build.build_variant("abw5_stamps", ["-DPV_STAMPS", "-DPV_OPERAND_F16"]))
    sys.exit(0)
import torch
lib = C.CDLL(os.path.join(_build.HERE, "libpeekvit_hip_abw5_stamps.so"))
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
dbg = torch.zeros(256 * 16 * 8, dtype=torch.int64, device=dev)
torch.cuda.synchronize()
for it in range(3):
    if it == 2:
        lib.pv_debug_set_attn_stamp_buffer(dbg.data_ptr())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    assert lib.pv_attention_bwd_lse_bf16(qkv.data_ptr(), dout.data_ptr(), att.data_ptr(), lse.data_ptr(), dqkv.data_ptr(), dbp.data_ptr(), B, S, H, dh, dh ** -0.5, st) == 0
    e1.record(); torch.cuda.synchronize()
print(f"launch {e0.elapsed_time(e1):.3f} ms = {e0.elapsed_time(e1) * 1e3 / 96:.1f} us per item")
d = dbg.view(256, 16, 8).cpu().double()
names = ["wait A", "barrier 1", "pass 1 / idle", "wait E", "barrier 2", "pass 2 / side work"]
for w in range(16):
    x = d[:, w]
    print(f"wave {w:2d}: " + "  ".join(f"{n} {(x[:, i + 1] - x[:, i]).median():.0f}" for i, n in enumerate(names)) + f"  item {(x[:, 6] - x[:, 0]).median():.0f}")
