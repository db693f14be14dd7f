"""Where the attention-backward kernel's time goes: the shipped kernel against two diagnostic builds that return after staging / after
pass 1 (scripts/raster_ab.py-style side-by-side libraries).  ViT-B/16 shape, B = 2048."""
import ctypes as C, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if "--build" in sys.argv:
    from peekvit_amd import _build
    _build.build()
    for tag, d in (("abw1", "-DPV_ATTN_BWD_STOP=1"), ("abw2", "-DPV_ATTN_BWD_STOP=2")):
        print(_build.build_variant(tag, [d]))
    sys.exit(0)
import torch
from peekvit_amd import _build
dev = "cuda:0"
B, S, H, dh = 2048, 197, 12, 64
D = H * dh
g = torch.Generator(device=dev).manual_seed(0)
qkv = (torch.randn(B, S, 3 * D, generator=g, device=dev) * 0.5).to(torch.bfloat16)
dout = (torch.randn(B, S, D, generator=g, device=dev) * 0.1).to(torch.bfloat16)
dqkv = torch.empty_like(qkv)
dbp = torch.empty(B, 3 * D, device=dev)
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
libs = {"full": _build.LIB, "staging only": os.path.join(_build.HERE, "libpeekvit_hip_abw1.so"), "staging + pass 1": os.path.join(_build.HERE, "libpeekvit_hip_abw2.so")}
fns = {}
for k, path in libs.items():
    lib = C.CDLL(path)
    lib.pv_attention_bwd_bf16.restype = C.c_int
    lib.pv_attention_bwd_bf16.argtypes = [C.c_void_p] * 4 + [C.c_int64] * 4 + [C.c_float, C.c_void_p]
    fns[k] = lib
times = {k: [] for k in fns}
for _ in range(4):
    for k, lib in fns.items():
        for _ in range(2):
            assert lib.pv_attention_bwd_bf16(qkv.data_ptr(), dout.data_ptr(), dqkv.data_ptr(), dbp.data_ptr(), B, S, H, dh, dh ** -0.5, stream) == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            lib.pv_attention_bwd_bf16(qkv.data_ptr(), dout.data_ptr(), dqkv.data_ptr(), dbp.data_ptr(), B, S, H, dh, dh ** -0.5, stream)
        e1.record(); torch.cuda.synchronize()
        times[k].append(e0.elapsed_time(e1) / 10)
for k, v in times.items():
    print(f"{k:18s} {statistics.median(v):.3f} ms")
