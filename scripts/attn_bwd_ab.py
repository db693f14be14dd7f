"""A/B of compile-time variants of the attention-backward kernel (side-by-side libraries, interleaved rounds in one process).
  python scripts/attn_bwd_ab.py --build      (here: cross-compiles the variants)      python scripts/attn_bwd_ab.py   (on the GPU box)"""
import ctypes as C, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# round 3: "shipped" = two key tiles per wave in pass 2 (rolled q-pair loop); kpw1 = round 2's one key tile per wave with three q-tile pairs per trip;
# kpw2u2 / kpw2u3 = two key tiles with two / three pairs per trip
# round 5: "shipped" = pv_attn_bwd2_kernel (two images in LDS at a time, 4 waves, 2-3 workgroups per CU); v1 = the round 1-4 kernel (four images, 8 waves, one per CU)
VARIANTS = {"v1": ["-DPV_ABW_V2=0"], "u2": ["-DPV_ABW2_P2_UNROLL=2"]}       # u2: pass 2 of the round-5 kernel with two query-tile pairs per loop trip
from peekvit_amd import _build
if "--build" in sys.argv:
    _build.build()
    for tag, d in VARIANTS.items():
        print(_build.build_variant("abw_" + tag, d))
    sys.exit(0)
import torch
dev = "cuda:0"
H, dh = 12, 64
D = H * dh
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
libs = {"shipped": _build.LIB}
libs.update({t: os.path.join(_build.HERE, f"libpeekvit_hip_abw_{t}.so") for t in VARIANTS})
fns = {}
for k, path in libs.items():
    if not os.path.exists(path):
        continue
    lib = C.CDLL(path)
    lib.pv_attention_bwd_bf16.restype = C.c_int
    lib.pv_attention_bwd_bf16.argtypes = [C.c_void_p] * 4 + [C.c_int64] * 4 + [C.c_float, C.c_void_p]
    fns[k] = lib
for B, S in ((2048, 197), (2048, 99), (2048, 50)):
    g = torch.Generator(device=dev).manual_seed(0)
    qkv = (torch.randn(B, S, 3 * D, generator=g, device=dev) * 0.5).to(torch.bfloat16)
    dout = (torch.randn(B, S, D, generator=g, device=dev) * 0.1).to(torch.bfloat16)
    dbp = torch.empty(B, 3 * D, device=dev)
    outs, times = {}, {k: [] for k in fns}
    for rnd in range(4):
        for k, lib in fns.items():
            dqkv = torch.zeros_like(qkv)
            for _ in range(2):
                assert lib.pv_attention_bwd_bf16(qkv.data_ptr(), dout.data_ptr(), dqkv.data_ptr(), dbp.data_ptr(), B, S, H, dh, dh ** -0.5, stream) == 0
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                lib.pv_attention_bwd_bf16(qkv.data_ptr(), dout.data_ptr(), dqkv.data_ptr(), dbp.data_ptr(), B, S, H, dh, dh ** -0.5, stream)
            e1.record(); torch.cuda.synchronize()
            times[k].append(e0.elapsed_time(e1) / 10)
            outs[k] = dqkv
    for k, v in times.items():
        err = float((outs[k].float() - outs["shipped"].float()).norm() / outs["shipped"].float().norm())
        print(f"B={B} S={S} {k:10s} {statistics.median(v):.3f} ms  rel diff to shipped: {err:.2e}")
