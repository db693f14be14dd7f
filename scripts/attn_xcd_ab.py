"""Round 5: XCD-aware (image, head) mapping of the attention kernels (pv_bh_map) against the plain blockIdx / H mapping (-DPV_BH_XCD=0), forward and
backward, at the head sizes of vit_tiny (32), vit_small (48) and ViT-B/16 (64).    python scripts/attn_xcd_ab.py --build ; python scripts/attn_xcd_ab.py"""
import ctypes as C, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from peekvit_amd import _build
if "--build" in sys.argv:
    _build.build()
    print(_build.build_variant("noxcd", ["-DPV_BH_XCD=0"]))
    sys.exit(0)
import torch
dev = "cuda:0"
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
libs = {"xcd": C.CDLL(_build.LIB), "plain": C.CDLL(os.path.join(_build.HERE, "libpeekvit_hip_noxcd.so"))}
for lib in libs.values():
    lib.pv_attention_bf16.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int64] * 4 + [C.c_void_p, C.c_void_p]
    lib.pv_attention_bwd_bf16.argtypes = [C.c_void_p] * 4 + [C.c_int64] * 4 + [C.c_float, C.c_void_p]
for name, B, S, H, dh in (("vit_small", 512, 197, 8, 48), ("vit_tiny", 32, 401, 8, 32), ("vit_tiny b512", 512, 401, 8, 32), ("vit_b_16", 2048, 197, 12, 64), ("rankvit s=99", 2048, 99, 12, 64)):
    D = H * dh
    qkv = (torch.randn(B, S, 3 * D, device=dev) * 0.5).to(torch.bfloat16)
    dout = (torch.randn(B, S, D, device=dev) * 0.1).to(torch.bfloat16)
    outs, res = {}, {}
    for kind in ("fwd", "bwd"):
        times = {k: [] for k in libs}
        for rnd in range(5):
            for k, lib in libs.items():
                o = torch.empty(B, S, D if kind == "fwd" else 3 * D, device=dev, dtype=torch.bfloat16)
                def run(n):
                    for _ in range(n):
                        rc = lib.pv_attention_bf16(qkv.data_ptr(), o.data_ptr(), B, S, H, dh, None, st) if kind == "fwd" else \
                            lib.pv_attention_bwd_bf16(qkv.data_ptr(), dout.data_ptr(), o.data_ptr(), None, B, S, H, dh, dh ** -0.5, st)
                        assert rc == 0
                run(3)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); run(20); e1.record(); torch.cuda.synchronize()
                times[k].append(e0.elapsed_time(e1) / 20 * 1e3)
                outs[(kind, k)] = o
        same = torch.equal(outs[(kind, "xcd")], outs[(kind, "plain")])
        print(f"{name:14s} B={B} S={S} H={H} dh={dh} {kind}: xcd {statistics.median(times['xcd']):8.1f} us   plain {statistics.median(times['plain']):8.1f} us   bit-identical {same}", flush=True)
