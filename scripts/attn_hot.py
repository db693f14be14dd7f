"""What the attention kernels cost without HBM: the shipped fp16 build against a -DPV_BH_HOT build in which every workgroup works on one of eight images
(operands stay in the L2s; results are meaningless, timings are the point).  profiles/r05_attn_bwd4_experiment.txt reads these numbers.
  python scripts/attn_hot.py --build   (here)        python scripts/attn_hot.py   (on the GPU box)"""
import ctypes as C, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from peekvit_amd import _build
if "--build" in sys.argv:
    _build.build()
    print(_build.build_variant("attn_hot", ["-DPV_BH_HOT=1", "-DPV_OPERAND_F16"]))
    sys.exit(0)
import torch
dev = "cuda:0"
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
P, I, F = C.c_void_p, C.c_int64, C.c_float
libs = {"shipped": C.CDLL(_build.LIB_F16), "operands in L2": C.CDLL(os.path.join(_build.HERE, "libpeekvit_hip_attn_hot.so"))}
for lib in libs.values():
    lib.pv_attention_bf16.argtypes = [P, P] + [I] * 4 + [P, P]
    lib.pv_attention_bwd_bf16.argtypes = [P] * 4 + [I] * 4 + [F, P]
for H, dh, B, S in ((12, 64, 2048, 197), (8, 48, 512, 197), (6, 64, 512, 197), (3, 64, 512, 197), (12, 64, 2048, 99), (12, 64, 2048, 50)):
    D = H * dh
    qkv = (torch.randn(B, S, 3 * D, device=dev) * 0.7).to(torch.float16)
    dout = (torch.randn(B, S, D, device=dev) * 0.1).to(torch.float16)
    out, dqkv, dbp = torch.empty(B, S, D, dtype=torch.float16, device=dev), torch.empty_like(qkv), torch.empty(B, 3 * D, device=dev)
    t = {(k, w): [] for k in libs for w in ("fwd", "bwd")}
    for rnd in range(3):
        for k, lib in libs.items():
            for w in ("fwd", "bwd"):
                def run():
                    if w == "fwd":
                        return lib.pv_attention_bf16(qkv.data_ptr(), out.data_ptr(), B, S, H, dh, None, stream)
                    return lib.pv_attention_bwd_bf16(qkv.data_ptr(), dout.data_ptr(), dqkv.data_ptr(), dbp.data_ptr(), B, S, H, dh, dh ** -0.5, stream)
                for _ in range(2):
                    assert run() == 0
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    run()
                e1.record(); torch.cuda.synchronize()
                t[(k, w)].append(e0.elapsed_time(e1) / 10)
    print(f"H={H} dh={dh} B={B} S={S}: " + "  ".join(f"{w} {k} {statistics.median(v):.3f} ms" for (k, w), v in t.items()), flush=True)
