"""Forward attention with operands in L2 (the -DPV_BH_HOT build of scripts/attn_bwd4_ab.py --build) against the shipped build: what the kernel costs without HBM."""
import ctypes as C, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from peekvit_amd import _build
import torch
dev = "cuda:0"
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
P, I = C.c_void_p, C.c_int64
libs = {"f16": C.CDLL(_build.LIB_F16), "f16/hot": C.CDLL(os.path.join(_build.HERE, "libpeekvit_hip_abw4_hot.so"))}
for lib in libs.values():
    lib.pv_attention_bf16.argtypes = [P, P] + [I] * 4 + [P, P]
for H, dh, B, S in ((12, 64, 2048, 197), (6, 64, 512, 197), (12, 64, 2048, 99), (12, 64, 2048, 50)):
    D = H * dh
    qkv = (torch.randn(B, S, 3 * D, device=dev) * 0.7).to(torch.float16)
    out = torch.empty(B, S, D, dtype=torch.float16, device=dev)
    t = {k: [] for k in libs}
    for rnd in range(3):
        for k, lib in libs.items():
            for _ in range(2):
                assert lib.pv_attention_bf16(qkv.data_ptr(), out.data_ptr(), B, S, H, dh, None, stream) == 0
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                lib.pv_attention_bf16(qkv.data_ptr(), out.data_ptr(), B, S, H, dh, None, stream)
            e1.record(); torch.cuda.synchronize()
            t[k].append(e0.elapsed_time(e1) / 10)
    print(f"H={H} dh={dh} B={B} S={S}: " + "  ".join(f"{k} {statistics.median(v):.3f} ms" for k, v in t.items()), flush=True)
