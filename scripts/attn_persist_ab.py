"""A/B of the persistent forward attention kernel (pv_attn_p_kernel) against the one-item-per-workgroup kernel, interleaved, ViT-B/16 and vit_small shapes."""
import ctypes as C, json, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from peekvit_amd import _lib, ops
_lib.set_operand("f16")
lib = _lib.load()
lib.pv_debug_set_attn_persist.restype, lib.pv_debug_set_attn_persist.argtypes = None, [C.c_int]
dev = "cuda:0"
res = {}
for name, B, S, H, dh in [("vit_b_16 B2048", 2048, 197, 12, 64), ("vit_small B512", 512, 197, 8, 48), ("vit_b_16 B256", 256, 197, 12, 64), ("vit_b_16 B1024 lse", 1024, 197, 12, 64)]:
    D = H * dh
    qkv = (torch.randn(B * S, 3 * D, device=dev) * 0.7).to(torch.float16)
    out = torch.empty(B * S, D, device=dev, dtype=torch.float16)
    lse = torch.empty(B, H, S, device=dev) if "lse" in name else None
    t = {0: [], 1: []}
    for rnd in range(6):
        for mode in (0, 1):
            lib.pv_debug_set_attn_persist(mode)
            for _ in range(3):
                ops.attention(qkv, out, B, S, H, dh, lse=lse)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops.attention(qkv, out, B, S, H, dh, lse=lse)
            e1.record(); torch.cuda.synchronize()
            t[mode].append(e0.elapsed_time(e1) / 20 * 1e3)
    res[name] = {"one_item_per_workgroup_us": round(statistics.median(t[0]), 1), "persistent_us": round(statistics.median(t[1]), 1)}
    print(name, res[name], flush=True)
lib.pv_debug_set_attn_persist(-1)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "attn_persist_ab.json"), "w"), indent=1)
