"""Diagnostic (round 6): does vit_small's attention (4 096 items on 768 workgroup slots = 5.33 rounds) lose its partial last round?  Time per item at batches
whose item counts are whole and fractional numbers of rounds."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from peekvit_amd import _lib, ops
_lib.set_operand("f16")
S, H, dh = 197, 8, 48
D = H * dh
for B in (384, 480, 512, 544, 576, 672, 768):
    qkv = (torch.randn(B * S, 3 * D, device="cuda") * 0.7).to(torch.float16)
    out = torch.empty(B * S, D, device="cuda", dtype=torch.float16)
    for _ in range(5):
        ops.attention(qkv, out, B, S, H, dh)
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.attention(qkv, out, B, S, H, dh)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 20 * 1e3)
    t = sorted(ts)[2]
    print(f"B={B:4d} items={B*H:5d} rounds={B*H/768:5.2f}  {t:7.1f} us  {t/(B*H)*1e3:6.2f} ns/item")
