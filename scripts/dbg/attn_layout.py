"""does the [B,S,3D] packed layout (128-B head slices at 4.6-KB stride) cost the attention kernel HBM efficiency?  Same kernel, same bytes and
FLOPs, on (a) the real layout B=2048,H=12 and (b) B*H 'images' of one head each, [B*H, S, 3*dh] (a head's q|k|v rows 384 B apart)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from peekvit_amd import engine, ops
B, S, H, dh = 2048, 197, 12, 64
with engine.precision("f16"):
    for tag, (b, h) in (("packed [B,S,3*H*dh]", (B, H)), ("per-head [B*H,S,3*dh]", (B * H, 1))):
        qkv = (torch.randn(b, S, 3 * h * dh, device="cuda:0") * 0.5).to(torch.float16)
        out = torch.empty(b, S, h * dh, dtype=torch.float16, device="cuda:0")
        for _ in range(3): ops.attention(qkv, out, b, S, h, dh)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): ops.attention(qkv, out, b, S, h, dh)
        e1.record(); torch.cuda.synchronize()
        print(f"{tag}: {e0.elapsed_time(e1) / 20:.4f} ms")
