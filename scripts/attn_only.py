"""The attention launch of ViT-B/16 at batch 2048, alone (for rocprofv3 --pmc passes and timing): python3 scripts/attn_only.py [iters] [S]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from peekvit_amd import engine, ops
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
B, S, H, dh = 2048, int(sys.argv[2]) if len(sys.argv) > 2 else 197, 12, 64
with engine.precision("f16"):
    qkv = (torch.randn(B, S, 3 * H * dh, device="cuda:0") * 0.5).to(torch.float16)
    out = torch.empty(B, S, H * dh, dtype=torch.float16, device="cuda:0")
    for _ in range(3):
        ops.attention(qkv, out, B, S, H, dh)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.attention(qkv, out, B, S, H, dh)
    e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
print(f"pv_attention_bf16 B={B} S={S} H={H} dh={dh}: {ms:.4f} ms  {4.0 * B * H * S * S * dh / ms / 1e9:.0f} TF/s  {8.0 * B * S * H * dh / ms / 1e6:.0f} GB/s")
