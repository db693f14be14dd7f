"""The attention-backward launch of a ViT-B/16 training step (batch 2048, fp16 operands), alone, for rocprofv3 --pmc passes:
  python3 scripts/attn_bwd_only.py [lse|recompute] [iters] [S]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from peekvit_amd import ops, engine
which = sys.argv[1] if len(sys.argv) > 1 else "lse"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 8
S = int(sys.argv[3]) if len(sys.argv) > 3 else 197
B, H, dh = 2048, 12, 64
D = H * dh
dev = "cuda:0"
with engine.precision("f16"):
    g = torch.Generator(device=dev).manual_seed(0)
    qkv = (torch.randn(B, S, 3 * D, generator=g, device=dev) * 0.7).to(torch.float16)
    dout = (torch.randn(B, S, D, generator=g, device=dev) * 0.1).to(torch.float16)
    att = torch.empty(B, S, D, dtype=torch.float16, device=dev)
    lse = torch.empty(B, H, S, dtype=torch.float32, device=dev)
    dqkv = torch.empty_like(qkv)
    dbp = torch.empty(B, 3 * D, device=dev)
    ops.attention(qkv, att, B, S, H, dh, lse=lse)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for it in range(iters + 2):
        if it == 2:
            e0.record()
        if which == "recompute":
            ops.attention_bwd(qkv, dout, dqkv, B, S, H, dh, dh ** -0.5, dbias_partial=dbp)
        else:
            ops.attention_bwd_lse(qkv, dout, att, lse, dqkv, B, S, H, dh, dh ** -0.5, dbias_partial=dbp)
    e1.record(); torch.cuda.synchronize()
print(f"{which} S={S}: {e0.elapsed_time(e1) / iters:.3f} ms per launch")
