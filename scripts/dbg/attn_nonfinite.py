import sys, torch
sys.path.insert(0, "/root/repo")
from peekvit_amd import _lib, ops
for (B,S,H,dh,amp) in [(2,99,8,32,1.0),(2,99,8,32,3.0),(2,99,8,32,6.0),(2,99,8,64,6.0),(2,197,8,32,6.0),(2,50,8,32,6.0),(1,26,2,48,8.0)]:
    D=H*dh
    g = torch.Generator(device="cuda").manual_seed(S * 7 + dh)
    qkv = torch.randn(B, S, 3 * D, generator=g, device="cuda")
    qkv[..., :2 * D] *= amp
    qkv[..., :D] *= dh ** -0.5
    for op in ("f16","bf16"):
        old=_lib.set_operand(op)
        dt = torch.float16 if op=="f16" else torch.bfloat16
        out = torch.empty(B*S, D, device="cuda", dtype=dt)
        ops.attention(qkv.view(B*S,3*D).to(dt), out, B, S, H, dh)
        _lib.set_operand(old)
        bad = ~torch.isfinite(out.float())
        print(B,S,H,dh,amp,op,"nonfinite", int(bad.sum()), "rows", int(bad.any(1).sum()), "of", B*S)
