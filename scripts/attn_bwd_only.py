"""The attention-backward launch of a ViT-B/16 training step (batch 2048, fp16 operands), alone, for rocprofv3 --pmc passes:
  python3 scripts/attn_bwd_only.py [bwd4|bwd2] [iters] [S]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from peekvit_amd import _build
which = sys.argv[1] if len(sys.argv) > 1 else "bwd4"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 8
S = int(sys.argv[3]) if len(sys.argv) > 3 else 197
B, H, dh = 2048, 12, 64
D = H * dh
dev = "cuda:0"
P, I, F = C.c_void_p, C.c_int64, C.c_float
lib = C.CDLL(os.environ.get("PV_LIB", _build.LIB_F16))
lib.pv_attention_bwd_bf16.argtypes = [P] * 4 + [I] * 4 + [F, P]
lib.pv_attention_lse_bf16.argtypes = [P, P, P] + [I] * 4 + [P, P]
lib.pv_attention_bwd_lse_bf16.argtypes = [P] * 6 + [I] * 4 + [F, P]
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
g = torch.Generator(device=dev).manual_seed(0)
qkv = (torch.randn(B, S, 3 * D, generator=g, device=dev) * 0.7).to(torch.float16)
dout = (torch.randn(B, S, D, generator=g, device=dev) * 0.1).to(torch.float16)
att = torch.empty(B, S, D, dtype=torch.float16, device=dev)
lse = torch.empty(B, H, S, dtype=torch.float32, device=dev)
flag = torch.zeros(64, dtype=torch.int32, device=dev)
dqkv = torch.empty_like(qkv)
dbp = torch.empty(B, 3 * D, device=dev)
assert lib.pv_attention_lse_bf16(qkv.data_ptr(), att.data_ptr(), lse.data_ptr(), B, S, H, dh, flag.data_ptr(), stream) == 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for it in range(iters + 2):
    if it == 2:
        e0.record()
    if which == "bwd2":
        rc = lib.pv_attention_bwd_bf16(qkv.data_ptr(), dout.data_ptr(), dqkv.data_ptr(), dbp.data_ptr(), B, S, H, dh, dh ** -0.5, stream)
    else:
        rc = lib.pv_attention_bwd_lse_bf16(qkv.data_ptr(), dout.data_ptr(), att.data_ptr(), lse.data_ptr(), dqkv.data_ptr(), dbp.data_ptr(), B, S, H, dh, dh ** -0.5, stream)
    assert rc == 0, rc
e1.record(); torch.cuda.synchronize()
print(f"{which} S={S}: {e0.elapsed_time(e1) / iters:.3f} ms per launch")
