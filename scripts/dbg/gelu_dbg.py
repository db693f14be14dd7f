import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from peekvit_amd import ops
from peekvit_amd._lib import PV_EPI_BIAS_GELU_BF16
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
for (M, N, K) in [(2560, 3072, 768), (2432, 768, 3072), (394, 2304, 768)]:
    a = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16)
    w = ((torch.rand(N, K, generator=g, device=dev) * 2 - 1) / math.sqrt(K)).to(torch.bfloat16)
    bias = (torch.rand(N, generator=g, device=dev) * 2 - 1) * 0.1
    out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    ops.gemm(a, w, bias, out, PV_EPI_BIAS_GELU_BF16)
    pre = a.double() @ w.double().t() + bias.double()
    ref = torch.nn.functional.gelu(pre)
    err = (out.double() - ref)
    rel = float(err.norm() / ref.norm())
    bad = err.abs() > 0.02 * ref.abs() + 1e-3
    print(M, N, K, "rel", rel, "bad", int(bad.sum()), "of", M * N)
    if bad.any():
        idx = bad.nonzero()[:12]
        for r, c in idx.tolist():
            print("   ", r, c, "pre", float(pre[r, c]), "ref", float(ref[r, c]), "got", float(out[r, c]))
        rows = bad.any(1).nonzero().flatten(); cols = bad.any(0).nonzero().flatten()
        print("   rows%128", sorted(set((rows % 128).tolist()))[:40], "cols%128", sorted(set((cols % 128).tolist()))[:64])
