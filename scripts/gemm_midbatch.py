"""Which tile (256^2 / 128^2) serves the four token GEMMs of ViT-B/16 best at MID batch sizes (wave quantisation of few 256^2 tiles on 256 CUs)?
Runs itself twice (PV_GEMM_TILE unset = the dispatch's choice, = 128 forced) and prints both per shape.  python scripts/gemm_midbatch.py"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from peekvit_amd import ops
    from peekvit_amd._lib import PV_EPI_BIAS_BF16, PV_EPI_BIAS_GELU_BF16, PV_EPI_BIAS_RES_F32
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(0)
    D, Mh = 768, 3072
    out = {}
    for B in (16, 32, 48, 64, 96, 128, 192, 256, 384):
        M = B * 197
        for name, N, K, epi in (("qkv", 3 * D, D, PV_EPI_BIAS_BF16), ("out", D, D, PV_EPI_BIAS_RES_F32), ("fc1", Mh, D, PV_EPI_BIAS_GELU_BF16), ("fc2", D, Mh, PV_EPI_BIAS_RES_F32)):
            a = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16)
            w = (torch.randn(N, K, generator=g, device=dev) * K ** -0.5).to(torch.bfloat16)
            bias = torch.randn(N, generator=g, device=dev)
            res = torch.randn(M, N, generator=g, device=dev) if epi == PV_EPI_BIAS_RES_F32 else None
            o = torch.empty((M, N), dtype=torch.float32 if res is not None else torch.bfloat16, device=dev)
            f = lambda: ops.gemm(a, w, bias, o, epi, res=res)
            for _ in range(5): f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 50
            e0.record()
            for _ in range(n): f()
            e1.record(); torch.cuda.synchronize()
            out[f"B{B} {name}"] = {"us": round(e0.elapsed_time(e1) / n * 1e3, 1), "tile": ops.gemm_tile_rows(M, N, K, epi)}
    print(json.dumps(out))
    sys.exit(0)
res = {}
for tag, env in (("dispatch", {}), ("forced128", {"PV_GEMM_TILE": "128"})):
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], capture_output=True, text=True, env=dict(os.environ, **env), timeout=600)
    res[tag] = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
for k in res["dispatch"]:
    d, f = res["dispatch"][k], res["forced128"][k]
    print(f"{k:12s} dispatch {d['tile']:3d}-row tile {d['us']:8.1f} us   forced 128: {f['us']:8.1f} us   {'128 wins by %.0f %%' % (100 * (d['us'] / f['us'] - 1)) if f['us'] < d['us'] * 0.97 and d['tile'] == 256 else ''}")
