"""A/B of the attention backward from the forward's statistics (pv_attention_bwd_lse_bf16) against the two-pass kernel that recomputes them
(pv_attention_bwd_bf16), both operand builds, with an fp64 check on a small batch.
  python scripts/attn_bwd4_ab.py --build   (here)        python scripts/attn_bwd4_ab.py   (on the GPU box)"""
import ctypes as C, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from peekvit_amd import _build
VARIANTS = {"hot": ["-DPV_BH_HOT=1"], "nw7": ["-DPV_ABW4_NW=7"]}     # hot: all workgroups on eight images (operands in L2): the kernels without their HBM traffic (timings only)
if "--build" in sys.argv:
    _build.build()
    for tag, d in VARIANTS.items():
        print(_build.build_variant("abw4_" + tag, d + ["-DPV_OPERAND_F16"]))
    sys.exit(0)
import torch
dev = "cuda:0"
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
P, I, F = C.c_void_p, C.c_int64, C.c_float


def load(path):
    lib = C.CDLL(path)
    lib.pv_attention_bwd_bf16.restype = C.c_int
    lib.pv_attention_bwd_bf16.argtypes = [P] * 4 + [I] * 4 + [F, P]
    lib.pv_attention_lse_bf16.restype = C.c_int
    lib.pv_attention_lse_bf16.argtypes = [P, P, P] + [I] * 4 + [P, P]
    lib.pv_attention_bwd_lse_bf16.restype = C.c_int
    lib.pv_attention_bwd_lse_bf16.argtypes = [P] * 6 + [I] * 4 + [F, P]
    return lib


def ref64(qkv, dout, B, S, H, dh, qscale):
    D = H * dh
    x = qkv.double().view(B, S, 3, H, dh).permute(2, 0, 3, 1, 4).clone().requires_grad_(True)
    q, k, v = x[0], x[1], x[2]
    o = torch.softmax(q @ k.transpose(-1, -2), -1) @ v
    o.permute(0, 2, 1, 3).reshape(B, S, D).backward(dout.double())
    g = x.grad.clone()
    g[0] *= qscale
    return g.permute(1, 3, 0, 2, 4).reshape(B, S, 3 * D)


libs = [("bf16", torch.bfloat16, _build.LIB), ("f16", torch.float16, _build.LIB_F16)]
for v in VARIANTS:
    p = os.path.join(_build.HERE, f"libpeekvit_hip_abw4_{v}.so")
    if os.path.exists(p):
        libs.append(("f16/" + v, torch.float16, p))
for tag, dt, path in libs:
    lib = load(path)
    for H, dh, B, S in ((12, 64, 2048, 197), (12, 64, 2048, 99), (12, 64, 2048, 50), (6, 64, 2048, 197), (3, 64, 2048, 197), (12, 32, 512, 401), (12, 48, 512, 197), (12, 64, 512, 1), (12, 64, 512, 17)):
        D = H * dh
        g = torch.Generator(device=dev).manual_seed(0)
        qkv = (torch.randn(B, S, 3 * D, generator=g, device=dev) * 0.7).to(dt)
        dout = (torch.randn(B, S, D, generator=g, device=dev) * 0.1).to(dt)
        att = torch.empty(B, S, D, dtype=dt, device=dev)
        lse = torch.empty(B, H, S, dtype=torch.float32, device=dev)
        flag = torch.zeros(64, dtype=torch.int32, device=dev)
        rc = lib.pv_attention_lse_bf16(qkv.data_ptr(), att.data_ptr(), lse.data_ptr(), B, S, H, dh, flag.data_ptr(), stream)
        assert rc == 0, rc
        dbps = {k: torch.full((B, 3 * D), float('nan'), device=dev) for k in ('bwd2', 'bwd4')}
        outs, times = {}, {"bwd2": [], "bwd4": []}
        for rnd in range(3):
            for k in ("bwd2", "bwd4"):
                dqkv = torch.zeros_like(qkv)
                def run():
                    if k == "bwd2":
                        return lib.pv_attention_bwd_bf16(qkv.data_ptr(), dout.data_ptr(), dqkv.data_ptr(), dbps[k].data_ptr(), B, S, H, dh, dh ** -0.5, stream)
                    return lib.pv_attention_bwd_lse_bf16(qkv.data_ptr(), dout.data_ptr(), att.data_ptr(), lse.data_ptr(), dqkv.data_ptr(), dbps[k].data_ptr(), B, S, H, dh, dh ** -0.5, stream)
                for _ in range(2):
                    rc = run()
                    assert rc == 0, (k, rc)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    run()
                e1.record(); torch.cuda.synchronize()
                times[k].append(e0.elapsed_time(e1) / 10)
                outs[k] = dqkv
        nb = 8
        r = ref64(qkv[:nb], dout[:nb], nb, S, H, dh, dh ** -0.5)
        line = f"{tag:9s} H={H} dh={dh} B={B} S={S}:"
        for k in ("bwd2", "bwd4"):
            o = outs[k][:nb].double()
            errs = [float((o[..., i * D:(i + 1) * D] - r[..., i * D:(i + 1) * D]).norm() / r[..., i * D:(i + 1) * D].norm()) for i in range(3)]
            line += f"  {k} {statistics.median(times[k]):.3f} ms err(dq,dk,dv) " + "/".join(f"{e:.1e}" for e in errs) + f" finite={bool(torch.isfinite(outs[k].float()).all())}"
        ref_db = outs["bwd4"].float().sum(1)          # column sums of the stored values
        for k in ("bwd2", "bwd4"):
            line += f"  {k} db-err {float((dbps[k] - ref_db).abs().max() / ref_db.abs().max()):.1e}"
        print(line, flush=True)
