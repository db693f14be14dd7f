"""A/B of builds of the LDS-resident attention kernel (ViT-B/16, batch 2048, fp16 operands): interleaved rounds in one process, error against
fp64 on a sample, JSON out.
    python scripts/attn_ab.py --build ; python scripts/attn_ab.py [out.json]
variants: r2 = round 2's sources; cur = the shipped library; noflag = without -mllvm -amdgpu-mfma-vgpr-form=1 (MFMA results through AGPRs +
v_accvgpr_read copies); g0 = the attention-score guard compiled out; noshift = probabilities packed without the 2^14 scale.
(profiles/r03_attention_ab.json also holds the removed variants of this round - packed exp argument, row sum from the matrix pipe - from
the commit that still had them.)"""
import ctypes as C, json, os, statistics, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from peekvit_amd import _build
V = {"nw8": ["-DPV_ATTN_NW=8"], "g0": ["-DPV_SCORE_GUARD=0"]}      # round 5: nw8 = 8 waves per workgroup (2 workgroups per CU) instead of 4 (3 per CU)
path = lambda t: _build.LIB_F16 if t == "cur" else os.path.join(_build.HERE, "libpeekvit_hip_r2f16.so") if t == "r2" else os.path.join(_build.HERE, f"libpv_attn_{t}.so")
if "--build" in sys.argv:
    src = os.path.join(_build.CSRC, "pv_attention.hip")
    for t, d in V.items():
        flags = [] if t == "noflag" else _build.FILE_FLAGS["pv_attention.hip"]
        subprocess.check_call([_build.HIPCC, *_build.FLAGS, "-DPV_OPERAND_F16", *flags, *d, "-shared", src, "-o", path(t)])
    sys.exit(0)
import torch
B, S, H, dh = (int(a) for a in os.environ.get("PV_AB_SHAPE", "2048,197,12,64").split(","))
dev = "cuda:0"
qkv = (torch.randn(B, S, 3 * H * dh, device=dev) * 0.7).to(torch.float16)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
libs, outs = {}, {}
flagbuf = torch.zeros(1, dtype=torch.int32, device=dev)
for t in ["r2", *V, "cur"]:
    if not os.path.exists(path(t)):
        continue
    lib = C.CDLL(path(t))
    lib.pv_attention_bf16.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int64] * 4 + ([C.c_void_p] if t == "r2" else [C.c_void_p, C.c_void_p])
    libs[t] = lib
    outs[t] = torch.empty(B, S, H * dh, dtype=torch.float16, device=dev)
def run(t, n):
    for _ in range(n):
        rc = libs[t].pv_attention_bf16(qkv.data_ptr(), outs[t].data_ptr(), B, S, H, dh, st) if t == "r2" else \
             libs[t].pv_attention_bf16(qkv.data_ptr(), outs[t].data_ptr(), B, S, H, dh, flagbuf.data_ptr(), st)
        assert rc == 0
for t in libs: run(t, 3)
torch.cuda.synchronize()
q, k, v = (qkv[:4].double().view(4, S, 3, H, dh).permute(2, 0, 3, 1, 4)[i] for i in range(3))
ref = (torch.softmax(q @ k.transpose(-1, -2), -1) @ v).permute(0, 2, 1, 3).reshape(4, S, H * dh)
times = {t: [] for t in libs}
for r in range(6):
    for t in libs:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(t, 10); e1.record(); torch.cuda.synchronize()
        times[t].append(e0.elapsed_time(e1) / 10)
res = {}
for t in libs:
    err = float((outs[t][:4].double() - ref).norm() / ref.norm())
    bad = int((~torch.isfinite(outs[t])).sum())
    if bad:
        idx = (~torch.isfinite(outs[t])).nonzero()[:6].tolist()
        print(f"   {t}: {bad} non-finite outputs, first at [image, token, column] {idx}")
    res[t] = {"median_ms": round(statistics.median(times[t]), 4), "min_ms": round(min(times[t]), 4), "rel_l2_vs_fp64": err}
    print(f"{t:7s} median {statistics.median(times[t]):.4f} ms  min {min(times[t]):.4f} ms   rel L2 vs fp64 {err:.3e}", flush=True)
if len(sys.argv) > 1:
    json.dump({"shape": [B, S, H, dh], "score_scale": "q, k ~ N(0, 0.49): scores sigma 3.9, max ~20", "variants": res}, open(sys.argv[1], "w"), indent=1)
