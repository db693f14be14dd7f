"""A/B of 256^2 GEMM builds on the bf16/fp16-output token GEMMs (QKV, fc1 + GELU; with and without the folded LayerNorm).

    python scripts/gemm_epi_ab.py --build                 # (build container) compile the variant libraries
    python scripts/gemm_epi_ab.py --rounds 8 --out gpurun_out/gemm_epi_ab.json

Variants (all fp16-operand builds = the default inference library):
  r2      round 2's sources (git show <rev>: ten-instruction GELU, one-pass epilogue, GELU table staged first)
  nopipe  this round's sources with -DPV_EPI_PIPE=0: short GELU + table staged after the first K tiles, one-pass epilogue
  cur     this round's shipped library: the software-pipelined 16-bit epilogue
(profiles/r03_gemm_epilogue_persist_ab.json additionally holds a persistent-launch build, "cur" there, against "nopersist" = what ships:
 slower, removed - pv_gemm.hip, comment above pv_gemm256_kernel)
Interleaved rounds in ONE process (cdna_hip_programming.md rule 24); reports median and minimum per variant and shape, checks that
`cur` and `nopipe` agree bit for bit and that every variant is within rounding of an fp64 reference on a sample of rows.
"""
import argparse
import ctypes as C
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

R2_REV = "d1078c2"


def variant_paths():
    from peekvit_amd import _build
    return {"r2": os.path.join(_build.HERE, "libpeekvit_hip_r2f16.so"),
            "nopipe": os.path.join(_build.HERE, "libpeekvit_hip_nopipef16.so"),
            "pfm0": os.path.join(_build.HERE, "libpeekvit_hip_pfm0f16.so"),      # -DPV_PF_MODE=0: persistent, no prefetch
            "pfm1": os.path.join(_build.HERE, "libpeekvit_hip_pfm1f16.so"),      # -DPV_PF_MODE=1: prefetch retired before the first store
            "nt1": os.path.join(_build.HERE, "libpeekvit_hip_nt1f16.so"),        # -DPV_STORE_NT=1: non-temporal 16-bit output stores
            "nt3": os.path.join(_build.HERE, "libpeekvit_hip_nt3f16.so"),        # -DPV_STORE_NT=3: + the fp32 residual stream
            "nopf": _build.LIB_F16,            # the shipped library with pv_debug_set_gemm_pf(0): one tile per workgroup
            "cur": _build.LIB_F16}


def build():
    from peekvit_amd import _build
    _build.build()
    _build.build_variant("nopipef16", ["-DPV_OPERAND_F16", "-DPV_EPI_PIPE=0"])
    _build.build_variant("pfm0f16", ["-DPV_OPERAND_F16", "-DPV_PF_MODE=0"])
    _build.build_variant("pfm1f16", ["-DPV_OPERAND_F16", "-DPV_PF_MODE=1"])
    # round 2's kernels from history, compiled outside the tree (only the .so comes back)
    tmp = "/tmp/pv_r2_src"
    os.makedirs(os.path.join(tmp, "peekvit_amd", "csrc"), exist_ok=True)
    os.makedirs(os.path.join(tmp, "include"), exist_ok=True)
    files = subprocess.check_output(["git", "-C", ROOT, "ls-tree", "--name-only", R2_REV, "peekvit_amd/csrc/", "include/"]).decode().split()
    for f in files:
        with open(os.path.join(tmp, f), "wb") as fh:
            fh.write(subprocess.check_output(["git", "-C", ROOT, "show", f"{R2_REV}:{f}"]))
    objs = []
    for src in sorted(f for f in files if f.endswith(".hip")):
        obj = os.path.join(tmp, os.path.basename(src)[:-4] + ".o")
        subprocess.check_call([_build.HIPCC, *_build.FLAGS, "-DPV_OPERAND_F16", "-c", os.path.join(tmp, src), "-o", obj])
        objs.append(obj)
    subprocess.check_call([_build.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", variant_paths()["r2"], *objs])
    print("built", variant_paths())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--rounds", type=int, default=8)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--M", type=int, default=403456)
    ap.add_argument("--repeat", type=int, default=10)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    if a.build:
        return build()
    import numpy as np
    import torch
    from peekvit_amd import _lib
    from peekvit_amd._lib import GemmArgs, PV_EPI_BIAS_BF16, PV_EPI_BIAS_GELU_BF16

    class GemmArgsV6(C.Structure):          # round 2's pv_gemm_args (ABI v6): no leading struct_size
        _fields_ = [f for f in GemmArgs._fields_ if f[0] != "struct_size"]

    dev = "cuda:0"
    libs = {}
    for tag, path in variant_paths().items():
        if not os.path.exists(path):
            print("missing", path)
            continue
        lib = C.CDLL(path)
        lib.pv_gemm_bf16.restype = C.c_int
        lib.pv_gemm_bf16.argtypes = [C.c_void_p, C.c_void_p]
        libs[tag] = lib
    only = [t for t in os.environ.get("AB_VARIANTS", "").split(",") if t]
    if only:
        libs = {t: l for t, l in libs.items() if t in only}
    stream = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device=dev).manual_seed(0)
    M = a.M
    from peekvit_amd._lib import PV_EPI_BIAS_RES_F32
    shapes = [("qkv", 2304, 768, PV_EPI_BIAS_BF16, False), ("qkv_fold", 2304, 768, PV_EPI_BIAS_BF16, True),
              ("fc1", 3072, 768, PV_EPI_BIAS_GELU_BF16, False), ("fc1_fold", 3072, 768, PV_EPI_BIAS_GELU_BF16, True),
              ("out", 768, 768, PV_EPI_BIAS_RES_F32, False), ("fc2", 768, 3072, PV_EPI_BIAS_RES_F32, False)]
    keep = [t for t in os.environ.get("AB_SHAPES", "").split(",") if t]
    if keep:
        shapes = [sh for sh in shapes if sh[0] in keep]
    result = {"M": M, "rounds": a.rounds, "iters": a.iters, "shapes": {}}
    for name, N, K, epi, fold in shapes:
        x = torch.randn(M, K, generator=g, device=dev)
        A = x.to(torch.float16)
        W = (torch.randn(N, K, generator=g, device=dev) * K ** -0.5).to(torch.float16)
        bias = torch.randn(N, generator=g, device=dev) * 0.1
        stat = torch.stack([torch.randn(M, generator=g, device=dev) * 0.05, 1.0 + 0.1 * torch.rand(M, generator=g, device=dev)], 1).contiguous()
        c1 = W.float().sum(1).contiguous()
        odt = torch.float32 if epi == PV_EPI_BIAS_RES_F32 else torch.float16
        outs = {t: torch.empty((M, N), dtype=odt, device=dev) for t in libs}
        resid = torch.randn(M, N, generator=g, device=dev) if epi == PV_EPI_BIAS_RES_F32 else None
        x16 = {t: torch.empty((M, N), dtype=torch.float16, device=dev) for t in libs} if resid is not None else None
        rstat = {t: torch.empty((N // 256, M, 2), dtype=torch.float32, device=dev) for t in libs} if resid is not None else None

        def args(out, tag):
            ga = GemmArgsV6() if tag == "r2" else GemmArgs()
            ga.A, ga.W, ga.out = A.data_ptr(), W.data_ptr(), out.data_ptr()
            ga.M, ga.N, ga.K, ga.lda, ga.ldw, ga.ldo = M, N, K, K, K, N
            ga.epilogue = epi
            if name.startswith("qkv"):
                ga.qcols, ga.qscale = 768, 0.125
            if fold:
                ga.fold_stat, ga.fold_c1, ga.fold_c2 = stat.data_ptr(), c1.data_ptr(), bias.data_ptr()
            else:
                ga.bias = bias.data_ptr()
            if resid is not None:                  # the residual GEMMs as the folded forward runs them: + 16-bit copy + row statistics
                ga.res, ga.ldr = resid.data_ptr(), N
                ga.x16_out, ga.rowstat_out = x16[tag].data_ptr(), rstat[tag].data_ptr()
            return ga

        gargs = {t: args(outs[t], t) for t in libs}

        def run(t, n):
            if hasattr(libs[t], "pv_debug_set_gemm_pf") and t != "r2" and t != "nopipe":
                libs[t].pv_debug_set_gemm_pf(0 if t == "nopf" else 1)
            for _ in range(n):
                rc = libs[t].pv_gemm_bf16(C.byref(gargs[t]), stream)
                assert rc == 0, (t, name, rc)

        for t in libs:
            run(t, 2)
        torch.cuda.synchronize()
        # correctness: fp64 reference on 512 sampled rows
        rows = torch.randint(0, M, (512,), generator=g, device=dev)
        ref = A[rows].double() @ W.double().T
        if fold:
            ref = stat[rows, 1:2].double() * (ref - stat[rows, 0:1].double() * c1.double()[None]) + bias.double()[None]
        else:
            ref = ref + bias.double()[None]
        if epi == PV_EPI_BIAS_GELU_BF16:
            ref = ref * 0.5 * torch.erfc(-ref / 2 ** 0.5)
        elif epi == PV_EPI_BIAS_RES_F32:
            ref = ref + resid[rows].double()
        else:
            ref[:, :768] *= 0.125
        errs = {}
        for t in libs:
            got = outs[t][rows].double()
            errs[t] = float((got - ref).norm() / ref.norm())
        same = bool(torch.equal(outs["cur"], outs["nopipe"])) if "cur" in outs and "nopipe" in outs else None
        if "cur" in outs and "nopf" in outs:
            eq = bool(torch.equal(outs["cur"], outs["nopf"]))
            if resid is not None:
                eq = eq and bool(torch.equal(x16["cur"], x16["nopf"])) and bool(torch.equal(rstat["cur"], rstat["nopf"]))
            print(f"{name:9s} cur == nopf bitwise (all outputs): {eq}", flush=True)
        if resid is not None and "r2" in outs:           # the fp32 epilogue was restructured too: same bits as round 2's
            print(f"{name:9s} cur == r2 bitwise: out {bool(torch.equal(outs['cur'], outs['r2']))}, x16 copy {bool(torch.equal(x16['cur'], x16['r2']))}, "
                  f"row statistics max rel diff {float(((rstat['cur'] - rstat['r2']).abs() / rstat['r2'].abs().clamp_min(1e-6)).max()):.1e}", flush=True)
        # glitch screen: the kernels are deterministic, so every relaunch must reproduce the first output bit for bit (a rare stale
        # register read - see pv_gelu_poly_g in pv_gemm.hip - shows up as a handful of differing elements)
        repeat_diff = {}
        for t in libs:
            first = outs[t].clone()
            nd = 0
            for _ in range(a.repeat):
                outs[t].zero_()
                run(t, 1)
                nd += int((outs[t] != first).sum())
            repeat_diff[t] = nd
        times = {t: [] for t in libs}
        for r in range(a.rounds):
            for t in libs:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                run(t, a.iters)
                e1.record()
                torch.cuda.synchronize()
                times[t].append(e0.elapsed_time(e1) / a.iters)
        fl = 2.0 * M * N * K
        row = {}
        for t in libs:
            med, mn = statistics.median(times[t]), min(times[t])
            row[t] = {"median_ms": round(med, 4), "min_ms": round(mn, 4), "tflops_median": round(fl / med / 1e9, 1), "rel_l2_vs_fp64": errs[t]}
            print(f"{name:9s} {t:7s} median {med:7.4f} ms  min {mn:7.4f} ms  {fl / med / 1e9:7.1f} TF/s   rel L2 vs fp64 {errs[t]:.2e}", flush=True)
        row["cur_equals_nopipe_bitwise"] = same
        row["elements_differing_over_relaunches"] = repeat_diff
        print(f"{name:9s} cur == nopipe bitwise: {same}; elements differing over {a.repeat} relaunches: {repeat_diff}", flush=True)
        result["shapes"][name] = row
        del A, W, x, outs, resid, x16, rstat
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        with open(a.out, "w") as f:
            json.dump(result, f, indent=1)


if __name__ == "__main__":
    main()
