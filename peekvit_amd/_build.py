"""Build libpeekvit_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libpeekvit_hip.so")
LIB_F16 = os.path.join(HERE, "libpeekvit_hip_f16.so")      # same sources, -DPV_OPERAND_F16: fp16 operands (precision mode "f16")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"]
# per-source extra flags.  pv_attention.hip: MFMA results in VGPRs.  hipcc otherwise puts the score / output tiles of the attention
# kernels into AGPRs and copies every one of them out with v_accvgpr_read_b32 before the softmax can touch it - 272 of a wave's
# 1 795 vector instructions in a kernel that is VALU-issue bound (round 3, rocprofv3 + ISA); with the flag: no copies, 116 VGPRs.
# pv_rowops.hip: no SLP vectorisation.  hipcc's SLP pass turns neighbouring scalar fp32 operations into v_pk_*_f32 and, where a scalar
# operand sits in an odd register or a pair is summed horizontally, sets an op_sel bit (the LOW result reads the HIGH register of a
# source pair) - the instruction form that returned wrong low results in lanes 48-63 while vector-memory loads were landing in VGPRs
# (round 4); these row kernels run under such loads all the time and are HBM-bound, so packing buys them nothing.
# tests/test_isa_audit.py keeps that form out of every kernel that computes under in-flight register loads.
FILE_FLAGS = {"pv_attention.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"], "pv_rowops.hip": ["-fno-slp-vectorize"]}


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _stale() -> bool:
    if not os.path.exists(LIB) or not os.path.exists(LIB_F16):
        return True
    t = min(os.path.getmtime(LIB), os.path.getmtime(LIB_F16))
    # (this file too: FLAGS / FILE_FLAGS are part of what the libraries were built with - round-4 review: a checkout that only changed a flag kept its old .so)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(os.path.dirname(HERE), "include", "peekvit_hip.h"), os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile every .hip source to an object (in parallel) and link the shared library."""
    if not force and not _stale():
        return LIB
    procs = []
    variants = [("build", [], LIB), ("build_f16", ["-DPV_OPERAND_F16"], LIB_F16)]
    objs = {lib: [] for _, _, lib in variants}
    for sub, defs, lib in variants:
        objdir = os.path.join(HERE, sub)
        os.makedirs(objdir, exist_ok=True)
        for src in sources():
            obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
            objs[lib].append(obj)
            cmd = [HIPCC, *FLAGS, *FILE_FLAGS.get(os.path.basename(src), []), *defs, "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out.decode(errors='replace')}")
    for _, _, lib in variants:
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs[lib]]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout.decode(errors='replace')}")
    return LIB


def build_variant(tag: str, defs, force: bool = False) -> str:
    """An EXPERIMENT build of the same sources: libpeekvit_hip_<tag>.so with extra -D flags (bf16 operands).  scripts/raster_ab.py
    loads such builds side by side with the shipped library to A/B compile-time knobs (cache policy of the GEMM staging loads) in
    one process; nothing in the package loads them."""
    lib = os.path.join(HERE, f"libpeekvit_hip_{tag}.so")
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(os.path.dirname(HERE), "include", "peekvit_hip.h")]
    if not force and os.path.exists(lib) and all(os.path.getmtime(d) <= os.path.getmtime(lib) for d in deps):
        return lib
    objdir = os.path.join(HERE, "build_" + tag)
    os.makedirs(objdir, exist_ok=True)
    procs, objs = [], []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        procs.append((src, subprocess.Popen([HIPCC, *FLAGS, *FILE_FLAGS.get(os.path.basename(src), []), *defs, "-c", src, "-o", obj],
                                            stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out.decode(errors='replace')}")
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout.decode(errors='replace')}")
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
