"""Diagnostic: build a -DPV_STAMPS copy of the GEMM and print where a tile's cycles go (prologue / K-loop / epilogue issue / store drain)."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from peekvit_amd._lib import GemmArgs
so = os.path.join(ROOT, "gpurun_out", "libpv_stamps.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DPV_STAMPS", "-DPV_OPERAND_F16",
                       os.path.join(ROOT, "peekvit_amd/csrc/pv_gemm.hip"), "-o", so])
lib = C.CDLL(so)
lib.pv_gemm_bf16.argtypes = [C.POINTER(GemmArgs), C.c_void_p]; lib.pv_debug_set_stamp_buffer.argtypes = [C.c_void_p]
dev = "cuda:0"; M = int(os.environ.get("M", 403456))
g = torch.Generator(device=dev).manual_seed(0)
SL = 16          # stamp slots per workgroup
SHAPES = {"vit_b_16": [("qkv", 2304, 768, 0, False), ("qkv_fold", 2304, 768, 0, True), ("out", 768, 768, 2, False), ("fc1", 3072, 768, 1, False),
                       ("fc1_fold", 3072, 768, 1, True), ("fc2", 768, 3072, 2, False)],
          "vit_small": [("qkv", 1152, 384, 0, False), ("fc1", 1536, 384, 1, False)],
          # the training step's two byte-heaviest GEMMs: fc1 forward with the pair epilogue (gelu | pre-activation planes), fc2's data gradient with gelu' (round 6)
          "train": [("fc1_gelu_fwd", 3072, 768, 1, False), ("fc1_pair", 3072, 768, 6, False), ("dgelu", 3072, 768, 7, False), ("dgrad_plain", 3072, 768, 0, False)]}      # (M=100864; its residual GEMMs run the full-row kernel)
for name, N, K, epi, fold in SHAPES[os.environ.get("MODEL", "vit_b_16")]:
    a = torch.randn(M, K, generator=g, device=dev).to(torch.float16)
    w = (torch.randn(N, K, generator=g, device=dev) * K ** -0.5).to(torch.float16)
    bias = torch.randn(N, generator=g, device=dev)
    stat = torch.stack([torch.randn(M, generator=g, device=dev) * 0.05, 1.0 + 0.1 * torch.rand(M, generator=g, device=dev)], 1).contiguous()
    c1 = w.float().sum(1).contiguous()
    out = torch.empty((M, 2 * N if epi == 6 else N), dtype=torch.float32 if epi == 2 else torch.float16, device=dev)
    res = torch.randn(M, N, generator=g, device=dev) if epi == 2 else torch.randn(M, N, generator=g, device=dev).to(torch.float16) if epi == 7 else None
    if epi == 7:
        bias = None
    nblk = ((M + 255) // 256) * ((N + 255) // 256)
    dbg = torch.zeros(nblk * SL, dtype=torch.int64, device=dev)
    lib.pv_debug_set_stamp_buffer(dbg.data_ptr())
    args = GemmArgs(A=a.data_ptr(), W=w.data_ptr(), bias=bias.data_ptr() if bias is not None else 0, out=out.data_ptr(), res=res.data_ptr() if res is not None else 0,
                    row_scale=0, pos=0, M=M, N=N, K=K, lda=K, ldw=K, ldo=out.shape[1], ldr=N, rows_per_img_in=0, rows_per_img_out=0, row_off=0,
                    qcols=0, qscale=1.0, epilogue=epi)
    if fold:
        args.bias, args.fold_stat, args.fold_c1, args.fold_c2 = 0, stat.data_ptr(), c1.data_ptr(), bias.data_ptr()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        assert lib.pv_gemm_bf16(C.byref(args), st) == 0
    torch.cuda.synchronize()
    d = dbg.view(nblk, SL).cpu().double()
    seg = lambda i, j: (d[:, j] - d[:, i])
    tot = seg(0, 4)
    # per-CU timeline from the 100 MHz wall clock: busy = sum of workgroup lifetimes, span = first start .. last end
    import collections
    raw = dbg.view(nblk, SL).cpu()
    hw, t0, t1 = raw[:, 5], raw[:, 13].double(), raw[:, 7].double()      # kernel entry .. after the last store has drained
    cu_key = ((hw >> 32) & 0xf) * 4096 + ((hw >> 8) & 0xf) + (((hw >> 12) & 0x1) << 4) + (((hw >> 13) & 0x7) << 5)   # xcc | cu_id, sh_id, se_id
    per = collections.defaultdict(list)
    for k_, a_, b_ in zip(cu_key.tolist(), t0.tolist(), t1.tolist()):
        per[k_].append((a_, b_))
    utils, gaps, lifes = [], [], []
    for k_, iv in per.items():
        iv.sort()
        busy = sum(b_ - a_ for a_, b_ in iv)
        utils.append(busy / max(iv[-1][1] - iv[0][0], 1))
        gaps += [iv[i + 1][0] - iv[i][1] for i in range(len(iv) - 1)]
        lifes += [b_ - a_ for a_, b_ in iv]
    gaps_t = torch.tensor(gaps) if gaps else torch.zeros(1)
    print(f"   CUs seen {len(per)}  workgroups/CU {nblk / max(len(per), 1):.1f}  lifetime median {torch.tensor(lifes).median() * 10:.0f} ns  "
          f"gap between consecutive workgroups on a CU: median {gaps_t.median() * 10:.0f} ns, mean {gaps_t.mean() * 10:.0f} ns  CU busy fraction {sum(utils) / len(utils):.3f}  "
          f"kernel span {(t1.max() - t0.min()) * 10 / 1e3:.1f} us")
    print(f"{name}: blocks {nblk}  (s_memtime ticks, median per block) prologue {seg(0,1).median():.0f}  kloop {seg(1,2).median():.0f} "
          f"({seg(1,2).median() / (K // 64):.0f}/ktile)  epilogue-issue {seg(2,3).median():.0f}  store-drain {seg(3,4).median():.0f}  total {tot.median():.0f}"
          f"   span first-start..last-end {(d[:,4].max() - d[:,0].min()):.0f}")
    print(f"      kernel entry -> first LDS-DMA issued (index math, bias loads, accumulator init) {seg(14, 0).median():.0f} ticks")
    if epi in (0, 1):
        print("      epilogue: K-loop end -> fold loads back %.0f | pass 0..3 arithmetic + image written %s | last read-back + store issue %.0f" % (
            seg(2, 8).median(), " ".join("%.0f" % seg(8 + q, 9 + q).median() for q in range(4)), seg(12, 3).median()))
    t_ms = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); lib.pv_gemm_bf16(C.byref(args), st); e1.record(); torch.cuda.synchronize()
        t_ms.append(e0.elapsed_time(e1))
    t_ms = sorted(t_ms)[2]
    print(f"      launch {t_ms:.3f} ms = {t_ms * 1e6 / (nblk / 256):.0f} ns per tile per CU; lifetime + gap = {torch.tensor(lifes).median() * 10 + gaps_t.median() * 10:.0f} ns")
