"""In-model A/B of full-row GEMM variants inside ONE process (vit_small, batch 512): pv_debug_set_fullrow_dp values interleaved, rounds of 100 forwards.
python scripts/vit_small_inproc_ab.py 1 2     (1 = NT/2 phases of 16 MFMAs per K-tile, 2 = two phases, 0 = plain loop)"""
import ctypes as C, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from peekvit_amd import _lib, synth
from peekvit_amd.models.vit import VisionTransformer
modes = [int(a) for a in sys.argv[1:]] or [1, 2]
libs = [_lib.load()]
try:
    libs.append(C.CDLL(os.path.join(ROOT, "peekvit_amd", "libpeekvit_hip_f16.so")))
except OSError:
    pass
cfg = synth.MODEL_CONFIGS["vit_small"]
m = VisionTransformer(**cfg); synth.load_synth_weights(m, cfg); m = m.eval().cuda()
x = torch.randn(512, 3, 224, 224, device="cuda").to(torch.bfloat16).float()
def run(n):
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    with torch.no_grad():
        for _ in range(n): m(x)
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
with torch.no_grad():
    for _ in range(20): m(x)
res = {k: [] for k in modes}
for r in range(8):
    for k in modes:
        for l in libs: l.pv_debug_set_fullrow_dp(k)
        run(10)
        res[k].append(run(100))
for k in modes:
    print(f"dp={k}: median {statistics.median(res[k]):.4f} ms  ({512 / statistics.median(res[k]):.1f} k img/s)  rounds " + " ".join(f"{v:.3f}" for v in res[k]))
