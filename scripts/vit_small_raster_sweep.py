"""vit_small (batch 512) forward time against the tile raster of the 256^2 GEMMs (QKV: 5 column tiles, fc1: 6; pv_debug_set_gemm_raster(gm, gc),
0 = the built-in choice), interleaved rounds in one process."""
import ctypes as C, os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from peekvit_amd import _lib, synth
from peekvit_amd.models.vit import VisionTransformer
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
cases = [(0, 0), (1, 0), (2, 0), (4, 0), (8, 0), (16, 0), (2, 3), (4, 3), (8, 3), (4, 2), (8, 2), (32, 0)]
res = {c: [] for c in cases}
for r in range(5):
    for c in cases:
        for l in libs: l.pv_debug_set_gemm_raster(*c)
        run(5)
        res[c].append(run(60))
for c in cases:
    print(f"gm={c[0]:2d} gc={c[1]}: median {statistics.median(res[c]):.4f} ms", flush=True)
