"""In-model A/B of one debug setter of the library inside ONE process, interleaved rounds:  python scripts/model_knob_ab.py <model> <batch> <pv_debug_set_*> <value> [value ...]"""
import ctypes as C, os, statistics, sys
ROOT = os.getcwd(); sys.path.insert(0, ROOT)
import torch
from peekvit_amd import _lib, synth
from peekvit_amd.models.vit import VisionTransformer
name, B = sys.argv[1], int(sys.argv[2]); setter = sys.argv[3]; vals = [int(v) for v in sys.argv[4:]]
libs = [_lib.load()]
try: libs.append(C.CDLL(os.path.join(ROOT, "peekvit_amd", "libpeekvit_hip_f16.so")))
except OSError: pass
cfg = synth.MODEL_CONFIGS[name]
m = VisionTransformer(**cfg); synth.load_synth_weights(m, cfg); m = m.eval().cuda()
x = torch.randn(B, 3, cfg["image_size"], cfg["image_size"], device="cuda").to(torch.bfloat16).float()
def run(n):
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    with torch.no_grad():
        for _ in range(n): m(x)
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n
with torch.no_grad():
    for _ in range(10): m(x)
n = 60 if B * cfg["image_size"] < 200000 else 8
res = {v: [] for v in vals}
for r in range(5):
    for v in vals:
        for l in libs: getattr(l, setter)(v)
        run(3); res[v].append(run(n))
for v in vals: print(f"{setter}({v}): median {statistics.median(res[v]):.4f} ms  " + " ".join(f"{t:.3f}" for t in res[v]), flush=True)
