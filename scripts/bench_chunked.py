"""Does a chunk-major forward (the whole 12-layer forward on one chunk of images after the other, every intermediate small enough
to stay in the 256 MiB Infinity Cache) beat the layer-major forward at batch 2048?  Per chunk size: one hipGraph of the forward
(peekvit_amd.graph.GraphedForward), replayed 2048 / chunk times; images/s, board power, sclk.  gpurun_out/bench_chunked.json"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import torch
from raster_ab import PowerSampler
from peekvit_amd import synth
from peekvit_amd.graph import GraphedForward
from peekvit_amd.models.vit import VisionTransformer

dev = torch.device("cuda:0")
pr = torch.cuda.get_device_properties(0)
sm = PowerSampler(f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0")
if not sm.cards:
    sm = PowerSampler(None)
sm.start()
cfg = synth.MODEL_CONFIGS[os.environ.get("MODEL", "vit_b_16")]
m = VisionTransformer(**cfg)
synth.load_synth_weights(m, cfg)
m = m.eval().to(dev)
B = int(os.environ.get("B", 2048))
x = torch.randn(B, 3, cfg["image_size"], cfg["image_size"], generator=torch.Generator(device=dev).manual_seed(0), device=dev)
out = {}
chunks = [int(c) for c in os.environ.get("CHUNKS", "64,96,128,160,192,256,512,2048").split(",")]
for chunk in chunks:
    n = B // chunk
    if chunk == B:
        run = lambda: m(x)
        with torch.no_grad():
            for _ in range(2): run()
    else:
        gf = GraphedForward(m, x[:chunk])
        def run():
            for _ in range(n):
                gf.graph.replay()
    torch.cuda.synchronize()
    reps = 6
    t0 = time.perf_counter()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    with torch.no_grad():
        for _ in range(reps): run()
    e1.record(); torch.cuda.synchronize()
    t1 = time.perf_counter()
    ms = e0.elapsed_time(e1) / reps
    out[chunk] = {"ms_per_2048": round(ms * 2048 / (n * chunk), 2), "img_s": round(n * chunk / ms * 1e3, 1), **sm.mean(t0 + 0.3 * (t1 - t0), t1)}
    print(chunk, out[chunk], flush=True)
    if chunk != B: del gf
sm.stop = True
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "bench_chunked.json"), "w"), indent=1)
