"""End-to-end images/sec of an evaluation loop INCLUDING the host->device copy (the reference's definition, validate/test.py:113-124:
len(dataset) / wall time of the loader loop), ViT-B/16 at batch 2048 on one MI355X, against the device-resident headline:
  (a) the reference's loop verbatim: synchronous fp32 copy, one .item() per batch
  (b) the same fp32 batches through harness.pipeline.DevicePrefetcher (copy of batch i+1 under the forward of batch i)
  (c) uint8 NHWC batches (normalisation fused into the patch gather) through the prefetcher
Host batches are pre-built pinned tensors (a real loader's decode / augment cost is not what is measured here)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from peekvit_amd import synth
from peekvit_amd.harness.pipeline import DevicePrefetcher
from peekvit_amd.models.vit import VisionTransformer

B, NB = int(os.environ.get("B", 2048)), int(os.environ.get("NB", 8))
dev = torch.device("cuda:0")
cfg = synth.MODEL_CONFIGS["vit_b_16"]
m = VisionTransformer(**cfg); synth.load_synth_weights(m, cfg); m = m.eval().to(dev)
g = torch.Generator().manual_seed(0)
f32 = [torch.randn(B, 3, 224, 224, generator=g).pin_memory() for _ in range(2)]
u8 = [torch.randint(0, 256, (B, 224, 224, 3), generator=g, dtype=torch.uint8).pin_memory() for _ in range(2)]
lab = torch.randint(0, 1000, (B,), generator=g).pin_memory()
out = {"batch": B, "batches": NB}


def loader(bufs):
    for i in range(NB):
        yield bufs[i % 2], lab


@torch.no_grad()
def run(kind):
    hits = torch.zeros((), dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if kind == "reference loop, fp32":
        c = 0
        for x, y in loader(f32):
            x, y = x.to(dev), y.to(dev)
            c += int((m(x).argmax(1) == y).sum().item())
    else:
        for x, y in DevicePrefetcher(loader(f32 if "fp32" in kind else u8), dev):
            hits += (m(x).argmax(1) == y).sum()
        hits.item()
    torch.cuda.synchronize()
    return B * NB / (time.perf_counter() - t0)


with torch.no_grad():
    x = f32[0].to(dev)
    for _ in range(3):
        m(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(NB):
        m(x)
    torch.cuda.synchronize()
    out["device-resident input (bench.py's value)"] = round(B * NB / (time.perf_counter() - t0), 1)
    # raw copy rates
    for name, t in (("fp32 NCHW", f32[0]), ("uint8 NHWC", u8[0])):
        d = torch.empty(t.shape, dtype=t.dtype, device=dev)
        d.copy_(t, non_blocking=True); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(4):
            d.copy_(t, non_blocking=True)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 4
        out[f"H2D {name} batch"] = {"GB": round(t.numel() * t.element_size() / 1e9, 3), "ms": round(dt * 1e3, 2), "GB/s": round(t.numel() * t.element_size() / dt / 1e9, 1)}
for kind in ("reference loop, fp32", "prefetcher, fp32", "prefetcher, uint8 NHWC"):
    run(kind)
    out[kind + " (img/s incl. H2D)"] = round(max(run(kind) for _ in range(2)), 1)
print(json.dumps(out, indent=1))
