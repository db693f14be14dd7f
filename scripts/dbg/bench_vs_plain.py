"""Why is bench.py's timed eager loop slower than a plain loop for a 0.5 ms forward?  Same process, same model: plain loop, then with the pieces bench.py adds."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from peekvit_amd import engine, synth
from peekvit_amd.models.vit import VisionTransformer
cfg = synth.MODEL_CONFIGS["vit_tiny"]
model = VisionTransformer(**cfg); synth.load_synth_weights(model, cfg); model = model.eval().to("cuda:0")
torch.manual_seed(0)
x = torch.randn(32, 3, cfg["image_size"], cfg["image_size"], device="cuda:0")
def loop(n=50):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n):
        out = model(x)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3
with torch.no_grad():
    for _ in range(10): model(x)
    print("plain            %.3f %.3f %.3f" % (loop(), loop(), loop()))
    with engine.precision("auto"):
        print("precision(auto)  %.3f %.3f %.3f" % (loop(), loop(), loop()))
    engine.guard_state(model)
    print("plain again      %.3f %.3f" % (loop(), loop()))
    import bench  # noqa
    print("after import bench %.3f %.3f" % (loop(), loop()))
    from oracle import vit_oracle  # noqa
    print("after import oracle %.3f %.3f" % (loop(), loop()))
    print("threads", torch.get_num_threads())
