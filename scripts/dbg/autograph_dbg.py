"""Diagnostic: eager vs engine-initiated hipGraph replay vs explicit GraphedForward on vit_tiny, batch 32."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from peekvit_amd import autograph, engine, synth
from peekvit_amd.graph import GraphedForward
from peekvit_amd.models.vit import VisionTransformer
cfg = synth.MODEL_CONFIGS["vit_tiny"]
m = VisionTransformer(**cfg); synth.load_synth_weights(m, cfg); m = m.eval().to("cuda:0")
x = torch.from_numpy(synth.synth_images(32, cfg["image_size"], seed=11)).to("cuda:0")
with torch.no_grad():
    outs = []
    for i in range(10):
        outs.append(m(x))
        print(i, "captures", autograph.captures, "replays", autograph.replays, "equal to first:", bool(torch.equal(outs[-1], outs[0])),
              "max diff", float((outs[-1] - outs[0]).abs().max()))
    autograph.ENABLED = False
    g = GraphedForward(m, x, warmup=1)
    y = g(x).clone()
    print("explicit graph equal to eager:", bool(torch.equal(y, outs[0])), float((y - outs[0]).abs().max()))
    g2 = GraphedForward(m, x, warmup=1, capture_error_mode="thread_local")
    y2 = g2(x).clone()
    print("explicit graph (thread_local) equal to eager:", bool(torch.equal(y2, outs[0])), float((y2 - outs[0]).abs().max()))
