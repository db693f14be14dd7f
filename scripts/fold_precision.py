"""Logits error of the LayerNorm-folding path (PEEKVIT_AMD_FOLD_LN=1) against the fp32 CPU oracle at a batch where folding is eligible."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from oracle import vit_oracle as O
from peekvit_amd import engine, ops, synth
from peekvit_amd.models.vit import VisionTransformer
for name, B in (("vit_b_16", 64), ("vit_small", 192), ("vit_tiny", 96)):
    cfg = synth.MODEL_CONFIGS[name]
    m = VisionTransformer(**cfg); synth.load_synth_weights(m, cfg); m = m.eval().cuda()
    x = torch.randn(B, 3, cfg["image_size"], cfg["image_size"], generator=torch.Generator().manual_seed(0))
    sd = {k: torch.from_numpy(v) for k, v in synth.synth_state_dict(cfg).items()}
    torch.set_num_threads(16)
    with torch.no_grad():
        ref = O.vit_forward(x, sd, cfg, "fp32")
        out = {}
        for fold in (False, True):
            engine._FOLD_LN = fold
            with ops.KernelTimer() as kt:
                got = m(x.cuda()).float().cpu()
            torch.cuda.synchronize()
            ks = kt.summary()
            out[fold] = (float((got - ref).norm() / ref.norm()), ks.get("pv_layernorm_bf16", {}).get("launches", 0), "pv_rowstat_finalize" in ks)
    print(name, "B", B, "plain err %.3e (LN launches %d)" % out[False][:2], "| fold err %.3e (LN launches %d, fold ran %s)" % out[True])
