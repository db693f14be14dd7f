"""ResidualViT-B/16 inference at batch 2048: the gate kernel with / without the masked copy of the tokens (pv_gemm_args.res_scaled) and with /
without the block's first LayerNorm, one process, interleaved rounds."""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from peekvit_amd import engine, ops, synth
from peekvit_amd.models.residualvit import ResidualVisionTransformer
cfg = synth.MODEL_CONFIGS["vit_b_16"]
extra = dict(residual_layers=["attention+mlp"] * cfg["num_layers"], gate_temp=1, add_input=False, gate_type="sigmoid", gate_threshold=0.5,
             gate_bias=10, add_budget_token="learnable")
m = ResidualVisionTransformer(**cfg, **extra)
synth.load_synth_weights(m, dict(cfg, **extra), "residualvit", seed=0)
m = m.cuda().eval(); m.set_budget(0.5)
x = torch.randn(2048, 3, 224, 224, device="cuda")
cases = [(True, True), (False, True), (False, False)]        # (masked copy NOT written, LN1 in gate)
res = {c: [] for c in cases}
with torch.no_grad():
    for rnd in range(4):
        for fold in cases:
            engine._GATE_NO_MASKED, engine._GATE_LN1 = fold
            for _ in range(2): m(x)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(5): m(x)
            torch.cuda.synchronize(); res[fold].append((time.perf_counter() - t0) / 5 * 1e3)
    for fold in cases:
        engine._GATE_NO_MASKED, engine._GATE_LN1 = fold
        with ops.KernelTimer() as kt:
            m(x)
        torch.cuda.synchronize()
        print("(no masked copy, LN1 in gate) =", fold, f"{statistics.median(res[fold]):.2f} ms", {k: round(v["ms"], 2) for k, v in sorted(kt.summary().items(), key=lambda kv: -kv[1]["ms"])[:6]})
