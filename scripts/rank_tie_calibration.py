"""Calibrate engine.RANK_TIE_GAP: RankViT-B/16 [3, 6, 9] @ 0.5 on random images - per image the narrowest relative gap at a keep boundary (fp16 operands)
against whether the image kept another token SET than the split-operand arithmetic (= the reference's fp32 ranking).  gpurun_out/rank_tie_calibration.json"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from peekvit_amd import engine, synth
from peekvit_amd.models.rankvit import RankVisionTransformer
dev = "cuda:0"
cfg = synth.MODEL_CONFIGS["vit_b_16"]
m = RankVisionTransformer(**cfg, rankvit_layers=[3, 6, 9])
synth.load_synth_weights(m, cfg)
m = m.eval().to(dev)
m.set_budget(0.5)
ranked = [b for b in m.encoder.layers if hasattr(b, "sort_and_drop") and b.current_budget != 1]
B, rounds = 256, int(os.environ.get("ROUNDS", "4"))
gaps, flips, errs = [], [], []
with torch.no_grad():
    for r in range(rounds):
        x = torch.randn(B, 3, 224, 224, generator=torch.Generator(device=dev).manual_seed(100 + r), device=dev).to(torch.bfloat16).float()
        with engine.precision("f16"), engine.rank_gaps(B, x.device) as gap:
            y = m(x)
            sets = [torch.sort(b.last_keep, dim=1).values.clone() for b in ranked]
        with engine.precision("bf16x3"):
            ref = m(x)
            rsets = [torch.sort(b.last_keep, dim=1).values.clone() for b in ranked]
        agree = torch.ones(B, dtype=torch.bool, device=dev)
        for a, b in zip(sets, rsets):
            agree &= (a == b).all(dim=1)
        gaps.append(gap.cpu()); flips.append((~agree).cpu())
        errs.append(((y - ref).norm(dim=1) / ref.norm(dim=1)).cpu())
gap, flip, err = torch.cat(gaps), torch.cat(flips), torch.cat(errs)
out = {"images": int(gap.numel()), "flipped": int(flip.sum()), "largest_gap_of_a_flipped_image": float(gap[flip].max()) if flip.any() else None,
       "gap_quantiles_of_flipped_images": [float(q) for q in torch.quantile(gap[flip], torch.tensor([0.5, 0.9, 0.99, 1.0]))] if flip.any() else None,
       "logit_error_of_flipped_images_median": float(err[flip].median()) if flip.any() else None,
       "logit_error_of_unflipped_images_max": float(err[~flip].max()),
       "fraction_of_images_under_threshold": {str(t): float((gap < t).float().mean()) for t in (1e-4, 2e-4, 3e-4, 4e-4, 5e-4, 6e-4, 8e-4, 1e-3, 2e-3)},
       "flipped_images_caught_by_threshold": {str(t): float((gap[flip] < t).float().mean()) if flip.any() else None for t in (1e-4, 2e-4, 3e-4, 4e-4, 5e-4, 6e-4, 8e-4, 1e-3, 2e-3)}}
print(json.dumps(out, indent=1))
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "rank_tie_calibration.json"), "w"), indent=1)
