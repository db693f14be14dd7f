"""Analytic FLOP / sparsity accounting with the conventions of the reference's counter.

The reference measures FLOPs with ptflops plus two custom hooks (utils/flops_count.py:27-145) that fire when the stock
`nn.Linear.forward` / `nn.MultiheadAttention.forward` run and that DISCOUNT tokens whose input row is all zeros
("masked" tokens of ResidualViT).  The MI355X path never calls those forwards, so the hooks cannot fire; this module
restates their arithmetic analytically, per image, from the model's shapes and the per-block masks:

  Linear (flops_count.py:27-39)        MACs = (in*out + out[bias]) * live_rows
  MultiheadAttention (:45-145)         MACs = L*D (q scale) + 3*L*D*D + 3*L*D (in-proj + bias)
                                              + H*(L*L*dh + L*L + L*L*dh) + L*D*(D+1) (out-proj), L = live tokens
  Conv2d, LayerNorm                    ptflops defaults: out_elems*(Cin*k*k) + out_elems (bias); 2 * numel (affine LN)
  compute_flops returns 2 * MACs       (flops_count.py:175-180)

Pinned: `hook_macs` (the Linear + MultiheadAttention part, i.e. everything the reference's OWN hooks count) reproduces, to the
integer, the `__flops__` the real reference's hooks accumulate on the real reference's models (tests/golden/flops_hooks.json, made
by oracle/make_golden_aux.py; ViT, RankViT, ResidualViT with and without zero rows).  Unpinned: the conv / LayerNorm terms, which
come from ptflops' built-in hooks - ptflops is not installed in the build image, so those two are restated from ptflops 0.7.2.2's
published conv / norm hooks and NOT cross-checked against a ptflops run.

Which rows are discounted (found when pinning; flops_count.py:14-24 tests `sum(row) == 0` on each module's INPUT): the attention
input `mask * LN1(x)` and the fc1 input `mask * LN2(x1)` have zero rows where the mask is 0; the fc2 input `gelu(fc1(0) + b)` is not
zero, so fc2 is counted on every row; inputs with a single token ([B,1,D], the budget-token gate) or of rank 2 (the head) are never
discounted.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch


def _blocks(model) -> list:
    """The transformer blocks of `encoder.layers`.  A module spliced between them that owns no Linear / MultiheadAttention - the
    evaluation harness's NoiseBlock (harness/noise.py) - fires none of the reference's hooks and is skipped here as well."""
    return [m for m in model.encoder.layers if hasattr(m, "self_attention")]


def _linear_macs(rows: float, fin: int, fout: int, bias: bool = True) -> float:
    return (fin * fout + (fout if bias else 0)) * rows


def _mha_macs(live: float, D: int, H: int) -> float:
    dh = D // H
    return live * D + 3 * live * D * D + 3 * live * D + H * (live * live * dh + live * live + live * live * dh) + live * D * (D + 1)


def block_macs(live_tokens: float, total_tokens: int, D: int, M: int, H: int) -> float:
    """One pre-LN block; `live_tokens` = rows that are not all-zero at the attention / MLP inputs."""
    ln = 2 * (2 * total_tokens * D)                           # two affine LayerNorms over every row
    return ln + _mha_macs(live_tokens, D, H) + _linear_macs(live_tokens, D, M) + _linear_macs(total_tokens, M, D)


def model_flops(model, seq_per_layer: Optional[Sequence[int]] = None, live_per_layer: Optional[Sequence[float]] = None) -> float:
    """2 * MACs per image for a VisionTransformer / RankVisionTransformer / ResidualVisionTransformer.

    seq_per_layer: tokens entering each block (RankViT shrinks it); live_per_layer: of those, how many are non-zero rows at
    the block's attention / MLP inputs (ResidualViT masks with relu(...) == 0 zero their rows).  Defaults: full sequence."""
    D, M, P = model.hidden_dim, model.mlp_dim, model.patch_size
    blocks = _blocks(model)
    H = blocks[0].self_attention.self_attention.num_heads
    L = len(blocks)
    Np = (model.image_size // P) ** 2
    S = Np + model.num_class_tokens + model.num_registers + (1 if getattr(model, "add_budget_token", False) else 0)
    seqs = list(seq_per_layer) if seq_per_layer is not None else [S] * L
    lives = list(live_per_layer) if live_per_layer is not None else seqs
    macs = Np * D * (3 * P * P) + Np * D                                   # conv_proj (+ bias)
    for s, lv in zip(seqs, lives):
        macs += block_macs(lv, s, D, M, H)
    macs += 2 * seqs[-1] * D                                               # encoder.ln
    macs += _linear_macs(1, D, model.num_classes)                          # head on the pooled class token
    if hasattr(blocks[0], "residual_gate"):
        for s in seqs:                                                     # gate projection D->1 on image tokens (+ budget gate)
            macs += _linear_macs(s - 2, D, 1) + _linear_macs(1, D, 1)
    return 2.0 * macs


@torch.no_grad()
def measured_flops(model, x: torch.Tensor):
    """Run one forward and account for what it actually did: RankViT sequence lengths and ResidualViT zero rows are read
    from the blocks after the pass (block.last_keep / block.mask).  Returns (flops_per_image, avg_sparsity)."""
    out = model(x)
    B = out.shape[0]
    seqs, lives = [], []
    S = (model.image_size // model.patch_size) ** 2 + model.num_class_tokens + model.num_registers
    S += 1 if getattr(model, "add_budget_token", False) else 0
    sparsity: List[float] = []
    for blk in _blocks(model):
        keep = getattr(blk, "last_keep", None)
        if keep is not None and getattr(blk, "current_budget", 1) != 1:
            S = 1 + keep.shape[1]
        seqs.append(S)
        mask = getattr(blk, "mask", None)
        if mask is not None:
            zero = float((mask == 0).sum().item()) / B                     # zero rows per image
            lives.append(S - zero)
            sparsity.append(zero / S)
        else:
            lives.append(S)
    return model_flops(model, seqs, lives), (sum(sparsity) / len(sparsity) if sparsity else 0.0)


@torch.no_grad()
def hook_macs(model, x: torch.Tensor):
    """MACs the reference's two custom hooks (utils/flops_count.py:27-39 Linear, :45-145 MultiheadAttention) would accumulate in
    `__flops__` over ONE forward of the whole batch `x`, per module (reference module names) and in total: integers, bit-equal to
    the reference's own counts (tests/golden/flops_hooks.json).  Runs the forward to read RankViT's per-layer sequence lengths and
    ResidualViT's masks off the blocks."""
    out = model(x)
    B = int(out.shape[0])
    D, M = model.hidden_dim, model.mlp_dim
    H = _blocks(model)[0].self_attention.self_attention.num_heads
    dh = D // H
    S = (model.image_size // model.patch_size) ** 2 + model.num_class_tokens + model.num_registers
    S += 1 if getattr(model, "add_budget_token", False) else 0
    per = {}
    for i, blk in enumerate(model.encoder.layers):
        if not hasattr(blk, "self_attention"):
            continue
        keep = getattr(blk, "last_keep", None)
        if keep is not None and getattr(blk, "current_budget", 1) != 1:
            S = 1 + int(keep.shape[1])
        mask = getattr(blk, "mask", None)
        zeros = [0] * B if mask is None else [int(z) for z in (mask.reshape(B, -1) == 0).sum(-1).tolist()]
        mha = 0
        for z in zeros:                                                    # per sequence: the attention term is quadratic in the live length
            live = S - z
            mha += live * D + 3 * live * D * D + 3 * live * D + H * (live * live * dh + live * live + live * live * dh) + live * D * (D + 1)
        p = f"encoder.layers.{i}."
        per[p + "self_attention.self_attention"] = mha
        per[p + "mlp.fc1"] = (D * M + M) * (B * S - sum(zeros))
        per[p + "mlp.fc2"] = (M * D + D) * (B * S)
        if hasattr(blk, "residual_gate") and mask is not None:
            per[p + "residual_gate.projection"] = (D + 1) * B * int(mask.shape[1])
            if getattr(blk, "budget_token_gate", None) is not None:
                per[p + "budget_token_gate"] = (D + 1) * B
    per["head"] = (D * model.num_classes + model.num_classes) * B
    mha_total = sum(v for k, v in per.items() if k.endswith("self_attention.self_attention"))
    return {"per_module_macs": per, "total_macs": {"mha": mha_total, "linear": sum(per.values()) - mha_total}}
