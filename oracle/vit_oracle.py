"""ORACLE (test infrastructure, NOT product code) - CPU restatement of peekvit's ViT encoder hot path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
the product package `peekvit_amd` never does (it fails loudly when its HIP library is missing).

Every function restates, in plain functional PyTorch-CPU fp32, the arithmetic that the reference
delegates to stock torch.nn modules, citing the reference file:line it
follows (paths relative to the reference checkout root).  Parity is PINNED: `oracle/make_golden.py`
imports the real reference in the build container and stores its outputs under tests/golden/;
tests/test_oracle_golden.py checks this restatement against those vectors bit-for-bit (fp32 mode).

Two arithmetic modes:
  * mode="fp32"  - the reference's own arithmetic (what the golden vectors hold).
  * mode="bf16"  - the SAME algorithm with the operand roundings of the MI355X path inserted
                   ("same-rounding-points" restatement): GEMM/attention
                   operands rounded to bf16, fp32 accumulation, fp32 residual stream, fp32
                   LayerNorm/softmax/GELU.  Used for tight op-level and end-to-end GPU parity.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Union

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


def rb(x: Tensor, mode: str) -> Tensor:
    """Round to the 16-bit operand type of the mode (bf16 or IEEE fp16, nearest-even) and return fp32 - identity in fp32 mode."""
    if mode == "bf16":
        return x.to(torch.bfloat16).to(torch.float32)
    if mode == "f16":
        return x.to(torch.float16).to(torch.float32)
    return x


def _t(sd: Dict[str, object], key: str) -> Tensor:
    v = sd[key]
    if not isinstance(v, torch.Tensor):
        v = torch.from_numpy(v)
    if v.requires_grad:                 # training-step checks (tests): leaf parameters stay attached to autograd
        return v
    return v.detach().to(torch.float32).cpu()


# ------------------------------------------------------------------------------------------------
# op-level restatements
# ------------------------------------------------------------------------------------------------
def patch_embed(x: Tensor, w: Tensor, b: Tensor, patch: int, mode: str = "fp32") -> Tensor:
    """models/vit.py:203-222 `_process_input`: stride-P conv -> [B,D,Np] -> permute -> [B,Np,D]."""
    n = x.shape[0]
    if mode != "fp32":
        # im2col + GEMM with bf16 operands, fp32 accumulate; K order (c, kh, kw) = conv weight layout
        cols = F.unfold(rb(x, mode), kernel_size=patch, stride=patch)          # [B, 3PP, Np]
        t = cols.transpose(1, 2) @ rb(w, mode).reshape(w.shape[0], -1).t() + b
        return t
    t = F.conv2d(x, w, b, stride=patch)                                        # vit.py:212
    t = t.reshape(n, w.shape[0], -1)                                           # vit.py:214
    return t.permute(0, 2, 1)                                                  # vit.py:220


def layer_norm(x: Tensor, w: Tensor, b: Tensor, eps: float) -> Tensor:
    """nn.LayerNorm(hidden_dim) instances: vit.py:37,42,88; residualvit.py:117,122 (eps 1e-6)."""
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def linear(x: Tensor, w: Tensor, b: Optional[Tensor], mode: str = "fp32") -> Tensor:
    """nn.Linear call sites blocks.py:77-78, vit.py:186; bf16 mode rounds both operands."""
    return F.linear(rb(x, mode), rb(w, mode), b)


def attention_core(q: Tensor, k: Tensor, v: Tensor, mode: str = "fp32") -> Tensor:
    """softmax(q k^T) v per head, q already scaled. q,k,v: [B,H,S,d].

    torch/nn/functional.py multi_head_attention_forward (need_weights branch 6576-6594):
    q_scaled = q * sqrt(1/d); attn = softmax(bmm(q_scaled, k^T)); out = bmm(attn, v).
    bf16 mode = the flash-style order of the HIP kernel: P = exp(s - max) rounded to bf16 feeds
    P.V (fp32 acc), normalised afterwards by the fp32 row sum of the UNROUNDED exponentials.
    """
    s = q @ k.transpose(-1, -2)
    if mode != "fp32":
        m = s.max(dim=-1, keepdim=True).values
        p = torch.exp(s - m)
        l = p.sum(dim=-1, keepdim=True)
        return (rb(p, mode) @ v) / l
    return torch.softmax(s, dim=-1) @ v


def mha(x: Tensor, in_w: Tensor, in_b: Tensor, out_w: Tensor, out_b: Tensor, num_heads: int,
        mode: str = "fp32") -> Tensor:
    """models/blocks.py:88-95 SelfAttention -> nn.MultiheadAttention(batch_first=True)(x,x,x).

    Packed in-proj (3D,D)+bias, q scaled by d**-0.5, per-head softmax(q k^T) v, out-proj.  The
    head-averaged weights that need_weights=True also returns are discarded (blocks.py:94-95).
    """
    B, S, D = x.shape
    d = D // num_heads
    qkv = linear(x, in_w, in_b, mode)                                          # [B,S,3D]
    q, k, v = qkv.split(D, dim=-1)
    q = q * (float(d) ** -0.5)
    q, k, v = (rb(t, mode).reshape(B, S, num_heads, d).transpose(1, 2) for t in (q, k, v))
    o = attention_core(q, k, v, mode)                                          # [B,H,S,d]
    o = o.transpose(1, 2).reshape(B, S, D)
    return linear(o, out_w, out_b, mode)


def mlp(x: Tensor, w1: Tensor, b1: Tensor, w2: Tensor, b2: Tensor, mode: str = "fp32") -> Tensor:
    """models/blocks.py:80-84: fc2(gelu_erf(fc1(x)))."""
    return linear(F.gelu(linear(x, w1, b1, mode)), w2, b2, mode)


def vit_block(x: Tensor, sd: Dict[str, object], prefix: str, num_heads: int, eps: float = 1e-5,
              mode: str = "fp32", mask: Optional[Tensor] = None) -> Tensor:
    """models/vit.py:45-55 ViTBlock.forward; with `mask` = residualvit.py:249-260 plain_forward."""
    g = lambda k: _t(sd, prefix + k)
    h = layer_norm(x, g("ln_1.weight"), g("ln_1.bias"), eps)
    if mask is not None:
        h = mask * h                                                           # residualvit.py:252
    a = mha(h, g("self_attention.self_attention.in_proj_weight"), g("self_attention.self_attention.in_proj_bias"),
            g("self_attention.self_attention.out_proj.weight"), g("self_attention.self_attention.out_proj.bias"),
            num_heads, mode)
    if mask is not None:
        a = mask * a                                                           # residualvit.py:254
    x = a + x                                                                  # vit.py:51 `x + input`
    y = layer_norm(x, g("ln_2.weight"), g("ln_2.bias"), eps)
    if mask is not None:
        y = mask * y                                                           # residualvit.py:258
    y = mlp(y, g("mlp.fc1.weight"), g("mlp.fc1.bias"), g("mlp.fc2.weight"), g("mlp.fc2.bias"), mode)
    return x + y                                                               # vit.py:55


def token_norms(tokens: Tensor) -> Tensor:
    """models/rankvit.py:63 torch.norm(input, dim=-1) on the non-CLS tokens: [B,N,D] -> [B,N]."""
    return torch.norm(tokens, dim=-1)


def rank_indices(norms: Tensor) -> Tensor:
    """models/rankvit.py:67 argsort(descending).  The reference sort is unstable; ties are
    reference-undefined.  The build defines ties as LOWEST INDEX FIRST,
    i.e. a stable descending sort, which is what this restatement and the HIP kernel implement."""
    return torch.argsort(norms, dim=-1, descending=True, stable=True)


def sort_and_drop(x: Tensor, budget: float, indices: Optional[Tensor] = None):
    """models/rankvit.py:55-77: rank non-CLS tokens by L2 norm, keep the first ceil(N*budget) in
    descending-norm order, prepend the class token.  Returns (out [B,1+k,D], kept indices [B,k])."""
    cls, tok = x[:, 0:1, :], x[:, 1:, :]                                       # rankvit.py:58-59
    if indices is None:
        indices = rank_indices(token_norms(tok))
    k = math.ceil(tok.shape[1] * budget)                                       # rankvit.py:74
    keep = indices[:, :k]
    out = torch.gather(tok, 1, keep.unsqueeze(-1).expand(-1, -1, tok.shape[-1]))  # rankvit.py:71,75
    return torch.cat([cls, out], dim=1), keep                                  # rankvit.py:77


def residual_gate(tokens: Tensor, w: Tensor, b: Tensor, temp: float, sigmoid_bias: float,
                  threshold: Union[float, Tensor]) -> Tensor:
    """models/residualvit.py:47-74 with gate_type='sigmoid' (blocks.py:62-69 SigmoidWithTemp):
    mask = relu(sigmoid((x.w + b)/temp + sigmoid_bias) - threshold), [B,N,D] -> [B,N,1].
    The gate's D->1 projection stays fp32 in every mode (it is a GEMV, not an MFMA GEMM)."""
    logit = F.linear(tokens, w, b)                                             # residualvit.py:54
    mask = torch.sigmoid(logit / temp + sigmoid_bias)                          # blocks.py:69
    return F.relu(mask - threshold)                                            # residualvit.py:62-69


def residual_block(x: Tensor, sd: Dict[str, object], prefix: str, num_heads: int, temp: float,
                   gate_bias: float, mode: str = "fp32", budget_token: str = "learnable",
                   gate_threshold: float = 0.5):
    """models/residualvit.py:197-244 forward_skip_attention_mlp (eval, sigmoid gate) for the two
    budget modes the in-scope configs can reach: a 'learnable' budget token (every residual*.yaml)
    and no budget token... the latter is a reference bug (shape mismatch), so
    only 'learnable' is restated.  Returns (block output, mask [B,N,1])."""
    assert budget_token == "learnable"
    g = lambda k: _t(sd, prefix + k)
    special, img = x[:, :1, :], x[:, 1:, :]                                    # residualvit.py:200-201
    btok, img = img[:, -1:, :], img[:, :-1, :]                                 # residualvit.py:206-207
    thr = torch.sigmoid(F.linear(btok, g("budget_token_gate.weight"), g("budget_token_gate.bias")))  # :212
    mask = residual_gate(img, g("residual_gate.projection.weight"), g("residual_gate.projection.bias"),
                         temp, gate_bias, thr)                                 # :217
    masked = torch.cat([special, mask * img, btok], dim=1)                     # :220-227
    ones = torch.ones(mask.shape[0], 1, 1)
    fwd_mask = torch.cat([ones, mask, ones], dim=1)                            # :230-235
    y = vit_block(masked, sd, prefix, num_heads, eps=1e-6, mode=mode, mask=fwd_mask)  # :237, :249-260
    return y, mask


# ------------------------------------------------------------------------------------------------
# whole-model restatements
# ------------------------------------------------------------------------------------------------
def embed_tokens(x: Tensor, sd: Dict[str, object], cfg: dict, mode: str = "fp32") -> Tensor:
    """vit.py:226-236 + vit.py:92: patches, [cls | registers | patches] concat, + pos_embedding."""
    t = patch_embed(x, _t(sd, "conv_proj.weight"), _t(sd, "conv_proj.bias"), cfg["patch_size"], mode)
    n = t.shape[0]
    if cfg.get("num_registers", 0) > 0:
        t = torch.cat([_t(sd, "register_tokens").expand(n, -1, -1), t], dim=1)   # vit.py:230-232
    t = torch.cat([_t(sd, "class_tokens").expand(n, -1, -1), t], dim=1)           # vit.py:235-236
    return t


def vit_forward(x: Tensor, sd: Dict[str, object], cfg: dict, mode: str = "fp32",
                rankvit_layers: Optional[Sequence[int]] = None,
                budget: Union[float, Sequence[float]] = 1.0, trace: Optional[dict] = None) -> Tensor:
    """models/vit.py:224-248 VisionTransformer.forward; with `rankvit_layers` it is
    models/rankvit.py:256-281 (blocks in rankvit_layers run sort_and_drop first when their budget != 1,
    rankvit.py:85-88; budget[i] if list, rankvit.py:287-288)."""
    x = x.to(torch.float32)
    t = embed_tokens(x, sd, cfg, mode)
    t = t + _t(sd, "encoder.pos_embedding")                                    # vit.py:92
    H, L = cfg["num_heads"], cfg["num_layers"]
    if trace is not None:
        trace["tokens"] = t.clone()
        trace["block_cls"], trace["keep"], trace["seq"] = [], {}, []
    for i in range(L):
        if rankvit_layers is not None and i in rankvit_layers:
            b = budget[i] if isinstance(budget, (list, tuple)) else budget
            if b != 1:                                                         # rankvit.py:85
                t, keep = sort_and_drop(t, b)
                if trace is not None:
                    trace["keep"][i] = keep
        if trace is not None:
            trace["seq"].append(t.shape[1])
        t = vit_block(t, sd, f"encoder.layers.{i}.", H, 1e-5, mode)
        if trace is not None:
            trace["block_cls"].append(t[:, 0].clone())
    t = layer_norm(t, _t(sd, "encoder.ln.weight"), _t(sd, "encoder.ln.bias"), 1e-5)   # vit.py:95
    if trace is not None:
        trace["encoder_cls"] = t[:, 0].clone()
    nc = cfg.get("num_class_tokens", 1)
    pooled = t[:, 0:nc].sum(dim=1)                                             # vit.py:242-243 (a SUM)
    return F.linear(pooled, _t(sd, "head.weight"), _t(sd, "head.bias"))        # vit.py:246 (fp32 in all modes)


def residualvit_forward(x: Tensor, sd: Dict[str, object], cfg: dict, budget: float, mode: str = "fp32",
                        trace: Optional[dict] = None) -> Tensor:
    """models/residualvit.py:587-616 forward (eval, add_budget_token='learnable', sigmoid gate,
    all layers 'attention+mlp' - the configuration of configs/model/residualvit_b_16.yaml)."""
    x = x.to(torch.float32)
    t = embed_tokens(x, sd, cfg, mode)
    n = t.shape[0]
    btok = _t(sd, "learnable_budget_token_1").expand(n, -1, -1) * torch.tensor(budget, dtype=torch.float32)  # :566-568
    t = t + _t(sd, "encoder.pos_embedding")                                    # residualvit.py:338-343
    t = torch.cat([t, btok], dim=1)                                            # :345
    H, L = cfg["num_heads"], cfg["num_layers"]
    if trace is not None:
        trace["masks"], trace["block_cls"] = [], []
    for i in range(L):
        t, mask = residual_block(t, sd, f"encoder.layers.{i}.", H, cfg.get("gate_temp", 1.0),
                                 cfg.get("gate_bias", 10.0), mode)
        if trace is not None:
            trace["masks"].append(mask.clone())
            trace["block_cls"].append(t[:, 0].clone())
    t = layer_norm(t, _t(sd, "encoder.ln.weight"), _t(sd, "encoder.ln.bias"), 1e-5)   # residualvit.py:348
    nc = cfg.get("num_class_tokens", 1)
    pooled = t[:, 0:nc].sum(dim=1)                                             # :610-611
    return F.linear(pooled, _t(sd, "head.weight"), _t(sd, "head.bias"))


def rank_seq_lengths(cfg: dict, rankvit_layers: Sequence[int], budget: float) -> List[int]:
    """Per-layer sequence lengths implied by rankvit.py:74 (each ranked layer keeps ceil(N*b) of ITS input)."""
    S = (cfg["image_size"] // cfg["patch_size"]) ** 2 + 1
    out = []
    for i in range(cfg["num_layers"]):
        if i in rankvit_layers and budget != 1:
            S = 1 + math.ceil((S - 1) * budget)
        out.append(S)
    return out
