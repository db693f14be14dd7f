"""Host-side executor of the MI355X forward path: sequences the C-ABI kernels for one encoder block,
the patch embedding and the classification head.  Pure plumbing: buffers come from PyTorch's caching
allocator, launches go to torch's current stream, arithmetic happens in libpeekvit_hip.so only.

HBM data layout:
  residual stream   fp32 [B, S, D]        (token-major, one 4*D-byte row per token)
  GEMM operands     bf16 [B*S, D|M]       (LayerNorm / attention / GELU outputs, K-contiguous)
  packed q|k|v      bf16 [B, S, 3D]       (nn.MultiheadAttention in_proj order; heads are 2*dh-byte column slices)
  weights           bf16 [N, K]           (nn.Linear (out,in) layout, cast once per parameter version)
"""
from __future__ import annotations

import math
import os
import weakref
from typing import Dict, Optional, Tuple

import torch
from torch import nn

from . import _lib, ops
import contextlib

from ._lib import (PV_EPI_BIAS_BF16, PV_EPI_BIAS_F32, PV_EPI_BIAS_GELU_BF16, PV_EPI_BIAS_GELU_SPLIT_BF16, PV_EPI_BIAS_POS_F32,
                   PV_EPI_BIAS_RES_F32, PeekvitHipError)

# Operand precision of the MFMA products:
#   "bf16"   (default) bf16 operands, fp32 accumulate: 4e-3 relative logits error vs the fp32 reference at random init
#   "f16"    IEEE fp16 operands (libpeekvit_hip_f16.so: same kernels, same MFMA rate, 2^-11 instead of 2^-8 operand rounding),
#            fp32 accumulate: 5e-4 relative logits error - meets BASELINE's 1e-3 at the speed of "bf16"; operand range 6e-5..65504
#   "bf16x3" every GEMM operand split v = hi + lo and concatenated along K ([a_hi|a_lo|a_hi] . [w_hi|w_hi|w_lo]^T on the same
#            MFMA kernel), exact-fp32 attention: meets BASELINE's 1e-3 (measured ~1e-5) at ~3x the GEMM work
_PRECISION = os.environ.get("PEEKVIT_AMD_PRECISION", "bf16")
_MODES = ("bf16", "f16", "bf16x3")
if _PRECISION not in _MODES:
    raise ValueError(f"PEEKVIT_AMD_PRECISION={_PRECISION!r}: expected one of {_MODES}")
_lib.OPERAND = "f16" if _PRECISION == "f16" else "bf16"


@contextlib.contextmanager
def precision(mode: str):
    global _PRECISION
    if mode not in _MODES:
        raise ValueError(f"unknown precision {mode!r}")
    old, _PRECISION = _PRECISION, mode
    old_op, _lib.OPERAND = _lib.OPERAND, ("f16" if mode == "f16" else "bf16")
    try:
        yield
    finally:
        _PRECISION, _lib.OPERAND = old, old_op


def backend_for(x: torch.Tensor, module: nn.Module, dropout_p: float = 0.0) -> str:
    """'hip' for GPU tensors outside autograd, else 'torch' (stock-op composite used on CPU tensors and - for the
    modules train_engine does not cover yet: RankViT / ResidualViT blocks - while autograd is recording).
    PEEKVIT_AMD_BACKEND=hip makes every non-eligible call raise instead of taking the composite path."""
    forced = os.environ.get("PEEKVIT_AMD_BACKEND", "")
    eligible = x.is_cuda and not torch.is_grad_enabled() and not (module.training and dropout_p > 0.0)
    if forced == "torch":
        return "torch"
    if forced == "hip" and not eligible:
        raise PeekvitHipError("PEEKVIT_AMD_BACKEND=hip but the call is not eligible for the HIP path "
                              "(needs a GPU tensor, torch.no_grad(), and inactive dropout)")
    return "hip" if eligible else "torch"


# ------------------------------------------------------------------------------------------------
# workspace arena: named scratch buffers per device, grown on demand, reused across blocks/calls
# ------------------------------------------------------------------------------------------------
class _Workspace:
    def __init__(self):
        self._bufs: Dict[Tuple[str, torch.device], torch.Tensor] = {}

    def get(self, name: str, shape, dtype, device) -> torch.Tensor:
        n = 1
        for s in shape:
            n *= int(s)
        nbytes = n * torch.empty((), dtype=dtype).element_size()
        key = (name, device)
        buf = self._bufs.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=device)
            self._bufs[key] = buf
        return buf[:nbytes].view(dtype).view(*shape)

    def clear(self):
        self._bufs.clear()


workspace = _Workspace()

# ------------------------------------------------------------------------------------------------
# bf16 weight cache: one cast per (parameter storage, version)
# ------------------------------------------------------------------------------------------------
_wcache: Dict[int, Tuple["weakref.ref", int, int, torch.Tensor]] = {}


def bf16_weight(p: torch.Tensor) -> torch.Tensor:
    """bf16 copy of a 2-D (or conv 4-D, viewed [out, -1]) fp32 parameter, refreshed when it changes."""
    key = (id(p), _lib.OPERAND)
    ent = _wcache.get(key)
    if ent is not None and ent[0]() is p and ent[1] == p._version and ent[2] == p.data_ptr():
        return ent[3]
    src = p.detach()
    if not src.is_contiguous():
        src = src.contiguous()
    w = ops.cast_bf16(src.view(src.shape[0], -1))
    _wcache[key] = (weakref.ref(p, lambda _r, k=key: _wcache.pop(k, None)), p._version, p.data_ptr(), w)
    return w


# ToTensor + Normalize constants of the reference's datasets (data/imagenette.py:73, data/imagenet.py)
IMAGENET_MEAN, IMAGENET_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


_w3cache: Dict[int, Tuple["weakref.ref", int, int, torch.Tensor]] = {}


def bf16x3_weight(p: torch.Tensor) -> torch.Tensor:
    """[w_hi | w_hi | w_lo] along K (bf16 [N, 3K]) of an fp32 parameter, refreshed when it changes."""
    key = id(p)
    ent = _w3cache.get(key)
    if ent is not None and ent[0]() is p and ent[1] == p._version and ent[2] == p.data_ptr():
        return ent[3]
    src = p.detach()
    src = (src if src.is_contiguous() else src.contiguous()).view(src.shape[0], -1)
    w = ops.split3(src, 1)
    _w3cache[key] = (weakref.ref(p, lambda _r, k=key: _w3cache.pop(k, None)), p._version, p.data_ptr(), w)
    return w


def _f32(p: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    if p is None:
        return None
    d = p.detach()
    return d if d.is_contiguous() else d.contiguous()


# ------------------------------------------------------------------------------------------------
# one pre-LN encoder block (reference models/vit.py:45-55; masked form models/residualvit.py:249-260)
# ------------------------------------------------------------------------------------------------
# Fused LayerNorm (row-block GEMM, pv_gemm_args.ln_out) is OPT-IN: measured on MI355X at B = 2048 it is bit-identical but
# slower than GEMM + standalone LN (out-proj 1.34 vs 1.14 ms, fc2 2.81 vs 2.02 ms): one workgroup per 256-row block runs its
# three column tiles one after the other, so the A panel is re-fetched through the fabric three times instead of being
# shared in time by neighbouring CUs.
_FUSE_LN = os.environ.get("PEEKVIT_AMD_FUSE_LN", "0") == "1"


def _ln_fusable(D: int, K: int) -> bool:
    """Shapes the row-block GEMM with fused LayerNorm accepts (include/peekvit_hip.h pv_gemm_args.ln_out)."""
    return _FUSE_LN and D % 256 == 0 and D <= 4096 and K % 128 == 0


def _ln_key(ln: nn.LayerNorm):
    return (id(ln.weight), ln.weight._version, ln.bias._version, float(ln.eps))


# LayerNorm FOLDING (opt-in, PEEKVIT_AMD_FOLD_LN=1): no LayerNorm pass at all - the residual GEMMs also emit the
# 16-bit copy of their rows + per-tile row statistics, and the in-proj / fc1 GEMMs run on that raw copy with gamma (.) W and correct
# in their epilogue: rstd * (acc - mean * c1) + c2.  Same math as LayerNorm -> Linear, but the operand that is rounded to 16 bits is
# the raw row instead of the normalised one: +20 % logits error (5.6e-3 bf16, 6.8e-4 f16), hence opt-in.
_FOLD_LN = os.environ.get("PEEKVIT_AMD_FOLD_LN", "0") == "1"
_foldcache: Dict[Tuple[int, int, str], tuple] = {}


def _fold_weights(w: torch.Tensor, b: Optional[torch.Tensor], ln: nn.LayerNorm):
    """(W' = operand(gamma (.) W) [N,D], c1 = sum_k W'[n,k], c2 = W beta + b) cached per (weight, LayerNorm, operand type) version."""
    key = (id(w), id(ln.weight), _lib.OPERAND)
    ver = (w._version, b._version if b is not None else -1, ln.weight._version, ln.bias._version, w.data_ptr())
    ent = _foldcache.get(key)
    if ent is not None and ent[0] == ver:
        return ent[1]
    with torch.no_grad():
        wf = w.detach().float()
        wg = ops.cast_bf16((wf * ln.weight.detach().float()).contiguous())
        c1 = wg.float().sum(1).contiguous()
        c2 = (wf @ ln.bias.detach().float() + (b.detach().float() if b is not None else 0.0)).contiguous()
    _foldcache[key] = (ver, (wg, c1, c2))
    return wg, c1, c2


def _fold_ok(R: int, D: int, M: int) -> bool:
    """Folding needs the 256-row tile kernel for all four token GEMMs (include/peekvit_hip.h): enough rows, 128-multiples."""
    return _FOLD_LN and _PRECISION in ("bf16", "f16") and R >= 2048 and D % 128 == 0 and M % 128 == 0


def block_forward(blk: nn.Module, x: torch.Tensor, eps: float, row_scale: Optional[torch.Tensor] = None,
                  next_ln: Optional[nn.LayerNorm] = None) -> torch.Tensor:
    """x: fp32 [B,S,D] contiguous on the GPU.  Returns a NEW fp32 [B,S,D] tensor.

    Launches: [LN1] -> QKV GEMM (+bias, q*dh^-0.5) -> attention -> out-proj GEMM (+bias, +residual, fused LN2)
              -> fc1 GEMM (+bias, GELU) -> fc2 GEMM (+bias, +residual, fused next-block LN1).
    LN1 is skipped when the producer of `x` (the previous block's fc2) already emitted it: that hand-off travels as the
    private attribute `x._pv_ln = (h_bf16, key)` and is used only if `key` matches this block's ln_1 (same parameter
    object, versions and eps).  `next_ln`: the LayerNorm the consumer of the output will apply first (encoder hint).
    row_scale [B,S] (ResidualViT fwd_mask) multiplies LN1 out, the attention branch and LN2 out.
    """
    if _PRECISION == "bf16x3":
        return _block_forward_x3(blk, x, eps, row_scale)
    if x.dtype != torch.float32:
        x = x.float()
    handoff = getattr(x, "_pv_ln", None)
    if not x.is_contiguous():
        x, handoff = x.contiguous(), None
    B, S, D = x.shape
    mha = blk.self_attention.self_attention
    H = mha.num_heads
    dh = D // H
    M = blk.mlp.fc1.out_features
    dev = x.device
    R = B * S

    h2 = workspace.get("h2", (R, D), _lib.operand_dtype(), dev)
    qkv = workspace.get("qkv", (R, 3 * D), _lib.operand_dtype(), dev)
    att = workspace.get("att", (R, D), _lib.operand_dtype(), dev)
    g = workspace.get("g", (R, M), _lib.operand_dtype(), dev)
    x1 = workspace.get("x1", (B, S, D), torch.float32, dev)
    out = torch.empty_like(x)

    if row_scale is None and _fold_ok(R, D, M):
        # ---- LayerNorm folded into the GEMMs: no LayerNorm launch except for a block whose input has no producer hand-off ----
        nt = (D + 255) // 256
        fold_in = getattr(x, "_pv_fold", None)
        if fold_in is not None and fold_in[2] == _ln_key(blk.ln_1) and fold_in[0].shape == (R, D):
            wg, c1, c2 = _fold_weights(mha.in_proj_weight, mha.in_proj_bias, blk.ln_1)
            stat = ops.rowstat_finalize(fold_in[1], D, blk.ln_1.eps, workspace.get("fold_stat", (R, 2), torch.float32, dev))
            ops.gemm(fold_in[0], wg, None, qkv, PV_EPI_BIAS_BF16, M=R, qcols=D, qscale=float(dh) ** -0.5, fold=(stat, c1, c2))
        else:
            h = workspace.get("h", (R, D), _lib.operand_dtype(), dev)
            ops.layernorm_bf16(x, _f32(blk.ln_1.weight), _f32(blk.ln_1.bias), eps, h, None)
            ops.gemm(h, bf16_weight(mha.in_proj_weight), _f32(mha.in_proj_bias), qkv, PV_EPI_BIAS_BF16, M=R, qcols=D, qscale=float(dh) ** -0.5)
        ops.attention(qkv, att, B, S, H, dh)
        x16 = workspace.get("fold_x16", (R, D), _lib.operand_dtype(), dev)
        part = workspace.get("fold_part", (nt, R, 2), torch.float32, dev)
        ops.gemm(att, bf16_weight(mha.out_proj.weight), _f32(mha.out_proj.bias), x1.view(R, D), PV_EPI_BIAS_RES_F32, M=R, res=x.view(R, D),
                 x16_out=x16, rowstat_out=part)
        wg, c1, c2 = _fold_weights(blk.mlp.fc1.weight, blk.mlp.fc1.bias, blk.ln_2)
        stat = ops.rowstat_finalize(part, D, blk.ln_2.eps, workspace.get("fold_stat", (R, 2), torch.float32, dev))
        ops.gemm(x16, wg, None, g, PV_EPI_BIAS_GELU_BF16, M=R, fold=(stat, c1, c2))
        emit = next_ln is not None and next_ln.normalized_shape == (D,)
        o16 = workspace.get("fold_o16", (R, D), _lib.operand_dtype(), dev) if emit else None
        opart = workspace.get("fold_opart", (nt, R, 2), torch.float32, dev) if emit else None
        ops.gemm(g, bf16_weight(blk.mlp.fc2.weight), _f32(blk.mlp.fc2.bias), out.view(R, D), PV_EPI_BIAS_RES_F32, M=R, res=x1.view(R, D),
                 x16_out=o16, rowstat_out=opart)
        if emit:
            out._pv_fold = (o16, opart, _ln_key(next_ln))
        return out

    if handoff is not None and row_scale is None and handoff[1] == _ln_key(blk.ln_1) and handoff[0].shape == (R, D):
        h = handoff[0]                                   # LN1(x), emitted by the producer's fused epilogue
    else:
        h = workspace.get("h", (R, D), _lib.operand_dtype(), dev)
        ops.layernorm_bf16(x, _f32(blk.ln_1.weight), _f32(blk.ln_1.bias), eps, h, row_scale)
    ops.gemm(h, bf16_weight(mha.in_proj_weight), _f32(mha.in_proj_bias), qkv, PV_EPI_BIAS_BF16, M=R,
             qcols=D, qscale=float(dh) ** -0.5)
    ops.attention(qkv, att, B, S, H, dh)
    fuse2 = _ln_fusable(D, D)
    ops.gemm(att, bf16_weight(mha.out_proj.weight), _f32(mha.out_proj.bias), x1.view(R, D), PV_EPI_BIAS_RES_F32, M=R,
             res=x.view(R, D), row_scale=row_scale,
             ln=(_f32(blk.ln_2.weight), _f32(blk.ln_2.bias), eps, h2, row_scale) if fuse2 else None)
    if not fuse2:
        ops.layernorm_bf16(x1, _f32(blk.ln_2.weight), _f32(blk.ln_2.bias), eps, h2, row_scale)
    ops.gemm(h2, bf16_weight(blk.mlp.fc1.weight), _f32(blk.mlp.fc1.bias), g, PV_EPI_BIAS_GELU_BF16, M=R)
    fuse_next = next_ln is not None and _ln_fusable(D, M) and next_ln.normalized_shape == (D,)
    hn = workspace.get("h", (R, D), _lib.operand_dtype(), dev) if fuse_next else None      # "h" is dead once QKV has consumed it
    ops.gemm(g, bf16_weight(blk.mlp.fc2.weight), _f32(blk.mlp.fc2.bias), out.view(R, D), PV_EPI_BIAS_RES_F32, M=R,
             res=x1.view(R, D),
             ln=(_f32(next_ln.weight), _f32(next_ln.bias), next_ln.eps, hn, None) if fuse_next else None)
    if fuse_next:
        out._pv_ln = (hn, _ln_key(next_ln))
    return out


def _block_forward_x3(blk: nn.Module, x: torch.Tensor, eps: float, row_scale: Optional[torch.Tensor]) -> torch.Tensor:
    """The same block in precision mode "bf16x3": split LN outputs / GELU outputs / attention outputs, split weights,
    fp32 q|k|v and exact-fp32 attention.  Residual stream, LayerNorm, softmax, GELU are fp32 as in the default mode."""
    x = x.float() if x.dtype != torch.float32 else x
    x = x if x.is_contiguous() else x.contiguous()
    B, S, D = x.shape
    mha = blk.self_attention.self_attention
    H = mha.num_heads
    dh = D // H
    M = blk.mlp.fc1.out_features
    dev, R = x.device, B * S
    h3 = workspace.get("h3", (R, 3 * D), _lib.operand_dtype(), dev)
    qkv32 = workspace.get("qkv32", (R, 3 * D), torch.float32, dev)
    att3 = workspace.get("att3", (R, 3 * D), _lib.operand_dtype(), dev)
    g3 = workspace.get("g3", (R, 3 * M), _lib.operand_dtype(), dev)
    x1 = workspace.get("x1", (B, S, D), torch.float32, dev)
    out = torch.empty_like(x)
    ops.layernorm_split(x, _f32(blk.ln_1.weight), _f32(blk.ln_1.bias), eps, h3, row_scale)
    ops.gemm(h3, bf16x3_weight(mha.in_proj_weight), _f32(mha.in_proj_bias), qkv32, PV_EPI_BIAS_F32, M=R, qcols=D, qscale=float(dh) ** -0.5)
    ops.attention_f32(qkv32, att3, B, S, H, dh)
    ops.gemm(att3, bf16x3_weight(mha.out_proj.weight), _f32(mha.out_proj.bias), x1.view(R, D), PV_EPI_BIAS_RES_F32, M=R,
             res=x.view(R, D), row_scale=row_scale)
    ops.layernorm_split(x1, _f32(blk.ln_2.weight), _f32(blk.ln_2.bias), eps, h3, row_scale)
    ops.gemm(h3, bf16x3_weight(blk.mlp.fc1.weight), _f32(blk.mlp.fc1.bias), g3, PV_EPI_BIAS_GELU_SPLIT_BF16, M=R)
    ops.gemm(g3, bf16x3_weight(blk.mlp.fc2.weight), _f32(blk.mlp.fc2.bias), out.view(R, D), PV_EPI_BIAS_RES_F32, M=R, res=x1.view(R, D))
    return out


def run_layers(layers: nn.Sequential, x: torch.Tensor) -> torch.Tensor:
    """Run an encoder's `layers` on the MI355X path, telling every block which LayerNorm its consumer applies first so the
    producer can fuse it (peephole over ADJACENT blocks only: `layers` stays an ordinary nn.Sequential).
    A consumer that transforms its input before ln_1 (RankViT block with an active budget, ResidualViT gated block,
    NoiseBlock, ...) gets no hint and simply normalises itself."""
    mods = list(layers)
    for i, layer in enumerate(mods):
        nxt = mods[i + 1] if i + 1 < len(mods) else None
        hint = None
        if nxt is not None and getattr(nxt, "_pv_plain_ln1", None) is not None and nxt._pv_plain_ln1():
            hint = nxt.ln_1
        if hasattr(layer, "_pv_next_ln"):
            # a PLAIN attribute (nn.Module.__setattr__ would register the neighbour's LayerNorm as a submodule of this block and
            # leak `layers.{i}._pv_next_ln.*` into named_parameters() / state_dict())
            object.__setattr__(layer, "_pv_next_ln", hint)
        x = layer(x)
    return x


# ------------------------------------------------------------------------------------------------
# patch embedding + special tokens + positional embedding (reference models/vit.py:203-236, :92)
# ------------------------------------------------------------------------------------------------
def embed_tokens(model: nn.Module, img: torch.Tensor, budget_token: Optional[torch.Tensor] = None,
                 budget: float = 0.0) -> torch.Tensor:
    """img fp32 [B,3,R,R] -> tokens fp32 [B,S_total,D] = [cls | registers | patches] + pos_embedding
    (+ one trailing budget-token row without positional embedding for ResidualViT)."""
    u8 = img.dtype == torch.uint8            # raw NHWC image: normalisation is fused into the patch gather
    if not u8 and img.dtype != torch.float32:
        img = img.float()
    if not img.is_contiguous():
        img = img.contiguous()
    B = img.shape[0]
    P, D = model.patch_size, model.hidden_dim
    Hh, Ww, Cin = (img.shape[1], img.shape[2], img.shape[3]) if u8 else (img.shape[2], img.shape[3], img.shape[1])
    Np = (Hh // P) * (Ww // P)
    n_special = model.num_class_tokens + model.num_registers
    S = n_special + Np + (1 if budget_token is not None else 0)
    K = Cin * P * P
    dev = img.device

    x3 = _PRECISION == "bf16x3" and not u8
    cols = workspace.get("cols", (B * Np, 3 * K if x3 else K), _lib.operand_dtype(), dev)
    if x3:
        ops.im2col_split(img, P, cols)
    elif u8:
        mean, std = getattr(model, "input_mean", IMAGENET_MEAN), getattr(model, "input_std", IMAGENET_STD)
        ops.im2col_u8(img, P, cols, mean, std)
    else:
        ops.im2col(img, P, cols)
    tokens = torch.empty((B, S, D), dtype=torch.float32, device=dev)
    pos = _f32(model.encoder.pos_embedding).view(-1, D)
    ops.gemm(cols, (bf16x3_weight if x3 else bf16_weight)(model.conv_proj.weight), _f32(model.conv_proj.bias), tokens.view(B * S, D),
             PV_EPI_BIAS_POS_F32, M=B * Np, pos=pos, rows_per_img_in=Np, rows_per_img_out=S, row_off=n_special)
    special = _f32(model.class_tokens).view(-1, D)
    if model.num_registers > 0:
        special = torch.cat([special, _f32(model.register_tokens).view(-1, D)], dim=0)
    ops.token_prologue(tokens, special, pos, _f32(budget_token), budget, n_special)
    return tokens


def pool_and_head(model: nn.Module, tokens: torch.Tensor) -> torch.Tensor:
    """Final LayerNorm on the class-token rows, sum over class tokens, fp32 head (models/vit.py:95,242-246)."""
    ln = model.encoder.ln
    pooled = ops.cls_pool(tokens, _f32(ln.weight), _f32(ln.bias), ln.eps, model.num_class_tokens)
    return ops.head(pooled, _f32(model.head.weight), _f32(model.head.bias))


def sort_and_drop(x: torch.Tensor, budget: float):
    """RankViT token ranking + compaction (models/rankvit.py:55-77).  Returns (tokens [B,1+k,D], keep int32 [B,k])."""
    if not x.is_contiguous():
        x = x.contiguous()
    N = x.shape[1] - 1
    k = math.ceil(N * budget)
    norms = ops.token_norm(x)
    keep = ops.rank_topk(norms, k)
    return ops.gather_tokens(x, keep), keep
