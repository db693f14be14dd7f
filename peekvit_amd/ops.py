"""Tensor-level wrappers over the C ABI (one function per entry point of include/peekvit_hip.h).

PyTorch is plumbing here: it owns device memory and the stream; every FLOP of the hot path is done by
libpeekvit_hip.so.  All wrappers launch on torch's CURRENT stream and never synchronise.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib
from ._lib import GemmArgs, check

# launch counter: tests assert that the HIP path (not some silent eager path) produced the result
launch_count = 0


class KernelTimer:
    """Per-launch HIP-event timing on the stream the kernels are launched on (bench.py's `roofline` leg).
    `with KernelTimer() as kt:` brackets every C-ABI launch with two events; `kt.summary()` (after a device
    synchronise) returns {kernel: {"launches", "ms", "flops", "bytes"}} where flops/bytes are the ALGORITHMIC
    work of the launches."""

    def __init__(self):
        self.records = []     # (name, start_event, end_event, flops, bytes, member)

    def __enter__(self):
        global _timer
        _timer = self
        return self

    def __exit__(self, *exc):
        global _timer
        _timer = None

    def summary(self):
        out = {}
        for name, e0, e1, flops, nbytes, _member in self.records:
            d = out.setdefault(name, {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
            d["launches"] += 1
            d["ms"] += e0.elapsed_time(e1)
            d["flops"] += flops
            d["bytes"] += nbytes
        return out

    def members(self, name):
        """The launches of one kernel name split by `member` (GEMMs: (N, K, epilogue) - the in-projection, out-projection, fc1, fc2 and the
        patch embedding are one C-ABI entry point but different roofline cases): {member: {"launches", "ms", "flops", "bytes"}}."""
        out = {}
        for n, e0, e1, flops, nbytes, member in self.records:
            if n != name:
                continue
            d = out.setdefault(member, {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
            d["launches"] += 1
            d["ms"] += e0.elapsed_time(e1)
            d["flops"] += flops
            d["bytes"] += nbytes
        return out


_timer: Optional[KernelTimer] = None


class _timed:
    """Context manager used by every wrapper: no-op unless a KernelTimer is active."""
    __slots__ = ("name", "flops", "nbytes", "e0", "dev", "member")

    def __init__(self, name, dev, flops=0.0, nbytes=0.0, member=None):
        self.name, self.flops, self.nbytes, self.dev, self.member = name, flops, nbytes, dev, member

    def __enter__(self):
        if _timer is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record(torch.cuda.current_stream(self.dev))

    def __exit__(self, *exc):
        if _timer is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record(torch.cuda.current_stream(self.dev))
            _timer.records.append((self.name, self.e0, e1, self.flops, self.nbytes, self.member))


def _stream(t: torch.Tensor):
    """torch's current stream on the tensor's device.  The launch goes to the CURRENT device (kernel attributes such as the > 64 KiB
    LDS opt-in are per device), so a tensor on another device is refused loudly: the model-level entry points make the input's
    device current (`engine.on_device`)."""
    idx = t.device.index
    if idx != torch.cuda.current_device():
        raise _lib.PeekvitHipError(f"tensor on {t.device} but the current device is cuda:{torch.cuda.current_device()}: wrap the call in "
                                   "`with torch.cuda.device(tensor.device):`")
    return C.c_void_p(raw_stream(idx))


def raw_stream(device_index: int) -> int:
    """hipStream_t of torch's current stream on that device, without building a torch.cuda.Stream object (~100 launches per forward: the eager
    small-batch path is host-bound)."""
    return torch._C._cuda_getCurrentRawStream(device_index)


# Operand-range guard (fp16-operand library, include/peekvit_hip.h `range_flag`): while `range_flag` holds a 1-element int32 GPU
# tensor the data-dependent operand producers (QKV / GELU GEMM epilogues, the fp32 patch gather) OR 1 into it when a value does
# not fit fp16.  engine.forward_auto zeroes it before and reads it after a forward.
# Per THREAD: another thread's guarded forward has its own flag word.
import threading as _threading
_tls = _threading.local()


def set_range_flag(flag: Optional[torch.Tensor]):
    _tls.range_flag = flag


def current_range_flag() -> Optional[torch.Tensor]:
    return getattr(_tls, "range_flag", None)


def _flag(dev):
    range_flag = current_range_flag()
    return C.c_void_p(range_flag.data_ptr()) if range_flag is not None and range_flag.device == dev else C.c_void_p(0)


def set_flag_word(k: int) -> int:
    """Which word of the flag tensor the ATTENTION launches of the calling thread raise their score bit in (engine.run_layers: 1 + layer index,
    so that a tripped forward knows WHICH layers to repeat in split precision); 0 = the common word.  Returns the previous value."""
    old = getattr(_tls, "flag_word", 0)
    _tls.flag_word = k
    return old


def _attn_flag(dev):
    flag = current_range_flag()
    if flag is None or flag.device != dev:
        return C.c_void_p(0)
    k = getattr(_tls, "flag_word", 0)
    return C.c_void_p(flag.data_ptr() + 4 * (k if k < flag.numel() else 0))


def workspace_bytes(use: int, *dims: int) -> int:
    """Scratch bytes of a C-ABI entry point for the given sizes: the library's own formula (pv_workspace_size, include/peekvit_hip.h PV_WS_*)."""
    arr = (C.c_int64 * len(dims))(*[int(d) for d in dims])
    n = int(_lib.load().pv_workspace_size(int(use), arr, len(dims)))
    if n < 0:
        check(n, "pv_workspace_size")
    return n


def _scratch_f32(use: int, dev, *dims: int) -> torch.Tensor:
    return torch.empty((workspace_bytes(use, *dims) // 4,), dtype=torch.float32, device=dev)


def _ptr(t: Optional[torch.Tensor]):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _chk(t: torch.Tensor, dtype, name: str):
    if not t.is_cuda:
        raise _lib.PeekvitHipError(f"{name}: expected a GPU tensor, got {t.device}")
    if t.dtype != dtype:
        raise _lib.PeekvitHipError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise _lib.PeekvitHipError(f"{name}: expected a contiguous tensor")
    return t


def _count():
    global launch_count
    launch_count += 1


def cast_bf16(src: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _chk(src, torch.float32, "src")
    if out is None:
        out = torch.empty(src.shape, dtype=_lib.operand_dtype(), device=src.device)
    with _timed("pv_cast_f32_bf16", src.device, 0.0, 6.0 * src.numel()):
        check(_lib.load().pv_cast_f32_bf16(_ptr(src), _ptr(out), src.numel(), _stream(src)), "pv_cast_f32_bf16")
    _count()
    return out


def patch_embed(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], pos: torch.Tensor, tokens: torch.Tensor, patch: int, row_off: int) -> torch.Tensor:
    """tokens[b, row_off + i, :] = patch_i(x[b]) . w^T + bias + pos[row_off + i, :] in ONE launch (pv_patch_embed_f32: no materialised patch matrix).
    x fp32 [B,C,R,R]; w 16-bit [D, C*P*P]; pos fp32 [>= row_off + Np, D]; tokens fp32 [B, S, D]."""
    _chk(x, torch.float32, "x"); _chk(w, _lib.operand_dtype(), "w"); _chk(pos, torch.float32, "pos"); _chk(tokens, torch.float32, "tokens")
    B, Cc, H, W = x.shape
    _, S, D = tokens.shape
    if H != W or w.shape != (D, Cc * patch * patch) or pos.shape[1] != D or pos.shape[0] < row_off + (H // patch) ** 2:      # (ResidualViT's budget-token row has no positional row)
        raise _lib.PeekvitHipError(f"patch_embed: shapes x {tuple(x.shape)}, w {tuple(w.shape)}, pos {tuple(pos.shape)}, tokens {tuple(tokens.shape)}")
    M = B * (H // patch) ** 2
    with _timed("pv_gemm_bf16", x.device, 2.0 * M * D * w.shape[1], 4.0 * x.numel() + 2.0 * w.numel() + 4.0 * M * D, member=(D, w.shape[1], _lib.PV_EPI_BIAS_POS_F32)):
        check(_lib.load().pv_patch_embed_f32(_ptr(x), _ptr(w), _ptr(bias), _ptr(pos), _ptr(tokens), B, Cc, H, patch, D, S, row_off, _flag(x.device), _stream(x)),
              "pv_patch_embed_f32")
    _count()
    return tokens


def im2col(x: torch.Tensor, patch: int, out: torch.Tensor) -> torch.Tensor:
    _chk(x, torch.float32, "x")
    B, Cc, H, W = x.shape
    with _timed("pv_im2col_bf16", x.device, 0.0, 6.0 * x.numel()):
        check(_lib.load().pv_im2col_bf16(_ptr(x), _ptr(out), B, Cc, H, W, patch, _flag(x.device), _stream(x)), "pv_im2col_bf16")
    _count()
    return out


def im2col_u8(x: torch.Tensor, patch: int, out: torch.Tensor, mean, std) -> torch.Tensor:
    """x: uint8 [B,H,W,3] (NHWC) -> bf16 patch matrix of the ToTensor+Normalize'd image."""
    _chk(x, torch.uint8, "x")
    B, H, W, Cc = x.shape
    if Cc != 3:
        raise _lib.PeekvitHipError("uint8 input must be NHWC with 3 channels")
    with _timed("pv_im2col_u8_bf16", x.device, 0.0, 3.0 * x.numel()):
        check(_lib.load().pv_im2col_u8_bf16(_ptr(x), _ptr(out), B, H, W, patch, *[float(v) for v in mean], *[float(v) for v in std],
                                            _stream(x)), "pv_im2col_u8_bf16")
    _count()
    return out


def token_prologue(tokens, special, pos, budget_token, budget: float, n_special: int):
    B, S, D = tokens.shape
    with _timed("pv_token_prologue", tokens.device, 0.0, 0.0):
        check(_lib.load().pv_token_prologue(_ptr(tokens), _ptr(special), _ptr(pos), _ptr(budget_token), float(budget),
                                            B, S, D, n_special, _stream(tokens)), "pv_token_prologue")
    _count()


def layernorm_bf16(x: torch.Tensor, gamma, beta, eps: float, out: torch.Tensor, row_scale=None):
    """x: fp32 [..., D] contiguous -> out bf16 same shape."""
    D = x.shape[-1]
    rows = x.numel() // D
    with _timed("pv_layernorm_bf16", x.device, 0.0, 6.0 * x.numel()):
        check(_lib.load().pv_layernorm_bf16(_ptr(x), D, _ptr(gamma), _ptr(beta), _ptr(row_scale), _ptr(out), rows, D,
                                            float(eps), _stream(x)), "pv_layernorm_bf16")
    _count()
    return out


def gemm(a: torch.Tensor, w: torch.Tensor, bias, out: torch.Tensor, epilogue: int, *, M=None, res=None,
         row_scale=None, pos=None, rows_per_img_in=0, rows_per_img_out=0, row_off=0, qcols=0, qscale=1.0, ln=None, ksplit=0, tag="", colsum_out=None, x16_out=None, rowstat_out=None, fold=None, rowsq_out=None, res_scaled=False):
    """out = epilogue(a[M,K] . w[N,K]^T).  a, w: bf16 2-D (row stride = shape[-1]).
    ksplit > 1: out is fp32 [ksplit, M, N] partial slices (PV_EPI_BIAS_F32), reduce with sum_slices().
    x16_out / rowstat_out (PV_EPI_BIAS_RES_F32) and fold = (stat [M,2], c1 [N], c2 [N]) (PV_EPI_BIAS_BF16 / _GELU_BF16, bias None): the
    producer / consumer halves of the folded LayerNorm (include/peekvit_hip.h).
    colsum_out (PV_EPI_GELU_GRAD_BF16): fp32 [N] tensor that receives the column sums of the output (bias gradient) - fused into the
    epilogue when the 256-row tile kernel serves the shape, a separate pv_colsum_f32 pass otherwise.
    ln = (gamma, beta, eps, ln_out_bf16, ln_row_scale | None): also emit bf16(LayerNorm(out)) (fused, PV_EPI_BIAS_RES_F32).
    res_scaled (with row_scale, PV_EPI_BIAS_RES_F32): out = row_scale * (res + a.w^T + bias) - ResidualViT with res = the unmasked tokens."""
    K = a.shape[-1]
    if M is None:
        M = a.numel() // K
    N = w.shape[0]
    range_flag = current_range_flag()
    args = GemmArgs(A=a.data_ptr(), W=w.data_ptr(), bias=bias.data_ptr() if bias is not None else 0,
                    out=out.data_ptr(), res=res.data_ptr() if res is not None else 0,
                    row_scale=row_scale.data_ptr() if row_scale is not None else 0,
                    pos=pos.data_ptr() if pos is not None else 0,
                    M=M, N=N, K=K, lda=a.stride(0) if a.dim() == 2 else K, ldw=w.stride(0) if w.dim() == 2 else w.shape[-1],
                    ldo=out.stride(-2) if out.dim() >= 2 else out.shape[-1],
                    ldr=(res.stride(0) if res.dim() == 2 else res.shape[-1]) if res is not None else 0,
                    rows_per_img_in=rows_per_img_in, rows_per_img_out=rows_per_img_out, row_off=row_off,
                    qcols=qcols, qscale=float(qscale), epilogue=epilogue,
                    ln_gamma=ln[0].data_ptr() if ln else 0, ln_beta=ln[1].data_ptr() if ln else 0,
                    ln_row_scale=ln[4].data_ptr() if ln and ln[4] is not None else 0,
                    ln_out=ln[3].data_ptr() if ln else 0, ln_eps=float(ln[2]) if ln else 0.0, ksplit=int(ksplit), colsum_partial=0,
                    x16_out=x16_out.data_ptr() if x16_out is not None else 0, rowstat_out=rowstat_out.data_ptr() if rowstat_out is not None else 0,
                    fold_stat=fold[0].data_ptr() if fold else 0, fold_c1=fold[1].data_ptr() if fold else 0, fold_c2=fold[2].data_ptr() if fold else 0,
                    range_flag=range_flag.data_ptr() if range_flag is not None and range_flag.device == a.device else 0,
                    rowsq_out=rowsq_out.data_ptr() if rowsq_out is not None else 0, res_scaled=int(res_scaled))
    part = None
    if colsum_out is not None and _lib.load().pv_gemm_tile_rows(C.byref(args)) == 256:
        part = _scratch_f32(_lib.PV_WS_GEMM_COLSUM_PARTIAL, a.device, M, N).view((M + 255) // 256, N)
        args.colsum_partial = part.data_ptr()
    # algorithmic bytes: both operands once, the output once (+ the residual rows it adds, + the 16-bit copies the fused / folded LayerNorm forms emit)
    nbytes = 2.0 * (M * K + N * K) + out.element_size() * M * N * (2 if res is not None else 1) + (2.0 * M * N if ln else 0.0) + (2.0 * M * N if x16_out is not None else 0.0)
    with _timed("pv_gemm_bf16" + tag, a.device, 2.0 * M * N * K, nbytes, member=(N, K, epilogue)):
        check(_lib.load().pv_gemm_bf16(C.byref(args), _stream(a)), "pv_gemm_bf16")
    _count()
    if colsum_out is not None:
        colsum(part if part is not None else out, colsum_out)
    return out


def rowstat_finalize(partials: torch.Tensor, D: int, eps: float, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """partials fp32 [tiles, rows, 2] (sum, sum of squares per column tile) -> fp32 [rows, 2] (mean, rstd) of rows of length D."""
    _chk(partials, torch.float32, "partials")
    T, rows, _ = partials.shape
    if out is None:
        out = torch.empty((rows, 2), dtype=torch.float32, device=partials.device)
    with _timed("pv_rowstat_finalize", partials.device, 0.0, 8.0 * (T + 1) * rows):
        check(_lib.load().pv_rowstat_finalize(_ptr(partials), _ptr(out), T, rows, D, float(eps), _flag(partials.device), _stream(partials)), "pv_rowstat_finalize")
    _count()
    return out


def gemm_tn(a: torch.Tensor, b: torch.Tensor, part: torch.Tensor, ksplit: int, tag: str = "[wgrad]"):
    """part[t] = (a[K_t, M])^T . b[K_t, N] over ksplit row slices: weight gradient straight from row-major bf16 activations
    (row-strided 2-D views allowed).  part fp32 [ksplit, M, N]; reduce with sum_slices()."""
    K, M = a.shape
    N = b.shape[1]
    assert b.shape[0] == K and part.shape == (max(ksplit, 1), M, N) and part.is_contiguous()
    args = GemmArgs(A=a.data_ptr(), W=b.data_ptr(), bias=0, out=part.data_ptr(), res=0, row_scale=0, pos=0, M=M, N=N, K=K,
                    lda=a.stride(0), ldw=b.stride(0), ldo=N, ldr=0, rows_per_img_in=0, rows_per_img_out=0, row_off=0, qcols=0,
                    qscale=1.0, epilogue=_lib.PV_EPI_BIAS_F32, ln_gamma=0, ln_beta=0, ln_row_scale=0, ln_out=0, ln_eps=0.0,
                    ksplit=int(ksplit))
    with _timed("pv_gemm_tn_bf16" + tag, a.device, 2.0 * M * N * K, 2.0 * K * (M + N) + 4.0 * max(ksplit, 1) * M * N):
        check(_lib.load().pv_gemm_tn_bf16(C.byref(args), _stream(a)), "pv_gemm_tn_bf16")
    _count()
    return part


def attention(qkv: torch.Tensor, out: torch.Tensor, B: int, S: int, H: int, dh: int, lse: Optional[torch.Tensor] = None):
    """lse (fp32 [B, H, S], optional; training forward): also receives log2(sum_k exp(s[q,k])) per row, for attention_bwd_lse."""
    with _timed("pv_attention_bf16", qkv.device, 4.0 * B * H * S * S * dh, 8.0 * B * S * H * dh):
        if lse is None:
            check(_lib.load().pv_attention_bf16(_ptr(qkv), _ptr(out), B, S, H, dh, _attn_flag(qkv.device), _stream(qkv)), "pv_attention_bf16")
        else:
            _chk(lse, torch.float32, "lse")
            if lse.numel() != B * H * S:
                raise _lib.PeekvitHipError("attention: lse must hold B * H * S values")
            check(_lib.load().pv_attention_lse_bf16(_ptr(qkv), _ptr(out), _ptr(lse), B, S, H, dh, _attn_flag(qkv.device), _stream(qkv)), "pv_attention_lse_bf16")
    _count()
    return out


def attention_bwd_lse_ok(S: int, dh: int) -> bool:
    """Shapes for which the training path takes the persistent backward (pv_attention_bwd_lse_bf16): where it serves (attention_bwd_lse_supported) AND is the
    faster kernel - 10 .. 13 query tiles (S = 197: 1.70 vs 1.97 ms, 177: 1.48 vs 1.64, 158: 1.25 vs 1.32; at 9 tiles, S = 129, the two are equal)."""
    return attention_bwd_lse_supported(S, dh) and S >= 145


def attention_bwd_lse_supported(S: int, dh: int) -> bool:
    """Shapes pv_attention_bwd_lse_bf16 accepts: 9 .. 13 query tiles of 16 (one wave each, at least three waves left for the side work), dh = 48 / 64."""
    return dh in (48, 64) and 129 <= S <= 208


def attention_rows(q: torch.Tensor, kv: torch.Tensor, out: torch.Tensor, B: int, S: int, nq: int, H: int, dh: int):
    """softmax(q k^T) v for the first `nq` rows of every image only: q 16-bit [B*nq, H*dh] (scaled), kv 16-bit [B*S, >= 2*H*dh] (k | v),
    out 16-bit [B*nq, H*dh].  Row strides are taken from the tensors (2-D views of wider buffers allowed)."""
    for t, name in ((q, "q"), (kv, "kv"), (out, "out")):
        if not (t.is_cuda and t.dtype == _lib.operand_dtype() and t.dim() == 2 and t.stride(1) == 1):
            raise _lib.PeekvitHipError(f"attention_rows: {name} must be a 2-D GPU tensor of the operand type with unit column stride")
    if q.shape[0] != B * nq or out.shape[0] != B * nq or kv.shape[0] != B * S or kv.shape[1] < 2 * H * dh:
        raise _lib.PeekvitHipError("attention_rows: shape mismatch")
    with _timed("pv_attention_rows_bf16", q.device, 4.0 * B * H * nq * S * dh, 4.0 * B * S * H * dh + 4.0 * B * nq * H * dh):
        check(_lib.load().pv_attention_rows_bf16(_ptr(q), q.stride(0), _ptr(kv), kv.stride(0), _ptr(out), out.stride(0),
                                                 B, S, nq, H, dh, _attn_flag(q.device), _stream(q)), "pv_attention_rows_bf16")
    _count()
    return out


def attention_rows_bwd(q, kv, out, dout, dq, dkv, B: int, S: int, H: int, dh: int, qscale: float):
    """Backward of attention_rows for one query row per image: q, out, dout, dq 16-bit [B, H*dh]; kv, dkv 16-bit [B*S, >= 2*H*dh]."""
    for t, name in ((q, "q"), (kv, "kv"), (out, "out"), (dout, "dout"), (dq, "dq"), (dkv, "dkv")):
        if not (t.is_cuda and t.dtype == _lib.operand_dtype() and t.dim() == 2 and t.stride(1) == 1):
            raise _lib.PeekvitHipError(f"attention_rows_bwd: {name} must be a 2-D GPU tensor of the operand type with unit column stride")
    if any(t.shape[0] != B for t in (q, out, dout, dq)) or kv.shape[0] != B * S or dkv.shape[0] != B * S:
        raise _lib.PeekvitHipError("attention_rows_bwd: shape mismatch")
    with _timed("pv_attention_rows_bwd_bf16", q.device, 10.0 * B * H * S * dh, 2.0 * 3 * 2 * B * S * H * dh):
        check(_lib.load().pv_attention_rows_bwd_bf16(_ptr(q), q.stride(0), _ptr(kv), kv.stride(0), _ptr(out), out.stride(0), _ptr(dout), dout.stride(0),
                                                     _ptr(dq), dq.stride(0), _ptr(dkv), dkv.stride(0), B, S, 1, H, dh, float(qscale), _stream(q)),
              "pv_attention_rows_bwd_bf16")
    _count()
    return dq, dkv


def cls_pool(x: torch.Tensor, gamma, beta, eps: float, num_cls: int) -> torch.Tensor:
    B, S, D = x.shape
    pooled = torch.empty((B, D), dtype=torch.float32, device=x.device)
    with _timed("pv_cls_pool", x.device, 0.0, 0.0):
        check(_lib.load().pv_cls_pool(_ptr(x), _ptr(gamma), _ptr(beta), _ptr(pooled), B, S, D, num_cls, float(eps),
                                      _stream(x)), "pv_cls_pool")
    _count()
    return pooled


def head(pooled: torch.Tensor, w: torch.Tensor, b) -> torch.Tensor:
    B, D = pooled.shape
    Cn = w.shape[0]
    logits = torch.empty((B, Cn), dtype=torch.float32, device=pooled.device)
    with _timed("pv_head_f32", pooled.device, 2.0 * B * D * Cn, 0.0):
        check(_lib.load().pv_head_f32(_ptr(pooled), _ptr(w), _ptr(b), _ptr(logits), B, D, Cn, _stream(pooled)), "pv_head_f32")
    _count()
    return logits


# ---- backward building blocks ---------------------------------------------------------------------------------------
def sum_slices(partials: torch.Tensor, out: torch.Tensor, accumulate: bool = False, base: Optional[torch.Tensor] = None, ln=None) -> torch.Tensor:
    """out (+)= partials.sum(0); partials fp32 [S, ...] contiguous, out fp32 of the trailing shape.  base (fp32, same shape as out, may be
    out): out = base + partials.sum(0) - the finish of a split-K GEMM with a residual; with ln = (gamma, beta, eps, out16 [rows, D]) the same
    pass also emits the 16-bit LayerNorm of the finished rows (out viewed as [rows, D])."""
    _chk(partials, torch.float32, "partials"); _chk(out, torch.float32, "out")
    S = partials.shape[0]
    with _timed("pv_sum_slices_f32", out.device, 0.0, 4.0 * (S + 1 + (base is not None)) * out.numel()):
        if ln is not None:
            _chk(base, torch.float32, "base"); _chk(ln[3], _lib.operand_dtype(), "ln out")
            D = ln[3].shape[-1]
            assert base.numel() == out.numel() == ln[3].numel() and base.is_contiguous() and out.is_contiguous()
            check(_lib.load().pv_sum_slices_add_ln_f32(_ptr(partials), _ptr(base), _ptr(out), out.numel() // D, D, S, _ptr(ln[0]), _ptr(ln[1]), float(ln[2]),
                                                       _ptr(ln[3]), _stream(out)), "pv_sum_slices_add_ln_f32")
        elif base is not None:
            _chk(base, torch.float32, "base")
            assert base.numel() == out.numel() and base.is_contiguous() and out.is_contiguous()
            check(_lib.load().pv_sum_slices_add_f32(_ptr(partials), _ptr(base), _ptr(out), out.numel(), S, _stream(out)), "pv_sum_slices_add_f32")
        else:
            check(_lib.load().pv_sum_slices_f32(_ptr(partials), _ptr(out), out.numel(), S, int(accumulate), _stream(out)), "pv_sum_slices_f32")
    _count()
    return out


def sum_slices_act(partials: torch.Tensor, out: torch.Tensor, gelu: bool = False, qcols: int = 0, qscale: float = 1.0) -> torch.Tensor:
    """out (16-bit [M, N], contiguous) = gelu(partials.sum(0)) or the sum with the first qcols columns scaled: finish of a split-K fc1 / in-proj."""
    _chk(partials, torch.float32, "partials"); _chk(out, _lib.operand_dtype(), "out")
    S, M, N = partials.shape
    assert out.shape == (M, N)
    with _timed("pv_sum_slices_f32", out.device, 0.0, (4.0 * S + 2.0) * out.numel()):
        check(_lib.load().pv_sum_slices_act_bf16(_ptr(partials), _ptr(out), M, N, S, int(gelu), int(qcols), float(qscale), _flag(out.device), _stream(out)),
              "pv_sum_slices_act_bf16")
    _count()
    return out


def transpose(src: torch.Tensor, out: Optional[torch.Tensor] = None, pad_to: int = 1, colsum_out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """bf16 [R,C] (row-strided view allowed) -> bf16 [C, ceil(R / pad_to) * pad_to], zero-filled beyond column R."""
    if not (src.is_cuda and src.dtype == _lib.operand_dtype() and src.dim() == 2 and src.stride(1) == 1):
        raise _lib.PeekvitHipError("transpose: expected a 2-D bf16 GPU tensor with unit column stride")
    R, Cc = src.shape
    ldd = (R + pad_to - 1) // pad_to * pad_to
    if out is None:
        out = torch.empty((Cc, ldd), dtype=_lib.operand_dtype(), device=src.device)
    assert out.shape == (Cc, ldd) and out.is_contiguous()
    ws = _scratch_f32(_lib.PV_WS_TRANSPOSE_COLSUM, src.device, R, Cc, ldd) if colsum_out is not None else None
    with _timed("pv_transpose_bf16", src.device, 0.0, 4.0 * src.numel()):
        check(_lib.load().pv_transpose_bf16(_ptr(src), src.stride(0), _ptr(out), R, Cc, ldd, _ptr(colsum_out), _ptr(ws), _stream(src)),
              "pv_transpose_bf16")
    _count()
    return out


def colsum(src: torch.Tensor, out: torch.Tensor, accumulate: bool = False) -> torch.Tensor:
    """out[C] (+)= src[R,C].sum(0) in fp32; src bf16 or fp32."""
    assert src.dtype in (_lib.operand_dtype(), torch.float32) and src.is_contiguous()
    _chk(out, torch.float32, "out")
    R, Cc = src.shape
    ws = _scratch_f32(_lib.PV_WS_COLSUM, src.device, R, Cc)
    with _timed("pv_colsum_f32", src.device, 0.0, float(src.element_size() * src.numel())):
        check(_lib.load().pv_colsum_f32(_ptr(src), int(src.dtype == _lib.operand_dtype()), _ptr(out), _ptr(ws), R, Cc, int(accumulate),
                                        _stream(src)), "pv_colsum_f32")
    _count()
    return out


def layernorm_bwd(x: torch.Tensor, dy: torch.Tensor, gamma: torch.Tensor, dres_in, dx_out: Optional[torch.Tensor], dgb: torch.Tensor, eps: float,
                  accumulate: bool = False, dx_bf16: Optional[torch.Tensor] = None):
    """dx_out = (dres_in or 0) + LN'(x)^T dy;  dgb [3,D] (+)= (dgamma, dbeta, colsum(dx)).  x fp32 [rows,D], dy bf16 [rows,D].
    dres_in may be a 16-BIT tensor and dx_out None (only the 16-bit copy dx_bf16 is written): the residual gradient handed over in 16 bits
    between the two LayerNorms of a block (pv_layernorm_bwd16; an option of the training path)."""
    _chk(x, torch.float32, "x"); _chk(dy, _lib.operand_dtype(), "dy"); _chk(dgb, torch.float32, "dgb")
    D = x.shape[-1]
    rows = x.numel() // D
    ws = _scratch_f32(_lib.PV_WS_LAYERNORM_BWD, x.device, rows, D)
    if dx_out is None or (dres_in is not None and dres_in.dtype != torch.float32):
        d16 = dres_in if dres_in is not None and dres_in.dtype != torch.float32 else None
        d32 = dres_in if dres_in is not None and dres_in.dtype == torch.float32 else None
        if d16 is not None:
            _chk(d16, _lib.operand_dtype(), "dres16")
        nb = 4.0 + 2.0 + (4.0 if dx_out is not None else 0.0) + (4.0 if d32 is not None else 0.0) + (2.0 if d16 is not None else 0.0) + (2.0 if dx_bf16 is not None else 0.0)
        with _timed("pv_layernorm_bwd", x.device, 0.0, nb * x.numel()):
            check(_lib.load().pv_layernorm_bwd16(_ptr(x), _ptr(dy), _ptr(gamma), _ptr(d32), _ptr(d16), _ptr(dx_out), _ptr(dx_bf16), _ptr(dgb), _ptr(ws), ws.numel(),
                                                 rows, D, float(eps), int(accumulate), _stream(x)), "pv_layernorm_bwd16")
        _count()
        return dx_out
    _chk(dx_out, torch.float32, "dx_out")
    with _timed("pv_layernorm_bwd", x.device, 0.0, (4.0 + 2.0 + 4.0 + (4.0 if dres_in is not None else 0.0) + (2.0 if dx_bf16 is not None else 0.0)) * x.numel()):
        check(_lib.load().pv_layernorm_bwd(_ptr(x), _ptr(dy), _ptr(gamma), _ptr(dres_in) if dres_in is not None else 0, _ptr(dx_out), _ptr(dx_bf16), _ptr(dgb),
                                           _ptr(ws), ws.numel(), rows, D, float(eps), int(accumulate), _stream(x)), "pv_layernorm_bwd")
    _count()
    return dx_out


def layernorm_bwd_masked(x, dy, gamma, beta, row_scale, dres_in, u, dx_out, dx_bf16, scale_copy: bool, dgb, dmask, dmask_accumulate: bool,
                         eps: float):
    """Backward of y = row_scale * LayerNorm(x) (ResidualViT): as layernorm_bwd, plus dmask [rows] (+)= rowdot(dy, LN(x))
    (+ rowdot(dx_out, u) when u is given); dx_bf16 = row_scale * dx_out when scale_copy."""
    _chk(x, torch.float32, "x"); _chk(dy, _lib.operand_dtype(), "dy"); _chk(dx_out, torch.float32, "dx_out")
    _chk(dgb, torch.float32, "dgb"); _chk(dmask, torch.float32, "dmask"); _chk(row_scale, torch.float32, "row_scale")
    D = x.shape[-1]
    rows = x.numel() // D
    ws = _scratch_f32(_lib.PV_WS_LAYERNORM_BWD, x.device, rows, D)
    with _timed("pv_layernorm_bwd", x.device, 0.0, 16.0 * x.numel()):
        check(_lib.load().pv_layernorm_bwd_masked(_ptr(x), _ptr(dy), _ptr(gamma), _ptr(beta), _ptr(row_scale), _ptr(dres_in), _ptr(u),
                                                  _ptr(dx_out), _ptr(dx_bf16), int(scale_copy), _ptr(dgb), _ptr(dmask), int(dmask_accumulate),
                                                  _ptr(ws), ws.numel(), rows, D, float(eps), _stream(x)), "pv_layernorm_bwd_masked")
    _count()
    return dx_out


def masked_residual(x: torch.Tensor, u: torch.Tensor, row_scale: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    """out = x + row_scale[row] * u  (x, out fp32 [rows, D]; u 16-bit [rows, D])."""
    _chk(x, torch.float32, "x"); _chk(u, _lib.operand_dtype(), "u"); _chk(out, torch.float32, "out")
    D = x.shape[-1]
    with _timed("pv_masked_residual", x.device, 0.0, 10.0 * x.numel()):
        check(_lib.load().pv_masked_residual(_ptr(x), _ptr(u), _ptr(row_scale), _ptr(out), x.numel() // D, D, _stream(x)), "pv_masked_residual")
    _count()
    return out


def gelu(pre: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _chk(pre, _lib.operand_dtype(), "pre")
    if out is None:
        out = torch.empty_like(pre)
    with _timed("pv_gelu_bf16", pre.device, 0.0, 4.0 * pre.numel()):
        check(_lib.load().pv_gelu_bf16(_ptr(pre), _ptr(out), pre.numel(), _stream(pre)), "pv_gelu_bf16")
    _count()
    return out


def gelu_bwd(pre: torch.Tensor, dg: torch.Tensor, dpre: Optional[torch.Tensor] = None) -> torch.Tensor:
    _chk(pre, _lib.operand_dtype(), "pre"); _chk(dg, _lib.operand_dtype(), "dg")
    if dpre is None:
        dpre = dg
    with _timed("pv_gelu_bwd_bf16", pre.device, 0.0, 6.0 * pre.numel()):
        check(_lib.load().pv_gelu_bwd_bf16(_ptr(pre), _ptr(dg), _ptr(dpre), pre.numel(), _stream(pre)), "pv_gelu_bwd_bf16")
    _count()
    return dpre


def attention_bwd(qkv: torch.Tensor, dout: torch.Tensor, dqkv: torch.Tensor, B: int, S: int, H: int, dh: int, qscale: float,
                  dbias_partial: Optional[torch.Tensor] = None):
    """dbias_partial (fp32 [B, 3*H*dh], optional) receives the per-image column sums of dqkv."""
    with _timed("pv_attention_bwd_bf16", qkv.device, 14.0 * B * H * S * S * dh, 14.0 * B * S * H * dh):
        check(_lib.load().pv_attention_bwd_bf16(_ptr(qkv), _ptr(dout), _ptr(dqkv), _ptr(dbias_partial), B, S, H, dh, float(qscale),
                                                _stream(qkv)), "pv_attention_bwd_bf16")
    _count()
    return dqkv


def attention_bwd_lse(qkv: torch.Tensor, dout: torch.Tensor, out: torch.Tensor, lse: torch.Tensor, dqkv: torch.Tensor, B: int, S: int, H: int, dh: int,
                      qscale: float, dbias_partial: Optional[torch.Tensor] = None):
    """attention_bwd from the forward's output `out` (16-bit [B, S, H*dh]) and row statistics `lse` (attention(..., lse=)): one persistent workgroup per CU."""
    _chk(lse, torch.float32, "lse")
    with _timed("pv_attention_bwd_bf16", qkv.device, 14.0 * B * H * S * S * dh, 16.0 * B * S * H * dh):
        check(_lib.load().pv_attention_bwd_lse_bf16(_ptr(qkv), _ptr(dout), _ptr(out), _ptr(lse), _ptr(dqkv), _ptr(dbias_partial), B, S, H, dh,
                                                    float(qscale), _stream(qkv)), "pv_attention_bwd_lse_bf16")
    _count()
    return dqkv


def wgrad(dy_t: torch.Tensor, x_t: torch.Tensor, out: torch.Tensor, accumulate: bool = False, ksplit: int = 0) -> torch.Tensor:
    """out[N_out, N_in] (+)= dY^T . X from the TRANSPOSED bf16 activations dy_t [N_out, M], x_t [N_in, M] (split-K over M)."""
    No, M = dy_t.shape
    Ni = x_t.shape[0]
    if ksplit <= 0:
        tiles = ((No + 255) // 256) * ((Ni + 255) // 256)
        ksplit = 1
        while tiles * ksplit < 512 and M % (ksplit * 2 * 128) == 0 and M // (ksplit * 2) >= 1024:
            ksplit *= 2
    part = _scratch_f32(_lib.PV_WS_GEMM_SPLITK, out.device, No, Ni, ksplit).view(ksplit, No, Ni)
    gemm(dy_t, x_t, None, part, _lib.PV_EPI_BIAS_F32, ksplit=ksplit)
    return sum_slices(part, out, accumulate)


# ---- precision mode "bf16x3" ---------------------------------------------------------------------------------------
def split3(src: torch.Tensor, order: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """fp32 [rows,K] -> bf16 [rows,3K]: order 0 = [hi|lo|hi] (activations), 1 = [hi|hi|lo] (weights)."""
    _chk(src, torch.float32, "src")
    rows, K = src.shape
    if out is None:
        out = torch.empty((rows, 3 * K), dtype=_lib.operand_dtype(), device=src.device)
    with _timed("pv_split3_f32_bf16", src.device, 0.0, 10.0 * src.numel()):
        check(_lib.load().pv_split3_f32_bf16(_ptr(src), _ptr(out), rows, K, order, _stream(src)), "pv_split3_f32_bf16")
    _count()
    return out


def im2col_split(x: torch.Tensor, patch: int, out: torch.Tensor) -> torch.Tensor:
    _chk(x, torch.float32, "x")
    B, Cc, H, W = x.shape
    with _timed("pv_im2col_split_bf16", x.device, 0.0, 10.0 * x.numel()):
        check(_lib.load().pv_im2col_split_bf16(_ptr(x), _ptr(out), B, Cc, H, W, patch, _stream(x)), "pv_im2col_split_bf16")
    _count()
    return out


def layernorm_split(x: torch.Tensor, gamma, beta, eps: float, out: torch.Tensor, row_scale=None):
    D = x.shape[-1]
    rows = x.numel() // D
    with _timed("pv_layernorm_split_bf16", x.device, 0.0, 10.0 * x.numel()):
        check(_lib.load().pv_layernorm_split_bf16(_ptr(x), D, _ptr(gamma), _ptr(beta), _ptr(row_scale), _ptr(out), rows, D,
                                                  float(eps), _stream(x)), "pv_layernorm_split_bf16")
    _count()
    return out


def attention_f32(qkv: torch.Tensor, out: torch.Tensor, B: int, S: int, H: int, dh: int):
    _chk(qkv, torch.float32, "qkv")
    with _timed("pv_attention_f32_split", qkv.device, 4.0 * B * H * S * S * dh, 18.0 * B * S * H * dh):
        check(_lib.load().pv_attention_f32_split(_ptr(qkv), _ptr(out), B, S, H, dh, _stream(qkv)), "pv_attention_f32_split")
    _count()
    return out


def attention_split(qkv: torch.Tensor, out: torch.Tensor, B: int, S: int, H: int, dh: int):
    """Attention with split-operand scores (the local fallback of the score guard): qkv fp32 [B*S, 3*H*dh] (q pre-scaled) -> out 16-bit [B*S, H*dh]."""
    _chk(qkv, torch.float32, "qkv"); _chk(out, _lib.operand_dtype(), "out")
    with _timed("pv_attention_split_bf16", qkv.device, 8.0 * B * H * S * S * dh, 14.0 * B * S * H * dh):
        check(_lib.load().pv_attention_split_bf16(_ptr(qkv), _ptr(out), B, S, H, dh, _stream(qkv)), "pv_attention_split_bf16")
    _count()
    return out


def token_norm(x: torch.Tensor) -> torch.Tensor:
    B, S, D = x.shape
    norms = torch.empty((B, S - 1), dtype=torch.float32, device=x.device)
    with _timed("pv_token_norm", x.device, 0.0, 4.0 * x.numel()):
        check(_lib.load().pv_token_norm(_ptr(x), _ptr(norms), B, S, D, _stream(x)), "pv_token_norm")
    _count()
    return norms


def rank_topk(norms: torch.Tensor, k: int, gap_min: Optional[torch.Tensor] = None) -> torch.Tensor:
    """keep int32 [B,k]; gap_min (fp32 [B], optional) is lowered to each image's relative gap at the keep boundary (include/peekvit_hip.h pv_rank_topk_gap)."""
    B, N = norms.shape
    keep = torch.empty((B, k), dtype=torch.int32, device=norms.device)
    if gap_min is not None:
        _chk(gap_min, torch.float32, "gap_min")
        assert gap_min.numel() == B
    with _timed("pv_rank_topk", norms.device, 0.0, 4.0 * (norms.numel() + B * k)):
        check(_lib.load().pv_rank_topk_gap(_ptr(norms), _ptr(keep), _ptr(gap_min), B, N, k, _stream(norms)), "pv_rank_topk_gap")
    _count()
    return keep


def gemm_tile_rows(M: int, N: int, K: int, epilogue: int) -> int:
    """The M-tile height (256 or 128) pv_gemm_bf16 would choose for this shape (features such as rowsq_out need 256)."""
    args = GemmArgs(A=16, W=16, out=16, res=16, M=M, N=N, K=K, lda=K, ldw=K, ldo=N, ldr=N, qscale=1.0, epilogue=epilogue)
    return int(_lib.load().pv_gemm_tile_rows(C.byref(args)))


def rank_topk_partials(rowsq: torch.Tensor, B: int, S: int, k: int, gap_min: Optional[torch.Tensor] = None) -> torch.Tensor:
    """keep int32 [B,k] from a producer GEMM's per-column-tile row sums of squares (rowsq fp32 [tiles, B*S]); gap_min as in rank_topk."""
    _chk(rowsq, torch.float32, "rowsq")
    tiles = rowsq.shape[0]
    keep = torch.empty((B, k), dtype=torch.int32, device=rowsq.device)
    if gap_min is not None:
        _chk(gap_min, torch.float32, "gap_min")
        assert gap_min.numel() == B
    with _timed("pv_rank_topk", rowsq.device, 0.0, 4.0 * (rowsq.numel() + B * k)):
        check(_lib.load().pv_rank_topk_partials_gap(_ptr(rowsq), tiles, _ptr(keep), _ptr(gap_min), B, S, k, _stream(rowsq)), "pv_rank_topk_partials_gap")
    _count()
    return keep


def gather_tokens(x: torch.Tensor, keep: torch.Tensor) -> torch.Tensor:
    B, S, D = x.shape
    k = keep.shape[1]
    out = torch.empty((B, k + 1, D), dtype=torch.float32, device=x.device)
    with _timed("pv_gather_tokens", x.device, 0.0, 8.0 * B * (k + 1) * D):
        check(_lib.load().pv_gather_tokens(_ptr(x), _ptr(keep), _ptr(out), B, S, k, D, _stream(x)), "pv_gather_tokens")
    _count()
    return out


def scatter_tokens(dy: torch.Tensor, keep: torch.Tensor, S_in: int) -> torch.Tensor:
    """Backward of gather_tokens: dy fp32 [B,1+k,D] -> dx fp32 [B,S_in,D] (zeros at dropped tokens)."""
    _chk(dy, torch.float32, "dy")
    B, k1, D = dy.shape
    dx = torch.empty((B, S_in, D), dtype=torch.float32, device=dy.device)
    with _timed("pv_scatter_tokens", dy.device, 0.0, 4.0 * B * (k1 + S_in) * D):
        check(_lib.load().pv_scatter_tokens(_ptr(dy), _ptr(keep), _ptr(dx), B, S_in, k1 - 1, D, _stream(dy)), "pv_scatter_tokens")
    _count()
    return dx


def residual_gate(x: torch.Tensor, x_out: torch.Tensor, wg, bg, wb, bb, temp: float, sigmoid_bias: float, thr_out: Optional[torch.Tensor] = None,
                  ln=None):
    """Returns (mask [B,N,1], row_scale [B,S]); x_out receives [cls | mask*img | budget]; thr_out (fp32 [B], optional): the thresholds;
    ln = (gamma, beta, eps, out 16-bit [B*S, D]): also row_scale * LayerNorm(x_out) from the same pass."""
    B, S, D = x.shape
    mask = torch.empty((B, S - 2, 1), dtype=torch.float32, device=x.device)
    row_scale = torch.empty((B, S), dtype=torch.float32, device=x.device)
    if ln is not None:
        _chk(ln[3], _lib.operand_dtype(), "ln out")
    with _timed("pv_residual_gate", x.device, 0.0, (8.0 + (2.0 if ln is not None else 0.0)) * x.numel()):
        check(_lib.load().pv_residual_gate(_ptr(x), _ptr(x_out), _ptr(wg), _ptr(bg), _ptr(wb), _ptr(bb), float(temp),
                                           float(sigmoid_bias), _ptr(mask), _ptr(row_scale), _ptr(thr_out),
                                           _ptr(ln[0]) if ln is not None else C.c_void_p(0), _ptr(ln[1]) if ln is not None else C.c_void_p(0),
                                           float(ln[2]) if ln is not None else 0.0, _ptr(ln[3]) if ln is not None else C.c_void_p(0),
                                           B, S, D, _stream(x)),
              "pv_residual_gate")
    _count()
    return mask, row_scale


def residual_gate_bwd(x: torch.Tensor, dxo: torch.Tensor, drow: torch.Tensor, wg, bg, wb, bb, temp: float, sigmoid_bias: float):
    """Backward of residual_gate: returns (dx [B,S,D], dwg [D], dbg [1], dwb [D], dbb [1]) for x, dxo fp32 [B,S,D], drow fp32 [B,S]."""
    _chk(x, torch.float32, "x"); _chk(dxo, torch.float32, "dxo"); _chk(drow, torch.float32, "drow")
    B, S, D = x.shape
    dev = x.device
    dx = torch.empty_like(x)
    dwg_p = torch.empty((B, D), dtype=torch.float32, device=dev)
    dwb_p = torch.empty((B, D), dtype=torch.float32, device=dev)
    scal_p = torch.empty((B, 4), dtype=torch.float32, device=dev)          # (dbg, dbb, 0, 0): four columns for pv_colsum_f32's 16-byte loads
    with _timed("pv_residual_gate_bwd", dev, 0.0, 12.0 * x.numel()):
        check(_lib.load().pv_residual_gate_bwd(_ptr(x), _ptr(dxo), _ptr(drow), _ptr(wg), _ptr(bg), _ptr(wb), _ptr(bb), float(temp), float(sigmoid_bias),
                                               _ptr(dx), _ptr(dwg_p), _ptr(dwb_p), _ptr(scal_p), B, S, D, _stream(x)), "pv_residual_gate_bwd")
    _count()
    dwg = colsum(dwg_p, torch.empty((D,), dtype=torch.float32, device=dev))
    dwb = colsum(dwb_p, torch.empty((D,), dtype=torch.float32, device=dev))
    scal = colsum(scal_p, torch.empty((4,), dtype=torch.float32, device=dev))
    return dx, dwg, scal[0:1], dwb, scal[1:2]
