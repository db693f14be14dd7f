"""ctypes binding of libpeekvit_hip.so (the C ABI declared in include/peekvit_hip.h).

The library is built in-tree by `peekvit_amd._build.build()` (hipcc, gfx950).  There is NO fallback:
if the shared object is missing or a symbol is absent, loading raises - the HIP path must be the
path that runs on a GPU box.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

# torch bundles its own libamdhip64.so; it MUST be in the process before our library is dlopen'ed, otherwise our
# NEEDED libamdhip64.so.7 resolves to /opt/rocm's copy and the process ends up with two HIP runtimes (our
# launches then fail against torch's streams/pointers).
import torch  # noqa: F401

from ._build import LIB

_i64, _f32, _p, _i32 = C.c_int64, C.c_float, C.c_void_p, C.c_int32

PV_EPI_BIAS_BF16, PV_EPI_BIAS_GELU_BF16, PV_EPI_BIAS_RES_F32, PV_EPI_BIAS_POS_F32 = 0, 1, 2, 3
PV_EPI_BIAS_F32, PV_EPI_BIAS_GELU_SPLIT_BF16 = 4, 5
PV_EPI_BIAS_GELU_PAIR_BF16, PV_EPI_GELU_GRAD_BF16 = 6, 7
PV_WS_TRANSPOSE_COLSUM, PV_WS_COLSUM, PV_WS_LAYERNORM_BWD, PV_WS_GEMM_COLSUM_PARTIAL, PV_WS_GEMM_SPLITK = 1, 2, 3, 4, 5


class GemmArgs(C.Structure):
    """Mirror of `pv_gemm_args` (include/peekvit_hip.h)."""
    _fields_ = [("struct_size", C.c_uint64), ("A", _p), ("W", _p), ("bias", _p), ("out", _p), ("res", _p), ("row_scale", _p), ("pos", _p),
                ("M", _i64), ("N", _i64), ("K", _i64), ("lda", _i64), ("ldw", _i64), ("ldo", _i64), ("ldr", _i64),
                ("rows_per_img_in", _i64), ("rows_per_img_out", _i64), ("row_off", _i64), ("qcols", _i64),
                ("qscale", _f32), ("epilogue", _i32),
                ("ln_gamma", _p), ("ln_beta", _p), ("ln_row_scale", _p), ("ln_out", _p), ("ln_eps", _f32),
                ("ksplit", _i32), ("colsum_partial", _p),
                ("x16_out", _p), ("rowstat_out", _p), ("fold_stat", _p), ("fold_c1", _p), ("fold_c2", _p), ("range_flag", _p), ("rowsq_out", _p),
                ("res_scaled", _i32)]

    def __init__(self, *args, **kw):
        super().__init__(*args, **kw)
        self.struct_size = C.sizeof(GemmArgs)      # since ABI v7: the library refuses a struct of another length


# name -> (restype, argtypes); every symbol include/peekvit_hip.h declares
SIGNATURES = {
    "pv_version": (C.c_int, []),
    "pv_gemm_args_size": (C.c_uint64, []),
    "pv_arch": (C.c_char_p, []),
    "pv_error_string": (C.c_char_p, [C.c_int]),
    "pv_cast_f32_bf16": (C.c_int, [_p, _p, _i64, _p]),
    "pv_im2col_bf16": (C.c_int, [_p, _p, _i64, _i64, _i64, _i64, _i64, _p, _p]),
    "pv_patch_embed_f32": (C.c_int, [_p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _i64, _i64, _i64, _p, _p]),
    "pv_im2col_u8_bf16": (C.c_int, [_p, _p, _i64, _i64, _i64, _i64, _f32, _f32, _f32, _f32, _f32, _f32, _p]),
    "pv_split3_f32_bf16": (C.c_int, [_p, _p, _i64, _i64, C.c_int, _p]),
    "pv_im2col_split_bf16": (C.c_int, [_p, _p, _i64, _i64, _i64, _i64, _i64, _p]),
    "pv_layernorm_split_bf16": (C.c_int, [_p, _i64, _p, _p, _p, _p, _i64, _i64, _f32, _p]),
    "pv_attention_f32_split": (C.c_int, [_p, _p, _i64, _i64, _i64, _i64, _p]),
    "pv_attention_split_bf16": (C.c_int, [_p, _p, _i64, _i64, _i64, _i64, _p]),
    "pv_workspace_size": (C.c_int64, [C.c_int, C.POINTER(C.c_int64), C.c_int]),
    "pv_gemm_tn_bf16": (C.c_int, [C.POINTER(GemmArgs), _p]),
    "pv_sum_slices_f32": (C.c_int, [_p, _p, _i64, _i64, C.c_int, _p]),
    "pv_sum_slices_add_f32": (C.c_int, [_p, _p, _p, _i64, _i64, _p]),
    "pv_sum_slices_act_bf16": (C.c_int, [_p, _p, _i64, _i64, _i64, C.c_int, _i64, _f32, _p, _p]),
    "pv_sum_slices_add_ln_f32": (C.c_int, [_p, _p, _p, _i64, _i64, _i64, _p, _p, _f32, _p, _p]),
    "pv_transpose_bf16": (C.c_int, [_p, _i64, _p, _i64, _i64, _i64, _p, _p, _p]),
    "pv_layernorm_bwd": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _f32, C.c_int, _p]),
    "pv_layernorm_bwd16": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _f32, C.c_int, _p]),
    "pv_layernorm_bwd_masked": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, C.c_int, _p, _p, C.c_int, _p, _i64, _i64, _i64, _f32, _p]),
    "pv_masked_residual": (C.c_int, [_p, _p, _p, _p, _i64, _i64, _p]),
    "pv_gelu_bf16": (C.c_int, [_p, _p, _i64, _p]),
    "pv_gelu_bwd_bf16": (C.c_int, [_p, _p, _p, _i64, _p]),
    "pv_colsum_f32": (C.c_int, [_p, C.c_int, _p, _p, _i64, _i64, C.c_int, _p]),
    "pv_scatter_tokens": (C.c_int, [_p, _p, _p, _i64, _i64, _i64, _i64, _p]),
    "pv_attention_bwd_bf16": (C.c_int, [_p, _p, _p, _p, _i64, _i64, _i64, _i64, _f32, _p]),
    "pv_attention_lse_bf16": (C.c_int, [_p, _p, _p, _i64, _i64, _i64, _i64, _p, _p]),
    "pv_attention_bwd_lse_bf16": (C.c_int, [_p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i64, _f32, _p]),
    "pv_token_prologue": (C.c_int, [_p, _p, _p, _p, _f32, _i64, _i64, _i64, _i64, _p]),
    "pv_layernorm_bf16": (C.c_int, [_p, _i64, _p, _p, _p, _p, _i64, _i64, _f32, _p]),
    "pv_operand_type": (C.c_int, []),
    "pv_gemm_bf16": (C.c_int, [C.POINTER(GemmArgs), _p]),
    "pv_gemm_tile_rows": (C.c_int, [C.POINTER(GemmArgs)]),
    "pv_rowstat_finalize": (C.c_int, [_p, _p, _i64, _i64, _i64, _f32, _p, _p]),
    "pv_attention_bf16": (C.c_int, [_p, _p, _i64, _i64, _i64, _i64, _p, _p]),
    "pv_attention_rows_bf16": (C.c_int, [_p, _i64, _p, _i64, _p, _i64, _i64, _i64, _i64, _i64, _i64, _p, _p]),
    "pv_attention_rows_bwd_bf16": (C.c_int, [_p, _i64, _p, _i64, _p, _i64, _p, _i64, _p, _i64, _p, _i64, _i64, _i64, _i64, _i64, _i64, _f32, _p]),
    "pv_cls_pool": (C.c_int, [_p, _p, _p, _p, _i64, _i64, _i64, _i64, _f32, _p]),
    "pv_head_f32": (C.c_int, [_p, _p, _p, _p, _i64, _i64, _i64, _p]),
    "pv_token_norm": (C.c_int, [_p, _p, _i64, _i64, _i64, _p]),
    "pv_rank_topk": (C.c_int, [_p, _p, _i64, _i64, _i64, _p]),
    "pv_rank_topk_partials": (C.c_int, [_p, _i64, _p, _i64, _i64, _i64, _p]),
    "pv_rank_topk_gap": (C.c_int, [_p, _p, _p, _i64, _i64, _i64, _p]),
    "pv_rank_topk_partials_gap": (C.c_int, [_p, _i64, _p, _p, _i64, _i64, _i64, _p]),
    "pv_gather_tokens": (C.c_int, [_p, _p, _p, _i64, _i64, _i64, _i64, _p]),
    "pv_residual_gate": (C.c_int, [_p, _p, _p, _p, _p, _p, _f32, _f32, _p, _p, _p, _p, _p, _f32, _p, _i64, _i64, _i64, _p]),
    "pv_residual_gate_bwd": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _f32, _f32, _p, _p, _p, _p, _i64, _i64, _i64, _p]),
}

ABI_VERSION = 10
_lock = threading.Lock()
_libs: dict = {}

# 16-bit operand type of the MFMA products (engine.precision): "bf16" -> libpeekvit_hip.so, "f16" -> libpeekvit_hip_f16.so (the
# same sources built with -DPV_OPERAND_F16).  ops.py / engine.py allocate operand tensors with operand_dtype().
# The selection is PER THREAD (engine.precision() switches it around a forward; another thread in the middle of its own forward must not see
# the switch): `_lib.OPERAND` reads the calling thread's value (module __getattr__), set_operand() writes it.
_OPERAND_DEFAULT = "bf16"
_tls = threading.local()
LIB_F16 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libpeekvit_hip_f16.so")


def current_operand() -> str:
    return getattr(_tls, "operand", _OPERAND_DEFAULT)


def set_operand(op: str) -> str:
    """Set the calling thread's operand type; returns the previous one."""
    old = current_operand()
    _tls.operand = op
    return old


def __getattr__(name):
    if name == "OPERAND":
        return current_operand()
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")


def operand_dtype():
    import torch as _t
    return _t.float16 if current_operand() == "f16" else _t.bfloat16


class PeekvitHipError(RuntimeError):
    pass


def lib_path(operand: str = "bf16") -> str:
    return LIB_F16 if operand == "f16" else LIB


def load(operand=None):
    """Load the shared library of the current (or given) operand type once and attach the declared signatures.  Raises if missing."""
    op = current_operand() if operand is None else operand
    lib = _libs.get(op)
    if lib is not None:
        return lib
    with _lock:
        lib = _libs.get(op)
        if lib is not None:
            return lib
        path = lib_path(op)
        if not os.path.exists(path):
            raise PeekvitHipError(
                f"{path} not found: the MI355X kernels are not built. Run `python -m peekvit_amd._build` "
                "(or __graft_entry__.build()); there is no fallback path.")
        lib = C.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)          # AttributeError if the symbol is not exported
            fn.restype, fn.argtypes = res, args
        if lib.pv_version() != ABI_VERSION:
            raise PeekvitHipError(f"{path} has ABI v{lib.pv_version()}, this package binds v{ABI_VERSION}: rebuild (python -m peekvit_amd._build)")
        if lib.pv_gemm_args_size() != C.sizeof(GemmArgs):
            raise PeekvitHipError(f"{path}: pv_gemm_args is {lib.pv_gemm_args_size()} bytes in the library, {C.sizeof(GemmArgs)} in this binding")
        if lib.pv_operand_type() != (1 if op == "f16" else 0):
            raise PeekvitHipError(f"{path} was built for a different operand type")
        _libs[op] = lib
    return lib


def check(code: int, what: str = "") -> None:
    if code != 0:
        msg = load().pv_error_string(code).decode()
        raise PeekvitHipError(f"{what}: {msg} (code {code})" if what else f"{msg} (code {code})")
