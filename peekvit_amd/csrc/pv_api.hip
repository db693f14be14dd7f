// ABI identification and error strings of libpeekvit_hip.so.
#include "pv_common.h"

extern "C" int pv_version(void) { return 10; }
extern "C" uint64_t pv_gemm_args_size(void) { return (uint64_t)sizeof(pv_gemm_args); }
extern "C" const char* pv_arch(void) { return "gfx950"; }
extern "C" int pv_operand_type(void) { return PV_OPERAND_CODE; }      // 0 = bf16 operands, 1 = fp16 operands (libpeekvit_hip_f16.so)
extern "C" const char* pv_error_string(int code) {
    switch (code) {
        case PV_OK: return "ok";
        case PV_ERR_INVALID_ARG: return "invalid argument (null pointer, bad size, misaligned pointer or leading dimension)";
        case PV_ERR_UNSUPPORTED: return "shape not supported by the gfx950 kernels";
        case PV_ERR_LAUNCH: return "HIP kernel launch failed";
        default: return "unknown error";
    }
}

// Scratch sizes of the entry points that take caller-provided workspaces (include/peekvit_hip.h, PV_WS_*): ONE place for the formulas the
// kernels' launchers assume (pv_rowops.hip: chunk height 64 rows up to 65536 rows, else 1024; LayerNorm backward: min(ceil(rows / 4), 1024)
// partial blocks of 3 * D floats; pv_gemm.hip: one row of column sums per 256-row tile).
extern "C" int64_t pv_workspace_size(int use, const int64_t* dims, int ndims) {
    if (!dims || ndims <= 0) return PV_ERR_INVALID_ARG;
    for (int i = 0; i < ndims; ++i)
        if (dims[i] <= 0) return PV_ERR_INVALID_ARG;
    switch (use) {
        case PV_WS_TRANSPOSE_COLSUM: {
            if (ndims != 3) return PV_ERR_INVALID_ARG;
            const int64_t C = dims[1], ldd = dims[2], q = ldd <= 65536 ? 64 : 1024;
            return (ldd + q - 1) / q * C * 4;
        }
        case PV_WS_COLSUM: {
            if (ndims != 2) return PV_ERR_INVALID_ARG;
            const int64_t R = dims[0], C = dims[1], q = R <= 65536 ? 64 : 1024;
            return (R + q - 1) / q * C * 4;
        }
        case PV_WS_LAYERNORM_BWD: {
            if (ndims != 2) return PV_ERR_INVALID_ARG;
            int64_t blocks = (dims[0] + 3) / 4;
            if (blocks > 1024) blocks = 1024;
            return blocks * 3 * dims[1] * 4;
        }
        case PV_WS_GEMM_COLSUM_PARTIAL:
            if (ndims != 2) return PV_ERR_INVALID_ARG;
            return (dims[0] + 255) / 256 * dims[1] * 4;
        case PV_WS_GEMM_SPLITK:
            if (ndims != 3) return PV_ERR_INVALID_ARG;
            return dims[2] * dims[0] * dims[1] * 4;
        default: return PV_ERR_INVALID_ARG;
    }
}
