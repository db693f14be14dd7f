// ABI identification and error strings of libpeekvit_hip.so.
#include "pv_common.h"

extern "C" int pv_version(void) { return 7; }
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
