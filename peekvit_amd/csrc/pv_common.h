// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels.  wave = 64 lanes, hard-coded.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/peekvit_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define PV_WAVE 64

// fp32 -> bf16 bits, round-to-nearest-even; a plain cast lowers to v_cvt_pk_bf16_f32 on gfx950 and keeps NaN a NaN.
__device__ __forceinline__ uint16_t pv_f2bf(float f) {
    __bf16 h = (__bf16)f;
    return __builtin_bit_cast(uint16_t, h);
}
__device__ __forceinline__ uint32_t pv_pack_bf16x2(float lo, float hi) {
    return (uint32_t)pv_f2bf(lo) | ((uint32_t)pv_f2bf(hi) << 16);
}
__device__ __forceinline__ float pv_bf2f(uint16_t b) {
    return __builtin_bit_cast(float, (uint32_t)b << 16);
}

__device__ __forceinline__ float pv_wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float pv_wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

// exact (erf) GELU, the default of F.gelu used at models/blocks.py:82
__device__ __forceinline__ float pv_gelu_erf(float x) {
    return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
}

// hipGetLastError() is per-thread and sticky across ALL users of the runtime (PyTorch leaves benign errors such as
// failed pointer-attribute queries behind), so every launch first clears it: pv_check_launch() then reports OUR launch.
#define PV_LAUNCH(...) do { (void)hipGetLastError(); hipLaunchKernelGGL(__VA_ARGS__); } while (0)

static inline int pv_check_launch() {
    return hipGetLastError() == hipSuccess ? PV_OK : PV_ERR_LAUNCH;
}

// grid sizing for HBM-bound row kernels: enough blocks to fill 256 CUs x 8, grid-stride the rest
static inline unsigned pv_stream_grid(int64_t work_items, int items_per_block) {
    int64_t blocks = (work_items + items_per_block - 1) / items_per_block;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (blocks < 1) blocks = 1;
    return (unsigned)blocks;
}
