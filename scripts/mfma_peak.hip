// Diagnostic (not part of the library): sustained dense bf16 MFMA rate of the whole chip with NO memory traffic - registers only.
// hipcc --offload-arch=gfx950 -O3 [-DRANDOM_OPERANDS] scripts/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
__global__ __launch_bounds__(512) void mfma_spin(float* out, int iters) {
    // RANDOM_OPERANDS: eight different pseudo-random operand pairs per lane, so consecutive MFMAs toggle the operand paths as real
    // data does; otherwise one constant pair (the datapath barely toggles: lower power, higher clock)
    bf16x8 av[8], bv[8];
    unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int j = 0; j < 8; ++j)
        for (int i = 0; i < 8; ++i) {
            h = h * 1664525u + 1013904223u;
#ifdef RANDOM_OPERANDS
            av[j][i] = (__bf16)(((int)(h >> 8) % 2001 - 1000) * 1e-3f);
            h = h * 1664525u + 1013904223u;
            bv[j][i] = (__bf16)(((int)(h >> 8) % 2001 - 1000) * 1e-3f);
#else
            av[j][i] = (__bf16)(0.001f * (threadIdx.x + i)); bv[j][i] = (__bf16)(0.002f * (threadIdx.x - i));
#endif
        }
#ifdef SHAPE_32x32x16     // same FLOPs per pair of instructions, half the operand registers read per FLOP
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i], bv[i], acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
#else
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[i], bv[i], acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#endif
    if (s == 12345.678f) out[0] = s;
}
int main() {
    float* d; hipMalloc(&d, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 1, iters = 100000;       // one 8-wave workgroup per CU = 2 waves per SIMD
    for (int rep = 0; rep < 40; ++rep) {
        hipEventRecord(e0);
        for (int k = 0; k < 10; ++k) hipLaunchKernelGGL(mfma_spin, dim3(blocks), dim3(512), 0, 0, d, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
#ifdef SHAPE_32x32x16
        const double flops = 10.0 * blocks * 8.0 /*waves*/ * iters * 4.0 * 32768.0;   // 32x32x16 MACs x 2
#else
        const double flops = 10.0 * blocks * 8.0 /*waves*/ * iters * 8.0 * 16384.0;   // 16x16x32 MACs x 2
#endif
        printf("rep %2d: %.1f ms  %.1f TFLOP/s dense bf16 (registers only)\n", rep, ms, flops / ms / 1e9);
    }
    return 0;
}
