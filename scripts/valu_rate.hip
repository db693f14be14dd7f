// Diagnostic (not part of the library): vector-ALU issue rate of one CU's SIMDs on gfx950 for the instruction kinds of the GEMM's 16-bit
// epilogue, at 1 / 2 / 4 waves per SIMD: cycles per wave-instruction per SIMD (s_memtime around an unrolled stream of independent
// instructions, median over the workgroups).  Round 4: the fc1 epilogue measures ~4.7 cycles per instruction per SIMD with two waves
// per SIMD - is that the pipe (then only fewer instructions help) or stalls (then scheduling helps)?
//   hipcc --offload-arch=gfx950 -O3 scripts/valu_rate.hip -o scripts/bin/valu_rate && scripts/bin/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int KIND>
__global__ __launch_bounds__(1024) void spin(unsigned long long* out, float* sink, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float acc[16];
    f32x2 acc2[8];
    f32x4 g[4];
    const float a = 1.0f + 1e-7f * threadIdx.x, b = 1e-9f * (threadIdx.x + 1);
    const f32x2 a2 = {a, a}, b2 = {b, b};
    for (int i = 0; i < 16; ++i) acc[i] = 0.5f + i;
    for (int i = 0; i < 8; ++i) acc2[i] = (f32x2){0.5f + i, 1.5f + i};
    for (int i = 0; i < 4; ++i) g[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const unsigned lds_addr = (threadIdx.x & 15) * 16 + ((threadIdx.x * 37u) & 63u) * 256;       // a conflict-free table gather (replica = lane & 15)
    for (int i = threadIdx.x; i < 16384 / 4; i += blockDim.x) reinterpret_cast<float*>(smem)[i] = 1e-3f * i;
    __syncthreads();
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) {            // 64 independent v_fma_f32
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
        } else if (KIND == 1) {     // 64 independent v_pk_fma_f32 (natural operand halves)
#pragma unroll
            for (int r = 0; r < 8; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(acc2[i]) : "v"(a2), "v"(b2));
        } else if (KIND == 2) {     // 48 v_fma_f32 + 16 integer ops (v_med3_i32 / v_lshl_add_u32): the epilogue's index arithmetic
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int i = 0; i < 12; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
#pragma unroll
                for (int i = 12; i < 16; i += 2) {
                    asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
                    asm volatile("v_lshl_add_u32 %0, %0, 8, %1" : "+v"(acc[i + 1]) : "v"(b));
                }
            }
        } else if (KIND == 3) {     // 56 v_fma_f32 + 8 ds_read_b128 gathers (one per 8 instructions, as in the epilogue), counted wait at the end
#pragma unroll
            for (int r = 0; r < 8; ++r) {
#pragma unroll
                for (int i = 0; i < 7; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc[(r * 7 + i) & 15]) : "v"(a), "v"(b));
                asm volatile("ds_read_b128 %0, %1" : "=v"(g[r & 3]) : "v"(lds_addr + (r & 3) * 16));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(g[0]), "+v"(g[1]), "+v"(g[2]), "+v"(g[3]));
        } else if (KIND == 4) {     // 32 v_fma_f32 + 32 v_cvt_pk_bf16_f32 / v_max3: the pack + range-track share
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
#pragma unroll
                for (int i = 8; i < 16; i += 2) {
                    asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "+v"(acc[i]) : "v"(a), "v"(b));
                    asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(acc[i + 1]) : "v"(a), "v"(b));
                }
            }
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i];
    for (int i = 0; i < 8; ++i) s += acc2[i][0] + acc2[i][1];
    for (int i = 0; i < 4; ++i) s += g[i][0];
    if (s == 123.456f) sink[0] = s;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND>
static void run(const char* name, int threads) {
    unsigned long long* d; float* sink;
    hipMalloc(&d, 256 * 16 * 8); hipMalloc(&sink, 4);
    const int iters = 2000;
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(spin<KIND>, dim3(256), dim3(threads), 16384, 0, d, sink, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * 16);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> v;
    const int waves = threads / 64;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < waves; ++w) v.push_back((double)h[b * 16 + w]);
    std::sort(v.begin(), v.end());
    const double med = v[v.size() / 2], per_wave = med / (iters * 64.0), per_simd = per_wave / (waves / 4.0);
    printf("%-44s waves/SIMD %d: %6.2f cycles per instruction per wave, %5.2f per SIMD\n", name, waves / 4, per_wave, per_simd);
    hipFree(d); hipFree(sink);
}

int main() {
    for (int threads : {256, 512, 1024}) {
        run<0>("v_fma_f32", threads);
        run<1>("v_pk_fma_f32", threads);
        run<2>("3/4 v_fma_f32 + 1/4 v_med3_i32 / v_lshl_add_u32", threads);
        run<3>("7/8 v_fma_f32 + 1/8 ds_read_b128 gather", threads);
        run<4>("1/2 v_fma_f32 + 1/2 v_cvt_pk_bf16 / v_max3", threads);
    }
    return 0;
}
