// Diagnostic (not part of the library): what the global -> LDS path of a CU sustains with NOTHING else running - the bound of the 256^2 GEMM's K loop
// (DESIGN.md section 10).  One 512-thread workgroup per CU issues 1-KiB LDS-DMA pieces (global_load_lds_dwordx4, 8 rows x 128 B with the GEMM's
// source swizzle) back to back into a 128 KiB ring, throttled by a counted s_waitcnt vmcnt like the K loop, from
//   (a) a window small enough to stay in the XCD's L2 (the W tiles + A panels a group of tiles shares),  (b) a 128 MiB window (Infinity Cache),
//   (c) a 4 GiB stream (HBM).   Reports bytes per cycle per CU (s_memtime) and the chip-wide rate.
//   hipcc --offload-arch=gfx950 -O3 scripts/lds_dma_rate.hip -o scripts/bin/lds_dma_rate && scripts/bin/lds_dma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

__global__ __launch_bounds__(512) void dma_spin(const char* __restrict__ src, size_t window, size_t stride_cu, int iters, unsigned long long* out, int inflight16) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // piece p of an iteration: rows p*64 + wid*8 + lane/8 of a 512-row x 128-B "K-tile" (64 KiB per workgroup and iteration)
    const size_t lane_off = (size_t)(wid * 8 + (lane >> 3)) * 128 + (size_t)(((lane & 7) ^ ((lane >> 3) & 7)) * 16);
    const size_t mask = window - 1, start = (blockIdx.x * stride_cu) & mask;          // (window is a power of two)
    unsigned long long t0, t1;
    __syncthreads();
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    size_t pos = start;
    for (int it = 0; it < iters; ++it) {
        char* l = smem + (it & 1) * 65536 + wid * 1024;
#pragma unroll
        for (int p = 0; p < 8; ++p)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + ((pos + (size_t)p * 8192 + lane_off) & mask)),
                                             (__attribute__((address_space(3))) void*)(l + p * 8192), 16, 0, 0);
        pos += 65536;
        if (inflight16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");      // two iterations of pieces in flight (the K loop keeps 4 - 8 per wave)
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

int main() {
    const size_t big = (size_t)4 << 30;
    char* src; unsigned long long* d;
    hipMalloc(&src, big); hipMemset(src, 1, big); hipMalloc(&d, 256 * 8);
    hipFuncSetAttribute((const void*)dma_spin, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    struct Case { const char* name; size_t window, stride; } cases[] = {
        {"L2-resident: every CU re-reads the same 2 MiB", (size_t)2 << 20, 0},
        {"L2-resident, per-XCD-ish: 8 windows of 2 MiB", (size_t)16 << 20, (size_t)2 << 20},
        {"Infinity Cache: 128 MiB window, CUs spread over it", (size_t)128 << 20, (size_t)512 << 10},
        {"HBM: 4 GiB stream, each CU its own 16 MiB", big, (size_t)16 << 20}};
    for (auto& c : cases)
        for (int inflight16 = 0; inflight16 < 2; ++inflight16) {
            const int iters = 2000;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(dma_spin, dim3(256), dim3(512), 131072, 0, src, c.window, c.stride, iters, d, inflight16);
                hipEventRecord(e1);
            }
            hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> h(256);
            hipMemcpy(h.data(), d, 256 * 8, hipMemcpyDeviceToHost);
            std::sort(h.begin(), h.end());
            const double bytes_cu = (double)iters * 65536.0;
            printf("%-52s vmcnt(%2d): %5.1f B/clk/CU (median CU), %6.2f TB/s chip-wide, %.0f cycles per 64 KiB\n", c.name, inflight16 ? 16 : 8, bytes_cu / (double)h[128],
                   256.0 * bytes_cu / (ms * 1e-3) / 1e12, (double)h[128] / iters);
        }
    return 0;
}
