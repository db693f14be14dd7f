// Diagnostic (not part of the library): the EMPIRICAL MFMA roofline of one MI355X under its package power cap, as a function of how many
// bytes move per FLOP.  Every token GEMM of this repo holds the package at 1400 W with the clock pulled to 1.8-2.0 GHz,
// so "fraction of 2.5 PFLOP/s" says little about the kernel: the question is what ANY kernel can sustain at the same LDS / L2 / HBM traffic
// per MFMA.  This program measures exactly that with a dependency-free loop:
//   per trip and wave: 16 x v_mfma_f32_16x16x32_bf16 on register operands (8 independent accumulators, operands that toggle like real data)
//                      + LDS  ds_read_b128 per lane            (-l N, a 256^2 GEMM tile needs 6 per 16 MFMAs)
//                      + L2   global_load_dwordx4 per lane from a small per-XCD-resident window (-c N, 1 KiB per wave and load)
//                      + HBM  global_load_dwordx4 per lane from a 4 GiB stream, every byte touched once (-m N per 8 trips)
//   loaded values BECOME the MFMA operands (no VALU work on them), so nothing is dead code and the operand paths toggle like real data.
// One 8-wave workgroup per CU (the GEMM's occupancy).  Prints TFLOP/s, LDS / L2 / HBM GB/s; power and sclk come from the hwmon sampler
// of scripts/power_roofline.py, which sweeps the mix.
//   hipcc --offload-arch=gfx950 -O3 scripts/power_roofline.hip -o scripts/bin/power_roofline
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int NL, int NC, int NM>      // LDS reads / L2 loads per trip, HBM loads per 8 trips
__global__ __launch_bounds__(512) void mix_kernel(const u32x4* __restrict__ l2win, const u32x4* __restrict__ stream, size_t stream_vecs, float* out, int trips) {
    __shared__ u32x4 lds[4096];        // 64 KiB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned h = tid * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int i = tid; i < 4096; i += 512) {
        h = h * 1664525u + 1013904223u;
        u32x4 v;
        for (int k = 0; k < 4; ++k) {
            h = h * 1664525u + 1013904223u;
            const unsigned lo = (h & 0x807fu) | ((120u + ((h >> 8) & 7u)) << 7), hi = ((h >> 16) & 0x807fu) | ((120u + ((h >> 28) & 7u)) << 7);
            v[k] = lo | (hi << 16);
        }
        lds[i] = v;                        // finite bf16 pairs with random sign / mantissa
    }
    __syncthreads();
    u32x4 av[8], bv[8];
    for (int j = 0; j < 8; ++j) {
        h = h * 1664525u + 1013904223u;
        av[j] = lds[(h >> 4) & 4095];
        h = h * 1664525u + 1013904223u;
        bv[j] = lds[(h >> 4) & 4095];
    }
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // this wave's private slice of the HBM stream: consecutive 1 KiB pieces, never revisited
    const size_t waves = (size_t)gridDim.x * 8, wv = (size_t)blockIdx.x * 8 + wave;
    const size_t per_wave = stream_vecs / waves / 64 * 64;
    const u32x4* sp = stream + wv * per_wave + lane;
    size_t spos = 0;
    // L2 window: 2 MiB, the waves of a workgroup walk it with a stride (hits after the first pass)
    const u32x4* cp = l2win + lane;
    unsigned cpos = (blockIdx.x * 977u + wave * 131u) & 2047u;       // in 1 KiB pieces (2048 pieces = 2 MiB)
    unsigned lpos = (wave * 512 + lane) & 4095;
    // Loaded values BECOME MFMA operands (no VALU work on them, as in a GEMM, where fragments go LDS -> registers -> MFMA): LDS reads replace
    // A operands and are used by the next trip; L2 / HBM loads land in pending registers and replace B operands one period later (4 / 8 trips),
    // like the prefetch distance of a pipelined GEMM, so a wait never sits right behind its own load.
    u32x4 pc[NC > 0 ? 4 * NC : 1], pm[NM > 0 ? NM : 1];
    for (int i = 0; i < (NC > 0 ? 4 * NC : 1); ++i) pc[i] = bv[i & 7];
    for (int i = 0; i < (NM > 0 ? NM : 1); ++i) pm[i] = bv[i & 7];
    for (int t = 0; t < trips; ++t) {
#pragma unroll
        for (int i = 0; i < 16; ++i)
            acc[i & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av[i & 7]), __builtin_bit_cast(bf16x8, bv[(i + (i >> 3)) & 7]), acc[i & 7], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            av[i & 7] = lds[lpos];
            lpos = (lpos + 64 * 8 + 1) & 4095;
        }
        if (NC > 0 && (t & 3) == 3) {
#pragma unroll
            for (int i = 0; i < 4 * NC; ++i) {
                asm volatile("" : "+v"(pc[i]));            // the load has to have arrived (and is never dead code), at no instruction cost
                if (i < 4) bv[i] = pc[i];
                pc[i] = cp[(size_t)cpos * 64];
                cpos = (cpos + 37u) & 2047u;
            }
        }
        if (NM > 0 && (t & 7) == 7) {
#pragma unroll
            for (int i = 0; i < NM; ++i) {
                asm volatile("" : "+v"(pm[i]));
                if (i < 4) bv[4 + i] = pm[i];
                pm[i] = __builtin_nontemporal_load(sp + spos);
                spos += 64;
                if (spos >= per_wave) spos = 0;
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678f) out[0] = s;
}

__global__ void fill_kernel(u32x4* p, size_t n) {          // pseudo-random bits: memory and fabric toggle like real activations
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u + 12345u;
        u32x4 v;
        for (int k = 0; k < 4; ++k) {      // two bf16 per dword: random sign and mantissa, exponent 120 .. 127 (finite, |v| in [2^-7, 2))
            h = h * 1664525u + 1013904223u;
            const unsigned lo = (h & 0x807fu) | ((120u + ((h >> 8) & 7u)) << 7), hi = ((h >> 16) & 0x807fu) | ((120u + ((h >> 28) & 7u)) << 7);
            v[k] = lo | (hi << 16);
        }
        p[i] = v;
    }
}

template <int NL, int NC, int NM>
static void run(const u32x4* l2win, const u32x4* stream, size_t vecs, float* out, int trips, double seconds) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((mix_kernel<NL, NC, NM>), dim3(256), dim3(512), 0, 0, l2win, stream, vecs, out, trips);     // warm up
    hipDeviceSynchronize();
    double total_ms = 0; int launches = 0;
    float best = 1e30f;
    while (total_ms < seconds * 1e3) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((mix_kernel<NL, NC, NM>), dim3(256), dim3(512), 0, 0, l2win, stream, vecs, out, trips);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        total_ms += ms; ++launches; best = ms < best ? ms : best;
    }
    const double ms = total_ms / launches;                       // the AVERAGE launch (DVFS settles over the run), not the best
    const double waves = 256.0 * 8, t = trips;
    const double flops = waves * t * 16 * 16384.0, lds_b = waves * t * NL * 1024.0, l2_b = waves * t * NC * 1024.0, hbm_b = waves * (t / 8) * NM * 1024.0;
    printf("{\"lds_per_16mfma\": %d, \"l2_per_16mfma\": %d, \"hbm_per_128mfma\": %d, \"ms\": %.3f, \"tflops\": %.1f, \"lds_GBps\": %.0f, \"l2_GBps\": %.0f, \"hbm_GBps\": %.0f, "
           "\"bytes_per_kflop_hbm\": %.3f}\n",
           NL, NC, NM, ms, flops / ms / 1e9, lds_b / ms / 1e6, l2_b / ms / 1e6, hbm_b / ms / 1e6, hbm_b / flops * 1e3);
    fflush(stdout);
}

int main(int argc, char** argv) {
    int nl = 0, nc = 0, nm = 0, trips = 20000; double seconds = 1.5;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "-l")) nl = atoi(argv[++i]);
        else if (!strcmp(argv[i], "-c")) nc = atoi(argv[++i]);
        else if (!strcmp(argv[i], "-m")) nm = atoi(argv[++i]);
        else if (!strcmp(argv[i], "-t")) trips = atoi(argv[++i]);
        else if (!strcmp(argv[i], "-s")) seconds = atof(argv[++i]);
    }
    const size_t stream_bytes = 4ull << 30, vecs = stream_bytes / 16;
    u32x4 *stream, *l2win; float* out;
    if (hipMalloc(&stream, stream_bytes) != hipSuccess || hipMalloc(&l2win, 2 << 20) != hipSuccess || hipMalloc(&out, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, stream, vecs);
    hipLaunchKernelGGL(fill_kernel, dim3(64), dim3(256), 0, 0, l2win, (size_t)(2 << 20) / 16);
    hipDeviceSynchronize();
#define CASE(L, C, M) if (nl == L && nc == C && nm == M) { run<L, C, M>(l2win, stream, vecs, out, trips, seconds); return 0; }
    CASE(0, 0, 0) CASE(6, 0, 0) CASE(6, 2, 0) CASE(6, 2, 2) CASE(6, 2, 4) CASE(6, 2, 6) CASE(6, 2, 8) CASE(6, 2, 10) CASE(6, 2, 12) CASE(6, 2, 16)
    CASE(0, 0, 8) CASE(0, 0, 16) CASE(6, 0, 8) CASE(3, 1, 8) CASE(6, 1, 4) CASE(6, 1, 8)
    printf("mix -l %d -c %d -m %d not compiled in\n", nl, nc, nm);
    return 2;
}
