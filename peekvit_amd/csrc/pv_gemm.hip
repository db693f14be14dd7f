// bf16 MFMA GEMM with fused epilogues for the ViT encoder linears (gfx950).
//   out = epilogue(A[M,K] . W[N,K]^T)   A = activations (bf16), W = nn.Linear weight layout (out,in) (bf16)
// Both operands are K-contiguous, so they are staged identically: LDS-DMA (global_load_lds_dwordx4) into a
// lane-linear LDS image whose 16-byte chunks are XOR-swizzled on the SOURCE address (chunk ^ (row & 7)),
// read back with conflict-free ds_read_b128 through the same involution.
// The MFMA is issued "swapped" (A-operand = W rows, B-operand = activation rows) so that each lane's four
// accumulator registers are four CONSECUTIVE output columns of one output row: the epilogue then loads
// bias/residual and stores the result with 8-byte (bf16) / 16-byte (fp32) vector accesses.
#include "pv_common.h"

struct GemmDev {
    const uint16_t* A;
    const uint16_t* W;
    const float* bias;
    void* out;
    const float* res;
    const float* row_scale;
    const float* pos;
    int M, N, K;
    int64_t lda, ldw, ldo, ldr;
    int rpi, rpo, row_off, qcols;
    float qscale;
    int tiles_m, tiles_n;
};

__device__ __forceinline__ void pv_glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// XCD-aware bijective remap (8 XCDs, blocks b and b+8 share an XCD): every XCD walks a CONTIGUOUS range of the
// tile list (n fastest), so co-resident blocks of one XCD share A panels and the weight matrix in that XCD's L2.
__device__ __forceinline__ int pv_xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, i = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
}

template <int EPI>
__device__ __forceinline__ void pv_epilogue_store(const GemmDev& p, int m, int n, f32x4 acc) {
    if (m >= p.M || n >= p.N) return;
    float4 b = p.bias ? *reinterpret_cast<const float4*>(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
    float v0 = acc[0] + b.x, v1 = acc[1] + b.y, v2 = acc[2] + b.z, v3 = acc[3] + b.w;
    if (EPI == PV_EPI_BIAS_BF16) {
        const float s = n < p.qcols ? p.qscale : 1.0f;
        u32x2 o = {pv_pack_bf16x2(v0 * s, v1 * s), pv_pack_bf16x2(v2 * s, v3 * s)};
        *reinterpret_cast<u32x2*>(reinterpret_cast<uint16_t*>(p.out) + (int64_t)m * p.ldo + n) = o;
    } else if (EPI == PV_EPI_BIAS_GELU_BF16) {
        u32x2 o = {pv_pack_bf16x2(pv_gelu_erf(v0), pv_gelu_erf(v1)), pv_pack_bf16x2(pv_gelu_erf(v2), pv_gelu_erf(v3))};
        *reinterpret_cast<u32x2*>(reinterpret_cast<uint16_t*>(p.out) + (int64_t)m * p.ldo + n) = o;
    } else if (EPI == PV_EPI_BIAS_RES_F32) {
        const float s = p.row_scale ? p.row_scale[m] : 1.0f;
        float4 r = *reinterpret_cast<const float4*>(p.res + (int64_t)m * p.ldr + n);
        float4 o = make_float4(r.x + s * v0, r.y + s * v1, r.z + s * v2, r.w + s * v3);
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + (int64_t)m * p.ldo + n) = o;
    } else {   // PV_EPI_BIAS_POS_F32
        const int img = m / p.rpi, pi = m - img * p.rpi;
        const int64_t orow = (int64_t)img * p.rpo + p.row_off + pi;
        float4 r = *reinterpret_cast<const float4*>(p.pos + (int64_t)(p.row_off + pi) * p.N + n);
        float4 o = make_float4(r.x + v0, r.y + v1, r.z + v2, r.w + v3);
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + orow * p.ldo + n) = o;
    }
}

// ------------------------------------------------------------------------------------------------
// 128 x 128 x 64 tile, 4 waves (2 x 2), 64 x 64 per wave = 4 x 4 MFMA 16x16x32 tiles, 2 LDS buffers (64 KiB)
// ------------------------------------------------------------------------------------------------
constexpr int G1_BM = 128, G1_BN = 128, G1_BK = 64;
constexpr int G1_TILE_BYTES = G1_BM * G1_BK * 2;   // 16 KiB per operand per stage

template <int EPI>
__global__ __launch_bounds__(256, 2) void pv_gemm128_kernel(const GemmDev p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wn = wid & 1;

    const int tile = pv_xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n);
    const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
    const int m0 = tm * G1_BM, n0 = tn * G1_BN;

    // ---- staging: thread -> (row, swizzled chunk) of the lane-linear LDS image -------------------------------
    const int srow = wid * 8 + (lane >> 3);                      // + 32 * i
    const int schunk = (lane & 7) ^ ((lane >> 3) & 7);           // logical 16-B chunk this lane fetches
    const uint16_t* ga[4];
    const uint16_t* gw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int ra = m0 + i * 32 + srow; ra = ra < p.M ? ra : p.M - 1;
        int rw = n0 + i * 32 + srow; rw = rw < p.N ? rw : p.N - 1;
        ga[i] = p.A + (int64_t)ra * p.lda + schunk * 8;
        gw[i] = p.W + (int64_t)rw * p.ldw + schunk * 8;
    }
    auto stage = [&](int buf, int kt) {
        char* la = smem + buf * (2 * G1_TILE_BYTES) + wid * 1024;
        char* lw = la + G1_TILE_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            pv_glds16(ga[i] + kt * G1_BK, la + i * 4096);
            pv_glds16(gw[i] + kt * G1_BK, lw + i * 4096);
        }
    };

    // ---- fragment read addresses (bytes inside one operand tile) ----------------------------------------------
    const int frow = lane & 15;
    const int fx0 = (((lane >> 4) ^ (lane & 7)) << 4);           // k-step 0; k-step 1 = fx0 ^ 64
    const int a_off = (wm * 64 + frow) * 128;                    // activation rows of this wave
    const int w_off = (wn * 64 + frow) * 128;                    // weight rows of this wave

    f32x4 acc[4][4];   // [nt][mt]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / G1_BK;
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
        const char* la = smem + cur * (2 * G1_TILE_BYTES);
        const char* lw = la + G1_TILE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int fx = fx0 ^ (ks << 6);
            bf16x8 xf[4], wf[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                xf[t] = *reinterpret_cast<const bf16x8*>(la + a_off + t * 2048 + fx);
                wf[t] = *reinterpret_cast<const bf16x8*>(lw + w_off + t * 2048 + fx);
            }
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], xf[mt], acc[nt][mt], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // ---- epilogue: lane holds out[m = ..+(lane&15)][n = ..+(lane>>4)*4 + 0..3] ---------------------------------
    const int em = m0 + wm * 64 + (lane & 15);
    const int en = n0 + wn * 64 + ((lane >> 4) << 2);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) pv_epilogue_store<EPI>(p, em + mt * 16, en + nt * 16, acc[nt][mt]);
}

template <int EPI>
static int pv_launch_gemm128(const GemmDev& p, hipStream_t stream) {
    static bool attr_set = false;
    const int lds = 2 * 2 * G1_TILE_BYTES;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pv_gemm128_kernel<EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    PV_LAUNCH(pv_gemm128_kernel<EPI>, dim3((unsigned)(p.tiles_m * p.tiles_n)), dim3(256), lds, stream, p);
    return pv_check_launch();
}

extern "C" int pv_gemm_bf16(const pv_gemm_args* a, void* stream) {
    if (!a || !a->A || !a->W || !a->out || a->M <= 0 || a->N <= 0 || a->K <= 0) return PV_ERR_INVALID_ARG;
    if (a->K % 64 || a->N % 4) return PV_ERR_UNSUPPORTED;
    if (a->lda % 8 || a->ldw % 8 || a->ldo % 4 || a->lda < a->K || a->ldw < a->K || a->ldo < a->N) return PV_ERR_INVALID_ARG;
    if (((uintptr_t)a->A & 15) || ((uintptr_t)a->W & 15) || ((uintptr_t)a->out & 15) || (a->bias && ((uintptr_t)a->bias & 15))) return PV_ERR_INVALID_ARG;
    if (a->M > 0x7fffffff || a->N > 0x7fffffff || a->K > 0x7fffffff) return PV_ERR_UNSUPPORTED;
    GemmDev p;
    p.A = a->A; p.W = a->W; p.bias = a->bias; p.out = a->out; p.res = a->res; p.row_scale = a->row_scale; p.pos = a->pos;
    p.M = (int)a->M; p.N = (int)a->N; p.K = (int)a->K;
    p.lda = a->lda; p.ldw = a->ldw; p.ldo = a->ldo; p.ldr = a->ldr;
    p.rpi = (int)a->rows_per_img_in; p.rpo = (int)a->rows_per_img_out; p.row_off = (int)a->row_off;
    p.qcols = (int)a->qcols; p.qscale = a->qscale;
    p.tiles_m = (p.M + G1_BM - 1) / G1_BM; p.tiles_n = (p.N + G1_BN - 1) / G1_BN;
    if ((int64_t)p.tiles_m * p.tiles_n > 0x7fffffff) return PV_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    switch (a->epilogue) {
        case PV_EPI_BIAS_BF16: return pv_launch_gemm128<PV_EPI_BIAS_BF16>(p, s);
        case PV_EPI_BIAS_GELU_BF16: return pv_launch_gemm128<PV_EPI_BIAS_GELU_BF16>(p, s);
        case PV_EPI_BIAS_RES_F32:
            if (!a->res || a->ldr % 4 || a->ldr < a->N || ((uintptr_t)a->res & 15)) return PV_ERR_INVALID_ARG;
            return pv_launch_gemm128<PV_EPI_BIAS_RES_F32>(p, s);
        case PV_EPI_BIAS_POS_F32:
            if (!a->pos || a->rows_per_img_in <= 0 || a->rows_per_img_out < a->rows_per_img_in + a->row_off || a->row_off < 0 || ((uintptr_t)a->pos & 15))
                return PV_ERR_INVALID_ARG;
            return pv_launch_gemm128<PV_EPI_BIAS_POS_F32>(p, s);
        default: return PV_ERR_INVALID_ARG;
    }
}
