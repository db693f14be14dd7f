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

// ---- the 16-bit OPERAND type of the MFMA products -----------------------------------------------------------------------
// Default: bf16.  Built with -DPV_OPERAND_F16 (libpeekvit_hip_f16.so, precision mode "f16") every operand tensor the kernels
// write or read - LayerNorm / GELU / attention outputs, q|k|v, weights, patch columns - is IEEE fp16 instead: same MFMA rate,
// 3 more mantissa bits (2^-11 vs 2^-8 rounding), range 6e-5 .. 65504 (activations after LayerNorm, weights and attention
// operands of a ViT sit well inside).  Accumulation, residual stream, LayerNorm / softmax / GELU arithmetic stay fp32.
// The helper names keep their "bf16" spelling; they convert to / from the operand type of the build.
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float pv_f32x2_t;
#ifdef PV_OPERAND_F16
typedef __attribute__((ext_vector_type(2))) _Float16 pv_h2_t;
typedef __attribute__((ext_vector_type(4))) _Float16 pv_h4_t;
typedef __attribute__((ext_vector_type(8))) _Float16 pv_h8_t;
#define PV_OPERAND_CODE 1
#define PV_MFMA_16x16x32(a, b, c, x_, y_, z_) __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(pv_h8_t, a), __builtin_bit_cast(pv_h8_t, b), c, 0, 0, 0)
__device__ __forceinline__ uint16_t pv_f2bf(float f) { return __builtin_bit_cast(uint16_t, (_Float16)f); }
__device__ __forceinline__ uint32_t pv_pack_bf16x2(float lo, float hi) {
    const pv_f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, pv_h2_t));      // round-to-nearest-even
}
__device__ __forceinline__ float pv_unpack_lo(uint32_t w) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(w & 0xffffu)); }
__device__ __forceinline__ float pv_unpack_hi(uint32_t w) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(w >> 16)); }
#else
#define PV_OPERAND_CODE 0
#define PV_MFMA_16x16x32(a, b, c, x_, y_, z_) __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0)
// fp32 -> bf16 bits, round-to-nearest-even; a plain cast lowers to v_cvt_pk_bf16_f32 on gfx950 and keeps NaN a NaN.
__device__ __forceinline__ uint16_t pv_f2bf(float f) {
    __bf16 h = (__bf16)f;
    return __builtin_bit_cast(uint16_t, h);
}
// two fp32 -> one dword of two bf16 (lo in bits 0-15): a vector convert lowers to ONE v_cvt_pk_bf16_f32
__device__ __forceinline__ uint32_t pv_pack_bf16x2(float lo, float hi) {
    const pv_f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ float pv_unpack_lo(uint32_t w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float pv_unpack_hi(uint32_t w) { return __builtin_bit_cast(float, w & 0xffff0000u); }
#endif
// ---- operand-range guard (ABI v5) --------------------------------------------------------------------------------------
// An fp16 operand overflows to +-inf above 65504 where bf16 would not.  The kernels of the fp16 build that turn DATA-DEPENDENT
// fp32 values into operands (GEMM epilogues that emit q|k|v and the GELU activations, the patch gather of a float image) keep
// the largest magnitude they pack in a register (one v_max3_f32 per two values) and, if it is not representable, OR 1 into the
// caller's `range_flag` word: the host then repeats that forward on the bf16 library (peekvit_amd.engine, mode "auto").  The
// other operand producers are bounded by construction and checked on the host once per parameter version: LayerNorm outputs
// (|y| <= max|gamma| * sqrt(D) + max|beta|), attention outputs (convex combinations of v rows), weights.  The bf16 build compiles
// the tracking away and never writes the flag.
#ifdef PV_OPERAND_F16
// v_maximum3_f32 (gfx950): IEEE-754-2019 maximum, a NaN operand gives NaN - v_max3_f32 / fmaxf drop it, and pv_range_commit's
// !(m <= 65504) then never saw a NaN the epilogue packed (round 2 ADVICE).  |a|, |b| are source modifiers: still ONE instruction per pair.
__device__ __forceinline__ void pv_range_track(float& m, float a, float b) { asm("v_maximum3_f32 %0, %1, |%2|, |%3|" : "=v"(m) : "v"(m), "v"(a), "v"(b)); }
__device__ __forceinline__ void pv_range_commit(float m, uint32_t* flag) {
    if (flag != nullptr && !(m <= 65504.0f)) atomicOr(flag, 1u);
}
#else
__device__ __forceinline__ void pv_range_track(float&, float, float) {}
__device__ __forceinline__ void pv_range_commit(float, uint32_t*) {}
#endif
// Attention-logit guard (ABI v7, bit 4 of the same flag word).  The error the 16-bit rounding of q and k leaves in a score grows with
// the score: delta_s ~ 2^-11 * sqrt(2) * |s| / sqrt(d_eff), and softmax turns it into a RELATIVE error of the probabilities of the
// keys that matter.  On Gaussian q, k (S = 197, d = 64) the attention output's relative error is 3.6e-4 at max|s| = 5, 8.6e-4 at 40,
// 1.1e-3 at 73, 1.6e-3 at 147 (fp16 operands) - beyond PV_SCORE_LIMIT the fp16 path cannot promise BASELINE's
// 1e-3, so the attention kernels of the fp16 build OR 4 into the flag when the magnitude of a row's largest score exceeds it, and the
// caller repeats the forward in a mode that keeps q, k in fp32 (engine mode "auto" -> "bf16x3").  One v_max per query tile.
#define PV_SCORE_LIMIT 32.0f
#ifndef PV_SCORE_GUARD
#define PV_SCORE_GUARD 1          // 0: compiled out (A/B, scripts/attn_ab.py)
#endif
// A row's maximum is tested WHERE IT IS COMPUTED (one compare per query tile, an atomic only when it trips).  Round 3 first carried a
// running maximum to the kernel's end: hipcc restructured the unrolled query-tile loop of the LDS-resident kernel around the extra
// live value - +6 % time, and (with the 2^14 probability scale) non-finite outputs in 0.4 % of the rows, a miscompile or a latent
// hazard that form exposed; profiles/r03_attention_ab.json.  The form below times and rounds exactly like the kernel without a guard.
#ifdef PV_OPERAND_F16
__device__ __forceinline__ void pv_score_guard(float row_max, uint32_t* flag) {
    if (PV_SCORE_GUARD && flag != nullptr && !(__builtin_fabsf(row_max) <= PV_SCORE_LIMIT)) atomicOr(flag, 4u);
}
#else
__device__ __forceinline__ void pv_score_guard(float, uint32_t*) {}
#endif
__device__ __forceinline__ uint32_t pv_pack_bf16x2_tracked(float lo, float hi, float& m) {
    pv_range_track(m, lo, hi);
    return pv_pack_bf16x2(lo, hi);
}

// split-precision helpers (precision mode "bf16x3"): v = hi + lo + O(2^-17 |v|) with hi = bf16(v), lo = bf16(v - hi)
struct PvHiLo { uint32_t hi, lo; };
__device__ __forceinline__ PvHiLo pv_split2(float a, float b) {
    PvHiLo r;
    r.hi = pv_pack_bf16x2(a, b);
    const float ah = pv_unpack_lo(r.hi), bh = pv_unpack_hi(r.hi);
    r.lo = pv_pack_bf16x2(a - ah, b - bh);
    return r;
}

__device__ __forceinline__ float pv_bf2f(uint16_t b) { return pv_unpack_lo((uint32_t)b); }

// K = 16 product (4 operand values per lane: lane group g supplies k = 4g .. 4g+3) on the K = 32 MFMA with the upper four slots
// of BOTH operands zero.  The native v_mfma_f32_16x16x16_bf16 is NOT used: hipcc (ROCm 7.2) allocates its destination over its
// A-operand registers (no early-clobber for the "4-pass" form), and on gfx950 that returned wrong values in exactly the
// overlapping registers (first seen in the attention backward: dq[2][0..1], run-to-run varying).  The zero-padded K = 32 form
// costs the same issue slots as the half-rate K = 16 instruction.
typedef __attribute__((ext_vector_type(8))) short pv_s16x8_t;
__device__ __forceinline__ bf16x8 pv_pad_k16(s16x4 v) {
    const s16x4 z = {0, 0, 0, 0};
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(v, z, 0, 1, 2, 3, 4, 5, 6, 7));
}
#define PV_MFMA_16x16x16(a, b, c, x_, y_, z_) PV_MFMA_16x16x32(pv_pad_k16(__builtin_bit_cast(s16x4, a)), pv_pad_k16(__builtin_bit_cast(s16x4, b)), c, 0, 0, 0)

// sum over the 16 lanes of a row (lanes 16r .. 16r+15) with DPP row operations - four full-rate VALU adds, no LDS crossbar:
// xor 1, xor 2 (quad_perm), then the two quads of each half (row_half_mirror), then the two halves (row_mirror).
__device__ __forceinline__ float pv_row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));    // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));    // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));   // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));   // row_mirror
    return v;
}
// wave-wide sum, the same value in every lane: DPP within the four 16-lane rows, then the four row sums through v_readlane
// (scalar registers) - no ds_bpermute round trips (the 6-step __shfl_xor butterfly costs six LDS-crossbar latencies).
__device__ __forceinline__ float pv_wave_sum(float v) {
    v = pv_row16_sum(v);
    const int vi = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(vi, 48));
    return (r0 + r1) + (r2 + r3);
}
// v[r] (r = 0..15) per lane -> the sum over all 64 lanes of v[r], delivered to lanes 4r .. 4r+3: a reduce-scatter that halves the
// number of rows a lane carries at every step (8 + 4 + 2 + 1 exchanges) and finishes with two plain steps: 17 cross-lane operations
// instead of 16 separate wave reductions.
__device__ __forceinline__ float pv_reduce16_rows(const float (&v)[16], int lane) {
    float t8[8], t4[4], t2[2];
    const bool b5 = lane & 32, b4 = lane & 16, b3 = lane & 8, b2 = lane & 4;
#pragma unroll
    for (int i = 0; i < 8; ++i) t8[i] = (b5 ? v[i + 8] : v[i]) + __shfl_xor(b5 ? v[i] : v[i + 8], 32, 64);
#pragma unroll
    for (int i = 0; i < 4; ++i) t4[i] = (b4 ? t8[i + 4] : t8[i]) + __shfl_xor(b4 ? t8[i] : t8[i + 4], 16, 64);
#pragma unroll
    for (int i = 0; i < 2; ++i) t2[i] = (b3 ? t4[i + 2] : t4[i]) + __shfl_xor(b3 ? t4[i] : t4[i + 2], 8, 64);
    float t1 = (b2 ? t2[1] : t2[0]) + __shfl_xor(b2 ? t2[0] : t2[1], 4, 64);
    t1 += __shfl_xor(t1, 2, 64);
    t1 += __shfl_xor(t1, 1, 64);
    return t1;                       // row ((lane >> 2) & 15)
}
// the same for 8 rows: v[r] (r = 0..7) per lane -> the sum over all 64 lanes of v[r], delivered to lanes 8r .. 8r+7 (4 + 2 + 1 exchanges that
// halve the rows a lane carries, then three plain steps)
__device__ __forceinline__ float pv_reduce8_rows(const float (&v)[8], int lane) {
    float t4[4], t2[2];
    const bool b5 = lane & 32, b4 = lane & 16, b3 = lane & 8;
#pragma unroll
    for (int i = 0; i < 4; ++i) t4[i] = (b5 ? v[i + 4] : v[i]) + __shfl_xor(b5 ? v[i] : v[i + 4], 32, 64);
#pragma unroll
    for (int i = 0; i < 2; ++i) t2[i] = (b4 ? t4[i + 2] : t4[i]) + __shfl_xor(b4 ? t4[i] : t4[i + 2], 16, 64);
    float t1 = (b3 ? t2[1] : t2[0]) + __shfl_xor(b3 ? t2[0] : t2[1], 8, 64);
    t1 += __shfl_xor(t1, 4, 64);
    t1 += __shfl_xor(t1, 2, 64);
    t1 += __shfl_xor(t1, 1, 64);
    return t1;                       // row ((lane >> 3) & 7)
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

// ---- LayerNorm row helpers (one wave per row, the row stays in registers; two-pass mean / variance in fp32).  Shared by the
// standalone LN kernel and the GEMM-fused LN pass so both round identically. -------------------------------------------
template <int NCH>
struct RowRegs {
    float4 v[NCH];
};

template <int NCH>
__device__ __forceinline__ void pv_load_row(RowRegs<NCH>& r, const float* __restrict__ xr, int nvec, int lane) {
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        int idx = lane + 64 * j;
        r.v[j] = idx < nvec ? reinterpret_cast<const float4*>(xr)[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

// normalise in place: v <- (v - mean) * rstd * gamma + beta   (lanes beyond nvec keep zeros).  gamma / beta already in registers
// (the lane's NCH float4 of each): ONE arithmetic for the standalone kernel and the GEMM-fused passes.
// A scalar fp32 add the compiler cannot fold into a packed (v_pk_add_f32) tree.  Round 4: v_pk_*_f32 whose LOW result reads the HIGH register
// of a source pair (an op_sel bit set - what hipcc emits for a horizontal add of a packed pair, or to broadcast a value that sits in an
// odd register) returned wrong low results in lanes 48-63 about 1e-5 of the time on gfx950 while vector-memory loads were returning into
// VGPRs (scripts/dbg/gelu_glitch.py); the sums below run under exactly such loads in the GEMM-fused LayerNorm
// epilogues.  Same operation, same rounding as `a + b`.
__device__ __forceinline__ float pv_add_s(float a, float b) {
    float r;
    asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// JB rows at once (round 4): the SAME per-row arithmetic, written step by step ACROSS the rows, so that the JB independent chains of
// cross-lane reductions (two wave sums per row, each a dependent chain of four DPP adds, four v_readlane and three scalar adds) interleave
// instead of running one after the other - the GEMM-fused LayerNorm passes are latency-bound on exactly these chains.  JB = 1 is the
// standalone kernel's form; every row rounds identically for any JB.
template <int NCH, int JB>
__device__ __forceinline__ void pv_ln_rows_regs(RowRegs<NCH> (&r)[JB], const float4 (&gm)[NCH], const float4 (&bt)[NCH], int D, int nvec, int lane, float eps) {
    // every operation rounded on its own: which multiply-adds hipcc contracts into FMAs depends on the code this is inlined into, and the
    // standalone kernel and the GEMM-fused passes must agree to the bit (tests/test_hip_ops.py found a last-bit difference at N = 512)
#pragma clang fp contract(off)
    float s[JB], mean[JB], q[JB], rstd[JB];
#pragma unroll
    for (int b = 0; b < JB; ++b) {
        s[b] = 0.f;
#pragma unroll
        for (int j = 0; j < NCH; ++j) s[b] += pv_add_s(r[b].v[j].x + r[b].v[j].y, r[b].v[j].z + r[b].v[j].w);
    }
#pragma unroll
    for (int b = 0; b < JB; ++b) mean[b] = pv_wave_sum(s[b]) / (float)D;
#pragma unroll
    for (int b = 0; b < JB; ++b) {
        q[b] = 0.f;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            if (lane + 64 * j < nvec) {
                float a = r[b].v[j].x - mean[b], bb = r[b].v[j].y - mean[b], c = r[b].v[j].z - mean[b], d = r[b].v[j].w - mean[b];
                q[b] += pv_add_s(a * a + bb * bb, c * c + d * d);
            }
        }
    }
#pragma unroll
    for (int b = 0; b < JB; ++b) rstd[b] = 1.0f / sqrtf(pv_wave_sum(q[b]) / (float)D + eps);
#pragma unroll
    for (int b = 0; b < JB; ++b) {
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            int idx = lane + 64 * j;
            if (idx < nvec) {
                const float4 g = gm[j], be = bt[j];
                r[b].v[j].x = (r[b].v[j].x - mean[b]) * rstd[b] * g.x + be.x;
                r[b].v[j].y = (r[b].v[j].y - mean[b]) * rstd[b] * g.y + be.y;
                r[b].v[j].z = (r[b].v[j].z - mean[b]) * rstd[b] * g.z + be.z;
                r[b].v[j].w = (r[b].v[j].w - mean[b]) * rstd[b] * g.w + be.w;
            }
        }
    }
}

// Sixteen lanes per row (round 4, the full-row GEMM's epilogue): lane l16 of a 16-lane DPP row holds the 16-byte chunks l16 + 16 k (k < KC =
// D / 64) of ITS token row - every lane busy at any D, reductions by DPP alone, four token rows per wave at once.  The arithmetic AND its
// order are pv_ln_rows_regs': the wave-per-row form gives lane L = l16 + 16 i the chunks L and L + 64, sums a lane's chunks first
// ((0 + S_i) + S_{i+4}), then the 16 lanes of each DPP row, then (r0 + r1) + (r2 + r3); here a lane forms the same four partials itself and
// runs the same DPP tree on each of them, so every row rounds identically to the standalone kernel's (tests/test_hip_ops.py, bitwise).
// gamma / beta: the lane's chunks are read from an LDS copy (gb = gamma[D] | beta[D] floats).
template <int KC>
__device__ __forceinline__ void pv_ln_row16(float4 (&v)[KC], const __attribute__((address_space(3))) char* gb, int D, int l16, float eps) {
#pragma clang fp contract(off)
    static_assert(KC >= 4 && KC <= 8, "D = 256 .. 512");
    float pp[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float s_ = 0.f;
        s_ += pv_add_s(v[i].x + v[i].y, v[i].z + v[i].w);
        if (i + 4 < KC) s_ += pv_add_s(v[(i + 4) % KC].x + v[(i + 4) % KC].y, v[(i + 4) % KC].z + v[(i + 4) % KC].w);
        pp[i] = pv_row16_sum(s_);
    }
    const float mean = ((pp[0] + pp[1]) + (pp[2] + pp[3])) / (float)D;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float q_ = 0.f;
        {
            const float a = v[i].x - mean, bb = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
            q_ += pv_add_s(a * a + bb * bb, c * c + d * d);
        }
        if (i + 4 < KC) {
            const float a = v[(i + 4) % KC].x - mean, bb = v[(i + 4) % KC].y - mean, c = v[(i + 4) % KC].z - mean, d = v[(i + 4) % KC].w - mean;
            q_ += pv_add_s(a * a + bb * bb, c * c + d * d);
        }
        pp[i] = pv_row16_sum(q_);
    }
    const float rstd = 1.0f / sqrtf(((pp[0] + pp[1]) + (pp[2] + pp[3])) / (float)D + eps);
#pragma unroll
    for (int k = 0; k < KC; ++k) {
        const f32x4 g = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(gb + (l16 + 16 * k) * 16);
        const f32x4 be = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(gb + D * 4 + (l16 + 16 * k) * 16);
        v[k].x = (v[k].x - mean) * rstd * g[0] + be[0];
        v[k].y = (v[k].y - mean) * rstd * g[1] + be[1];
        v[k].z = (v[k].z - mean) * rstd * g[2] + be[2];
        v[k].w = (v[k].w - mean) * rstd * g[3] + be[3];
    }
}

template <int NCH>
__device__ __forceinline__ void pv_ln_row_regs(RowRegs<NCH>& r, const float4 (&gm)[NCH], const float4 (&bt)[NCH], int D, int nvec, int lane, float eps) {
    RowRegs<NCH> one[1] = {r};
    pv_ln_rows_regs<NCH, 1>(one, gm, bt, D, nvec, lane, eps);
    r = one[0];
}

template <int NCH>
__device__ __forceinline__ void pv_ln_load_affine(float4 (&gm)[NCH], float4 (&bt)[NCH], const float* __restrict__ gamma, const float* __restrict__ beta,
                                                  int nvec, int lane) {
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int idx = lane + 64 * j < nvec ? lane + 64 * j : nvec - 1;
        gm[j] = reinterpret_cast<const float4*>(gamma)[idx];
        bt[j] = reinterpret_cast<const float4*>(beta)[idx];
    }
}

template <int NCH>
__device__ __forceinline__ void pv_ln_row(RowRegs<NCH>& r, const float* __restrict__ gamma, const float* __restrict__ beta, int D,
                                          int nvec, int lane, float eps) {
    float4 gm[NCH], bt[NCH];
    pv_ln_load_affine<NCH>(gm, bt, gamma, beta, nvec, lane);
    pv_ln_row_regs<NCH>(r, gm, bt, D, nvec, lane, eps);
}

// hipGetLastError() is per-thread and sticky across ALL users of the runtime (PyTorch leaves benign errors such as
// failed pointer-attribute queries behind), so every launch first clears it: pv_check_launch() then reports OUR launch.
#define PV_LAUNCH(...) do { (void)hipGetLastError(); hipLaunchKernelGGL(__VA_ARGS__); } while (0)

// fast exact-erf GELU: Phi(x) from erfc(|x|/sqrt2) = t*(a1+t*(a2+t*(a3+t*(a4+t*a5))))*exp(-x^2/2), t = 1/(1+p|x|/sqrt2)
// (Abramowitz-Stegun 7.1.26, |erf error| <= 1.5e-7).  Measured against fp64: max abs error 6e-7, relative L2 9e-8 - the
// same class as torch's own fp32 F.gelu (7e-8) - at ~16 instructions (two transcendental) instead of ocml erff's ~27.
__device__ __forceinline__ float pv_gelu_fast(float x) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    const float u = x * 0.84932180028801904f;                       // sqrt(0.5*log2(e)): exp(-x^2/2) = exp2(-u*u)
    const float e = __builtin_amdgcn_exp2f(-(u * u));
    float poly = fmaf(t, 1.061405429f, -1.453152027f);
    poly = fmaf(t, poly, 1.421413741f);
    poly = fmaf(t, poly, -0.284496736f);
    poly = fmaf(t, poly, 0.254829592f);
    const float h = (poly * t) * 0.5f * e;                          // 0.5 * erfc(|x|/sqrt2)
    return x * (x < 0.f ? h : 1.0f - h);
}

// two GELUs at once on packed fp32 (v_pk_mul_f32 / v_pk_fma_f32 issue one instruction for two lanes-values): same
// arithmetic per element as pv_gelu_fast, bit-identical results (packed ops round each half like their scalar forms).
typedef __attribute__((ext_vector_type(2))) float f32x2;
__device__ __forceinline__ f32x2 pv_gelu_fast2(f32x2 x) {
    const f32x2 ax = {fabsf(x[0]), fabsf(x[1])};
    const f32x2 z = ax * 0.70710678118654752f;
    const f32x2 d = __builtin_elementwise_fma(z, (f32x2){0.3275911f, 0.3275911f}, (f32x2){1.0f, 1.0f});
    const f32x2 t = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    const f32x2 u = x * 0.84932180028801904f;
    const f32x2 nu2 = -(u * u);
    const f32x2 e = {__builtin_amdgcn_exp2f(nu2[0]), __builtin_amdgcn_exp2f(nu2[1])};
    f32x2 poly = __builtin_elementwise_fma(t, (f32x2){1.061405429f, 1.061405429f}, (f32x2){-1.453152027f, -1.453152027f});
    poly = __builtin_elementwise_fma(t, poly, (f32x2){1.421413741f, 1.421413741f});
    poly = __builtin_elementwise_fma(t, poly, (f32x2){-0.284496736f, -0.284496736f});
    poly = __builtin_elementwise_fma(t, poly, (f32x2){0.254829592f, 0.254829592f});
    const f32x2 h = (poly * t) * 0.5f * e;
    const f32x2 phi = {x[0] < 0.f ? h[0] : 1.0f - h[0], x[1] < 0.f ? h[1] : 1.0f - h[1]};
    return x * phi;
}

// once-per-DEVICE latch for per-device function attributes (hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the
// CURRENT device only): `static PvPerDevice latch; if (latch.first_use()) hipFuncSetAttribute(...)`.  The current device is the
// one the caller's stream belongs to (peekvit_amd.ops refuses to launch otherwise).  A benign race at worst repeats the call.
struct PvPerDevice {
    bool seen[64] = {};
    bool first_use() {
        int d = 0;
        if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) return true;
        if (seen[d]) return false;
        seen[d] = true;
        return true;
    }
};

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
