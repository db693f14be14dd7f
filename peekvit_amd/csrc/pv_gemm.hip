// bf16 MFMA GEMM with fused epilogues for the ViT encoder linears (gfx950).
//   out = epilogue(A[M,K] . W[N,K]^T)   A = activations (bf16), W = nn.Linear weight layout (out,in) (bf16)
// Both operands are K-contiguous, so they are staged identically: LDS-DMA (global_load_lds_dwordx4) into a
// lane-linear LDS image whose 16-byte chunks are XOR-swizzled on the SOURCE address (chunk ^ (row & 7)),
// read back with conflict-free ds_read_b128 through the same involution.
// The MFMA is issued "swapped" (A-operand = W rows, B-operand = activation rows) so that each lane's four
// accumulator registers are four CONSECUTIVE output columns of one output row: the epilogue then loads
// bias/residual and stores the result with 8-byte (bf16) / 16-byte (fp32) vector accesses.
#include "pv_common.h"
#include "pv_gelu_table.h"
#include <type_traits>
#include <cstdlib>

typedef __attribute__((ext_vector_type(8))) short s16x8;

// 1: the software-pipelined 16-bit epilogue of the 256^2 kernel (round 3); 0: round 2's one-pass form, kept for A/B builds (scripts/gemm_epi_ab.py)
// cache policy of the 256^2 epilogues' output stores: non-temporal (bit 0 = the 16-bit outputs, bit 1 = the fp32 residual stream).  The
// outputs are 0.6 - 2.5 GB per launch and nothing of them is re-read from L2, while the operands ARE (each A panel by up to 12 column tiles):
// with default stores they push the panels out.  Round 3, same box, libraries swapped between runs of bench.py (scripts/lib_ab.sh): forward
// 74.5 -> 73.2 ms; per shape (scripts/gemm_epi_ab.py) QKV + fold -2 %, fc1 + fold -1 %, fc2 -0.6 %.  The same idea measured and NOT adopted:
// non-temporal LOADS of the residual rows (-0.8 % = slower), non-temporal stores of the attention output (-1.5 %), nt on attention's K / V
// LDS-DMA (+-0).  0 restores default stores (A/B).
#ifndef PV_STORE_NT
#define PV_STORE_NT 3
#endif
#define PV_STORE16(ptr, val) do { if (PV_STORE_NT & 1) __builtin_nontemporal_store((val), (ptr)); else *(ptr) = (val); } while (0)
#define PV_STORE32(ptr, val) do { if (PV_STORE_NT & 2) __builtin_nontemporal_store((val), (ptr)); else *(ptr) = (val); } while (0)
// prefetching persistent launch: 0 = no prefetch (plain persistent loop), 1 = the whole prefetch issued at the start of the epilogue and
// retired before the first store, 2 (shipped) = prefetch issued in pieces under the first pass of the epilogue (16-bit) / behind the first
// residual rows (fp32) and retired by a counted wait at the end of the epilogue
#ifndef PV_PF_MODE
#define PV_PF_MODE 2
#endif
#ifndef PV_EPI_PIPE
#define PV_EPI_PIPE 1
#endif

// exact-erf GELU by table: gelu(x) itself as a piecewise cubic in x on 64 entries of width 11/64 over [-5.5, 5.5) (zero below,
// identity above; pv_gelu_table.h pv_gelu_cub, scripts/gen_gelu_table.py; |error| <= 7.3e-7, relative L2 5e-8 vs fp64 - the class of
// torch's fp32 F.gelu).  The fc1 epilogue is VALU-issue bound (4 cycles per instruction per SIMD, packed fp32 included), so the form is
// chosen for instruction count - 5.5 VALU + ONE 16-byte gather per value (round 2's Phi(x)-in-interval-coordinates form took 10):
//   bits  = x * S + MAGIC        (v_pk_fma_f32 for two values; the magic addend leaves the entry index in the low mantissa bits)
//   bits  = clamp(bits) as int   (v_med3_i32: huge / negative / NaN inputs land on the identity or the zero entry)
//   addr  = (bits << 8) + base   (v_lshl_add_u32; entry i is 256 bytes = 16 replicas above entry i-1)
//   (r0, r1) = (c1, c3) * x + (c0, c2)   (one v_pk_fma_f32 on the entry's register pairs: storage order {c0, c2, c1, c3})
//   y = x * (x * r1) + r0        (not x^2 * r1: x^2 overflows where the identity entry must still return x)
// Every 16-byte entry is stored 16 times and a lane reads replica (lane & 15), so the ds_read_b128 lane groups (16 lanes each) are
// bank-conflict free whatever entries the lanes need.  The SAME arithmetic runs from LDS (256^2 kernel) or from global memory (128^2
// kernel), so both kernels round an element identically.  `tab` = table base + this lane's replica (f32x4 elements).
// The scalar steps are inline asm, not C operators: hipcc's SLP pass otherwise pairs the steps of two values into packed instructions
// behind nine v_mov shuffles per four values.  The operations and their roundings are fmaf's / the product's.
__device__ __forceinline__ float pv_fma_s(float a, float b, float c) {
    float r;
    asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float pv_mul_s(float a, float b) {
    float r;
    asm("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float pv_gelu_poly(float x, const f32x4 c) {
    const pv_f32x2_t c02 = {c[0], c[1]}, c13 = {c[2], c[3]}, xx = {x, x};
    const pv_f32x2_t r = __builtin_elementwise_fma(c13, xx, c02);
    return pv_fma_s(x, pv_mul_s(x, r[1]), r[0]);
}
// 0x4B400000 + entry index of two values (adjacent accumulator registers)
__device__ __forceinline__ void pv_gelu_bits2(float x0, float x1, uint32_t& b0, uint32_t& b1) {
    const pv_f32x2_t xs = {x0, x1}, sc = {PV_GELU_CUB_SCALE, PV_GELU_CUB_SCALE}, mg = {PV_GELU_CUB_MAGIC, PV_GELU_CUB_MAGIC};
    const pv_f32x2_t t = __builtin_elementwise_fma(xs, sc, mg);
    // (the WHOLE vector is bit-cast, then indexed: hipcc 7.2 compiles __builtin_bit_cast(int, t[1]) - a bit-cast of a vector ELEMENT - as a
    // read of element 0, and both values took the first one's table entry; found in the ISA, reproduced in a ten-line kernel)
    typedef __attribute__((ext_vector_type(2))) int pv_i32x2_t;
    const pv_i32x2_t ti = __builtin_bit_cast(pv_i32x2_t, t);
    const int lo = 0x4B400000, hi = 0x4B400000 + PV_GELU_CUB_N - 1;
    b0 = (uint32_t)min(max(ti[0], lo), hi);
    b1 = (uint32_t)min(max(ti[1], lo), hi);
}
// The polynomial for an entry that has just arrived from GLOBAL memory (128^2 kernel): scalar FMAs for (r0, r1).
// Cause of round 3's "stale dword" (found in round 4): hipcc broadcasts an x that sits in an ODD register with
//   v_pk_fma_f32 ..., op_sel:[0,1,0]        (the LOW result reads the HIGH register of the src1 pair)
// and on gfx950 that form returns a low result computed as if the source were zero, in lanes 48-63, for ~1e-5 of the values, WHILE vector-memory
// loads (here: the next table gathers) are returning into VGPRs.  scripts/dbg/gelu_glitch.py over the PV_GELU_GLOBAL_MODE builds below
// (scripts/dbg/build_gelu_variants.py), 40 launches of a 2560 x 3072 x 768 GEMM each: mode 0 (hipcc's packed form) 37/40 launches wrong, every wrong
// element computed by an op_sel:[0,1,0] instruction, all in lanes 48-63; 2 (idle cycles after the wait) 29/40; 4 (consumers as plain C) 36/40;
// 9 (inline-asm v_pk_fma_f32 op_sel:[0,1,0], fresh registers, ALL values) 40/40; 1 / 5 (scalar FMAs), 3 / 6 / 7 / 8 (inline-asm packed FMA without
// that bit: fresh destination, op_sel_hi broadcast, in place over the addend / the multiplicand) 0/40.  The 256^2 kernel evaluates the same packed
// form from LDS with no register-destination load in flight and is bitwise relaunch-stable and elementwise exact (tests/test_hip_ops.py);
// tests/test_isa_audit.py keeps the form out of every kernel that computes under such loads.
// Same operations, same roundings as pv_gelu_poly: both kernels still round an element identically.
#ifndef PV_GELU_GLOBAL_MODE
#define PV_GELU_GLOBAL_MODE 1
#endif
__device__ __forceinline__ float pv_gelu_poly_g(float x, f32x4 c) {
#if PV_GELU_GLOBAL_MODE == 0
    return pv_gelu_poly(x, c);
#elif PV_GELU_GLOBAL_MODE == 2      // diagnostic (round 4): the packed form, 4+ idle cycles between the wait for the entry and its first read
    asm volatile("s_nop 3" : "+v"(c));
    return pv_gelu_poly(x, c);
#elif PV_GELU_GLOBAL_MODE == 3      // diagnostic: the packed FMA writes FRESH registers (early-clobber), never the load's destination registers
    pv_f32x2_t r;
    const pv_f32x2_t c02 = {c[0], c[1]}, c13 = {c[2], c[3]}, xx = {x, x};
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=&v"(r) : "v"(c13), "v"(xx), "v"(c02));
    return pv_fma_s(x, pv_mul_s(x, r[1]), r[0]);
#elif PV_GELU_GLOBAL_MODE == 4      // diagnostic: packed FMA as shipped, its CONSUMERS as compiler-visible C (no inline asm behind the packed result)
    const pv_f32x2_t c02 = {c[0], c[1]}, c13 = {c[2], c[3]}, xx = {x, x};
    const pv_f32x2_t r = __builtin_elementwise_fma(c13, xx, c02);
    return fmaf(x, x * r[1], r[0]);
#elif PV_GELU_GLOBAL_MODE == 6      // diagnostic: packed asm, FRESH destination, x broadcast by op_sel_hi (the modifier hipcc's form carries)
    pv_f32x2_t r;
    const pv_f32x2_t c02 = {c[0], c[1]}, c13 = {c[2], c[3]}, xx = {x, 0.0f};
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=&v"(r) : "v"(c13), "v"(xx), "v"(c02));
    return pv_fma_s(x, pv_mul_s(x, r[1]), r[0]);
#elif PV_GELU_GLOBAL_MODE == 9      // diagnostic: packed asm, FRESH destination, x taken from the HIGH register of its pair for BOTH results (op_sel:[0,1,0])
    pv_f32x2_t r;
    const pv_f32x2_t c02 = {c[0], c[1]}, c13 = {c[2], c[3]}, xx = {0.0f, x};
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]" : "=&v"(r) : "v"(c13), "v"(xx), "v"(c02));
    return pv_fma_s(x, pv_mul_s(x, r[1]), r[0]);
#elif PV_GELU_GLOBAL_MODE == 7      // diagnostic: packed asm, no op_sel, destination = the ADDEND's registers (in place over c0 | c2, as hipcc's form)
    pv_f32x2_t c02 = {c[0], c[1]};
    const pv_f32x2_t c13 = {c[2], c[3]}, xx = {x, x};
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(c02) : "v"(c13), "v"(xx));
    return pv_fma_s(x, pv_mul_s(x, c02[1]), c02[0]);
#elif PV_GELU_GLOBAL_MODE == 8      // diagnostic: packed asm, no op_sel, destination = the MULTIPLICAND's registers (over c1 | c3)
    pv_f32x2_t c13 = {c[2], c[3]};
    const pv_f32x2_t c02 = {c[0], c[1]}, xx = {x, x};
    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(c13) : "v"(xx), "v"(c02));
    return pv_fma_s(x, pv_mul_s(x, c13[1]), c13[0]);
#elif PV_GELU_GLOBAL_MODE == 5      // diagnostic: scalar FMAs as INLINE ASM (an asm statement is the first reader of the loaded registers)
    const float r0 = pv_fma_s(c[2], x, c[0]), r1 = pv_fma_s(c[3], x, c[1]);
    return pv_fma_s(x, pv_mul_s(x, r1), r0);
#else
    const float r0 = fmaf(c[2], x, c[0]), r1 = fmaf(c[3], x, c[1]);
    return pv_fma_s(x, pv_mul_s(x, r1), r0);
#endif
}
__device__ __forceinline__ void pv_gelu_lut2(float& x0, float& x1, const f32x4* tab) {          // table in global memory, two values in place
    uint32_t b0, b1;
    pv_gelu_bits2(x0, x1, b0, b1);
    const f32x4 c0 = tab[(b0 - 0x4B400000u) * PV_GELU_CUB_REP], c1 = tab[(b1 - 0x4B400000u) * PV_GELU_CUB_REP];
    x0 = pv_gelu_poly_g(x0, c0); x1 = pv_gelu_poly_g(x1, c1);
}
__device__ __forceinline__ void pv_gelu_lut2(float& x0, float& x1, const __attribute__((address_space(3))) f32x4* tab) {   // table in LDS
    uint32_t b0, b1;
    pv_gelu_bits2(x0, x1, b0, b1);
    // entry i lives i * 256 bytes above `tab`: (bits << 8) = 0x40000000 + i * 256 (mod 2^32) -> ONE v_lshl_add_u32 with a per-lane constant
    const uint32_t base = (uint32_t)(uintptr_t)tab - 0x40000000u;
    const f32x4 c0 = *(const __attribute__((address_space(3))) f32x4*)(uintptr_t)((b0 << 8) + base);
    const f32x4 c1 = *(const __attribute__((address_space(3))) f32x4*)(uintptr_t)((b1 << 8) + base);
    x0 = pv_gelu_poly(x0, c0); x1 = pv_gelu_poly(x1, c1);
}

// gelu'(x) for the training pair epilogue (round 6): the DERIVATIVE OF THE INTERVAL'S CUBIC, c1 + x (2 c2 + 3 c3 x), from the very table entry the
// forward GELU gathers - no second table, no second gather.  Against fp64 gelu'(x) = Phi(x) + x phi(x): |error| <= 1.0e-4 (the derivative of a fit is
// one order less accurate than the fit: 8e-7), relative L2 1.2e-5 - a quarter of the rounding of the 16-bit plane it is stored in.  Rounds 1-5 saved the
// pre-activation instead and evaluated gelu' in the DATA-GRADIENT GEMM's epilogue from a 4096-entry linear table: 23.4 k ticks per 256^2 tile against
// 6.4 k for a plain 16-bit epilogue (profiles/r06_gemm_stamps_train.txt: ~45 vector instructions and four gathers per row of eight rows of four
// passes); a conflict-free cubic table there measured 26.2 k.  Now that epilogue multiplies by the saved derivative.
// Scalar arithmetic (the 128^2 kernel evaluates it under in-flight register loads; every kernel rounds an element identically).
__device__ __forceinline__ float pv_gelu_dpoly(float x, const f32x4 c) {        // c = {c0, c2, c1, c3}
    const float t = pv_fma_s(pv_mul_s(c[3], 3.0f), x, pv_add_s(c[1], c[1]));      // (3 c3, not 3 x: the outermost entries have c3 = 0, and 0 * (3 x) would be NaN once 3 x overflows)
    return pv_fma_s(x, t, c[2]);
}
__device__ __forceinline__ void pv_gelu_dlut2(float& x0, float& x1, const f32x4* tab) {          // table in global memory, two values in place
    uint32_t b0, b1;
    pv_gelu_bits2(x0, x1, b0, b1);
    const f32x4 c0 = tab[(b0 - 0x4B400000u) * PV_GELU_CUB_REP], c1 = tab[(b1 - 0x4B400000u) * PV_GELU_CUB_REP];
    x0 = pv_gelu_dpoly(x0, c0); x1 = pv_gelu_dpoly(x1, c1);
}
__device__ __forceinline__ void pv_gelu_dlut2(float& x0, float& x1, const __attribute__((address_space(3))) f32x4* tab) {   // table in LDS
    uint32_t b0, b1;
    pv_gelu_bits2(x0, x1, b0, b1);
    const uint32_t base = (uint32_t)(uintptr_t)tab - 0x40000000u;
    const f32x4 c0 = *(const __attribute__((address_space(3))) f32x4*)(uintptr_t)((b0 << 8) + base);
    const f32x4 c1 = *(const __attribute__((address_space(3))) f32x4*)(uintptr_t)((b1 << 8) + base);
    x0 = pv_gelu_dpoly(x0, c0); x1 = pv_gelu_dpoly(x1, c1);
}

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
    int fr_full, fr_half;      // full-row kernel, split remainder: fr_full 128-row tiles (a multiple of 8) + fr_half 64-row tiles; 0, 0 = tiles_m plain tiles
    int fr_big;                // ... or (round 6) fr_full 128-row tiles + fr_big 160-row tiles over the rows behind them
    int gm;                    // M-blocks per tile-order group (256^2 kernel): inside a group n is the SLOW index
    int gc;                    // column tiles per raster chunk (256^2 kernel): the tile list is chunk-major, so an XCD keeps the
                               // same gc weight tiles while it streams the activation row panels of its share of the rows
    uint32_t* range_flag;      // operand-range guard of the fp16 build (pv_common.h), or null
    float* rowsq_out;          // PV_EPI_BIAS_RES_F32: [tiles_n][M] sum of squares of each output row's segment (RankViT norms), or null
    int res_scaled;            // PV_EPI_BIAS_RES_F32 with row_scale: the residual row is scaled too (ResidualViT: res = the unmasked tokens)
    int k_last;                // TN kernel: K extent of the last split-K slice (the slices need not be equal)
    float* colsum_partial;     // PV_EPI_GELU_GRAD_BF16: [tiles_m][N] column sums of the stored tile rows (bias gradient), or null
    // LayerNorm folding (the default for large batches): the PRODUCER (PV_EPI_BIAS_RES_F32) also emits the 16-bit copy of its
    // output rows and per-(column tile, row) partial (sum, sum of squares); the CONSUMER (PV_EPI_BIAS_BF16 / _GELU_BF16) runs on that
    // copy with W' = gamma (.) W and finishes  out = rstd[m] * (acc - mean[m] * c1[n]) + c2[n]  before its usual epilogue.
    uint16_t* x16_out;
    float* rowstat_out;        // [tiles_n][M][2]
    const float* fold_stat;    // [M][2] (mean, rstd)
    const float* fold_c1;      // [N] = sum_k W'[n][k]
    const float* fold_c2;      // [N] = sum_k beta[k] W[n][k] + bias[n]
    int ksplit;                // split-K (wgrad): blocks [t*ntiles, (t+1)*ntiles) compute K slice t into out + t*split_stride
    int k_slice;
    int64_t split_stride;      // elements of `out` between consecutive slices
    const float* ln_gamma;     // fused LayerNorm of the finished output rows (row-block kernel): out rows -> ln_out bf16
    const float* ln_beta;
    const float* ln_row_scale;
    uint16_t* ln_out;
    float ln_eps;
#ifdef PV_STAMPS
    unsigned long long* dbg;   // diagnostic build only: per-block s_memtime stamps (never read by any kernel)
#endif
};

#ifdef PV_STAMPS
static unsigned long long* g_pv_dbg = nullptr;
extern "C" void pv_debug_set_stamp_buffer(void* p) { g_pv_dbg = (unsigned long long*)p; }
#define PV_STAMP(i)                                                                                  \
    do {                                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                           \
        unsigned long long t_;                                                                       \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                    \
        __builtin_amdgcn_sched_barrier(0);                                                           \
        if (threadIdx.x == 0 && p.dbg) p.dbg[pv_stamp_slot * 16 + (i)] = t_;                         \
    } while (0)
#else
#define PV_STAMP(i)
#endif

// AUX = cache-policy bits of the load (0 default, 1 sc0, 2 nt, 16 sc1): a literal, hence a template argument
template <int AUX = 0>
__device__ __forceinline__ void pv_glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, AUX);
}
// cache policy of the 256^2 kernel's activation (A) and weight (W) staging loads; build-time knobs for the L2 experiments
// (scripts/bench_gemm.py loads differently built libraries side by side in one process)
#ifndef PV_GEMM_A_AUX
#define PV_GEMM_A_AUX 0
#endif
#ifndef PV_GEMM_W_AUX
#define PV_GEMM_W_AUX 0
#endif

// XCD-aware bijective remap (8 XCDs, blocks b and b+8 share an XCD): every XCD walks a CONTIGUOUS range of the
// tile list (n fastest), so co-resident blocks of one XCD share A panels and the weight matrix in that XCD's L2.
__device__ __forceinline__ int pv_xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, i = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
}

// Per-lane epilogue of the 128^2 kernel.  `acc` already contains the bias (accumulators are INITIALISED with it, in both
// kernels, so that an output element is rounded identically whichever kernel/tile computes it: batch invariance).
// second operand of the 128^2 kernel's epilogue for output element group (m, n .. n+3): the residual / positional row segment or the
// saved pre-activation (as fp32).  Fetched for ALL 16 groups of a lane before the first store (a store may alias the residual, so
// the compiler would otherwise serialise load -> wait -> store sixteen times); out-of-range groups read a clamped address.
template <int EPI>
__device__ __forceinline__ f32x4 pv_epilogue_fetch(const GemmDev& p, int m, int n) {
    m = m < p.M ? m : p.M - 1;
    n = n + 4 <= p.N ? n : p.N - 4;
    if (EPI == PV_EPI_BIAS_RES_F32) {
        return *reinterpret_cast<const f32x4*>(p.res + (int64_t)m * p.ldr + n);
    } else if (EPI == PV_EPI_GELU_GRAD_BF16) {
        const u32x2 w = *reinterpret_cast<const u32x2*>(reinterpret_cast<const uint16_t*>(p.res) + (int64_t)m * p.ldr + n);
        return (f32x4){pv_unpack_lo(w[0]), pv_unpack_hi(w[0]), pv_unpack_lo(w[1]), pv_unpack_hi(w[1])};
    } else if (EPI == PV_EPI_BIAS_POS_F32) {
        const int img = m / p.rpi, pi = m - img * p.rpi;
        return *reinterpret_cast<const f32x4*>(p.pos + (int64_t)(p.row_off + pi) * p.N + n);
    }
    return (f32x4){0.f, 0.f, 0.f, 0.f};
}

template <int EPI, bool GUARD = true>
__device__ __forceinline__ void pv_epilogue_store(const GemmDev& p, int m, int n, f32x4 acc, f32x4 r, float row_scale, float& vmax) {
    if (GUARD && (m >= p.M || n >= p.N)) return;
    const float v0 = acc[0], v1 = acc[1], v2 = acc[2], v3 = acc[3];
    if (EPI == PV_EPI_BIAS_BF16) {
        const float s = n < p.qcols ? p.qscale : 1.0f;
        u32x2 o = {pv_pack_bf16x2_tracked(v0 * s, v1 * s, vmax), pv_pack_bf16x2_tracked(v2 * s, v3 * s, vmax)};
        *reinterpret_cast<u32x2*>(reinterpret_cast<uint16_t*>(p.out) + (int64_t)m * p.ldo + n) = o;
    } else if (EPI == PV_EPI_BIAS_GELU_BF16) {
        const f32x4* tab = reinterpret_cast<const f32x4*>(pv_gelu_cub) + (threadIdx.x & 15);
        float y0 = v0, y1 = v1, y2 = v2, y3 = v3;
        pv_gelu_lut2(y0, y1, tab); pv_gelu_lut2(y2, y3, tab);
        u32x2 o = {pv_pack_bf16x2_tracked(y0, y1, vmax), pv_pack_bf16x2_tracked(y2, y3, vmax)};
        *reinterpret_cast<u32x2*>(reinterpret_cast<uint16_t*>(p.out) + (int64_t)m * p.ldo + n) = o;
    } else if (EPI == PV_EPI_BIAS_RES_F32) {
        const float s = row_scale, t = p.res_scaled ? row_scale : 1.0f;
        float4 o = make_float4(fmaf(s, v0, t * r[0]), fmaf(s, v1, t * r[1]), fmaf(s, v2, t * r[2]), fmaf(s, v3, t * r[3]));
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + (int64_t)m * p.ldo + n) = o;
    } else if (EPI == PV_EPI_BIAS_F32) {
        const float s = n < p.qcols ? p.qscale : 1.0f;
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + (int64_t)m * p.ldo + n) = make_float4(v0 * s, v1 * s, v2 * s, v3 * s);
    } else if (EPI == PV_EPI_BIAS_GELU_SPLIT_BF16) {
        const f32x4* tab = reinterpret_cast<const f32x4*>(pv_gelu_cub) + (threadIdx.x & 15);
        float y0 = v0, y1 = v1, y2 = v2, y3 = v3;
        pv_gelu_lut2(y0, y1, tab); pv_gelu_lut2(y2, y3, tab);
        const PvHiLo a = pv_split2(y0, y1), b = pv_split2(y2, y3);
        uint16_t* o = reinterpret_cast<uint16_t*>(p.out) + (int64_t)m * p.ldo + n;
        *reinterpret_cast<u32x2*>(o) = (u32x2){a.hi, b.hi};
        *reinterpret_cast<u32x2*>(o + p.N) = (u32x2){a.lo, b.lo};
        *reinterpret_cast<u32x2*>(o + 2 * p.N) = (u32x2){a.hi, b.hi};
    } else if (EPI == PV_EPI_BIAS_GELU_PAIR_BF16) {
        const f32x4* tab = reinterpret_cast<const f32x4*>(pv_gelu_cub) + (threadIdx.x & 15);
        uint16_t* o = reinterpret_cast<uint16_t*>(p.out) + (int64_t)m * p.ldo + n;
        float y0 = v0, y1 = v1, y2 = v2, y3 = v3, d0 = v0, d1 = v1, d2 = v2, d3 = v3;
        pv_gelu_lut2(y0, y1, tab); pv_gelu_lut2(y2, y3, tab);
        pv_gelu_dlut2(d0, d1, tab); pv_gelu_dlut2(d2, d3, tab);
        *reinterpret_cast<u32x2*>(o) = (u32x2){pv_pack_bf16x2_tracked(y0, y1, vmax), pv_pack_bf16x2_tracked(y2, y3, vmax)};
        *reinterpret_cast<u32x2*>(o + p.N) = (u32x2){pv_pack_bf16x2(d0, d1), pv_pack_bf16x2(d2, d3)};
    } else if (EPI == PV_EPI_GELU_GRAD_BF16) {
        u32x2 o = {pv_pack_bf16x2(pv_mul_s(v0, r[0]), pv_mul_s(v1, r[1])), pv_pack_bf16x2(pv_mul_s(v2, r[2]), pv_mul_s(v3, r[3]))};      // r = the saved gelu'(pre)
        *reinterpret_cast<u32x2*>(reinterpret_cast<uint16_t*>(p.out) + (int64_t)m * p.ldo + n) = o;
    } else {   // PV_EPI_BIAS_POS_F32
        const int img = m / p.rpi, pi = m - img * p.rpi;
        const int64_t orow = (int64_t)img * p.rpo + p.row_off + pi;
        float4 o = make_float4(r[0] + v0, r[1] + v1, r[2] + v2, r[3] + v3);
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.out) + orow * p.ldo + n) = o;
    }
}

// ------------------------------------------------------------------------------------------------
// 128 x 128 x 64 tile, 4 waves (2 x 2), 64 x 64 per wave = 4 x 4 MFMA 16x16x32 tiles, 2 LDS buffers (64 KiB), 2 workgroups / CU
// (measured dead end, r2: the same tile with K steps of 32 through a 4-stage LDS-DMA ring, counted vmcnt and ONE raw barrier per
// step - bit-identical - is 0-10 % SLOWER on every small / short-K shape tried (scripts/gemm128_ab.py, profiles/r02_gemm128_ab.json):
// a barrier per 16 MFMAs costs more than the deeper prefetch returns when two workgroups per CU already overlap each other)
// ------------------------------------------------------------------------------------------------
constexpr int G1_BM = 128, G1_BN = 128, G1_BK = 64;
constexpr int G1_TILE_BYTES = G1_BM * G1_BK * 2;   // 16 KiB per operand per stage

template <int EPI>
__global__ __launch_bounds__(256, 2) void pv_gemm128_kernel(const GemmDev p_in) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wn = wid & 1;

    GemmDev p = p_in;
    const int ntiles = p.tiles_m * p.tiles_n;
    const int slice = blockIdx.x / ntiles;                       // split-K slice (0 when ksplit <= 1)
    if (p.ksplit > 1) {
        p.A += (int64_t)slice * p.k_slice; p.W += (int64_t)slice * p.k_slice; p.K = p.k_slice;
        p.out = reinterpret_cast<float*>(p.out) + slice * p.split_stride;
        if (slice) p.bias = nullptr;
    }
    const int tile = pv_xcd_remap(blockIdx.x - slice * ntiles, ntiles);
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

    const int en = n0 + wn * 64 + ((lane >> 4) << 2);
    f32x4 acc[4][4];   // [nt][mt], initialised with the bias of the lane's 4 columns
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
        if (p.bias && en + i * 16 < p.N) b4 = *reinterpret_cast<const f32x4*>(p.bias + en + i * 16);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = b4;
    }

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
                    acc[nt][mt] = PV_MFMA_16x16x32(wf[nt], xf[mt], acc[nt][mt], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // ---- epilogue: lane holds out[m = ..+(lane&15)][n = ..+(lane>>4)*4 + 0..3] ---------------------------------
    const int em = m0 + wm * 64 + (lane & 15);
    float vmax = 0.f;          // operand-range guard (fp16 build)
    f32x4 rr[4][4];
    float rs[4] = {1.0f, 1.0f, 1.0f, 1.0f};
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        if (EPI == PV_EPI_BIAS_RES_F32 && p.row_scale) rs[mt] = p.row_scale[em + mt * 16 < p.M ? em + mt * 16 : p.M - 1];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) rr[mt][nt] = pv_epilogue_fetch<EPI>(p, em + mt * 16, en + nt * 16);
    }
    if (EPI == PV_EPI_BIAS_RES_F32 || EPI == PV_EPI_GELU_GRAD_BF16 || EPI == PV_EPI_BIAS_POS_F32) {
        // every fetch has returned before the first store is issued (keeps the 16 loads in flight together)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) asm volatile("" : "+v"(rr[mt][nt]));
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) asm volatile("" : "+v"(rs[mt]));
        __builtin_amdgcn_sched_barrier(0);
    }
    if (m0 + G1_BM <= p.M && n0 + G1_BN <= p.N) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) pv_epilogue_store<EPI, false>(p, em + mt * 16, en + nt * 16, acc[nt][mt], rr[mt][nt], rs[mt], vmax);
    } else {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) pv_epilogue_store<EPI, true>(p, em + mt * 16, en + nt * 16, acc[nt][mt], rr[mt][nt], rs[mt], vmax);
    }
    if (EPI == PV_EPI_BIAS_BF16 || EPI == PV_EPI_BIAS_GELU_BF16) pv_range_commit(vmax, p.range_flag);
}

// ------------------------------------------------------------------------------------------------
// Patch embedding WITHOUT a materialised patch matrix (round 6; models/vit.py:203-222: conv_proj with kernel = stride = P, reshape, permute,
// + positional embedding).  The 128 x 128 tile kernel above with its A operand gathered from the fp32 NCHW image on the fly: a thread's 16-byte
// chunk of the LDS image (8 consecutive K values of one patch row) is 8 consecutive pixels of one image row - two 16-byte loads, converted and
// written where the LDS-DMA would have put them, so the fragment reads and the MFMA order are the tile kernel's: bit-identical to
// pv_im2col_bf16 + pv_gemm_bf16(PV_EPI_BIAS_POS_F32).  The pixels of K-tile kt + 1 fly under the MFMAs of K-tile kt.  What it saves is the
// patch matrix's round trip (vit_small, batch 512: 154 MB written by pv_im2col and read back: 90 + 120 us of a 4.8 ms forward); a lane's eight
// rows-of-patches are neighbours in the image (8 patches x 64 B of one image row per wave instruction).  For narrow models only: each of the N / 128
// column tiles gathers the pixels again (from L2), and at D = 768 the 256^2 kernel on a materialised matrix is the faster form.
// ------------------------------------------------------------------------------------------------
struct PatchGeom { const float* img; int C, R, P, Wp, Np; };      // image fp32 [B, C, R, R]; Wp = R / P patches per row, Np = Wp * Wp per image

__global__ __launch_bounds__(256, 2) void pv_patch_embed_kernel(const GemmDev p, const PatchGeom gm) {
    constexpr int EPI = PV_EPI_BIAS_POS_F32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wn = wid & 1;
    const int ntiles = p.tiles_m * p.tiles_n;
    const int tile = pv_xcd_remap(blockIdx.x, ntiles);
    const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
    const int m0 = tm * G1_BM, n0 = tn * G1_BN;

    const int srow = wid * 8 + (lane >> 3);                      // + 32 * i
    const int schunk = (lane & 7) ^ ((lane >> 3) & 7);           // logical 16-B chunk (8 K values) this lane provides
    const float* gpx[4];                                         // first pixel of the row's patch (channel 0, patch row 0)
    const uint16_t* gw[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int ra = m0 + i * 32 + srow; ra = ra < p.M ? ra : p.M - 1;
        const int b = ra / gm.Np, pi = ra - b * gm.Np, py = pi / gm.Wp, px = pi - py * gm.Wp;
        gpx[i] = gm.img + ((int64_t)b * gm.C * gm.R + (int64_t)py * gm.P) * gm.R + px * gm.P;
        int rw = n0 + i * 32 + srow; rw = rw < p.N ? rw : p.N - 1;
        gw[i] = p.W + (int64_t)rw * p.ldw + schunk * 8;
    }
    float4 pxv[4][2];
    auto fetch = [&](int kt) {                                   // the 8 pixels of this lane's chunk of K-tile kt, for its four rows
        const int k0 = kt * G1_BK + schunk * 8, pp = gm.P * gm.P;
        const int c = k0 / pp, rem = k0 - c * pp, ky = rem / gm.P, kx = rem - ky * gm.P;
        const int64_t off = ((int64_t)c * gm.R + ky) * gm.R + kx;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            pxv[i][0] = *reinterpret_cast<const float4*>(gpx[i] + off);
            pxv[i][1] = *reinterpret_cast<const float4*>(gpx[i] + off + 4);
        }
    };
    float vmax = 0.f;                                            // operand-range guard of the packed pixels (fp16 build), as pv_im2col_bf16
    auto put = [&](int buf) {
        char* la = smem + buf * (2 * G1_TILE_BYTES) + wid * 1024 + lane * 16;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const u32x4 v = {pv_pack_bf16x2_tracked(pxv[i][0].x, pxv[i][0].y, vmax), pv_pack_bf16x2_tracked(pxv[i][0].z, pxv[i][0].w, vmax),
                             pv_pack_bf16x2_tracked(pxv[i][1].x, pxv[i][1].y, vmax), pv_pack_bf16x2_tracked(pxv[i][1].z, pxv[i][1].w, vmax)};
            *reinterpret_cast<u32x4*>(la + i * 4096) = v;
        }
    };
    auto stage_w = [&](int buf, int kt) {
        char* lw = smem + buf * (2 * G1_TILE_BYTES) + G1_TILE_BYTES + wid * 1024;
#pragma unroll
        for (int i = 0; i < 4; ++i) pv_glds16(gw[i] + kt * G1_BK, lw + i * 4096);
    };

    const int frow = lane & 15;
    const int fx0 = (((lane >> 4) ^ (lane & 7)) << 4);
    const int a_off = (wm * 64 + frow) * 128;
    const int w_off = (wn * 64 + frow) * 128;
    const int en = n0 + wn * 64 + ((lane >> 4) << 2);
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
        if (p.bias && en + i * 16 < p.N) b4 = *reinterpret_cast<const f32x4*>(p.bias + en + i * 16);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = b4;
    }

    const int nk = p.K / G1_BK;
    stage_w(0, 0);
    fetch(0);
    put(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) { stage_w(cur ^ 1, kt + 1); fetch(kt + 1); }
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
                    acc[nt][mt] = PV_MFMA_16x16x32(wf[nt], xf[mt], acc[nt][mt], 0, 0, 0);
        }
        if (kt + 1 < nk) put(cur ^ 1);                           // (buffer cur ^ 1 was last read before the previous barrier)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    const int em = m0 + wm * 64 + (lane & 15);
    float dummy = 0.f;
    f32x4 rr[4][4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) rr[mt][nt] = pv_epilogue_fetch<EPI>(p, em + mt * 16, en + nt * 16);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) asm volatile("" : "+v"(rr[mt][nt]));
    __builtin_amdgcn_sched_barrier(0);
    if (m0 + G1_BM <= p.M && n0 + G1_BN <= p.N) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) pv_epilogue_store<EPI, false>(p, em + mt * 16, en + nt * 16, acc[nt][mt], rr[mt][nt], 1.0f, dummy);
    } else {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) pv_epilogue_store<EPI, true>(p, em + mt * 16, en + nt * 16, acc[nt][mt], rr[mt][nt], 1.0f, dummy);
    }
    pv_range_commit(vmax, p.range_flag);
}

// models/vit.py:203-222 in one launch: tokens[b, row_off + i, :] = patch_i(img[b]) . W^T + bias + pos[row_off + i, :].  img fp32 [B, C, R, R] (NCHW,
// 16-byte aligned, R % P == 0, P % 8 == 0, C P P % 64 == 0), W 16-bit [D, C P P] (the conv weight's own layout), tokens fp32 [B, S, D].
extern "C" int pv_patch_embed_f32(const float* img, const uint16_t* W, const float* bias, const float* pos, float* tokens, int64_t B, int64_t C, int64_t R,
                                  int64_t P, int64_t D, int64_t S, int64_t row_off, uint32_t* range_flag, void* stream) {
    if (!img || !W || !pos || !tokens || B <= 0 || C <= 0 || R <= 0 || P <= 0 || D <= 0 || S <= 0 || row_off < 0) return PV_ERR_INVALID_ARG;
    if (R % P || P % 8 || (C * P * P) % G1_BK || D % 4 || R % 4 || ((uintptr_t)img & 15) || ((uintptr_t)W & 15) || ((uintptr_t)tokens & 15) || ((uintptr_t)pos & 15) ||
        (bias && ((uintptr_t)bias & 15)))
        return PV_ERR_UNSUPPORTED;
    const int64_t Wp = R / P, Np = Wp * Wp, M = B * Np;
    if (row_off + Np > S || M > 0x7fffffff || B * S > 0x7fffffff) return PV_ERR_INVALID_ARG;
    GemmDev p = {};
    p.W = W; p.bias = bias; p.out = tokens; p.pos = pos;
    p.M = (int)M; p.N = (int)D; p.K = (int)(C * P * P);
    p.ldw = p.K; p.ldo = D;
    p.rpi = (int)Np; p.rpo = (int)S; p.row_off = (int)row_off;
    p.qscale = 1.0f;
    p.tiles_m = (int)((M + G1_BM - 1) / G1_BM); p.tiles_n = (int)((D + G1_BN - 1) / G1_BN);
    p.range_flag = range_flag;
    const PatchGeom gm = {img, (int)C, (int)R, (int)P, (int)Wp, (int)Np};
    static PvPerDevice attr_set;
    const int lds = 2 * 2 * G1_TILE_BYTES;
    if (attr_set.first_use()) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pv_patch_embed_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    PV_LAUNCH(pv_patch_embed_kernel, dim3((unsigned)(p.tiles_m * p.tiles_n)), dim3(256), lds, (hipStream_t)stream, p, gm);
    return pv_check_launch();
}

template <int EPI>
static int pv_launch_gemm128(const GemmDev& p, hipStream_t stream) {
    static PvPerDevice attr_set;
    const int lds = 2 * 2 * G1_TILE_BYTES;
    if (attr_set.first_use()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pv_gemm128_kernel<EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    }
    PV_LAUNCH(pv_gemm128_kernel<EPI>, dim3((unsigned)(p.tiles_m * p.tiles_n * (p.ksplit > 1 ? p.ksplit : 1))), dim3(256), lds, stream, p);
    return pv_check_launch();
}

// ------------------------------------------------------------------------------------------------
// 256 x 256 x 64 tile, 8 waves (2 M x 4 N, 128 x 64 outputs per wave = 128 accumulator VGPRs), 128 KiB LDS,
// one workgroup per CU.  Deep pipeline in plain HIP: LDS-DMA stays in flight ACROSS raw s_barriers behind a
// counted s_waitcnt vmcnt (never 0 in the loop); the two wave groups (waves 0-3 / 4-7 = the two waves of every
// SIMD) run staggered by one barrier, so in every barrier interval one wave per SIMD issues its 16 MFMAs while its
// partner issues the next phase's ds_reads and LDS-DMA.
//
// LDS: 2 buffers (K-tile parity) x 4 half-tile slots {A0, A1, B0, B1} of 128 rows x 64 k (16 KiB, swizzled as
// above).  Per K-tile t (buffer t&1) a wave runs 4 phases, each = [load interval | barrier | MFMA interval | barrier]:
//   P1: read A(m0) 8x b128 + B(n0) 4x   MFMA (m0,n0)     stage B0(t+1)
//   P2: read A(m1) 8x                   MFMA (m1,n0)     stage B1(t+1)
//   P3: read B(n1) 4x                   MFMA (m1,n1)     stage A0(t+2)      (A slots: last read in P2)
//   P4: -                               MFMA (m0,n1)     stage A1(t+2), s_waitcnt vmcnt(4)
// Hazards (q = phase index, group g in {0,1} loads in interval 2q+g, every wave drains lgkmcnt BEFORE the barrier
// that ends its load interval):
//   WAR  a slot last read in phase q is free from interval 2q+2 on -> restaging in phase q+1 is safe for both groups;
//   RAW  vmcnt(4) in P4 retires everything but the two youngest half-tiles (A0/A1 of t+2): all of tile t+1 has
//        landed for THIS thread; after one more barrier pair it has for every thread -> first read in P1 of t+1.
// (Measured, round 3: the same K-tile in TWO phases of 32 MFMAs - Q1 = all reads but B(n1) + both B halves staged, Q2 = B(n1) + both A
//  halves staged; half the barriers, same slots and hazards, bit-identical - is within +-1.3 % on every token GEMM
//  (profiles/r03_gemm_two_phase_ab.json: fc1 +1.0 %, fc2 +0.7 %, QKV -0.5 %, out-proj -0.7 %): the ~250 cycles per K-tile the barriers
//  cost come back as clock, not as time, like every other saving inside the K loop since round 1.  Not kept.)
// ------------------------------------------------------------------------------------------------
constexpr int G2_BM = 256, G2_BN = 256, G2_BK = 64;
constexpr int G2_HALF = 128 * G2_BK * 2;     // 16 KiB half-tile slot
constexpr int G2_BUF = 4 * G2_HALF;          // 64 KiB per K-tile buffer
constexpr int G2_LDS = 2 * G2_BUF;           // 128 KiB

// one 256 x 256 output tile at (m0, n0): prologue, pipelined K loop, transposed epilogue.  Uses smem[0, G2_LDS) (+ the GELU table).
// PF = the prefetching persistent launch (pv_gemm256_pf_kernel, round 3): `first` = this workgroup's first tile (its K-tile 0 is staged
// here); otherwise K-tile 0 is ALREADY in buffer 0, staged by the previous tile's epilogue, which - when `has_next` - stages the first
// K-tile of tile (nm0, nn0) in turn.  PF = false: the tile is self-contained (one tile per workgroup, the rows kernel).
// FULL (round 4): the tile lies entirely inside the matrix (every tile but the last row / column tile) - the row / column clamps of the
// staging addresses and the bounds guards of the epilogue's stores compile away.  The kernels pick the instantiation per tile.
template <int EPI, bool PF = false, bool FULL = false>
__device__ __forceinline__ void pv_gemm256_tile(const GemmDev& p, char* smem, const int m0, const int n0, const bool first = true,
                                                const bool has_next = false, const int nm0 = 0, const int nn0 = 0, const int wid_pf = 0, const int slot_pf = 0) {
    int tid_ = threadIdx.x;
    // PF: the thread index is REBUILT per tile from the wave index (an SGPR of the persistent loop) and a volatile v_mbcnt pair, so that no
    // VGPR lives across tiles and nothing derived from the thread index (staging offsets, fragment bases) is hoisted out of the loop and
    // kept live through the epilogue, whose register budget is full.  (hipcc spilled 53 VGPRs without this, and a scratch reload is a
    // vector-memory operation: the s_waitcnt vmcnt(0) in front of its use also waits for every LDS-DMA and store in flight.)
    if (PF) {
        int l_;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l_));
        tid_ = wid_pf * 64 + l_;
    }
    int tid = tid_, lane = tid & 63;       // (PF: rebuilt once more at the start of the epilogue)
    const int wid = PF ? wid_pf : __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 2, wc = wid & 3;
#ifdef PV_STAMPS
    const size_t pv_stamp_slot = PF ? (size_t)slot_pf : (size_t)blockIdx.x;      // stamps per TILE: the persistent launch passes the tile's list position
#endif
    // PF: the tile's 256 bias values travel through LDS (1 KiB behind the table and the fold constants), staged with K-tile 0: a register load at
    // the start of a tile would put an s_waitcnt vmcnt(0) - the previous tile's stores included - in front of the first MFMA
    constexpr int PFM = PV_PF_MODE;
    constexpr int PF_BIAS_BASE = G2_LDS + ((EPI == PV_EPI_BIAS_GELU_BF16 || EPI == PV_EPI_BIAS_GELU_SPLIT_BF16 || EPI == PV_EPI_BIAS_GELU_PAIR_BF16) ? PV_GELU_CUB_N * PV_GELU_CUB_REP * 16 : 0) + 4096;
    PV_STAMP(14);
#ifdef PV_STAMPS
    if (threadIdx.x == 0 && p.dbg) { unsigned long long rt_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt_)::"memory"); p.dbg[pv_stamp_slot * 16 + 13] = rt_; }
#endif

    // ---- LDS-DMA sources: per half-tile two 1-KiB pieces per wave (rows j*64 + wid*8 + lane/8), swizzled chunk ----
    const int srow = wid * 8 + (lane >> 3);
    const int schunk = (lane & 7) ^ ((lane >> 3) & 7);
    // block-uniform bases (SGPR pairs) + per-lane 32-bit byte offsets: keeps the 8 DMA sources in 8 VGPRs
    const char* const a_blk = reinterpret_cast<const char*>(p.A + (int64_t)m0 * p.lda);
    const char* const w_blk = reinterpret_cast<const char*>(p.W + (int64_t)n0 * p.ldw);
    uint32_t oa[2][2], ow[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            int ra = h * 128 + j * 64 + srow; ra = (FULL || m0 + ra < p.M) ? ra : p.M - 1 - m0;
            // LDS row q of every aligned 32-row block holds W row perm(q) = ((q&15)>>2)*8 + (q>>4)*4 + (q&3): with the
            // MFMA C layout (row = 4*(lane>>4)+reg) a lane then owns 8 CONSECUTIVE output columns per tile pair.
            const int q = j * 64 + srow;
            int rw = h * 128 + (q & ~31) + (((q & 15) >> 2) << 3) + (((q >> 4) & 1) << 2) + (q & 3);
            rw = (FULL || n0 + rw < p.N) ? rw : p.N - 1 - n0;
            oa[h][j] = (uint32_t)(ra * (int)p.lda + schunk * 8) * 2u;
            ow[h][j] = (uint32_t)(rw * (int)p.ldw + schunk * 8) * 2u;
        }
    char* const lds_piece = smem + wid * 1024;       // + slot + j * 8192
    auto stage_a = [&](int buf, int h, int kt) {
        const char* src = a_blk + kt * (G2_BK * 2);
        pv_glds16<PV_GEMM_A_AUX>(src + oa[h][0], lds_piece + buf * G2_BUF + h * G2_HALF);
        pv_glds16<PV_GEMM_A_AUX>(src + oa[h][1], lds_piece + buf * G2_BUF + h * G2_HALF + 8192);
    };
    auto stage_b = [&](int buf, int h, int kt) {
        const char* src = w_blk + kt * (G2_BK * 2);
        pv_glds16<PV_GEMM_W_AUX>(src + ow[h][0], lds_piece + buf * G2_BUF + (2 + h) * G2_HALF);
        pv_glds16<PV_GEMM_W_AUX>(src + ow[h][1], lds_piece + buf * G2_BUF + (2 + h) * G2_HALF + 8192);
    };
    // PF: K-tile 0 (all four half-tile slots of buffer 0) of ANOTHER tile - the next one of this workgroup's list - in eight 1-KiB-per-wave
    // pieces (0-3: A half h = pc >> 1, part j = pc & 1; 4-7: W likewise) + piece 8 = the tile's 256 bias values (one float per lane of waves
    // 0 - 3).  Nothing here may depend on a value carried (= spilled) across the K loop: the epilogue rebuilds `lane` before the first piece.
    auto stage_next_piece = [&](int tm0, int tn0, int pc) __attribute__((always_inline)) {
        if (pc == 8) {
            if (p.bias && wid < 4) {          // (wave-uniform)
                const int c_ = tn0 + wid * 64 + lane < p.N ? tn0 + wid * 64 + lane : p.N - 1;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.bias + c_),
                                                 (__attribute__((address_space(3))) void*)(smem + PF_BIAS_BASE + wid * 256), 4, 0, 0);
            }
            return;
        }
        const int h = (pc >> 1) & 1, jj = pc & 1;
        const int sr = wid * 8 + (lane >> 3), sc = (lane & 7) ^ ((lane >> 3) & 7);
        if (pc < 4) {
            const char* const na = reinterpret_cast<const char*>(p.A + (int64_t)tm0 * p.lda);
            int ra = h * 128 + jj * 64 + sr; ra = tm0 + ra < p.M ? ra : p.M - 1 - tm0;
            pv_glds16<PV_GEMM_A_AUX>(na + (uint32_t)(ra * (int)p.lda + sc * 8) * 2u, lds_piece + h * G2_HALF + jj * 8192);
        } else {
            const char* const nw = reinterpret_cast<const char*>(p.W + (int64_t)tn0 * p.ldw);
            const int q = jj * 64 + sr;
            int rw = h * 128 + (q & ~31) + (((q & 15) >> 2) << 3) + (((q >> 4) & 1) << 2) + (q & 3);
            rw = tn0 + rw < p.N ? rw : p.N - 1 - tn0;
            pv_glds16<PV_GEMM_W_AUX>(nw + (uint32_t)(rw * (int)p.ldw + sc * 8) * 2u, lds_piece + (2 + h) * G2_HALF + jj * 8192);
        }
    };

    // ---- fragment read addresses: one LDS-address-space base per (buffer, k-step), every read = base + immediate ----
    typedef __attribute__((address_space(3))) const char lds_cc;
    const int frow = (lane & 15) * 128;
    const int fx0 = ((lane >> 4) ^ (lane & 7)) << 4;
    lds_cc* const lds0 = (lds_cc*)smem;
    lds_cc* a_rd[2][2];   // [buf][ks]
    lds_cc* b_rd[2][2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            a_rd[b][ks] = lds0 + b * G2_BUF + wr * G2_HALF + frow + (fx0 ^ (ks << 6));
            b_rd[b][ks] = lds0 + b * G2_BUF + (2 + (wc >> 1)) * G2_HALF + (wc & 1) * 8192 + frow + (fx0 ^ (ks << 6));
            asm volatile("" : "+v"(a_rd[b][ks]));   // opaque: keep these 8 bases in VGPRs, never re-derive per read
            asm volatile("" : "+v"(b_rd[b][ks]));
        }

    // accumulators start from the bias: lane (g = lane>>4) owns columns en0 + (nt>>1)*32 + g*8 + (nt&1)*4 + 0..3 of tile nt
    int en0 = n0 + wc * 64 + ((lane >> 4) << 3);
    f32x4 acc[4][8];   // [nt][mt]
    if (!PF) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
            if (p.bias && en0 + (i >> 1) * 32 + (i & 1) * 4 < p.N) b4 = *reinterpret_cast<const f32x4*>(p.bias + en0 + (i >> 1) * 32 + (i & 1) * 4);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = b4;
        }
    }
    bf16x8 af[2][4][2];   // [m half][mt][ks]
    bf16x8 bfr[2][2];     // [nt][ks] of the CURRENT n half (n0 lives P1-P2, n1 lives P3-P4: one register set)

#define G2_READ_A(BUF, MH)                                                                                  \
    _Pragma("unroll") for (int t_ = 0; t_ < 4; ++t_) _Pragma("unroll") for (int ks_ = 0; ks_ < 2; ++ks_)  \
        af[MH][t_][ks_] = *reinterpret_cast<const __attribute__((address_space(3))) bf16x8*>(a_rd[BUF][ks_] + ((MH) * 4 + t_) * 2048);
#define G2_READ_B(BUF, NH)                                                                                  \
    _Pragma("unroll") for (int t_ = 0; t_ < 2; ++t_) _Pragma("unroll") for (int ks_ = 0; ks_ < 2; ++ks_)  \
        bfr[t_][ks_] = *reinterpret_cast<const __attribute__((address_space(3))) bf16x8*>(b_rd[BUF][ks_] + ((NH) * 2 + t_) * 2048);
#define G2_SYNC_LOADS()                                  \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   \
    __builtin_amdgcn_s_barrier();                        \
    __builtin_amdgcn_sched_barrier(0);
#define G2_MFMA(MH, NH)                                                                                              \
    __builtin_amdgcn_s_setprio(1);                                                                                   \
    _Pragma("unroll") for (int ks_ = 0; ks_ < 2; ++ks_) _Pragma("unroll") for (int n_ = 0; n_ < 2; ++n_)            \
        _Pragma("unroll") for (int m_ = 0; m_ < 4; ++m_)                                                            \
            acc[(NH) * 2 + n_][(MH) * 4 + m_] = PV_MFMA_16x16x32(                            \
                bfr[n_][ks_], af[MH][m_][ks_], acc[(NH) * 2 + n_][(MH) * 4 + m_], 0, 0, 0);                     \
    __builtin_amdgcn_s_setprio(0);                                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                               \
    __builtin_amdgcn_s_barrier();                                                                                    \
    __builtin_amdgcn_sched_barrier(0);

    // one K-tile.  SB: stage the B halves of tile kt+1 (P1,P2).  SA: stage the A halves of tile kt+2 (P3,P4); without
    // SA the youngest loads in flight are B(kt+1) themselves, so P4 must drain (vmcnt(0)) instead of vmcnt(4).
    auto ktile = [&](auto buf_c, auto sb_c, auto sa_c, int kt) {
        constexpr int BUF = decltype(buf_c)::value;
        constexpr bool SB = decltype(sb_c)::value, SA = decltype(sa_c)::value;
        // P1
        G2_READ_B(BUF, 0)
        __builtin_amdgcn_sched_barrier(0);
        G2_READ_A(BUF, 0)
        if (SB) stage_b(BUF ^ 1, 0, kt + 1);
        G2_SYNC_LOADS()
        G2_MFMA(0, 0)
        // P2
        G2_READ_A(BUF, 1)
        if (SB) stage_b(BUF ^ 1, 1, kt + 1);
        G2_SYNC_LOADS()
        G2_MFMA(1, 0)
        // P3
        G2_READ_B(BUF, 1)
        if (SA) stage_a(BUF, 0, kt + 2);
        G2_SYNC_LOADS()
        G2_MFMA(1, 1)
        // P4
        if (SA) {
            stage_a(BUF, 1, kt + 2);
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        G2_SYNC_LOADS()
        G2_MFMA(0, 1)
    };
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;
    using T = std::true_type;
    using F = std::false_type;

    // ---- prologue: tile 0 complete + A halves of tile 1 in flight ------------------------------------------------
    const int nk = p.K / G2_BK;        // even, >= 2 (checked on the host)
    if (PF) __builtin_assume(nk >= 4);   // (the launcher keeps K < 256 off the persistent kernel: the main loop runs at least once, no bypass edge)
    PV_STAMP(0);
#ifdef PV_STAMPS
    if (threadIdx.x == 0 && p.dbg) { unsigned long long rt_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt_)::"memory"); p.dbg[pv_stamp_slot * 16 + 6] = rt_; }
#endif
    const bool own_tile0 = !PF || !PFM || first;          // (workgroup-uniform)
    if (PF && (first || !PFM) && p.bias && wid < 4) {       // (wave-uniform) this tile's bias values: OLDER than K-tile 0, so the wait for that covers them
        const int c_ = n0 + wid * 64 + lane < p.N ? n0 + wid * 64 + lane : p.N - 1;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.bias + c_),
                                         (__attribute__((address_space(3))) void*)(smem + PF_BIAS_BASE + wid * 256), 4, 0, 0);
    }
    if (own_tile0) { stage_a(0, 0, 0); stage_a(0, 1, 0); stage_b(0, 0, 0); stage_b(0, 1, 0); }
    stage_a(1, 0, 1); stage_a(1, 1, 1);
    constexpr bool HAS_TAB = EPI == PV_EPI_BIAS_GELU_BF16 || EPI == PV_EPI_BIAS_GELU_SPLIT_BF16 || EPI == PV_EPI_BIAS_GELU_PAIR_BF16;
    constexpr int NTAB = HAS_TAB ? 2 : 0;
    if (HAS_TAB && own_tile0) {
        // GELU table (16 KiB replicated cubic) into the LDS above the staging buffers.  Issued AFTER the
        // first K tiles (round 3; round 2 issued it first, so the K loop could not start before the table had landed: fc1's prologue took
        // 4.0 k cycles against QKV's 2.4 k): the wait below leaves it in flight, and the first counted wait of the K loop - which leaves
        // only the 4 youngest operations, all issued later - covers it long before the epilogue reads it.
#pragma unroll
        for (int i = 0; i < NTAB; ++i)
            pv_glds16(reinterpret_cast<const char*>(pv_gelu_cub) + (i * 512 + tid) * 16,
                      smem + G2_LDS + (i * 512 + wid * 64) * 16);
    }
    // LayerNorm folding (consumer): the tile's row statistics [256][2] and column constants c1[256], c2[256] (4 KiB) behind the table,
    // one 4-byte LDS-DMA each per thread (clamped at the ragged edges: those rows / columns are never stored) - two more operations
    // that the counted waits of the K loop cover; the epilogue reads them from LDS instead of waiting for L2
    const bool fold_dma = PV_EPI_PIPE && (EPI == PV_EPI_BIAS_BF16 || EPI == PV_EPI_BIAS_GELU_BF16) && p.fold_stat != nullptr;
    if (fold_dma) {
        constexpr int FOLD_BASE = G2_LDS + (EPI == PV_EPI_BIAS_GELU_BF16 ? PV_GELU_CUB_N * PV_GELU_CUB_REP * 16 : 0);
        const int r_ = m0 + (tid >> 1) < p.M ? m0 + (tid >> 1) : p.M - 1;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.fold_stat + 2 * (int64_t)r_ + (tid & 1)),
                                         (__attribute__((address_space(3))) void*)(smem + FOLD_BASE + wid * 256), 4, 0, 0);
        const int c_ = n0 + (tid & 255) < p.N ? n0 + (tid & 255) : p.N - 1;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((tid < 256 ? p.fold_c1 : p.fold_c2) + c_),
                                         (__attribute__((address_space(3))) void*)(smem + FOLD_BASE + 2048 + wid * 256), 4, 0, 0);
    }
    // tile 0 has landed for this thread; the A halves of tile 1 (4 operations), the table and the fold constants stay in flight.
    // (PF, not the first tile: K-tile 0 landed during the previous epilogue - every wave drained its share before its first store - and the
    //  barrier at the tile boundary has been passed: nothing to wait for; what was just issued is covered by the K loop's first counted wait.)
    if (own_tile0) {
        const int inflight = 4 + NTAB + (fold_dma ? 2 : 0);      // workgroup-uniform; the immediate must be literal
        if (inflight == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (inflight == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (inflight == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    __builtin_amdgcn_sched_barrier(0);
    PV_STAMP(1);
    if (PF) {          // accumulators start from the bias: the same values as the register loads of the one-tile kernel, read from the LDS copy
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
            if (p.bias) b4 = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>((const __attribute__((address_space(3))) char*)smem + PF_BIAS_BASE + (wc * 64 + ((lane >> 4) << 3) + (i >> 1) * 32 + (i & 1) * 4) * 4);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = b4;
        }
    }
    if (wr == 1) __builtin_amdgcn_s_barrier();   // stagger: group 1 runs one barrier interval behind group 0

    int kt = 0;
    for (; kt + 4 <= nk; kt += 2) {              // tiles with kt+2 < nk: full staging
        ktile(B0{}, T{}, T{}, kt);
        ktile(B1{}, T{}, T{}, kt + 1);
    }
    ktile(B0{}, T{}, F{}, kt);                   // tile nk-2: only B(nk-1) left to stage, then drain
    ktile(B1{}, F{}, F{}, kt + 1);               // tile nk-1
    if (wr == 0) __builtin_amdgcn_s_barrier();   // balance the stagger barrier
    PV_STAMP(2);
#undef G2_READ_A
#undef G2_READ_B
#undef G2_SYNC_LOADS
#undef G2_MFMA

    // ---- epilogue: transpose the C tile through LDS (the staging buffers are free now) so that every global access is a
    // whole contiguous row segment: one wave-instruction = 1 KiB of one (fp32) or two (bf16) output rows ------------
    typedef __attribute__((address_space(3))) char lds_c;
    lds_c* const cimg = (lds_c*)smem;
    if (PF) {                  // fresh lane index for the epilogue: no thread-index arithmetic is carried through the K loop
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
        tid = wid * 64 + lane;
        en0 = n0 + wc * 64 + ((lane >> 4) << 3);
    }
    const int g = lane >> 4, i16 = lane & 15;
    // PF: the next tile's first K-tile goes into buffer 0 during this epilogue - its images live in buffer 1 only - so that the 64 KiB travel
    // under the epilogue's arithmetic instead of in front of the next K loop.  It is OLDER than every store of the epilogue, so a wait that
    // leaves the stores in flight retires it: mode 1 issues it here and waits (vmcnt(0)) in front of the first store; mode 2 (shipped) issues
    // it in pieces below and waits with a count at the end of the epilogue.
    const bool pf_next = PF && PFM && has_next;
    if (PFM == 1 && pf_next) {
#pragma unroll
        for (int pc = 0; pc < 9; ++pc) stage_next_piece(nm0, nn0, pc);
    }
    if (PV_EPI_PIPE && (EPI == PV_EPI_BIAS_BF16 || EPI == PV_EPI_BIAS_GELU_BF16 || EPI == PV_EPI_BIAS_GELU_PAIR_BF16)) {
        // Round 3: the 16-bit epilogue as a software pipeline (in-kernel stamps of round 2's form, scripts/stamp_gemm.py: fc1 + folded
        // LayerNorm spent 18.9 k ticks here against 27.7 k in its K loop).  What was wrong, from the stamps and the ISA:
        //   - GELU ran latency-bound: hipcc kept 2 table gathers in flight per wave (256 VGPRs, all accumulators + 48 fold constants
        //     live) and SLP-packed the polynomial into v_pk_fma_f32 behind v_mov shuffles: 3.6 k ticks per 32 values per lane;
        //   - the 128 KiB of stores followed all the arithmetic, and a wave blocks at a store's ISSUE once the CU's memory path
        //     (~28 B/clk) has its queue full: the path and the VALU were never busy together;
        //   - the fold statistics were fetched from L2 at the start of the epilogue: 1.6 - 1.8 k ticks of exposed latency.
        // Now: (1) the fold constants (row statistics, c1, c2: 4 KiB) are staged into LDS by the PROLOGUE's LDS-DMA and applied to all
        // accumulators first (pure VALU), after which their registers are dead; (2) the lane's 16 units of 8 values (one 16-byte
        // chunk of one image row each) run through [8 gathers issued | previous unit: polynomial, pack, LDS image write], the
        // gathers of unit n+1 in flight under the arithmetic of unit n (inline-asm ds_read_b128 + counted lgkmcnt, results threaded
        // through the wait so nothing is consumed early); (3) four units = one pass of 64 image rows: after its barrier every wave
        // reads 8 whole rows back and issues their four 1-KiB stores ONE PER UNIT of the next pass, i.e. at the rate the memory path
        // drains them.  Arithmetic per element is unchanged (bit-identical to the one-pass form, scripts/gemm_epi_ab.py).
        float vmax = 0.f;          // operand-range guard (fp16 build): largest magnitude this lane packs
        const bool fold = p.fold_stat != nullptr;
        constexpr int FOLD_BASE = G2_LDS + (EPI != PV_EPI_BIAS_BF16 ? PV_GELU_CUB_N * PV_GELU_CUB_REP * 16 : 0);
        // PAIR (training forward of fc1): TWO output planes from the same accumulators - 32 virtual units, PLANE-MINOR (round 6): virtual unit vn is
        // accumulator unit n = vn >> 1, plane vn & 1 - plane 0 its GELU, plane 1 gelu'(pre-activation) = the derivative of the SAME gathered cubic (saved
        // for backward; rounds 1-5 packed the raw pre-activation in units 0-15 and the GELU in units 16-31).  One set of gathers per accumulator unit
        // serves both planes (issued two virtual units ahead).  A pass (4 virtual units) is now ONE 16-row tile of both wave groups x both planes:
        // 64 image row slots = plane * 32 + wave group * 16 + row; waves 0-3 read back and store plane 0, waves 4-7 plane 1; 8 passes.
        constexpr bool PAIR2 = EPI == PV_EPI_BIAS_GELU_PAIR_BF16;
        constexpr int NU = PAIR2 ? 32 : 16;
        if (fold) {
            // x = rstd[m] * (acc - mean[m] * c1[n]) + c2[n]: the accumulators hold x16 . (gamma (.) W)^T
            f32x4 k1[2][2], k2[2][2];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const int col = wc * 64 + u * 32 + g * 8 + hh * 4;
                    k1[u][hh] = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(cimg + FOLD_BASE + 2048 + col * 4);
                    k2[u][hh] = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(cimg + FOLD_BASE + 3072 + col * 4);
                }
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) {
                const pv_f32x2_t st = *reinterpret_cast<const __attribute__((address_space(3))) pv_f32x2_t*>(cimg + FOLD_BASE + (wr * 128 + mt * 16 + i16) * 8);
                const float f_mean = st[0], f_rstd = st[1];
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc[2 * u][mt][e] = fmaf(f_rstd, fmaf(-f_mean, k1[u][0][e], acc[2 * u][mt][e]), k2[u][0][e]);
                        acc[2 * u + 1][mt][e] = fmaf(f_rstd, fmaf(-f_mean, k1[u][1][e], acc[2 * u + 1][mt][e]), k2[u][1][e]);
                    }
            }
        }
        PV_STAMP(8);
        const uint32_t tabc = (uint32_t)(uintptr_t)((const __attribute__((address_space(3))) f32x4*)(cimg + G2_LDS) + i16) - 0x40000000u;
        const bool ocol_ok = n0 + (lane & 31) * 8 < p.N;
        // + 32 q + 2 j: the rows this wave stores in pass q (PAIR: + 16 q + 2 j, wave group (wid >> 1) & 1, rows (wid & 1) * 8 .. of the pass's 16-row tile)
        const int rb_row0 = (EPI == PV_EPI_BIAS_GELU_PAIR_BF16 ? ((wid >> 1) & 1) * 128 + (wid & 1) * 8 : (wid >> 2) * 128 + (wid & 3) * 8) + (lane >> 5);
        // The image of a pass: 64 rows x 512 B = 32 KiB, local row lr = (wave group) * 32 + (tile row % 32); two such regions alternate
        // (pass q -> region q & 1) inside K-tile buffer 1 ONLY: buffer 0 stays free, so the prefetching launch can stage the next
        // tile's first K-tile into it while this epilogue runs.  Region q & 1 is rewritten in pass q + 2, after barrier q + 1 - by
        // which every wave has completed its read-back of pass q (it needed the rows for the stores it issued during pass q + 1).
        lds_c* const img = cimg + G2_BUF;
        const int rb_lrow0 = (wid >> 2) * 32 + (wid & 3) * 8 + (lane >> 5);      // + 2 j
        uint32_t rb_addr[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = rb_lrow0 + 2 * j;
            rb_addr[j] = (uint32_t)(uintptr_t)img + row * 512 + (((lane & 31) ^ (row & 7)) << 4);
        }
        f32x4 cf[2][8];            // gathered table entries of two units in flight (GELU units only)
        u32x4 rb[4];               // rows read back from the image, waiting for their store slot
        // virtual unit vn: accumulator unit n = vn & 15 (16-row tile mt = n >> 1, column half u = n & 1), plane vn >> 4
        auto acc_unit = [&](int vn) -> int { return PAIR2 ? vn >> 1 : vn; };                           // accumulator unit: 16-row tile mt = n >> 1, column half u = n & 1
        auto cf_slot = [&](int vn) -> int { return acc_unit(vn) & 1; };
        auto unit_x = [&](int vn, int e) -> float { const int n = acc_unit(vn); return acc[2 * (n & 1) + (e >> 2)][n >> 1][e & 3]; };
        auto is_gelu = [&](int vn) -> bool { return EPI == PV_EPI_BIAS_GELU_BF16 || PAIR2; };         // units that evaluate a table entry per value
        auto gathers = [&](int vn) -> bool { return EPI == PV_EPI_BIAS_GELU_BF16 || (PAIR2 && (vn & 1) == 0); };      // ... and ISSUE its gathers (PAIR: once per accumulator unit)
        auto is_deriv = [&](int vn) -> bool { return PAIR2 && (vn & 1) == 1; };
        auto issue = [&](int vn) __attribute__((always_inline)) {
            if (gathers(vn)) {
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    uint32_t b0, b1;
                    pv_gelu_bits2(unit_x(vn, e), unit_x(vn, e + 1), b0, b1);
                    asm volatile("ds_read_b128 %0, %1" : "=v"(cf[cf_slot(vn)][e]) : "v"((b0 << 8) + tabc));
                    asm volatile("ds_read_b128 %0, %1" : "=v"(cf[cf_slot(vn)][e + 1]) : "v"((b1 << 8) + tabc));
                }
            }
        };
        // the lgkmcnt wait that makes unit vn's gathers (and everything older: the read-back rows of the previous pass) valid.  LDS
        // operations return in order, so "at most N outstanding" = all but the N youngest have completed; the values pass through the
        // statement, so no consumer can be scheduled above it.  with_rb: the read-back registers are threaded through as well (first
        // unit after a read-back).
        auto wait_unit = [&](int vn, bool next_in_flight, bool with_rb) __attribute__((always_inline)) {
            if (gathers(vn)) {
                f32x4(&c)[8] = cf[cf_slot(vn)];
                if (next_in_flight && with_rb)
                    asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]),
                                 "+v"(rb[0]), "+v"(rb[1]), "+v"(rb[2]), "+v"(rb[3]));
                else if (next_in_flight)
                    asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]));
                else if (with_rb)
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]),
                                 "+v"(rb[0]), "+v"(rb[1]), "+v"(rb[2]), "+v"(rb[3]));
                else
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]));
            } else if (with_rb) {
                // (a plain unit whose successor's gathers are already in flight - unit 15 of PAIR - never carries a read-back: 15 & 3 != 0)
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(rb[0]), "+v"(rb[1]), "+v"(rb[2]), "+v"(rb[3]));
            }
        };
        auto finish = [&](int vn) __attribute__((always_inline)) {
            const int n = acc_unit(vn), mt = n >> 1, u = n & 1;
            float y[8];
            if (is_gelu(vn)) {
                // pv_gelu_poly for the 8 values in lock step (step by step ACROSS the values): no instruction depends on its predecessor,
                // so the wave issues back to back (value by value, hipcc padded every dependent pair of asm steps with an s_nop)
                float x[8];
                if (is_deriv(vn)) {
                    // pv_gelu_dpoly in lock step: c1 + x (2 c2 + 3 c3 x) from the entry {c0, c2, c1, c3}
                    float t[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) { x[e] = unit_x(vn, e); y[e] = pv_mul_s(cf[cf_slot(vn)][e][3], 3.0f); }
#pragma unroll
                    for (int e = 0; e < 8; ++e) t[e] = pv_add_s(cf[cf_slot(vn)][e][1], cf[cf_slot(vn)][e][1]);
#pragma unroll
                    for (int e = 0; e < 8; ++e) t[e] = pv_fma_s(y[e], x[e], t[e]);
#pragma unroll
                    for (int e = 0; e < 8; ++e) y[e] = pv_fma_s(x[e], t[e], cf[cf_slot(vn)][e][2]);
                } else {
                pv_f32x2_t r[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    x[e] = unit_x(vn, e);
                    const f32x4 c = cf[cf_slot(vn)][e];
                    r[e] = __builtin_elementwise_fma((pv_f32x2_t){c[2], c[3]}, (pv_f32x2_t){x[e], x[e]}, (pv_f32x2_t){c[0], c[1]});
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) y[e] = pv_mul_s(x[e], r[e][1]);
#pragma unroll
                for (int e = 0; e < 8; ++e) y[e] = pv_fma_s(x[e], y[e], r[e][0]);
                }
            } else {
                // the lane's 8 consecutive columns start at en0 + 32u: q-scaling is uniform per such chunk (qcols % 8 == 0)
                const float qs = (EPI == PV_EPI_BIAS_BF16 && en0 + u * 32 < p.qcols) ? p.qscale : 1.0f;
#pragma unroll
                for (int e = 0; e < 8; ++e) y[e] = unit_x(vn, e) * qs;
            }
            const u32x4 pk = {pv_pack_bf16x2_tracked(y[0], y[1], vmax), pv_pack_bf16x2_tracked(y[2], y[3], vmax),
                              pv_pack_bf16x2_tracked(y[4], y[5], vmax), pv_pack_bf16x2_tracked(y[6], y[7], vmax)};
            // image slot of the row: pass = 4 virtual units -> region (pass & 1); rows of the pass: two 16-row tiles of both wave groups (PAIR: one tile x both planes)
            const int lrow = PAIR2 ? (vn & 1) * 32 + wr * 16 + i16 : wr * 32 + (mt & 1) * 16 + i16, c = wc * 8 + u * 4 + g;
            const int region = PAIR2 ? mt & 1 : (mt >> 1) & 1;
            *reinterpret_cast<__attribute__((address_space(3))) u32x4*>(img + region * 32768 + lrow * 512 + ((c ^ (i16 & 7)) << 4)) = pk;
        };
        // Output addresses (round 4).  The stores of this epilogue leave in a fixed order (pass by pass, four 2-row stores per wave and pass), so
        // ONE running per-lane pointer walks them: first row of the wave + (lane >> 5) rows + (lane & 31) 16-byte chunks, advanced after every
        // store by a workgroup-uniform byte distance (2 rows, or 26 rows to the next pass; the training pair's planes are N elements apart) -
        // one 64-bit vector add per store.  Round 3 kept four 64-bit row pointers, which the allocator (251 of 256 registers live) re-derived
        // before every store: 7 vector instructions + a compare + an exec-mask branch per 1-KiB store, 16 % of this issue-bound epilogue's
        // instructions.  Full tiles (template argument FULL) have no bounds guard.
        // (a workgroup-uniform 64-bit base in scalar registers + ONE 32-bit running byte offset per lane: the tile spans < 4 MiB)
        const uint32_t row_bytes = (uint32_t)p.ldo * 2u;
        char* const obase = reinterpret_cast<char*>(p.out) + ((int64_t)m0 * p.ldo + n0) * 2;
        uint32_t ooff = (uint32_t)rb_row0 * row_bytes + (uint32_t)(lane & 31) * 16u
                        + ((PAIR2 && wid >= 4) ? (uint32_t)p.N * 2u : 0u);                 // (the pair: waves 4-7 store the derivative plane, N columns to the right)
        {
            auto store_row = [&](int vq, int j) __attribute__((always_inline)) {       // pass vq (PAIR: 16 tile rows per pass, otherwise 32)
                if (FULL || (m0 + rb_row0 + (PAIR2 ? 16 : 32) * vq + 2 * j < p.M && ocol_ok)) PV_STORE16(reinterpret_cast<u32x4*>(obase + ooff), rb[j]);
                // to the next store in program order: (vq, j + 1), else (vq + 1, 0)
                if (j < 3) ooff += 2u * row_bytes;
                else ooff += (PAIR2 ? 10u : 26u) * row_bytes;
            };
            issue(0);
#pragma unroll
            for (int vn = 0; vn < NU; ++vn) {
                const int vq = vn >> 2;
                // the next accumulator unit's gathers fly under this one's arithmetic (PAIR: under both planes of it)
                if (PAIR2) { if ((vn & 1) == 0 && vn + 2 < NU) issue(vn + 2); }
                else if (vn + 1 < NU) issue(vn + 1);
                if (PFM == 2 && vn < 4 && pf_next) {
                    // the next tile's first K-tile, two pieces per unit of the first pass: the memory path is idle until the first store (unit 4),
                    // and a burst of all 64 KiB at the start of the epilogue held every wave at the ISSUE of its loads for ~1 k cycles (stamps)
                    if (vn == 0) stage_next_piece(nm0, nn0, 8);
                    stage_next_piece(nm0, nn0, 2 * vn);
                    stage_next_piece(nm0, nn0, 2 * vn + 1);
                }
                wait_unit(vn, PAIR2 ? vn + 2 < NU : (vn + 1 < NU && gathers(vn + 1)), vq > 0 && (vn & 3) == 0);
                finish(vn);
                if (PFM == 1 && vn == 4 && pf_next) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the next tile's K-tile 0 has landed (before any store)
                if (vq > 0) store_row(vq - 1, vn & 3);                  // one 1-KiB store of the previous pass per unit
                if ((vn & 3) == 3) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // this pass's image writes (and the gathers issued before them)
                    if (vq < 4) { PV_STAMP(9 + vq); }
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < 4; ++j)      // local row rb_lrow0 + 2 j of region q & 1: base + immediate
                        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(rb[j]) : "v"(rb_addr[j]), "n"((vq & 1) * 32768));
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(rb[0]), "+v"(rb[1]), "+v"(rb[2]), "+v"(rb[3]));
#pragma unroll
            for (int j = 0; j < 4; ++j) store_row(NU / 4 - 1, j);
        }
        pv_range_commit(vmax, p.range_flag);
    } else
    if (EPI == PV_EPI_BIAS_BF16 || EPI == PV_EPI_BIAS_GELU_BF16 || EPI == PV_EPI_BIAS_GELU_SPLIT_BF16 || EPI == PV_EPI_BIAS_GELU_PAIR_BF16) {
        // bf16 image: 256 rows x 512 B, 16-B chunk c of row r stored at chunk c ^ (r & 7).  SPLIT (precision mode): the fp32
        // results stay in the accumulators; pass 0 stores their bf16 "hi" image to planes 0 and 2 of the [M, 3N] output, pass 1
        // the "lo" image (v - hi) to plane 1.
        // PAIR (training forward): pass 0 stores gelu'(pre-activation) to plane 1 of the [M, 2N] output, pass 1 its GELU to plane 0.
        constexpr bool SPLIT = EPI == PV_EPI_BIAS_GELU_SPLIT_BF16;
        constexpr bool PAIR = EPI == PV_EPI_BIAS_GELU_PAIR_BF16;
        float vmax = 0.f;          // operand-range guard (fp16 build): largest magnitude this lane packs
        // folded LayerNorm (consumer): the lane's 2 x 8 column constants, fetched ONCE (columns en0 + 32u + 0..7, clamped reads)
        f32x4 fc1[2][2], fc2[2][2];
        if ((EPI == PV_EPI_BIAS_BF16 || EPI == PV_EPI_BIAS_GELU_BF16) && p.fold_stat) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                int cb = en0 + u * 32; cb = cb + 8 <= p.N ? cb : p.N - 8;
                fc1[u][0] = *reinterpret_cast<const f32x4*>(p.fold_c1 + cb); fc1[u][1] = *reinterpret_cast<const f32x4*>(p.fold_c1 + cb + 4);
                fc2[u][0] = *reinterpret_cast<const f32x4*>(p.fold_c2 + cb); fc2[u][1] = *reinterpret_cast<const f32x4*>(p.fold_c2 + cb + 4);
            }
        }
        // folded LayerNorm (consumer): the (mean, rstd) of this lane's 8 rows, fetched together BEFORE the image loop (inside it every
        // load would be followed by its own wait: eight L2 round trips per tile)
        float f_mean8[8], f_rstd8[8];
        if ((EPI == PV_EPI_BIAS_BF16 || EPI == PV_EPI_BIAS_GELU_BF16) && p.fold_stat) {
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) {
                const int row_ = wr * 128 + mt * 16 + i16;
                const int mrow = m0 + row_ < p.M ? m0 + row_ : p.M - 1;
                const float2 st = *reinterpret_cast<const float2*>(p.fold_stat + 2 * (int64_t)mrow);
                f_mean8[mt] = st.x; f_rstd8[mt] = st.y;
            }
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) { asm volatile("" : "+v"(f_mean8[mt])); asm volatile("" : "+v"(f_rstd8[mt])); }
        }
#pragma unroll
        for (int pass = 0; pass < (SPLIT || PAIR ? 2 : 1); ++pass) {
            if (pass == 1) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();                      // pass 0's image has been read by every wave
            }
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) {
                const int row = wr * 128 + mt * 16 + i16;
                float f_mean = 0.f, f_rstd = 1.f;
                if ((EPI == PV_EPI_BIAS_BF16 || EPI == PV_EPI_BIAS_GELU_BF16) && p.fold_stat) { f_mean = f_mean8[mt]; f_rstd = f_rstd8[mt]; }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    f32x4 lo = acc[2 * u][mt], hi = acc[2 * u + 1][mt];
                    if ((EPI == PV_EPI_BIAS_BF16 || EPI == PV_EPI_BIAS_GELU_BF16) && p.fold_stat) {
                        // folded LayerNorm: the accumulators hold x16 . (gamma (.) W)^T
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            lo[e] = fmaf(f_rstd, fmaf(-f_mean, fc1[u][0][e], lo[e]), fc2[u][0][e]);
                            hi[e] = fmaf(f_rstd, fmaf(-f_mean, fc1[u][1][e], hi[e]), fc2[u][1][e]);
                        }
                    }
                    // the lane's 8 consecutive columns start at en0 + 32u: q-scaling is uniform per such chunk (qcols % 8 == 0)
                    const float qs = (EPI == PV_EPI_BIAS_BF16 && en0 + u * 32 < p.qcols) ? p.qscale : 1.0f;
                    u32x4 pk;
                    if (PAIR) {
                        {
                            const __attribute__((address_space(3))) f32x4* tab = (const __attribute__((address_space(3))) f32x4*)(cimg + G2_LDS) + i16;
#pragma unroll
                            for (int e = 0; e < 4; e += 2) {
                                float t0 = lo[e], t1 = lo[e + 1], t2 = hi[e], t3 = hi[e + 1];
                                if (pass == 1) { pv_gelu_lut2(t0, t1, tab); pv_gelu_lut2(t2, t3, tab); }
                                else { pv_gelu_dlut2(t0, t1, tab); pv_gelu_dlut2(t2, t3, tab); }
                                lo[e] = t0; lo[e + 1] = t1; hi[e] = t2; hi[e + 1] = t3;
                            }
                        }
                        pk = (u32x4){pv_pack_bf16x2_tracked(lo[0], lo[1], vmax), pv_pack_bf16x2_tracked(lo[2], lo[3], vmax),
                                     pv_pack_bf16x2_tracked(hi[0], hi[1], vmax), pv_pack_bf16x2_tracked(hi[2], hi[3], vmax)};
                    } else if (EPI == PV_EPI_BIAS_GELU_BF16 || EPI == PV_EPI_BIAS_GELU_SPLIT_BF16) {
                        if (pass == 0) {
                            const __attribute__((address_space(3))) f32x4* tab = (const __attribute__((address_space(3))) f32x4*)(cimg + G2_LDS) + i16;
#pragma unroll
                            for (int e = 0; e < 4; e += 2) {
                                float t0 = lo[e], t1 = lo[e + 1], t2 = hi[e], t3 = hi[e + 1];
                                pv_gelu_lut2(t0, t1, tab); pv_gelu_lut2(t2, t3, tab);
                                lo[e] = t0; lo[e + 1] = t1; hi[e] = t2; hi[e + 1] = t3;
                            }
                            if (SPLIT) { acc[2 * u][mt] = lo; acc[2 * u + 1][mt] = hi; }
                        }
                        if (SPLIT) {
                            const PvHiLo s0 = pv_split2(lo[0], lo[1]), s1 = pv_split2(lo[2], lo[3]), s2 = pv_split2(hi[0], hi[1]), s3 = pv_split2(hi[2], hi[3]);
                            pk = pass == 0 ? (u32x4){s0.hi, s1.hi, s2.hi, s3.hi} : (u32x4){s0.lo, s1.lo, s2.lo, s3.lo};
                        } else {
                            pk = (u32x4){pv_pack_bf16x2_tracked(lo[0], lo[1], vmax), pv_pack_bf16x2_tracked(lo[2], lo[3], vmax),
                                         pv_pack_bf16x2_tracked(hi[0], hi[1], vmax), pv_pack_bf16x2_tracked(hi[2], hi[3], vmax)};
                        }
                    } else {
                        pk = (u32x4){pv_pack_bf16x2_tracked(lo[0] * qs, lo[1] * qs, vmax), pv_pack_bf16x2_tracked(lo[2] * qs, lo[3] * qs, vmax),
                                     pv_pack_bf16x2_tracked(hi[0] * qs, hi[1] * qs, vmax), pv_pack_bf16x2_tracked(hi[2] * qs, hi[3] * qs, vmax)};
                    }
                    const int c = wc * 8 + u * 4 + g;
                    *reinterpret_cast<__attribute__((address_space(3))) u32x4*>(cimg + row * 512 + ((c ^ (i16 & 7)) << 4)) = pk;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            uint16_t* const ob = reinterpret_cast<uint16_t*>(p.out) + n0 + (lane & 31) * 8;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int row = wid * 32 + 2 * j + (lane >> 5);
                const u32x4 v = *reinterpret_cast<const __attribute__((address_space(3))) u32x4*>(cimg + row * 512 + (((lane & 31) ^ (row & 7)) << 4));
                if (m0 + row < p.M && n0 + (lane & 31) * 8 < p.N) {
                    uint16_t* o = ob + (int64_t)(m0 + row) * p.ldo;
                    if (PAIR) {
                        *reinterpret_cast<u32x4*>(pass == 0 ? o + p.N : o) = v;
                    } else if (!SPLIT) {
                        *reinterpret_cast<u32x4*>(o) = v;
                    } else if (pass == 0) {
                        *reinterpret_cast<u32x4*>(o) = v;
                        *reinterpret_cast<u32x4*>(o + 2 * p.N) = v;
                    } else {
                        *reinterpret_cast<u32x4*>(o + p.N) = v;
                    }
                }
            }
        }
        if (!SPLIT) pv_range_commit(vmax, p.range_flag);
    } else {
        // fp32 image: 128 rows x 1 KiB per pass (pass ps = the rows of wave group wr == ps), chunk c of row r at c ^ (r & 7)
        // PV_EPI_GELU_GRAD_BF16 (round 6): its rows are 16-BIT on both sides (the saved derivative in, the gradient out: 512 B each), and a wave instruction that
        // moves a row of them at 8 bytes per lane is half a request - the epilogue ran at 14.7 B/clk where the fp32 forms reach 23.  Here a lane owns 8 columns
        // and a wave instruction TWO rows (lanes 0-31 / 32-63): 16-byte loads of the derivative, two 16-byte reads of the fp32 image, 16-byte stores.
        constexpr bool DG = EPI == PV_EPI_GELU_GRAD_BF16;
        float cs8[DG ? 8 : 1];                      // this lane's 8 columns summed over the rows it stores
        if constexpr (DG) {
#pragma unroll
            for (int e = 0; e < 8; ++e) cs8[e] = 0.f;
        }
        u32x4 dg[DG ? 2 : 1][DG ? 4 : 1];          // the derivative rows of two passes in flight
        const int dg_col = n0 + 8 * (lane & 31);
        const bool dg_col_ok = FULL || dg_col + 8 <= p.N;
        float vmax = 0.f;                           // operand-range guard of the x16_out copy (fp16 build)
        // FOUR passes of 64 tile rows; the fp32 image of a pass (64 rows x 1 KiB, chunk c of row r at c ^ (r & 7)) lives in K-tile buffer 1
        // ONLY - buffer 0 stays free for the prefetching launch to stage the next tile's first K-tile during this epilogue.  Pass q: the
        // wave group that holds rows 64 q .. 64 q + 63 (wr = q >> 1, its 16-row tiles 4 (q & 1) .. + 3) writes them, then every wave takes 8
        // whole rows: residual / positional / pre-activation row + image row -> result row (1 KiB per instruction).
        // Round 3: the rows of pass q + 2 are requested BEFORE the stores of pass q + 1 are issued - VMEM order L0 L1 S0 L2 S1 L3 S2 S3 -
        // because the counter behind s_waitcnt vmcnt is in order: round 2 issued its second batch of loads behind the first batch of
        // stores, so the wait for the rows also waited for 192 KiB of stores to drain, and the CU's memory path (the bound of this
        // epilogue: 640 KiB per tile at ~28 B/clk) idled in between.  Same arithmetic per element: bit-identical outputs.
        lds_c* const img = cimg + G2_BUF;
        const bool col_ok = FULL || n0 + lane * 4 < p.N;  // ragged last column tile (N % 256 != 0)
        const int ncol = col_ok ? n0 + lane * 4 : 0;
        f32x4 rr[2][8];
        float rsc[2] = {1.0f, 1.0f};                       // PV_EPI_BIAS_RES_F32 row scale: lane j (< 8) holds slot j's (loaded with the rows: never after a store)
        auto row_of = [&](int q, int j) -> int64_t {       // output row of slot j of pass q (clamped at the ragged bottom edge)
            int m = m0 + q * 64 + wid * 8 + j;
            m = (FULL || m < p.M) ? m : p.M - 1;
            if (EPI == PV_EPI_BIAS_POS_F32) { const int img_ = m / p.rpi, pi = m - img_ * p.rpi; return (int64_t)img_ * p.rpo + p.row_off + pi; }
            return m;
        };
        auto fetch = [&](int q) __attribute__((always_inline)) {
            if constexpr (DG) {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    int m = m0 + q * 64 + wid * 8 + 2 * jj + (lane >> 5);
                    m = (FULL || m < p.M) ? m : p.M - 1;
                    dg[q & 1][jj] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const uint16_t*>(p.res) + (int64_t)m * p.ldr + (dg_col_ok ? dg_col : 0));
                }
                return;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                int m = m0 + q * 64 + wid * 8 + j;
                m = (FULL || m < p.M) ? m : p.M - 1;
                f32x4& r = rr[q & 1][j];
                if (EPI == PV_EPI_BIAS_F32) {
                    r = (f32x4){0.f, 0.f, 0.f, 0.f};
                } else if (EPI == PV_EPI_BIAS_RES_F32) {
                    r = *reinterpret_cast<const f32x4*>(p.res + (int64_t)m * p.ldr + ncol);
                } else {
                    const int img_ = m / p.rpi, pi = m - img_ * p.rpi;
                    r = *reinterpret_cast<const f32x4*>(p.pos + (int64_t)(p.row_off + pi) * p.N + ncol);
                }
            }
            if (EPI == PV_EPI_BIAS_RES_F32 && p.row_scale) {
                int m = m0 + q * 64 + wid * 8 + (lane & 7);
                rsc[q & 1] = p.row_scale[m < p.M ? m : p.M - 1];
            }
        };
        fetch(0);
        fetch(1);
        if (PFM == 2 && pf_next) {          // behind the rows the first passes wait for, in front of every store
#pragma unroll
            for (int pc = 0; pc < 9; ++pc) stage_next_piece(nm0, nn0, pc);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (q > 0) __builtin_amdgcn_s_barrier();      // the previous pass's image has been consumed by every wave
            if (wr == (q >> 1)) {
#pragma unroll
                for (int ml = 0; ml < 4; ++ml) {
                    const int mt = 4 * (q & 1) + ml, row = ml * 16 + i16;
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        const int c = wc * 16 + (nt >> 1) * 8 + g * 2 + (nt & 1);
                        *reinterpret_cast<__attribute__((address_space(3))) f32x4*>(img + row * 1024 + ((c ^ (i16 & 7)) << 4)) = acc[nt][mt];
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (PFM == 1 && q == 0 && pf_next) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the next tile's K-tile 0 (and this tile's first rows) before any store
            float fs[8], fq[8];            // LayerNorm folding (producer) / rank norms: per-lane partial (sum, sum of squares) of the 8 rows
            if constexpr (DG) {
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    const int row = wid * 8 + 2 * jj + (lane >> 5), c0 = 2 * (lane & 31);
                    const f32x4 va = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(img + row * 1024 + ((c0 ^ (row & 7)) << 4));
                    const f32x4 vb = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(img + row * 1024 + (((c0 + 1) ^ (row & 7)) << 4));
                    const u32x4 d = dg[q & 1][jj];
                    // (scalar products: the next derivative rows are still returning into registers)
                    const u32x4 pk = {pv_pack_bf16x2(pv_mul_s(va[0], pv_unpack_lo(d[0])), pv_mul_s(va[1], pv_unpack_hi(d[0]))),
                                      pv_pack_bf16x2(pv_mul_s(va[2], pv_unpack_lo(d[1])), pv_mul_s(va[3], pv_unpack_hi(d[1]))),
                                      pv_pack_bf16x2(pv_mul_s(vb[0], pv_unpack_lo(d[2])), pv_mul_s(vb[1], pv_unpack_hi(d[2]))),
                                      pv_pack_bf16x2(pv_mul_s(vb[2], pv_unpack_lo(d[3])), pv_mul_s(vb[3], pv_unpack_hi(d[3])))};
                    const int mrow = m0 + q * 64 + row;
                    if (FULL || (mrow < p.M && dg_col_ok)) {
                        PV_STORE16(reinterpret_cast<u32x4*>(reinterpret_cast<uint16_t*>(p.out) + (int64_t)mrow * p.ldo + dg_col), pk);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {               // column sums of exactly the stored values
                            cs8[2 * e] = pv_add_s(cs8[2 * e], pv_unpack_lo(pk[e]));
                            cs8[2 * e + 1] = pv_add_s(cs8[2 * e + 1], pv_unpack_hi(pk[e]));
                        }
                    }
                }
            } else
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int row = wid * 8 + j;                        // local image row; tile row 64 q + row
                const int64_t orow = row_of(q, j);
                const f32x4 v = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(img + row * 1024 + ((lane ^ (row & 7)) << 4));
                const f32x4 r = rr[q & 1][j];
                // row scale of slot j: lane j of the set holds it (one register per set instead of eight)
                const float sc = (EPI == PV_EPI_BIAS_RES_F32 && p.row_scale)
                                     ? __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rsc[q & 1]), j)) : 1.0f;
                f32x4 o;
                if (EPI == PV_EPI_BIAS_F32) {
                    const float qs = ncol < p.qcols ? p.qscale : 1.0f;
                    o = (f32x4){v[0] * qs, v[1] * qs, v[2] * qs, v[3] * qs};
                } else if (EPI == PV_EPI_BIAS_RES_F32) {
                    const float tr = p.res_scaled ? sc : 1.0f;
                    o = (f32x4){fmaf(sc, v[0], tr * r[0]), fmaf(sc, v[1], tr * r[1]), fmaf(sc, v[2], tr * r[2]), fmaf(sc, v[3], tr * r[3])};
                }
                else o = (f32x4){r[0] + v[0], r[1] + v[1], r[2] + v[2], r[3] + v[3]};
                const bool ok = FULL || (m0 + q * 64 + row < p.M && col_ok);
                if (ok) PV_STORE32(reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + orow * p.ldo + ncol), o);
                if (EPI == PV_EPI_BIAS_RES_F32 && p.rowsq_out)      // (workgroup-uniform) token norms for the next block's ranking
                    fq[j] = ok ? pv_add_s(pv_add_s(o[0] * o[0], o[1] * o[1]), pv_add_s(o[2] * o[2], o[3] * o[3])) : 0.f;     // reduced over the lanes after the pass (pv_add_s: no packed horizontal add under the row loads in flight)
                if (EPI == PV_EPI_BIAS_RES_F32 && p.x16_out) {      // (workgroup-uniform) LayerNorm folding, producer side
                    if (ok)
                        PV_STORE16(reinterpret_cast<u32x2*>(p.x16_out + orow * (int64_t)p.N + ncol), ((u32x2){pv_pack_bf16x2_tracked(o[0], o[1], vmax), pv_pack_bf16x2_tracked(o[2], o[3], vmax)}));
                    // this lane's share of the row's (sum, sum of squares); the 64-lane reduction of the 8 rows follows the pass
                    fs[j] = ok ? pv_add_s(pv_add_s(o[0], o[1]), pv_add_s(o[2], o[3])) : 0.f;
                    fq[j] = ok ? pv_add_s(pv_add_s(o[0] * o[0], o[1] * o[1]), pv_add_s(o[2] * o[2], o[3] * o[3])) : 0.f;
                }
            }
            // the rows of pass q + 2 into the slot set just consumed: requested before the NEXT pass's stores, so the wait for them (two
            // passes from now) has only this pass's stores in front of it, long since drained
            if (q + 2 < 4) fetch(q + 2);
            {
                // 8 rows x 64 lanes -> 8 totals in 10 cross-lane steps per quantity (halving the rows a lane carries at every exchange)
                // instead of 8 full wave reductions; lanes 8r .. 8r+7 end up with row r's totals
                const int r_ = (lane >> 3) & 7, mrow = m0 + q * 64 + wid * 8 + r_;
                if (EPI == PV_EPI_BIAS_RES_F32 && p.rowsq_out && !p.x16_out) {
                    const float tq = pv_reduce8_rows(fq, lane);
                    if ((lane & 7) == 0 && mrow < p.M) p.rowsq_out[(int64_t)(n0 / G2_BN) * p.M + mrow] = tq;
                }
                if (EPI == PV_EPI_BIAS_RES_F32 && p.x16_out) {
                    const float ts = pv_reduce8_rows(fs, lane), tq = pv_reduce8_rows(fq, lane);
                    if ((lane & 7) == 0 && mrow < p.M)
                        *reinterpret_cast<float2*>(p.rowstat_out + ((int64_t)(n0 / G2_BN) * p.M + mrow) * 2) = make_float2(ts, tq);
                }
            }
        }
        if (EPI == PV_EPI_BIAS_RES_F32 && p.x16_out) pv_range_commit(vmax, p.range_flag);
        if (EPI == PV_EPI_GELU_GRAD_BF16 && p.colsum_partial) {      // (workgroup-uniform) combine the 8 waves through LDS: [8][256] floats
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                            // the last image has been consumed by every wave
            if constexpr (DG) {
#pragma unroll
                for (int e = 0; e < 8; ++e) cs8[e] = pv_add_s(cs8[e], __shfl_xor(cs8[e], 32, 64));       // the two rows of an instruction: lanes l and l + 32 hold the same columns
                if (lane < 32) {
                    *reinterpret_cast<__attribute__((address_space(3))) f32x4*>(cimg + G2_BUF + wid * 1024 + lane * 32) = (f32x4){cs8[0], cs8[1], cs8[2], cs8[3]};
                    *reinterpret_cast<__attribute__((address_space(3))) f32x4*>(cimg + G2_BUF + wid * 1024 + lane * 32 + 16) = (f32x4){cs8[4], cs8[5], cs8[6], cs8[7]};
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (tid < 256 && n0 + tid < p.N) {
                float t = 0.f;
#pragma unroll
                for (int w = 0; w < 8; ++w) t += *reinterpret_cast<const __attribute__((address_space(3))) float*>(cimg + G2_BUF + w * 1024 + tid * 4);
                p.colsum_partial[(int64_t)(m0 / G2_BM) * p.N + n0 + tid] = t;
            }
        }
    }
    if (PFM == 2 && pf_next) {
        // the next tile's K-tile 0 is older than every store of this epilogue: a COUNTED wait that leaves the stores in flight retires it.
        // A full tile issues at least 16 (16-bit outputs: 2 rows per instruction) / 32 (fp32 and gelu' epilogues: 1 row) stores per wave
        // after the prefetch; a ragged tile may skip some, so it drains.
        const bool full = FULL || (m0 + G2_BM <= p.M && n0 + G2_BN <= p.N);
        // (the gelu' product's fp32-image form issues 16 two-row stores per wave since round 6, like the 16-bit pipelines)
        constexpr bool B16 = EPI == PV_EPI_BIAS_BF16 || EPI == PV_EPI_BIAS_GELU_BF16 || EPI == PV_EPI_BIAS_GELU_PAIR_BF16 || EPI == PV_EPI_GELU_GRAD_BF16;
        if (!full) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (B16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
    }
    PV_STAMP(3);
#ifdef PV_STAMPS
    if (threadIdx.x == 0 && p.dbg) {          // slot 5: which CU ran this workgroup; slot 6/7: constant-rate (100 MHz) wall clock at start / end
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        p.dbg[pv_stamp_slot * 16 + 5] = ((unsigned long long)(xcc & 0xf) << 32) | hw;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    PV_STAMP(4);
    if (threadIdx.x == 0 && p.dbg) { unsigned long long rt_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt_)::"memory"); p.dbg[pv_stamp_slot * 16 + 7] = rt_; }
#endif
}

// (Measured dead end, round 3 - again: the same tile function in a PERSISTENT launch, one workgroup per CU walking its XCD-contiguous share
// of the list behind one barrier per tile, to save the 1.0 - 1.4 us the dispatcher leaves between two workgroups of a CU.  Bit-identical
// and SLOWER on every token GEMM with loads at a tile's start: QKV + fold 1.428 vs 1.357 ms, out-proj 0.904 vs 0.829, fc2 1.911 vs 1.837,
// fc1 + fold 2.056 vs 1.999 (profiles/r03_gemm_epilogue_persist_ab.json, "cur" = persistent).  The counter behind s_waitcnt vmcnt is in
// order: the next tile's first wait for its LDS-DMA also waits for the previous tile's 128 - 384 KiB of stores, which a fresh workgroup
// never has to - the hardware hand-off lets them drain under the successor's prologue.  Removed.)
template <int EPI>
__global__ __launch_bounds__(512) void pv_gemm256_kernel(const GemmDev p_in) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // tile order: every XCD walks a contiguous range of a list in which groups of `gm` M-blocks are swept with n as the slow
    // index, so the ~32 co-running blocks of an XCD touch gm A panels and ~32/gm weight tiles at a time (L2 = 4 MiB per XCD)
    GemmDev p = p_in;
    const int ntiles = p.tiles_m * p.tiles_n;
    const int slice = blockIdx.x / ntiles;                       // split-K slice (0 when ksplit <= 1)
    if (p.ksplit > 1) {
        p.A += (int64_t)slice * p.k_slice; p.W += (int64_t)slice * p.k_slice; p.K = p.k_slice;
        p.out = reinterpret_cast<float*>(p.out) + slice * p.split_stride;
        if (slice) p.bias = nullptr;
    }
    const int tile = pv_xcd_remap(blockIdx.x - slice * ntiles, ntiles);
    // chunk-major list: chunk c = column tiles [c*gc, c*gc + cw); inside a chunk groups of gm row panels, n slow inside a group
    const int nfull = p.tiles_n / p.gc;
    const int ch = min(tile / (p.gc * p.tiles_m), nfull);
    const int r = tile - ch * p.gc * p.tiles_m;
    const int cw = ch < nfull ? p.gc : p.tiles_n - nfull * p.gc;
    const int grp = r / (p.gm * cw), rem = r - grp * (p.gm * cw);
    const int gsz = min(p.gm, p.tiles_m - grp * p.gm);
    const int tnl = rem / gsz, tm = grp * p.gm + (rem - tnl * gsz);
    const int m0 = tm * G2_BM, n0 = (ch * p.gc + tnl) * G2_BN;
    if (m0 + G2_BM <= p.M && n0 + G2_BN <= p.N) pv_gemm256_tile<EPI, false, true>(p, smem, m0, n0);
    else pv_gemm256_tile<EPI, false, false>(p, smem, m0, n0);
}

// Prefetching persistent launch (round 3).  One workgroup per CU walks its XCD-contiguous share of the tile list; the epilogue of tile t
// stages K-tile 0 of tile t+1 (pv_gemm256_tile<EPI, true>), so a tile no longer pays kernel entry (0.7 k cycles), the wait for its first 64
// KiB (2.6 - 3.2 k cycles at batch 2048; 5 - 8 k when few rounds of tiles keep all CUs' prologues in step, vit_small) nor the 1.0 - 1.4 us
// the dispatcher leaves between two workgroups of a CU.  The plain persistent loop (measured above pv_gemm256_kernel) lost because the
// next tile's first wait sat behind the previous tile's stores in the in-order vmcnt queue; here the prefetch is issued BEFORE any store
// of the epilogue (in pieces under its first pass / behind its first residual rows) and retired by a COUNTED wait at its end that leaves
// the stores in flight (PV_PF_MODE 2; mode 1 = all of it at the epilogue's start, retired in front of the first store).  A tile's
// arithmetic is untouched: outputs are bit-identical.  Per-tile stamps of both launches: profiles/r03_gemm_stamps_*.txt; what had to be
// true before it was faster than one tile per workgroup (no VGPR spill, bias through LDS, no K < 256 bypass edge): DESIGN.md section 4.
template <int EPI>
__global__ __launch_bounds__(512) void pv_gemm256_pf_kernel(const GemmDev p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int ntiles = p.tiles_m * p.tiles_n;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, per = gridDim.x >> 3;       // gridDim.x is a multiple of 8
    const int q8 = ntiles >> 3, r8 = ntiles & 7;
    const int start = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int cnt = q8 + (xcd < r8 ? 1 : 0);
    const int nfull = p.tiles_n / p.gc;
    auto decode = [&](int tile, int& m0, int& n0) {       // chunk-major list: chunk c = column tiles [c*gc, c*gc + cw); groups of gm row panels, n slow
        const int ch = min(tile / (p.gc * p.tiles_m), nfull);
        const int r = tile - ch * p.gc * p.tiles_m;
        const int cw = ch < nfull ? p.gc : p.tiles_n - nfull * p.gc;
        const int grp = r / (p.gm * cw), rem = r - grp * (p.gm * cw);
        const int gsz = min(p.gm, p.tiles_m - grp * p.gm);
        const int tnl = rem / gsz, tm = grp * p.gm + (rem - tnl * gsz);
        m0 = tm * G2_BM; n0 = (ch * p.gc + tnl) * G2_BN;
    };
    const int wid = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    int m0 = 0, n0 = 0, nm0 = 0, nn0 = 0;
    if (j < cnt) decode(start + j, m0, n0);
    for (int i = j; i < cnt; i += per) {
        const bool has_next = i + per < cnt;
        if (has_next) decode(start + i + per, nm0, nn0);
        if (m0 + G2_BM <= p.M && n0 + G2_BN <= p.N) pv_gemm256_tile<EPI, true, true>(p, smem, m0, n0, i == j, has_next, nm0, nn0, wid, start + i);
        else pv_gemm256_tile<EPI, true, false>(p, smem, m0, n0, i == j, has_next, nm0, nn0, wid, start + i);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // this wave's LDS reads of the epilogue have returned ...
        __builtin_amdgcn_s_barrier();                          // ... and every wave's: the next tile may overwrite buffer 1; its K-tile 0 is in buffer 0
        __builtin_amdgcn_sched_barrier(0);
        m0 = nm0; n0 = nn0;
    }
}

// ------------------------------------------------------------------------------------------------
// TN variant for weight gradients: out[m][n] = sum_k A[k][m] * B[k][n] straight from the row-major activations (A = dY
// [K = token rows, M = out features], B = X [K, N = in features]) - no transposed copies.  Same 256 x 256 x 64 tile, 4-phase
// schedule, counted vmcnt and wave-group stagger as pv_gemm256_tile; what changes is the LDS image and the fragment reads:
//   half-tile slot = 64 k-rows x 128 columns (256-byte rows), staged by LDS-DMA in 1-KiB pieces of 4 k-rows (16 lanes read
//   256 contiguous bytes of one k-row);  16-byte chunk c of k-row r lives at chunk c ^ s(r), s(r) = ((r&3)<<2)|(((r>>2)&1)<<1);
//   a fragment (16 columns x 32 k) = two ds_read_b64_tr_b16 (k-rows kb+4g+q and kb+16+4g+q, q = 0..3): the 32 lanes of a
//   half-wave then cover 8 k-rows x 32 B on all 64 banks exactly once.  Both operands use the same k order inside a fragment,
//   so the products are unchanged.  Slots are ordered {A0: buf0, buf1 | A1 | B0 | B1} so that buffer and k offsets are
//   ds_read immediates on 8 + 4 lane-constant bases.
// Output: fp32 split-K slices out[slice][M][ldo] (slice = K range), reduced by pv_sum_slices_f32.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void pv_gemm256_tn_kernel(const GemmDev p_in) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    GemmDev p = p_in;
    const int ntiles = p.tiles_m * p.tiles_n;
    // (slice, tile) list in slice-major order; every XCD (blocks b, b + 8, ...) walks a CONTIGUOUS range of it, so the ~32 workgroups an XCD
    // runs together are (nearly) one K slice: each dY column block they stream is shared by tiles_n of them and each X column block by
    // tiles_m of them in that XCD's L2.  (r1 remapped inside a slice with the slice's local block index as "XCD", which is the real
    // XCD only for slices that start at a multiple of 8 blocks.)   PV_TN_RASTER=0 restores it for A/B.
    int slice, tile;
    if (p.gm != 0) {
        const int l = pv_xcd_remap(blockIdx.x, ntiles * p.ksplit);
        slice = l / ntiles; tile = l - slice * ntiles;
    } else {
        slice = blockIdx.x / ntiles; tile = pv_xcd_remap(blockIdx.x - slice * ntiles, ntiles);
    }
    p.A += (int64_t)slice * p.k_slice * p.lda; p.W += (int64_t)slice * p.k_slice * p.ldw;
    p.K = slice == p.ksplit - 1 ? p.k_last : p.k_slice;          // the last slice takes the remainder
    p.out = reinterpret_cast<float*>(p.out) + slice * p.split_stride;
    const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
    const int m0 = tm * G2_BM, n0 = tn * G2_BN;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 2, wc = wid & 3;
    constexpr int SLOT = 2 * G2_HALF;                 // one operand half: both K-tile buffers, 32 KiB

    // ---- LDS-DMA sources -------------------------------------------------------------------------------------------
    const int sw_s = ((lane >> 4) << 2) | ((wid & 1) << 1);          // s(r) of this lane's k-row (r&3 = lane>>4, (r>>2)&1 = wid&1)
    const int scol = ((lane & 15) ^ sw_s) * 8;                       // logical column (elements) inside the 128-column half
    const char* const a_blk = reinterpret_cast<const char*>(p.A + m0);
    const char* const w_blk = reinterpret_cast<const char*>(p.W + n0);
    // one per-lane byte offset per operand (piece j = 0, half 0); piece 1 = 32 k-rows further, half 1 = 256 bytes further: both
    // are workgroup-uniform additions to the scalar base.  M, N are multiples of 128 (host check), so a half-tile is entirely
    // inside or entirely outside the matrix: an outside half re-reads half 0 (its results are never stored).
    const uint32_t oa = (uint32_t)((4 * wid + (lane >> 4)) * (int)p.lda + scol) * 2u;
    const uint32_t ow = (uint32_t)((4 * wid + (lane >> 4)) * (int)p.ldw + scol) * 2u;
    const int a_h1 = m0 + 256 <= p.M ? 256 : 0, w_h1 = n0 + 256 <= p.N ? 256 : 0;
    char* const lds_piece = smem + wid * 1024;
    const int64_t a_step = (int64_t)G2_BK * p.lda * 2, w_step = (int64_t)G2_BK * p.ldw * 2;
    auto stage_a = [&](int buf, int h, int kt) __attribute__((always_inline)) {
        const char* src = a_blk + kt * a_step + (h ? a_h1 : 0);
        pv_glds16(src + oa, lds_piece + h * SLOT + buf * G2_HALF);
        pv_glds16(src + (a_step >> 1) + oa, lds_piece + h * SLOT + buf * G2_HALF + 8192);
    };
    auto stage_b = [&](int buf, int h, int kt) __attribute__((always_inline)) {
        const char* src = w_blk + kt * w_step + (h ? w_h1 : 0);
        pv_glds16(src + ow, lds_piece + (2 + h) * SLOT + buf * G2_HALF);
        pv_glds16(src + (w_step >> 1) + ow, lds_piece + (2 + h) * SLOT + buf * G2_HALF + 8192);
    };

    // ---- transposed fragment read bases ------------------------------------------------------------------------------
    typedef __attribute__((address_space(3))) const char lds_cc;
    const int g = lane >> 4, i16 = lane & 15, tq = i16 >> 2, tp = i16 & 3;
    const int rl = 4 * g + tq;
    const int sw_l = (tq << 2) | ((g & 1) << 1);
    lds_cc* const lds0 = (lds_cc*)smem;
    lds_cc* a_rd[8];
    lds_cc* b_rd[4];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        a_rd[t] = lds0 + wr * SLOT + rl * 256 + (((2 * t + (tp >> 1)) ^ sw_l) << 4) + ((tp & 1) << 3);
        asm volatile("" : "+v"(a_rd[t]));
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        b_rd[t] = lds0 + (2 + (wc >> 1)) * SLOT + rl * 256 + (((2 * ((wc & 1) * 4 + t) + (tp >> 1)) ^ sw_l) << 4) + ((tp & 1) << 3);
        asm volatile("" : "+v"(b_rd[t]));
    }

    f32x4 acc[4][8];   // [nt][mt]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 af[2][4][2];
    bf16x8 bfr[2][2];
    typedef __attribute__((address_space(3))) s16x4 lds_s4;

    // transposed reads as inline asm: the compiler's wait-count pass puts s_waitcnt vmcnt(0) in front of every ds_read_tr builtin
    // while LDS-DMA is in flight (it cannot see that the pieces being staged are not the ones being read); the asm results are
    // consumed only after the explicit s_waitcnt lgkmcnt(0) + barrier of TN_SYNC_LOADS.
#define TN_FRAG(DST, BASE, OFF)                                                                                              \
    do {                                                                                                                     \
        s16x4 lo_, hi_;                                                                                                      \
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo_) : "v"(BASE), "n"(OFF));                               \
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi_) : "v"(BASE), "n"((OFF) + 4096));                      \
        DST = __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo_, hi_, 0, 1, 2, 3, 4, 5, 6, 7));                         \
    } while (0)
#define TN_READ_A(BUF, MH)                                                                                  \
    _Pragma("unroll") for (int t_ = 0; t_ < 4; ++t_) _Pragma("unroll") for (int ks_ = 0; ks_ < 2; ++ks_)  \
        TN_FRAG(af[MH][t_][ks_], a_rd[(MH) * 4 + t_], (BUF) * G2_HALF + ks_ * 8192);
#define TN_READ_B(BUF, NH)                                                                                  \
    _Pragma("unroll") for (int t_ = 0; t_ < 2; ++t_) _Pragma("unroll") for (int ks_ = 0; ks_ < 2; ++ks_)  \
        TN_FRAG(bfr[t_][ks_], b_rd[(NH) * 2 + t_], (BUF) * G2_HALF + ks_ * 8192);
#define TN_SYNC_LOADS()                                  \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   \
    __builtin_amdgcn_s_barrier();                        \
    __builtin_amdgcn_sched_barrier(0);
#define TN_MFMA(MH, NH)                                                                                              \
    __builtin_amdgcn_s_setprio(1);                                                                                   \
    _Pragma("unroll") for (int ks_ = 0; ks_ < 2; ++ks_) _Pragma("unroll") for (int n_ = 0; n_ < 2; ++n_)            \
        _Pragma("unroll") for (int m_ = 0; m_ < 4; ++m_)                                                            \
            acc[(NH) * 2 + n_][(MH) * 4 + m_] = PV_MFMA_16x16x32(                            \
                bfr[n_][ks_], af[MH][m_][ks_], acc[(NH) * 2 + n_][(MH) * 4 + m_], 0, 0, 0);                     \
    __builtin_amdgcn_s_setprio(0);                                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                               \
    __builtin_amdgcn_s_barrier();                                                                                    \
    __builtin_amdgcn_sched_barrier(0);

    auto ktile = [&](auto buf_c, auto sb_c, auto sa_c, int kt) __attribute__((always_inline)) {
        constexpr int BUF = decltype(buf_c)::value;
        constexpr bool SB = decltype(sb_c)::value, SA = decltype(sa_c)::value;
        TN_READ_B(BUF, 0)
        __builtin_amdgcn_sched_barrier(0);
        TN_READ_A(BUF, 0)
        if (SB) stage_b(BUF ^ 1, 0, kt + 1);
        TN_SYNC_LOADS()
        TN_MFMA(0, 0)
        TN_READ_A(BUF, 1)
        if (SB) stage_b(BUF ^ 1, 1, kt + 1);
        TN_SYNC_LOADS()
        TN_MFMA(1, 0)
        TN_READ_B(BUF, 1)
        if (SA) stage_a(BUF, 0, kt + 2);
        TN_SYNC_LOADS()
        TN_MFMA(1, 1)
        if (SA) {
            stage_a(BUF, 1, kt + 2);
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        TN_SYNC_LOADS()
        TN_MFMA(0, 1)
    };
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;
    using T = std::true_type;
    using F = std::false_type;

    const int nk = p.K / G2_BK;        // even, >= 2 (checked on the host)
    stage_a(0, 0, 0); stage_a(0, 1, 0); stage_b(0, 0, 0); stage_b(0, 1, 0);
    stage_a(1, 0, 1); stage_a(1, 1, 1);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (wr == 1) __builtin_amdgcn_s_barrier();
    int kt = 0;
    for (; kt + 4 <= nk; kt += 2) {
        ktile(B0{}, T{}, T{}, kt);
        ktile(B1{}, T{}, T{}, kt + 1);
    }
    ktile(B0{}, T{}, F{}, kt);
    ktile(B1{}, F{}, F{}, kt + 1);
    if (wr == 0) __builtin_amdgcn_s_barrier();
#undef TN_FRAG
#undef TN_READ_A
#undef TN_READ_B
#undef TN_SYNC_LOADS
#undef TN_MFMA

    // ---- epilogue: fp32 image (128 rows x 1 KiB per pass) so that every global store is a whole 1-KiB row segment ---------
    typedef __attribute__((address_space(3))) char lds_c;
    lds_c* const cimg = (lds_c*)smem;
    const bool col_ok = n0 + lane * 4 < p.N;
    const int ncol = col_ok ? n0 + lane * 4 : 0;
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
        if (ps == 1) __builtin_amdgcn_s_barrier();
        if (wr == ps) {
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) {
                const int row = mt * 16 + i16;
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const int c = wc * 16 + nt * 4 + g;             // lane holds n = wc*64 + nt*16 + 4g + 0..3 of row m = mt*16 + i16
                    *reinterpret_cast<__attribute__((address_space(3))) f32x4*>(cimg + row * 1024 + ((c ^ (i16 & 7)) << 4)) = acc[nt][mt];
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int row = wid * 16 + j;
            const f32x4 v = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(cimg + row * 1024 + ((lane ^ (row & 7)) << 4));
            const int m = m0 + ps * 128 + row;
            if (m < p.M && col_ok) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + (int64_t)m * p.ldo + ncol) = v;
        }
    }
}

// out fp32 [ksplit][M][ldo] partial products of A^T . B over K slices; A bf16 [K, M] (row stride lda), B = args->W bf16 [K, N]
// (row stride ldw).  M, N multiples of 128, K a multiple of 128 with at least ksplit blocks of 128 rows.
extern "C" int pv_gemm_tn_bf16(const pv_gemm_args* a, void* stream) {
    if (!a || a->struct_size != sizeof(pv_gemm_args)) return PV_ERR_INVALID_ARG;      // nothing past the first field is read before this
    if (!a->A || !a->W || !a->out || a->M < 8 || a->N < 8 || a->K <= 0) return PV_ERR_INVALID_ARG;
    const int ks = a->ksplit > 1 ? a->ksplit : 1;
    if (a->M % 128 || a->N % 128 || a->K % (2 * G2_BK) || a->K / (2 * G2_BK) < ks) return PV_ERR_UNSUPPORTED;
    if (a->lda % 8 || a->ldw % 8 || a->ldo % 4 || a->lda < a->M || a->ldw < a->N || a->ldo < a->N) return PV_ERR_INVALID_ARG;
    if (((uintptr_t)a->A & 15) || ((uintptr_t)a->W & 15) || ((uintptr_t)a->out & 15)) return PV_ERR_INVALID_ARG;
    if (a->epilogue != PV_EPI_BIAS_F32 || a->bias || a->qcols) return PV_ERR_INVALID_ARG;
    if (a->M > 0x7fffffff || a->N > 0x7fffffff || a->K > 0x7fffffff || 64 * a->lda * 2 > 0x7fffffff || 64 * a->ldw * 2 > 0x7fffffff) return PV_ERR_UNSUPPORTED;
    GemmDev p = {};
    p.A = a->A; p.W = a->W; p.out = a->out;
    p.M = (int)a->M; p.N = (int)a->N; p.K = (int)a->K;
    p.lda = a->lda; p.ldw = a->ldw; p.ldo = a->ldo;
    // K / 128 blocks of rows are dealt to the slices as evenly as whole blocks allow; the last slice takes the remainder
    p.ksplit = ks; p.k_slice = (int)(a->K / (2 * G2_BK) / ks) * (2 * G2_BK); p.k_last = (int)a->K - p.k_slice * (ks - 1);
    p.split_stride = a->M * a->ldo;
    static const int tn_raster = [] { const char* e = getenv("PV_TN_RASTER"); return e ? atoi(e) : 1; }();
    p.tiles_m = (p.M + G2_BM - 1) / G2_BM; p.tiles_n = (p.N + G2_BN - 1) / G2_BN; p.gm = tn_raster; p.gc = p.tiles_n;
    if ((int64_t)p.tiles_m * p.tiles_n * ks > 0x7fffffff) return PV_ERR_UNSUPPORTED;
    static PvPerDevice attr_set;
    if (attr_set.first_use()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pv_gemm256_tn_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, G2_LDS);
    }
    PV_LAUNCH(pv_gemm256_tn_kernel, dim3((unsigned)(p.tiles_m * p.tiles_n * ks)), dim3(512), G2_LDS, (hipStream_t)stream, p);
    return pv_check_launch();
}

// ------------------------------------------------------------------------------------------------
// Row-block kernel with FUSED LayerNorm: one workgroup computes ALL column tiles of its 256 output rows (N = hidden dim),
// then normalises the rows it has just written - re-read from the XCD's L2 with L1-bypassing loads instead of from HBM by a
// separate LayerNorm launch - and emits the bf16 operand of the next GEMM.  The per-row arithmetic is the standalone LN
// kernel's (pv_ln_row), so the result is bit-identical to GEMM + pv_layernorm_bf16.
//   replaces models/vit.py:51 + :53 (out-proj residual, then ln_2) and models/vit.py:55 + next block's :48 (fc2 residual, ln_1)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void pv_load16_l2(f32x4& dst, const void* ptr) {     // sc1: served by L2, never by a (possibly stale) L1 line
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(dst) : "v"(ptr) : "memory");
}

template <int NCH>
__device__ __forceinline__ void pv_fused_ln_rows(const GemmDev& p, const int m0) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int D = p.N, nvec = D >> 2;
    const float* const xo = reinterpret_cast<const float*>(p.out);
    constexpr int RB = NCH <= 4 ? 4 : 1;                     // rows in flight per wave (registers: RB * NCH * 4)
    for (int r0 = 0; r0 < 32; r0 += RB) {
        f32x4 raw[RB][NCH];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            int m = m0 + wid * 32 + r0 + rb;
            m = m < p.M ? m : p.M - 1;
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                const int idx = lane + 64 * j;
                if (idx < nvec) pv_load16_l2(raw[rb][j], xo + (int64_t)m * p.ldo + idx * 4);
                else raw[rb][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
        }
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int j = 0; j < NCH; ++j) asm volatile("s_waitcnt vmcnt(0)" : "+v"(raw[rb][j])::"memory");
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            const int m = m0 + wid * 32 + r0 + rb;
            if (m >= p.M) break;
            RowRegs<NCH> r;
#pragma unroll
            for (int j = 0; j < NCH; ++j) r.v[j] = make_float4(raw[rb][j][0], raw[rb][j][1], raw[rb][j][2], raw[rb][j][3]);
            pv_ln_row<NCH>(r, p.ln_gamma, p.ln_beta, D, nvec, lane, p.ln_eps);
            const float sc = p.ln_row_scale ? p.ln_row_scale[m] : 1.0f;
            u32x2* o = reinterpret_cast<u32x2*>(p.ln_out + (int64_t)m * D);
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                const int idx = lane + 64 * j;
                if (idx < nvec) {
                    u32x2 pk = {pv_pack_bf16x2(r.v[j].x * sc, r.v[j].y * sc), pv_pack_bf16x2(r.v[j].z * sc, r.v[j].w * sc)};
                    o[idx] = pk;
                }
            }
        }
    }
}

template <int EPI>
__global__ __launch_bounds__(512) void pv_gemm256_rows_kernel(const GemmDev p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int m0 = pv_xcd_remap(blockIdx.x, p.tiles_m) * G2_BM;
    for (int tn = 0; tn < p.tiles_n; ++tn) {
        pv_gemm256_tile<EPI>(p, smem, m0, tn * G2_BN);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the epilogue's LDS reads are done before the next tile's DMA
        __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // every row segment of this block has reached L2
    __builtin_amdgcn_s_barrier();
    const int nch = (p.N / 4 + 63) / 64;
    if (nch <= 1) pv_fused_ln_rows<1>(p, m0);
    else if (nch == 2) pv_fused_ln_rows<2>(p, m0);
    else if (nch == 3) pv_fused_ln_rows<3>(p, m0);
    else if (nch == 4) pv_fused_ln_rows<4>(p, m0);
    else if (nch <= 8) pv_fused_ln_rows<8>(p, m0);
    else pv_fused_ln_rows<16>(p, m0);
}

// prefetching persistent 256^2 launches: PV_GEMM_PF=0 or pv_debug_set_gemm_pf(0) restores one tile per workgroup (A/B, scripts/gemm_epi_ab.py)
#ifndef PV_GEMM_PF_DEFAULT
#define PV_GEMM_PF_DEFAULT 1
#endif
static int g_pv_pf = -1;
extern "C" void pv_debug_set_gemm_pf(int on) { g_pv_pf = on; }
static bool pv_gemm_pf_enabled() {
    static const int env = [] { const char* e = getenv("PV_GEMM_PF"); return e ? atoi(e) : PV_GEMM_PF_DEFAULT; }();
    return g_pv_pf >= 0 ? g_pv_pf != 0 : env != 0;
}
static int pv_cu_count() {              // per device (the current one = the stream's: peekvit_amd.ops refuses otherwise)
    static int cached[64] = {};
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) return 0;
    if (cached[d] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess) n = 0;
        cached[d] = n > 0 ? n : -1;
    }
    return cached[d] > 0 ? cached[d] : 0;
}

template <int EPI>
static int pv_launch_gemm256(const GemmDev& p, hipStream_t stream) {
    static PvPerDevice attr_set;
    constexpr int lds = G2_LDS + ((EPI == PV_EPI_BIAS_GELU_BF16 || EPI == PV_EPI_BIAS_GELU_SPLIT_BF16 || EPI == PV_EPI_BIAS_GELU_PAIR_BF16)
                                        ? PV_GELU_CUB_N * PV_GELU_CUB_REP * 16 : 0)                   // + 16 KiB replicated cubic GELU table
                        + ((EPI == PV_EPI_BIAS_BF16 || EPI == PV_EPI_BIAS_GELU_BF16) ? 4096 : 0);         // + 4 KiB folded-LayerNorm constants
    // (persistent launch: + 1 KiB of bias values behind the table and the fold slot; the gelu' epilogue has no bias)
    constexpr int lds_pf = EPI == PV_EPI_GELU_GRAD_BF16 ? lds
                         : G2_LDS + ((EPI == PV_EPI_BIAS_GELU_BF16 || EPI == PV_EPI_BIAS_GELU_SPLIT_BF16 || EPI == PV_EPI_BIAS_GELU_PAIR_BF16) ? PV_GELU_CUB_N * PV_GELU_CUB_REP * 16 : 0) + 4096 + 1024;
    // the prefetching persistent launch: every epilogue but the one-pass forms (bf16x3's SPLIT; -DPV_EPI_PIPE=0 builds) keeps buffer 0 free
    constexpr bool PF_OK = EPI != PV_EPI_BIAS_GELU_SPLIT_BF16 && (PV_EPI_PIPE || (EPI != PV_EPI_BIAS_BF16 && EPI != PV_EPI_BIAS_GELU_BF16 && EPI != PV_EPI_BIAS_GELU_PAIR_BF16));
    if (attr_set.first_use()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pv_gemm256_kernel<EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (PF_OK) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pv_gemm256_pf_kernel<EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_pf);
    }
    if (PF_OK) {
        const int cus = pv_cu_count();
        if (pv_gemm_pf_enabled() && p.ksplit <= 1 && p.K >= 4 * G2_BK && (int64_t)p.tiles_m * p.tiles_n >= 2 * (int64_t)cus && cus >= 8 &&
            !(EPI == PV_EPI_GELU_GRAD_BF16 && p.bias)) {
            PV_LAUNCH(pv_gemm256_pf_kernel<EPI>, dim3((unsigned)(cus & ~7)), dim3(512), lds_pf, stream, p);
            return pv_check_launch();
        }
    }
    PV_LAUNCH(pv_gemm256_kernel<EPI>, dim3((unsigned)(p.tiles_m * p.tiles_n * (p.ksplit > 1 ? p.ksplit : 1))), dim3(512), lds, stream, p);
    return pv_check_launch();
}

template <int EPI>
static int pv_launch_gemm256_rows(const GemmDev& p, hipStream_t stream) {
    static PvPerDevice attr_set;
    if (attr_set.first_use()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pv_gemm256_rows_kernel<EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, G2_LDS);
    }
    PV_LAUNCH(pv_gemm256_rows_kernel<EPI>, dim3((unsigned)p.tiles_m), dim3(512), G2_LDS, stream, p);
    return pv_check_launch();
}

// ------------------------------------------------------------------------------------------------
// Full-row tile for NARROW hidden dims: 128 rows x N (N = 256, 384 or 512 = the whole output row), 8 waves as 2 (M) x 4 (N),
// 64 x N/4 outputs per wave, A 128 x 64 + W N x 64 per K tile in two LDS buffers (96 / 128 / 160 KiB), one workgroup per CU.
// The residual GEMMs of vit_small / vit_tiny widths (out-proj, fc2: PV_EPI_BIAS_RES_F32) are memory- and epilogue-bound rather than
// MFMA-bound (rocprof: MFMA busy 23 %; in-kernel clock 1.2 - 2.1 GHz back to back, 2.1 GHz `sclk` at 1.2 kW in the model: close to, not
// at, the power cap - profiles/r04_fullrow_stamps_dp.txt); with the whole row in one workgroup
//   - N = 384 wastes no quarter-empty 256-wide column tile,
//   - the LayerNorm that consumes the row next (ln_2 after out-proj, the next block's ln_1 after fc2) runs on the finished fp32
//     values while they are still in the workgroup's LDS: its launch and its re-read of the residual stream disappear
//     (models/vit.py:51+53, :55 + the next block's :48).  The per-row arithmetic is pv_ln_row (the standalone kernel's), the
//     residual add is the 256^2 epilogue's fmaf, K is accumulated in the same order: outputs are bit-identical to
//     pv_gemm_bf16 followed by pv_layernorm_bf16 (tests/test_hip_ops.py).
// K loop (DP = false, K not a multiple of 128): 2 buffers, LDS-DMA of tile kt+1 in flight under the MFMAs of tile kt, one barrier per
// K tile (the 128^2 kernel's structure).  DP = true (round 4, the default): the deep-pipelined loop described where it starts - 1.8 k
// instead of 3.0 k ticks per K-tile.
// (Measured dead end, round 4: the same loop as a ring of FOUR 32-deep stages with three stages of LDS-DMA in flight behind a counted
//  vmcnt - to take the DMA latency per K tile off the critical path (3.3 k cycles per 64 of K against the 2.3 k the global -> LDS path
//  needs, scripts/fullrow_probe.py) - is bit-identical and 6 - 10 % SLOWER (N = 384: K = 384 97.8 -> 103.8 us, K = 1536 203.7 -> 223.5 us):
//  a 32-deep stage has 64-byte rows, so every 1-KiB DMA piece touches 16 half cache lines instead of 8 whole ones, and the staging path
//  is what bounds this loop.  A ring of 64-deep stages would need 192 KiB at N = 384.)
// (Measured dead ends, round 4, profiles/r04_fullrow_*: de-phasing the CUs - half of them starting on a 64-row tile, or half a tile late - so that
//  one half's epilogues run under the other half's K loops: no gain, the delay only adds; a persistent launch with the next tile's first
//  two K-tiles staged from inside the epilogue and the parameters in LDS: bit-identical, prologue 4.1 -> 2.4 k ticks, vit_small 2 % SLOWER
//  in the model (101.1 -> 99.0 k img/s) - a tile's wall time stayed put while its tick count moved with the clock.)
// Epilogue: two passes of 64 rows; the wave group that owns the rows writes bias-initialised accumulators to an fp32 LDS image
// (16-byte chunk c of row r at chunk c ^ (r & 7)), then every wave takes 8 whole rows, four at a time with sixteen lanes per row
// (round 4): residual row from HBM (requested ahead), fmaf, fp32 row store, LayerNorm (pv_ln_row16), 16-bit row store.
// ------------------------------------------------------------------------------------------------
template <int N_>
__device__ __forceinline__ void pv_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory"); }

// Immediates of the counted waits of the deep-pipelined full-row K loop, from a compile-time model of its LDS-DMA issue order (the same model
// as scripts/fullrow_vmcnt_model.py).  NPH phases per K-tile, PW pieces per wave and W group, 2 pieces of A.  Per K-tile s a wave issues
//   phase 0: W group NPH-1 of K-tile s+1      phase 1: A and W group 0 of K-tile s+2      phase j >= 2: W group j-1 of K-tile s+2
// (prologue: all of K-tile 0, then K-tile 1 but its last group).  The wait at the end of phase j's load interval must leave in flight
// exactly the pieces issued AFTER the ones phase j+1 (or phase 0 of the next K-tile) reads.  v[mode][j]: mode 0 steady state, 1 K-tile
// nk-2, 2 K-tile nk-1 (its last phase waits for nothing).
struct PvFrCounts { int pro; int v[3][4]; };
constexpr PvFrCounts pv_fr_counts(int NPH, int PW, int AP = 2) {      // AP: A pieces per wave and K-tile (2; 3 for the 160-row tile)
    constexpr int NK = 8;                                // long enough for a steady state between prologue and tail
    int kind[512] = {}, tile[512] = {};                  // kind 0 = A, 1 + g = W group g
    int n = 0;
    auto issue = [&](int k, int t, int cnt) { for (int i = 0; i < cnt; ++i) { kind[n] = k; tile[n] = t; ++n; } };
    auto younger = [&](int k, int t) { int last = -1; for (int i = 0; i < n; ++i) if (kind[i] == k && tile[i] == t) last = i; return n - 1 - last; };
    PvFrCounts c = {};
    issue(0, 0, AP);
    for (int g = 0; g < NPH; ++g) issue(1 + g, 0, PW);
    issue(0, 1, AP);
    for (int g = 0; g < NPH - 1; ++g) issue(1 + g, 1, PW);
    c.pro = younger(1, 0);                               // A(0) is older than W group 0 of K-tile 0
    for (int s = 0; s < NK; ++s)
        for (int j = 0; j < NPH; ++j) {
            if (j == 0) { if (s + 1 < NK) issue(NPH, s + 1, PW); }
            else if (s + 2 < NK) { if (j == 1) { issue(0, s + 2, AP); issue(1, s + 2, PW); } else issue(j, s + 2, PW); }
            int w = 0;
            if (j < NPH - 1) w = younger(2 + j, s);
            else if (s + 1 < NK) w = younger(1, s + 1);
            const int mode = s == NK - 1 ? 2 : s == NK - 2 ? 1 : 0;
            if (mode != 0 || s == 3) c.v[mode][j] = w;
        }
    return c;
}

// Round 6: the body is a device function of MT = row tiles (16 rows) per wave group, so that one launch can mix 128-row tiles (MT = 4) with a LAST ROUND
// of 160-row tiles (MT = 5, deep-pipelined loop at N <= 384 only).  788 tiles of 128 rows on 256 CUs are 3.08 rounds; round 3's split remainder ran
// the 0.08 as a fourth round of 64-row tiles whose K loop is as long as a full tile's (the loop is paced by its staging, not by its MFMAs): 21 % of fc2's
// time on 16 % of the CUs.  Two rounds of 128-row tiles + one round of 221 tiles of 160 rows cover the same rows in 3.25 tile times: 5 + 5 row tiles on the two
// waves of a SIMD = 1.25 x the MFMAs per K-tile, the same W staging, 24 instead of 16 KiB of A (waves 4-7 repeat their first piece into a spare
// slot so that every wave issues the same number of pieces: the counted waits stay one table), three epilogue passes (64 + 64 + 32 rows), one batch
// of residual rows in registers instead of two (120 accumulator registers).  Per row the arithmetic is unchanged: bit-identical.
template <int NT, int DPH, int MT>      // N = 64 * NT, NT in {4, 6, 8}: n-tiles (16 columns) per wave; DPH = phases per K-tile of the deep-pipelined K loop (round 4), 0 = the plain loop
__device__ __forceinline__ void pv_fullrow_body(const GemmDev& p, char* const smem, const int m0, const bool half) {
    constexpr bool DP = DPH > 0;
    static_assert(MT == 4 || (MT == 5 && DP && NT <= 6), "160-row tiles: deep-pipelined loop, N <= 384");
    constexpr int N = 64 * NT, BM = 32 * MT, BK = 64;
    constexpr int AP = (BM + 63) / 64;                    // A pieces (8 rows x 128 B) per wave and K-tile
    constexpr int A_BYTES = AP * 64 * BK * 2;             // 16 KiB (24 KiB: 160 rows + 32 spare)
    constexpr int BUF = A_BYTES + N * BK * 2;             // one K-tile buffer
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 2, wn = wid & 3;
#ifdef PV_STAMPS
    const size_t pv_stamp_slot = (size_t)blockIdx.x;
    if (threadIdx.x == 0 && p.dbg) { unsigned long long rt_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt_)::"memory"); p.dbg[pv_stamp_slot * 16 + 5] = rt_; p.dbg[pv_stamp_slot * 16 + 7] = (unsigned long long)(half ? 1 : MT == 5 ? 2 : 0); }
#endif
    PV_STAMP(0);
    const bool skip_mm = half && wm == 1;              // (wave-uniform)
    const int g = lane >> 4, i16 = lane & 15;

    // accumulators start from the bias.  Ordinary loads, consumed before any LDS-DMA is issued - or (round 4, deep-pipelined loop at N <= 384,
    // where 4 N bytes of LDS are left behind the two K-tile buffers) an LDS-DMA of the N floats issued FIRST, so that it is older than every
    // piece and the prologue's counted wait covers it: the bias loads' latency (~2 k ticks per tile) no longer sits in front of the first piece,
    // and no register load sits in the queue for the compiler to drain with vmcnt(0).
    constexpr bool BIAS_LDS = DP && NT <= 6;
    constexpr int BIAS_OFF = 2 * BUF;
    f32x4 acc[NT][MT];  // [nt][mt]: out[row wm*16*MT + mt*16 + i16][col wn*16*NT + nt*16 + 4g + 0..3]
    if constexpr (BIAS_LDS) {
        if (p.bias && wid < NT)                                // (wave-uniform) 64 floats per wave
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(p.bias + wid * 64 + lane),
                                             (__attribute__((address_space(3))) void*)(smem + BIAS_OFF + wid * 256), 4, 0, 0);
    } else {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
            if (p.bias) b4 = *reinterpret_cast<const f32x4*>(p.bias + wn * 16 * NT + nt * 16 + 4 * g);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = b4;
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) asm volatile("" : "+v"(acc[nt][mt]));
        __builtin_amdgcn_sched_barrier(0);
    }

    // Epilogue row mapping (round 4): a pass is 64 rows; wave `wid` owns rows wid * 8 .. + 7 of it in two batches of four, SIXTEEN lanes per row
    // (row = wid * 8 + 4 b + (lane >> 4); lane l16 holds the 16-byte chunks l16 + 16 k, k < KC = N / 64): every lane busy at N = 384 (the
    // wave-per-row form of rounds 2-3 left a quarter of the lanes without a second chunk), one wave instruction = four 256-byte row segments,
    // LayerNorm reductions by DPP alone (pv_ln_row16, same summation order as the standalone kernel).
    // The residual rows run ahead of their use: batch 0 of pass 0 is issued before the LAST K-tile of the deep-pipelined loop (its counted
    // waits leave the loads in flight), batch 1 at the start of the epilogue, pass 1's batches as pass 0 consumes the registers.
    constexpr int KC = NT;
    const int q4 = lane >> 4, l16 = lane & 15;
    constexpr int RB = (NT <= 6) ? 2 : 1;                  // batches of residual rows in registers (N = 512: 128 accumulator registers leave room for one; the 160-row tile at N = 384 fits two with no register to spare)
    constexpr int NPASS = (BM + 63) / 64;                  // epilogue passes of 64 rows (a wave owns 8: two batches of four); the 160-row tile: 64 + 64 + 32 (4 rows: one batch)
    f32x4 rr[RB][KC];
    float sc[RB], lsc[RB];
    auto res_load = [&](int ps, int b) __attribute__((always_inline)) {
        const int m = m0 + ps * 64 + ((MT == 5 && ps == 2) ? wid * 4 + q4 : wid * 8 + 4 * b + q4), mr = m < p.M ? m : p.M - 1;
        sc[b % RB] = p.row_scale ? p.row_scale[mr] : 1.0f;        // (workgroup-uniform branches)
        lsc[b % RB] = p.ln_row_scale ? p.ln_row_scale[mr] : 1.0f;
        const float* src = p.res + (int64_t)mr * p.ldr + l16 * 4;
#pragma unroll
        for (int k = 0; k < KC; ++k) rr[b % RB][k] = *reinterpret_cast<const f32x4*>(src + 64 * k);
    };
    // LayerNorm affine parameters: one float of each per thread now, written to LDS behind the first image (gamma[N] | beta[N])
    float ln_g1 = 0.f, ln_b1 = 0.f;
    if (p.ln_out && tid < N) { ln_g1 = p.ln_gamma[tid]; ln_b1 = p.ln_beta[tid]; }
    constexpr bool RES_AHEAD = DP && RB == 2;

  if constexpr (DP) {
    // ---- deep-pipelined K loop (round 4) ----------------------------------------------------------------------------------------------
    // The 256^2 kernel's structure on the 128 x N tile: LDS-DMA stays in flight across raw s_barriers behind COUNTED s_waitcnt vmcnt, the
    // two wave groups (wm = 0 / 1 = the two waves of every SIMD) run staggered by one barrier.  A K-tile is NP phases (shipped: 2) of 8 NTP
    // MFMAs, NTP = NT / NP: phase j multiplies all four row tiles of the wave by its column tiles NTP j .. NTP j + NTP - 1.  "W group j" = the
    // 64 NTP weight rows the four column groups read in phase j (rows wn * 16 NT + 16 NTP j + ..) = NTP 1-KiB pieces per wave; A = two pieces.
    // Every piece is staged TWO K-tiles ahead into the slot its predecessor was last read from, one phase after that read:
    //   phase 0 of K-tile s: reads A(s), Wg0(s)      stages Wg(NP-1)(s+1) -> buffer (s+1) & 1     (last read: phase NP-1 of K-tile s-1)
    //   phase 1            : reads Wg1(s)            stages A(s+2), Wg0(s+2) -> buffer s & 1
    //   phase j >= 2       : reads Wgj(s)            stages Wg(j-1)(s+2)
    // so a piece has more than a K-tile (2.5 - 3 k cycles) to arrive, and NT + 2 pieces per wave are issued per K-tile in a fixed order.
    // The wait that covers what phase j+1 reads sits at the end of phase j's load interval (one barrier pair before the first read, so it
    // holds for every wave of both groups); its immediate = the pieces issued after the needed ones, from the compile-time model of the
    // issue order above (pv_fr_counts; N = 384, two phases: steady state {8, 8}, K-tile nk-2 {8, 3}, K-tile nk-1 {0}).
    // A 64-row tile still stages both A pieces (the counts assume a fixed number of operations per K-tile) and its group 1 multiplies the
    // clamped rows like any others (never stored: a branch around the MFMAs made hipcc spill the fragments).
    constexpr int NP = DP ? DPH : 1;                     // phases per K-tile
    constexpr int NTP = NT / NP;                         // column tiles (16 columns) of a wave per phase = pieces per wave and W group
    static_assert(NT % NP == 0 && NTP >= 2 && NTP <= 4, "phases of 2 - 4 column tiles");
    constexpr PvFrCounts CNT = pv_fr_counts(NP, NTP, AP);
    const int srow8 = lane >> 3;
    const int schunk = (lane & 7) ^ (srow8 & 7);
    const char* ga[AP];
    const char* gw[NTP];
    int lw_off[NTP];                                     // (wave-uniform) LDS byte offset of the wave's W piece i of group 0 inside a buffer
#pragma unroll
    for (int j = 0; j < AP; ++j) {
        // (160-row tile: piece 2 of waves 4 - 7 would be rows 160 - 191: they fetch their piece 0 again - an L1 / L2 hit - into that spare LDS slot)
        const int jr = (j * 64 + wid * 8 < BM) ? j : 0;
        int ra = m0 + jr * 64 + wid * 8 + srow8; ra = ra < p.M ? ra : p.M - 1;
        ga[j] = reinterpret_cast<const char*>(p.A + (int64_t)ra * p.lda + schunk * 8);
    }
    // W group j = the 16 NTP weight rows each of the four column groups reads in phase j (rows wn * 16 NT + 16 NTP j + ..): 8 NTP pieces of 8 rows,
    // 2 NTP per column group; the wave's pieces are i * 8 + wid
#pragma unroll
    for (int i = 0; i < NTP; ++i) {
        const int pc = i * 8 + wid, r0 = (pc / (2 * NTP)) * 16 * NT + (pc % (2 * NTP)) * 8;
        gw[i] = reinterpret_cast<const char*>(p.W + (int64_t)(r0 + srow8) * p.ldw + schunk * 8);
        lw_off[i] = A_BYTES + r0 * 128;
    }
    const int64_t wg_stride = 32 * NTP * p.ldw;         // bytes between W groups (16 NTP rows)
    auto stage_a = [&](int buf, int kt) {
#pragma unroll
        for (int j = 0; j < AP; ++j) pv_glds16(ga[j] + kt * (BK * 2), smem + buf * BUF + j * 8192 + wid * 1024);
    };
    auto stage_w = [&](int buf, int grp, int kt) {
#pragma unroll
        for (int i = 0; i < NTP; ++i) pv_glds16(gw[i] + grp * wg_stride + kt * (BK * 2), smem + buf * BUF + lw_off[i] + grp * (2048 * NTP));
    };

    typedef __attribute__((address_space(3))) const char lds_cc;
    const int fx0 = ((lane >> 4) ^ (lane & 7)) << 4;
    lds_cc* a_rd[2][2];                                  // [buf][ks]   + mt * 2048
    lds_cc* w_rd[2][2];                                  // [buf][ks]   + nt * 2048
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            a_rd[b][ks] = (lds_cc*)smem + b * BUF + (wm * 16 * MT + i16) * 128 + (fx0 ^ (ks << 6));
            w_rd[b][ks] = (lds_cc*)smem + b * BUF + A_BYTES + (wn * 16 * NT + i16) * 128 + (fx0 ^ (ks << 6));
            asm volatile("" : "+v"(a_rd[b][ks]));
            asm volatile("" : "+v"(w_rd[b][ks]));
        }
    bf16x8 xf[MT][2], wf[NTP][2];

    const int nk = p.K / BK;
    // prologue, in the steady-state issue order: all of K-tile 0, then K-tile 1 but its last W group (phase 0 of K-tile 0 issues that)
    stage_a(0, 0);
#pragma unroll
    for (int gq = 0; gq < NP; ++gq) stage_w(0, gq, 0);
    stage_a(1, 1);
#pragma unroll
    for (int gq = 0; gq < NP - 1; ++gq) stage_w(1, gq, 1);
    pv_wait_vmcnt<CNT.pro>();                            // A(0), Wg0(0) (and the older bias) have landed: everything younger stays in flight
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (BIAS_LDS) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
            if (p.bias) b4 = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>((const __attribute__((address_space(3))) char*)smem + BIAS_OFF + (wn * 16 * NT + nt * 16 + 4 * g) * 4);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = b4;
        }
    }
    PV_STAMP(1);
    if (wm == 1) __builtin_amdgcn_s_barrier();          // stagger: group 1 runs one barrier interval behind group 0

    // MODE 0: steady state (s + 2 < nk), 1: K-tile nk - 2, 2 / 3: K-tile nk - 1 (3: behind the residual-row loads)
    auto phase = [&](auto buf_c, auto mode_c, auto j_c, int kt) __attribute__((always_inline)) {
        constexpr int B = decltype(buf_c)::value, MODE = decltype(mode_c)::value, j = decltype(j_c)::value;
        if constexpr (j == 0) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    xf[mt][ks] = *reinterpret_cast<const __attribute__((address_space(3))) bf16x8*>(a_rd[B][ks] + mt * 2048);
        }
#pragma unroll
        for (int t_ = 0; t_ < NTP; ++t_)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                wf[t_][ks] = *reinterpret_cast<const __attribute__((address_space(3))) bf16x8*>(w_rd[B][ks] + (NTP * j + t_) * 2048);
        if constexpr (j == 0) { if constexpr (MODE <= 1) stage_w(B ^ 1, NP - 1, kt + 1); }
        else if constexpr (MODE == 0) {
            if constexpr (j == 1) { stage_a(B, kt + 2); stage_w(B, 0, kt + 2); }
            else stage_w(B, j - 1, kt + 2);
        }
        // the counted wait for what the NEXT phase reads (nothing after the last phase of the last K-tile)
        if constexpr (MODE <= 1) pv_wait_vmcnt<CNT.v[MODE][j]>();
        else if constexpr (MODE == 2) { if constexpr (j < NP - 1) pv_wait_vmcnt<CNT.v[2][j]>(); }
        else if constexpr (j < NP - 1) pv_wait_vmcnt<CNT.v[2][j] + KC>();      // MODE 3: + the (at least) KC residual loads issued in front of this K-tile
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int t_ = 0; t_ < NTP; ++t_)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[NTP * j + t_][mt] = PV_MFMA_16x16x32(wf[t_][ks], xf[mt][ks], acc[NTP * j + t_][mt], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    auto ktile = [&](auto buf_c, auto mode_c, int kt) __attribute__((always_inline)) {
        phase(buf_c, mode_c, std::integral_constant<int, 0>{}, kt);
        phase(buf_c, mode_c, std::integral_constant<int, 1>{}, kt);
        if constexpr (NP > 2) phase(buf_c, mode_c, std::integral_constant<int, 2>{}, kt);
        if constexpr (NP > 3) phase(buf_c, mode_c, std::integral_constant<int, 3>{}, kt);
    };
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;
    using M0 = std::integral_constant<int, 0>;
    using M1 = std::integral_constant<int, 1>;
    using M2 = std::integral_constant<int, 2>;
    using M3 = std::integral_constant<int, 3>;
    int kt = 0;
    for (; kt + 4 <= nk; kt += 2) {
        ktile(B0{}, M0{}, kt);
        ktile(B1{}, M0{}, kt + 1);
    }
    ktile(B0{}, M1{}, kt);                               // K-tile nk - 2 (nk is even: the launcher sends other K to the plain loop)
    if constexpr (RES_AHEAD) {
        res_load(0, 0);                                  // batch 0 of pass 0; batch 1 has batch 0's arithmetic to arrive under
        __builtin_amdgcn_sched_barrier(0);
        ktile(B1{}, M3{}, kt + 1);                       // K-tile nk - 1 (>= KC register loads younger than every LDS-DMA)
    } else ktile(B1{}, M2{}, kt + 1);
    if (wm == 0) __builtin_amdgcn_s_barrier();          // balance the stagger barrier
    PV_STAMP(2);
  } else {
    // ---- LDS-DMA sources: 1-KiB pieces of 8 rows x 128 B per wave, swizzled chunk (chunk ^ (row & 7)) on the source side ----
    const int srow = wid * 8 + (lane >> 3);
    const int schunk = (lane & 7) ^ ((lane >> 3) & 7);
    const uint16_t* ga[2];
    const uint16_t* gw[NT];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        int ra = m0 + j * 64 + srow; ra = ra < p.M ? ra : p.M - 1;
        ga[j] = p.A + (int64_t)ra * p.lda + schunk * 8;
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) gw[j] = p.W + (int64_t)(j * 64 + srow) * p.ldw + schunk * 8;
    auto stage = [&](int buf, int kt) {
        char* l = smem + buf * BUF + wid * 1024;
#pragma unroll
        for (int j = 0; j < 2; ++j)
            if (j == 0 || !half) pv_glds16(ga[j] + kt * BK, l + j * 8192);          // (workgroup-uniform: every wave issues the same number of pieces)
#pragma unroll
        for (int j = 0; j < NT; ++j) pv_glds16(gw[j] + kt * BK, l + A_BYTES + j * 8192);
    };

    typedef __attribute__((address_space(3))) const char lds_cc;
    const int fx0 = ((lane >> 4) ^ (lane & 7)) << 4;
    lds_cc* a_rd = (lds_cc*)smem + (wm * 64 + i16) * 128 + fx0;                      // + buf * BUF + mt * 2048, ks: ^ 64
    lds_cc* w_rd = (lds_cc*)smem + A_BYTES + (wn * 16 * NT + i16) * 128 + fx0;       // + buf * BUF + nt * 2048
    asm volatile("" : "+v"(a_rd));
    asm volatile("" : "+v"(w_rd));

    const int nk = p.K / BK;
    stage(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                      // tile kt has landed for every wave; buffer cur ^ 1 is no longer read
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
        if (skip_mm) continue;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 xf[4], wf[NT];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
                xf[mt] = *reinterpret_cast<const __attribute__((address_space(3))) bf16x8*>((lds_cc*)((uintptr_t)a_rd ^ (ks << 6)) + cur * BUF + mt * 2048);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                wf[nt] = *reinterpret_cast<const __attribute__((address_space(3))) bf16x8*>((lds_cc*)((uintptr_t)w_rd ^ (ks << 6)) + cur * BUF + nt * 2048);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) acc[nt][mt] = PV_MFMA_16x16x32(wf[nt], xf[mt], acc[nt][mt], 0, 0, 0);
        }
    }

  }

    // ---- epilogue ------------------------------------------------------------------------------------------------------
    typedef __attribute__((address_space(3))) char lds_c;
    lds_c* const cimg = (lds_c*)smem;
    constexpr int GB_OFF = 64 * N * 4;                     // gamma | beta behind the 64-row fp32 image (2 N floats; the K-tile buffers are 256 N + 32 KiB)
    if constexpr (!RES_AHEAD) res_load(0, 0);
    if constexpr (RB == 2) res_load(0, 1);
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
        if (ps == 1 && half) break;                        // (workgroup-uniform) a 64-row tile has no second pass
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                      // K loop (ps = 0) / the previous pass's image reads (ps = 1) are done
#ifdef PV_STAMPS
        if (ps == 0) PV_STAMP(8);
#endif
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int gmt = wm * MT + mt;                                   // (wave-uniform) row tile of the workgroup's tile: pass gmt / 4, image rows (gmt % 4) * 16 ..
            if ((gmt >> 2) != ps) continue;
            const int row = (gmt & 3) * 16 + i16;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int c = wn * 4 * NT + nt * 4 + g;                     // 16-byte chunk of the row
                *reinterpret_cast<__attribute__((address_space(3))) f32x4*>(cimg + row * (N * 4) + ((c ^ (i16 & 7)) << 4)) = acc[nt][mt];
            }
        }
        if (ps == 0 && p.ln_out && tid < N) {              // (the region is beyond every row of the image)
            *reinterpret_cast<__attribute__((address_space(3))) float*>(cimg + GB_OFF + tid * 4) = ln_g1;
            *reinterpret_cast<__attribute__((address_space(3))) float*>(cimg + GB_OFF + N * 4 + tid * 4) = ln_b1;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#ifdef PV_STAMPS
        if (ps == 0) PV_STAMP(10);
#endif
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const bool short_pass = MT == 5 && ps == 2;                     // (compile-time after unrolling) the 32-row pass: one batch of four rows per wave
            if (short_pass && b == 1) break;
            const int row = short_pass ? wid * 4 + q4 : wid * 8 + 4 * b + q4, m = m0 + ps * 64 + row;
            float4 v[KC];
#pragma unroll
            for (int k = 0; k < KC; ++k) {
                const int c = l16 + 16 * k;
                const f32x4 im = *reinterpret_cast<const __attribute__((address_space(3))) f32x4*>(cimg + row * (N * 4) + ((c ^ (row & 7)) << 4));
                // (scalar FMAs by inline asm: the packed form with an op_sel bit misreads lanes 48-63 while residual rows are still returning,
                //  pv_common.h pv_add_s)
                v[k] = make_float4(pv_fma_s(sc[b % RB], im[0], rr[b % RB][k][0]), pv_fma_s(sc[b % RB], im[1], rr[b % RB][k][1]),
                                   pv_fma_s(sc[b % RB], im[2], rr[b % RB][k][2]), pv_fma_s(sc[b % RB], im[3], rr[b % RB][k][3]));
            }
            const float lsc_b = lsc[b % RB];
            // the next rows into the registers just consumed (workgroup-uniform branches): two batches in flight -> the same batch of pass 1,
            // one -> the next batch
            __builtin_amdgcn_sched_barrier(0);
            if (RB == 2) { if (ps + 1 < NPASS && !half && !(MT == 5 && ps == 1 && b == 1)) res_load(ps + 1, b); }      // (the 160-row tile's third pass has one batch)
            else if (b == 0 && !short_pass) res_load(ps, 1);
            else if (ps + 1 < NPASS && !half) res_load(ps + 1, 0);
            if (m < p.M) {
                float* o = reinterpret_cast<float*>(p.out) + (int64_t)m * p.ldo + l16 * 4;
#pragma unroll
                for (int k = 0; k < KC; ++k) PV_STORE32(reinterpret_cast<f32x4*>(o + 64 * k), ((f32x4){v[k].x, v[k].y, v[k].z, v[k].w}));
            }
#ifdef PV_STAMPS
            if (ps == 0 && b == 0) PV_STAMP(11);
#endif
            if (p.ln_out) {                                // (workgroup-uniform)
                // p.N (= N), not the constant: the standalone kernel divides by a RUN-TIME D, and hipcc's division by a run-time value and by
                // a power-of-two constant differ in the last bit (found by the bit-identity test at N = 512)
                pv_ln_row16<KC>(v, (const __attribute__((address_space(3))) char*)cimg + GB_OFF, p.N, l16, p.ln_eps);
                if (m < p.M) {
                    u32x2* o = reinterpret_cast<u32x2*>(p.ln_out + (int64_t)m * N) + l16;
#pragma unroll
                    for (int k = 0; k < KC; ++k) {
                        u32x2 pk = {pv_pack_bf16x2(pv_mul_s(v[k].x, lsc_b), pv_mul_s(v[k].y, lsc_b)), pv_pack_bf16x2(pv_mul_s(v[k].z, lsc_b), pv_mul_s(v[k].w, lsc_b))};
                        PV_STORE16(o + 16 * k, pk);
                    }
                }
            }
#ifdef PV_STAMPS
            if (ps == 0 && b == 0) PV_STAMP(12);
#endif
        }
        PV_STAMP(ps < 2 ? 3 + ps : 13);
    }
#ifdef PV_STAMPS
    if (threadIdx.x == 0 && p.dbg) { unsigned long long rt_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt_)::"memory"); p.dbg[pv_stamp_slot * 16 + 6] = rt_; }
#endif
}

template <int NT, int DPH = 0>
__global__ __launch_bounds__(512) void pv_gemm_fullrow_kernel(const GemmDev p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BM = 128;
    // Split remainder (round 3): 788 tiles of 128 rows on 256 CUs are 3.08 rounds - a quarter of the last one idle in every XCD.  With
    // fr_full > 0 the launch is whole rounds of 128-row tiles plus ONE round of 64-row tiles over the remaining rows; every XCD (workgroup id
    // & 7) takes an equal share of both lists, full tiles first.  A 64-row tile is this same code with wave group 1 (rows 64 - 127) idle:
    // no A staging and no MFMAs for it, one epilogue pass - per row the arithmetic is unchanged.
    // Round 6: fr_big > 0 - fr_big tiles of 160 rows FIRST in the grid (the longest tiles start first), over the rows behind the fr_full (a multiple
    // of 8, possibly 0) 128-row tiles; again every XCD takes an equal share of both lists.
    int m0_;
    bool half = false, big = false;
    if (p.fr_big > 0) {
        const int nb8 = (p.fr_big + 7) >> 3;
        const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
        if (j < nb8) {
            const int q = p.fr_big >> 3, r = p.fr_big & 7;
            if (j >= q + (x < r ? 1 : 0)) return;           // (whole workgroup: before any barrier)
            m0_ = p.fr_full * BM + ((x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + j) * 160;
            big = true;
        } else m0_ = (x * (p.fr_full >> 3) + (j - nb8)) * BM;
    } else if (p.fr_full > 0) {
        const int x = blockIdx.x & 7, j = blockIdx.x >> 3, fper = p.fr_full >> 3, q = p.fr_half >> 3, r = p.fr_half & 7;
        if (j < fper) m0_ = (x * fper + j) * BM;
        else {
            const int hj = j - fper;
            if (hj >= q + (x < r ? 1 : 0)) return;          // (whole workgroup: before any barrier)
            m0_ = p.fr_full * BM + ((x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + hj) * 64;
            half = true;
        }
    } else m0_ = pv_xcd_remap(blockIdx.x, p.tiles_m) * BM;
    if constexpr (DPH > 0 && NT <= 6) {
        if (big) { pv_fullrow_body<NT, DPH, 5>(p, smem, m0_, false); return; }      // (workgroup-uniform)
    }
    pv_fullrow_body<NT, DPH, 4>(p, smem, m0_, half);
}

static int g_pv_frdp = -1;         // -1: PV_FULLROW_DP / default; 0 / 1: A/B override (scripts/fullrow_ab4.py)
extern "C" void pv_debug_set_fullrow_dp(int on) { g_pv_frdp = on; }
static int pv_fullrow_dp_mode() {           // 0: plain K loop, otherwise the deep-pipelined one
    static const int env = [] { const char* e = getenv("PV_FULLROW_DP"); return e ? atoi(e) : 1; }();
    return g_pv_frdp >= 0 ? g_pv_frdp : env;
}
static int g_pv_frsplit = -1;
extern "C" void pv_debug_set_fullrow_split(int on) { g_pv_frsplit = on; }
static int pv_fullrow_split_mode() {       // 0: plain 128-row tiles, 1: split remainder (round 3), 2: a last round of 160-row tiles where it applies, else as 1 (round 6)
    static const int env = [] { const char* e = getenv("PV_FULLROW_SPLIT"); return e ? atoi(e) : 2; }();
    return g_pv_frsplit >= 0 ? g_pv_frsplit : env;
}
static bool pv_fullrow_split_enabled() { return pv_fullrow_split_mode() != 0; }

template <int NT>
static int pv_launch_gemm_fullrow(const GemmDev& p, hipStream_t stream) {
    static PvPerDevice attr_set;
    constexpr int lds = 2 * (128 * 64 * 2 + 64 * NT * 64 * 2);
    constexpr int lds_big = 2 * (192 * 64 * 2 + 64 * NT * 64 * 2) + 256 * NT;      // 160-row tiles (24 KiB of A per K-tile buffer) + the bias
    if (attr_set.first_use()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pv_gemm_fullrow_kernel<NT, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pv_gemm_fullrow_kernel<NT, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, NT <= 6 ? lds_big : lds);      // (+ the bias)
    }
    // split remainder: whole rounds of 128-row tiles, then the rest as ONE round of 64-row tiles - when there is more than one round and the
    // remainder fits half a round (otherwise plain 128-row tiles; PV_FULLROW_SPLIT=0 / pv_debug_set_fullrow_split(0): A/B)
    GemmDev q = p;
    q.fr_full = 0; q.fr_half = 0; q.fr_big = 0;
    unsigned grid = (unsigned)p.tiles_m;
    const int cus = pv_cu_count() & ~7;
    const bool dp = p.K % 128 == 0 && pv_fullrow_dp_mode() != 0;
    // round 6: the remainder of the last whole round is small (<= 32 rows per CU): take one round of 128-row tiles back and cover its rows and
    // the remainder with ONE round of 160-row tiles: 3.08 rounds of work in 2 + 1.25 tile times instead of 3 + ~0.85
    if (pv_fullrow_split_mode() == 2 && NT <= 6 && dp && cus >= 8 && p.tiles_m > cus) {
        const int rounds = p.tiles_m / cus;
        const int64_t rem = p.M - (int64_t)rounds * cus * 128;
        if (rem > 0 && rem <= (int64_t)cus * 32) {
            q.fr_full = (rounds - 1) * cus;
            q.fr_big = (int)((p.M - (int64_t)q.fr_full * 128 + 159) / 160);
            grid = 8u * (unsigned)((q.fr_big + 7) / 8) + (unsigned)q.fr_full;
        }
    }
    if (pv_fullrow_split_mode() == 3 && NT <= 6 && dp && cus >= 8 && p.tiles_m > cus) {      // experiment: 160-row tiles only
        q.fr_full = 0; q.fr_big = (int)((p.M + 159) / 160);
        grid = 8u * (unsigned)((q.fr_big + 7) / 8);
    }
    if (q.fr_big == 0 && pv_fullrow_split_enabled() && cus >= 8 && p.tiles_m > cus) {
        const int full = p.tiles_m / cus * cus;
        const int64_t rem = p.M - (int64_t)full * 128;
        if (rem > 0 && rem <= (int64_t)cus * 64) {
            q.fr_full = full; q.fr_half = (int)((rem + 63) / 64);
            grid = 8u * (unsigned)(full / 8 + (q.fr_half + 7) / 8);
        }
    }
    // the deep-pipelined loop (K-tiles in pairs) with TWO phases of 8 NT MFMAs per K-tile: phases of 16 MFMAs (NT / 2 of them, the first form of
    // round 4) cost a barrier pair more per K-tile at N = 384 / 512 and measured 0.4 % behind in the model (scripts/vit_small_inproc_ab.py)
    if (dp) PV_LAUNCH((pv_gemm_fullrow_kernel<NT, 2>), dim3(grid), dim3(512), q.fr_big > 0 ? lds_big : NT <= 6 ? lds + 256 * NT : lds, stream, q);
    else PV_LAUNCH((pv_gemm_fullrow_kernel<NT, 0>), dim3(grid), dim3(512), lds, stream, q);
    return pv_check_launch();
}

// tile-raster override for the L2 experiments (scripts/bench_gemm.py): gm / gc <= 0 restore the built-in choice
static int g_pv_raster_gm = 0, g_pv_raster_gc = 0;
extern "C" void pv_debug_set_gemm_raster(int gm, int gc) { g_pv_raster_gm = gm; g_pv_raster_gc = gc; }
static int g_pv_fullrow = -1;      // -1: PV_GEMM_FULLROW / default; 0 / 1: A/B override (scripts/fullrow_ab.py)
extern "C" void pv_debug_set_gemm_fullrow(int on) { g_pv_fullrow = on; }

static int pv_gemm_dispatch(const pv_gemm_args* a, void* stream, bool query_only);
extern "C" int pv_gemm_bf16(const pv_gemm_args* a, void* stream) { return pv_gemm_dispatch(a, stream, false); }
extern "C" int pv_gemm_tile_rows(const pv_gemm_args* a) {
    if (!a || a->struct_size != sizeof(pv_gemm_args)) return PV_ERR_INVALID_ARG;
    pv_gemm_args q = *a;
    q.colsum_partial = nullptr; q.x16_out = nullptr; q.rowstat_out = nullptr; q.fold_stat = nullptr; q.rowsq_out = nullptr;
    return pv_gemm_dispatch(&q, nullptr, true);
}
static int pv_gemm_dispatch(const pv_gemm_args* a, void* stream, bool query_only) {
    if (!a || a->struct_size != sizeof(pv_gemm_args)) return PV_ERR_INVALID_ARG;      // nothing past the first field is read before this
    if (!a->A || !a->W || !a->out || a->M <= 0 || a->N <= 0 || a->K <= 0) return PV_ERR_INVALID_ARG;
    if (a->K % 64 || a->N % 4) return PV_ERR_UNSUPPORTED;
    if (a->lda % 8 || a->ldw % 8 || a->ldo % 4 || a->lda < a->K || a->ldw < a->K || a->ldo < a->N) return PV_ERR_INVALID_ARG;
    if (((uintptr_t)a->A & 15) || ((uintptr_t)a->W & 15) || ((uintptr_t)a->out & 15) || (a->bias && ((uintptr_t)a->bias & 15))) return PV_ERR_INVALID_ARG;
    if (a->M > 0x7fffffff || a->N > 0x7fffffff || a->K > 0x7fffffff) return PV_ERR_UNSUPPORTED;
    if ((uintptr_t)a->range_flag & 3) return PV_ERR_INVALID_ARG;
    GemmDev p;
    p.range_flag = a->range_flag;
    p.rowsq_out = a->rowsq_out;
    if (a->rowsq_out && (a->epilogue != PV_EPI_BIAS_RES_F32 || ((uintptr_t)a->rowsq_out & 3) || a->ln_out || a->x16_out)) return PV_ERR_INVALID_ARG;
    p.A = a->A; p.W = a->W; p.bias = a->bias; p.out = a->out; p.res = a->res; p.row_scale = a->row_scale; p.pos = a->pos;
    p.res_scaled = a->res_scaled;
    if (a->res_scaled && (a->epilogue != PV_EPI_BIAS_RES_F32 || !a->row_scale)) return PV_ERR_INVALID_ARG;
    p.M = (int)a->M; p.N = (int)a->N; p.K = (int)a->K;
    p.lda = a->lda; p.ldw = a->ldw; p.ldo = a->ldo; p.ldr = a->ldr;
    p.rpi = (int)a->rows_per_img_in; p.rpo = (int)a->rows_per_img_out; p.row_off = (int)a->row_off;
    p.qcols = (int)a->qcols; p.qscale = a->qscale;
#ifdef PV_STAMPS
    p.dbg = g_pv_dbg;
#endif
    p.colsum_partial = a->colsum_partial;
    p.x16_out = a->x16_out; p.rowstat_out = a->rowstat_out; p.fold_stat = a->fold_stat; p.fold_c1 = a->fold_c1; p.fold_c2 = a->fold_c2;
    if ((a->x16_out != nullptr) != (a->rowstat_out != nullptr)) return PV_ERR_INVALID_ARG;
    if (a->x16_out && (a->epilogue != PV_EPI_BIAS_RES_F32 || ((uintptr_t)a->x16_out & 7) || ((uintptr_t)a->rowstat_out & 7))) return PV_ERR_INVALID_ARG;
    if (a->fold_stat && ((a->epilogue != PV_EPI_BIAS_BF16 && a->epilogue != PV_EPI_BIAS_GELU_BF16) || a->bias || !a->fold_c1 || !a->fold_c2 || a->N < 8 ||
                         ((uintptr_t)a->fold_stat & 7) || ((uintptr_t)a->fold_c1 & 15) || ((uintptr_t)a->fold_c2 & 15))) return PV_ERR_INVALID_ARG;
    if (a->colsum_partial && (a->epilogue != PV_EPI_GELU_GRAD_BF16 || ((uintptr_t)a->colsum_partial & 15))) return PV_ERR_INVALID_ARG;
    p.ksplit = a->ksplit > 1 ? a->ksplit : 1; p.k_slice = (int)(a->K / p.ksplit); p.split_stride = a->M * a->ldo;
    if (p.ksplit > 1) {
        // split-K: fp32 partial slices out[t] (t < ksplit) of M*ldo floats each, summed by pv_sum_slices_f32; bias goes to slice 0
        if (a->epilogue != PV_EPI_BIAS_F32 || a->qcols != 0 || a->ln_out || a->K % (p.ksplit * 64)) return PV_ERR_INVALID_ARG;
    }
    p.ln_gamma = a->ln_gamma; p.ln_beta = a->ln_beta; p.ln_row_scale = a->ln_row_scale; p.ln_out = a->ln_out; p.ln_eps = a->ln_eps;
    hipStream_t s = (hipStream_t)stream;
    if (a->ln_out) {
        // fused LayerNorm: whole rows per workgroup -> N is the hidden dim, a multiple of 256, K a multiple of 128
        if (a->epilogue != PV_EPI_BIAS_RES_F32 || !a->ln_gamma || !a->ln_beta) return PV_ERR_INVALID_ARG;
        if (!(a->N == 256 || a->N == 384 || a->N == 512) && (a->N % G2_BN || a->N > 4096 || a->K % (2 * G2_BK))) return PV_ERR_UNSUPPORTED;
        if (a->ldo != a->N) return PV_ERR_UNSUPPORTED;
        if (((uintptr_t)a->ln_out & 7) || ((uintptr_t)a->ln_gamma & 15) || ((uintptr_t)a->ln_beta & 15)) return PV_ERR_INVALID_ARG;
    }
    if (a->epilogue == PV_EPI_BIAS_RES_F32 && (!a->res || a->ldr % 4 || a->ldr < a->N || ((uintptr_t)a->res & 15))) return PV_ERR_INVALID_ARG;
    if (a->epilogue == PV_EPI_GELU_GRAD_BF16 && (!a->res || a->ldr % 4 || a->ldr < a->N || ((uintptr_t)a->res & 7))) return PV_ERR_INVALID_ARG;
    if (a->epilogue == PV_EPI_BIAS_POS_F32 &&
        (!a->pos || a->rows_per_img_in <= 0 || a->rows_per_img_out < a->rows_per_img_in + a->row_off || a->row_off < 0 || ((uintptr_t)a->pos & 15)))
        return PV_ERR_INVALID_ARG;
    // kernel choice: the deep-pipelined 256^2 tile for the big token GEMMs, the 128^2 tile for everything else
    static const int force = [] { const char* e = getenv("PV_GEMM_TILE"); return e ? atoi(e) : 0; }();
    // 256^2 tile: N a multiple of 128 (a ragged last column tile is clamped on load and guarded on store; worth it while
    // it wastes <= 25 % of the tiles), K a multiple of 128, enough rows to fill the chip
    const int tn256 = (p.N + G2_BN - 1) / G2_BN;
    const bool n_ok = p.N % 128 == 0 && (int64_t)tn256 * G2_BN * 3 <= (int64_t)p.N * 4;
    // (the 256^2 epilogue applies the q-scale per 8-column chunk, the 128^2 one per 4 columns)
    const int k_eff = p.ksplit > 1 ? p.k_slice : p.K;
    // "enough rows to fill the chip", measured per shape (scripts/gemm128_ab.py, profiles/r02_gemm128_epilogue_ab.json): the 256^2 tile
    // wins from ~128 tiles on (vit_tiny at batch 32: 153 tiles 12.9 vs 14.9 us) and loses below (51 tiles: 16.9 vs 9.8 us); with a
    // quarter of the last column tile empty AND a short K (N = 384, K = 384: out-proj of vit_small) the 128^2 tile wins as well
    // (96.9 vs 107.7 us), at K = 1536 the two tie.
    const int64_t tiles256 = (int64_t)((p.M + G2_BM - 1) / G2_BM) * tn256;
    const bool ragged_short = (int64_t)tn256 * G2_BN * 3 >= (int64_t)p.N * 4 && k_eff <= 512;
    // an epilogue feature only the 256-row tile kernel has (the caller asked pv_gemm_tile_rows, or insists): take that kernel if the shape allows
    const bool feat = a->colsum_partial || a->x16_out || a->fold_stat || a->rowsq_out;
    const bool big = !((a->epilogue == PV_EPI_BIAS_BF16 || a->epilogue == PV_EPI_BIAS_F32) && p.qcols % 8) &&
                     (force == 256 || (force != 128 && n_ok && k_eff % (2 * G2_BK) == 0 &&
                                       ((tiles256 >= 128 && !ragged_short) || feat ||
                                        (p.ksplit > 1 && (int64_t)p.M * p.N >= 256 * 256 && tiles256 * p.ksplit >= 64))));   // (a few-row split-K GEMM - small-batch residual GEMMs - fills more CUs with 128^2 tiles)
    // (the 256^2 kernel's gelu' product moves its 16-bit rows 16 bytes per lane since round 6)
    if (big && a->epilogue == PV_EPI_GELU_GRAD_BF16 && (p.N % 8 || a->ldr % 8 || a->ldo % 8 || ((uintptr_t)a->res & 15) || ((uintptr_t)a->out & 15))) return PV_ERR_UNSUPPORTED;
    if (big && (k_eff % (2 * G2_BK) || k_eff < 2 * G2_BK)) return PV_ERR_UNSUPPORTED;
    if ((a->colsum_partial || a->x16_out || a->fold_stat || a->rowsq_out) && !big) return PV_ERR_UNSUPPORTED;   // 256-row tile kernel only (pv_gemm_tile_rows)
    if (query_only) return big ? G2_BM : G1_BM;
    if ((a->epilogue == PV_EPI_BIAS_BF16 || a->epilogue == PV_EPI_BIAS_F32) && p.qcols % 4) return PV_ERR_UNSUPPORTED;
    static const int gm_env = [] { const char* e = getenv("PV_GEMM_GM"); return e ? atoi(e) : 0; }();
    const int bm = big ? G2_BM : G1_BM, bn = big ? G2_BN : G1_BN;
    p.tiles_m = (p.M + bm - 1) / bm; p.tiles_n = (p.N + bn - 1) / bn;
    // tile raster of the 256^2 kernel, measured per shape at M = 403456 with interleaved rounds and PMC passes
    // (scripts/raster_ab.py, profiles/r02_raster_ab.json): 12 column tiles (fc1) -> two chunks of 6 columns x groups of 6 row
    // panels (+1.6 %), 9 column tiles (QKV) -> groups of 8 row panels (+1.8 %) over the r1 choice gm = 4; N = 768: gm = 1.
    // None of the rasters / cache policies moves the L2-side fetch below ~4x the algorithmic read (4 MiB of L2 per XCD against
    // 32 co-resident 256-wide tiles), and fetch volume does not order the timings: the package sits at its power cap in all of them.
    p.gm = p.tiles_n >= 8 ? (p.tiles_n >= 12 ? 6 : 8) : (p.tiles_n >= 6 ? 4 : 1);
    p.gc = p.tiles_n >= 12 ? 6 : p.tiles_n;
    if (gm_env > 0) p.gm = gm_env;
    if (g_pv_raster_gm > 0) p.gm = g_pv_raster_gm;
    if (g_pv_raster_gc > 0) p.gc = g_pv_raster_gc < p.tiles_n ? g_pv_raster_gc : p.tiles_n;
    if ((int64_t)p.tiles_m * p.tiles_n > 0x7fffffff) return PV_ERR_UNSUPPORTED;
    // full-row tile (128 x N, N = 256 / 384 / 512): the residual GEMMs of narrow models, with or without the fused LayerNorm
    static const int fullrow_env = [] { const char* e = getenv("PV_GEMM_FULLROW"); return e ? atoi(e) : 1; }();
    const bool fullrow_shape = a->epilogue == PV_EPI_BIAS_RES_F32 && (p.N == 256 || p.N == 384 || p.N == 512) && p.ksplit <= 1 && !feat &&
                               (g_pv_fullrow >= 0 ? g_pv_fullrow != 0 : fullrow_env != 0) && force == 0;
    if (a->res_scaled && (fullrow_shape || a->ln_out)) return PV_ERR_UNSUPPORTED;      // 256- / 128-row tile kernels only
    if (fullrow_shape && (a->ln_out || (p.M + 127) / 128 >= 96)) {
        p.tiles_m = (p.M + 127) / 128; p.tiles_n = 1;
        return p.N == 256 ? pv_launch_gemm_fullrow<4>(p, s) : p.N == 384 ? pv_launch_gemm_fullrow<6>(p, s) : pv_launch_gemm_fullrow<8>(p, s);
    }
    if (a->ln_out) {
        p.gm = 1; p.gc = p.tiles_n;
        p.tiles_m = (p.M + G2_BM - 1) / G2_BM; p.tiles_n = p.N / G2_BN;
        return pv_launch_gemm256_rows<PV_EPI_BIAS_RES_F32>(p, s);
    }
    switch (a->epilogue) {
        case PV_EPI_BIAS_BF16: return big ? pv_launch_gemm256<PV_EPI_BIAS_BF16>(p, s) : pv_launch_gemm128<PV_EPI_BIAS_BF16>(p, s);
        case PV_EPI_BIAS_GELU_BF16: return big ? pv_launch_gemm256<PV_EPI_BIAS_GELU_BF16>(p, s) : pv_launch_gemm128<PV_EPI_BIAS_GELU_BF16>(p, s);
        case PV_EPI_BIAS_RES_F32: return big ? pv_launch_gemm256<PV_EPI_BIAS_RES_F32>(p, s) : pv_launch_gemm128<PV_EPI_BIAS_RES_F32>(p, s);
        case PV_EPI_BIAS_POS_F32: return big ? pv_launch_gemm256<PV_EPI_BIAS_POS_F32>(p, s) : pv_launch_gemm128<PV_EPI_BIAS_POS_F32>(p, s);
        case PV_EPI_BIAS_F32: return big ? pv_launch_gemm256<PV_EPI_BIAS_F32>(p, s) : pv_launch_gemm128<PV_EPI_BIAS_F32>(p, s);
        case PV_EPI_BIAS_GELU_SPLIT_BF16:
            if (a->ldo < 3 * a->N) return PV_ERR_INVALID_ARG;
            return big ? pv_launch_gemm256<PV_EPI_BIAS_GELU_SPLIT_BF16>(p, s) : pv_launch_gemm128<PV_EPI_BIAS_GELU_SPLIT_BF16>(p, s);
        case PV_EPI_BIAS_GELU_PAIR_BF16:
            if (a->ldo < 2 * a->N) return PV_ERR_INVALID_ARG;
            return big ? pv_launch_gemm256<PV_EPI_BIAS_GELU_PAIR_BF16>(p, s) : pv_launch_gemm128<PV_EPI_BIAS_GELU_PAIR_BF16>(p, s);
        case PV_EPI_GELU_GRAD_BF16: return big ? pv_launch_gemm256<PV_EPI_GELU_GRAD_BF16>(p, s) : pv_launch_gemm128<PV_EPI_GELU_GRAD_BF16>(p, s);
        default: return PV_ERR_INVALID_ARG;
    }
}
