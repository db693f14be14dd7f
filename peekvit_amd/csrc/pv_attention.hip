// Fused multi-head self-attention core for ViT-length sequences on gfx950 (S <= 416: LDS-resident K/V; longer: pv_attn_stream_kernel).
//   out[b,s,h*dh:(h+1)*dh] = softmax(q k^T) v          q pre-scaled by dh^-0.5 in the in-proj epilogue
// One workgroup (4 waves) per (image, head).  The whole K and V of the head live in LDS (<= 53 KiB each), so
// softmax is single pass: no online rescale.  Per 16-query tile a wave computes
//   S^T = K . Q^T   (MFMA 16x16x32, A = K rows from LDS, B = Q rows straight from global)
// "swapped", so a lane holds 4 consecutive KEYS of ONE query per accumulator tile: the row max / sum are
// in-lane reductions plus two cross-lane steps, and the bf16-packed P^T registers are directly the B operand of
//   O^T = V^T . P^T (A = V^T via ds_read_b64_tr_b16 transposed LDS reads of the row-major V image)
// in a permuted-but-consistent k order (key slot j of lane group g: j<4 -> key 32t+4g+j, j>=4 -> key 32t+16+4g+j-4).
// LDS images are XOR-swizzled per 128-byte line (chunk ^ (line & 7)): conflict-free for both read kinds.
#include "pv_common.h"
#include <type_traits>

typedef __attribute__((ext_vector_type(8))) short s16x8;

template <int CPR>   // 16-byte chunks per row (DHP / 8)
__device__ __forceinline__ int pv_swz(int row, int chunk) {
    const int L = row * CPR + chunk, line = L >> 3, pos = L & 7;
    return ((line << 3) + (pos ^ (line & 7))) << 4;
}

// (image, head) of workgroup i, XCD-aware (round 5).  The hardware deals consecutive workgroups to the eight XCDs in turn, each with its own L2, and
// blockIdx -> (b = i / H, h = i % H) therefore spreads the heads of ONE image over all eight L2s.  A head's slice of a token row is 2 * dh bytes:
// at dh = 64 exactly one 128-byte line, but at dh = 48 (vit_small) 96 bytes that straddle lines and at dh = 32 (vit_tiny) half a line, so every L2
// fetched the lines it shares with its neighbours' heads again: rocprofv3 FETCH_SIZE 465 MB per launch against 232 MB of q|k|v at vit_small, batch 512
// (profiles/r05_vit_small_kernel_summary.json) - the kernel ran at 5.5 TB/s of fabric traffic, half of it redundant.  Here workgroups 8q + x, q = k H ..
// k H + H - 1, are the H heads of image 8k + x: one image's heads share an XCD and are dispatched back to back.  `PV_BH_XCD=0` restores i / H, i % H.
#ifndef PV_BH_XCD
#define PV_BH_XCD 1
#endif
__device__ __forceinline__ void pv_bh_map(int i, int B, int H, int& b, int& h) {
    const int nfull = PV_BH_XCD ? (B >> 3) * 8 * H : 0;        // workgroups of complete groups of eight images
    if (i < nfull) {
        const int q = i >> 3, k = q / H;
        b = k * 8 + (i & 7);
        h = q - k * H;
    } else {
        const int r = i - nfull;
        b = (nfull / H) + r / H;
        h = r % H;
    }
#ifdef PV_BH_HOT       // experiment (scripts/attn_hot.py): every workgroup works on one of eight images, whose operands stay in the L2s - what the kernels cost without HBM
    b &= 7;
#endif
}

#ifdef PV_STAMPS
__device__ unsigned long long* d_pv_adbg;     // diagnostic build only: stamps go to a buffer no kernel reads
extern "C" void pv_debug_set_attn_stamp_buffer(void* p) { (void)hipMemcpyToSymbol(HIP_SYMBOL(d_pv_adbg), &p, sizeof(p)); }
#define PV_ASTAMP(i)                                                                                       \
    do {                                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
        unsigned long long t_;                                                                             \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                          \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
        if ((threadIdx.x & 63) == 0 && d_pv_adbg) d_pv_adbg[((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + (i)] = t_;   \
    } while (0)
#define PV_CSTAMP(i)       /* sixteen-wave persistent kernel (pv_attn_bwd5_kernel; scripts/stamp_attn_bwd5.py): the last-but-one item of every workgroup */  \
    do {                                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
        unsigned long long t_;                                                                             \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                          \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
        if ((threadIdx.x & 63) == 0 && d_pv_adbg && pv_stamp_on) d_pv_adbg[((size_t)blockIdx.x * 16 + (threadIdx.x >> 6)) * 8 + (i)] = t_;   \
    } while (0)
#else
#define PV_ASTAMP(i)
#define PV_CSTAMP(i)
#endif

typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;

// Probabilities are packed as p * 2^PV_P_SHIFT in the fp16 build (the fp16 MFMA flushes subnormal operands; the factor cancels in O / l).
// Round 4: 2^10 instead of round 3's 2^14 - p <= 1 leaves SIX bits of fp16 headroom instead of two (an exponent computed against a
// maximum that is off by up to 4 score units still packs a finite value; round 3's carried-maximum experiment turned exactly such
// values into inf and then NaN rows), and everything down to p = 6e-8 stays a normal number: at most 197 x 6e-8 = 1.2e-5 of a row's
// mass can flush, two orders below the contract.  The streaming kernel (S > 416, wide heads) now applies the same scale.
#ifndef PV_P_SHIFT
#ifdef PV_OPERAND_F16
#define PV_P_SHIFT 10.0f
#else
#define PV_P_SHIFT 0.0f
#endif
#endif
#define PV_P_UNSHIFT (1.0f / (float)(1 << (int)PV_P_SHIFT))       // 2^-PV_P_SHIFT, exact

#ifndef PV_ATTN_NW
#define PV_ATTN_NW 4         // waves per workgroup of pv_attn_kernel (A/B: 8 waves x 2 workgroups per CU instead of 4 x 3; scripts/attn_ab.py)
#endif
template <int DH, int NKT, bool LSE = false>     // NKT = number of 16-key tiles = ceil(S / 16); LSE: the training forward, which also writes the rows' log-sum-exp
__global__ __launch_bounds__(PV_ATTN_NW * 64) void pv_attn_kernel(const uint16_t* __restrict__ qkv, uint16_t* __restrict__ out, int S, int H, uint32_t* flag, int B, float* __restrict__ lse) {
    constexpr int DHP = (DH + 31) / 32 * 32;
    constexpr int CPR = DHP / 8;
    constexpr int KS = DHP / 32;        // k-steps of the QK^T product
    constexpr int NKT32 = NKT / 2;      // full 32-key steps of the PV product (an odd last tile uses the K=16 MFMA)
    constexpr int SP = NKT * 16;        // padded key count: LDS holds exactly SP rows of K and of V
    constexpr int NDT = DH / 16;        // 16-wide output d tiles
    constexpr int NCH = SP * CPR;       // 16-byte chunks per K (or V) image
    constexpr int NW = PV_ATTN_NW, NT = NW * 64;
    constexpr int NIT = (NCH + NT - 1) / NT;
    constexpr int MAXQT = (NKT + NW - 1) / NW;   // q tiles per wave
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Ks = smem;
    char* Vs = smem + SP * DHP * 2;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, i16 = lane & 15;
    int b, h;
    pv_bh_map(blockIdx.x, B, H, b, h);
    PV_ASTAMP(0);
    const int D = H * DH;
    const int64_t ld = 3 * (int64_t)D;
    const uint16_t* qb = qkv + (int64_t)b * S * ld + h * DH;

    // ---- Q^T fragments of every q tile of this wave (B operand), issued FIRST (needed first):
    // lane (g,i16) holds Q[q0+i16][ks*32 + 8g .. +8] ---------------------------------------------------------------------
    const int nqt = (S + 15) >> 4;
    bf16x8 qf[MAXQT][KS];
#pragma unroll
    for (int t = 0; t < MAXQT; ++t) {
        int qr = (wid + NW * t) * 16 + i16;
        qr = qr < S ? qr : S - 1;
        const uint16_t* qp = qb + (int64_t)qr * ld;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int dcol = ks * 32 + 8 * g;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (dcol < DH) v = *reinterpret_cast<const u32x4*>(qp + dcol);
            qf[t][ks] = __builtin_bit_cast(bf16x8, v);
        }
    }
    // ---- stage K, then V, by LDS-DMA: every piece issued up front; K is waited for first so QK^T starts under V's flight ----
    // physical chunk P = it*256 + tid; the swizzle permutes inside a 128-byte line only, so L>>3 = P>>3 and the lane's logical
    // chunk-in-line (tid&7)^((tid>>3)&7) is the same for every iteration: row = it*(256/CPR) + lane term, c = lane term.
    // Rows >= S duplicate row S-1: their scores are masked to -inf below and P = 0 multiplies the (finite) duplicate V rows.
    const int lsw = (tid & 7) ^ ((tid >> 3) & 7);
    const int r_lane = CPR == 8 ? (tid >> 3) : 2 * (tid >> 3) + (lsw >> 2);
    int c_lane = CPR == 8 ? lsw : (lsw & 3);
    if (c_lane * 8 >= DH) c_lane = 0;          // DH = 48: the pad chunks only ever meet zero Q columns / unused d tiles
    const uint16_t* const kv_src = qb + D + c_lane * 8;
#pragma unroll
    for (int kv = 0; kv < 2; ++kv) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            if (NCH % NT == 0 || it * NT + wid * 64 < NCH) {
                int row = it * (NT / CPR) + r_lane;
                row = row < S ? row : S - 1;
                const uint16_t* src = kv_src + (int64_t)row * ld + kv * D;
                const size_t dst = (size_t)(it * NT + wid * 64) * 16;   // wave-uniform byte offset
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)((kv ? Vs : Ks) + dst), 16, 0, 0);
            }
        }
    }
    // number of V pieces THIS wave issued (the youngest operations): waiting until only they remain retires Q and K
    int nv = 0;
#pragma unroll
    for (int it = 0; it < NIT; ++it) nv += (NCH % NT == 0 || it * NT + wid * 64 < NCH) ? 1 : 0;

    // ---- LDS read bases: lane-constant swizzle terms hoisted, every read below is base + immediate ------------------------
    typedef __attribute__((address_space(3))) const char lds_cc;
    lds_cc* kbase[KS];     // K fragment of key tile kt, k-step ks: kbase[ks] + kt * (16 rows)
    lds_cc* vbase[NDT];    // V transposed read of key tile kt, d tile dt: vbase[dt] + kt * (16 rows)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        kbase[ks] = (lds_cc*)Ks + pv_swz<CPR>(i16, ks * 4 + g);
        asm volatile("" : "+v"(kbase[ks]));
    }
    {
        const int tq_ = i16 >> 2, tp_ = i16 & 3;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
            vbase[dt] = (lds_cc*)Vs + pv_swz<CPR>(4 * g + tq_, dt * 2 + (tp_ >> 1)) + ((tp_ & 1) << 3);
            asm volatile("" : "+v"(vbase[dt]));
        }
    }
    PV_ASTAMP(1);
    // K (and Q) landed: all but this wave's nv youngest pieces.  nv is NIT or NIT-1 (wave-uniform): two immediates.
    if (nv == NIT) {
        if (NIT == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else if (NIT == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else if (NIT == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else if (NIT == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (NIT == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else if (NIT == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (NIT == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        if (NIT == 2) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else if (NIT == 3) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else if (NIT == 4) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else if (NIT == 5) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (NIT == 6) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else if (NIT == 7) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    PV_ASTAMP(2);
    bool v_ready = false;

#pragma unroll
    for (int t = 0; t < MAXQT; ++t) {
        const int qt = wid + NW * t;
        if (qt >= nqt) break;
        const int q0 = qt << 4;
        // ---- S^T tiles: sc[kt][r] = score(query q0+i16, key kt*16 + 4g + r) -----------------------------------
        f32x4 sc[NKT];
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                // rows kt*16 + i16: 16 rows = 16*DHP*2 bytes further, and (row & 7) / the line parity are unchanged
                const bf16x8 kf = *reinterpret_cast<const __attribute__((address_space(3))) bf16x8*>(kbase[ks] + kt * (16 * DHP * 2));
                a = PV_MFMA_16x16x32(kf, qf[t][ks], a, 0, 0, 0);
            }
            sc[kt] = a;
        }
        // mask padded keys: SP - 16 < S <= SP, so only the last 16-key tile can hold them
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if ((NKT - 1) * 16 + 4 * g + r >= S) sc[NKT - 1][r] = -INFINITY;
        // ---- softmax numerator: in-lane over 4*NKT keys, then across the 4 lane groups ------------------------
        float m = -INFINITY;
        // Round 5: the maxima below are INLINE ASM, which the hazard recognizer does not look into: hipcc (ROCm 7.2) scheduled them two
        // instructions behind the MFMA that writes their operands (dh = 32, second query tile of a wave: v_mfma v[6:9] ... v_max3 v32, v32, v6, v7
        // three lines later), so a maximum could be formed from what those registers held BEFORE - harmless as long as every exp2(s - m) stays
        // finite, non-finite output rows as soon as a missed score exceeds the stale maximum by more than the packing headroom (fp16 build:
        // 2^6; scripts/dbg/attn_nonfinite.py: 4 of 198 rows at S = 99, dh = 32 on N(0,1) scores, 70 rows at scores ~50; the cause of round 3's
        // "carried maximum" NaN rows).  Fix: every score tile is pinned behind its MFMA by an (empty) volatile asm, then one volatile asm of
        // 20 wait states - more than the 18 the longest MFMA needs before a vector read - through which the running maximum passes: the
        // maxima cannot start before it, and it cannot start before the last MFMA has been issued.
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) asm volatile("" : "+v"(sc[kt]));
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" : "+v"(m));
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {      // v_max3: two scores per instruction, no canonicalising v_max pairs
            asm("v_max3_f32 %0, %0, %1, %2" : "+v"(m) : "v"(sc[kt][0]), "v"(sc[kt][1]));
            asm("v_max3_f32 %0, %0, %1, %2" : "+v"(m) : "v"(sc[kt][2]), "v"(sc[kt][3]));
        }
        m = fmaxf(m, __shfl_xor(m, 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        pv_score_guard(m, flag);
        // p = exp(s - m) = exp2(s*log2e - m*log2e): one FMA + v_exp per score.
        // (+ PV_P_SHIFT: the probabilities are packed as p * 2^10.  The fp16 MFMA path flushes subnormal operands, and with scores spread
        //  over ~17 units more than half of a row's p = exp(s - m) lie below fp16's smallest normal 6.1e-5 - up to 0.7 % of a row's mass;
        //  scaled, everything down to 6e-8 stays normal, and the factor cancels in O / l.  Free: it rides in the FMA's addend.)
        // Measured and NOT kept (round 3, scripts/attn_ab.py, profiles/r03_attention_ab.json): the exp argument as a packed FMA and the
        // row sum as one more MFMA tile (P^T times a tile of ones) take 30 % of the wave's VALU instructions away and not one percent of
        // the kernel's time - it is not VALU-bound, whatever the 54 % VALU issue utilisation suggests (DESIGN.md sections 4 and 12).
        const float nm = -m * 1.44269504088896340736f + PV_P_SHIFT;
        float l = 0.f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pe = __builtin_amdgcn_exp2f(fmaf(sc[kt][r], 1.44269504088896340736f, nm));
                sc[kt][r] = pe;
                l += pe;
            }
        }
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        // training forward (round 5): the row's log-sum-exp in the exp2 domain, m log2(e) + log2(sum exp(s - m)): the persistent backward kernel
        // (pv_attn_bwd5_kernel) forms p = exp2(s log2(e) - lse) without the row maximum and sum.  (l carries 2^PV_P_SHIFT in the fp16 build.)
        if constexpr (LSE)
            if (g == 0 && q0 + i16 < S)
                lse[((int64_t)b * H + h) * S + q0 + i16] = m * 1.44269504088896340736f + (__builtin_amdgcn_logf(l) - PV_P_SHIFT);
        if (!v_ready) {            // first tile of the wave: V must have landed (for every wave) before the first PV product
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            v_ready = true;
        }
        // ---- O^T = V^T . P^T ---------------------------------------------------------------------------------
        f32x4 o[NDT];
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tt = 0; tt < NKT32; ++tt) {
            u32x4 pw = {pv_pack_bf16x2(sc[2 * tt][0], sc[2 * tt][1]), pv_pack_bf16x2(sc[2 * tt][2], sc[2 * tt][3]),
                        pv_pack_bf16x2(sc[2 * tt + 1][0], sc[2 * tt + 1][1]), pv_pack_bf16x2(sc[2 * tt + 1][2], sc[2 * tt + 1][3])};
            const bf16x8 pf = __builtin_bit_cast(bf16x8, pw);
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vbase[dt] + tt * (32 * DHP * 2)));
                s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vbase[dt] + tt * (32 * DHP * 2) + 16 * DHP * 2));
                const s16x8 vv = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
                o[dt] = PV_MFMA_16x16x32(__builtin_bit_cast(bf16x8, vv), pf, o[dt], 0, 0, 0);
            }
        }
        if (NKT & 1) {             // odd last 16-key tile: K = 16 MFMA, lane group g contributes keys 4g..4g+3 directly
            constexpr int kt = NKT - 1;
            u32x2 pw = {pv_pack_bf16x2(sc[kt][0], sc[kt][1]), pv_pack_bf16x2(sc[kt][2], sc[kt][3])};
            const s16x4 pf = __builtin_bit_cast(s16x4, pw);
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(vbase[dt] + kt * (16 * DHP * 2)));
                o[dt] = PV_MFMA_16x16x16(v0, pf, o[dt], 0, 0, 0);
            }
        }
        // ---- normalise and store: lane holds out[q0+i16][h*DH + dt*16 + 4g + 0..3] -----------------------------
        if (q0 + i16 < S) {
            const float inv = 1.0f / l;
            uint16_t* op = out + ((int64_t)b * S + q0 + i16) * D + h * DH + 4 * g;
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                u32x2 ov = {pv_pack_bf16x2(o[dt][0] * inv, o[dt][1] * inv), pv_pack_bf16x2(o[dt][2] * inv, o[dt][3] * inv)};
                *reinterpret_cast<u32x2*>(op + dt * 16) = ov;
            }
        }
    }
    if (!v_ready) {                // a wave without any q tile still has to join the V barrier
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    PV_ASTAMP(3);
}

template <int DH, int NKT>
static int pv_launch_attn(const uint16_t* qkv, uint16_t* out, int64_t B, int S, int H, uint32_t* flag, hipStream_t stream, float* lse = nullptr) {
    constexpr int DHP = (DH + 31) / 32 * 32;
    constexpr int lds = 2 * NKT * 16 * DHP * 2;
    static PvPerDevice attr_set;
    if (attr_set.first_use()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pv_attn_kernel<DH, NKT, false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pv_attn_kernel<DH, NKT, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    }
    // (two instantiations: the inference kernel keeps the code it had before the statistics existed)
    if (lse) PV_LAUNCH((pv_attn_kernel<DH, NKT, true>), dim3((unsigned)(B * H)), dim3(PV_ATTN_NW * 64), lds, stream, qkv, out, S, H, flag, (int)B, lse);
    else PV_LAUNCH((pv_attn_kernel<DH, NKT, false>), dim3((unsigned)(B * H)), dim3(PV_ATTN_NW * 64), lds, stream, qkv, out, S, H, flag, (int)B, lse);
    return pv_check_launch();
}

// ------------------------------------------------------------------------------------------------
// Long sequences (S > 416: larger images / smaller patches than the LDS-resident kernel above can hold): the same swapped
// S^T = K.Q^T / O^T = V^T.P^T products, streamed over 64-key blocks with an online softmax.  One workgroup = 64 queries of one
// (image, head) (a 16-query tile per wave); K and V blocks pass through one 16-KiB LDS buffer.  A lane holds ONE query in every
// accumulator (scores and O^T alike), so the running max / rescale are per-lane scalars; the row sum is kept per lane group and
// combined once at the end.
// ------------------------------------------------------------------------------------------------
template <int DH>
__global__ __launch_bounds__(256) void pv_attn_stream_kernel(const uint16_t* __restrict__ qkv, uint16_t* __restrict__ out, int S, int H, int nqb, uint32_t* flag) {
    constexpr int DHP = (DH + 31) / 32 * 32, CPR = DHP / 8, KS = DHP / 32, NDT = DH / 16, KB = 64;
    __shared__ __attribute__((aligned(16))) char Ks[KB * DHP * 2];
    __shared__ __attribute__((aligned(16))) char Vs[KB * DHP * 2];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int g = lane >> 4, i16 = lane & 15;
    const int qb = blockIdx.x % nqb, bh = blockIdx.x / nqb;
    const int b = bh / H, h = bh - b * H;
    const int D = H * DH;
    const int64_t ld = 3 * (int64_t)D;
    const uint16_t* base = qkv + (int64_t)b * S * ld + h * DH;
    const int q0 = qb * 64 + wid * 16;
    bf16x8 qf[KS];
    {
        int qr = q0 + i16; qr = qr < S ? qr : S - 1;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int dcol = ks * 32 + 8 * g;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (dcol < DH) v = *reinterpret_cast<const u32x4*>(base + (int64_t)qr * ld + dcol);
            qf[ks] = __builtin_bit_cast(bf16x8, v);
        }
    }
    const int tq_ = i16 >> 2, tp_ = i16 & 3;
    int koff[KS], voff[NDT];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) koff[ks] = pv_swz<CPR>(i16, ks * 4 + g);
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) voff[dt] = pv_swz<CPR>(4 * g + tq_, dt * 2 + (tp_ >> 1)) + ((tp_ & 1) << 3);
    f32x4 o[NDT];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m = -INFINITY, l = 0.f;                  // l: this lane group's share of the row sum
    constexpr float LOG2E = 1.44269504088896340736f;
    for (int k0 = 0; k0 < S; k0 += KB) {
        __syncthreads();                           // the previous block has been consumed
        for (int e = tid; e < KB * CPR; e += 256) {
            const int row = e / CPR, c = e - row * CPR;
            int kr = k0 + row; kr = kr < S ? kr : S - 1;
            u32x4 kv = {0u, 0u, 0u, 0u}, vv = kv;
            if (c * 8 < DH) {
                kv = *reinterpret_cast<const u32x4*>(base + (int64_t)kr * ld + D + c * 8);
                vv = *reinterpret_cast<const u32x4*>(base + (int64_t)kr * ld + 2 * D + c * 8);
            }
            const int off = pv_swz<CPR>(row, c);
            *reinterpret_cast<u32x4*>(Ks + off) = kv;
            *reinterpret_cast<u32x4*>(Vs + off) = vv;
        }
        __syncthreads();
        f32x4 sc[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                a = PV_MFMA_16x16x32(*reinterpret_cast<const bf16x8*>(Ks + koff[ks] + kt * (16 * DHP * 2)), qf[ks], a, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (k0 + kt * 16 + 4 * g + r >= S) a[r] = -INFINITY;
            sc[kt] = a;
        }
        float bm = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) bm = fmaxf(fmaxf(bm, fmaxf(sc[kt][0], sc[kt][1])), fmaxf(sc[kt][2], sc[kt][3]));
        bm = fmaxf(bm, __shfl_xor(bm, 16, 64));
        bm = fmaxf(bm, __shfl_xor(bm, 32, 64));
        const float mn = fmaxf(m, bm);             // finite from the first block on (key 0 is never masked)
        const float alpha = __builtin_amdgcn_exp2f((m - mn) * LOG2E);
        const float nm = -mn * LOG2E + PV_P_SHIFT;      // p * 2^PV_P_SHIFT (fp16 build): o and l carry the same factor, it cancels in o / l
        float ps = 0.f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pe = __builtin_amdgcn_exp2f(fmaf(sc[kt][r], LOG2E, nm));
                sc[kt][r] = pe;
                ps += pe;
            }
        l = fmaf(l, alpha, ps);
        m = mn;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) o[dt] = o[dt] * alpha;
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            u32x4 pw = {pv_pack_bf16x2(sc[2 * tt][0], sc[2 * tt][1]), pv_pack_bf16x2(sc[2 * tt][2], sc[2 * tt][3]),
                        pv_pack_bf16x2(sc[2 * tt + 1][0], sc[2 * tt + 1][1]), pv_pack_bf16x2(sc[2 * tt + 1][2], sc[2 * tt + 1][3])};
            const bf16x8 pf = __builtin_bit_cast(bf16x8, pw);
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(Vs + voff[dt] + tt * (32 * DHP * 2)));
                s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(Vs + voff[dt] + tt * (32 * DHP * 2) + 16 * DHP * 2));
                const s16x8 vv = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
                o[dt] = PV_MFMA_16x16x32(__builtin_bit_cast(bf16x8, vv), pf, o[dt], 0, 0, 0);
            }
        }
    }
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    pv_score_guard(m, flag);           // the row's final maximum (every block's maximum has passed through it)
    if (q0 + i16 < S) {
        const float inv = 1.0f / l;
        uint16_t* op = out + ((int64_t)b * S + q0 + i16) * D + h * DH + 4 * g;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
            u32x2 ov = {pv_pack_bf16x2(o[dt][0] * inv, o[dt][1] * inv), pv_pack_bf16x2(o[dt][2] * inv, o[dt][3] * inv)};
            *reinterpret_cast<u32x2*>(op + dt * 16) = ov;
        }
    }
}

template <int DH>
static int pv_launch_attn_stream(const uint16_t* qkv, uint16_t* out, int64_t B, int S, int H, uint32_t* flag, hipStream_t stream) {
    const int nqb = (S + 63) / 64;
    if (B * H * nqb > 0x7fffffff) return PV_ERR_UNSUPPORTED;
    PV_LAUNCH(pv_attn_stream_kernel<DH>, dim3((unsigned)(B * H * nqb)), dim3(256), 0, stream, qkv, out, S, H, nqb, flag);
    return pv_check_launch();
}

template <int DH>
static int pv_dispatch_attn(const uint16_t* qkv, uint16_t* out, int64_t B, int S, int H, uint32_t* flag, hipStream_t stream, float* lse = nullptr) {
    switch ((S + 15) / 16) {
#define PV_ATTN_CASE(N) case N: return pv_launch_attn<DH, N>(qkv, out, B, S, H, flag, stream, lse);
        PV_ATTN_CASE(1) PV_ATTN_CASE(2) PV_ATTN_CASE(3) PV_ATTN_CASE(4) PV_ATTN_CASE(5) PV_ATTN_CASE(6) PV_ATTN_CASE(7)
        PV_ATTN_CASE(8) PV_ATTN_CASE(9) PV_ATTN_CASE(10) PV_ATTN_CASE(11) PV_ATTN_CASE(12) PV_ATTN_CASE(13) PV_ATTN_CASE(14)
        PV_ATTN_CASE(15) PV_ATTN_CASE(16) PV_ATTN_CASE(17) PV_ATTN_CASE(18) PV_ATTN_CASE(19) PV_ATTN_CASE(20) PV_ATTN_CASE(21)
        PV_ATTN_CASE(22) PV_ATTN_CASE(23) PV_ATTN_CASE(24) PV_ATTN_CASE(25) PV_ATTN_CASE(26)
#undef PV_ATTN_CASE
        default: return lse ? PV_ERR_UNSUPPORTED : pv_launch_attn_stream<DH>(qkv, out, B, S, H, flag, stream);      // (the streaming kernel keeps no statistics)
    }
}

// ------------------------------------------------------------------------------------------------
// Precision mode ("bf16x3"): the same attention in exact fp32 on the f32-input MFMA (v_mfma_f32_16x16x4_f32, 1/16 of the
// bf16 rate - attention is 4 % of the model's FLOPs).  qkv fp32 [B,S,3D] -> out bf16 [B*S, 3D] = [hi | lo | hi] planes (the
// split A operand of the out-proj GEMM).  Same swapped S^T = K.Q^T layout: lane (g, q) holds keys 4g+r of a 16-key tile, so
// at PV step s the lane group g contributes key 4g+s on both operands (A = V[key][d], B = its own P register s).
// ------------------------------------------------------------------------------------------------
template <int DH, int NKT>
__global__ __launch_bounds__(256) void pv_attn_f32_kernel(const float* __restrict__ qkv, uint16_t* __restrict__ out, int S, int H) {
    constexpr int RS = DH + 1;            // padded LDS row (floats): breaks the power-of-two stride
    constexpr int SP = NKT * 16;
    constexpr int NDT = DH / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* Ks = reinterpret_cast<float*>(smem);
    float* Vs = Ks + SP * RS;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int g = lane >> 4, i16 = lane & 15;
    const int b = blockIdx.x / H, h = blockIdx.x - b * H;
    const int D = H * DH;
    const int64_t ld = 3 * (int64_t)D;
    const float* qb = qkv + (int64_t)b * S * ld + h * DH;
    for (int e = tid; e < SP * (DH / 4); e += 256) {
        const int row = e / (DH / 4), c4 = (e - row * (DH / 4)) * 4;
        float4 kv = make_float4(0.f, 0.f, 0.f, 0.f), vv = kv;
        if (row < S) {
            kv = *reinterpret_cast<const float4*>(qb + (int64_t)row * ld + D + c4);
            vv = *reinterpret_cast<const float4*>(qb + (int64_t)row * ld + 2 * D + c4);
        }
        float* kd = Ks + row * RS + c4;
        float* vd = Vs + row * RS + c4;
        kd[0] = kv.x; kd[1] = kv.y; kd[2] = kv.z; kd[3] = kv.w;
        vd[0] = vv.x; vd[1] = vv.y; vd[2] = vv.z; vd[3] = vv.w;
    }
    __syncthreads();
    const int nqt = (S + 15) >> 4;
    for (int qt = wid; qt < nqt; qt += 4) {
        const int q0 = qt << 4;
        int qr = q0 + i16; qr = qr < S ? qr : S - 1;
        float qv[DH / 4];                  // B operand: Q[q][4*step + g]
#pragma unroll
        for (int st = 0; st < DH / 4; ++st) qv[st] = qb[(int64_t)qr * ld + 4 * st + g];
        f32x4 sc[NKT];
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int st = 0; st < DH / 4; ++st) a = __builtin_amdgcn_mfma_f32_16x16x4f32(Ks[(kt * 16 + i16) * RS + 4 * st + g], qv[st], a, 0, 0, 0);
            sc[kt] = a;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if ((NKT - 1) * 16 + 4 * g + r >= S) sc[NKT - 1][r] = -INFINITY;
        float m = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) m = fmaxf(fmaxf(m, fmaxf(sc[kt][0], sc[kt][1])), fmaxf(sc[kt][2], sc[kt][3]));
        m = fmaxf(m, __shfl_xor(m, 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float l = 0.f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pe = expf(sc[kt][r] - m);
                sc[kt][r] = pe;
                l += pe;
            }
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        f32x4 o[NDT];
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int st = 0; st < 4; ++st)
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt)
                    o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(Vs[(kt * 16 + 4 * g + st) * RS + dt * 16 + i16], sc[kt][st], o[dt], 0, 0, 0);
        if (q0 + i16 < S) {
            const float inv = 1.0f / l;
            uint16_t* op = out + ((int64_t)b * S + q0 + i16) * 3 * D + h * DH + 4 * g;
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                const PvHiLo s0 = pv_split2(o[dt][0] * inv, o[dt][1] * inv), s1 = pv_split2(o[dt][2] * inv, o[dt][3] * inv);
                *reinterpret_cast<u32x2*>(op + dt * 16) = (u32x2){s0.hi, s1.hi};
                *reinterpret_cast<u32x2*>(op + dt * 16 + D) = (u32x2){s0.lo, s1.lo};
                *reinterpret_cast<u32x2*>(op + dt * 16 + 2 * D) = (u32x2){s0.hi, s1.hi};
            }
        }
    }
}

template <int DH, int NKT>
static int pv_launch_attn_f32(const float* qkv, uint16_t* out, int64_t B, int S, int H, hipStream_t stream) {
    constexpr int lds = 2 * NKT * 16 * (DH + 1) * 4;
    static PvPerDevice attr_set;
    if (attr_set.first_use()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pv_attn_f32_kernel<DH, NKT>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    }
    PV_LAUNCH((pv_attn_f32_kernel<DH, NKT>), dim3((unsigned)(B * H)), dim3(256), lds, stream, qkv, out, S, H);
    return pv_check_launch();
}

template <int DH>
static int pv_dispatch_attn_f32(const float* qkv, uint16_t* out, int64_t B, int S, int H, hipStream_t stream) {
    switch ((S + 15) / 16) {
#define PV_ATTN_CASE(N) case N: return pv_launch_attn_f32<DH, N>(qkv, out, B, S, H, stream);
        PV_ATTN_CASE(1) PV_ATTN_CASE(2) PV_ATTN_CASE(3) PV_ATTN_CASE(4) PV_ATTN_CASE(5) PV_ATTN_CASE(6) PV_ATTN_CASE(7)
        PV_ATTN_CASE(8) PV_ATTN_CASE(9) PV_ATTN_CASE(10) PV_ATTN_CASE(11) PV_ATTN_CASE(12) PV_ATTN_CASE(13)
#undef PV_ATTN_CASE
        default: break;
    }
    if constexpr (DH == 32) {                   // fp32 K and V of one head must fit the LDS: S <= 208 at dh = 64/48, <= 416 at dh = 32
        switch ((S + 15) / 16) {
#define PV_ATTN_CASE(N) case N: return pv_launch_attn_f32<DH, N>(qkv, out, B, S, H, stream);
            PV_ATTN_CASE(14) PV_ATTN_CASE(15) PV_ATTN_CASE(16) PV_ATTN_CASE(17) PV_ATTN_CASE(18) PV_ATTN_CASE(19) PV_ATTN_CASE(20)
            PV_ATTN_CASE(21) PV_ATTN_CASE(22) PV_ATTN_CASE(23) PV_ATTN_CASE(24) PV_ATTN_CASE(25) PV_ATTN_CASE(26)
#undef PV_ATTN_CASE
            default: break;
        }
    }
    return PV_ERR_UNSUPPORTED;
}

// ------------------------------------------------------------------------------------------------
// LOCAL fallback of the attention-score guard (round 5): the same attention with the SCORES computed from split operands.
//   in : qkv fp32 [B,S,3D] (q pre-scaled) - the in-projection of THIS layer computed in split precision (bf16x3 GEMM, PV_EPI_BIAS_F32)
//   out: 16-bit [B*S, D], the operand of the ordinary out-projection GEMM
// The rounding of q and k to 16 bits leaves an error in a score that grows with the score, and the softmax turns it into a relative error of
// the probabilities (PV_SCORE_LIMIT above).  Here q = q_hi + q_lo, k = k_hi + k_lo (two 16-bit halves each: 22 mantissa bits in the fp16 build)
// and S^T = k_hi.q_hi + (k_hi.q_lo + k_lo.q_hi): three MFMA products per score tile instead of one, everything behind the scores - p = exp(s - m)
// packed to 16 bits, O^T = V^T.P^T with v rounded once - exactly pv_attn_kernel's (an error of 2^-11 in p or v is not amplified by anything).
// fp16 build: the lo halves are packed times 2^11 (exact) into an accumulator of their own - unscaled, the lo half of every |q| < 0.125 would
// be an fp16 subnormal, which the MFMA flushes - and the accumulators are joined by one FMA per score.
// One workgroup (4 waves) per (image, head), K_hi | K_lo | V images in LDS (80 KiB at S = 197, dh = 64: two workgroups per CU), staged through
// registers (the split has to happen on the way); rows >= S duplicate row S - 1 and are masked like pv_attn_kernel's.
// ------------------------------------------------------------------------------------------------
#ifdef PV_OPERAND_F16
#define PV_LO_SCALE 2048.0f
#else
#define PV_LO_SCALE 1.0f
#endif
__device__ __forceinline__ void pv_split8(const float4 a, const float4 b, u32x4& hi, u32x4& lo) {
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t h = pv_pack_bf16x2(v[2 * i], v[2 * i + 1]);
        hi[i] = h;
        lo[i] = pv_pack_bf16x2((v[2 * i] - pv_unpack_lo(h)) * PV_LO_SCALE, (v[2 * i + 1] - pv_unpack_hi(h)) * PV_LO_SCALE);
    }
}

template <int DH, int NKT>
__global__ __launch_bounds__(256, 2) void pv_attn_split_kernel(const float* __restrict__ qkv, uint16_t* __restrict__ out, int S, int H, int B) {
    constexpr int DHP = (DH + 31) / 32 * 32, CPR = DHP / 8, KS = DHP / 32, NKT32 = NKT / 2, SP = NKT * 16, NDT = DH / 16, IMG = SP * DHP * 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const Kh = smem;
    char* const Kl = smem + IMG;
    char* const Vs = smem + 2 * IMG;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, i16 = lane & 15;
    int b, h;
    pv_bh_map(blockIdx.x, B, H, b, h);
    const int D = H * DH;
    const int64_t ld = 3 * (int64_t)D;
    const float* qb = qkv + (int64_t)b * S * ld + h * DH;

    // ---- stage K (split) and V: one 16-byte operand chunk (8 values) per thread and trip ----
    for (int e = tid; e < SP * CPR; e += 256) {
        const int row = e / CPR, c = e - row * CPR;
        const int r = row < S ? row : S - 1;
        u32x4 kh = {0u, 0u, 0u, 0u}, kl = kh, vv = kh;
        if (c * 8 < DH) {
            const float* rp = qb + (int64_t)r * ld + c * 8;
            const float4 k0 = *reinterpret_cast<const float4*>(rp + D), k1 = *reinterpret_cast<const float4*>(rp + D + 4);
            const float4 v0 = *reinterpret_cast<const float4*>(rp + 2 * D), v1 = *reinterpret_cast<const float4*>(rp + 2 * D + 4);
            pv_split8(k0, k1, kh, kl);
            vv = (u32x4){pv_pack_bf16x2(v0.x, v0.y), pv_pack_bf16x2(v0.z, v0.w), pv_pack_bf16x2(v1.x, v1.y), pv_pack_bf16x2(v1.z, v1.w)};
        }
        const int off = pv_swz<CPR>(row, c);
        *reinterpret_cast<u32x4*>(Kh + off) = kh;
        *reinterpret_cast<u32x4*>(Kl + off) = kl;
        *reinterpret_cast<u32x4*>(Vs + off) = vv;
    }
    // Q of a 16-query tile: lane (g, i16) holds Q[q0 + i16][ks*32 + 8g .. +8] as hi / lo fragments
    auto qload = [&](int qt, float4 (&raw)[KS][2]) __attribute__((always_inline)) {
        int qr = qt * 16 + i16;
        qr = qr < S ? qr : S - 1;
        const float* qp = qb + (int64_t)qr * ld;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int dcol = ks * 32 + 8 * g;
            raw[ks][0] = make_float4(0.f, 0.f, 0.f, 0.f);
            raw[ks][1] = raw[ks][0];
            if (dcol < DH) { raw[ks][0] = *reinterpret_cast<const float4*>(qp + dcol); raw[ks][1] = *reinterpret_cast<const float4*>(qp + dcol + 4); }
        }
    };
    const int nqt = (S + 15) >> 4;
    float4 raw[KS][2];
    if (wid < nqt) qload(wid, raw);
    typedef __attribute__((address_space(3))) const char lds_cc;
    int koff[KS], voff[NDT];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) koff[ks] = pv_swz<CPR>(i16, ks * 4 + g);
    {
        const int tq_ = i16 >> 2, tp_ = i16 & 3;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) voff[dt] = pv_swz<CPR>(4 * g + tq_, dt * 2 + (tp_ >> 1)) + ((tp_ & 1) << 3);
    }
    __syncthreads();

    for (int qt = wid; qt < nqt; qt += 4) {
        const int q0 = qt << 4;
        bf16x8 qh[KS], ql[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            u32x4 hi, lo;
            pv_split8(raw[ks][0], raw[ks][1], hi, lo);
            qh[ks] = __builtin_bit_cast(bf16x8, hi);
            ql[ks] = __builtin_bit_cast(bf16x8, lo);
        }
        if (qt + 4 < nqt) qload(qt + 4, raw);            // the next tile's rows arrive under this tile's products
        f32x4 sc[NKT];
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            f32x4 ah = {0.f, 0.f, 0.f, 0.f}, al = ah;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8 kh = *reinterpret_cast<const bf16x8*>(Kh + koff[ks] + kt * (16 * DHP * 2));
                const bf16x8 kl = *reinterpret_cast<const bf16x8*>(Kl + koff[ks] + kt * (16 * DHP * 2));
                ah = PV_MFMA_16x16x32(kh, qh[ks], ah, 0, 0, 0);
                al = PV_MFMA_16x16x32(kh, ql[ks], al, 0, 0, 0);
                al = PV_MFMA_16x16x32(kl, qh[ks], al, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) sc[kt][r] = fmaf(al[r], 1.0f / PV_LO_SCALE, ah[r]);
            // (without this hipcc hoists the K fragments of every key tile to the top of the loop: 256 registers and 70 spilled at 13 tiles)
            if (kt & 1) __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if ((NKT - 1) * 16 + 4 * g + r >= S) sc[NKT - 1][r] = -INFINITY;
        float m = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) m = fmaxf(fmaxf(m, fmaxf(sc[kt][0], sc[kt][1])), fmaxf(sc[kt][2], sc[kt][3]));
        m = fmaxf(m, __shfl_xor(m, 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        const float nm = -m * 1.44269504088896340736f + PV_P_SHIFT;
        float l = 0.f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pe = __builtin_amdgcn_exp2f(fmaf(sc[kt][r], 1.44269504088896340736f, nm));
                sc[kt][r] = pe;
                l += pe;
            }
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        f32x4 o[NDT];
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tt = 0; tt < NKT32; ++tt) {
            u32x4 pw = {pv_pack_bf16x2(sc[2 * tt][0], sc[2 * tt][1]), pv_pack_bf16x2(sc[2 * tt][2], sc[2 * tt][3]),
                        pv_pack_bf16x2(sc[2 * tt + 1][0], sc[2 * tt + 1][1]), pv_pack_bf16x2(sc[2 * tt + 1][2], sc[2 * tt + 1][3])};
            const bf16x8 pf = __builtin_bit_cast(bf16x8, pw);
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(Vs + voff[dt] + tt * (32 * DHP * 2)));
                s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(Vs + voff[dt] + tt * (32 * DHP * 2) + 16 * DHP * 2));
                const s16x8 vv = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
                o[dt] = PV_MFMA_16x16x32(__builtin_bit_cast(bf16x8, vv), pf, o[dt], 0, 0, 0);
            }
        }
        if (NKT & 1) {
            constexpr int kt = NKT - 1;
            u32x2 pw = {pv_pack_bf16x2(sc[kt][0], sc[kt][1]), pv_pack_bf16x2(sc[kt][2], sc[kt][3])};
            const s16x4 pf = __builtin_bit_cast(s16x4, pw);
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(Vs + voff[dt] + kt * (16 * DHP * 2)));
                o[dt] = PV_MFMA_16x16x16(v0, pf, o[dt], 0, 0, 0);
            }
        }
        if (q0 + i16 < S) {
            const float inv = 1.0f / l;
            uint16_t* op = out + ((int64_t)b * S + q0 + i16) * D + h * DH + 4 * g;
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                u32x2 ov = {pv_pack_bf16x2(o[dt][0] * inv, o[dt][1] * inv), pv_pack_bf16x2(o[dt][2] * inv, o[dt][3] * inv)};
                *reinterpret_cast<u32x2*>(op + dt * 16) = ov;
            }
        }
    }
}

template <int DH, int NKT>
static int pv_launch_attn_split(const float* qkv, uint16_t* out, int64_t B, int S, int H, hipStream_t stream) {
    constexpr int DHP = (DH + 31) / 32 * 32;
    constexpr int lds = 3 * NKT * 16 * DHP * 2;
    static_assert(lds <= 160 * 1024, "K_hi | K_lo | V of one head must fit the LDS");
    static PvPerDevice attr_set;
    if (attr_set.first_use()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pv_attn_split_kernel<DH, NKT>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    }
    PV_LAUNCH((pv_attn_split_kernel<DH, NKT>), dim3((unsigned)(B * H)), dim3(256), lds, stream, qkv, out, S, H, (int)B);
    return pv_check_launch();
}

template <int DH>
static int pv_dispatch_attn_split(const float* qkv, uint16_t* out, int64_t B, int S, int H, hipStream_t stream) {
    switch ((S + 15) / 16) {
#define PV_ATTN_CASE(N) case N: return pv_launch_attn_split<DH, N>(qkv, out, B, S, H, stream);
        PV_ATTN_CASE(1) PV_ATTN_CASE(2) PV_ATTN_CASE(3) PV_ATTN_CASE(4) PV_ATTN_CASE(5) PV_ATTN_CASE(6) PV_ATTN_CASE(7)
        PV_ATTN_CASE(8) PV_ATTN_CASE(9) PV_ATTN_CASE(10) PV_ATTN_CASE(11) PV_ATTN_CASE(12) PV_ATTN_CASE(13)
#undef PV_ATTN_CASE
        default: break;
    }
    if constexpr (DH == 32) {
        switch ((S + 15) / 16) {
#define PV_ATTN_CASE(N) case N: return pv_launch_attn_split<DH, N>(qkv, out, B, S, H, stream);
            PV_ATTN_CASE(14) PV_ATTN_CASE(15) PV_ATTN_CASE(16) PV_ATTN_CASE(17) PV_ATTN_CASE(18) PV_ATTN_CASE(19) PV_ATTN_CASE(20)
            PV_ATTN_CASE(21) PV_ATTN_CASE(22) PV_ATTN_CASE(23) PV_ATTN_CASE(24) PV_ATTN_CASE(25) PV_ATTN_CASE(26)
#undef PV_ATTN_CASE
            default: break;
        }
    }
    return PV_ERR_UNSUPPORTED;
}

extern "C" int pv_attention_split_bf16(const float* qkv, uint16_t* out, int64_t B, int64_t S, int64_t H, int64_t dh, void* stream) {
    if (!qkv || !out || B <= 0 || S <= 0 || H <= 0 || dh <= 0) return PV_ERR_INVALID_ARG;
    if (((uintptr_t)qkv & 15) || ((uintptr_t)out & 15)) return PV_ERR_INVALID_ARG;
    if (B * H > 0x7fffffff) return PV_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    switch (dh) {          // S <= 208 at dh = 48 / 64, S <= 416 at dh = 32 (three 16-bit images of the head in LDS)
        case 32: return pv_dispatch_attn_split<32>(qkv, out, B, (int)S, (int)H, s);
        case 48: return pv_dispatch_attn_split<48>(qkv, out, B, (int)S, (int)H, s);
        case 64: return pv_dispatch_attn_split<64>(qkv, out, B, (int)S, (int)H, s);
        default: return PV_ERR_UNSUPPORTED;
    }
}

// ------------------------------------------------------------------------------------------------
// Attention backward (train/train.py:118 loss.backward() through models/blocks.py:32-37); dh in {32, 48, 64}, S <= 208 (416 at dh = 32).
//   in : qkv bf16 [B,S,3D] (q pre-scaled by qscale, as the forward wrote it), dout bf16 [B,S,D] = dL/d(attention output)
//   out: dqkv bf16 [B,S,3D] = dL/d(in-proj output BEFORE the q pre-scale)  (the A operand of the in-proj dgrad / wgrad GEMMs)
// One workgroup (8 waves) per (image, head); Q, K, V, dO of the head live in LDS (swizzled row-major images).  Probabilities
// are recomputed (nothing but qkv and the output gradient is read).  Two wave-private passes, no atomics:
//   pass 1, wave owns a 16-query tile, swapped S^T / dP^T = K.Q^T / V.dO^T tiles (lane = 4 keys of one query): softmax row
//           statistics m, 1/l and D = rowsum(P o dP), dS = P o (dP - D), dQ^T = K^T . dS^T (A = K^T by transposed LDS reads,
//           B = the bf16-packed dS registers, K=16 MFMA); statistics go to LDS.
//   pass 2, wave owns a 16-key tile, S / dP = Q.K^T / dO.V^T tiles (lane = 4 queries of one key) recomputed with the saved
//           statistics: dV^T = dO^T . P, dK^T = Q^T . dS (A by transposed LDS reads of dO and Q, B from registers).
// ------------------------------------------------------------------------------------------------
template <int DH, int NKT, int NW>     // NW waves per workgroup: 8, or 4 when the 2*NKT accumulator tiles need the full register file
__global__ __launch_bounds__(NW * 64) void pv_attn_bwd_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ dout,
                                                          uint16_t* __restrict__ dqkv, float* __restrict__ dbp, int S, int H, float qscale) {
    constexpr int DHP = (DH + 31) / 32 * 32, CPR = DHP / 8, TB = 16 * DHP * 2;      // TB: bytes of 16 LDS rows
    constexpr int SP = NKT * 16, IMG = SP * DHP * 2, NDT = DH / 16, KS = DHP / 32;
    constexpr float LOG2E = 1.44269504088896340736f;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const Qs = smem;
    char* const Ks = smem + IMG;
    char* const Vs = smem + 2 * IMG;
    char* const Os = smem + 3 * IMG;
    float* const st_m = reinterpret_cast<float*>(smem + 4 * IMG);     // per query: log2(1/l) - m log2(e), and D
    float* const st_d = st_m + SP;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, i16 = lane & 15;
    const int b = blockIdx.x / H, h = blockIdx.x - b * H;
    const int D = H * DH;
    const int64_t ld = 3 * (int64_t)D;
    const uint16_t* qb = qkv + (int64_t)b * S * ld + h * DH;
    const uint16_t* ob = dout + (int64_t)b * S * D + h * DH;
    uint16_t* gb = dqkv + (int64_t)b * S * ld + h * DH;

    // ---- stage the four images; rows >= S are zero: zero Q / dO rows make every padded-query contribution vanish ------
    {
        for (int e = tid; e < SP * CPR; e += NW * 64) {
            const int row = e / CPR, c = e - row * CPR;
            u32x4 q = {0u, 0u, 0u, 0u}, k = q, v = q, o = q;
            if (row < S && c * 8 < DH) {
                const uint16_t* rp = qb + (int64_t)row * ld + c * 8;
                q = *reinterpret_cast<const u32x4*>(rp);
                k = *reinterpret_cast<const u32x4*>(rp + D);
                v = *reinterpret_cast<const u32x4*>(rp + 2 * D);
                o = *reinterpret_cast<const u32x4*>(ob + (int64_t)row * D + c * 8);
            }
            const int off = pv_swz<CPR>(row, c);
            *reinterpret_cast<u32x4*>(Qs + off) = q;
            *reinterpret_cast<u32x4*>(Ks + off) = k;
            *reinterpret_cast<u32x4*>(Vs + off) = v;
            *reinterpret_cast<u32x4*>(Os + off) = o;
        }
    }
    __syncthreads();

    // lane-constant LDS offsets: plain fragment X[tile*16 + i16][ks*32 + 8g ..+8] and transposed fragment
    // X[tile*16 + 4g + j][dt*16 + i16] (j = 0..3); 16 rows further = + TB bytes with the same swizzle term
    int foff[KS], toff[NDT];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) foff[ks] = pv_swz<CPR>(i16, ks * 4 + g);
    {
        const int tq_ = i16 >> 2, tp_ = i16 & 3;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) toff[dt] = pv_swz<CPR>(4 * g + tq_, dt * 2 + (tp_ >> 1)) + ((tp_ & 1) << 3);
    }
    auto frag = [&](const char* X, int tile, int ks) __attribute__((always_inline)) {
        return *reinterpret_cast<const bf16x8*>(X + foff[ks] + tile * TB);
    };
    auto tfrag = [&](const char* X, int tile, int dt) __attribute__((always_inline)) {
        return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(X + toff[dt] + tile * TB));
    };
    const int nqt = (S + 15) >> 4;
    // column sums of the stored (operand-rounded) gradient rows of this (image, head): the in-proj bias gradient, per lane here
    f32x4 cq[NDT], ck[NDT], cv[NDT];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) { cq[dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; ck[dt] = cq[dt]; cv[dt] = cq[dt]; }

    // =============================== pass 1: per 16-query tile ===============================
    for (int qt = wid; qt < nqt; qt += NW) {
        const int q0 = qt << 4;
        bf16x8 qf[KS], of[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) { qf[ks] = frag(Qs, qt, ks); of[ks] = frag(Os, qt, ks); }
        f32x4 sc[NKT], dp[NKT];
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f}, c = a;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                a = PV_MFMA_16x16x32(frag(Ks, kt, ks), qf[ks], a, 0, 0, 0);
                c = PV_MFMA_16x16x32(frag(Vs, kt, ks), of[ks], c, 0, 0, 0);
            }
            sc[kt] = a; dp[kt] = c;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if ((NKT - 1) * 16 + 4 * g + r >= S) sc[NKT - 1][r] = -INFINITY;
        float m = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) m = fmaxf(fmaxf(m, fmaxf(sc[kt][0], sc[kt][1])), fmaxf(sc[kt][2], sc[kt][3]));
        m = fmaxf(m, __shfl_xor(m, 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        const float nm = -m * LOG2E;
        float l = 0.f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pe = __builtin_amdgcn_exp2f(fmaf(sc[kt][r], LOG2E, nm));
                sc[kt][r] = pe;
                l += pe;
            }
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        const float inv = 1.0f / l;
        float dd = 0.f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                sc[kt][r] *= inv;                                  // P
                dd = fmaf(sc[kt][r], dp[kt][r], dd);
            }
        dd += __shfl_xor(dd, 16, 64);
        dd += __shfl_xor(dd, 32, 64);
        // one statistic per query for pass 2, log2(1/l) - m log2(e): p = exp2(s log2(e) + that), no second LDS read and no multiply per element
        if (g == 0) { st_m[q0 + i16] = nm + __builtin_amdgcn_logf(inv); st_d[q0 + i16] = dd; }
        f32x4 dq[NDT];
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) dq[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // pairs of key tiles feed the K = 32 MFMA (twice the rate of the K = 16 form): slot j < 4 of lane group g = key 32t + 4g + j,
        // slot j >= 4 = key 32t + 16 + 4g + j - 4, the same order on both operands; an odd last tile uses the K = 16 MFMA
        auto ds_pack = [&](int kt) __attribute__((always_inline)) {
            return (u32x2){pv_pack_bf16x2(sc[kt][0] * (dp[kt][0] - dd), sc[kt][1] * (dp[kt][1] - dd)),
                           pv_pack_bf16x2(sc[kt][2] * (dp[kt][2] - dd), sc[kt][3] * (dp[kt][3] - dd))};
        };
#pragma unroll
        for (int tt = 0; tt < NKT / 2; ++tt) {
            const u32x2 d0 = ds_pack(2 * tt), d1 = ds_pack(2 * tt + 1);
            const bf16x8 dsf = __builtin_bit_cast(bf16x8, (u32x4){d0[0], d0[1], d1[0], d1[1]});
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                const s16x8 kk = __builtin_shufflevector(tfrag(Ks, 2 * tt, dt), tfrag(Ks, 2 * tt + 1, dt), 0, 1, 2, 3, 4, 5, 6, 7);
                dq[dt] = PV_MFMA_16x16x32(__builtin_bit_cast(bf16x8, kk), dsf, dq[dt], 0, 0, 0);
            }
        }
        if (NKT & 1) {
            const s16x4 dsf = __builtin_bit_cast(s16x4, ds_pack(NKT - 1));
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) dq[dt] = PV_MFMA_16x16x16(tfrag(Ks, NKT - 1, dt), dsf, dq[dt], 0, 0, 0);
        }
        if (q0 + i16 < S) {       // dq[dt][r] = dL/dq'[q0+i16][dt*16 + 4g + r]; the in-proj output is q'/qscale
            uint16_t* op = gb + (int64_t)(q0 + i16) * ld + 4 * g;
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                const u32x2 ov = {pv_pack_bf16x2(dq[dt][0] * qscale, dq[dt][1] * qscale), pv_pack_bf16x2(dq[dt][2] * qscale, dq[dt][3] * qscale)};
                *reinterpret_cast<u32x2*>(op + dt * 16) = ov;
                cq[dt] += (f32x4){pv_unpack_lo(ov[0]), pv_unpack_hi(ov[0]), pv_unpack_lo(ov[1]), pv_unpack_hi(ov[1])};
            }
        }
    }
    __syncthreads();

    // =============================== pass 2: per unit of KPW 16-key tiles ===============================
    // Round 3: a wave owns TWO key tiles when that turns two rounds into one (9 <= NKT <= 16 with 8 waves: 13 tiles = 7 units; it was 8 + 5
    // tiles = two rounds at 81 %).  Both passes are paced by LDS reads (r2 analysis in DESIGN.md section 9), and everything a unit reads per
    // q-tile pair - the Q and dO fragments for S / dP, their transposed fragments for dK / dV, the row statistics - is read ONCE for both key
    // tiles: the MFMAs double per LDS byte.  Arithmetic per element is unchanged (same products, same accumulation order over the q tiles).
#ifndef PV_ABW_KPW
#define PV_ABW_KPW ((NW == 8 && NKT >= 9 && NKT <= 16) ? 2 : 1)
#endif
    constexpr int KPW = PV_ABW_KPW;
    for (int ku = wid; ku * KPW < nqt; ku += NW) {
        bf16x8 kf[KPW][KS], vf[KPW][KS];
        bool key_ok[KPW];
        int k0[KPW];
#pragma unroll
        for (int t = 0; t < KPW; ++t) {
            const int kt = ku * KPW + t, ktc = kt < NKT ? kt : NKT - 1;       // a unit's second tile may lie past the last one: no valid key
            k0[t] = kt << 4;
            key_ok[t] = kt < nqt && k0[t] + i16 < S;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) { kf[t][ks] = frag(Ks, ktc, ks); vf[t][ks] = frag(Vs, ktc, ks); }
        }
        f32x4 dv[KPW][NDT], dk[KPW][NDT];
#pragma unroll
        for (int t = 0; t < KPW; ++t)
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) { dv[t][dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; dk[t][dt] = dv[t][dt]; }
        // probabilities / dS of one q tile against the unit's key tiles from the saved row statistics: p[r], ds[r] for query qt*16 + 4g + r
        auto pds = [&](int qt, u32x2 (&pw)[KPW], u32x2 (&dw)[KPW]) __attribute__((always_inline)) {
            bf16x8 qq_[KS], oo_[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) { qq_[ks] = frag(Qs, qt, ks); oo_[ks] = frag(Os, qt, ks); }
            const float4 m4 = *reinterpret_cast<const float4*>(st_m + qt * 16 + 4 * g);
            const float4 d4 = *reinterpret_cast<const float4*>(st_d + qt * 16 + 4 * g);
            // fp16 build (round 5, fp16 training): P is packed as p * 2^PV_P_SHIFT like the forward kernel's (the fp16 MFMA flushes subnormal
            // operands: unshifted, every p < 6.1e-5 would vanish from dV = dO^T . P); dS = P o (dP - D) keeps its own magnitude - the
            // factor is taken out again inside the bracket, (dP - D) * 2^-PV_P_SHIFT as one FMA against the pre-scaled D - and dV is
            // multiplied by 2^-PV_P_SHIFT (exact) where it is stored.  PV_P_SHIFT = 0 (bf16 build): the arithmetic of rounds 1-4, bit for bit.
#ifdef PV_OPERAND_F16
            const float mm[4] = {m4.x + PV_P_SHIFT, m4.y + PV_P_SHIFT, m4.z + PV_P_SHIFT, m4.w + PV_P_SHIFT};
            const float dd[4] = {d4.x * PV_P_UNSHIFT, d4.y * PV_P_UNSHIFT, d4.z * PV_P_UNSHIFT, d4.w * PV_P_UNSHIFT};
#else
            const float mm[4] = {m4.x, m4.y, m4.z, m4.w}, dd[4] = {d4.x, d4.y, d4.z, d4.w};
#endif
#pragma unroll
            for (int t = 0; t < KPW; ++t) {
                f32x4 s = {0.f, 0.f, 0.f, 0.f}, c = s;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    s = PV_MFMA_16x16x32(qq_[ks], kf[t][ks], s, 0, 0, 0);
                    c = PV_MFMA_16x16x32(oo_[ks], vf[t][ks], c, 0, 0, 0);
                }
                float p[4], ds[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    p[r] = key_ok[t] ? __builtin_amdgcn_exp2f(fmaf(s[r], LOG2E, mm[r])) : 0.f;
#ifdef PV_OPERAND_F16
                    ds[r] = p[r] * fmaf(c[r], PV_P_UNSHIFT, -dd[r]);
#else
                    ds[r] = p[r] * (c[r] - dd[r]);
#endif
                }
                pw[t] = (u32x2){pv_pack_bf16x2(p[0], p[1]), pv_pack_bf16x2(p[2], p[3])};
                dw[t] = (u32x2){pv_pack_bf16x2(ds[0], ds[1]), pv_pack_bf16x2(ds[2], ds[3])};
            }
        };
        // q tiles in pairs on the K = 32 MFMA (same slot order on both operands as in pass 1), an odd last tile on the K = 16 form
        // one key tile per wave: three pairs per iteration at S = 193..208 (two iterations): the LDS reads of the next pair are issued under the
        // MFMAs of this one (2.28 -> 2.13 ms at ViT-B/16, batch 2048); shorter sequences spill with it (NKT = 7: 0.93 -> 2.07 ms) and keep the rolled loop
#ifndef PV_ABW_P2_UNROLL
#define PV_ABW_P2_UNROLL (KPW == 1 ? (NKT == 13 ? 3 : 1) : 1)
#endif
#pragma unroll PV_ABW_P2_UNROLL
        for (int tt = 0; tt < NKT / 2; ++tt) {
            u32x2 p0[KPW], d0[KPW], p1[KPW], d1[KPW];
            pds(2 * tt, p0, d0);
            pds(2 * tt + 1, p1, d1);
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                const s16x8 oo = __builtin_shufflevector(tfrag(Os, 2 * tt, dt), tfrag(Os, 2 * tt + 1, dt), 0, 1, 2, 3, 4, 5, 6, 7);
                const s16x8 qq = __builtin_shufflevector(tfrag(Qs, 2 * tt, dt), tfrag(Qs, 2 * tt + 1, dt), 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                for (int t = 0; t < KPW; ++t) {
                    const bf16x8 pf = __builtin_bit_cast(bf16x8, (u32x4){p0[t][0], p0[t][1], p1[t][0], p1[t][1]});
                    const bf16x8 dsf = __builtin_bit_cast(bf16x8, (u32x4){d0[t][0], d0[t][1], d1[t][0], d1[t][1]});
                    dv[t][dt] = PV_MFMA_16x16x32(__builtin_bit_cast(bf16x8, oo), pf, dv[t][dt], 0, 0, 0);
                    dk[t][dt] = PV_MFMA_16x16x32(__builtin_bit_cast(bf16x8, qq), dsf, dk[t][dt], 0, 0, 0);
                }
            }
        }
        if (NKT & 1) {
            u32x2 pw[KPW], dw[KPW];
            pds(NKT - 1, pw, dw);
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                const s16x4 oo = tfrag(Os, NKT - 1, dt), qq = tfrag(Qs, NKT - 1, dt);
#pragma unroll
                for (int t = 0; t < KPW; ++t) {
                    dv[t][dt] = PV_MFMA_16x16x16(oo, __builtin_bit_cast(s16x4, pw[t]), dv[t][dt], 0, 0, 0);
                    dk[t][dt] = PV_MFMA_16x16x16(qq, __builtin_bit_cast(s16x4, dw[t]), dk[t][dt], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < KPW; ++t)
            if (key_ok[t]) {             // d*[dt][r] = dL/d{k,v}[k0+i16][dt*16 + 4g + r]
                uint16_t* op = gb + (int64_t)(k0[t] + i16) * ld + 4 * g;
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt) {
                    const u32x2 kv = {pv_pack_bf16x2(dk[t][dt][0], dk[t][dt][1]), pv_pack_bf16x2(dk[t][dt][2], dk[t][dt][3])};
#ifdef PV_OPERAND_F16
                    dv[t][dt] = dv[t][dt] * PV_P_UNSHIFT;
#endif
                    const u32x2 vv = {pv_pack_bf16x2(dv[t][dt][0], dv[t][dt][1]), pv_pack_bf16x2(dv[t][dt][2], dv[t][dt][3])};
                    *reinterpret_cast<u32x2*>(op + D + dt * 16) = kv;
                    *reinterpret_cast<u32x2*>(op + 2 * D + dt * 16) = vv;
                    ck[dt] += (f32x4){pv_unpack_lo(kv[0]), pv_unpack_hi(kv[0]), pv_unpack_lo(kv[1]), pv_unpack_hi(kv[1])};
                    cv[dt] += (f32x4){pv_unpack_lo(vv[0]), pv_unpack_hi(vv[0]), pv_unpack_lo(vv[1]), pv_unpack_hi(vv[1])};
                }
            }
    }
    if (dbp) {        // (workgroup-uniform) sum over the 16 rows a lane group holds, then over the waves through LDS
        __syncthreads();                                  // every wave has left pass 2: the Q image is free
        float* red = reinterpret_cast<float*>(smem);      // [NW][3][DH]
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float a = pv_row16_sum(cq[dt][r]), bsum = pv_row16_sum(ck[dt][r]), c = pv_row16_sum(cv[dt][r]);
                if (i16 == 0) {
                    const int col = dt * 16 + 4 * g + r;
                    red[(wid * 3 + 0) * DH + col] = a; red[(wid * 3 + 1) * DH + col] = bsum; red[(wid * 3 + 2) * DH + col] = c;
                }
            }
        __syncthreads();
        for (int e = tid; e < 3 * DH; e += NW * 64) {
            const int part = e / DH, col = e - part * DH;
            float t = 0.f;
            for (int w = 0; w < NW; ++w) t += red[(w * 3 + part) * DH + col];
            dbp[(int64_t)b * 3 * D + part * D + h * DH + col] = t;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Attention backward, round 5 form (`pv_attn_bwd2_kernel`, PV_ABW_V2): the same two wave-private passes and the same arithmetic per element as
// pv_attn_bwd_kernel above, restaged so that TWO (three for S <= 192) workgroups share a CU and one's staging / stores run under the other's
// MFMAs.  The round 1-4 kernel held Q, K, V and dO of its head in LDS at once (106 KiB at S = 197, dh = 64: ONE workgroup per CU) and
// staged them through registers before anything else ran: per workgroup 21.5 us against 5.6 us of MFMA issue - each CU serialised load ->
// compute -> store (rocprofv3, round 4: mfma_busy 0.21).  Here a workgroup is 4 waves and its LDS holds two images at a time:
//   phase 1  K | V by LDS-DMA (the forward kernel's lane-linear swizzled images); pass 1 takes its Q / dO fragments straight from global
//            memory (a 16-query tile per wave, like the forward kernel's Q);
//   phase 2  Q | dO by LDS-DMA into the SAME space, issued behind the barrier that ends pass 1; pass 2 takes its K / V fragments (the
//            wave's own key tiles) straight from global memory.
// Rows >= S of an image duplicate row S - 1 (LDS-DMA cannot write zeros): padded KEYS are masked in both passes (score -inf / key_ok),
// padded QUERIES get the statistic -inf, i.e. p = 0, so their (finite, duplicate) rows contribute exact zeros.  dh = 48: the pad chunk
// of an image row duplicates chunk 0; every product over the padded columns has a zero on its register side.
// ------------------------------------------------------------------------------------------------
#ifndef PV_ABW_V2
#define PV_ABW_V2 1
#endif
template <int DH, int NKT, int WGS>     // WGS: workgroups per CU the register budget is set for (launch bounds)
__global__ __launch_bounds__(256, WGS) void pv_attn_bwd2_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ dout,
                                                                  uint16_t* __restrict__ dqkv, float* __restrict__ dbp, int S, int H, float qscale, int B) {
    constexpr int NW = 4;
    constexpr int DHP = (DH + 31) / 32 * 32, CPR = DHP / 8, TB = 16 * DHP * 2;
    constexpr int SP = NKT * 16, IMG = SP * DHP * 2, NDT = DH / 16, KS = DHP / 32;
    constexpr int NCH = SP * CPR, NIT = (NCH + 255) / 256;
    constexpr float LOG2E = 1.44269504088896340736f;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const A_ = smem;                 // phase 1: K, phase 2: Q
    char* const B_ = smem + IMG;           // phase 1: V, phase 2: dO
    float* const st_m = reinterpret_cast<float*>(smem + 2 * IMG);
    float* const st_d = st_m + SP;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, i16 = lane & 15;
    int b, h;
    pv_bh_map(blockIdx.x, B, H, b, h);
    const int D = H * DH;
    const int64_t ld = 3 * (int64_t)D;
    const uint16_t* qb = qkv + (int64_t)b * S * ld + h * DH;
    const uint16_t* ob = dout + (int64_t)b * S * D + h * DH;
    uint16_t* gb = dqkv + (int64_t)b * S * ld + h * DH;

    // ---- LDS-DMA staging of two row-major [S, DH] matrices as swizzled images (pv_attn_kernel's scheme: physical chunk it*256 + tid) ----
    const int lsw = (tid & 7) ^ ((tid >> 3) & 7);
    const int r_lane = CPR == 8 ? (tid >> 3) : 2 * (tid >> 3) + (lsw >> 2);
    int c_lane = CPR == 8 ? lsw : (lsw & 3);
    if (c_lane * 8 >= DH) c_lane = 0;
    auto stage = [&](const uint16_t* a, int64_t lda_, const uint16_t* bsrc, int64_t ldb_) __attribute__((always_inline)) {
#pragma unroll
        for (int which = 0; which < 2; ++which) {
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                if (NCH % 256 == 0 || it * 256 + wid * 64 < NCH) {
                    int row = it * (256 / CPR) + r_lane;
                    row = row < S ? row : S - 1;
                    const uint16_t* src = (which ? bsrc + (int64_t)row * ldb_ : a + (int64_t)row * lda_) + c_lane * 8;
                    const size_t dst = (size_t)(it * 256 + wid * 64) * 16;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                     (__attribute__((address_space(3))) void*)((which ? B_ : A_) + dst), 16, 0, 0);
                }
            }
        }
    };
    stage(qb + D, ld, qb + 2 * D, ld);                         // K | V

    int foff[KS], toff[NDT];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) foff[ks] = pv_swz<CPR>(i16, ks * 4 + g);
    {
        const int tq_ = i16 >> 2, tp_ = i16 & 3;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) toff[dt] = pv_swz<CPR>(4 * g + tq_, dt * 2 + (tp_ >> 1)) + ((tp_ & 1) << 3);
    }
    auto frag = [&](const char* X, int tile, int ks) __attribute__((always_inline)) {
        return *reinterpret_cast<const bf16x8*>(X + foff[ks] + tile * TB);
    };
    auto tfrag = [&](const char* X, int tile, int dt) __attribute__((always_inline)) {
        return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(X + toff[dt] + tile * TB));
    };
    // a 16-row tile's plain fragments from GLOBAL memory: lane (g, i16) holds X[tile*16 + i16][ks*32 + 8g .. +8], zero beyond DH, row clamped
    auto gfrag = [&](const uint16_t* X, int64_t ldx, int tile, bf16x8 (&f)[KS]) __attribute__((always_inline)) {
        int r = tile * 16 + i16;
        r = r < S ? r : S - 1;
        const uint16_t* rp = X + (int64_t)r * ldx;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int dcol = ks * 32 + 8 * g;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (dcol < DH) v = *reinterpret_cast<const u32x4*>(rp + dcol);
            f[ks] = __builtin_bit_cast(bf16x8, v);
        }
    };
    const int nqt = (S + 15) >> 4;
    f32x4 cq[NDT], ck[NDT], cv[NDT];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) { cq[dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; ck[dt] = cq[dt]; cv[dt] = cq[dt]; }

    // first tile's Q / dO fragments travel with the K | V pieces
    bf16x8 qf[KS], of[KS];
    if (wid < nqt) { gfrag(qb, ld, wid, qf); gfrag(ob, D, wid, of); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // =============================== pass 1: per 16-query tile (K | V in LDS) ===============================
    for (int qt = wid; qt < nqt; qt += NW) {
        const int q0 = qt << 4;
        f32x4 sc[NKT], dp[NKT];
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f}, c = a;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                a = PV_MFMA_16x16x32(frag(A_, kt, ks), qf[ks], a, 0, 0, 0);
                c = PV_MFMA_16x16x32(frag(B_, kt, ks), of[ks], c, 0, 0, 0);
            }
            sc[kt] = a; dp[kt] = c;
        }
        // the next tile's fragments are requested now: they arrive under this tile's softmax and dQ products
        bf16x8 qn[KS], on[KS];
        const bool more = qt + NW < nqt;
        if (more) { gfrag(qb, ld, qt + NW, qn); gfrag(ob, D, qt + NW, on); }
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if ((NKT - 1) * 16 + 4 * g + r >= S) sc[NKT - 1][r] = -INFINITY;
        float m = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) m = fmaxf(fmaxf(m, fmaxf(sc[kt][0], sc[kt][1])), fmaxf(sc[kt][2], sc[kt][3]));
        m = fmaxf(m, __shfl_xor(m, 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        const float nm = -m * LOG2E;
        float l = 0.f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pe = __builtin_amdgcn_exp2f(fmaf(sc[kt][r], LOG2E, nm));
                sc[kt][r] = pe;
                l += pe;
            }
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        const float inv = 1.0f / l;
        float dd = 0.f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                sc[kt][r] *= inv;
                dd = fmaf(sc[kt][r], dp[kt][r], dd);
            }
        dd += __shfl_xor(dd, 16, 64);
        dd += __shfl_xor(dd, 32, 64);
        if (g == 0) {          // a padded query (duplicate of row S - 1) gets p = 0 in pass 2
            const bool okq = q0 + i16 < S;
            st_m[q0 + i16] = okq ? nm + __builtin_amdgcn_logf(inv) : -INFINITY;
            st_d[q0 + i16] = okq ? dd : 0.f;
        }
        f32x4 dq[NDT];
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) dq[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        auto ds_pack = [&](int kt) __attribute__((always_inline)) {
            return (u32x2){pv_pack_bf16x2(sc[kt][0] * (dp[kt][0] - dd), sc[kt][1] * (dp[kt][1] - dd)),
                           pv_pack_bf16x2(sc[kt][2] * (dp[kt][2] - dd), sc[kt][3] * (dp[kt][3] - dd))};
        };
#pragma unroll
        for (int tt = 0; tt < NKT / 2; ++tt) {
            const u32x2 d0 = ds_pack(2 * tt), d1 = ds_pack(2 * tt + 1);
            const bf16x8 dsf = __builtin_bit_cast(bf16x8, (u32x4){d0[0], d0[1], d1[0], d1[1]});
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                const s16x8 kk = __builtin_shufflevector(tfrag(A_, 2 * tt, dt), tfrag(A_, 2 * tt + 1, dt), 0, 1, 2, 3, 4, 5, 6, 7);
                dq[dt] = PV_MFMA_16x16x32(__builtin_bit_cast(bf16x8, kk), dsf, dq[dt], 0, 0, 0);
            }
        }
        if (NKT & 1) {
            const s16x4 dsf = __builtin_bit_cast(s16x4, ds_pack(NKT - 1));
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) dq[dt] = PV_MFMA_16x16x16(tfrag(A_, NKT - 1, dt), dsf, dq[dt], 0, 0, 0);
        }
        if (q0 + i16 < S) {
            uint16_t* op = gb + (int64_t)(q0 + i16) * ld + 4 * g;
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                const u32x2 ov = {pv_pack_bf16x2(dq[dt][0] * qscale, dq[dt][1] * qscale), pv_pack_bf16x2(dq[dt][2] * qscale, dq[dt][3] * qscale)};
                *reinterpret_cast<u32x2*>(op + dt * 16) = ov;
                cq[dt] += (f32x4){pv_unpack_lo(ov[0]), pv_unpack_hi(ov[0]), pv_unpack_lo(ov[1]), pv_unpack_hi(ov[1])};
            }
        }
        if (more) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) { qf[ks] = qn[ks]; of[ks] = on[ks]; }
        }
    }
    __syncthreads();                       // every wave has left pass 1: the K | V images are free, the statistics are written

    // =============================== phase 2: Q | dO images, pass 2 per unit of KPW 16-key tiles ===============================
    stage(qb, ld, ob, D);
    constexpr int KPW = (NKT >= 5) ? 2 : 1;         // 4 waves: two key tiles per wave from five tiles on (13 tiles = 7 units = two rounds)
    bf16x8 kf[KPW][KS], vf[KPW][KS];
    auto load_unit = [&](int ku) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < KPW; ++t) {
            const int kt = ku * KPW + t, ktc = kt < NKT ? kt : NKT - 1;
            gfrag(qb + D, ld, ktc, kf[t]);
            gfrag(qb + 2 * D, ld, ktc, vf[t]);
        }
    };
    if (wid * KPW < nqt) load_unit(wid);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int ku = wid; ku * KPW < nqt; ku += NW) {
        bool key_ok[KPW];
        int k0[KPW];
#pragma unroll
        for (int t = 0; t < KPW; ++t) {
            const int kt = ku * KPW + t;
            k0[t] = kt << 4;
            key_ok[t] = kt < nqt && k0[t] + i16 < S;
        }
        f32x4 dv[KPW][NDT], dk[KPW][NDT];
#pragma unroll
        for (int t = 0; t < KPW; ++t)
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) { dv[t][dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; dk[t][dt] = dv[t][dt]; }
        auto pds = [&](int qt, u32x2 (&pw)[KPW], u32x2 (&dw)[KPW]) __attribute__((always_inline)) {
            bf16x8 qq_[KS], oo_[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) { qq_[ks] = frag(A_, qt, ks); oo_[ks] = frag(B_, qt, ks); }
            const float4 m4 = *reinterpret_cast<const float4*>(st_m + qt * 16 + 4 * g);
            const float4 d4 = *reinterpret_cast<const float4*>(st_d + qt * 16 + 4 * g);
#ifdef PV_OPERAND_F16
            const float mm[4] = {m4.x + PV_P_SHIFT, m4.y + PV_P_SHIFT, m4.z + PV_P_SHIFT, m4.w + PV_P_SHIFT};
            const float dd[4] = {d4.x * PV_P_UNSHIFT, d4.y * PV_P_UNSHIFT, d4.z * PV_P_UNSHIFT, d4.w * PV_P_UNSHIFT};
#else
            const float mm[4] = {m4.x, m4.y, m4.z, m4.w}, dd[4] = {d4.x, d4.y, d4.z, d4.w};
#endif
#pragma unroll
            for (int t = 0; t < KPW; ++t) {
                f32x4 s_ = {0.f, 0.f, 0.f, 0.f}, c = s_;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    s_ = PV_MFMA_16x16x32(qq_[ks], kf[t][ks], s_, 0, 0, 0);
                    c = PV_MFMA_16x16x32(oo_[ks], vf[t][ks], c, 0, 0, 0);
                }
                float p[4], ds[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    p[r] = key_ok[t] ? __builtin_amdgcn_exp2f(fmaf(s_[r], LOG2E, mm[r])) : 0.f;
#ifdef PV_OPERAND_F16
                    ds[r] = p[r] * fmaf(c[r], PV_P_UNSHIFT, -dd[r]);
#else
                    ds[r] = p[r] * (c[r] - dd[r]);
#endif
                }
                pw[t] = (u32x2){pv_pack_bf16x2(p[0], p[1]), pv_pack_bf16x2(p[2], p[3])};
                dw[t] = (u32x2){pv_pack_bf16x2(ds[0], ds[1]), pv_pack_bf16x2(ds[2], ds[3])};
            }
        };
#ifndef PV_ABW2_P2_UNROLL
#define PV_ABW2_P2_UNROLL 1
#endif
#pragma unroll PV_ABW2_P2_UNROLL
        for (int tt = 0; tt < NKT / 2; ++tt) {
            u32x2 p0[KPW], d0[KPW], p1[KPW], d1[KPW];
            pds(2 * tt, p0, d0);
            pds(2 * tt + 1, p1, d1);
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                const s16x8 oo = __builtin_shufflevector(tfrag(B_, 2 * tt, dt), tfrag(B_, 2 * tt + 1, dt), 0, 1, 2, 3, 4, 5, 6, 7);
                const s16x8 qq = __builtin_shufflevector(tfrag(A_, 2 * tt, dt), tfrag(A_, 2 * tt + 1, dt), 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
                for (int t = 0; t < KPW; ++t) {
                    const bf16x8 pf = __builtin_bit_cast(bf16x8, (u32x4){p0[t][0], p0[t][1], p1[t][0], p1[t][1]});
                    const bf16x8 dsf = __builtin_bit_cast(bf16x8, (u32x4){d0[t][0], d0[t][1], d1[t][0], d1[t][1]});
                    dv[t][dt] = PV_MFMA_16x16x32(__builtin_bit_cast(bf16x8, oo), pf, dv[t][dt], 0, 0, 0);
                    dk[t][dt] = PV_MFMA_16x16x32(__builtin_bit_cast(bf16x8, qq), dsf, dk[t][dt], 0, 0, 0);
                }
            }
        }
        if (NKT & 1) {
            u32x2 pw[KPW], dw[KPW];
            pds(NKT - 1, pw, dw);
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                const s16x4 oo = tfrag(B_, NKT - 1, dt), qq = tfrag(A_, NKT - 1, dt);
#pragma unroll
                for (int t = 0; t < KPW; ++t) {
                    dv[t][dt] = PV_MFMA_16x16x16(oo, __builtin_bit_cast(s16x4, pw[t]), dv[t][dt], 0, 0, 0);
                    dk[t][dt] = PV_MFMA_16x16x16(qq, __builtin_bit_cast(s16x4, dw[t]), dk[t][dt], 0, 0, 0);
                }
            }
        }
        // the wave's next unit: its K / V fragments are requested before this unit's stores
        const bool more = (ku + NW) * KPW < nqt;
        bool ok_now[KPW];
        int k0_now[KPW];
#pragma unroll
        for (int t = 0; t < KPW; ++t) { ok_now[t] = key_ok[t]; k0_now[t] = k0[t]; }
        if (more) load_unit(ku + NW);
#pragma unroll
        for (int t = 0; t < KPW; ++t)
            if (ok_now[t]) {
                uint16_t* op = gb + (int64_t)(k0_now[t] + i16) * ld + 4 * g;
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt) {
                    const u32x2 kv = {pv_pack_bf16x2(dk[t][dt][0], dk[t][dt][1]), pv_pack_bf16x2(dk[t][dt][2], dk[t][dt][3])};
#ifdef PV_OPERAND_F16
                    dv[t][dt] = dv[t][dt] * PV_P_UNSHIFT;
#endif
                    const u32x2 vv = {pv_pack_bf16x2(dv[t][dt][0], dv[t][dt][1]), pv_pack_bf16x2(dv[t][dt][2], dv[t][dt][3])};
                    *reinterpret_cast<u32x2*>(op + D + dt * 16) = kv;
                    *reinterpret_cast<u32x2*>(op + 2 * D + dt * 16) = vv;
                    ck[dt] += (f32x4){pv_unpack_lo(kv[0]), pv_unpack_hi(kv[0]), pv_unpack_lo(kv[1]), pv_unpack_hi(kv[1])};
                    cv[dt] += (f32x4){pv_unpack_lo(vv[0]), pv_unpack_hi(vv[0]), pv_unpack_lo(vv[1]), pv_unpack_hi(vv[1])};
                }
            }
    }
    if (dbp) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);      // [NW][3][DH]
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float a = pv_row16_sum(cq[dt][r]), bsum = pv_row16_sum(ck[dt][r]), c = pv_row16_sum(cv[dt][r]);
                if (i16 == 0) {
                    const int col = dt * 16 + 4 * g + r;
                    red[(wid * 3 + 0) * DH + col] = a; red[(wid * 3 + 1) * DH + col] = bsum; red[(wid * 3 + 2) * DH + col] = c;
                }
            }
        __syncthreads();
        for (int e = tid; e < 3 * DH; e += NW * 64) {
            const int part = e / DH, col = e - part * DH;
            float t = 0.f;
            for (int w = 0; w < NW; ++w) t += red[(w * 3 + part) * DH + col];
            dbp[(int64_t)b * 3 * D + part * D + h * DH + col] = t;
        }
    }
}

template <int DH, int NKT>
static int pv_launch_attn_bwd2(const uint16_t* qkv, const uint16_t* dout, uint16_t* dqkv, float* dbp, int64_t B, int S, int H, float qscale, hipStream_t stream) {
    constexpr int DHP = (DH + 31) / 32 * 32;
    constexpr int lds = 2 * NKT * 16 * DHP * 2 + 2 * NKT * 16 * 4;
    // three workgroups per CU (<= 168 registers per wave) only where the kernel fits them without spilling (dh = 32 up to eleven key tiles: hipcc's
    // own count, -Rpass-analysis=kernel-resource-usage); a scratch reload is a vector-memory operation that waits for every LDS-DMA piece in flight
    constexpr int WGS = (DH == 32 && NKT <= 11 && 3 * lds <= 160 * 1024) ? 3 : 2;
    static PvPerDevice attr_set;
    if (attr_set.first_use()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pv_attn_bwd2_kernel<DH, NKT, WGS>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    }
    PV_LAUNCH((pv_attn_bwd2_kernel<DH, NKT, WGS>), dim3((unsigned)(B * H)), dim3(256), lds, stream, qkv, dout, dqkv, dbp, S, H, qscale, (int)B);
    return pv_check_launch();
}

template <int DH, int NKT>
static int pv_launch_attn_bwd(const uint16_t* qkv, const uint16_t* dout, uint16_t* dqkv, float* dbp, int64_t B, int S, int H, float qscale, hipStream_t stream) {
    constexpr int DHP = (DH + 31) / 32 * 32;
    constexpr int lds = 4 * NKT * 16 * DHP * 2 + 2 * NKT * 16 * 4;
    constexpr int NW = NKT <= 13 ? 8 : 4;
    static_assert(lds <= 160 * 1024, "Q, K, V, dO of one head must fit the LDS");
    static PvPerDevice attr_set;
    if (attr_set.first_use()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pv_attn_bwd_kernel<DH, NKT, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    }
    PV_LAUNCH((pv_attn_bwd_kernel<DH, NKT, NW>), dim3((unsigned)(B * H)), dim3(NW * 64), lds, stream, qkv, dout, dqkv, dbp, S, H, qscale);
    return pv_check_launch();
}

// ------------------------------------------------------------------------------------------------
// Persistent attention backward from the forward's row statistics (round 5, `pv_attn_bwd5_kernel`, entry pv_attention_bwd_lse_bf16).
// What the measurements on the two-pass kernels above say (profiles/r05_attn_bwd4_experiment.txt): their arithmetic fills half of the SIMD
// cycles, their traffic half of the HBM time, and 58 % of the wave cycles are spent waiting - load and compute ALTERNATE inside a workgroup,
// two workgroups are all the LDS holds, and every buffer has an owner that idles while it fills; operands are fetched twice (as LDS images
// for one pass, as register fragments for the other).  Here ONE workgroup of 16 waves per CU walks over its (image, head) items:
//   * wave w < ceil(S / 16) owns query tile w in pass 1 and key tile w in pass 2 (one tile each: no tile switch, no prefetch registers);
//   * pass 1 streams over key-tile pairs with p = exp2(s log2e - lse), dS = p (dP - D) (lse from the forward, the dP accumulator starts
//     at -D): two score tiles live, < 128 registers, four waves per SIMD;
//   * K | V of item j sit in buffer X for pass 1 while Q | dO of item j land in buffer Y; the wave's own K / V fragments for pass 2 are
//     read from X as pass 1 ends, so X is free for K | V of item j + 1 while pass 2 runs on Y: every LDS-DMA has a whole pass to land,
//     every operand byte is fetched once;
//   * the three waves without a tile do the item's side work while the others compute: D = rowsum(dO o O) and lse -> LDS for the NEXT
//     item (pass 2), the bias-gradient column sums (query third: the stored dQ tiles transposed through a private LDS scratch and summed
//     by an MFMA against ones; key third: 0 identically - sum_k dS[q,k] = D - D; value third: column sums of dO by MFMAs over the image).
// Two raw barriers per item; counted vmcnt waits (loads, stores and LDS-DMA retire in issue order).  No atomics: bit-reproducible.
// ------------------------------------------------------------------------------------------------
#ifndef PV_ABW5_PRIO
#define PV_ABW5_PRIO 0          // wave priority by age in the persistent backward (A/B: -DPV_ABW5_PRIO=1; see the kernel)
#endif
#ifdef PV_OPERAND_F16
#define PV_ONE16 0x3C00
#else
#define PV_ONE16 0x3F80
#endif
template <int DH, int NKT>
__global__ __launch_bounds__(1024) void pv_attn_bwd5_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ dout, const uint16_t* __restrict__ att,
                                                            const float* __restrict__ lse, uint16_t* __restrict__ dqkv, float* __restrict__ dbp, int S, int H,
                                                            float qscale, int B, int n_items) {
    constexpr int NW = 16, NT = NW * 64;
    constexpr int DHP = (DH + 31) / 32 * 32, CPR = DHP / 8, TB = 16 * DHP * 2;
    constexpr int SP = NKT * 16, IMG = SP * DHP * 2, NDT = DH / 16, KS = DHP / 32;
    constexpr int NCH = SP * CPR, NIT = (NCH + NT - 1) / NT;
    constexpr float LOG2E = 1.44269504088896340736f;
    static_assert(CPR == 8, "row images of 128 bytes (dh = 48 / 64)");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const X0 = smem;                 // K
    char* const X1 = smem + IMG;           // V
    char* const Y0 = smem + 2 * IMG;       // Q
    char* const Y1 = smem + 3 * IMG;       // dO
    float* const st = reinterpret_cast<float*>(smem + 4 * IMG);         // [2 parities][m | d][SP]: PV_P_SHIFT - lse (-inf: padded query), -D
    float* const red_q = st + 4 * SP;                                     // [NW][DH] column sums of the dQ tiles
    const int tid = threadIdx.x;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    char* const scr = reinterpret_cast<char*>(red_q + NW * DH) + wid * TB;    // this wave's 16 x DHP scratch tile (dQ, transposed read)
    const int D = H * DH;
    const int64_t ld = 3 * (int64_t)D;
    const int nqt = (S + 15) >> 4;
    const bool tile_wave = wid < nqt;
    // Everything that depends on the lane alone is recomputed at the top of every item from an opaque copy of the thread index: left loop-invariant,
    // hipcc hoists ~40 such offsets and masks out of the item loop, spills them, and reloads them under the LDS-DMA (= s_waitcnt vmcnt(0) in pass 1).
    int lane, g, i16, foff[KS], toff[NDT];
    unsigned off3[NIT], off1[NIT];         // per-lane BYTE offsets of the staged rows for the two row strides (q | k | v rows: 3D elements, dO rows: D)
    auto relane = [&]() __attribute__((always_inline)) {
        int t = tid;
        asm volatile("" : "+v"(t));
        lane = t & 63;
        g = lane >> 4;
        i16 = lane & 15;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) foff[ks] = pv_swz<CPR>(i16, ks * 4 + g);
        const int tq_ = i16 >> 2, tp_ = i16 & 3;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) toff[dt] = pv_swz<CPR>(4 * g + tq_, dt * 2 + (tp_ >> 1)) + ((tp_ & 1) << 3);
        const int lsw = (t & 7) ^ ((t >> 3) & 7);
        const int c_lane = lsw * 8 >= DH ? 0 : lsw;
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            int row = i * (NT / CPR) + (t >> 3);
            row = row < S ? row : S - 1;
            off3[i] = (unsigned)(row * 3 * D + c_lane * 8) * 2u;
            off1[i] = (unsigned)(row * D + c_lane * 8) * 2u;
        }
    };
    relane();
    auto stage = [&](char* d0, char* d1, const uint16_t* a, bool a3, const uint16_t* bsrc, bool b3) __attribute__((always_inline)) {
#pragma unroll
        for (int which = 0; which < 2; ++which) {
#pragma unroll
            for (int i = 0; i < NIT; ++i) {
                if (NCH % NT == 0 || i * NT + wid * 64 < NCH) {
                    const char* src = reinterpret_cast<const char*>(which ? bsrc : a) + ((which ? b3 : a3) ? off3[i] : off1[i]);
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                     (__attribute__((address_space(3))) void*)((which ? d1 : d0) + (size_t)(i * NT + wid * 64) * 16), 16, 0, 0);
                }
            }
        }
    };
    // transposed reads as inline asm: hipcc's wait-count pass puts s_waitcnt vmcnt(0) in front of every ds_read_tr BUILTIN while an LDS-DMA is in flight (it
    // cannot see that the image being filled is not the one being read) - which would park pass 1 until Q | dO have landed.  trN reads NDT fragments of one
    // tile (tr2N: of two consecutive tiles, joined for the K = 32 MFMA); the s_waitcnt that follows names them as operands, so every consumer depends on it.
    typedef __attribute__((address_space(3))) char lds_c;
    auto tr_issue = [&](const char* X, int tile, s16x4 (&f)[NDT]) __attribute__((always_inline)) {
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f[dt]) : "v"((lds_c*)(X + toff[dt] + tile * TB)));
    };
    // ... and so are the plain fragment reads of the two passes: any LDS read hipcc can see after an LDS-DMA was issued waits for that DMA (its wait-count
    // pass has no alias information to tell the image being filled from the one being read).
    auto fr_issue = [&](const char* X, int tile, bf16x8 (&f)[KS]) __attribute__((always_inline)) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) asm volatile("ds_read_b128 %0, %1" : "=v"(f[ks]) : "v"((lds_c*)(X + foff[ks] + tile * TB)));
    };
    // the same with the tile and image offsets as INSTRUCTION immediates: `base[ks]` = image 0 + foff[ks] + (first tile of the loop trip) * TB, one add per trip
    auto fr_issue_i = [&](lds_c* const (&base)[KS], auto off_c, bf16x8 (&f)[KS]) __attribute__((always_inline)) {
        constexpr int OFF = decltype(off_c)::value;
        static_assert(OFF >= 0 && OFF < 65536, "DS offset field");
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f[ks]) : "v"(base[ks]), "n"(OFF));
    };
    auto tr_issue_i = [&](lds_c* const (&base)[NDT], auto off_c, s16x4 (&f)[NDT]) __attribute__((always_inline)) {
        constexpr int OFF = decltype(off_c)::value;
        static_assert(OFF >= 0 && OFF < 65536, "DS offset field");
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f[dt]) : "v"(base[dt]), "n"(OFF));
    };
    auto fr_wait = [&](bf16x8 (&f)[KS]) __attribute__((always_inline)) {
        if constexpr (KS == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1])::"memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0])::"memory");
    };
    auto tr_wait = [&](s16x4 (&f)[NDT]) __attribute__((always_inline)) {
        if constexpr (NDT == 4) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3])::"memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2])::"memory");
    };
    // global addresses are a uniform base + a 32-bit per-lane byte offset throughout (64-bit per-lane pointers kept across the item loop cost two registers each)
    auto gfrag = [&](const uint16_t* Xg, int ldx, int tile, bf16x8 (&f)[KS]) __attribute__((always_inline)) {
        int r = tile * 16 + i16;
        r = r < S ? r : S - 1;
        const unsigned o = (unsigned)(r * ldx + 8 * g) * 2u;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            u32x4 v = {0u, 0u, 0u, 0u};
            if (ks * 32 + 8 * g < DH) v = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(Xg) + o + ks * 64);
            f[ks] = __builtin_bit_cast(bf16x8, v);
        }
    };
#if PV_ABW5_PRIO
    // Experiment (off): the SIMD's instruction arbiter serves the OLDEST ready wave first - of the four waves of a SIMD (w, w + 4, w + 8, w + 12) the youngest
    // finishes every pass last (stamps: pass 2 takes wave 0 10 k ticks and wave 12 20 k).  Priority by age, youngest highest, only REVERSES the order (wave 12
    // 9.7 k, wave 0 19.6 k; 1.70 vs 1.73 ms per launch): the four tiles of SIMD 0 cost 20 k ticks of that SIMD's issue whoever goes first - the passes are
    // issue-bound, and the thirteenth tile (a fourth wave on one SIMD, three on the others) is what sets the pass length.
    switch (wid >> 2) {
        case 1: __builtin_amdgcn_s_setprio(1); break;
        case 2: __builtin_amdgcn_s_setprio(2); break;
        case 3: __builtin_amdgcn_s_setprio(3); break;
        default: break;
    }
#endif
    int it = blockIdx.x, par = 0;
    bf16x8 qf[KS], of[KS], af[KS];       // this wave's rows of Q, dO and of the forward's output O (for D = rowsum(dO o O)), prefetched one item ahead
    float lse_v = 0.f;                   // lse of query wid * 16 + i16
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) { qf[ks] = __builtin_bit_cast(bf16x8, (u32x4){0u, 0u, 0u, 0u}); of[ks] = qf[ks]; af[ks] = qf[ks]; }
    auto own_rows = [&](int bn, int hn) __attribute__((always_inline)) {
        gfrag(qkv + (int64_t)bn * S * ld + hn * DH, 3 * D, wid, qf);
        gfrag(dout + (int64_t)bn * S * D + hn * DH, D, wid, of);
        gfrag(att + (int64_t)bn * S * D + hn * DH, D, wid, af);
        int r = wid * 16 + i16;
        r = r < S ? r : S - 1;
        lse_v = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(lse + ((int64_t)bn * H + hn) * S) + (unsigned)r * 4u);
    };
    {
        int b, h;
        pv_bh_map(it, B, H, b, h);
        const uint16_t* qb = qkv + (int64_t)b * S * ld + h * DH;
        stage(X0, X1, qb + D, true, qb + 2 * D, true);
        if (tile_wave) own_rows(b, h);
    }

    for (;;) {
        relane();
#ifdef PV_STAMPS
        const bool pv_stamp_on = it + (int)gridDim.x < n_items;          // (a steady-state item: the last one has no side work for a successor)
#endif
        int b, h;
        pv_bh_map(it, B, H, b, h);
        const uint16_t* const qb = qkv + (int64_t)b * S * ld + h * DH;
        const uint16_t* const ob = dout + (int64_t)b * S * D + h * DH;
        uint16_t* const gb = dqkv + (int64_t)b * S * ld + h * DH;
        // ---- [A] K | V of this item, this wave's fragments and (LDS) the row statistics are in place ----
        PV_CSTAMP(0);
        // hipcc's own wait for the prefetched fragments goes HERE, on every path (its wait-count pass is path-insensitive: pinned under `if (tile_wave)` only,
        // the fragments still count as in flight where pass 1 first uses them, and the wait it puts there - vmcnt(0) - would hold pass 1 until Q | dO have landed)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[ks]), "+v"(of[ks]), "+v"(af[ks]));
        asm volatile("" : "+v"(lse_v));
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        float* const sm = st + par * 2 * SP;
        float* const sd = sm + SP;
        float mq = 0.f, nd = 0.f;
        if (tile_wave) {
            // this tile's row statistics: D = rowsum(dO o O) over the lane's 16 columns, then over the four lanes of the query; to LDS for pass 2 of every wave
            float dsum = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const u32x4 dv_ = __builtin_bit_cast(u32x4, of[ks]), ov_ = __builtin_bit_cast(u32x4, af[ks]);
#pragma unroll
                for (int j = 0; j < 4; ++j) dsum = fmaf(pv_unpack_lo(dv_[j]), pv_unpack_lo(ov_[j]), fmaf(pv_unpack_hi(dv_[j]), pv_unpack_hi(ov_[j]), dsum));
            }
            dsum += __shfl_xor(dsum, 16, 64);
            dsum += __shfl_xor(dsum, 32, 64);
            const bool okq = wid * 16 + i16 < S;
            mq = okq ? -lse_v : -INFINITY;
            nd = okq ? -dsum : 0.f;
            if (g == 0) { sm[wid * 16 + i16] = mq + PV_P_SHIFT; sd[wid * 16 + i16] = nd; }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        PV_CSTAMP(1);
        __builtin_amdgcn_s_barrier();
        PV_CSTAMP(2);
        stage(Y0, Y1, qb, true, ob, false);                  // Q | dO land while pass 1 runs
        bf16x8 kf[KS], vf[KS];
        // =============================== pass 1: dQ of query tile `wid`, streaming over pairs of key tiles (K | V in X) ===============================
        if (tile_wave) {
            const int q0 = wid << 4;
            f32x4 dq[NDT];
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) dq[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            auto tile_ds = [&](const bf16x8 (&kx)[KS], const bf16x8 (&vx)[KS], int kt, bool masked) __attribute__((always_inline)) {
                f32x4 a = {0.f, 0.f, 0.f, 0.f}, c = {nd, nd, nd, nd};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    a = PV_MFMA_16x16x32(kx[ks], qf[ks], a, 0, 0, 0);
                    c = PV_MFMA_16x16x32(vx[ks], of[ks], c, 0, 0, 0);
                }
                float ds[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float p = __builtin_amdgcn_exp2f(fmaf(a[r], LOG2E, mq));
                    if (masked) p = kt * 16 + 4 * g + r < S ? p : 0.f;          // (only the last tile holds padded keys)
                    ds[r] = p * c[r];
                }
                return (u32x2){pv_pack_bf16x2(ds[0], ds[1]), pv_pack_bf16x2(ds[2], ds[3])};
            };
            auto pair = [&](int k0_, bool masked) __attribute__((always_inline)) {
                bf16x8 k0f[KS], v0f[KS], k1f[KS], v1f[KS];
                s16x4 ka[NDT], kb[NDT];
                lds_c* fb[KS];
                lds_c* tb[NDT];
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) fb[ks] = (lds_c*)(X0 + foff[ks] + k0_ * TB);
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt) tb[dt] = (lds_c*)(X0 + toff[dt] + k0_ * TB);
                fr_issue_i(fb, std::integral_constant<int, 0>{}, k0f);
                fr_issue_i(fb, std::integral_constant<int, IMG>{}, v0f);
                fr_issue_i(fb, std::integral_constant<int, TB>{}, k1f);
                fr_issue_i(fb, std::integral_constant<int, IMG + TB>{}, v1f);
                tr_issue_i(tb, std::integral_constant<int, 0>{}, ka);
                tr_issue_i(tb, std::integral_constant<int, TB>{}, kb);
                fr_wait(k0f); fr_wait(v0f); fr_wait(k1f); fr_wait(v1f);
                const u32x2 d0 = tile_ds(k0f, v0f, k0_, false), d1 = tile_ds(k1f, v1f, k0_ + 1, masked);
                const bf16x8 dsf = __builtin_bit_cast(bf16x8, (u32x4){d0[0], d0[1], d1[0], d1[1]});
                tr_wait(ka);
                tr_wait(kb);
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt) {
                    const s16x8 kk = __builtin_shufflevector(ka[dt], kb[dt], 0, 1, 2, 3, 4, 5, 6, 7);
                    dq[dt] = PV_MFMA_16x16x32(__builtin_bit_cast(bf16x8, kk), dsf, dq[dt], 0, 0, 0);
                }
            };
#pragma unroll 1
            for (int tt = 0; tt < (NKT - 1) / 2; ++tt) pair(2 * tt, false);
            if (NKT & 1) {
                bf16x8 k0f[KS], v0f[KS];
                s16x4 ka[NDT];
                fr_issue(X0, NKT - 1, k0f);
                fr_issue(X1, NKT - 1, v0f);
                tr_issue(X0, NKT - 1, ka);
                fr_wait(k0f); fr_wait(v0f);
                const s16x4 dsf = __builtin_bit_cast(s16x4, tile_ds(k0f, v0f, NKT - 1, true));
                tr_wait(ka);
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt) dq[dt] = PV_MFMA_16x16x16(ka[dt], dsf, dq[dt], 0, 0, 0);
            } else {
                pair(NKT - 2, true);
            }
            const bool okq = q0 + i16 < S;
            char* const op = reinterpret_cast<char*>(gb) + (unsigned)((q0 + i16) * 3 * D + 4 * g) * 2u;
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                const u32x2 ov = {pv_pack_bf16x2(dq[dt][0] * qscale, dq[dt][1] * qscale), pv_pack_bf16x2(dq[dt][2] * qscale, dq[dt][3] * qscale)};
                if (okq) *reinterpret_cast<u32x2*>(op + dt * 32) = ov;
                if (dbp) *reinterpret_cast<u32x2*>(scr + pv_swz<CPR>(i16, dt * 2 + (g >> 1)) + ((g & 1) << 3)) = okq ? ov : (u32x2){0u, 0u};
            }
            // this wave's K / V fragments for pass 2, from the images that pass 1 is done with
            fr_issue(X0, wid, kf);
            fr_issue(X1, wid, vf);
            fr_wait(kf); fr_wait(vf);
            if (DH % 32 != 0 && 32 * (KS - 1) + 8 * g >= DH) {         // dh = 48: the images' pad columns hold copies of chunk 0, and pass 2 multiplies these fragments with image rows
                kf[KS - 1] = __builtin_bit_cast(bf16x8, (u32x4){0u, 0u, 0u, 0u});
                vf[KS - 1] = kf[KS - 1];
            }
            if (dbp) {             // column sums of the stored dQ tile: its transpose against ones
                const s16x4 ones = {(short)PV_ONE16, (short)PV_ONE16, (short)PV_ONE16, (short)PV_ONE16};
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the scratch tile is written (LDS operations of a wave execute in order; the read below is asm)
                s16x4 sa[NDT];
                tr_issue(scr, 0, sa);
                tr_wait(sa);
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt) {
                    const f32x4 t = PV_MFMA_16x16x16(sa[dt], ones, ((f32x4){0.f, 0.f, 0.f, 0.f}), 0, 0, 0);
                    if (i16 == 0) *reinterpret_cast<f32x4*>(red_q + wid * DH + dt * 16 + 4 * g) = t;
                }
            }
        }
        PV_CSTAMP(3);
        // ---- [E] Q | dO landed (issued before this wave's NDT stores of dQ), every wave is done with X ----
        if (tile_wave) {
            if (NDT == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        }
        PV_CSTAMP(4);
        __builtin_amdgcn_s_barrier();
        PV_CSTAMP(5);
        const int itn = it + gridDim.x;
        const bool more = itn < n_items;
        int bn = b, hn = h;
        if (more) pv_bh_map(itn, B, H, bn, hn);
        const uint16_t* const qbn = qkv + (int64_t)bn * S * ld + hn * DH;
        const uint16_t* const obn = dout + (int64_t)bn * S * D + hn * DH;
        if (more) {
            stage(X0, X1, qbn + D, true, qbn + 2 * D, true);      // K | V of the next item land while pass 2 runs
            if (tile_wave) {
                // this wave's Q / dO rows of the next item: pass 2 has no 16 registers to hold them, so they are only TOUCHED here (one dword per 128-byte
                // line, result discarded: the lines come to the L2) and loaded where the pass-2 loop ends, from the L2
                int r = wid * 16 + i16;
                r = r < S ? r : S - 1;
                // (as LDS-DMA into this wave's scratch tile, which is idle in pass 2: an inline-asm load into a register returns LATE, and a register hipcc
                //  believes dead - it spilled the three touch destinations at once - is reused while the load is still in flight)
                typedef const __attribute__((address_space(1))) void* gptr;
                typedef __attribute__((address_space(3))) void* lptr;
                __builtin_amdgcn_global_load_lds((gptr)(reinterpret_cast<const char*>(qbn) + (unsigned)(r * 3 * D + g * 16) * 2u), (lptr)(scr), 4, 0, 0);
                __builtin_amdgcn_global_load_lds((gptr)(reinterpret_cast<const char*>(obn) + (unsigned)(r * D + g * 16) * 2u), (lptr)(scr + 256), 4, 0, 0);
                __builtin_amdgcn_global_load_lds((gptr)(reinterpret_cast<const char*>(att + (int64_t)bn * S * D + hn * DH) + (unsigned)(r * D + g * 16) * 2u),
                                                 (lptr)(scr + 512), 4, 0, 0);
            }
        }
        if (tile_wave) {
            // =============================== pass 2: dK, dV of key tile `wid` over all query tiles (Q | dO in Y) ===============================
            f32x4 dv[NDT], dk[NDT];
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) { dv[dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; dk[dt] = dv[dt]; }
            // lane (g, i16) = key i16 of the tile, queries 4g .. 4g + 3.  No key mask: a padded key (a duplicate of key S - 1) only feeds its own dK / dV
            // rows, which are not stored; a padded query has st_m = -inf -> p = 0.
            // (fb / sb: Y0 + foff + qt0 * TB and the statistics row of query tile qt0; OFF_T = 0 or TB: the tile of the pair)
            auto pds = [&](lds_c* const (&fb)[KS], lds_c* sb, auto off_t, u32x2& pw, u32x2& dw) __attribute__((always_inline)) {
                constexpr int OFF_T = decltype(off_t)::value;
                bf16x8 qx[KS], ox[KS];
                f32x4 c, m4;
                fr_issue_i(fb, std::integral_constant<int, OFF_T>{}, qx);
                fr_issue_i(fb, std::integral_constant<int, IMG + OFF_T>{}, ox);
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(c) : "v"(sb), "n"(SP * 4 + (OFF_T ? 64 : 0)));
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(m4) : "v"(sb), "n"(OFF_T ? 64 : 0));
                fr_wait(qx); fr_wait(ox);
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(c), "+v"(m4)::"memory");
                f32x4 s_ = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    s_ = PV_MFMA_16x16x32(qx[ks], kf[ks], s_, 0, 0, 0);
                    c = PV_MFMA_16x16x32(ox[ks], vf[ks], c, 0, 0, 0);
                }
                float p[4], ds[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    p[r] = __builtin_amdgcn_exp2f(fmaf(s_[r], LOG2E, m4[r]));          // p 2^PV_P_SHIFT
#ifdef PV_OPERAND_F16
                    ds[r] = p[r] * (c[r] * PV_P_UNSHIFT);
#else
                    ds[r] = p[r] * c[r];
#endif
                }
                pw = (u32x2){pv_pack_bf16x2(p[0], p[1]), pv_pack_bf16x2(p[2], p[3])};
                dw = (u32x2){pv_pack_bf16x2(ds[0], ds[1]), pv_pack_bf16x2(ds[2], ds[3])};
            };
#pragma unroll 1
            for (int tt = 0; tt < NKT / 2; ++tt) {
                u32x2 p0, d0, p1, d1;
                lds_c* fb[KS];
                lds_c* tb[NDT];
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) fb[ks] = (lds_c*)(Y0 + foff[ks] + 2 * tt * TB);
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt) tb[dt] = (lds_c*)(Y0 + toff[dt] + 2 * tt * TB);
                lds_c* const sb = (lds_c*)reinterpret_cast<const char*>(sm + 2 * tt * 16 + 4 * g);
                pds(fb, sb, std::integral_constant<int, 0>{}, p0, d0);
                pds(fb, sb, std::integral_constant<int, TB>{}, p1, d1);
                const bf16x8 pf = __builtin_bit_cast(bf16x8, (u32x4){p0[0], p0[1], p1[0], p1[1]});
                const bf16x8 dsf = __builtin_bit_cast(bf16x8, (u32x4){d0[0], d0[1], d1[0], d1[1]});
                {
                    s16x4 oa[NDT], ob_[NDT];
                    tr_issue_i(tb, std::integral_constant<int, IMG>{}, oa);
                    tr_issue_i(tb, std::integral_constant<int, IMG + TB>{}, ob_);
                    tr_wait(oa);
                    tr_wait(ob_);
#pragma unroll
                    for (int dt = 0; dt < NDT; ++dt)
                        dv[dt] = PV_MFMA_16x16x32(__builtin_bit_cast(bf16x8, __builtin_shufflevector(oa[dt], ob_[dt], 0, 1, 2, 3, 4, 5, 6, 7)), pf, dv[dt], 0, 0, 0);
                }
                {
                    s16x4 qa[NDT], qb_[NDT];
                    tr_issue_i(tb, std::integral_constant<int, 0>{}, qa);
                    tr_issue_i(tb, std::integral_constant<int, TB>{}, qb_);
                    tr_wait(qa);
                    tr_wait(qb_);
#pragma unroll
                    for (int dt = 0; dt < NDT; ++dt)
                        dk[dt] = PV_MFMA_16x16x32(__builtin_bit_cast(bf16x8, __builtin_shufflevector(qa[dt], qb_[dt], 0, 1, 2, 3, 4, 5, 6, 7)), dsf, dk[dt], 0, 0, 0);
                }
            }
            if (NKT & 1) {
                u32x2 pw, dw;
                lds_c* fb[KS];
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) fb[ks] = (lds_c*)(Y0 + foff[ks] + (NKT - 1) * TB);
                pds(fb, (lds_c*)reinterpret_cast<const char*>(sm + (NKT - 1) * 16 + 4 * g), std::integral_constant<int, 0>{}, pw, dw);
                s16x4 oa[NDT], qa[NDT];
                tr_issue(Y1, NKT - 1, oa);
                tr_issue(Y0, NKT - 1, qa);
                tr_wait(oa);
                tr_wait(qa);
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt) {
                    dv[dt] = PV_MFMA_16x16x16(oa[dt], __builtin_bit_cast(s16x4, pw), dv[dt], 0, 0, 0);
                    dk[dt] = PV_MFMA_16x16x16(qa[dt], __builtin_bit_cast(s16x4, dw), dk[dt], 0, 0, 0);
                }
            }
            if (more) own_rows(bn, hn);
            const int key = wid * 16 + i16;
            if (key < S) {
                char* const opk = reinterpret_cast<char*>(gb + D) + (unsigned)(key * 3 * D + 4 * g) * 2u;
                char* const opv = reinterpret_cast<char*>(gb + 2 * D) + (unsigned)(key * 3 * D + 4 * g) * 2u;
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt) {
#ifdef PV_OPERAND_F16
                    dv[dt] = dv[dt] * PV_P_UNSHIFT;
#endif
                    *reinterpret_cast<u32x2*>(opk + dt * 32) = (u32x2){pv_pack_bf16x2(dk[dt][0], dk[dt][1]), pv_pack_bf16x2(dk[dt][2], dk[dt][3])};
                    *reinterpret_cast<u32x2*>(opv + dt * 32) = (u32x2){pv_pack_bf16x2(dv[dt][0], dv[dt][1]), pv_pack_bf16x2(dv[dt][2], dv[dt][3])};
                }
            }
        } else {
            // ---- the waves without a tile: this item's bias-gradient thirds, the next item's row statistics ----
            if (dbp && wid == NW - 2 && lane < DH) {                  // query third: the tile waves' column sums (written before the barrier above)
                float t = 0.f;
                for (int w = 0; w < nqt; ++w) t += red_q[w * DH + lane];
                dbp[(int64_t)b * 3 * D + h * DH + lane] = t;
            }
            if (dbp && wid == NW - 1) {                               // value third: column sums of dO (rows < S) by MFMAs over the image; key third: 0
                f32x4 acc[NDT];
#pragma unroll
                for (int dt = 0; dt < NDT; ++dt) acc[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                const s16x4 ones = {(short)PV_ONE16, (short)PV_ONE16, (short)PV_ONE16, (short)PV_ONE16};
                const bf16x8 ones8 = __builtin_bit_cast(bf16x8, __builtin_shufflevector(ones, ones, 0, 1, 2, 3, 4, 5, 6, 7));
                // four query tiles per trip: their sixteen transposed reads are issued before the first is waited for (one tile per trip, each behind its own
                // lgkmcnt(0), took this wave 16 k ticks - longer than pass 2, with every other wave waiting at the next barrier)
                constexpr int NQ4 = (NKT - 1) / 4;
#pragma unroll 1
                for (int q4 = 0; q4 < NQ4; ++q4) {
                    lds_c* tb[NDT];
#pragma unroll
                    for (int dt = 0; dt < NDT; ++dt) tb[dt] = (lds_c*)(Y1 + toff[dt] + q4 * 4 * TB);
                    s16x4 t0_[NDT], t1_[NDT], t2_[NDT], t3_[NDT];
                    tr_issue_i(tb, std::integral_constant<int, 0>{}, t0_);
                    tr_issue_i(tb, std::integral_constant<int, TB>{}, t1_);
                    tr_issue_i(tb, std::integral_constant<int, 2 * TB>{}, t2_);
                    tr_issue_i(tb, std::integral_constant<int, 3 * TB>{}, t3_);
                    tr_wait(t0_); tr_wait(t1_); tr_wait(t2_); tr_wait(t3_);
#pragma unroll
                    for (int dt = 0; dt < NDT; ++dt) {
                        acc[dt] = PV_MFMA_16x16x32(__builtin_bit_cast(bf16x8, __builtin_shufflevector(t0_[dt], t1_[dt], 0, 1, 2, 3, 4, 5, 6, 7)), ones8, acc[dt], 0, 0, 0);
                        acc[dt] = PV_MFMA_16x16x32(__builtin_bit_cast(bf16x8, __builtin_shufflevector(t2_[dt], t3_[dt], 0, 1, 2, 3, 4, 5, 6, 7)), ones8, acc[dt], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int qt = NQ4 * 4; qt < NKT; ++qt) {              // the last one to four tiles, one at a time (only the last can hold padded rows)
                    s16x4 msk;
#pragma unroll
                    for (int j = 0; j < 4; ++j) msk[j] = qt * 16 + 4 * g + j < S ? (short)PV_ONE16 : (short)0;
                    s16x4 oa[NDT];
                    tr_issue(Y1, qt, oa);
                    tr_wait(oa);
#pragma unroll
                    for (int dt = 0; dt < NDT; ++dt) acc[dt] = PV_MFMA_16x16x16(oa[dt], qt == NKT - 1 ? msk : ones, acc[dt], 0, 0, 0);
                }
                if (i16 == 0) {
                    float* o = dbp + (int64_t)b * 3 * D + h * DH + 4 * g;
#pragma unroll
                    for (int dt = 0; dt < NDT; ++dt) {
                        *reinterpret_cast<f32x4*>(o + D + dt * 16) = (f32x4){0.f, 0.f, 0.f, 0.f};
                        *reinterpret_cast<f32x4*>(o + 2 * D + dt * 16) = acc[dt];
                    }
                }
            }
        }
        PV_CSTAMP(6);
        if (!more) break;
        it = itn;
        par ^= 1;
    }
}

static int pv_attn_cu_count() {              // per device (the current one = the stream's)
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

template <int DH, int NKT>
static int pv_launch_attn_bwd5(const uint16_t* qkv, const uint16_t* dout, const uint16_t* att, const float* lse, uint16_t* dqkv, float* dbp, int64_t B, int S, int H,
                               float qscale, hipStream_t stream) {
    constexpr int DHP = (DH + 31) / 32 * 32;
    constexpr int lds = 4 * NKT * 16 * DHP * 2 + 4 * NKT * 16 * 4 + 16 * DH * 4 + 16 * 16 * DHP * 2;
    static_assert(lds <= 160 * 1024, "one workgroup per CU");
    static PvPerDevice attr_set;
    if (attr_set.first_use()) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(pv_attn_bwd5_kernel<DH, NKT>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    }
    const int cus = pv_attn_cu_count();
    if (cus <= 0) return PV_ERR_LAUNCH;
    const int64_t items = B * H;
    const unsigned grid = (unsigned)(items < cus ? items : cus);
    PV_LAUNCH((pv_attn_bwd5_kernel<DH, NKT>), dim3(grid), dim3(1024), lds, stream, qkv, dout, att, lse, dqkv, dbp, S, H, qscale, (int)B, (int)items);
    return pv_check_launch();
}

template <int DH>
static int pv_dispatch_attn_bwd5(const uint16_t* qkv, const uint16_t* dout, const uint16_t* att, const float* lse, uint16_t* dqkv, float* dbp, int64_t B, int S,
                                 int H, float qscale, hipStream_t s) {
    switch ((S + 15) / 16) {             // sequences of 129 .. 208 tokens: one tile per wave, at least three waves for the side work
#define PV_ATTN_CASE(N) case N: return pv_launch_attn_bwd5<DH, N>(qkv, dout, att, lse, dqkv, dbp, B, S, H, qscale, s);
        PV_ATTN_CASE(9) PV_ATTN_CASE(10) PV_ATTN_CASE(11) PV_ATTN_CASE(12) PV_ATTN_CASE(13)
#undef PV_ATTN_CASE
        default: return PV_ERR_UNSUPPORTED;
    }
}

extern "C" int pv_attention_bwd_lse_bf16(const uint16_t* qkv, const uint16_t* dout, const uint16_t* out, const float* lse, uint16_t* dqkv,
                                         float* dbias_partial, int64_t B, int64_t S, int64_t H, int64_t dh, float qscale, void* stream) {
    if (!qkv || !dout || !out || !lse || !dqkv || B <= 0 || S <= 0 || H <= 0 || dh <= 0) return PV_ERR_INVALID_ARG;
    if (((uintptr_t)qkv & 15) || ((uintptr_t)dout & 15) || ((uintptr_t)out & 15) || ((uintptr_t)dqkv & 15) || ((uintptr_t)lse & 3) || ((uintptr_t)dbias_partial & 15))
        return PV_ERR_INVALID_ARG;
    if (B * H > 0x7fffffff) return PV_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    switch (dh) {          // 129 <= S <= 208 at dh = 48 / 64; other shapes: pv_attention_bwd_bf16
        case 48: return pv_dispatch_attn_bwd5<48>(qkv, dout, out, lse, dqkv, dbias_partial, B, (int)S, (int)H, qscale, s);
        case 64: return pv_dispatch_attn_bwd5<64>(qkv, dout, out, lse, dqkv, dbias_partial, B, (int)S, (int)H, qscale, s);
        default: return PV_ERR_UNSUPPORTED;
    }
}

template <int DH>
static int pv_dispatch_attn_bwd(const uint16_t* qkv, const uint16_t* dout, uint16_t* dqkv, float* dbp, int64_t B, int S, int H, float qscale, hipStream_t s) {
    switch ((S + 15) / 16) {
#if PV_ABW_V2
#define PV_ATTN_CASE(N) case N: return pv_launch_attn_bwd2<DH, N>(qkv, dout, dqkv, dbp, B, S, H, qscale, s);
#else
#define PV_ATTN_CASE(N) case N: return pv_launch_attn_bwd<DH, N>(qkv, dout, dqkv, dbp, B, S, H, qscale, s);
#endif
        PV_ATTN_CASE(1) PV_ATTN_CASE(2) PV_ATTN_CASE(3) PV_ATTN_CASE(4) PV_ATTN_CASE(5) PV_ATTN_CASE(6) PV_ATTN_CASE(7)
        PV_ATTN_CASE(8) PV_ATTN_CASE(9) PV_ATTN_CASE(10) PV_ATTN_CASE(11) PV_ATTN_CASE(12) PV_ATTN_CASE(13)
#undef PV_ATTN_CASE
        default: break;
    }
    if constexpr (DH == 32) {                  // 64-byte LDS rows: twice the sequence fits
        switch ((S + 15) / 16) {
#define PV_ATTN_CASE(N) case N: return pv_launch_attn_bwd<DH, N>(qkv, dout, dqkv, dbp, B, S, H, qscale, s);
            PV_ATTN_CASE(14) PV_ATTN_CASE(15) PV_ATTN_CASE(16) PV_ATTN_CASE(17) PV_ATTN_CASE(18) PV_ATTN_CASE(19) PV_ATTN_CASE(20)
            PV_ATTN_CASE(21) PV_ATTN_CASE(22) PV_ATTN_CASE(23) PV_ATTN_CASE(24) PV_ATTN_CASE(25) PV_ATTN_CASE(26)
#undef PV_ATTN_CASE
            default: break;
        }
    }
    return PV_ERR_UNSUPPORTED;
}

extern "C" int pv_attention_bwd_bf16(const uint16_t* qkv, const uint16_t* dout, uint16_t* dqkv, float* dbias_partial, int64_t B, int64_t S,
                                     int64_t H, int64_t dh, float qscale, void* stream) {
    if (!qkv || !dout || !dqkv || B <= 0 || S <= 0 || H <= 0 || dh <= 0) return PV_ERR_INVALID_ARG;
    if (((uintptr_t)qkv & 15) || ((uintptr_t)dout & 15) || ((uintptr_t)dqkv & 15)) return PV_ERR_INVALID_ARG;
    if (B * H > 0x7fffffff) return PV_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    switch (dh) {          // S <= 208 at dh = 48 / 64, S <= 416 at dh = 32
        case 32: return pv_dispatch_attn_bwd<32>(qkv, dout, dqkv, dbias_partial, B, (int)S, (int)H, qscale, s);
        case 48: return pv_dispatch_attn_bwd<48>(qkv, dout, dqkv, dbias_partial, B, (int)S, (int)H, qscale, s);
        case 64: return pv_dispatch_attn_bwd<64>(qkv, dout, dqkv, dbias_partial, B, (int)S, (int)H, qscale, s);
        default: return PV_ERR_UNSUPPORTED;
    }
}

// ---- attention for the FIRST nq ROWS of every image only (queries), against all S keys -----------------------------------------------
// The last encoder block of a ViT: only the class-token rows of its output are consumed (models/vit.py:242-246), so its attention needs
// q for those rows alone but k, v of every token.  HBM-bound: K and V of the head are read once (2 * S * dh operands), nothing is staged.
// One wave per (image, head, query row).  CPL lanes share a key (one 16-byte chunk of the head dimension each), 64 / CPL keys per step;
// every lane group runs its own online softmax over its keys, the groups are merged at the end.  All arithmetic fp32 (p is not rounded).
template <int DH>
__global__ __launch_bounds__(256) void pv_attn_rows_kernel(const uint16_t* __restrict__ q, int64_t ldq, const uint16_t* __restrict__ kv, int64_t ldkv,
                                                           uint16_t* __restrict__ out, int64_t ldo, int S, int H, int nq, int64_t total, uint32_t* flag) {
    constexpr int NCH = DH / 8;
    constexpr int CPL = NCH <= 4 ? 4 : (NCH <= 8 ? 8 : 16);
    constexpr int KPI = 64 / CPL;
    const int lane = threadIdx.x & 63;
    const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= total) return;
    const int qi = (int)(w % nq);
    const int64_t bh = w / nq;
    const int h = (int)(bh % H);
    const int64_t b = bh / H;
    const int c = lane % CPL, g = lane / CPL;
    const bool act = c < NCH;
    const int cc = act ? c : 0;
    const int64_t vcol = (int64_t)H * DH;

    float qf[8];
    {
        const u32x4 qv = *reinterpret_cast<const u32x4*>(q + (b * nq + qi) * ldq + h * DH + cc * 8);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            qf[2 * i] = act ? pv_unpack_lo(qv[i]) : 0.f;
            qf[2 * i + 1] = act ? pv_unpack_hi(qv[i]) : 0.f;
        }
    }
    const uint16_t* kbase = kv + (b * S) * ldkv + h * DH + cc * 8;
    float m = -INFINITY, l = 0.f, o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = 0.f;
#pragma unroll 4
    for (int j0 = 0; j0 < S; j0 += KPI) {
        const int key = j0 + g;
        const bool valid = key < S;
        const int64_t off = (int64_t)(valid ? key : S - 1) * ldkv;
        const u32x4 kk = *reinterpret_cast<const u32x4*>(kbase + off);
        const u32x4 vv = *reinterpret_cast<const u32x4*>(kbase + off + vcol);
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) s += qf[2 * i] * pv_unpack_lo(kk[i]) + qf[2 * i + 1] * pv_unpack_hi(kk[i]);
#pragma unroll
        for (int d = 1; d < CPL; d <<= 1) s += __shfl_xor(s, d, 64);
        if (valid) {
            const float mn = fmaxf(m, s);
            const float sc = __expf(m - mn), pr = __expf(s - mn);        // first key of the group: m = -inf, sc = 0
            l = l * sc + pr;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                o[2 * i] = o[2 * i] * sc + pr * pv_unpack_lo(vv[i]);
                o[2 * i + 1] = o[2 * i + 1] * sc + pr * pv_unpack_hi(vv[i]);
            }
            m = mn;
        }
    }
    // merge the KPI lane groups (a group that saw no key keeps m = -inf, l = 0 and contributes nothing)
#pragma unroll
    for (int d = CPL; d < 64; d <<= 1) {
        const float m2 = __shfl_xor(m, d, 64), l2 = __shfl_xor(l, d, 64);
        const float mn = fmaxf(m, m2);
        const float a = m == -INFINITY ? 0.f : __expf(m - mn), a2 = m2 == -INFINITY ? 0.f : __expf(m2 - mn);
        l = l * a + l2 * a2;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = o[i] * a + __shfl_xor(o[i], d, 64) * a2;
        m = mn;
    }
    pv_score_guard(m, flag);           // after the merge every lane holds the row's maximum
    if (g == 0 && act) {
        const float inv = 1.0f / l;
        u32x4 r;
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = pv_pack_bf16x2(o[2 * i] * inv, o[2 * i + 1] * inv);
        *reinterpret_cast<u32x4*>(out + (b * nq + qi) * ldo + h * DH + c * 8) = r;
    }
}

template <int DH>
static int pv_launch_attn_rows(const uint16_t* q, int64_t ldq, const uint16_t* kv, int64_t ldkv, uint16_t* out, int64_t ldo, int64_t B, int S, int nq,
                               int H, uint32_t* flag, hipStream_t stream) {
    const int64_t total = B * H * nq;
    PV_LAUNCH(pv_attn_rows_kernel<DH>, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, stream, q, ldq, kv, ldkv, out, ldo, S, H, nq, total, flag);
    return pv_check_launch();
}

extern "C" int pv_attention_rows_bf16(const uint16_t* q, int64_t ldq, const uint16_t* kv, int64_t ldkv, uint16_t* out, int64_t ldo, int64_t B,
                                      int64_t S, int64_t nq, int64_t H, int64_t dh, uint32_t* range_flag, void* stream) {
    if (!q || !kv || !out || B <= 0 || S <= 0 || nq <= 0 || H <= 0 || dh <= 0 || ((uintptr_t)range_flag & 3)) return PV_ERR_INVALID_ARG;
    uint32_t* const flag = range_flag;
    if (((uintptr_t)q & 15) || ((uintptr_t)kv & 15) || ((uintptr_t)out & 15) || (ldq & 7) || (ldkv & 7) || (ldo & 7)) return PV_ERR_INVALID_ARG;
    if (ldq < H * dh || ldo < H * dh || ldkv < 2 * H * dh) return PV_ERR_INVALID_ARG;
    if (B * H * nq > 0x7fffffffLL * 4 || S > 0x3fffffff) return PV_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    switch (dh) {
        case 32: return pv_launch_attn_rows<32>(q, ldq, kv, ldkv, out, ldo, B, (int)S, (int)nq, (int)H, flag, s);
        case 48: return pv_launch_attn_rows<48>(q, ldq, kv, ldkv, out, ldo, B, (int)S, (int)nq, (int)H, flag, s);
        case 64: return pv_launch_attn_rows<64>(q, ldq, kv, ldkv, out, ldo, B, (int)S, (int)nq, (int)H, flag, s);
        case 80: return pv_launch_attn_rows<80>(q, ldq, kv, ldkv, out, ldo, B, (int)S, (int)nq, (int)H, flag, s);
        case 96: return pv_launch_attn_rows<96>(q, ldq, kv, ldkv, out, ldo, B, (int)S, (int)nq, (int)H, flag, s);
        case 128: return pv_launch_attn_rows<128>(q, ldq, kv, ldkv, out, ldo, B, (int)S, (int)nq, (int)H, flag, s);
        default: return PV_ERR_UNSUPPORTED;
    }
}

// Backward of pv_attn_rows_kernel for ONE query row per image (nq = 1: the class token).  One wave per (image, head):
//   pass 1  (m, l) of the softmax over the keys (same online form as the forward), delta = dout . out
//   pass 2  per key j: p = exp(s_j - m) / l, dp = dout . v_j, ds = p (dp - delta);  dv_j = p dout,  dk_j = ds q  (written straight to the key's row
//           of dkv: no reduction, every key has one query),  dq += ds k_j (reduced over the lane groups at the end, times qscale: the
//           gradient of the UNSCALED in-projection output, like pv_attn_bwd_kernel).
// HBM-bound: k | v read twice (the second pass from the L2 / MALL for most heads), dk | dv written once.
template <int DH>
__global__ __launch_bounds__(256) void pv_attn_rows_bwd_kernel(const uint16_t* __restrict__ q, int64_t ldq, const uint16_t* __restrict__ kv, int64_t ldkv,
                                                               const uint16_t* __restrict__ out, int64_t ldo, const uint16_t* __restrict__ dout, int64_t lddo,
                                                               uint16_t* __restrict__ dq, int64_t lddq, uint16_t* __restrict__ dkv, int64_t lddkv,
                                                               int S, int H, float qscale, int64_t total) {
    constexpr int NCH = DH / 8;
    constexpr int CPL = NCH <= 4 ? 4 : (NCH <= 8 ? 8 : 16);
    constexpr int KPI = 64 / CPL;
    const int lane = threadIdx.x & 63;
    const int64_t bh = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (bh >= total) return;
    const int h = (int)(bh % H);
    const int64_t b = bh / H;
    const int c = lane % CPL, g = lane / CPL;
    const bool act = c < NCH;
    const int cc = act ? c : 0;
    const int64_t vcol = (int64_t)H * DH;
    const int col = h * DH + cc * 8;

    float qf[8], df[8];
    float delta = 0.f;
    {
        const u32x4 qv = *reinterpret_cast<const u32x4*>(q + b * ldq + col);
        const u32x4 dv = *reinterpret_cast<const u32x4*>(dout + b * lddo + col);
        const u32x4 ov = *reinterpret_cast<const u32x4*>(out + b * ldo + col);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            qf[2 * i] = act ? pv_unpack_lo(qv[i]) : 0.f;
            qf[2 * i + 1] = act ? pv_unpack_hi(qv[i]) : 0.f;
            df[2 * i] = act ? pv_unpack_lo(dv[i]) : 0.f;
            df[2 * i + 1] = act ? pv_unpack_hi(dv[i]) : 0.f;
            delta += df[2 * i] * pv_unpack_lo(ov[i]) + df[2 * i + 1] * pv_unpack_hi(ov[i]);
        }
#pragma unroll
        for (int d = 1; d < CPL; d <<= 1) delta += __shfl_xor(delta, d, 64);
    }
    const uint16_t* kbase = kv + (b * S) * ldkv + col;
    uint16_t* dbase = dkv + (b * S) * lddkv + col;
    float m = -INFINITY, l = 0.f;
#pragma unroll 4
    for (int j0 = 0; j0 < S; j0 += KPI) {
        const int key = j0 + g;
        const bool valid = key < S;
        const u32x4 kk = *reinterpret_cast<const u32x4*>(kbase + (int64_t)(valid ? key : S - 1) * ldkv);
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) s += qf[2 * i] * pv_unpack_lo(kk[i]) + qf[2 * i + 1] * pv_unpack_hi(kk[i]);
#pragma unroll
        for (int d = 1; d < CPL; d <<= 1) s += __shfl_xor(s, d, 64);
        if (valid) {
            const float mn = fmaxf(m, s);
            l = l * __expf(m - mn) + __expf(s - mn);
            m = mn;
        }
    }
#pragma unroll
    for (int d = CPL; d < 64; d <<= 1) {
        const float m2 = __shfl_xor(m, d, 64), l2 = __shfl_xor(l, d, 64);
        const float mn = fmaxf(m, m2);
        l = l * (m == -INFINITY ? 0.f : __expf(m - mn)) + l2 * (m2 == -INFINITY ? 0.f : __expf(m2 - mn));
        m = mn;
    }
    const float inv = 1.0f / l;
    float dqa[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) dqa[i] = 0.f;
#pragma unroll 4
    for (int j0 = 0; j0 < S; j0 += KPI) {
        const int key = j0 + g;
        const bool valid = key < S;
        const int64_t off = (int64_t)(valid ? key : S - 1) * ldkv;
        const u32x4 kk = *reinterpret_cast<const u32x4*>(kbase + off);
        const u32x4 vv = *reinterpret_cast<const u32x4*>(kbase + off + vcol);
        float s = 0.f, dp = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            s += qf[2 * i] * pv_unpack_lo(kk[i]) + qf[2 * i + 1] * pv_unpack_hi(kk[i]);
            dp += df[2 * i] * pv_unpack_lo(vv[i]) + df[2 * i + 1] * pv_unpack_hi(vv[i]);
        }
#pragma unroll
        for (int d = 1; d < CPL; d <<= 1) {
            s += __shfl_xor(s, d, 64);
            dp += __shfl_xor(dp, d, 64);
        }
        const float pr = __expf(s - m) * inv;
        const float ds = pr * (dp - delta);
        if (valid && act) {
            u32x4 rk, rv;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                rk[i] = pv_pack_bf16x2(ds * qf[2 * i], ds * qf[2 * i + 1]);
                rv[i] = pv_pack_bf16x2(pr * df[2 * i], pr * df[2 * i + 1]);
            }
            *reinterpret_cast<u32x4*>(dbase + (int64_t)key * lddkv) = rk;
            *reinterpret_cast<u32x4*>(dbase + (int64_t)key * lddkv + vcol) = rv;
        }
        if (valid) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                dqa[2 * i] += ds * pv_unpack_lo(kk[i]);
                dqa[2 * i + 1] += ds * pv_unpack_hi(kk[i]);
            }
        }
    }
#pragma unroll
    for (int d = CPL; d < 64; d <<= 1) {
#pragma unroll
        for (int i = 0; i < 8; ++i) dqa[i] += __shfl_xor(dqa[i], d, 64);
    }
    if (g == 0 && act) {
        u32x4 r;
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = pv_pack_bf16x2(dqa[2 * i] * qscale, dqa[2 * i + 1] * qscale);
        *reinterpret_cast<u32x4*>(dq + b * lddq + col) = r;
    }
}

template <int DH>
static int pv_launch_attn_rows_bwd(const uint16_t* q, int64_t ldq, const uint16_t* kv, int64_t ldkv, const uint16_t* out, int64_t ldo, const uint16_t* dout,
                                   int64_t lddo, uint16_t* dq, int64_t lddq, uint16_t* dkv, int64_t lddkv, int64_t B, int S, int H, float qscale,
                                   hipStream_t stream) {
    const int64_t total = B * H;
    PV_LAUNCH(pv_attn_rows_bwd_kernel<DH>, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, stream, q, ldq, kv, ldkv, out, ldo, dout, lddo, dq, lddq, dkv,
              lddkv, S, H, qscale, total);
    return pv_check_launch();
}

extern "C" int pv_attention_rows_bwd_bf16(const uint16_t* q, int64_t ldq, const uint16_t* kv, int64_t ldkv, const uint16_t* out, int64_t ldo,
                                          const uint16_t* dout, int64_t lddo, uint16_t* dq, int64_t lddq, uint16_t* dkv, int64_t lddkv, int64_t B,
                                          int64_t S, int64_t nq, int64_t H, int64_t dh, float qscale, void* stream) {
    if (!q || !kv || !out || !dout || !dq || !dkv || B <= 0 || S <= 0 || nq <= 0 || H <= 0 || dh <= 0) return PV_ERR_INVALID_ARG;
    if (((uintptr_t)q | (uintptr_t)kv | (uintptr_t)out | (uintptr_t)dout | (uintptr_t)dq | (uintptr_t)dkv) & 15) return PV_ERR_INVALID_ARG;
    if ((ldq | ldkv | ldo | lddo | lddq | lddkv) & 7) return PV_ERR_INVALID_ARG;
    if (ldq < H * dh || ldo < H * dh || lddo < H * dh || lddq < H * dh || ldkv < 2 * H * dh || lddkv < 2 * H * dh) return PV_ERR_INVALID_ARG;
    if (nq != 1 || B * H > 0x7fffffffLL * 4 || S > 0x3fffffff) return PV_ERR_UNSUPPORTED;      // one query row per image (the class token)
    hipStream_t s = (hipStream_t)stream;
#define PV_ROWS_BWD(N) case N: return pv_launch_attn_rows_bwd<N>(q, ldq, kv, ldkv, out, ldo, dout, lddo, dq, lddq, dkv, lddkv, B, (int)S, (int)H, qscale, s);
    switch (dh) {
        PV_ROWS_BWD(32) PV_ROWS_BWD(48) PV_ROWS_BWD(64) PV_ROWS_BWD(80) PV_ROWS_BWD(96) PV_ROWS_BWD(128)
        default: return PV_ERR_UNSUPPORTED;
    }
#undef PV_ROWS_BWD
}

extern "C" int pv_attention_f32_split(const float* qkv, uint16_t* out, int64_t B, int64_t S, int64_t H, int64_t dh, void* stream) {
    if (!qkv || !out || B <= 0 || S <= 0 || H <= 0 || dh <= 0) return PV_ERR_INVALID_ARG;
    if (((uintptr_t)qkv & 15) || ((uintptr_t)out & 7)) return PV_ERR_INVALID_ARG;
    if (B * H > 0x7fffffff) return PV_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    switch (dh) {
        case 32: return pv_dispatch_attn_f32<32>(qkv, out, B, (int)S, (int)H, s);
        case 48: return pv_dispatch_attn_f32<48>(qkv, out, B, (int)S, (int)H, s);
        case 64: return pv_dispatch_attn_f32<64>(qkv, out, B, (int)S, (int)H, s);
        default: return PV_ERR_UNSUPPORTED;
    }
}

extern "C" int pv_attention_lse_bf16(const uint16_t* qkv, uint16_t* out, float* lse, int64_t B, int64_t S, int64_t H, int64_t dh, uint32_t* range_flag, void* stream) {
    if (!qkv || !out || !lse || B <= 0 || S <= 0 || H <= 0 || dh <= 0) return PV_ERR_INVALID_ARG;
    if (((uintptr_t)qkv & 15) || ((uintptr_t)out & 15) || ((uintptr_t)lse & 3) || ((uintptr_t)range_flag & 3)) return PV_ERR_INVALID_ARG;
    if (B * H > 0x7fffffff || S > 416) return PV_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    switch (dh) {
        case 32: return pv_dispatch_attn<32>(qkv, out, B, (int)S, (int)H, range_flag, s, lse);
        case 48: return pv_dispatch_attn<48>(qkv, out, B, (int)S, (int)H, range_flag, s, lse);
        case 64: return pv_dispatch_attn<64>(qkv, out, B, (int)S, (int)H, range_flag, s, lse);
        default: return PV_ERR_UNSUPPORTED;
    }
}

extern "C" int pv_attention_bf16(const uint16_t* qkv, uint16_t* out, int64_t B, int64_t S, int64_t H, int64_t dh, uint32_t* range_flag, void* stream) {
    if (!qkv || !out || B <= 0 || S <= 0 || H <= 0 || dh <= 0) return PV_ERR_INVALID_ARG;
    if (((uintptr_t)qkv & 15) || ((uintptr_t)out & 15) || ((uintptr_t)range_flag & 3)) return PV_ERR_INVALID_ARG;
    uint32_t* const flag = range_flag;
    if (B * H > 0x7fffffff || S > 0x3fffffff) return PV_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    switch (dh) {
        case 32: return pv_dispatch_attn<32>(qkv, out, B, (int)S, (int)H, flag, s);
        case 48: return pv_dispatch_attn<48>(qkv, out, B, (int)S, (int)H, flag, s);
        case 64: return pv_dispatch_attn<64>(qkv, out, B, (int)S, (int)H, flag, s);
        // wider heads (ViT-H: 80, 96, 128): the streaming kernel for every sequence length
        case 80: return pv_launch_attn_stream<80>(qkv, out, B, (int)S, (int)H, flag, s);
        case 96: return pv_launch_attn_stream<96>(qkv, out, B, (int)S, (int)H, flag, s);
        case 128: return pv_launch_attn_stream<128>(qkv, out, B, (int)S, (int)H, flag, s);
        default: return PV_ERR_UNSUPPORTED;
    }
}
