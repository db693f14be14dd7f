/* peekvit_hip.h - C ABI of libpeekvit_hip.so: the MI355X (gfx950) kernels behind peekvit's ViT
 * encoder hot path (SURVEY.md section 8b "inner boundary").
 *
 * The reference (alessiodevoto/peekvit) has no FFI: its arithmetic is dispatched by stock torch.nn
 * modules to ATen kernels.  Each entry point below replaces the ATen op(s) behind ONE reference call
 * site (cited as <reference file>:<line>) and is what a binding for that call site would bind
 * (INTEGRATION.md shows the ctypes stub).
 *
 * Conventions (all entry points):
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless said otherwise;
 *   - stateless, re-entrant: never allocates, never synchronises, never owns memory; the caller passes
 *     outputs/workspaces and the hipStream_t (as void*) to launch on; safe to capture in a hipGraph;
 *   - returns PV_OK (0) or a negative PV_ERR_* code; pv_error_string() explains it;
 *   - "bf16" buffers are uint16_t bit patterns (round-to-nearest-even from fp32);
 *   - row-major everywhere; `ld*` are leading dimensions in ELEMENTS.
 */
#ifndef PEEKVIT_HIP_H
#define PEEKVIT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PV_OK 0
#define PV_ERR_INVALID_ARG (-1)   /* null pointer, non-positive size, misaligned leading dimension */
#define PV_ERR_UNSUPPORTED (-2)   /* shape outside what the kernels are built for (see each entry) */
#define PV_ERR_LAUNCH (-3)        /* hipGetLastError() != hipSuccess after the launch */

/* ABI version (bumped on any signature change) and build target ("gfx950"). */
int pv_version(void);
const char* pv_arch(void);
/* 16-bit operand type this library was built for: 0 = bf16 (libpeekvit_hip.so), 1 = IEEE fp16 (libpeekvit_hip_f16.so, built
 * from the same sources with -DPV_OPERAND_F16).  Every "bf16" tensor in the signatures below is of that type. */
int pv_operand_type(void);
const char* pv_error_string(int code);

/* Scratch sizing (ABI v8; SURVEY.md section 8b lists it in the minimum export set).  Every entry point that needs scratch takes it from the
 * caller (`ws`, `colsum_ws`, `pv_gemm_args.colsum_partial`, the split-K slices); this returns the number of BYTES the named use needs for the
 * given sizes, so that a C caller does not have to copy formulas out of the comments below (peekvit_amd/ops.py sizes its buffers with it).
 * dims / ndims per `use`:
 *   PV_WS_TRANSPOSE_COLSUM    {R, C, ldd}      pv_transpose_bf16(colsum_ws)
 *   PV_WS_COLSUM              {R, C}           pv_colsum_f32(ws)
 *   PV_WS_LAYERNORM_BWD       {rows, D}        pv_layernorm_bwd / pv_layernorm_bwd_masked(ws)
 *   PV_WS_GEMM_COLSUM_PARTIAL {M, N}           pv_gemm_args.colsum_partial (PV_EPI_GELU_GRAD_BF16 on the 256-row tile kernel)
 *   PV_WS_GEMM_SPLITK         {M, N, ksplit}   the fp32 [ksplit, M, N] slices of pv_gemm_bf16 / pv_gemm_tn_bf16 with ksplit > 1
 * Returns a negative PV_ERR_* for an unknown use, a wrong ndims or a non-positive size. */
#define PV_WS_TRANSPOSE_COLSUM 1
#define PV_WS_COLSUM 2
#define PV_WS_LAYERNORM_BWD 3
#define PV_WS_GEMM_COLSUM_PARTIAL 4
#define PV_WS_GEMM_SPLITK 5
int64_t pv_workspace_size(int use, const int64_t* dims, int ndims);

/* fp32 -> bf16 cast of a contiguous buffer (weights packing at load time). n elements. */
int pv_cast_f32_bf16(const float* src, uint16_t* dst, int64_t n, void* stream);

/* Patch gather ("im2col") for the stride-P, kernel-P convolution of
 *   models/vit.py:212  `x = self.conv_proj(x)`  (+ reshape/permute 214-220).
 * x: fp32 [B,C,H,W] contiguous NCHW.  cols: bf16 [B*(H/P)*(W/P), C*P*P], K order (c,kh,kw) = the
 * conv weight's own layout, so conv_proj.weight viewed as [D, C*P*P] is the GEMM weight. */
/* range_flag (ABI v5, optional, device uint32_t): operand-range guard of the fp16-operand build - the kernel ORs 1 into it
 * when a value it packs is not finite in fp16 (|v| > 65504); the bf16 build never writes it.  See pv_gemm_args.range_flag. */
int pv_im2col_bf16(const float* x, uint16_t* cols, int64_t B, int64_t C, int64_t H, int64_t W, int64_t P,
                   uint32_t* range_flag, void* stream);

/* models/vit.py:203-222 WITHOUT a materialised patch matrix (ABI v10): tokens[b, row_off + i, :] = patch_i(img[b]) . W^T + bias + pos[row_off + i, :] in one
 * launch - a 128 x 128 tile GEMM whose activation operand is gathered from the image while it is staged.  img fp32 [B,C,R,R] (NCHW); W 16-bit [D, C*P*P] =
 * conv_proj.weight in its own layout; pos fp32 [S, D]; tokens fp32 [B, S, D] (rows [row_off, row_off + (R/P)^2) of every image are written).  R % P == 0,
 * P % 8 == 0, C*P*P % 64 == 0.  Bit-identical to pv_im2col_bf16 + pv_gemm_bf16(PV_EPI_BIAS_POS_F32); the faster form for D <= 512 (each 128-column tile
 * gathers the pixels again).  range_flag as pv_im2col_bf16. */
int pv_patch_embed_f32(const float* img, const uint16_t* W, const float* bias, const float* pos, float* tokens, int64_t B, int64_t C, int64_t R,
                       int64_t P, int64_t D, int64_t S, int64_t row_off, uint32_t* range_flag, void* stream);

/* The same patch gather straight from the DataLoader's RAW image: x uint8 [B,H,W,3] (NHWC); ToTensor + Normalize of
 *   data/imagenette.py:73 (x/255, then (x - mean[c]) / std[c], fp32, this op order) are applied per element, so `cols` is
 * bit-identical to pv_im2col_bf16 of the normalised fp32 NCHW tensor at a quarter of the input bytes.  P % 8 == 0. */
int pv_im2col_u8_bf16(const uint8_t* x, uint16_t* cols, int64_t B, int64_t H, int64_t W, int64_t P, float mean0,
                      float mean1, float mean2, float std0, float std1, float std2, void* stream);

/* Token prologue rows that are not produced by the patch GEMM:
 *   models/vit.py:230-236 (cat registers, cat class tokens) + models/vit.py:92 (+ pos_embedding),
 *   models/residualvit.py:566-568,345 (append learnable_budget_token_1 * budget, no pos-embedding).
 * tokens: fp32 [B,S_total,D].  Writes rows [0,n_special) = special[r] + pos[r] and, when
 * budget_token != NULL, row S_total-1 = budget_token * budget.  special: [n_special,D] (class tokens
 * then registers), pos: [>=n_special, D]. */
int pv_token_prologue(float* tokens, const float* special, const float* pos, const float* budget_token,
                      float budget, int64_t B, int64_t S_total, int64_t D, int64_t n_special, void* stream);

/* LayerNorm over the last dim, fp32 in -> bf16 out (the GEMM operand):
 *   nn.LayerNorm call sites models/vit.py:48,53 (eps 1e-5), models/residualvit.py:252,258 (eps 1e-6,
 *   followed by `mask *`: row_scale, may be NULL).  x: [rows, D] with row stride ldx; out: [rows, D]
 * contiguous.  D % 4 == 0, D <= 4096. */
int pv_layernorm_bf16(const float* x, int64_t ldx, const float* gamma, const float* beta, const float* row_scale,
                      uint16_t* out, int64_t rows, int64_t D, float eps, void* stream);

/* Epilogues of pv_gemm_bf16 (what is fused behind the MFMA accumulator). */
#define PV_EPI_BIAS_BF16 0      /* out bf16 = (acc + bias[n]) * (n < qcols ? qscale : 1)   [QKV in-proj]      */
#define PV_EPI_BIAS_GELU_BF16 1 /* out bf16 = gelu_erf(acc + bias[n])                       [MLP fc1]          */
#define PV_EPI_BIAS_RES_F32 2   /* out f32  = res[m,n] + row_scale[m]*(acc + bias[n])       [out-proj, fc2]    */
#define PV_EPI_BIAS_POS_F32 3   /* out f32 row (m/rpi)*rpo+row_off+(m%rpi) = acc+bias[n]+pos[(row_off+m%rpi),n] [patch embed] */

/* precision mode "bf16x3" (split operands concatenated along K): */
#define PV_EPI_BIAS_F32 4             /* out f32 = (acc + bias[n]) * (n < qcols ? qscale : 1)           [in-proj, fp32 q|k|v]  */
#define PV_EPI_BIAS_GELU_SPLIT_BF16 5 /* out bf16 [M, 3N] = [hi | lo | hi] of gelu_erf(acc + bias[n]), ldo >= 3N [MLP fc1]    */
/* training path (ABI v4): */
#define PV_EPI_BIAS_GELU_PAIR_BF16  6   /* out bf16 [M, 2N] (ldo >= 2N): cols [0,N) = gelu(acc+bias), cols [N,2N) = gelu'(acc+bias) (saved for backward; ABI v10 -
                                           * v4-v9 saved acc+bias itself and PV_EPI_GELU_GRAD_BF16 evaluated gelu' from it) */
#define PV_EPI_GELU_GRAD_BF16       7   /* out bf16 = (acc+bias) * d[m][n], d = the saved gelu' plane of PV_EPI_BIAS_GELU_PAIR_BF16: a 16-bit matrix passed in `res` (row stride ldr elements) */

typedef struct pv_gemm_args {
    /* ABI v7: sizeof(pv_gemm_args) as the CALLER's binding knows it - the first field, so that it can be read whatever the caller's
     * struct length is.  pv_gemm_bf16 / pv_gemm_tn_bf16 / pv_gemm_tile_rows return PV_ERR_INVALID_ARG when it differs from the
     * library's own size (pv_gemm_args_size()): a binding made from an older header hands over a shorter struct, and the library must
     * not read the fields it lacks (round 2 appended res_scaled without this check). */
    uint64_t struct_size;
    const uint16_t* A;       /* bf16 [M,K], row stride lda            (activations)                    */
    const uint16_t* W;       /* bf16 [N,K], row stride ldw            (nn.Linear weight layout (out,in)) */
    const float* bias;       /* fp32 [N] or NULL                                                        */
    void* out;               /* bf16 or fp32 [*, N], row stride ldo                                     */
    const float* res;        /* fp32 residual [M, *], row stride ldr  (PV_EPI_BIAS_RES_F32; may alias out) */
    const float* row_scale;  /* fp32 [M] or NULL                      (PV_EPI_BIAS_RES_F32)              */
    const float* pos;        /* fp32 [rows_per_img_out, N]            (PV_EPI_BIAS_POS_F32)              */
    int64_t M, N, K;
    int64_t lda, ldw, ldo, ldr;
    int64_t rows_per_img_in;   /* rpi: Np                              (PV_EPI_BIAS_POS_F32)             */
    int64_t rows_per_img_out;  /* rpo: S_total                                                          */
    int64_t row_off;           /* first patch row inside an image's token block (= n_special)           */
    int64_t qcols;             /* PV_EPI_BIAS_BF16: columns [0,qcols) are multiplied by qscale          */
    float qscale;
    int32_t epilogue;          /* PV_EPI_*                                                              */
    /* optional fused LayerNorm of the finished output rows (PV_EPI_BIAS_RES_F32 only, ABI v2): when ln_out != NULL the
     * kernel also writes ln_out[m,:] = bf16(LayerNorm(out[m,:]; ln_gamma, ln_beta, ln_eps) * ln_row_scale[m]) -
     * bit-identical to pv_layernorm_bf16 on `out`.  Replaces models/vit.py:51+53 (residual, then ln_2) and
     * models/vit.py:55 + the next block's :48.  Needs N % 256 == 0, N <= 4096, K % 128 == 0, ldo == N. */
    const float* ln_gamma;     /* fp32 [N]                                                              */
    const float* ln_beta;      /* fp32 [N]                                                              */
    const float* ln_row_scale; /* fp32 [M] or NULL                                                      */
    uint16_t* ln_out;          /* bf16 [M,N] contiguous, or NULL (no fused LayerNorm)                   */
    float ln_eps;
    /* split-K for weight gradients (ABI v4): ksplit > 1 with PV_EPI_BIAS_F32 makes the launch compute `ksplit` partial
     * products over K slices of K/ksplit (K % (64*ksplit) == 0) into out[t] = out + t*M*ldo (fp32), bias added by slice 0;
     * pv_sum_slices_f32 reduces them.  dW = dY^T . X (train.py:118 loss.backward()) = this GEMM on transposed activations. */
    int32_t ksplit;
    /* PV_EPI_GELU_GRAD_BF16 only, optional: fp32 [ceil(M/256), N] receives the column sums of the stored output values per
     * 256-row block (summed over the blocks = the fc1 bias gradient).  Served by the 256-row tile kernel only: ask
     * pv_gemm_tile_rows() first and pass NULL when it answers 128. */
    float* colsum_partial;
    /* LayerNorm folding (opt-in; 256-row tile kernel only).  Producer, PV_EPI_BIAS_RES_F32: x16_out bf16 [M, N] (row stride N)
     * receives the 16-bit copy of the output rows, rowstat_out fp32 [ceil(N/256), M, 2] the (sum, sum of squares) of each row's
     * segment per column tile.  Consumer, PV_EPI_BIAS_BF16 / PV_EPI_BIAS_GELU_BF16 with bias == NULL: A is that copy, W is
     * gamma (.) W, and  acc <- fold_stat[m].rstd * (acc - fold_stat[m].mean * fold_c1[n]) + fold_c2[n]  precedes the epilogue
     * (fold_stat fp32 [M, 2] from pv_rowstat_finalize; fold_c1[n] = sum_k W'[n,k]; fold_c2[n] = sum_k beta[k] W[n,k] + bias[n]). */
    uint16_t* x16_out;
    float* rowstat_out;
    const float* fold_stat;
    const float* fold_c1;
    const float* fold_c2;
    /* Operand-range guard (ABI v5, optional): device uint32_t the fp16-operand build ORs 1 into when a 16-bit OUTPUT value of
     * this launch (PV_EPI_BIAS_BF16, PV_EPI_BIAS_GELU_BF16, x16_out) is not finite in fp16, i.e. |v| > 65504 - where the bf16
     * build would still be in range.  The caller zeroes it before a forward and reads it afterwards (peekvit_amd.engine mode
     * "auto": repeat that forward on the bf16 library).  The bf16 build ignores it. */
    uint32_t* range_flag;
    /* RankViT row norms fused into the producer (ABI v5, optional; PV_EPI_BIAS_RES_F32, 256-row tile kernel only - ask
     * pv_gemm_tile_rows): rowsq_out fp32 [ceil(N/256), M] receives, per column tile, the sum of squares of each finished output
     * row's segment.  models/rankvit.py:63 `torch.norm(input, dim=-1)` of the NEXT block is then sqrt(sum over tiles)
     * (pv_rank_topk_partials) and the separate pass over the tokens (pv_token_norm) disappears. */
    float* rowsq_out;
    /* ResidualViT (ABI v6, PV_EPI_BIAS_RES_F32 with row_scale, 256- / 128-row tile kernels): nonzero = the residual row is scaled too,
     * out = row_scale[m] * (res + acc + bias) - models/residualvit.py:226,254 with res = the UNMASKED tokens, so the masked copy of the
     * tokens (mask * img) never has to be written: masked + mask * branch = mask * (img + branch). */
    int32_t res_scaled;
} pv_gemm_args;

/* sizeof(pv_gemm_args) of THIS library build (ABI v7): a binding asserts it equals its own struct's size when it loads the library. */
uint64_t pv_gemm_args_size(void);

/* (mean, rstd) per row from the producer's partial sums: partials fp32 [tiles, rows, 2] -> stat fp32 [rows, 2]; D = row length.
 * range_flag (ABI v7, optional): the device word of the operand-range guard; bit 2 is ORed in when a row's |mean| * rstd exceeds 1 -
 * folding rounds the RAW row to 16 bits, and a row whose mean is large against its spread loses the spread (operand noise grows by
 * sqrt(1 + (mean/std)^2)): the caller repeats the forward with the LayerNorm applied before the rounding. */
int pv_rowstat_finalize(const float* partials, float* stat, int64_t tiles, int64_t rows, int64_t D, float eps, uint32_t* range_flag, void* stream);

/* The M-tile height (256 or 128) pv_gemm_bf16 would choose for these arguments (no launch). */
int pv_gemm_tile_rows(const pv_gemm_args* args);

/* out = epilogue(A . W^T): bf16 MFMA operands, fp32 accumulation.  Replaces the addmm/mm behind
 *   nn.MultiheadAttention in-proj / out-proj   models/blocks.py:91,94
 *   MLP fc1 / fc2 (+ F.gelu)                   models/blocks.py:81-83
 *   conv_proj as a GEMM over pv_im2col_bf16    models/vit.py:212
 *   residual adds                              models/vit.py:51,55
 * Requires K % 64 == 0, N % 4 == 0, lda/ldw % 8 == 0, ldo % 4 == 0; any M, N. */
int pv_gemm_bf16(const pv_gemm_args* args /* HOST pointer */, void* stream);

/* Fused multi-head self-attention core, softmax(q k^T) v per head, no mask, q already scaled:
 *   torch F.multi_head_attention_forward as called by models/blocks.py:94 (the head-averaged weights it
 *   also returns are discarded there and are not computed here).
 * qkv: bf16 [B,S,3*H*dh] packed q|k|v (nn.MultiheadAttention in_proj layout), out: bf16 [B,S,H*dh].
 * dh in {32,48,64}: S <= 416 keeps K and V of a head in the LDS (single-pass softmax), longer sequences stream 64-key blocks
 * with an online softmax; dh in {80,96,128}: the streaming kernel for every S.
 * range_flag (ABI v7, optional): the device word of the operand-range guard; the fp16-operand build ORs 4 into it when the magnitude of
 * a query's largest score exceeds 32 - the 16-bit rounding of q and k leaves an error proportional to the score in it, and beyond that
 * the softmax no longer meets BASELINE's 1e-3 (measured sensitivity: DESIGN.md section 6).  The bf16 build ignores it. */
int pv_attention_bf16(const uint16_t* qkv, uint16_t* out, int64_t B, int64_t S, int64_t H, int64_t dh, uint32_t* range_flag, void* stream);

/* The same attention for the FIRST nq ROWS of every image only (queries) against all S keys: the last encoder block, of whose output
 * only the class-token rows are read (models/vit.py:242-246; the reference computes all S rows and drops the rest).
 * q: 16-bit [B*nq rows, ldq] (already scaled), kv: 16-bit [B*S rows, ldkv] with k in columns [0, H*dh) and v in [H*dh, 2*H*dh),
 * out: 16-bit [B*nq rows, ldo].  dh in {32,48,64,80,96,128}, any S; ld* in elements, multiples of 8. */
int pv_attention_rows_bf16(const uint16_t* q, int64_t ldq, const uint16_t* kv, int64_t ldkv, uint16_t* out, int64_t ldo, int64_t B,
                           int64_t S, int64_t nq, int64_t H, int64_t dh, uint32_t* range_flag /* as pv_attention_bf16 */, void* stream);

/* Backward of pv_attention_rows_bf16 for nq = 1 (one class token; other nq: PV_ERR_UNSUPPORTED).  out = the forward's result, dout = its
 * gradient; dq [B, lddq] is the gradient of the UNSCALED q (times qscale, like pv_attention_bwd_bf16), dkv [B*S, lddkv] = dk | dv of every
 * token.  Part of loss.backward() (train/train.py:118) for the last encoder block. */
int pv_attention_rows_bwd_bf16(const uint16_t* q, int64_t ldq, const uint16_t* kv, int64_t ldkv, const uint16_t* out, int64_t ldo,
                               const uint16_t* dout, int64_t lddo, uint16_t* dq, int64_t lddq, uint16_t* dkv, int64_t lddkv, int64_t B,
                               int64_t S, int64_t nq, int64_t H, int64_t dh, float qscale, void* stream);

/* ---- precision mode "bf16x3" (opt-in): split operands v = hi + lo, concatenated along K so that the
 * SAME bf16 MFMA GEMM computes a_hi.w_hi + a_lo.w_hi + a_hi.w_lo; meets the 1e-3 logits tolerance at ~3x the GEMM work. ---- */

/* fp32 [rows,K] -> bf16 [rows,3K]: order 0 = [hi | lo | hi] (activations), order 1 = [hi | hi | lo] (weights). K % 4 == 0. */
int pv_split3_f32_bf16(const float* src, uint16_t* dst, int64_t rows, int64_t K, int order, void* stream);
/* pv_im2col_bf16 / pv_layernorm_bf16 writing [hi | lo | hi] rows of 3K / 3D. */
int pv_im2col_split_bf16(const float* x, uint16_t* cols, int64_t B, int64_t C, int64_t H, int64_t W, int64_t P, void* stream);
int pv_layernorm_split_bf16(const float* x, int64_t ldx, const float* gamma, const float* beta, const float* row_scale,
                            uint16_t* out, int64_t rows, int64_t D, float eps, void* stream);
/* Attention core in exact fp32 (f32-input MFMA): qkv fp32 [B,S,3*H*dh] (q pre-scaled) -> out bf16 [B*S, 3*H*dh] in
 * [hi | lo | hi] planes.  dh in {32,48,64}, S <= 208. */
int pv_attention_f32_split(const float* qkv, uint16_t* out, int64_t B, int64_t S, int64_t H, int64_t dh, void* stream);
/* LOCAL fallback of the attention-score guard (ABI v8): models/blocks.py:93-95 for ONE layer whose scores left PV_SCORE_LIMIT.
 * qkv fp32 [B,S,3D] (q pre-scaled: that layer's in-projection computed in split precision - pv_layernorm_split_bf16 +
 * pv_gemm_bf16 on the [hi|lo|hi] . [hi|hi|lo] operands with PV_EPI_BIAS_F32); out: 16-bit [B*S, D] = the ordinary operand of the
 * out-projection GEMM of the library's operand type.  Scores from split operands (q = q_hi + q_lo, k likewise: three MFMA products),
 * probabilities and P.V as in pv_attention_bf16.  dh in {32, 48, 64}; S <= 208 (416 at dh = 32). */
int pv_attention_split_bf16(const float* qkv, uint16_t* out, int64_t B, int64_t S, int64_t H, int64_t dh, void* stream);

/* Weight-gradient GEMM without transposed copies ("TN"): out[t][m][n] = sum over K slice t of A[k][m] * W[k][n], with
 * A = dY bf16 [K, M] (row stride lda), W = X bf16 [K, N] (row stride ldw), out fp32 [ksplit][M][ldo] partial slices for
 * pv_sum_slices_f32.  Same struct as pv_gemm_bf16 (epilogue must be PV_EPI_BIAS_F32, bias NULL).  M, N multiples of 128;
 * K a multiple of 128 (>= 128 * ksplit; slice t covers rows [t*ks, ...) with ks = floor(K/128/ksplit)*128, the last slice the rest).  dW = dY^T . X of train/train.py:118 loss.backward(). */
int pv_gemm_tn_bf16(const pv_gemm_args* args, void* stream);

/* ---- backward building blocks (SURVEY.md section 2b "B*": what train/train.py:118 `loss.backward()` needs) ---- */

/* out[n] (+)= sum_t partials[t*n_elems + n]: reduces split-K slices (accumulate != 0 adds to the existing out). fp32. */
int pv_sum_slices_f32(const float* partials, float* out, int64_t n_elems, int64_t slices, int accumulate, void* stream);
/* out = base + sum over the slices (base may alias out): finishes a split-K pv_gemm_bf16 (PV_EPI_BIAS_F32, ksplit > 1, bias in slice 0) as
 * bias + residual add - the small-batch form of PV_EPI_BIAS_RES_F32 (models/vit.py:51,55), where the token rows alone would occupy a
 * dozen of the 256 CUs through a long K loop. */
int pv_sum_slices_add_f32(const float* partials, const float* base, float* out, int64_t n_elems, int64_t slices, void* stream);
/* Finish of a split-K GEMM with a 16-bit output [M, N] (contiguous): out = gelu(sum of slices) when `gelu` (fc1 + F.gelu, models/blocks.py:81-82),
 * else the sum with columns < qcols times qscale (the in-projection's q pre-scale); range_flag as in pv_gemm_args. */
int pv_sum_slices_act_bf16(const float* partials, uint16_t* out, int64_t M, int64_t N, int64_t slices, int gelu, int64_t qcols, float qscale,
                           uint32_t* range_flag, void* stream);
/* The same finish over rows of length D that also writes ln_out = 16-bit LayerNorm(out row; gamma, beta, eps) - the LayerNorm the consumer of
 * the finished rows applies first (models/vit.py:48,53), bit-identical to pv_layernorm_bf16 on `out`. */
int pv_sum_slices_add_ln_f32(const float* partials, const float* base, float* out, int64_t rows, int64_t D, int64_t slices, const float* gamma,
                             const float* beta, float eps, uint16_t* ln_out, void* stream);
/* bf16 [R,C] (row stride lds >= C) -> bf16 [C,ldd] (ldd >= R; columns R..ldd-1 zero-filled so that the GEMM K = ldd can be
 * a multiple of 64): the K-contiguous operands of dW = (dY^T) . (X^T)^T. */
/* colsum_out (fp32 [C], optional): also the column sums of src (the bias gradient when src = dY), from the same pass;
 * colsum_ws: fp32 scratch [ceil(ldd/q)*C] with q = 64 for ldd <= 65536, else 1024; required with colsum_out. */
int pv_transpose_bf16(const uint16_t* src, int64_t lds, uint16_t* dst, int64_t R, int64_t C, int64_t ldd, float* colsum_out,
                      float* colsum_ws, void* stream);
/* LayerNorm backward (models/blocks.py:60,77): x fp32 [rows,D] (the saved LN input), dy bf16 [rows,D], gamma fp32 [D];
 * dx_out = (dres_in or 0) + dL/dx, fp32 [rows,D] (may alias dres_in); dx_bf16 (optional): the same values as bf16 (the
 * operand of the next data/weight-gradient GEMMs); dgb fp32 [3,D] (+)= (dgamma, dbeta, column sums of dx - of its bf16
 * values when dx_bf16 is given: the bias gradient of the linear layer that produced the LN input's branch).
 * ws: fp32 scratch of >= min(ceil(rows/4),1024)*3*D floats.  D % 4 == 0, D <= 1024. */
int pv_layernorm_bwd(const float* x, const uint16_t* dy, const float* gamma, const float* dres_in, float* dx_out, uint16_t* dx_bf16,
                     float* dgb, float* ws, int64_t ws_floats, int64_t rows, int64_t D, float eps, int accumulate, void* stream);
/* The same with the residual gradient handed over in 16 BITS between the two LayerNorms of a block (ABI v8, an option of the training path): dres16
 * (16-bit [rows,D], optional) instead of the fp32 dres_in (at most one of them), and dx_out may be NULL when only dx_bf16 is wanted. */
int pv_layernorm_bwd16(const float* x, const uint16_t* dy, const float* gamma, const float* dres_in, const uint16_t* dres16, float* dx_out,
                       uint16_t* dx_bf16, float* dgb, float* ws, int64_t ws_floats, int64_t rows, int64_t D, float eps, int accumulate, void* stream);
/* Masked form for ResidualViT training (models/residualvit.py:249-260): the forward was y = row_scale[row] * LayerNorm(x).
 * dmask fp32 [rows] (+)= sum_d dy * LayerNorm(x) (+ sum_d dx_out * u when u, the bf16 branch output of x1 = x + m*u, is given);
 * dy is then scaled by row_scale and the plain backward follows.  scale_copy != 0 writes dx_bf16 = row_scale * dx_out (the gradient
 * of u) - dgb[2] holds the column sums of that copy. */
int pv_layernorm_bwd_masked(const float* x, const uint16_t* dy, const float* gamma, const float* beta, const float* row_scale,
                            const float* dres_in, const uint16_t* u, float* dx_out, uint16_t* dx_bf16, int scale_copy, float* dgb,
                            float* dmask, int dmask_accumulate, float* ws, int64_t ws_floats, int64_t rows, int64_t D, float eps,
                            void* stream);
/* out = x + row_scale[row] * u: the masked residual add (residualvit.py:254) of the training forward, u = bf16 branch output. */
int pv_masked_residual(const float* x, const uint16_t* u, const float* row_scale, float* out, int64_t rows, int64_t D, void* stream);
/* Training-path GELU (models/blocks.py:82) on bf16 streams: out = gelu(pre);  dpre = dg * gelu'(pre) (may alias dg). n % 8 == 0. */
int pv_gelu_bf16(const uint16_t* pre, uint16_t* out, int64_t n, void* stream);
int pv_gelu_bwd_bf16(const uint16_t* pre, const uint16_t* dg, uint16_t* dpre, int64_t n, void* stream);
/* Column sums of a bf16 (C % 8 == 0) or fp32 (C % 4 == 0) [R,C] matrix into fp32 [C] (bias gradients: db = sum_m dY[m,:]);
 * ws: fp32 scratch [ceil(R/q)*C] with q = 64 for R <= 65536, else 1024 (short matrices are cut finer so that they fill the chip). */
int pv_colsum_f32(const void* src, int src_is_bf16, float* out, float* ws, int64_t R, int64_t C, int accumulate, void* stream);

/* Backward of pv_gather_tokens (models/rankvit.py:55-77 under loss.backward()): dy fp32 [B,1+k,D], keep int32 [B,k] ->
 * dx fp32 [B,S_in,D] with dx[b,0] = dy[b,0], dx[b,1+keep[b,i]] = dy[b,1+i], zeros elsewhere (every row written). */
int pv_scatter_tokens(const float* dy, const int32_t* keep, float* dx, int64_t B, int64_t S_in, int64_t k, int64_t D, void* stream);

/* Attention backward for one block (models/blocks.py:32-37 under train/train.py:118 loss.backward()):
 * qkv bf16 [B,S,3D] as the forward in-proj wrote it (q columns pre-scaled by qscale), dout bf16 [B,S,D] = dL/d(attention
 * output); dqkv bf16 [B,S,3D] = dL/d(in-proj output before the q pre-scale).  Probabilities are recomputed.
 * dbias_partial (fp32 [B, 3D], optional): per-image column sums of dqkv (of the stored 16-bit values); their sum over B is the
 * in-proj bias gradient.  dh in {32, 48, 64}; S <= 208 (S <= 416 at dh = 32): Q, K, V and dO of one head live in the LDS. */
int pv_attention_bwd_bf16(const uint16_t* qkv, const uint16_t* dout, uint16_t* dqkv, float* dbias_partial, int64_t B, int64_t S,
                          int64_t H, int64_t dh, float qscale, void* stream);
/* The same backward as ONE persistent workgroup per CU that walks over its (image, head) items, from the forward's row statistics (ABI v9; the
 * training path's choice where it applies):
 *   pv_attention_lse_bf16     = pv_attention_bf16 that also writes lse fp32 [B,H,S] = log2(sum_k exp(s[q,k])) (base-2 log-sum-exp of the scaled scores);
 *                               S <= 416 (the resident-K/V kernel), PV_ERR_UNSUPPORTED beyond.
 *   pv_attention_bwd_lse_bf16 : `out` = the forward's 16-bit output [B,S,D], `lse` as above.  p = exp2(s log2(e) - lse) and D = rowsum(dO o out) replace the
 *                               recomputed row maximum / sum / sum(P o dP); K | V of the next item land while pass 2 of this one runs, every operand is fetched
 *                               once.  dbias_partial as above, in closed form where there is one: the key third is identically 0 (softmax is invariant under a
 *                               shift of all keys; autograd returns rounding noise there), the value third is the per-image column sums of dout (rows of this
 *                               head), the query third the column sums of the stored dQ.  dh in {48, 64}, 129 <= S <= 208; PV_ERR_UNSUPPORTED otherwise. */
int pv_attention_lse_bf16(const uint16_t* qkv, uint16_t* out, float* lse, int64_t B, int64_t S, int64_t H, int64_t dh, uint32_t* range_flag,
                          void* stream);
int pv_attention_bwd_lse_bf16(const uint16_t* qkv, const uint16_t* dout, const uint16_t* out, const float* lse, uint16_t* dqkv,
                              float* dbias_partial, int64_t B, int64_t S, int64_t H, int64_t dh, float qscale, void* stream);

/* Final LayerNorm on the class-token rows only + sum over class tokens:
 *   models/vit.py:95 `self.ln(input)` restricted to the rows models/vit.py:242-243 consume.
 * x: fp32 [B,S,D]; pooled: fp32 [B,D] = sum_{c<num_cls} LN(x[b,c]). */
int pv_cls_pool(const float* x, const float* gamma, const float* beta, float* pooled, int64_t B, int64_t S,
                int64_t D, int64_t num_cls, float eps, void* stream);

/* Classification head in fp32: models/vit.py:246 `self.head(x)`.  logits[B,C] = pooled[B,D] . w[C,D]^T + b. */
int pv_head_f32(const float* pooled, const float* w, const float* b, float* logits, int64_t B, int64_t D,
                int64_t C, void* stream);

/* Token magnitudes: models/rankvit.py:63 `torch.norm(input, dim=-1)` on rows [1,S) of x fp32 [B,S,D]
 * -> norms fp32 [B,S-1]. */
int pv_token_norm(const float* x, float* norms, int64_t B, int64_t S, int64_t D, void* stream);

/* Descending rank + top-k: models/rankvit.py:67,74-75 argsort(descending)[:, :k].
 * norms fp32 [B,N] -> keep int32 [B,k] in sorted (descending) order; ties: lowest index first
 * (the reference sort is unstable).  N <= 4096. */
int pv_rank_topk(const float* norms, int32_t* keep, int64_t B, int64_t N, int64_t k, void* stream);

/* The same ranking from the per-column-tile sums of squares a producer GEMM left behind (pv_gemm_args.rowsq_out):
 * rowsq fp32 [tiles, B*S] over the rows of x[B,S,D] (row 0 of every image = class token, not ranked); norm of token i of image b
 * = sqrt(sum_t rowsq[t][b*S + 1 + i]).  keep int32 [B,k] as pv_rank_topk (N = S-1 <= 4096). */
int pv_rank_topk_partials(const float* rowsq, int64_t tiles, int32_t* keep, int64_t B, int64_t S, int64_t k, void* stream);
/* The same two rankings that also report each image's keep BOUNDARY (ABI v10): gap_min fp32 [B] (device, optional) is lowered to
 * min(gap_min[b], (n_{k-1} - n_k) / n_{k-1}), the relative gap between the image's last kept and first dropped norm (untouched when k == N).  The caller
 * fills it with +inf before the forward and passes the same array to every ranked layer: what remains is the narrowest boundary an image crossed.
 * models/rankvit.py:63-77 ranks fp32 norms; here they carry the 16-bit layers' rounding (~1e-4 relative), so an image whose gap is of that size may keep
 * another token than the reference - the host side repairs exactly those images in split precision when asked to (PEEKVIT_AMD_RANK_REPAIR). */
int pv_rank_topk_gap(const float* norms, int32_t* keep, float* gap_min, int64_t B, int64_t N, int64_t k, void* stream);
int pv_rank_topk_partials_gap(const float* rowsq, int64_t tiles, int32_t* keep, float* gap_min, int64_t B, int64_t S, int64_t k, void* stream);

/* Compaction gather: models/rankvit.py:71,75-77 gather + slice + cat class token.
 * x fp32 [B,S_in,D], keep int32 [B,k] (indices into rows 1..S_in-1, i.e. 0-based among non-CLS tokens)
 * -> out fp32 [B,1+k,D] with out[b,0]=x[b,0], out[b,1+j]=x[b,1+keep[b,j]].  D % 4 == 0. */
int pv_gather_tokens(const float* x, const int32_t* keep, float* out, int64_t B, int64_t S_in, int64_t k,
                     int64_t D, void* stream);

/* Residual gate + token masking: models/residualvit.py:197-235 (eval, sigmoid gate, learnable budget
 * token) with models/residualvit.py:47-74 and models/blocks.py:62-69:
 *   thr[b]    = sigmoid(x[b,S-1] . wb + bb)                               (budget_token_gate, :212)
 *   mask[b,i] = relu(sigmoid((x[b,1+i] . wg + bg)/temp + sbias) - thr[b])   i < S-2            (:217)
 *   x_out[b]  = [x[b,0] | mask[b,i] * x[b,1+i] | x[b,S-1]]                  (:220-227, masked_input)
 * x_in: fp32 [B,S,D] (layout [cls | N | budget], one special token); x_out: same shape, may alias x_in, or NULL (the masked tokens are
 * not written: the caller uses pv_gemm_args.res_scaled and ln_out instead);
 * mask_out: fp32 [B,S-2] (block.mask); row_scale: fp32 [B,S] = [1, mask, 1] (:230-235 fwd_mask) for the
 * LN / out-proj epilogues.
 * thr_out: fp32 [B] or NULL - the per-image threshold (what the reference leaves in residual_gate.threshold, :66).
 * ln_out: 16-bit [B,S,D] or NULL - also row_scale * LayerNorm(x_out; ln_gamma, ln_beta, ln_eps), the first thing the masked block does with
 * these rows (:251), from the registers the row is in (bit-identical to pv_layernorm_bf16 with that row_scale). */
int pv_residual_gate(const float* x_in, float* x_out, const float* wg, const float* bg, const float* wb,
                     const float* bb, float temp, float sigmoid_bias, float* mask_out, float* row_scale, float* thr_out,
                     const float* ln_gamma, const float* ln_beta, float ln_eps, uint16_t* ln_out,
                     int64_t B, int64_t S, int64_t D, void* stream);
/* Its backward (training: loss.backward() through the gate).  dx_out = dL/d(x_out) [B,S,D], drow = dL/d(row_scale) [B,S] (the mask
 * gradient of the masked block plus that of any auxiliary loss on block.mask); dx_in [B,S,D]; parameter gradients as per-image partials
 * dwg_part, dwb_part fp32 [B,D] and scal_part fp32 [B,4] = (dbg, dbb, 0, 0) - sum them over B (pv_colsum_f32). */
int pv_residual_gate_bwd(const float* x_in, const float* dx_out, const float* drow, const float* wg, const float* bg, const float* wb,
                         const float* bb, float temp, float sigmoid_bias, float* dx_in, float* dwg_part, float* dwb_part,
                         float* scal_part, int64_t B, int64_t S, int64_t D, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PEEKVIT_HIP_H */
