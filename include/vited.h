/*
 * vited.h - C ABI of libvited_hip.so: the MI355X (gfx950) kernels behind the ViT encoder-decoder
 * hot path of glmanhtu/vit-ed.
 *
 * The reference has no FFI of its own (it is pure Python on PyTorch/timm); its boundary for this
 * path is the module contract of models/vision_transformer.py:13-420.  Each entry point below
 * replaces the ATen/cuDNN/SDPA call that the cited reference line dispatches implicitly, so a
 * maintainer binds them with ctypes from the Attention/Block/CrossBlock
 * forward methods.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes only; no C++ types, no exceptions, no torch types.
 *  - every pointer is a DEVICE pointer owned by the caller (inputs, outputs, saved tensors and
 *    workspace); kernels never allocate, free, or retain pointers past the call.
 *  - `stream` is a hipStream_t passed as void*; every launch goes to it; no entry synchronises,
 *    so all of them are hipGraph-capturable.
 *  - return value: 0 (VITED_OK) or a VITED_ERR_* code; vited_strerror() names it.
 *  - dtype arguments select the activation type (VITED_F32 or VITED_BF16); parameters, the
 *    residual stream, statistics, gradients of parameters and all accumulation are fp32.
 *  - strides (ld*, *_bs, *_ts) are in ELEMENTS of the tensor they describe.
 */
#ifndef VITED_H
#define VITED_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VITED_ABI_VERSION 1

#define VITED_OK 0
#define VITED_ERR_BAD_ARG 1     /* null pointer, non-positive size, misaligned pointer/stride */
#define VITED_ERR_UNSUPPORTED 2 /* shape/dtype combination no kernel covers */
#define VITED_ERR_LAUNCH 3      /* hipGetLastError() after the launch */
#define VITED_ERR_WORKSPACE 4   /* workspace smaller than vited_*_workspace_bytes() */

#define VITED_F32 0
#define VITED_BF16 1

/* GEMM epilogues (vited_gemm) */
#define VITED_EPI_STORE 0          /* out = T(acc + bias)                                          */
#define VITED_EPI_GELU 1           /* out = T(z), out2 = T(gelu_erf(z)), z = acc + bias            */
#define VITED_EPI_RESIDUAL 2       /* out_f32[orow] = residual[rrow] + acc + bias (row remap below) */
#define VITED_EPI_MUL_GELU_GRAD 3  /* out = T(acc * gelu_erf'(aux)), aux = saved pre-activation    */
#define VITED_EPI_STORE_F32 4      /* out_f32 = acc + bias (logits of the head stay fp32)           */
#define VITED_EPI_MUL 5            /* out = T((acc + bias) * aux), aux = a saved factor (gelu'(z))   */
#define VITED_EPI_GELU_GRAD 6      /* out = T(gelu_erf'(z)), out2 = T(gelu_erf(z)), z = acc + bias:
                                      what fc1 saves for backward (timm Mlp: the backward of GELU then is
                                      one multiply, VITED_EPI_MUL, instead of an erf/exp per element)  */

/* B-operand layouts (vited_gemm) */
#define VITED_B_NK 0 /* B is [N, K] row-major: out = A . B^T  (nn.Linear weight)                   */
#define VITED_B_KN 1 /* B is [K, N] row-major: out = A . B                                         */

int vited_abi_version(void);
const char* vited_strerror(int code);

/* Which implementation the last vited_gemm / vited_attention_* call on this thread dispatched to:
 * 0 = none yet, 1 = portable fp32-FMA kernel, 2 = bf16 MFMA kernel, 3 = persistent bf16 MFMA GEMM
 * (opt-in, VITED_NT=as).  Test/diagnostic use. */
int vited_last_gemm_path(void);
int vited_last_attention_path(void);

/* ---- data movement -------------------------------------------------------------------------- */

/* dst[i] = (dst_dtype) src[i].  Used for the bf16 shadow of the fp32 master parameters. */
int vited_cast(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t n, void* stream);

/* dst[c, r] = (dst_dtype) src[r, c]; src is fp32 [rows, cols].  Transposed bf16 weight shadow used
 * by the input-gradient GEMMs. */
int vited_cast_transpose(const float* src, void* dst, int dst_dtype, int64_t rows, int64_t cols, void* stream);

/* Refresh every bf16 weight shadow of a model in ONE launch (after an optimizer step; replaces one
 * vited_cast + one vited_cast_transpose per weight).  desc is a DEVICE array of count x 6 int64:
 *   {src fp32 [rows, cols] address, dst bf16 [rows, cols] address or 0, dst_t bf16 [cols, rows] address or 0,
 *    rows, cols, first_tile}
 * where first_tile is the running sum of ceil(rows/64) * ceil(cols/64) over the preceding entries and
 * total_tiles the sum over all of them.  The reference has no counterpart: under autocast PyTorch
 * re-casts each weight inside every F.linear call (vision_transformer.py:49-80 via misc/engine.py:208). */
int vited_cast_weights(const int64_t* desc, int count, int64_t total_tiles, void* stream);

/* Patch extraction for timm PatchEmbed's Conv2d(k = s = p) (used at vision_transformer.py:383,391):
 * out[(b * G*G + py*G + px), (c*p + i)*p + j] = img[idx(b), c, py*p + i, px*p + j]
 * img is fp32 with batch stride img_bs (so x[:, 0] / x[:, 1] of the stacked pair tensor need no
 * copy, vision_transformer.py:408); batch_index (nullable int64[B]) gathers images by index
 * (hisfrag.py:153,226-227) without materialising the gathered copy. */
int vited_patchify(const float* img, int64_t img_bs, const int64_t* batch_index, void* out, int out_dtype,
                   int64_t batch, int chans, int img_size, int patch, void* stream);

/* The same from uint8 pixels, with the input pipeline's ToTensor + Normalize(mean, std) (data/transforms.py:14-18, applied per
 * sample on the host by the reference's loader, data/datasets/div2k_patch.py:108-162) folded in:
 * value = (pixel / 255 - mean[c]) / std[c].  mean / std are HOST arrays of `chans` (<= 4) floats.  A batch then crosses PCIe and
 * HBM at 1 byte per pixel (misc/engine.py:203-204 copies fp32). */
int vited_patchify_u8(const uint8_t* img, int64_t img_bs, const int64_t* batch_index, void* out, int out_dtype,
                      int64_t batch, int chans, int img_size, int patch, const float* mean, const float* std, void* stream);

/* Patch-pair assembly on the device (data/datasets/div2k_patch.py:108-121,155-162; transforms.py:14-18): per sample a uint8
 * region [chans, 2 S, 3 S] (the 3-column x 2-row grid of S x S cells the reference cuts with transforms.crop(patch, 3, 2)),
 * the two cells of its pair (cells int32 [batch, 2], 0..5 row-major) and the erosion size e = ceil(S (1 - erosion_ratio))
 * (erode int32 [batch], e <= S).  out uint8 [batch, 2, chans, S, S] = Resize(S)(CenterCrop(e)(cell)) with Pillow's 8-bit
 * bilinear resample, bit for bit; ToTensor + Normalize then happen inside vited_patchify_u8.  cells / erode are DEVICE arrays. */
int vited_crop_pairs_u8(const uint8_t* src, int64_t src_bs, const int* cells, const int* erode, uint8_t* out, int64_t batch,
                        int chans, int img_size, void* stream);

/* out[b, r] = (out_dtype) in[b, row_offset + r] for r < rows: drops the cls row of a token-gradient
 * tensor before the patch-embed weight gradient. in is fp32 [batch, in_rows, dim]. */
int vited_slice_rows_cast(const float* in, void* out, int out_dtype, int64_t batch, int64_t in_rows,
                          int64_t row_offset, int64_t rows, int64_t dim, void* stream);

/* x[b, 0, :] = cls[:] + pos[0, :] : the cls row of timm _pos_embed (vision_transformer.py:392). */
int vited_write_cls_row(float* x, const float* cls, const float* pos, int64_t batch, int64_t rows_per_batch,
                        int64_t dim, void* stream);

/* out[r] = sum_b in[b, r] (fp32 out; in is `in_dtype` [batch, width]).  Deterministic two-pass;
 * workspace >= vited_sum_rows_workspace_bytes().  Serves bias gradients (column sums) and the
 * pos_embed / cls_token gradients (batch sums). */
int64_t vited_sum_rows_workspace_bytes(int64_t batch, int64_t width);
int vited_sum_rows(const void* in, int in_dtype, int64_t in_ld, float* out, int64_t batch, int64_t width,
                   float* workspace, int64_t workspace_bytes, void* stream);

/* ---- LayerNorm (nn.LayerNorm(eps=1e-6), vision_transformer.py:101,114,231,244,245,258,348,400) */

/* y[r, :] = (x[r, :] - mean) * rstd * gamma + beta ; saves mean/rstd (fp32 [rows]). */
int vited_layernorm_fwd(const float* x, int64_t x_ld, const float* gamma, const float* beta, void* y,
                        int y_dtype, int64_t y_ld, float* mean, float* rstd, int64_t rows, int64_t dim,
                        float eps, void* stream);

/* dx_out = (dx_in ? dx_in : 0) + LN'(dy); optional low-precision copy of dx_out (dx_lp, may be
 * null); dgamma/dbeta receive the column sums: overwritten, or added onto their current content when
 * `accumulate` != 0 (accumulation straight into a parameter's .grad).  workspace >= *_workspace_bytes. */
int64_t vited_layernorm_bwd_workspace_bytes(int64_t rows, int64_t dim);
int vited_layernorm_bwd(const void* dy, int dy_dtype, int64_t dy_ld, const float* x, int64_t x_ld,
                        const float* gamma, const float* mean, const float* rstd, const float* dx_in,
                        int64_t dx_in_ld, float* dx_out, int64_t dx_out_ld, void* dx_lp, int dx_lp_dtype,
                        int64_t dx_lp_ld, float* dgamma, float* dbeta, int accumulate, int64_t rows, int64_t dim,
                        float* workspace, int64_t workspace_bytes, void* stream);

/* ---- Linear / GEMM (nn.Linear: qkv :34, proj :38, q :151, kv :152, timm Mlp fc1/fc2, head) ---- */

/* acc[m, n] = sum_k A[m, k] * B(n, k), fp32 accumulation; A is `dtype` [M, K] (row stride lda);
 * B is `dtype`, laid out per b_layout (row stride ldb); then the epilogue (VITED_EPI_*):
 *   bias      fp32 [N] or null
 *   aux       `dtype` [M, N] (row stride ldo)        - MUL_GELU_GRAD and MUL
 *   residual  fp32                                     - RESIDUAL only
 *   out/out2  `dtype` [M, N] (row stride ldo); for RESIDUAL and STORE_F32 `out` is fp32
 * RESIDUAL row remap (patch-embed writes tokens behind a cls row and adds a broadcast pos_embed):
 *   orow = (m / rows_per_batch) * out_rows_per_batch + m % rows_per_batch + row_offset
 *   rrow = residual_bcast ? (m % rows_per_batch + row_offset) : orow
 * with rows_per_batch == 0 meaning the identity map (orow = rrow = m). */
int vited_gemm(const void* A, int64_t lda, const void* B, int64_t ldb, int b_layout, int dtype, int64_t M,
               int64_t N, int64_t K, int epilogue, const float* bias, const void* aux,
               const float* residual, void* out, void* out2, int64_t ldo, int64_t rows_per_batch,
               int64_t out_rows_per_batch, int64_t row_offset, int residual_bcast, void* stream);

/* dW[n, k] = sum_m dY[m, n] * X[m, k] (fp32 [N, K]) and, if dbias != null, dbias[n] = sum_m dY[m, n];
 * overwritten, or added onto their current content when `accumulate` != 0.  dY / X are `dtype`.
 * workspace >= *_workspace_bytes. */
int64_t vited_linear_bwd_weight_workspace_bytes(int64_t M, int64_t N, int64_t K);
int vited_linear_bwd_weight(const void* dY, int64_t lddy, const void* X, int64_t ldx, int dtype, int64_t M,
                            int64_t N, int64_t K, float* dW, float* dbias, int accumulate, float* workspace,
                            int64_t workspace_bytes, void* stream);

/* Several weight gradients in ONE launch: the dW / dbias of every Linear of one transformer block (Block / CrossBlock backward,
 * vision_transformer.py:124-127, 268-272), queued by the caller and flushed together.  Arrays have `count` (<= 40) entries, all
 * host memory; entry i is the product of vited_linear_bwd_weight with the same meanings (dbias[i] may be null).  Sharing the
 * chip's workgroup slots between the products cuts the number of row splits - and the fp32 partial slabs - several-fold.
 * bf16 only, every K a multiple of 384 and every M >= 4096 (vited_linear_bwd_weight_batched_supported); otherwise the caller
 * issues vited_linear_bwd_weight per product. */
int vited_linear_bwd_weight_batched_supported(int count, const int64_t* M, const int64_t* N, const int64_t* K, int dtype);
int64_t vited_linear_bwd_weight_batched_workspace_bytes(int count, const int64_t* M, const int64_t* N, const int64_t* K);
int vited_linear_bwd_weight_batched(int count, const void* const* dY, const int64_t* lddy, const void* const* X,
                                    const int64_t* ldx, const int64_t* M, const int64_t* N, const int64_t* K, float* const* dW,
                                    float* const* dbias, int dtype, int accumulate, float* workspace, int64_t workspace_bytes,
                                    void* stream);

/* ---- Linear fused with the LayerNorm on the other side of it (row-complete 384-wide tile, bf16 MFMA; gemm_row.hip) ----
 * The reference's blocks are chains  x = x + f(norm(x))  (Block.forward vision_transformer.py:124-127, CrossBlock.forward
 * :268-272, norm_layer :348): every residual Linear (attn.proj :38, cross_attn.proj :156, timm Mlp fc2) is followed by the next
 * sub-block's LayerNorm, and every Linear that consumes a LayerNorm's output (qkv :34, q :151, kv :152, fc1) is followed, in
 * backward, by that LayerNorm's backward.  These two entries do each pair in ONE kernel so the LayerNorm is not a separate
 * pass over the fp32 residual stream.  bf16 operands, N == 384 (the embed width of every shipped pjs config), K % 64 == 0;
 * vited_linear_layernorm_supported() tells whether a shape is covered - otherwise the caller runs vited_gemm +
 * vited_layernorm_fwd / vited_layernorm_bwd. */
int vited_linear_layernorm_supported(int64_t M, int64_t N, int64_t K);

/* y = residual + a . w^T + bias   (fp32 [M, N], row stride ldy; residual fp32 row stride ldr, may alias y row for row)
 * h = LayerNorm(y; gamma, beta, eps) bf16 [M, N] (row stride ldh), mean / rstd fp32 [M]   - or h == null: no LayerNorm
 *   a bf16 [M, K] (lda), w bf16 [N, K] (ldw), bias fp32 [N] or null */
int vited_linear_residual_layernorm_fwd(const void* a, int64_t lda, const void* w, int64_t ldw, const float* bias,
                                        const float* residual, int64_t ldr, float* y, int64_t ldy, const float* gamma,
                                        const float* beta, float eps, void* h, int64_t ldh, float* mean, float* rstd,
                                        int64_t M, int64_t N, int64_t K, void* stream);

/* dh = dy . wt^T  (wt = the transposed weight shadow, bf16 [N, K]: dX of y = LN(x) W^T), never written anywhere;
 * dx_out = (dx_in ? dx_in : 0) + LN'(dh; x, mean, rstd, gamma)  fp32 (dx_out may alias dx_in), optional bf16 copy dx_lp;
 * dgamma / dbeta: column sums of dh * xhat / dh, overwritten or (accumulate != 0) added.  workspace >= *_workspace_bytes.
 * With dgamma == dbeta == null the column partials stay in `workspace` as [vited_linear_layernorm_bwd_partial_rows(M)][2][N]
 * fp32 for vited_layernorm_bwd_finish_batched, which sums the partials of several LayerNorms in one launch. */
/* The same with the contraction dim cut into `segments` column blocks of seg_k, block j read from the tensor at
 * dy + j * seg_stride (elements): one [M, seg_k] tensor per decoder block (each contiguous for its own attention backward), all
 * contracted against wt [N, segments * seg_k] in one kernel. */
int vited_linear_layernorm_bwd_segmented(const void* dy, int64_t lddy, int64_t seg_k, int64_t seg_stride, int64_t segments,
                                         const void* wt, int64_t ldwt, const float* x, int64_t ldx, const float* gamma,
                                         const float* mean, const float* rstd, const float* dx_in, int64_t dx_in_ld,
                                         float* dx_out, int64_t dx_out_ld, void* dx_lp, int64_t dx_lp_ld, float* dgamma,
                                         float* dbeta, int accumulate, int64_t M, int64_t N, float* workspace,
                                         int64_t workspace_bytes, void* stream);
int64_t vited_linear_layernorm_bwd_partial_rows(int64_t M);
int vited_layernorm_bwd_finish_batched(int count, const float* const* partial, const int* nparts, float* const* dgamma,
                                       float* const* dbeta, const int* accumulate, int64_t dim, void* stream);
int64_t vited_linear_layernorm_bwd_workspace_bytes(int64_t M, int64_t N);
int vited_linear_layernorm_bwd(const void* dy, int64_t lddy, const void* wt, int64_t ldwt, const float* x, int64_t ldx,
                               const float* gamma, const float* mean, const float* rstd, const float* dx_in,
                               int64_t dx_in_ld, float* dx_out, int64_t dx_out_ld, void* dx_lp, int64_t dx_lp_ld,
                               float* dgamma, float* dbeta, int accumulate, int64_t M, int64_t N, int64_t K,
                               float* workspace, int64_t workspace_bytes, void* stream);

/* ---- norm_context + kv projection of all decoder blocks as one GEMM (context_fold.hip) ----
 * Every CrossBlock normalises the SAME encoder features with its own norm_context (vision_transformer.py:245,269-270) before
 * its kv projection (:152,177-179).  With xhat = LayerNorm(features; 1, 0):  kv_l = xhat (W_l o gamma_l)^T + (W_l beta_l + b_l).
 * vited_fold_context_weights writes the folded bf16 weights of `count` (<= 16) blocks stacked [count * N, K], their transpose
 * [K, count * N] and the folded fp32 bias [count * N]; vited_unfold_context_grads turns the gradient of the folded weights /
 * bias (dwf [count * N, K], dbf [count * N], fp32) back into dW_l, db_l (may be null), dgamma_l, dbeta_l - overwritten, or added
 * when accumulate != 0.  Pointer arrays are host arrays of device pointers. */
int vited_fold_context_weights(int count, const float* const* w, const float* const* bias, const float* const* gamma,
                               const float* const* beta, int64_t N, int64_t K, void* w_out, void* wt_out, float* bias_out,
                               void* stream);
int vited_unfold_context_grads(int count, const float* dwf, const float* dbf, const float* const* w, const float* const* gamma,
                               const float* const* beta, float* const* dw, float* const* dbias, float* const* dgamma, float* const* dbeta, int64_t N,
                               int64_t K, int accumulate, void* stream);

/* ---- fused MLP branch of a block: y = x + fc2(gelu(fc1(LayerNorm(x)))) (vision_transformer.py:126,271; timm Mlp :115,:259) ---- */

/* One kernel for the second half of Block.forward / CrossBlock.forward (SURVEY.md section 8(b) "optional fused mlp"): LayerNorm
 * (eps as given), fc1 + bias, exact-erf GELU, fc2 + bias and the residual add, bf16 MFMA with fp32 accumulation; the LayerNorm
 * output and the hidden activation stay on chip.  Covers dim 384 / hidden 1536 (every shipped pjs config); other shapes return
 * VITED_ERR_UNSUPPORTED and the caller runs vited_layernorm_fwd + vited_gemm(GELU_GRAD) + vited_gemm(RESIDUAL) instead.
 *   x, y        fp32 [rows, dim] (row strides ldx, ldy); y may not alias x
 *   w1, w2      bf16 [hidden, dim], [dim, hidden] dense (the nn.Linear weights in the activation dtype); b1, b2, gamma, beta fp32
 *   h, gd, u, mean, rstd   what the backward kernels read - LN(x) bf16 [rows, dim], gelu'(z) and gelu(z) bf16 [rows, hidden]
 *               (dense), row statistics fp32 [rows] - all five given, or all five null for inference (nothing is saved) */
int vited_mlp_fwd(const float* x, int64_t ldx, const float* gamma, const float* beta, const void* w1, const float* b1,
                  const void* w2, const float* b2, float* y, int64_t ldy, void* h, void* gd, void* u, float* mean,
                  float* rstd, int64_t rows, int64_t dim, int64_t hidden, float eps, void* stream);

/* One encoder Block forward (Block.forward, vision_transformer.py:124-127) behind one call: the launch sequence
 * LayerNorm -> qkv GEMM -> attention -> proj + residual -> fused MLP branch on `stream` (5 launches; 7 when the MLP shape is
 * outside vited_mlp_fwd's cover).  bf16 activations, fp32 residual stream, inference form (nothing saved for backward).
 *   x, y       fp32 [batch * tokens, dim] dense; y may not alias x
 *   wqkv [3 dim, dim], wproj [dim, dim], w1 [hidden, dim], w2 [dim, hidden]   bf16 dense; biases and LayerNorm parameters fp32
 *   workspace  >= vited_block_workspace_bytes(), 256-byte aligned */
int64_t vited_block_workspace_bytes(int64_t batch, int64_t tokens, int64_t dim, int64_t hidden, int heads);
int vited_block_fwd(const float* x, float* y, int64_t batch, int64_t tokens, int64_t dim, int heads, int64_t hidden,
                    const float* ln1_g, const float* ln1_b, const void* wqkv, const float* bqkv, const void* wproj,
                    const float* bproj, const float* ln2_g, const float* ln2_b, const void* w1, const float* b1,
                    const void* w2, const float* b2, float eps, void* workspace, int64_t workspace_bytes, void* stream);

/* One decoder CrossBlock forward (CrossBlock.forward, vision_transformer.py:268-272: self-attention branch, cross-attention of the
 * image-2 tokens over the image-1 features, MLP branch) behind one call - a launch sequence on `stream`, inference form.
 *   x          fp32 [batch * tokens, dim]       the image-2 token stream (cls + patches)
 *   context    fp32 [batch * ctx_tokens, dim]   the encoder features of image 1
 *   y          fp32 [batch * tokens, dim]; may not alias x
 *   ln1 / lnq / lnc / ln2   norm1, norm_cross, norm_context, norm2 (gamma, beta fp32)
 *   wqkv [3 dim, dim], wproj, wq, wcproj [dim, dim], wkv [2 dim, dim], w1 [hidden, dim], w2 [dim, hidden]   bf16 dense; biases fp32
 *   workspace  >= vited_cross_block_workspace_bytes(), 256-byte aligned */
int64_t vited_cross_block_workspace_bytes(int64_t batch, int64_t tokens, int64_t ctx_tokens, int64_t dim, int64_t hidden, int heads);
int vited_cross_block_fwd(const float* x, const float* context, float* y, int64_t batch, int64_t tokens, int64_t ctx_tokens,
                          int64_t dim, int heads, int64_t hidden, const float* ln1_g, const float* ln1_b, const void* wqkv,
                          const float* bqkv, const void* wproj, const float* bproj, const float* lnq_g, const float* lnq_b,
                          const float* lnc_g, const float* lnc_b, const void* wq, const float* bq, const void* wkv,
                          const float* bkv, const void* wcproj, const float* bcproj, const float* ln2_g, const float* ln2_b,
                          const void* w1, const float* b1, const void* w2, const float* b2, float eps, void* workspace,
                          int64_t workspace_bytes, void* stream);

/* ---- optimizer step on the flat gradient buffer ----------------- */

/* Gradient clip + AdamW + bf16 weight-shadow refresh + gradient zeroing in two launches; replaces
 * clip_grad_norm_ + optimizer.step() + zero_grad() of misc/utils.py:215-223 / misc/engine.py:231 and the
 * torch.optim.AdamW that misc/optimizer.py:25-27 builds (same update rule; parity in tests/test_gpu_optim.py).
 *   desc        DEVICE array of count x 10 int64, one row per parameter:
 *               {p fp32 [rows, cols], g fp32 (its slice of grad_flat), exp_avg, exp_avg_sq, shadow bf16 [rows, cols] or 0,
 *                shadow_t bf16 [cols, rows] or 0, rows, cols, first_tile, group}
 *               first_tile = running sum of ceil(rows/64) * ceil(cols/64); total_tiles the sum over all rows.
 *   grad_flat   the contiguous fp32 gradient buffer every g points into (the L2 norm is taken over all of it)
 *   hyper       DEVICE fp32 array: [0] = number of updates done so far (the call increments it: bias correction uses
 *               the incremented value), [1..7] unused, then 8 floats per parameter group
 *               {lr, beta1, beta2, eps, weight_decay, 0, 0, 0} - device-resident so that a replayed hipGraph follows
 *               lr_scheduler.step_update (misc/engine.py:228)
 *   max_norm    clip_grad_norm_ threshold (<= 0: no clipping); norm_out (nullable) receives the pre-clip norm
 *   zero_grad   != 0: every g is zeroed after it was read
 *   workspace   >= vited_adamw_workspace_bytes() */
int64_t vited_adamw_workspace_bytes(void);
int vited_adamw_step(const int64_t* desc, int count, int64_t total_tiles, const float* grad_flat, int64_t grad_numel,
                     float* hyper, float max_norm, int zero_grad, float* norm_out, float* workspace,
                     int64_t workspace_bytes, void* stream);

/* ---- attention core (F.scaled_dot_product_attention, vision_transformer.py:63-66,183-186) ----- */

/* o[b, i, h, :] = softmax_j(scale * q[b,i,h,:] . k[b,j,h,:]) v[b,j,h,:]   (no mask, no dropout)
 * q/k/v are addressed as ptr + b*bs + token*ts + h*head_dim (+d), so the packed qkv [B,N,3,h,hd]
 * (:58) and kv [B,Nc,2,h,hd] (:178) projections are consumed in place.  o is [B, Nq, H*hd] with
 * token stride o_ts; lse (fp32 [B, H, Nq]) = log sum exp of the scaled scores, saved for backward. */
int vited_attention_fwd(const void* q, int64_t q_bs, int64_t q_ts, const void* k, int64_t k_bs, int64_t k_ts,
                        const void* v, int64_t v_bs, int64_t v_ts, void* o, int64_t o_bs, int64_t o_ts,
                        float* lse, int dtype, int64_t batch, int heads, int64_t nq, int64_t nk, int head_dim,
                        float scale, void* stream);

/* The same with an indirection on the key/value side: batch item b attends over k / v of batch item kv_index[b]
 * (kv_index: DEVICE int64 [batch], values in [0, number of k/v batch items); null = identity).  Serves the pairwise
 * similarity-matrix inference (hisfrag.py:218-231): the cross-attention keys/values of an image-1 row block are projected
 * ONCE and every (i, j) pair of a pair batch reads row i's - no materialised features[i] gather (hisfrag.py:227), no
 * per-pair norm_context + kv projection (vision_transformer.py:174-179 re-runs them for every pair).  Forward only. */
int vited_attention_fwd_indexed(const void* q, int64_t q_bs, int64_t q_ts, const void* k, int64_t k_bs, int64_t k_ts,
                                const void* v, int64_t v_bs, int64_t v_ts, const int64_t* kv_index, void* o, int64_t o_bs,
                                int64_t o_ts, float* lse, int dtype, int64_t batch, int heads, int64_t nq, int64_t nk,
                                int head_dim, float scale, void* stream);

/* dq/dk/dv in the same strided layouts; delta (fp32 [B, H, Nq]) is scratch for rowsum(dO * O). */
int vited_attention_bwd(const void* q, int64_t q_bs, int64_t q_ts, const void* k, int64_t k_bs, int64_t k_ts,
                        const void* v, int64_t v_bs, int64_t v_ts, const void* o, const void* d_o,
                        int64_t o_bs, int64_t o_ts, const float* lse, float* delta, void* dq, int64_t dq_bs,
                        int64_t dq_ts, void* dk, int64_t dk_bs, int64_t dk_ts, void* dv, int64_t dv_bs,
                        int64_t dv_ts, int dtype, int64_t batch, int heads, int64_t nq, int64_t nk,
                        int head_dim, float scale, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VITED_H */
