// vited_block_fwd: one encoder Block forward (vision_transformer.py:124-127) behind ONE C-ABI call - the "optional fused block"
// of SURVEY.md section 8(b).  It is a fixed launch sequence on the caller's stream, not a single kernel:
//     LayerNorm -> qkv GEMM -> attention core (consumes the packed qkv in place) -> proj GEMM + residual epilogue ->
//     fused MLP branch (vited_mlp_fwd: LayerNorm + fc1 + GELU + fc2 + residual in one kernel; the unfused three-kernel
//     sequence when the shape is outside that kernel's cover)
// i.e. 5 launches (7 unfused) against the 13+ ATen kernels of the reference's Block.forward.  Inference form: nothing is saved
// for a backward (training goes through vited_layernorm_fwd / vited_gemm / vited_attention_fwd individually, which save).
// (A single-kernel block was measured against this sequence and lost - DESIGN.md section 9.)
#include "common.h"

static inline int64_t align256(int64_t v) { return (v + 255) & ~(int64_t)255; }

extern "C" int64_t vited_block_workspace_bytes(int64_t batch, int64_t tokens, int64_t dim, int64_t hidden, int heads) {
    const int64_t M = batch * tokens;
    // h (bf16 [M, D]) | qkv (bf16 [M, 3D]) | o (bf16 [M, D]) | lse (f32 [B, H, N]) | x' (f32 [M, D]) | mean, rstd (f32 [M]) |
    // unfused MLP only: gd, u (bf16 [M, hidden])
    return align256(M * dim * 2) + align256(M * 3 * dim * 2) + align256(M * dim * 2) + align256(batch * heads * tokens * 4) +
           align256(M * dim * 4) + 2 * align256(M * 4) + 2 * align256(M * hidden * 2);
}

extern "C" int vited_block_fwd(const float* x, float* y, int64_t batch, int64_t tokens, int64_t dim, int heads, int64_t hidden,
                               const float* ln1_g, const float* ln1_b, const void* wqkv, const float* bqkv, const void* wproj,
                               const float* bproj, const float* ln2_g, const float* ln2_b, const void* w1, const float* b1,
                               const void* w2, const float* b2, float eps, void* workspace, int64_t workspace_bytes, void* stream) {
    if (!x || !y || !ln1_g || !ln1_b || !wqkv || !wproj || !bproj || !ln2_g || !ln2_b || !w1 || !b1 || !w2 || !b2 || !workspace)
        return VITED_ERR_BAD_ARG;
    if (batch <= 0 || tokens <= 0 || dim <= 0 || heads <= 0 || hidden <= 0 || dim % heads) return VITED_ERR_BAD_ARG;
    if (workspace_bytes < vited_block_workspace_bytes(batch, tokens, dim, hidden, heads)) return VITED_ERR_WORKSPACE;
    const int64_t M = batch * tokens;
    const int hd = (int)(dim / heads);
    char* w = (char*)workspace;
    void* h = w;            w += align256(M * dim * 2);
    char* qkv = w;          w += align256(M * 3 * dim * 2);
    void* o = w;            w += align256(M * dim * 2);
    float* lse = (float*)w; w += align256(batch * heads * tokens * 4);
    float* xa = (float*)w;  w += align256(M * dim * 4);
    float* mean = (float*)w; w += align256(M * 4);
    float* rstd = (float*)w; w += align256(M * 4);
    void* gd = w;           w += align256(M * hidden * 2);
    void* u = w;
    int rc;
    // x' = x + proj(attention(qkv(LN1(x))))
    if ((rc = vited_layernorm_fwd(x, dim, ln1_g, ln1_b, h, VITED_BF16, dim, mean, rstd, M, dim, eps, stream))) return rc;
    if ((rc = vited_gemm(h, dim, wqkv, dim, VITED_B_NK, VITED_BF16, M, 3 * dim, dim, VITED_EPI_STORE, bqkv, nullptr, nullptr, qkv, nullptr,
                         3 * dim, 0, 0, 0, 0, stream))) return rc;
    if ((rc = vited_attention_fwd(qkv, tokens * 3 * dim, 3 * dim, qkv + dim * 2, tokens * 3 * dim, 3 * dim, qkv + 2 * dim * 2, tokens * 3 * dim,
                                  3 * dim, o, tokens * dim, dim, lse, VITED_BF16, batch, heads, tokens, tokens, hd, 1.0f / sqrtf((float)hd),
                                  stream))) return rc;
    if ((rc = vited_gemm(o, dim, wproj, dim, VITED_B_NK, VITED_BF16, M, dim, dim, VITED_EPI_RESIDUAL, bproj, nullptr, x, xa, nullptr, dim, 0, 0,
                         0, 0, stream))) return rc;
    // y = x' + fc2(gelu(fc1(LN2(x'))))
    rc = vited_mlp_fwd(xa, dim, ln2_g, ln2_b, w1, b1, w2, b2, y, dim, nullptr, nullptr, nullptr, nullptr, nullptr, M, dim, hidden, eps, stream);
    if (rc != VITED_ERR_UNSUPPORTED) return rc;
    if ((rc = vited_layernorm_fwd(xa, dim, ln2_g, ln2_b, h, VITED_BF16, dim, mean, rstd, M, dim, eps, stream))) return rc;
    if ((rc = vited_gemm(h, dim, w1, dim, VITED_B_NK, VITED_BF16, M, hidden, dim, VITED_EPI_GELU_GRAD, b1, nullptr, nullptr, gd, u, hidden, 0, 0,
                         0, 0, stream))) return rc;
    return vited_gemm(u, hidden, w2, hidden, VITED_B_NK, VITED_BF16, M, dim, hidden, VITED_EPI_RESIDUAL, b2, nullptr, xa, y, nullptr, dim, 0, 0, 0,
                      0, stream);
}

// ------------------------------------------------------------------------------------------------
// vited_cross_block_fwd: one decoder CrossBlock forward (vision_transformer.py:268-272) behind one C-ABI call - the
// "cross_block" entry of SURVEY.md section 8(b).  Launch sequence on the caller's stream (inference form, nothing saved):
//     LayerNorm(norm1) -> qkv GEMM -> self-attention -> proj + residual WITH norm_cross fused (gemm_row.hip) ->
//     q GEMM | LayerNorm(norm_context) -> kv GEMM -> cross-attention (queries from image 2, keys / values from the image-1
//     features) -> cross-proj + residual -> fused MLP branch (vited_mlp_fwd; the three-kernel sequence outside its cover)
// 11 launches (13 unfused) against the ~30 ATen kernels of the reference's CrossBlock.forward.
// ------------------------------------------------------------------------------------------------
extern "C" int64_t vited_cross_block_workspace_bytes(int64_t batch, int64_t tokens, int64_t ctx_tokens, int64_t dim, int64_t hidden, int heads) {
    const int64_t M = batch * tokens, Mc = batch * ctx_tokens;
    // h (bf16 [max(M, Mc), D]) | qkv (bf16 [M, 3D]) | o (bf16 [M, D]) | lse (f32 [B, H, N]) | xa, xb (f32 [M, D]) | mean, rstd (f32 [max]) |
    // hq, q (bf16 [M, D]) | kv (bf16 [Mc, 2D]) | unfused MLP only: gd, u (bf16 [M, hidden])
    const int64_t Mx = M > Mc ? M : Mc;
    return align256(Mx * dim * 2) + align256(M * 3 * dim * 2) + align256(M * dim * 2) + align256(batch * heads * tokens * 4) +
           2 * align256(M * dim * 4) + 2 * align256(Mx * 4) + 2 * align256(M * dim * 2) + align256(Mc * 2 * dim * 2) +
           2 * align256(M * hidden * 2);
}

extern "C" int vited_cross_block_fwd(const float* x, const float* context, float* y, int64_t batch, int64_t tokens, int64_t ctx_tokens,
                                     int64_t dim, int heads, int64_t hidden, const float* ln1_g, const float* ln1_b, const void* wqkv,
                                     const float* bqkv, const void* wproj, const float* bproj, const float* lnq_g, const float* lnq_b,
                                     const float* lnc_g, const float* lnc_b, const void* wq, const float* bq, const void* wkv,
                                     const float* bkv, const void* wcproj, const float* bcproj, const float* ln2_g, const float* ln2_b,
                                     const void* w1, const float* b1, const void* w2, const float* b2, float eps, void* workspace,
                                     int64_t workspace_bytes, void* stream) {
    if (!x || !context || !y || !ln1_g || !ln1_b || !wqkv || !wproj || !bproj || !lnq_g || !lnq_b || !lnc_g || !lnc_b || !wq || !wkv ||
        !wcproj || !bcproj || !ln2_g || !ln2_b || !w1 || !b1 || !w2 || !b2 || !workspace)
        return VITED_ERR_BAD_ARG;
    if (batch <= 0 || tokens <= 0 || ctx_tokens <= 0 || dim <= 0 || heads <= 0 || hidden <= 0 || dim % heads) return VITED_ERR_BAD_ARG;
    if (workspace_bytes < vited_cross_block_workspace_bytes(batch, tokens, ctx_tokens, dim, hidden, heads)) return VITED_ERR_WORKSPACE;
    const int64_t M = batch * tokens, Mc = batch * ctx_tokens, Mx = M > Mc ? M : Mc;
    const int hd = (int)(dim / heads);
    const float scale = 1.0f / sqrtf((float)hd);
    char* w = (char*)workspace;
    void* h = w;             w += align256(Mx * dim * 2);
    char* qkv = w;           w += align256(M * 3 * dim * 2);
    void* o = w;             w += align256(M * dim * 2);
    float* lse = (float*)w;  w += align256(batch * heads * tokens * 4);
    float* xa = (float*)w;   w += align256(M * dim * 4);
    float* xb = (float*)w;   w += align256(M * dim * 4);
    float* mean = (float*)w; w += align256(Mx * 4);
    float* rstd = (float*)w; w += align256(Mx * 4);
    void* hq = w;            w += align256(M * dim * 2);
    void* q = w;             w += align256(M * dim * 2);
    char* kv = w;            w += align256(Mc * 2 * dim * 2);
    void* gd = w;            w += align256(M * hidden * 2);
    void* u = w;
    int rc;
    // x' = x + proj(attention(qkv(norm1(x)))), and hq = norm_cross(x') from the same kernel when the row-complete tile covers the shape
    if ((rc = vited_layernorm_fwd(x, dim, ln1_g, ln1_b, h, VITED_BF16, dim, mean, rstd, M, dim, eps, stream))) return rc;
    if ((rc = vited_gemm(h, dim, wqkv, dim, VITED_B_NK, VITED_BF16, M, 3 * dim, dim, VITED_EPI_STORE, bqkv, nullptr, nullptr, qkv, nullptr,
                         3 * dim, 0, 0, 0, 0, stream))) return rc;
    if ((rc = vited_attention_fwd(qkv, tokens * 3 * dim, 3 * dim, qkv + dim * 2, tokens * 3 * dim, 3 * dim, qkv + 2 * dim * 2, tokens * 3 * dim,
                                  3 * dim, o, tokens * dim, dim, lse, VITED_BF16, batch, heads, tokens, tokens, hd, scale, stream))) return rc;
    if (vited_linear_layernorm_supported(M, dim, dim)) {
        if ((rc = vited_linear_residual_layernorm_fwd(o, dim, wproj, dim, bproj, x, dim, xa, dim, lnq_g, lnq_b, eps, hq, dim, mean, rstd, M, dim,
                                                      dim, stream))) return rc;
    } else {
        if ((rc = vited_gemm(o, dim, wproj, dim, VITED_B_NK, VITED_BF16, M, dim, dim, VITED_EPI_RESIDUAL, bproj, nullptr, x, xa, nullptr, dim, 0,
                             0, 0, 0, stream))) return rc;
        if ((rc = vited_layernorm_fwd(xa, dim, lnq_g, lnq_b, hq, VITED_BF16, dim, mean, rstd, M, dim, eps, stream))) return rc;
    }
    // x'' = x' + proj(attention(q(norm_cross(x')), kv(norm_context(context))))   (kv columns [2][h][hd], :178)
    if ((rc = vited_gemm(hq, dim, wq, dim, VITED_B_NK, VITED_BF16, M, dim, dim, VITED_EPI_STORE, bq, nullptr, nullptr, q, nullptr, dim, 0, 0, 0, 0,
                         stream))) return rc;
    if ((rc = vited_layernorm_fwd(context, dim, lnc_g, lnc_b, h, VITED_BF16, dim, mean, rstd, Mc, dim, eps, stream))) return rc;
    if ((rc = vited_gemm(h, dim, wkv, dim, VITED_B_NK, VITED_BF16, Mc, 2 * dim, dim, VITED_EPI_STORE, bkv, nullptr, nullptr, kv, nullptr, 2 * dim,
                         0, 0, 0, 0, stream))) return rc;
    if ((rc = vited_attention_fwd(q, tokens * dim, dim, kv, ctx_tokens * 2 * dim, 2 * dim, kv + dim * 2, ctx_tokens * 2 * dim, 2 * dim, o,
                                  tokens * dim, dim, lse, VITED_BF16, batch, heads, tokens, ctx_tokens, hd, scale, stream))) return rc;
    if ((rc = vited_gemm(o, dim, wcproj, dim, VITED_B_NK, VITED_BF16, M, dim, dim, VITED_EPI_RESIDUAL, bcproj, nullptr, xa, xb, nullptr, dim, 0, 0,
                         0, 0, stream))) return rc;
    // y = x'' + fc2(gelu(fc1(norm2(x''))))
    rc = vited_mlp_fwd(xb, dim, ln2_g, ln2_b, w1, b1, w2, b2, y, dim, nullptr, nullptr, nullptr, nullptr, nullptr, M, dim, hidden, eps, stream);
    if (rc != VITED_ERR_UNSUPPORTED) return rc;
    if ((rc = vited_layernorm_fwd(xb, dim, ln2_g, ln2_b, h, VITED_BF16, dim, mean, rstd, M, dim, eps, stream))) return rc;
    if ((rc = vited_gemm(h, dim, w1, dim, VITED_B_NK, VITED_BF16, M, hidden, dim, VITED_EPI_GELU_GRAD, b1, nullptr, nullptr, gd, u, hidden, 0, 0,
                         0, 0, stream))) return rc;
    return vited_gemm(u, hidden, w2, hidden, VITED_B_NK, VITED_BF16, M, dim, hidden, VITED_EPI_RESIDUAL, b2, nullptr, xb, y, nullptr, dim, 0, 0, 0,
                      0, stream);
}
