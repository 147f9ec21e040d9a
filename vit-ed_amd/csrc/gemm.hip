// C-ABI GEMM entry points: argument validation and dispatch between the bf16 MFMA kernels and the
// portable fp32-FMA kernels.
#include <stdlib.h>
#include <string.h>

#include "gemm_kernels.h"

static thread_local int g_last_gemm_path = 0;
extern "C" int vited_last_gemm_path(void) { return g_last_gemm_path; }

extern "C" int vited_gemm(const void* A, int64_t lda, const void* B, int64_t ldb, int b_layout, int dtype, int64_t M,
                          int64_t N, int64_t K, int epilogue, const float* bias, const void* aux, const float* residual,
                          void* out, void* out2, int64_t ldo, int64_t rows_per_batch, int64_t out_rows_per_batch,
                          int64_t row_offset, int residual_bcast, void* stream) {
    if (!A || !B || !out || M <= 0 || N <= 0 || K <= 0 || lda < K || ldo < N) return VITED_ERR_BAD_ARG;
    if (b_layout != VITED_B_NK && b_layout != VITED_B_KN) return VITED_ERR_BAD_ARG;
    if (ldb < (b_layout == VITED_B_NK ? K : N)) return VITED_ERR_BAD_ARG;
    if ((epilogue == VITED_EPI_GELU || epilogue == VITED_EPI_GELU_GRAD) && !out2) return VITED_ERR_BAD_ARG;
    if (epilogue == VITED_EPI_RESIDUAL && !residual) return VITED_ERR_BAD_ARG;
    if ((epilogue == VITED_EPI_MUL_GELU_GRAD || epilogue == VITED_EPI_MUL) && !aux) return VITED_ERR_BAD_ARG;
    if (rows_per_batch < 0 || (rows_per_batch > 0 && (out_rows_per_batch < rows_per_batch + row_offset || row_offset < 0)))
        return VITED_ERR_BAD_ARG;
    EpiParams ep;
    ep.bias = bias;
    ep.aux = aux;
    ep.residual = residual;
    ep.out = out;
    ep.out2 = out2;
    ep.ldo = ldo;
    ep.rows_per_batch = epilogue == VITED_EPI_RESIDUAL ? rows_per_batch : 0;
    ep.out_rows_per_batch = out_rows_per_batch;
    ep.row_offset = row_offset;
    ep.residual_bcast = residual_bcast;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == VITED_BF16 && b_layout == VITED_B_NK && gemm_nt_mfma_supported(A, lda, B, ldb, M, N, K, epilogue, ep)) {
        g_last_gemm_path = 2;
        return gemm_nt_mfma(A, lda, B, ldb, M, N, K, epilogue, ep, s);
    }
    g_last_gemm_path = 1;
    return gemm_portable(A, lda, B, ldb, b_layout, dtype, M, N, K, epilogue, ep, s);
}

static inline int64_t tn_portable_splits(int64_t M, int64_t N, int64_t K) {
    const int64_t tiles = ceil_div64(N, 64) * ceil_div64(K, 64);
    int64_t s = ceil_div64(1024, tiles);
    const int64_t max_s = ceil_div64(M, 256);
    if (s > max_s) s = max_s;
    if (s < 1) s = 1;
    const int64_t rps = ceil_div64(ceil_div64(M, s), 16) * 16;
    return ceil_div64(M, rps);
}

static inline int64_t max64(int64_t a, int64_t b) { return a > b ? a : b; }

extern "C" int64_t vited_linear_bwd_weight_workspace_bytes(int64_t M, int64_t N, int64_t K) {
    const int64_t s = max64(tn_portable_splits(M, N, K), gemm_tn_mfma_splits(M, N, K));
    // slabs for dW + the row-sum workspace for dbias (placed after the slabs)
    return (s * N * K + s * N + N) * (int64_t)sizeof(float) + vited_sum_rows_workspace_bytes(M, N) + 256;
}

extern "C" int vited_linear_bwd_weight(const void* dY, int64_t lddy, const void* X, int64_t ldx, int dtype, int64_t M,
                                       int64_t N, int64_t K, float* dW, float* dbias, int accumulate, float* workspace,
                                       int64_t workspace_bytes, void* stream) {
    if (!dY || !X || !dW || M <= 0 || N <= 0 || K <= 0 || lddy < N || ldx < K) return VITED_ERR_BAD_ARG;
    if (dtype != VITED_F32 && dtype != VITED_BF16) return VITED_ERR_UNSUPPORTED;
    if (!workspace || workspace_bytes < vited_linear_bwd_weight_workspace_bytes(M, N, K)) return VITED_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const bool mfma = dtype == VITED_BF16 && gemm_tn_mfma_supported(dY, lddy, X, ldx, M, N, K);
    const int64_t splits = mfma ? gemm_tn_mfma_splits(M, N, K) : tn_portable_splits(M, N, K);
    // partial products always go to workspace slabs when accumulating (the sum pass adds onto dW / dbias)
    const bool via_slabs = splits > 1 || accumulate;
    float* slab = via_slabs ? workspace : dW;
    float* bias_tmp = workspace + splits * N * K;            // [splits][N] (MFMA) or [N] (portable)
    int rc;
    if (mfma) {
        g_last_gemm_path = 2;
        float* bias_slab = !dbias ? nullptr : (via_slabs ? bias_tmp : dbias);
        rc = gemm_tn_mfma(dY, lddy, X, ldx, M, N, K, splits, slab, bias_slab, s);
        if (rc != VITED_OK) return rc;
        if (via_slabs) {
            rc = dbias ? sum_slabs_pair(workspace, N * K, dW, N * K, bias_slab, N, dbias, N, splits, s, accumulate)
                       : sum_rows_f32_single_pass(workspace, N * K, dW, splits, N * K, s, accumulate);
        }
        return rc;
    }
    g_last_gemm_path = 1;
    rc = gemm_tn_portable(dY, lddy, X, ldx, dtype, M, N, K, splits, slab, s);
    if (rc != VITED_OK) return rc;
    if (via_slabs) {
        rc = sum_rows_f32_single_pass(workspace, N * K, dW, splits, N * K, s, accumulate);
        if (rc != VITED_OK) return rc;
    }
    if (dbias) {
        float* ws2 = bias_tmp + N;
        const int64_t ws2_bytes = workspace_bytes - (splits * N * K + N) * (int64_t)sizeof(float);
        if (accumulate) {
            rc = vited_sum_rows(dY, dtype, lddy, bias_tmp, M, N, ws2, ws2_bytes, stream);
            if (rc == VITED_OK) rc = sum_rows_f32_single_pass(bias_tmp, N, dbias, 1, N, s, 1);
        } else {
            rc = vited_sum_rows(dY, dtype, lddy, dbias, M, N, ws2, ws2_bytes, stream);
        }
    }
    return rc;
}

// ---- several weight gradients in one launch ----------------------------------------------------------------------------
extern "C" int vited_linear_bwd_weight_batched_supported(int count, const int64_t* M, const int64_t* N, const int64_t* K, int dtype) {
    return dtype == VITED_BF16 && gemm_tn_batch_supported(count, M, N, K) ? 1 : 0;
}

extern "C" int64_t vited_linear_bwd_weight_batched_workspace_bytes(int count, const int64_t* M, const int64_t* N, const int64_t* K) {
    if (!gemm_tn_batch_supported(count, M, N, K)) return 0;
    int64_t splits, ws, bs;
    gemm_tn_batch_layout(count, M, N, K, &splits, &ws, &bs);
    return splits * (ws + bs) * (int64_t)sizeof(float) + 256;
}

extern "C" int vited_linear_bwd_weight_batched(int count, const void* const* dY, const int64_t* lddy, const void* const* X,
                                               const int64_t* ldx, const int64_t* M, const int64_t* N, const int64_t* K,
                                               float* const* dW, float* const* dbias, int dtype, int accumulate, float* workspace,
                                               int64_t workspace_bytes, void* stream) {
    if (count < 1 || !dY || !lddy || !X || !ldx || !M || !N || !K || !dW || !dbias || !workspace) return VITED_ERR_BAD_ARG;
    if (dtype != VITED_BF16 || !gemm_tn_batch_supported(count, M, N, K)) return VITED_ERR_UNSUPPORTED;
    if (workspace_bytes < vited_linear_bwd_weight_batched_workspace_bytes(count, M, N, K)) return VITED_ERR_WORKSPACE;
    int64_t splits, ws, bs;
    gemm_tn_batch_layout(count, M, N, K, &splits, &ws, &bs);
    int has_bias[40];
    int64_t widths[40], nbias[40];
    for (int i = 0; i < count; ++i) {
        if (!dY[i] || !X[i] || !dW[i] || lddy[i] < N[i] || ldx[i] < K[i]) return VITED_ERR_BAD_ARG;
        has_bias[i] = dbias[i] != nullptr;
        widths[i] = N[i] * K[i];
        nbias[i] = N[i];
    }
    float* base = (float*)(((uintptr_t)workspace + 15) & ~(uintptr_t)15);
    float* slabs = base;
    float* bias_slabs = base + splits * ws;
    hipStream_t s = (hipStream_t)stream;
    g_last_gemm_path = 2;
    int rc = gemm_tn_batch(count, dY, lddy, X, ldx, M, N, K, has_bias, slabs, bias_slabs, s);
    if (rc != VITED_OK) return rc;
    return sum_slabs_batch(count, slabs, ws, bias_slabs, bs, splits, widths, dW, nbias, dbias, accumulate, s);
}
