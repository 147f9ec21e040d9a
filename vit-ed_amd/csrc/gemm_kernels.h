// Internal launchers behind vited_gemm / vited_linear_bwd_weight (one translation unit each).
#pragma once
#include "gemm_epilogue.h"

// portable fp32-FMA kernels (gemm_portable.hip)
int gemm_portable(const void* A, int64_t lda, const void* B, int64_t ldb, int b_layout, int dtype, int64_t M, int64_t N,
                  int64_t K, int epilogue, const EpiParams& ep, hipStream_t s);
int gemm_tn_portable(const void* dY, int64_t lddy, const void* X, int64_t ldx, int dtype, int64_t M, int64_t N, int64_t K,
                     int64_t splits, float* out, hipStream_t s);

// bf16 MFMA kernels (gemm_mfma.hip)
bool gemm_nt_mfma_supported(const void* A, int64_t lda, const void* B, int64_t ldb, int64_t M, int64_t N, int64_t K,
                            int epilogue, const EpiParams& ep);
int gemm_nt_mfma(const void* A, int64_t lda, const void* B, int64_t ldb, int64_t M, int64_t N, int64_t K, int epilogue,
                 const EpiParams& ep, hipStream_t s);
bool gemm_tn_mfma_supported(const void* dY, int64_t lddy, const void* X, int64_t ldx, int64_t M, int64_t N, int64_t K);
int64_t gemm_tn_mfma_splits(int64_t M, int64_t N, int64_t K);
int gemm_tn_mfma(const void* dY, int64_t lddy, const void* X, int64_t ldx, int64_t M, int64_t N, int64_t K, int64_t splits,
                 float* out, float* bias_out /* [splits][N] or null */, hipStream_t s);

// several weight-gradient products in one launch of the wide TN kernel (gemm_mfma.hip)
bool gemm_tn_batch_supported(int count, const int64_t* M, const int64_t* N, const int64_t* K);
void gemm_tn_batch_layout(int count, const int64_t* M, const int64_t* N, const int64_t* K, int64_t* splits, int64_t* slab_stride,
                          int64_t* bias_stride);
int gemm_tn_batch(int count, const void* const* dY, const int64_t* lddy, const void* const* X, const int64_t* ldx, const int64_t* M,
                  const int64_t* N, const int64_t* K, const int* has_bias, float* slabs, float* bias_slabs, hipStream_t s);
// elementwise.hip: the slabs of such a launch summed onto the products' dW / dbias in one pass
int sum_slabs_batch(int count, const float* slabs, int64_t slab_stride, const float* bias_slabs, int64_t bias_stride, int64_t splits,
                    const int64_t* widths, float* const* dW, const int64_t* nbias, float* const* dbias, int accumulate, hipStream_t s);

// elementwise.hip: out[r] = sum_b in[b, r] in ONE pass (batch is small: the split-M slabs)
int sum_rows_f32_single_pass(const float* in, int64_t in_ld, float* out, int64_t batch, int64_t width, hipStream_t s,
                             int accumulate = 0);
int sum_slabs_pair(const float* in_a, int64_t ld_a, float* out_a, int64_t width_a, const float* in_b, int64_t ld_b, float* out_b,
                   int64_t width_b, int64_t batch, hipStream_t s, int accumulate);
