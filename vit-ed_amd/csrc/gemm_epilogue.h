// Epilogues shared by the portable and the MFMA GEMM kernels (see VITED_EPI_* in vited.h).
#pragma once
#include "common.h"

struct EpiParams {
    const float* bias;      // [N] or null
    const void* aux;        // T [M, N] (ldo)   - MUL_GELU_GRAD, MUL
    const float* residual;  // fp32              - RESIDUAL
    void* out;              // T [M, N] (ldo), fp32 for RESIDUAL
    void* out2;             // T [M, N] (ldo)   - GELU, GELU_GRAD
    int64_t ldo;
    int64_t rows_per_batch, out_rows_per_batch, row_offset;
    int residual_bcast;
};

// one output element (m, n) with accumulator value acc
template <typename T, int EPI>
__device__ __forceinline__ void epilogue_store(const EpiParams& p, int64_t m, int64_t n, float acc) {
    if (p.bias) acc += p.bias[n];
    if constexpr (EPI == VITED_EPI_STORE) {
        ((T*)p.out)[m * p.ldo + n] = from_f32<T>(acc);
    } else if constexpr (EPI == VITED_EPI_STORE_F32) {
        ((float*)p.out)[m * p.ldo + n] = acc;
    } else if constexpr (EPI == VITED_EPI_GELU) {
        ((T*)p.out)[m * p.ldo + n] = from_f32<T>(acc);
        ((T*)p.out2)[m * p.ldo + n] = from_f32<T>(gelu_f(acc));
    } else if constexpr (EPI == VITED_EPI_GELU_GRAD) {
        ((T*)p.out)[m * p.ldo + n] = from_f32<T>(gelu_grad_f(acc));
        ((T*)p.out2)[m * p.ldo + n] = from_f32<T>(gelu_f(acc));
    } else if constexpr (EPI == VITED_EPI_MUL) {
        ((T*)p.out)[m * p.ldo + n] = from_f32<T>(acc * to_f32(((const T*)p.aux)[m * p.ldo + n]));
    } else if constexpr (EPI == VITED_EPI_RESIDUAL) {
        int64_t orow = m, rrow = m;
        if (p.rows_per_batch > 0) {
            const int64_t b = m / p.rows_per_batch, r = m - b * p.rows_per_batch + p.row_offset;
            orow = b * p.out_rows_per_batch + r;
            rrow = p.residual_bcast ? r : orow;
        }
        ((float*)p.out)[orow * p.ldo + n] = p.residual[rrow * p.ldo + n] + acc;
    } else {  // VITED_EPI_MUL_GELU_GRAD
        const float z = to_f32(((const T*)p.aux)[m * p.ldo + n]);
        ((T*)p.out)[m * p.ldo + n] = from_f32<T>(acc * gelu_grad_f(z));
    }
}
