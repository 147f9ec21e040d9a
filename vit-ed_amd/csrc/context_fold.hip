// norm_context + kv projection of ALL decoder blocks as one GEMM (CrossBlock, models/vision_transformer.py:245,269-270 and
// CrossAttention :152,177-179).
//
// Every CrossBlock applies ITS OWN LayerNorm (norm_context: gamma_l, beta_l) to the SAME encoder features before its kv
// projection.  The normalised features xhat = (x - mean) * rstd do not depend on the block, so
//     kv_l = (xhat * gamma_l + beta_l) W_l^T + b_l = xhat (W_l o gamma_l)^T + (W_l beta_l + b_l)
// and the c_depth LayerNorm launches + c_depth kv GEMMs become ONE LayerNorm (gamma = 1, beta = 0) and ONE GEMM against the
// folded weights of all blocks stacked ([c_depth * 2 D, D]); in backward the c_depth input-gradient GEMMs + LayerNorm backwards
// become one row-complete kernel with K = c_depth * 2 D, and the gradients of the folded weights unfold as
//     dW_l[n, k] = dW'_l[n, k] gamma_l[k] + db'_l[n] beta_l[k]   (W_l enters W'_l AND b'_l),
//     dgamma_l[k] = sum_n dW'_l[n, k] W_l[n, k],   dbeta_l[k] = sum_n db'_l[n] W_l[n, k],   db_l = db'_l.
// The two kernels here touch c_depth * 2 D * D elements (2.4 M at config A): microseconds.
#include "common.h"

#define CF_MAX_BLOCKS 16

struct FoldArgs {
    const float* w[CF_MAX_BLOCKS];       // [N, K] kv weights
    const float* bias[CF_MAX_BLOCKS];    // [N] or null
    const float* gamma[CF_MAX_BLOCKS];   // [K]
    const float* beta[CF_MAX_BLOCKS];    // [K]
    bf16* w_out;                         // [count * N, K]   folded weights (forward NT operand)
    bf16* wt_out;                        // [K, count * N]   their transpose (input-gradient NT operand)
    float* bias_out;                     // [count * N]
    int count, N, K;
};

// one wave per output row (l, n): lanes over k
__global__ void __launch_bounds__(256) fold_context_kernel(const FoldArgs a) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);          // l * N + n
    if (row >= a.count * a.N) return;
    const int l = row / a.N, n = row - l * a.N;
    const float* __restrict__ w = a.w[l] + (int64_t)n * a.K;
    const float* __restrict__ g = a.gamma[l];
    const float* __restrict__ be = a.beta[l];
    const int64_t ldt = (int64_t)a.count * a.N;
    float dot = 0.f;
    for (int k = lane; k < a.K; k += 64) {
        const float wv = w[k];
        const bf16 f = (bf16)(wv * g[k]);
        a.w_out[(int64_t)row * a.K + k] = f;
        a.wt_out[(int64_t)k * ldt + row] = f;
        dot = fmaf(wv, be[k], dot);
    }
    dot = wave_sum(dot);
    if (lane == 0) a.bias_out[row] = dot + (a.bias[l] ? a.bias[l][n] : 0.f);
}

extern "C" int vited_fold_context_weights(int count, const float* const* w, const float* const* bias, const float* const* gamma,
                                          const float* const* beta, int64_t N, int64_t K, void* w_out, void* wt_out, float* bias_out,
                                          void* stream) {
    if (count < 1 || count > CF_MAX_BLOCKS || !w || !bias || !gamma || !beta || !w_out || !wt_out || !bias_out || N <= 0 || K <= 0)
        return VITED_ERR_BAD_ARG;
    FoldArgs a = {};
    for (int i = 0; i < count; ++i) {
        if (!w[i] || !gamma[i] || !beta[i]) return VITED_ERR_BAD_ARG;
        a.w[i] = w[i]; a.bias[i] = bias[i]; a.gamma[i] = gamma[i]; a.beta[i] = beta[i];
    }
    a.w_out = (bf16*)w_out; a.wt_out = (bf16*)wt_out; a.bias_out = bias_out;
    a.count = count; a.N = (int)N; a.K = (int)K;
    hipLaunchKernelGGL(fold_context_kernel, dim3((unsigned)ceil_div64((int64_t)count * N, 4)), dim3(256), 0, (hipStream_t)stream, a);
    return vited_check_launch();
}

struct UnfoldArgs {
    const float* w[CF_MAX_BLOCKS];
    const float* gamma[CF_MAX_BLOCKS];
    const float* beta[CF_MAX_BLOCKS];
    float* dw[CF_MAX_BLOCKS];            // [N, K]
    float* dbias[CF_MAX_BLOCKS];         // [N] or null
    float* dgamma[CF_MAX_BLOCKS];        // [K]
    float* dbeta[CF_MAX_BLOCKS];         // [K]
    const float* dwf;                    // [count * N, K]  gradient of the folded weights
    const float* dbf;                    // [count * N]     gradient of the folded bias
    int count, N, K, accumulate;
};

// block = (l, 64 columns k): 1024 threads = 64 columns x 16 row groups (48 rows each at N = 768: the loads of a thread are
// independent, 4 in flight); column sums over n combined through LDS
#define UF_GROUPS 16
__global__ void __launch_bounds__(64 * UF_GROUPS) unfold_context_kernel(const UnfoldArgs a) {
    __shared__ float red[2][UF_GROUPS][64];
    const int l = blockIdx.y, kx = threadIdx.x & 63, ng = threadIdx.x >> 6;
    const int k = blockIdx.x * 64 + kx;
    const bool ok = k < a.K;
    const float* __restrict__ w = a.w[l];
    const float* __restrict__ dwf = a.dwf + (int64_t)l * a.N * a.K;
    const float* __restrict__ dbf = a.dbf + (int64_t)l * a.N;
    float* __restrict__ dw = a.dw[l];
    const float g = ok ? a.gamma[l][k] : 0.f, be = ok ? a.beta[l][k] : 0.f;
    float sg = 0.f, sb = 0.f;
#pragma unroll 4
    for (int n = ng; n < a.N; n += UF_GROUPS) {
        if (ok) {
            const int64_t o = (int64_t)n * a.K + k;
            const float d = dwf[o], wv = w[o], dbn = dbf[n];
            sg = fmaf(d, wv, sg);
            sb = fmaf(dbn, wv, sb);
            const float v = fmaf(d, g, dbn * be);
            dw[o] = a.accumulate ? dw[o] + v : v;
        }
    }
    red[0][ng][kx] = sg;
    red[1][ng][kx] = sb;
    __syncthreads();
    if (ng == 0 && ok) {
        float tg = 0.f, tb = 0.f;
#pragma unroll
        for (int w = 0; w < UF_GROUPS; ++w) {
            tg += red[0][w][kx];
            tb += red[1][w][kx];
        }
        a.dgamma[l][k] = a.accumulate ? a.dgamma[l][k] + tg : tg;
        a.dbeta[l][k] = a.accumulate ? a.dbeta[l][k] + tb : tb;
    }
    if (blockIdx.x == 0 && a.dbias[l]) {
        for (int n = threadIdx.x; n < a.N; n += 64 * UF_GROUPS) a.dbias[l][n] = a.accumulate ? a.dbias[l][n] + dbf[n] : dbf[n];
    }
}

extern "C" int vited_unfold_context_grads(int count, const float* dwf, const float* dbf, const float* const* w, const float* const* gamma,
                                          const float* const* beta, float* const* dw, float* const* dbias, float* const* dgamma, float* const* dbeta, int64_t N,
                                          int64_t K, int accumulate, void* stream) {
    if (count < 1 || count > CF_MAX_BLOCKS || !dwf || !dbf || !w || !gamma || !beta || !dw || !dbias || !dgamma || !dbeta || N <= 0 || K <= 0)
        return VITED_ERR_BAD_ARG;
    UnfoldArgs a = {};
    for (int i = 0; i < count; ++i) {
        if (!w[i] || !gamma[i] || !beta[i] || !dw[i] || !dgamma[i] || !dbeta[i]) return VITED_ERR_BAD_ARG;
        a.w[i] = w[i]; a.gamma[i] = gamma[i]; a.beta[i] = beta[i]; a.dw[i] = dw[i]; a.dbias[i] = dbias[i]; a.dgamma[i] = dgamma[i]; a.dbeta[i] = dbeta[i];
    }
    a.dwf = dwf; a.dbf = dbf; a.count = count; a.N = (int)N; a.K = (int)K; a.accumulate = accumulate;
    hipLaunchKernelGGL(unfold_context_kernel, dim3((unsigned)ceil_div64(K, 64), (unsigned)count), dim3(64 * UF_GROUPS), 0, (hipStream_t)stream, a);
    return vited_check_launch();
}
