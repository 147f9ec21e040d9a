// LayerNorm forward/backward for the ViT-ED residual stream (fp32 in, activation dtype out).
// HBM-bound: one 64-lane wave per token row, row held in registers (dim 384 = 6 values/lane),
// butterfly reductions, no LDS in the forward.  The backward also produces the per-column
// gamma/beta partial sums (deterministic: per-workgroup partials + one finishing pass).
#include "common.h"

#define LN_MAX_PER_LANE 16  // supports dim <= 1024

template <typename T>
__global__ void __launch_bounds__(256)
layernorm_fwd_kernel(const float* __restrict__ x, int64_t x_ld, const float* __restrict__ gamma,
                     const float* __restrict__ beta, T* __restrict__ y, int64_t y_ld, float* __restrict__ mean,
                     float* __restrict__ rstd, int64_t rows, int dim, float eps) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int per = (dim + 63) / 64;
    for (int64_t r = wave; r < rows; r += nwaves) {
        const float* xr = x + r * x_ld;
        float v[LN_MAX_PER_LANE];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < LN_MAX_PER_LANE; ++j) {
            if (j < per) {
                const int c = j * 64 + lane;
                v[j] = c < dim ? xr[c] : 0.f;
                s += v[j];
            }
        }
        const float mu = wave_sum(s) / dim;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < LN_MAX_PER_LANE; ++j) {
            if (j < per) {
                const int c = j * 64 + lane;
                const float d = c < dim ? v[j] - mu : 0.f;
                q += d * d;
            }
        }
        const float rs = rsqrtf(wave_sum(q) / dim + eps);
        T* yr = y + r * y_ld;
#pragma unroll
        for (int j = 0; j < LN_MAX_PER_LANE; ++j) {
            if (j < per) {
                const int c = j * 64 + lane;
                if (c < dim) yr[c] = from_f32<T>((v[j] - mu) * rs * gamma[c] + beta[c]);
            }
        }
        if (lane == 0) {
            mean[r] = mu;
            rstd[r] = rs;
        }
    }
}

extern "C" int vited_layernorm_fwd(const float* x, int64_t x_ld, const float* gamma, const float* beta, void* y,
                                   int y_dtype, int64_t y_ld, float* mean, float* rstd, int64_t rows, int64_t dim,
                                   float eps, void* stream) {
    if (!x || !gamma || !beta || !y || !mean || !rstd || rows <= 0 || dim <= 0 || x_ld < dim || y_ld < dim) return VITED_ERR_BAD_ARG;
    if (dim > 64 * LN_MAX_PER_LANE) return VITED_ERR_UNSUPPORTED;
    int64_t blocks = ceil_div64(rows, 4);
    if (blocks > 8192) blocks = 8192;
    hipStream_t s = (hipStream_t)stream;
    if (y_dtype == VITED_BF16)
        hipLaunchKernelGGL((layernorm_fwd_kernel<bf16>), dim3(blocks), dim3(256), 0, s, x, x_ld, gamma, beta, (bf16*)y, y_ld, mean, rstd, rows, (int)dim, eps);
    else if (y_dtype == VITED_F32)
        hipLaunchKernelGGL((layernorm_fwd_kernel<float>), dim3(blocks), dim3(256), 0, s, x, x_ld, gamma, beta, (float*)y, y_ld, mean, rstd, rows, (int)dim, eps);
    else
        return VITED_ERR_UNSUPPORTED;
    return vited_check_launch();
}

// ------------------------------------------------------------------------------------------------
// backward
//   xhat = (x - mean) * rstd ; g = dy * gamma
//   dx = rstd * (g - mean_c(g) - xhat * mean_c(g * xhat))  (+ dx_in)
//   dgamma = sum_rows dy * xhat ; dbeta = sum_rows dy
// Each wave walks rows grid-stride and keeps its column partials in registers; the 4 waves of a
// workgroup combine through LDS and write one partial row pair to the workspace
// [gridDim.x][2][dim]; ln_bwd_finish sums those rows.
// ------------------------------------------------------------------------------------------------
template <typename T, typename L>
__global__ void __launch_bounds__(256)
layernorm_bwd_kernel(const T* __restrict__ dy, int64_t dy_ld, const float* __restrict__ x, int64_t x_ld,
                     const float* __restrict__ gamma, const float* __restrict__ mean, const float* __restrict__ rstd,
                     const float* __restrict__ dx_in, int64_t dx_in_ld, float* __restrict__ dx_out, int64_t dx_out_ld,
                     L* __restrict__ dx_lp, int64_t dx_lp_ld, float* __restrict__ partial, int64_t rows, int dim) {
    extern __shared__ float lds[];  // [4 waves][2][dim]
    const int lane = threadIdx.x & 63;
    const int wid = threadIdx.x >> 6;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int per = (dim + 63) / 64;
    float dg[LN_MAX_PER_LANE], db[LN_MAX_PER_LANE], gm[LN_MAX_PER_LANE];
#pragma unroll
    for (int j = 0; j < LN_MAX_PER_LANE; ++j) {
        dg[j] = 0.f;
        db[j] = 0.f;
        const int c = j * 64 + lane;
        gm[j] = (j < per && c < dim) ? gamma[c] : 0.f;
    }
    for (int64_t r = wave; r < rows; r += nwaves) {
        const float mu = mean[r], rs = rstd[r];
        const T* dyr = dy + r * dy_ld;
        const float* xr = x + r * x_ld;
        float xh[LN_MAX_PER_LANE], g[LN_MAX_PER_LANE];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < LN_MAX_PER_LANE; ++j) {
            if (j < per) {
                const int c = j * 64 + lane;
                const bool ok = c < dim;
                const float d = ok ? to_f32(dyr[c]) : 0.f;
                xh[j] = ok ? (xr[c] - mu) * rs : 0.f;
                g[j] = d * gm[j];
                s1 += g[j];
                s2 += g[j] * xh[j];
                dg[j] += d * xh[j];
                db[j] += d;
            }
        }
        const float c1 = wave_sum(s1) / dim, c2 = wave_sum(s2) / dim;
#pragma unroll
        for (int j = 0; j < LN_MAX_PER_LANE; ++j) {
            if (j < per) {
                const int c = j * 64 + lane;
                if (c < dim) {
                    float v = rs * (g[j] - c1 - xh[j] * c2);
                    if (dx_in) v += dx_in[r * dx_in_ld + c];
                    dx_out[r * dx_out_ld + c] = v;
                    if (dx_lp) dx_lp[r * dx_lp_ld + c] = from_f32<L>(v);
                }
            }
        }
    }
    // combine the 4 waves' column partials
    float* my = lds + (size_t)wid * 2 * dim;
#pragma unroll
    for (int j = 0; j < LN_MAX_PER_LANE; ++j) {
        if (j < per) {
            const int c = j * 64 + lane;
            if (c < dim) {
                my[c] = dg[j];
                my[dim + c] = db[j];
            }
        }
    }
    __syncthreads();
    float* out = partial + (size_t)blockIdx.x * 2 * dim;
    for (int c = threadIdx.x; c < 2 * dim; c += blockDim.x)
        out[c] = (lds[c] + lds[2 * dim + c]) + (lds[4 * dim + c] + lds[6 * dim + c]);
}

__global__ void ln_bwd_finish_kernel(const float* __restrict__ partial, int nparts, int dim, float* __restrict__ dgamma,
                                     float* __restrict__ dbeta) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= 2 * dim) return;
    float a0 = 0.f, a1 = 0.f;
    int p = 0;
    for (; p + 1 < nparts; p += 2) {
        a0 += partial[(size_t)p * 2 * dim + c];
        a1 += partial[(size_t)(p + 1) * 2 * dim + c];
    }
    if (p < nparts) a0 += partial[(size_t)p * 2 * dim + c];
    const float v = a0 + a1;
    if (c < dim) dgamma[c] = v; else dbeta[c - dim] = v;
}

static inline int64_t ln_bwd_blocks(int64_t rows) {
    int64_t b = ceil_div64(rows, 4 * 8);  // >= 8 rows per wave so the partial pass stays small
    if (b > 1024) b = 1024;
    if (b < 1) b = 1;
    return b;
}

extern "C" int64_t vited_layernorm_bwd_workspace_bytes(int64_t rows, int64_t dim) {
    return ln_bwd_blocks(rows) * 2 * dim * (int64_t)sizeof(float);
}

template <typename T, typename L>
static int ln_bwd_launch(const void* dy, int64_t dy_ld, const float* x, int64_t x_ld, const float* gamma, const float* mean,
                         const float* rstd, const float* dx_in, int64_t dx_in_ld, float* dx_out, int64_t dx_out_ld, void* dx_lp,
                         int64_t dx_lp_ld, float* dgamma, float* dbeta, int64_t rows, int dim, float* ws, hipStream_t s) {
    const int64_t blocks = ln_bwd_blocks(rows);
    const size_t lds = (size_t)4 * 2 * dim * sizeof(float);
    hipLaunchKernelGGL((layernorm_bwd_kernel<T, L>), dim3(blocks), dim3(256), lds, s, (const T*)dy, dy_ld, x, x_ld, gamma, mean,
                       rstd, dx_in, dx_in_ld, dx_out, dx_out_ld, (L*)dx_lp, dx_lp_ld, ws, rows, dim);
    hipLaunchKernelGGL(ln_bwd_finish_kernel, dim3((2 * dim + 255) / 256), dim3(256), 0, s, ws, (int)blocks, dim, dgamma, dbeta);
    return vited_check_launch();
}

extern "C" int vited_layernorm_bwd(const void* dy, int dy_dtype, int64_t dy_ld, const float* x, int64_t x_ld,
                                   const float* gamma, const float* mean, const float* rstd, const float* dx_in,
                                   int64_t dx_in_ld, float* dx_out, int64_t dx_out_ld, void* dx_lp, int dx_lp_dtype,
                                   int64_t dx_lp_ld, float* dgamma, float* dbeta, int64_t rows, int64_t dim,
                                   float* workspace, int64_t workspace_bytes, void* stream) {
    if (!dy || !x || !gamma || !mean || !rstd || !dx_out || !dgamma || !dbeta || rows <= 0 || dim <= 0) return VITED_ERR_BAD_ARG;
    if (dim > 64 * LN_MAX_PER_LANE) return VITED_ERR_UNSUPPORTED;
    if (!workspace || workspace_bytes < vited_layernorm_bwd_workspace_bytes(rows, dim)) return VITED_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int d = (int)dim;
    if (dx_lp && dx_lp_dtype != VITED_BF16) return VITED_ERR_UNSUPPORTED;
    if (dy_dtype == VITED_BF16)
        return ln_bwd_launch<bf16, bf16>(dy, dy_ld, x, x_ld, gamma, mean, rstd, dx_in, dx_in_ld, dx_out, dx_out_ld, dx_lp, dx_lp_ld, dgamma, dbeta, rows, d, workspace, s);
    if (dy_dtype == VITED_F32)
        return ln_bwd_launch<float, bf16>(dy, dy_ld, x, x_ld, gamma, mean, rstd, dx_in, dx_in_ld, dx_out, dx_out_ld, dx_lp, dx_lp_ld, dgamma, dbeta, rows, d, workspace, s);
    return VITED_ERR_UNSUPPORTED;
}
