// LayerNorm forward/backward for the ViT-ED residual stream (fp32 in, activation dtype out).
// HBM-bound streaming kernels: a 32-lane half-wave owns one token row, every lane moves 16-byte
// vectors (dim 384 = 3 float4 per lane), row statistics are DPP row sums + one swizzle inside the half-wave,
// no LDS in the forward.  The backward also produces the gamma/beta column sums: per-lane register
// partials over the rows a lane sees -> one LDS combine per workgroup -> [blocks][2][dim] partial
// slabs -> a small deterministic finishing pass (no atomics).
#include "common.h"

#define LN_MAX_VPL 8  // float4 vectors per lane: dim <= 32 * 4 * 8 = 1024

__device__ __forceinline__ float half_sum(float v) { return half_wave_sum(v); }   // over the 32 lanes of a half-wave (common.h)

template <typename T> struct Vec4;
template <> struct Vec4<float> {
    static __device__ __forceinline__ f32x4 load(const float* p) { return *(const f32x4*)p; }
    static __device__ __forceinline__ void store(float* p, const f32x4& v) { *(f32x4*)p = v; }
};
template <> struct Vec4<bf16> {
    static __device__ __forceinline__ f32x4 load(const bf16* p) {
        const bf16x4 v = *(const bf16x4*)p;
        return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
    }
    static __device__ __forceinline__ void store(bf16* p, const f32x4& v) {
        *(bf16x4*)p = bf16x4{(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
    }
};

template <typename T, int VPL>
__global__ void __launch_bounds__(256)
layernorm_fwd_kernel(const float* __restrict__ x, int64_t x_ld, const float* __restrict__ gamma,
                     const float* __restrict__ beta, T* __restrict__ y, int64_t y_ld, float* __restrict__ mean,
                     float* __restrict__ rstd, int64_t rows, int dim, float eps) {
    const int hl = threadIdx.x & 31;
    const int64_t half = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 5;
    const int64_t nhalf = ((int64_t)gridDim.x * blockDim.x) >> 5;
    f32x4 gm[VPL], bt[VPL];
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
        const int c = (j * 32 + hl) * 4;
        gm[j] = c < dim ? *(const f32x4*)(gamma + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        bt[j] = c < dim ? *(const f32x4*)(beta + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const float inv_d = 1.f / dim;
    for (int64_t r = half; r < rows; r += nhalf) {
        const float* xr = x + r * x_ld;
        f32x4 v[VPL];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < VPL; ++j) {
            const int c = (j * 32 + hl) * 4;
            v[j] = c < dim ? *(const f32x4*)(xr + c) : f32x4{0.f, 0.f, 0.f, 0.f};
            s += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
        }
        const float mu = half_sum(s) * inv_d;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < VPL; ++j) {
            const int c = (j * 32 + hl) * 4;
            if (c < dim) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d = v[j][e] - mu;
                    q = fmaf(d, d, q);
                }
            }
        }
        const float rs = rsqrtf(half_sum(q) * inv_d + eps);
        T* yr = y + r * y_ld;
#pragma unroll
        for (int j = 0; j < VPL; ++j) {
            const int c = (j * 32 + hl) * 4;
            if (c < dim) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (v[j][e] - mu) * rs * gm[j][e] + bt[j][e];
                Vec4<T>::store(yr + c, o);
            }
        }
        if (hl == 0) {
            mean[r] = mu;
            rstd[r] = rs;
        }
    }
}

template <typename T>
static void ln_fwd_launch(int vpl, dim3 grid, hipStream_t s, const float* x, int64_t x_ld, const float* gamma, const float* beta,
                          void* y, int64_t y_ld, float* mean, float* rstd, int64_t rows, int dim, float eps) {
#define L(V) hipLaunchKernelGGL((layernorm_fwd_kernel<T, V>), grid, dim3(256), 0, s, x, x_ld, gamma, beta, (T*)y, y_ld, mean, rstd, rows, dim, eps)
    switch (vpl) {
        case 1: L(1); break;
        case 2: L(2); break;
        case 3: L(3); break;
        case 4: L(4); break;
        case 5: L(5); break;
        case 6: L(6); break;
        case 7: L(7); break;
        default: L(8); break;
    }
#undef L
}

static inline bool ln_shape_ok(int64_t dim, int64_t a, int64_t b, int64_t c) {
    return dim % 4 == 0 && dim <= 128 * LN_MAX_VPL && a % 4 == 0 && b % 4 == 0 && c % 4 == 0;
}

extern "C" int vited_layernorm_fwd(const float* x, int64_t x_ld, const float* gamma, const float* beta, void* y,
                                   int y_dtype, int64_t y_ld, float* mean, float* rstd, int64_t rows, int64_t dim,
                                   float eps, void* stream) {
    if (!x || !gamma || !beta || !y || !mean || !rstd || rows <= 0 || dim <= 0 || x_ld < dim || y_ld < dim) return VITED_ERR_BAD_ARG;
    if (!ln_shape_ok(dim, x_ld, y_ld, 0)) return VITED_ERR_UNSUPPORTED;
    if (((uintptr_t)x | (uintptr_t)gamma | (uintptr_t)beta) & 15 || ((uintptr_t)y & 7)) return VITED_ERR_BAD_ARG;
    int64_t blocks = ceil_div64(rows, 8 * 2);
    if (blocks > 2048) blocks = 2048;
    const int vpl = (int)ceil_div64(dim, 128);
    hipStream_t s = (hipStream_t)stream;
    if (y_dtype == VITED_BF16)
        ln_fwd_launch<bf16>(vpl, dim3(blocks), s, x, x_ld, gamma, beta, y, y_ld, mean, rstd, rows, (int)dim, eps);
    else if (y_dtype == VITED_F32)
        ln_fwd_launch<float>(vpl, dim3(blocks), s, x, x_ld, gamma, beta, y, y_ld, mean, rstd, rows, (int)dim, eps);
    else
        return VITED_ERR_UNSUPPORTED;
    return vited_check_launch();
}

// ------------------------------------------------------------------------------------------------
// backward
//   xhat = (x - mean) * rstd ; g = dy * gamma
//   dx = rstd * (g - mean_c(g) - xhat * mean_c(g * xhat))  (+ dx_in)
//   dgamma = sum_rows dy * xhat ; dbeta = sum_rows dy
// ------------------------------------------------------------------------------------------------
template <typename T, int VPL>
__global__ void __launch_bounds__(256)
layernorm_bwd_kernel(const T* __restrict__ dy, int64_t dy_ld, const float* __restrict__ x, int64_t x_ld,
                     const float* __restrict__ gamma, const float* __restrict__ mean, const float* __restrict__ rstd,
                     const float* __restrict__ dx_in, int64_t dx_in_ld, float* __restrict__ dx_out, int64_t dx_out_ld,
                     bf16* __restrict__ dx_lp, int64_t dx_lp_ld, float* __restrict__ partial, int64_t rows, int dim) {
    extern __shared__ __attribute__((aligned(16))) float lds[];  // [8 half-waves][2][dim]
    const int hl = threadIdx.x & 31;
    const int hid = threadIdx.x >> 5;
    const int64_t half = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 5;
    const int64_t nhalf = ((int64_t)gridDim.x * blockDim.x) >> 5;
    f32x4 gm[VPL], dg[VPL], db[VPL];
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
        const int c = (j * 32 + hl) * 4;
        gm[j] = c < dim ? *(const f32x4*)(gamma + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        dg[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        db[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const float inv_d = 1.f / dim;
    for (int64_t r = half; r < rows; r += nhalf) {
        const float mu = mean[r], rs = rstd[r];
        const T* dyr = dy + r * dy_ld;
        const float* xr = x + r * x_ld;
        f32x4 xh[VPL], gg[VPL], res[VPL];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < VPL; ++j) {   // the incoming residual gradient rides along with the first loads, not behind the row sums
            const int c = (j * 32 + hl) * 4;
            res[j] = (dx_in && c < dim) ? *(const f32x4*)(dx_in + r * dx_in_ld + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int j = 0; j < VPL; ++j) {
            const int c = (j * 32 + hl) * 4;
            if (c < dim) {
                const f32x4 d = Vec4<T>::load(dyr + c);
                const f32x4 xv = *(const f32x4*)(xr + c);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    xh[j][e] = (xv[e] - mu) * rs;
                    gg[j][e] = d[e] * gm[j][e];
                    s1 += gg[j][e];
                    s2 = fmaf(gg[j][e], xh[j][e], s2);
                    dg[j][e] = fmaf(d[e], xh[j][e], dg[j][e]);
                    db[j][e] += d[e];
                }
            } else {
                xh[j] = f32x4{0.f, 0.f, 0.f, 0.f};
                gg[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        const float c1 = half_sum(s1) * inv_d, c2 = half_sum(s2) * inv_d;
#pragma unroll
        for (int j = 0; j < VPL; ++j) {
            const int c = (j * 32 + hl) * 4;
            if (c < dim) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = rs * (gg[j][e] - c1 - xh[j][e] * c2) + res[j][e];
                *(f32x4*)(dx_out + r * dx_out_ld + c) = v;
                if (dx_lp) Vec4<bf16>::store(dx_lp + r * dx_lp_ld + c, v);
            }
        }
    }
    // combine the 8 half-waves' column partials through LDS, one partial row pair per workgroup
    float* my = lds + (size_t)hid * 2 * dim;
#pragma unroll
    for (int j = 0; j < VPL; ++j) {
        const int c = (j * 32 + hl) * 4;
        if (c < dim) {
            *(f32x4*)(my + c) = dg[j];
            *(f32x4*)(my + dim + c) = db[j];
        }
    }
    __syncthreads();
    float* out = partial + (size_t)blockIdx.x * 2 * dim;
    for (int c = threadIdx.x; c < 2 * dim; c += blockDim.x) {
        float acc = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) acc += lds[(size_t)w * 2 * dim + c];
        out[c] = acc;
    }
}

// out[c] = sum_p partial[p][c]; block = 16 columns x 16 partial-row groups (a coalesced 64-byte read per
// group and step), LDS combine: 4x more workgroups and 4x shorter loops than a 64-column block
#define LNF_GROUPS 64   // row groups per workgroup (1024 threads = 16 columns x 64 groups): 12 dependent loads per thread at 768 partial rows
__global__ void __launch_bounds__(16 * LNF_GROUPS)
ln_bwd_finish_kernel(const float* __restrict__ partial, int nparts, int width, float* __restrict__ dgamma,
                     float* __restrict__ dbeta, int dim, int accumulate) {
    __shared__ float red[LNF_GROUPS][17];
    const int cx = threadIdx.x & 15, py = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cx;
    float a0 = 0.f, a1 = 0.f;
    if (c < width) {
        int p = py;
        for (; p + LNF_GROUPS < nparts; p += 2 * LNF_GROUPS) {
            a0 += partial[(size_t)p * width + c];
            a1 += partial[(size_t)(p + LNF_GROUPS) * width + c];
        }
        if (p < nparts) a0 += partial[(size_t)p * width + c];
    }
    red[py][cx] = a0 + a1;
    __syncthreads();
    if (py == 0 && c < width) {
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < LNF_GROUPS; ++k) v += red[k][cx];
        float* dst = c < dim ? dgamma + c : dbeta + (c - dim);
        *dst = accumulate ? *dst + v : v;
    }
}

// shared with the fused Linear + LayerNorm backward (gemm_row.hip): sum [nparts][2][dim] partial rows onto dgamma / dbeta
int ln_bwd_finish(const float* partial, int nparts, int dim, float* dgamma, float* dbeta, int accumulate, hipStream_t s) {
    hipLaunchKernelGGL(ln_bwd_finish_kernel, dim3((2 * dim + 15) / 16), dim3(16 * LNF_GROUPS), 0, s, partial, nparts, 2 * dim, dgamma, dbeta, dim,
                       accumulate);
    return vited_check_launch();
}

// Several LayerNorms' column partials finished by ONE launch (the row-complete Linear + LayerNorm backward kernels of one
// Function's backward leave their [tiles][2][dim] partials in caller-owned buffers; functions.py flushes them together).
#define LNF_MAX_BATCH 16
struct LnFinishBatch {
    const float* partial[LNF_MAX_BATCH];
    float* dgamma[LNF_MAX_BATCH];
    float* dbeta[LNF_MAX_BATCH];
    int nparts[LNF_MAX_BATCH];
    int accumulate[LNF_MAX_BATCH];
};

__global__ void __launch_bounds__(16 * LNF_GROUPS)
ln_bwd_finish_batch_kernel(const LnFinishBatch b, int dim) {
    __shared__ float red[LNF_GROUPS][17];
    const int e = blockIdx.y;
    const float* __restrict__ partial = b.partial[e];
    const int nparts = b.nparts[e], width = 2 * dim;
    const int cx = threadIdx.x & 15, py = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cx;
    float a0 = 0.f, a1 = 0.f;
    if (c < width) {
        int p = py;
        for (; p + LNF_GROUPS < nparts; p += 2 * LNF_GROUPS) {
            a0 += partial[(size_t)p * width + c];
            a1 += partial[(size_t)(p + LNF_GROUPS) * width + c];
        }
        if (p < nparts) a0 += partial[(size_t)p * width + c];
    }
    red[py][cx] = a0 + a1;
    __syncthreads();
    if (py == 0 && c < width) {
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < LNF_GROUPS; ++k) v += red[k][cx];
        float* dst = c < dim ? b.dgamma[e] + c : b.dbeta[e] + (c - dim);
        *dst = b.accumulate[e] ? *dst + v : v;
    }
}

extern "C" int vited_layernorm_bwd_finish_batched(int count, const float* const* partial, const int* nparts, float* const* dgamma,
                                                  float* const* dbeta, const int* accumulate, int64_t dim, void* stream) {
    if (count < 1 || !partial || !nparts || !dgamma || !dbeta || !accumulate || dim <= 0) return VITED_ERR_BAD_ARG;
    for (int i0 = 0; i0 < count; i0 += LNF_MAX_BATCH) {
        const int n = count - i0 < LNF_MAX_BATCH ? count - i0 : LNF_MAX_BATCH;
        LnFinishBatch b = {};
        for (int i = 0; i < n; ++i) {
            if (!partial[i0 + i] || !dgamma[i0 + i] || !dbeta[i0 + i] || nparts[i0 + i] < 1) return VITED_ERR_BAD_ARG;
            b.partial[i] = partial[i0 + i]; b.dgamma[i] = dgamma[i0 + i]; b.dbeta[i] = dbeta[i0 + i];
            b.nparts[i] = nparts[i0 + i]; b.accumulate[i] = accumulate[i0 + i];
        }
        hipLaunchKernelGGL(ln_bwd_finish_batch_kernel, dim3((unsigned)((2 * dim + 15) / 16), (unsigned)n), dim3(16 * LNF_GROUPS), 0,
                           (hipStream_t)stream, b, (int)dim);
    }
    return vited_check_launch();
}

static inline int64_t ln_bwd_blocks(int64_t rows) {
    int64_t b = ceil_div64(rows, 8 * 4);  // >= 4 rows per half-wave
    if (b > 768) b = 768;                 // 3 workgroups per CU: measured best of 512 / 768 / 1024 / 1536 / 2048 at 66,560 rows
    if (b < 1) b = 1;
    return b;
}

extern "C" int64_t vited_layernorm_bwd_workspace_bytes(int64_t rows, int64_t dim) {
    return ln_bwd_blocks(rows) * 2 * dim * (int64_t)sizeof(float);
}

template <typename T>
static int ln_bwd_launch(const void* dy, int64_t dy_ld, const float* x, int64_t x_ld, const float* gamma, const float* mean,
                         const float* rstd, const float* dx_in, int64_t dx_in_ld, float* dx_out, int64_t dx_out_ld, void* dx_lp,
                         int64_t dx_lp_ld, float* dgamma, float* dbeta, int accumulate, int64_t rows, int dim, float* ws, hipStream_t s) {
    const int64_t blocks = ln_bwd_blocks(rows);
    const size_t lds = (size_t)8 * 2 * dim * sizeof(float);
    const int vpl = (int)ceil_div64(dim, 128);
#define L(V) hipLaunchKernelGGL((layernorm_bwd_kernel<T, V>), dim3(blocks), dim3(256), lds, s, (const T*)dy, dy_ld, x, x_ld, gamma, mean, \
                                rstd, dx_in, dx_in_ld, dx_out, dx_out_ld, (bf16*)dx_lp, dx_lp_ld, ws, rows, dim)
    switch (vpl) {
        case 1: L(1); break;
        case 2: L(2); break;
        case 3: L(3); break;
        case 4: L(4); break;
        case 5: L(5); break;
        case 6: L(6); break;
        case 7: L(7); break;
        default: L(8); break;
    }
#undef L
    hipLaunchKernelGGL(ln_bwd_finish_kernel, dim3((2 * dim + 15) / 16), dim3(16 * LNF_GROUPS), 0, s, ws, (int)blocks, 2 * dim, dgamma, dbeta, dim, accumulate);
    return vited_check_launch();
}

extern "C" int vited_layernorm_bwd(const void* dy, int dy_dtype, int64_t dy_ld, const float* x, int64_t x_ld,
                                   const float* gamma, const float* mean, const float* rstd, const float* dx_in,
                                   int64_t dx_in_ld, float* dx_out, int64_t dx_out_ld, void* dx_lp, int dx_lp_dtype,
                                   int64_t dx_lp_ld, float* dgamma, float* dbeta, int accumulate, int64_t rows, int64_t dim,
                                   float* workspace, int64_t workspace_bytes, void* stream) {
    if (!dy || !x || !gamma || !mean || !rstd || !dx_out || !dgamma || !dbeta || rows <= 0 || dim <= 0) return VITED_ERR_BAD_ARG;
    if (!ln_shape_ok(dim, x_ld, dy_ld, dx_out_ld) || (dx_in && dx_in_ld % 4) || (dx_lp && dx_lp_ld % 4)) return VITED_ERR_UNSUPPORTED;
    if (((uintptr_t)x | (uintptr_t)gamma | (uintptr_t)dx_out | (uintptr_t)dx_in) & 15) return VITED_ERR_BAD_ARG;
    if (((uintptr_t)dy | (uintptr_t)dx_lp) & 7) return VITED_ERR_BAD_ARG;
    if (dy_dtype == VITED_F32 && ((uintptr_t)dy & 15)) return VITED_ERR_BAD_ARG;
    if (!workspace || workspace_bytes < vited_layernorm_bwd_workspace_bytes(rows, dim)) return VITED_ERR_WORKSPACE;
    if (dx_lp && dx_lp_dtype != VITED_BF16) return VITED_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const int d = (int)dim;
    if (dy_dtype == VITED_BF16)
        return ln_bwd_launch<bf16>(dy, dy_ld, x, x_ld, gamma, mean, rstd, dx_in, dx_in_ld, dx_out, dx_out_ld, dx_lp, dx_lp_ld, dgamma, dbeta, accumulate, rows, d, workspace, s);
    if (dy_dtype == VITED_F32)
        return ln_bwd_launch<float>(dy, dy_ld, x, x_ld, gamma, mean, rstd, dx_in, dx_in_ld, dx_out, dx_out_ld, dx_lp, dx_lp_ld, dgamma, dbeta, accumulate, rows, d, workspace, s);
    return VITED_ERR_UNSUPPORTED;
}
