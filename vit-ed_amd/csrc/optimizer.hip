// Fused gradient clip + AdamW + bf16 weight-shadow refresh + gradient zeroing on the flat gradient buffer
// (replaces clip_grad_norm_ + torch.optim.AdamW.step + vited_cast_weights + zero_grad:
// misc/utils.py:215-223, misc/optimizer.py:25-46, misc/engine.py:231).
//
//   launch 1  adamw_sumsq_kernel : partial[b] = sum of squares of a slice of the flat gradient (fixed slices and a
//                                  fixed reduction order: deterministic); block 0 also advances the step counter.
//   launch 2  adamw_update_kernel: one workgroup per 64 x 64 tile of one parameter (device descriptor table, as
//                                  vited_cast_weights).  Every workgroup re-reduces the partials in the same order, so
//                                  all of them see the same norm and clip coefficient; then per element
//                                      g' = g * min(1, max_norm / (norm + 1e-6))
//                                      p  = p * (1 - lr * wd);  m = lerp(m, g', 1 - b1);  v = b2 * v + (1 - b2) * g'^2
//                                      p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
//                                  (torch.optim.AdamW's update), the new p is also written to its bf16 [rows, cols] shadow
//                                  and, through an LDS tile, to the transposed bf16 [cols, rows] shadow, and g is zeroed.
// HBM traffic per element: 16 B read (g, p, m, v) + 16 B written (g, p, m, v) + 4 B of shadows = the floor for this update.
// All hyper-parameters that change between steps (learning rate per group, step count) live in a device array, so a
// hipGraph replay of the two launches follows the scheduler.
#include "common.h"

#define AD_DESC_WORDS 10   // {p, g, m, v, shadow, shadow_t, rows, cols, first_tile, group}
#define AD_HYPER_HEADER 8  // hyper[0] = step count (float), [1..7] reserved; then 8 floats per group
#define AD_GROUP_WORDS 8   // {lr, beta1, beta2, eps, weight_decay, -, -, -}
#define AD_PARTIALS 1024

__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) red[wave] = v;
    __syncthreads();
    const float total = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    return total;
}

__global__ void __launch_bounds__(256)
adamw_sumsq_kernel(const float* __restrict__ g, int64_t n, float* __restrict__ partials, float* __restrict__ hyper) {
    __shared__ float red[4];
    // fixed contiguous slice per block, so the sum does not depend on the launch
    const int64_t per = ((n + AD_PARTIALS - 1) / AD_PARTIALS + 3) & ~(int64_t)3;
    const int64_t lo = (int64_t)blockIdx.x * per;
    int64_t hi = lo + per;
    hi = hi < n ? hi : n;
    float s = 0.f;
    const bool al = ((uintptr_t)g & 15) == 0;
    if (al) {
        for (int64_t i = lo + (int64_t)threadIdx.x * 4; i < hi; i += 1024) {
            if (i + 3 < hi) {
                const f32x4 q = *(const f32x4*)(g + i);
                s += q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
            } else {
                for (int64_t j = i; j < hi; ++j) s += g[j] * g[j];
            }
        }
    } else {
        for (int64_t i = lo + threadIdx.x; i < hi; i += 256) s += g[i] * g[i];
    }
    s = block_sum_256(s, red);
    if (threadIdx.x == 0) {
        partials[blockIdx.x] = s;
        if (blockIdx.x == 0) hyper[0] += 1.0f;   // step count t of this update (read by launch 2, which runs after this kernel)
    }
}

__global__ void __launch_bounds__(256)
adamw_update_kernel(const int64_t* __restrict__ desc, int count, const float* __restrict__ partials,
                    const float* __restrict__ hyper, float max_norm, int zero_grad, float* __restrict__ norm_out) {
    __shared__ bf16 tile[64][66];
    __shared__ float red[4];
    // ---- the same norm in every workgroup (same partials, same order)
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < AD_PARTIALS / 256; ++i) s += partials[threadIdx.x + 256 * i];
    const float norm = sqrtf(block_sum_256(s, red));
    float clip = 1.0f;
    if (max_norm > 0.f) {
        clip = max_norm / (norm + 1e-6f);
        clip = clip < 1.0f ? clip : 1.0f;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && norm_out) *norm_out = norm;

    const int64_t blk = blockIdx.x;
    int lo = 0, hi = count - 1;     // last descriptor whose first_tile <= blk
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (desc[(int64_t)mid * AD_DESC_WORDS + 8] <= blk) lo = mid; else hi = mid - 1;
    }
    const int64_t* d = desc + (int64_t)lo * AD_DESC_WORDS;
    float* __restrict__ p = (float*)d[0];
    float* __restrict__ g = (float*)d[1];
    float* __restrict__ m = (float*)d[2];
    float* __restrict__ v = (float*)d[3];
    bf16* __restrict__ dst = (bf16*)d[4];
    bf16* __restrict__ dst_t = (bf16*)d[5];
    const int64_t rows = d[6], cols = d[7];
    const float* hg = hyper + AD_HYPER_HEADER + d[9] * AD_GROUP_WORDS;
    const float step = hyper[0];
    const float lr = hg[0], b1 = hg[1], b2 = hg[2], eps = hg[3], wd = hg[4];
    const float bc1 = 1.0f - powf(b1, step), bc2 = 1.0f - powf(b2, step);
    const float step_size = lr / bc1, inv_sqrt_bc2 = 1.0f / sqrtf(bc2), decay = 1.0f - lr * wd;

    const int64_t t = blk - d[8], tiles_c = (cols + 63) >> 6;
    const int64_t r0 = (t / tiles_c) << 6, c0 = (t % tiles_c) << 6;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;   // 16 x 16: 4 columns x 4 rows per thread
    const bool vec = (cols & 3) == 0 && (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int lr_ = ty + 16 * j;
        const int64_t r = r0 + lr_, c = c0 + tx * 4;
        float pn[4] = {0.f, 0.f, 0.f, 0.f};
        if (r < rows && c < cols) {
            const int64_t o = r * cols + c;
            const bool full = vec && c + 3 < cols;
            float gv[4], pv[4], mv[4], vv[4];
            if (full) {
                const f32x4 qg = *(const f32x4*)(g + o), qp = *(const f32x4*)(p + o), qm = *(const f32x4*)(m + o), qv = *(const f32x4*)(v + o);
#pragma unroll
                for (int e = 0; e < 4; ++e) { gv[e] = qg[e]; pv[e] = qp[e]; mv[e] = qm[e]; vv[e] = qv[e]; }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bool in = c + e < cols;
                    gv[e] = in ? g[o + e] : 0.f; pv[e] = in ? p[o + e] : 0.f; mv[e] = in ? m[o + e] : 0.f; vv[e] = in ? v[o + e] : 0.f;
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float ge = gv[e] * clip;
                const float pe = pv[e] * decay;
                mv[e] = mv[e] + (ge - mv[e]) * (1.0f - b1);
                vv[e] = vv[e] * b2 + (1.0f - b2) * ge * ge;
                const float denom = sqrtf(vv[e]) * inv_sqrt_bc2 + eps;
                pn[e] = pe - step_size * (mv[e] / denom);
            }
            if (full) {
                *(f32x4*)(p + o) = f32x4{pn[0], pn[1], pn[2], pn[3]};
                *(f32x4*)(m + o) = f32x4{mv[0], mv[1], mv[2], mv[3]};
                *(f32x4*)(v + o) = f32x4{vv[0], vv[1], vv[2], vv[3]};
                if (zero_grad) *(f32x4*)(g + o) = f32x4{0.f, 0.f, 0.f, 0.f};
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (c + e < cols) { p[o + e] = pn[e]; m[o + e] = mv[e]; v[o + e] = vv[e]; if (zero_grad) g[o + e] = 0.f; }
            }
        }
        if (!dst && !dst_t) continue;
        bf16 b[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { b[e] = (bf16)pn[e]; tile[lr_][tx * 4 + e] = b[e]; }
        if (dst && r < rows) {
            if ((cols & 3) == 0 && c + 3 < cols) *(bf16x4*)(dst + r * cols + c) = bf16x4{b[0], b[1], b[2], b[3]};
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (c + e < cols) dst[r * cols + c + e] = b[e];
            }
        }
    }
    if (!dst_t) return;
    __syncthreads();
    const bool vec_t = (rows & 3) == 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int lc = ty + 16 * j;                 // column of the source tile = row of the transposed shadow
        const int64_t c = c0 + lc, r = r0 + tx * 4;
        if (c >= cols) continue;
        bf16 b[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) b[e] = tile[tx * 4 + e][lc];
        if (vec_t && r + 3 < rows) *(bf16x4*)(dst_t + c * rows + r) = bf16x4{b[0], b[1], b[2], b[3]};
        else {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (r + e < rows) dst_t[c * rows + r + e] = b[e];
        }
    }
}

extern "C" int64_t vited_adamw_workspace_bytes(void) { return (int64_t)AD_PARTIALS * sizeof(float); }

extern "C" int vited_adamw_step(const int64_t* desc, int count, int64_t total_tiles, const float* grad_flat, int64_t grad_numel,
                                float* hyper, float max_norm, int zero_grad, float* norm_out, float* workspace,
                                int64_t workspace_bytes, void* stream) {
    if (!desc || !grad_flat || !hyper || !workspace || count <= 0 || grad_numel <= 0 || total_tiles <= 0 || total_tiles > 0x7fffffff)
        return VITED_ERR_BAD_ARG;
    if (workspace_bytes < vited_adamw_workspace_bytes()) return VITED_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(adamw_sumsq_kernel, dim3(AD_PARTIALS), dim3(256), 0, s, grad_flat, grad_numel, workspace, hyper);
    hipLaunchKernelGGL(adamw_update_kernel, dim3((unsigned)total_tiles), dim3(256), 0, s, desc, count, workspace, hyper, max_norm,
                       zero_grad, norm_out);
    return vited_check_launch();
}
