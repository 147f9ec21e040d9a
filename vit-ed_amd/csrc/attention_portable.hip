// Portable (fp32 VALU) attention core: one thread per query row (forward, dQ) or per key row
// (dK/dV), K/V or Q/dO tiles broadcast from LDS, flash-style streaming softmax so nothing of size
// Nq x Nk is ever materialised.  This is the whole fp32 compute mode and the catch-all for head dims
// the MFMA attention kernels do not cover.  Math is fp32 whatever the storage type T.
#include "attention_kernels.h"

#define AP_THREADS 128
#define AP_KT 32  // keys (or queries) per LDS tile

template <typename T, int HD>
__device__ __forceinline__ void load_row(const T* p, float* r, float mul) {
#pragma unroll
    for (int d = 0; d < HD; ++d) r[d] = to_f32(p[d]) * mul;
}

// cooperative tile load: rows [j0, j0+AP_KT) of a strided [n, HD] head slice -> LDS fp32, zero padded
template <typename T, int HD>
__device__ __forceinline__ void load_tile(const T* base, int64_t ts, int64_t j0, int64_t n, float (*dst)[HD]) {
    for (int e = threadIdx.x; e < AP_KT * HD; e += AP_THREADS) {
        const int jj = e / HD, d = e - jj * HD;
        const int64_t j = j0 + jj;
        dst[jj][d] = j < n ? to_f32(base[j * ts + d]) : 0.f;
    }
}

template <typename T, int HD>
__global__ void __launch_bounds__(AP_THREADS)
attn_fwd_portable_kernel(AttnArgs a) {
    __shared__ __attribute__((aligned(16))) float Ks[AP_KT][HD];
    __shared__ __attribute__((aligned(16))) float Vs[AP_KT][HD];
    const int64_t b = blockIdx.z;
    const int h = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * AP_THREADS + threadIdx.x;
    const bool live = i < a.nq;
    const int64_t bkv = a.kv_index ? a.kv_index[b] : b;
    const T* kb = (const T*)a.k + bkv * a.k_bs + (int64_t)h * HD;
    const T* vb = (const T*)a.v + bkv * a.v_bs + (int64_t)h * HD;
    float qr[HD], acc[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) acc[d] = 0.f;
    if (live) load_row<T, HD>((const T*)a.q + b * a.q_bs + i * a.q_ts + (int64_t)h * HD, qr, a.scale);
    else {
#pragma unroll
        for (int d = 0; d < HD; ++d) qr[d] = 0.f;
    }
    float m = -INFINITY, l = 0.f;
    for (int64_t j0 = 0; j0 < a.nk; j0 += AP_KT) {
        load_tile<T, HD>(kb, a.k_ts, j0, a.nk, Ks);
        load_tile<T, HD>(vb, a.v_ts, j0, a.nk, Vs);
        __syncthreads();
        const int cnt = (int)((a.nk - j0) < AP_KT ? (a.nk - j0) : AP_KT);
        float s[AP_KT];
        float tmax = -INFINITY;
#pragma unroll
        for (int jj = 0; jj < AP_KT; ++jj) {
            float d0 = 0.f, d1 = 0.f;
#pragma unroll
            for (int d = 0; d < HD; d += 2) {
                d0 = fmaf(qr[d], Ks[jj][d], d0);
                d1 = fmaf(qr[d + 1], Ks[jj][d + 1], d1);
            }
            s[jj] = jj < cnt ? d0 + d1 : -INFINITY;
            tmax = fmaxf(tmax, s[jj]);
        }
        const float mn = fmaxf(m, tmax);
        const float alpha = __expf(m - mn);  // m = -inf on the first tile -> 0
        l *= alpha;
#pragma unroll
        for (int d = 0; d < HD; ++d) acc[d] *= alpha;
#pragma unroll
        for (int jj = 0; jj < AP_KT; ++jj) {
            const float p = __expf(s[jj] - mn);  // masked keys: exp(-inf) = 0
            l += p;
#pragma unroll
            for (int d = 0; d < HD; ++d) acc[d] = fmaf(p, Vs[jj][d], acc[d]);
        }
        m = mn;
        __syncthreads();
    }
    if (live) {
        const float inv = 1.f / l;
        T* o = (T*)a.o + b * a.o_bs + i * a.o_ts + (int64_t)h * HD;
#pragma unroll
        for (int d = 0; d < HD; ++d) o[d] = from_f32<T>(acc[d] * inv);
        a.lse[(b * a.heads + h) * a.nq + i] = m + __logf(l);
    }
}

// dQ (+ delta = rowsum(dO * O)), one thread per query row
template <typename T, int HD>
__global__ void __launch_bounds__(AP_THREADS)
attn_bwd_dq_portable_kernel(AttnArgs a) {
    __shared__ __attribute__((aligned(16))) float Ks[AP_KT][HD];
    __shared__ __attribute__((aligned(16))) float Vs[AP_KT][HD];
    const int64_t b = blockIdx.z;
    const int h = blockIdx.y;
    const int64_t i = (int64_t)blockIdx.x * AP_THREADS + threadIdx.x;
    const bool live = i < a.nq;
    const T* kb = (const T*)a.k + b * a.k_bs + (int64_t)h * HD;
    const T* vb = (const T*)a.v + b * a.v_bs + (int64_t)h * HD;
    float qr[HD], dor[HD], dq[HD];
    float delta = 0.f, lse = 0.f;
#pragma unroll
    for (int d = 0; d < HD; ++d) dq[d] = 0.f;
    if (live) {
        load_row<T, HD>((const T*)a.q + b * a.q_bs + i * a.q_ts + (int64_t)h * HD, qr, a.scale);
        load_row<T, HD>((const T*)a.d_o + b * a.o_bs + i * a.o_ts + (int64_t)h * HD, dor, 1.f);
        const T* orow = (const T*)a.o + b * a.o_bs + i * a.o_ts + (int64_t)h * HD;
#pragma unroll
        for (int d = 0; d < HD; ++d) delta = fmaf(dor[d], to_f32(orow[d]), delta);
        lse = a.lse[(b * a.heads + h) * a.nq + i];
        a.delta[(b * a.heads + h) * a.nq + i] = delta;
    } else {
#pragma unroll
        for (int d = 0; d < HD; ++d) { qr[d] = 0.f; dor[d] = 0.f; }
    }
    for (int64_t j0 = 0; j0 < a.nk; j0 += AP_KT) {
        load_tile<T, HD>(kb, a.k_ts, j0, a.nk, Ks);
        load_tile<T, HD>(vb, a.v_ts, j0, a.nk, Vs);
        __syncthreads();
        const int cnt = (int)((a.nk - j0) < AP_KT ? (a.nk - j0) : AP_KT);
        for (int jj = 0; jj < cnt; ++jj) {
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int d = 0; d < HD; ++d) {
                s = fmaf(qr[d], Ks[jj][d], s);
                dp = fmaf(dor[d], Vs[jj][d], dp);
            }
            const float p = __expf(s - lse);
            const float ds = p * (dp - delta);
#pragma unroll
            for (int d = 0; d < HD; ++d) dq[d] = fmaf(ds, Ks[jj][d], dq[d]);
        }
        __syncthreads();
    }
    if (live) {
        T* o = (T*)a.dq + b * a.dq_bs + i * a.dq_ts + (int64_t)h * HD;
#pragma unroll
        for (int d = 0; d < HD; ++d) o[d] = from_f32<T>(dq[d] * a.scale);
    }
}

// dK / dV, one thread per key row; Q/dO tiles (+ lse, delta) broadcast from LDS
template <typename T, int HD>
__global__ void __launch_bounds__(AP_THREADS)
attn_bwd_dkv_portable_kernel(AttnArgs a) {
    __shared__ __attribute__((aligned(16))) float Qs[AP_KT][HD];
    __shared__ __attribute__((aligned(16))) float Ds[AP_KT][HD];
    __shared__ float Ls[AP_KT], Dl[AP_KT];
    const int64_t b = blockIdx.z;
    const int h = blockIdx.y;
    const int64_t j = (int64_t)blockIdx.x * AP_THREADS + threadIdx.x;
    const bool live = j < a.nk;
    const T* qb = (const T*)a.q + b * a.q_bs + (int64_t)h * HD;
    const T* dob = (const T*)a.d_o + b * a.o_bs + (int64_t)h * HD;
    float kr[HD], vr[HD], dk[HD], dv[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) { dk[d] = 0.f; dv[d] = 0.f; }
    if (live) {
        load_row<T, HD>((const T*)a.k + b * a.k_bs + j * a.k_ts + (int64_t)h * HD, kr, a.scale);
        load_row<T, HD>((const T*)a.v + b * a.v_bs + j * a.v_ts + (int64_t)h * HD, vr, 1.f);
    } else {
#pragma unroll
        for (int d = 0; d < HD; ++d) { kr[d] = 0.f; vr[d] = 0.f; }
    }
    for (int64_t i0 = 0; i0 < a.nq; i0 += AP_KT) {
        load_tile<T, HD>(qb, a.q_ts, i0, a.nq, Qs);
        load_tile<T, HD>(dob, a.o_ts, i0, a.nq, Ds);
        if (threadIdx.x < AP_KT) {
            const int64_t i = i0 + threadIdx.x;
            Ls[threadIdx.x] = i < a.nq ? a.lse[(b * a.heads + h) * a.nq + i] : 0.f;
            Dl[threadIdx.x] = i < a.nq ? a.delta[(b * a.heads + h) * a.nq + i] : 0.f;
        }
        __syncthreads();
        const int cnt = (int)((a.nq - i0) < AP_KT ? (a.nq - i0) : AP_KT);
        for (int ii = 0; ii < cnt; ++ii) {
            float s = 0.f, dp = 0.f;
#pragma unroll
            for (int d = 0; d < HD; ++d) {
                s = fmaf(Qs[ii][d], kr[d], s);  // kr carries the softmax scale
                dp = fmaf(Ds[ii][d], vr[d], dp);
            }
            const float p = __expf(s - Ls[ii]);
            const float ds = p * (dp - Dl[ii]);
#pragma unroll
            for (int d = 0; d < HD; ++d) {
                dv[d] = fmaf(p, Ds[ii][d], dv[d]);
                dk[d] = fmaf(ds, Qs[ii][d], dk[d]);
            }
        }
        __syncthreads();
    }
    if (live) {
        T* ok = (T*)a.dk + b * a.dk_bs + j * a.dk_ts + (int64_t)h * HD;
        T* ov = (T*)a.dv + b * a.dv_bs + j * a.dv_ts + (int64_t)h * HD;
#pragma unroll
        for (int d = 0; d < HD; ++d) {
            ok[d] = from_f32<T>(dk[d] * a.scale);
            ov[d] = from_f32<T>(dv[d]);
        }
    }
}

template <typename T, int HD>
static int launch_fwd(const AttnArgs& a, hipStream_t s) {
    dim3 grid((unsigned)ceil_div64(a.nq, AP_THREADS), a.heads, (unsigned)a.batch);
    hipLaunchKernelGGL((attn_fwd_portable_kernel<T, HD>), grid, dim3(AP_THREADS), 0, s, a);
    return vited_check_launch();
}

template <typename T, int HD>
static int launch_bwd(const AttnArgs& a, hipStream_t s) {
    dim3 gq((unsigned)ceil_div64(a.nq, AP_THREADS), a.heads, (unsigned)a.batch);
    dim3 gk((unsigned)ceil_div64(a.nk, AP_THREADS), a.heads, (unsigned)a.batch);
    hipLaunchKernelGGL((attn_bwd_dq_portable_kernel<T, HD>), gq, dim3(AP_THREADS), 0, s, a);
    hipLaunchKernelGGL((attn_bwd_dkv_portable_kernel<T, HD>), gk, dim3(AP_THREADS), 0, s, a);
    return vited_check_launch();
}

int attention_fwd_portable(const AttnArgs& a, int dtype, hipStream_t s) {
    if (dtype == VITED_F32) {
        if (a.head_dim == 32) return launch_fwd<float, 32>(a, s);
        if (a.head_dim == 64) return launch_fwd<float, 64>(a, s);
    } else if (dtype == VITED_BF16) {
        if (a.head_dim == 32) return launch_fwd<bf16, 32>(a, s);
        if (a.head_dim == 64) return launch_fwd<bf16, 64>(a, s);
    }
    return VITED_ERR_UNSUPPORTED;
}

int attention_bwd_portable(const AttnArgs& a, int dtype, hipStream_t s) {
    if (dtype == VITED_F32) {
        if (a.head_dim == 32) return launch_bwd<float, 32>(a, s);
        if (a.head_dim == 64) return launch_bwd<float, 64>(a, s);
    } else if (dtype == VITED_BF16) {
        if (a.head_dim == 32) return launch_bwd<bf16, 32>(a, s);
        if (a.head_dim == 64) return launch_bwd<bf16, 64>(a, s);
    }
    return VITED_ERR_UNSUPPORTED;
}
