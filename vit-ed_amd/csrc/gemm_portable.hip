// Portable fp32-FMA GEMM kernels (any dtype in, fp32 accumulate).  They are the whole fp32 compute
// mode (the rtol-1e-3 parity path) and the catch-all for shapes the bf16 MFMA kernels do not tile
// (K % 64 != 0, tiny N such as the 4-logit head, config T's D = 32).
// 64x64x16 LDS tile, 256 threads, 4x4 outputs per thread.
#include "gemm_epilogue.h"
#include "gemm_kernels.h"

#define PT_BM 64
#define PT_BN 64
#define PT_BK 16
#define PT_LD 68  // padded leading dim (floats): 16-B aligned rows, conflict-free b128 reads

// A [M, K] row-major (lda).  B: B_KN ? [K, N] (ldb) : [N, K] (ldb).
template <typename T, int EPI, bool B_KN>
__global__ void __launch_bounds__(256)
gemm_portable_kernel(const T* __restrict__ A, int64_t lda, const T* __restrict__ B, int64_t ldb, int64_t M, int64_t N,
                     int64_t K, EpiParams ep) {
    __shared__ __attribute__((aligned(16))) float As[PT_BK][PT_LD];
    __shared__ __attribute__((aligned(16))) float Bs[PT_BK][PT_LD];
    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;
    const int64_t m0 = (int64_t)blockIdx.y * PT_BM, n0 = (int64_t)blockIdx.x * PT_BN;
    float acc[4][4] = {};
    for (int64_t k0 = 0; k0 < K; k0 += PT_BK) {
        {   // A tile: thread -> row tid/4, k (tid%4)*4..+3
            const int r = tid >> 2, kq = (tid & 3) * 4;
            const int64_t m = m0 + r;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int64_t k = k0 + kq + j;
                As[kq + j][r] = (m < M && k < K) ? to_f32(A[m * lda + k]) : 0.f;
            }
        }
        if constexpr (!B_KN) {
            const int r = tid >> 2, kq = (tid & 3) * 4;
            const int64_t n = n0 + r;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int64_t k = k0 + kq + j;
                Bs[kq + j][r] = (n < N && k < K) ? to_f32(B[n * ldb + k]) : 0.f;
            }
        } else {
            const int kr = tid >> 4, nq = (tid & 15) * 4;
            const int64_t k = k0 + kr;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int64_t n = n0 + nq + j;
                Bs[kr][nq + j] = (n < N && k < K) ? to_f32(B[k * ldb + n]) : 0.f;
            }
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < PT_BK; ++kk) {
            const f32x4 a = *(const f32x4*)&As[kk][ty * 4];
            const f32x4 b = *(const f32x4*)&Bs[kk][tx * 4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t m = m0 + ty * 4 + i;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t n = n0 + tx * 4 + j;
            if (n < N) epilogue_store<T, EPI>(ep, m, n, acc[i][j]);
        }
    }
}

template <typename T, int EPI>
static void launch_portable(const void* A, int64_t lda, const void* B, int64_t ldb, int b_layout, int64_t M, int64_t N,
                            int64_t K, const EpiParams& ep, hipStream_t s) {
    dim3 grid((unsigned)ceil_div64(N, PT_BN), (unsigned)ceil_div64(M, PT_BM));
    if (b_layout == VITED_B_KN)
        hipLaunchKernelGGL((gemm_portable_kernel<T, EPI, true>), grid, dim3(256), 0, s, (const T*)A, lda, (const T*)B, ldb, M, N, K, ep);
    else
        hipLaunchKernelGGL((gemm_portable_kernel<T, EPI, false>), grid, dim3(256), 0, s, (const T*)A, lda, (const T*)B, ldb, M, N, K, ep);
}

template <typename T>
static int dispatch_portable(const void* A, int64_t lda, const void* B, int64_t ldb, int b_layout, int64_t M, int64_t N,
                             int64_t K, int epilogue, const EpiParams& ep, hipStream_t s) {
    switch (epilogue) {
        case VITED_EPI_STORE: launch_portable<T, VITED_EPI_STORE>(A, lda, B, ldb, b_layout, M, N, K, ep, s); break;
        case VITED_EPI_GELU: launch_portable<T, VITED_EPI_GELU>(A, lda, B, ldb, b_layout, M, N, K, ep, s); break;
        case VITED_EPI_RESIDUAL: launch_portable<T, VITED_EPI_RESIDUAL>(A, lda, B, ldb, b_layout, M, N, K, ep, s); break;
        case VITED_EPI_MUL_GELU_GRAD: launch_portable<T, VITED_EPI_MUL_GELU_GRAD>(A, lda, B, ldb, b_layout, M, N, K, ep, s); break;
        case VITED_EPI_STORE_F32: launch_portable<T, VITED_EPI_STORE_F32>(A, lda, B, ldb, b_layout, M, N, K, ep, s); break;
        case VITED_EPI_MUL: launch_portable<T, VITED_EPI_MUL>(A, lda, B, ldb, b_layout, M, N, K, ep, s); break;
        case VITED_EPI_GELU_GRAD: launch_portable<T, VITED_EPI_GELU_GRAD>(A, lda, B, ldb, b_layout, M, N, K, ep, s); break;
        default: return VITED_ERR_BAD_ARG;
    }
    return vited_check_launch();
}

int gemm_portable(const void* A, int64_t lda, const void* B, int64_t ldb, int b_layout, int dtype, int64_t M, int64_t N,
                  int64_t K, int epilogue, const EpiParams& ep, hipStream_t s) {
    if (dtype == VITED_F32) return dispatch_portable<float>(A, lda, B, ldb, b_layout, M, N, K, epilogue, ep, s);
    if (dtype == VITED_BF16) return dispatch_portable<bf16>(A, lda, B, ldb, b_layout, M, N, K, epilogue, ep, s);
    return VITED_ERR_UNSUPPORTED;
}

// ------------------------------------------------------------------------------------------------
// weight gradient: dW[n, k] = sum_m dY[m, n] X[m, k].  Split over M: block z handles rows
// [z*rows_per_split, ...), writes its 64x64 fp32 tile to slab z (or straight to dW when S == 1).
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256)
gemm_tn_portable_kernel(const T* __restrict__ dY, int64_t lddy, const T* __restrict__ X, int64_t ldx, int64_t M,
                        int64_t N, int64_t K, int64_t rows_per_split, float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) float As[PT_BK][PT_LD];
    __shared__ __attribute__((aligned(16))) float Bs[PT_BK][PT_LD];
    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;
    const int64_t n0 = (int64_t)blockIdx.y * 64, kc0 = (int64_t)blockIdx.x * 64;
    const int64_t mb = (int64_t)blockIdx.z * rows_per_split;
    int64_t me = mb + rows_per_split;
    if (me > M) me = M;
    float acc[4][4] = {};
    const int mr = tid >> 4, cq = (tid & 15) * 4;
    for (int64_t m0 = mb; m0 < me; m0 += PT_BK) {
        const int64_t m = m0 + mr;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t n = n0 + cq + j, k = kc0 + cq + j;
            As[mr][cq + j] = (m < me && n < N) ? to_f32(dY[m * lddy + n]) : 0.f;
            Bs[mr][cq + j] = (m < me && k < K) ? to_f32(X[m * ldx + k]) : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < PT_BK; ++kk) {
            const f32x4 a = *(const f32x4*)&As[kk][ty * 4];
            const f32x4 b = *(const f32x4*)&Bs[kk][tx * 4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }
    float* o = out + (int64_t)blockIdx.z * N * K;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t n = n0 + ty * 4 + i;
        if (n >= N) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t k = kc0 + tx * 4 + j;
            if (k < K) o[n * K + k] = acc[i][j];
        }
    }
}

int gemm_tn_portable(const void* dY, int64_t lddy, const void* X, int64_t ldx, int dtype, int64_t M, int64_t N, int64_t K,
                     int64_t splits, float* out, hipStream_t s) {
    dim3 grid((unsigned)ceil_div64(K, 64), (unsigned)ceil_div64(N, 64), (unsigned)splits);
    const int64_t rps = ceil_div64(ceil_div64(M, splits), PT_BK) * PT_BK;
    if (dtype == VITED_F32)
        hipLaunchKernelGGL((gemm_tn_portable_kernel<float>), grid, dim3(256), 0, s, (const float*)dY, lddy, (const float*)X, ldx, M, N, K, rps, out);
    else if (dtype == VITED_BF16)
        hipLaunchKernelGGL((gemm_tn_portable_kernel<bf16>), grid, dim3(256), 0, s, (const bf16*)dY, lddy, (const bf16*)X, ldx, M, N, K, rps, out);
    else
        return VITED_ERR_UNSUPPORTED;
    return vited_check_launch();
}
