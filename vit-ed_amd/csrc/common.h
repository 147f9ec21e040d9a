// Shared device helpers for the ViT-ED gfx950 kernels (CDNA4: wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vited.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define VITED_WAVE 64

template <typename T> struct dtype_of;
template <> struct dtype_of<float> { static constexpr int value = VITED_F32; };
template <> struct dtype_of<bf16> { static constexpr int value = VITED_BF16; };

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16 from_f32<bf16>(float v) { return (bf16)v; }

// full-wave (64 lane) butterfly reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

// exact (erf) GELU and its derivative: nn.GELU() default, timm Mlp act
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
    const float pdf = 0.39894228040143268f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}

// Cheap erf-GELU for the bf16 MFMA epilogues (the libm erff above costs more VALU time than the
// K=384 GEMM it follows): Abramowitz-Stegun 7.1.26, |erf error| <= 1.5e-7, i.e. far below one bf16
// ulp (measured over [-12, 12]: |cdf error| <= 2.7e-7); one v_exp_f32 + one v_rcp_f32 + 11 plain VALU operations, and gelu' reuses the
// same exponential because exp(-(x/sqrt2)^2) is also the Gaussian density's exponent.
__device__ __forceinline__ void gelu_parts_fast(float x, float& cdf, float& e) {
    // 64 activations per lane go through this in the fc1 epilogue, so the form below is the one with the fewest instructions
    // (9 % fewer than the textbook arrangement bought 1.1 % of that kernel: vector issue is not what bounds it): v = |x| sqrt(log2(e)/2) serves both the exponent (e = 2^(-v^2) = exp(-x^2/2), one multiply
    // with a negated operand) and the rational argument (t = 1 / (1 + p v) with A&S's p rescaled); 0.5 is folded into the polynomial;
    // the reflection cdf(x >= 0) = 1 - h, cdf(x < 0) = h is  step(x) - copysign(h, x)  with step from one clamped FMA - no compare /
    // select through VCC, and the negative tail keeps h's relative accuracy (no 0.5 - (0.5 - h) cancellation).
    const float v = fabsf(x) * 0.84932180028801907f;                       // |x| * sqrt(0.5 * log2(e))
    e = __builtin_amdgcn_exp2f(-v * v);                                     // exp(-x^2/2)
    const float t = __builtin_amdgcn_rcpf(fmaf(0.27273748087922250f, v, 1.0f));   // 0.3275911 / sqrt(log2(e)): same t as A&S 7.1.26
    float p = fmaf(t, 0.5307027145f, -0.7265760135f);                       // A&S coefficients x 0.5
    p = fmaf(t, p, 0.7107068705f);
    p = fmaf(t, p, -0.142248368f);
    p = fmaf(t, p, 0.127414796f);
    const float h = (p * t) * e;                                            // 0.5 * erfc(|x| / sqrt2)
    const float step = __builtin_amdgcn_fmed3f(fmaf(x, 3.0e38f, 1.0f), 0.f, 1.f);   // 1 for x >= 0, 0 for x < 0 (clamp modifier)
    cdf = step - __builtin_copysignf(h, x);
}
__device__ __forceinline__ float gelu_fast(float x) {
    float cdf, e;
    gelu_parts_fast(x, cdf, e);
    return x * cdf;
}
__device__ __forceinline__ float gelu_grad_fast(float x) {
    float cdf, e;
    gelu_parts_fast(x, cdf, e);
    return fmaf(x * 0.39894228040143268f, e, cdf);
}

// Sum over the 32 lanes of a half-wave (the LayerNorm kernels' row owner); every lane gets the sum.  Four DPP row rotations
// (vector instructions, no LDS) sum the two 16-lane rows, ONE ds_swizzle adds the other row: the 5-step __shfl_xor butterfly it
// replaces compiles to five DEPENDENT ds_bpermute_b32 (an LDS round trip each), and a LayerNorm row needs two of them in series.
template <int N> __device__ __forceinline__ float dpp_row_ror(float v) {   // lane i of a 16-lane row takes lane (i + N) % 16's value
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + N, 0xf, 0xf, true));
}
__device__ __forceinline__ float half_wave_sum(float v) {
    v += dpp_row_ror<8>(v);
    v += dpp_row_ror<4>(v);
    v += dpp_row_ror<2>(v);
    v += dpp_row_ror<1>(v);
    // bit-mask swizzle: and 0x1f, or 0, xor 0x10 -> lane i reads lane i ^ 16 of its 32-lane group
    return v + __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401F));
}

static inline int vited_check_launch() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? VITED_OK : VITED_ERR_LAUNCH;
}

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }
