// Epilogue of the bf16 MFMA NT GEMM.  The 64 x 64 accumulator tile of a wave goes through a
// per-wave LDS scratch 16 rows at a time and comes back in a store-friendly ownership:
//   bf16 outputs : lane -> 2 x (row, 8 consecutive columns): 8 lanes x 16 B = one 128-byte row
//                  segment, 8 rows per store instruction;
//   fp32 outputs : lane -> 4 x (row, 4 consecutive columns): 16 lanes x 16 B = one 256-byte row
//                  segment, 4 rows per store instruction.
// Every global access of the epilogue (bias, residual, saved pre-activation, outputs) is therefore a
// full-cache-line, 16-byte-per-lane access.  Operands that do not depend on the accumulators are
// fetched into registers one sub-tile ahead of their use (EpiPrefetch), after the main loop.
#pragma once
#include "gemm_epilogue.h"

// Every output of these epilogues is written once and read by a LATER kernel (saved activations, the residual stream, gradients):
// stores carry the streaming (non-temporal) hint, so they do not displace the L2 lines the co-resident workgroups re-read
// (the weight panel, the activation panel shared by the n-tiles of a row block).  Measured: fc1 + GELU' 177 -> 139 us and
// dz 172 -> 141 us isolated (M = 65,536), step 27.2 -> 26.8 ms; with the fp32 epilogues and the row-complete kernels' stores
// as well 26.55 ms.  (The same hint on the LayerNorm kernels' stores costs 0.3 ms, on the attention outputs nothing: not used there.)
#define EPI_STORE16(ptr, val) __builtin_nontemporal_store(val, ptr)
#define EPI_STORE16F(ptr, val) __builtin_nontemporal_store(val, ptr)
#define EPI_LOAD16(ptr) __builtin_nontemporal_load(ptr)     // residual rows / saved activations: read exactly once

#define SCRATCH_LD 68  // floats per row of the per-wave epilogue scratch (16 rows)
#define SCRATCH_BYTES (16 * SCRATCH_LD * 4)

template <int EPI> struct EpiTraits {
    static constexpr bool F32_OUT = (EPI == VITED_EPI_RESIDUAL || EPI == VITED_EPI_STORE_F32);
    static constexpr int PIECES = F32_OUT ? 4 : 2;      // (row, vector) pieces per lane per 16-row sub-tile
    static constexpr int WIDTH = F32_OUT ? 4 : 8;       // consecutive columns per piece
    static constexpr int ROWS_PER_PIECE = 16 / PIECES;  // rows covered by one store instruction
    __device__ static __forceinline__ int row(int lane, int piece) {
        return F32_OUT ? piece * 4 + (lane >> 4) : piece * 8 + (lane >> 3);
    }
    __device__ static __forceinline__ int col(int lane) { return F32_OUT ? (lane & 15) * 4 : (lane & 7) * 8; }
};

template <int EPI> struct EpiPrefetch {
    f32x4 res[EPI == VITED_EPI_RESIDUAL ? 4 : 1][4];          // [sub-tile][piece]
    bf16x8 aux[(EPI == VITED_EPI_MUL_GELU_GRAD || EPI == VITED_EPI_MUL) ? 4 : 1][2];    // [sub-tile][piece]
    f32x4 bias[2];                                            // this lane's 4 or 8 columns
};

__device__ __forceinline__ void remap_rows(const EpiParams& p, int64_t m, int64_t& orow, int64_t& rrow) {
    orow = m;
    rrow = m;
    if (p.rows_per_batch > 0) {
        // 32-bit division (gemm_nt_mfma_supported bounds M): the 64-bit one is a ~100-instruction routine, issued 32 times per
        // lane and tile by the patch-embedding GEMM's epilogue (K = 192: three K-steps of MFMA work per tile)
        const int64_t b = (uint32_t)m / (uint32_t)p.rows_per_batch, r = m - b * p.rows_per_batch + p.row_offset;
        orow = b * p.out_rows_per_batch + r;
        rrow = p.residual_bcast ? r : orow;
    }
}

// mtile = first row of the wave's 64-row slab, ntile = first column of its 64-column slab
template <int EPI>
__device__ __forceinline__ void epilogue_prefetch_bias(const EpiParams& p, EpiPrefetch<EPI>& pf, int64_t ntile, int64_t N, int lane) {
    using T = EpiTraits<EPI>;
    const int64_t n = ntile + T::col(lane);
    const bool ncol = n < N;
    pf.bias[0] = (p.bias && ncol) ? *(const f32x4*)(p.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
    pf.bias[1] = (p.bias && ncol && T::WIDTH == 8) ? *(const f32x4*)(p.bias + n + 4) : f32x4{0.f, 0.f, 0.f, 0.f};
}

// residual / saved pre-activation of the 16-row sub-tile i
template <int EPI>
__device__ __forceinline__ void epilogue_prefetch_subtile(const EpiParams& p, EpiPrefetch<EPI>& pf, int i, int64_t mtile, int64_t ntile,
                                                          int64_t M, int64_t N, int lane) {
    using T = EpiTraits<EPI>;
    const int64_t n = ntile + T::col(lane);
    const bool ncol = n < N;
    if constexpr (EPI == VITED_EPI_RESIDUAL) {
#pragma unroll
        for (int pc = 0; pc < 4; ++pc) {
            const int64_t m = mtile + i * 16 + T::row(lane, pc);
            int64_t orow, rrow;
            remap_rows(p, m, orow, rrow);
            pf.res[i][pc] = (m < M && ncol) ? EPI_LOAD16((const f32x4*)(p.residual + rrow * p.ldo + n)) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    if constexpr (EPI == VITED_EPI_MUL_GELU_GRAD || EPI == VITED_EPI_MUL) {
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) {
            const int64_t m = mtile + i * 16 + T::row(lane, pc);
            pf.aux[i][pc] = (m < M && ncol) ? EPI_LOAD16((const bf16x8*)((const bf16*)p.aux + m * p.ldo + n))
                                            : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        }
    }
}

template <int EPI>
__device__ __forceinline__ void epilogue_prefetch(const EpiParams& p, EpiPrefetch<EPI>& pf, int64_t mtile, int64_t ntile, int64_t M,
                                                  int64_t N, int lane) {
    epilogue_prefetch_bias<EPI>(p, pf, ntile, N, lane);
#pragma unroll
    for (int i = 0; i < 4; ++i) epilogue_prefetch_subtile<EPI>(p, pf, i, mtile, ntile, M, N, lane);
}

// one 16-row sub-tile i of the wave's slab: scratch (fp32 [16][SCRATCH_LD]) -> global
template <int EPI>
__device__ __forceinline__ void epilogue_subtile(const EpiParams& p, const EpiPrefetch<EPI>& pf, const float* sc, int i,
                                                 int64_t mtile, int64_t ntile, int64_t M, int64_t N, int lane) {
    using T = EpiTraits<EPI>;
    const int64_t n = ntile + T::col(lane);
    if (n >= N) return;
#pragma unroll
    for (int pc = 0; pc < T::PIECES; ++pc) {
        const int r = T::row(lane, pc);
        const int64_t m = mtile + i * 16 + r;
        if (m >= M) continue;
        const float* s = sc + r * SCRATCH_LD + T::col(lane);
        if constexpr (T::F32_OUT) {
            f32x4 v = *(const f32x4*)s;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += pf.bias[0][e];
            int64_t orow = m, rrow = m;
            if constexpr (EPI == VITED_EPI_RESIDUAL) {
                remap_rows(p, m, orow, rrow);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += pf.res[i][pc][e];
            }
            EPI_STORE16F((f32x4*)((float*)p.out + orow * p.ldo + n), v);
        } else {
            const f32x4 lo = *(const f32x4*)s, hi = *(const f32x4*)(s + 4);
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = lo[e] + pf.bias[0][e];
                v[4 + e] = hi[e] + pf.bias[1][e];
            }
            if constexpr (EPI == VITED_EPI_MUL_GELU_GRAD) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= gelu_grad_fast((float)pf.aux[i][pc][e]);
            }
            if constexpr (EPI == VITED_EPI_MUL) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] *= (float)pf.aux[i][pc][e];
            }
            if constexpr (EPI == VITED_EPI_GELU_GRAD) {
                bf16x8 pd, pg;     // one exponential serves both: exp(-z^2/2) is the Gaussian density AND the erfc tail
#pragma unroll
                for (int e = 0; e < 8; ++e) {
#ifdef NT_DBG_NO_GELU
                    pd[e] = (bf16)v[e];
                    pg[e] = (bf16)(v[e] * 0.5f);
#else
                    float cdf, ex;
                    gelu_parts_fast(v[e], cdf, ex);
                    pd[e] = (bf16)fmaf(v[e] * 0.39894228040143268f, ex, cdf);
                    pg[e] = (bf16)(v[e] * cdf);
#endif
                }
#ifdef NT_DBG_SKIP_STORES
                asm volatile("" :: "v"(pd), "v"(pg));
#else
                EPI_STORE16((bf16x8*)((bf16*)p.out + m * p.ldo + n), pd);
#ifdef NT_DBG_SKIP_OUT2
                asm volatile("" :: "v"(pg));
#else
                EPI_STORE16((bf16x8*)((bf16*)p.out2 + m * p.ldo + n), pg);
#endif
#endif
                continue;
            }
            bf16x8 pk;
#pragma unroll
            for (int e = 0; e < 8; ++e) pk[e] = (bf16)v[e];
            // the plain-store outputs (qkv / kv / q, do) are read by the very NEXT kernel (attention, the next GEMM): a normal store
            // leaves them where that kernel finds them (step 25.35 -> 25.2 ms against the streaming hint); everything else streams
            // (u = gelu(z) and dz also feed the next kernel, but at 200 MB each a normal store costs 0.45 ms: measured)
            if constexpr (EPI == VITED_EPI_STORE) *(bf16x8*)((bf16*)p.out + m * p.ldo + n) = pk;
            else EPI_STORE16((bf16x8*)((bf16*)p.out + m * p.ldo + n), pk);
            if constexpr (EPI == VITED_EPI_GELU) {
                bf16x8 pg;
#pragma unroll
                for (int e = 0; e < 8; ++e) pg[e] = (bf16)gelu_fast(v[e]);
                *(bf16x8*)((bf16*)p.out2 + m * p.ldo + n) = pg;
            }
        }
    }
}
