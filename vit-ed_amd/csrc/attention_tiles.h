// Shared device helpers of the bf16 MFMA attention kernels (attention_mfma.hip: short sequences,
// attention_flash.hip: tiled online-softmax for long sequences).
#pragma once
#include "attention_kernels.h"

#define SM_MAX_TILES 5  // 5 x 16 = 80 tokens

__device__ __forceinline__ bf16x4 tr_read4(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)p);
}

template <int HD> struct SmallCfg {
    static constexpr int ROW_BYTES = HD * 2;          // one token's head slice
    static constexpr int KCH = HD / 32;               // 32-wide k-steps of a q.k dot product
    static constexpr int DT = HD / 16;                // 16-wide tiles of the head dim
    static constexpr int TILE_BYTES = (SM_MAX_TILES + 1) * 16 * ROW_BYTES;  // +1: zero phantom tile of an odd last k-step
    // XOR on the 32-byte slot index so the 8 rows x 32 B a half-wave reads per tr-read hit 64 distinct banks
    __device__ static __forceinline__ int swz(int r) { return HD == 32 ? ((r >> 2) & 1) << 5 : ((r >> 1) & 3) << 5; }
    __device__ static __forceinline__ int off(int r, int byte_in_row) { return r * ROW_BYTES + (byte_in_row ^ swz(r)); }
};

// 16-row x HD tile -> registers in MFMA row-fragment form (lane: row fr, 8 elements at 8*(g + 4*c)),
// rows clamped to n-1.
template <int HD>
__device__ __forceinline__ void load_row_frags(const bf16* base, int64_t ts, int row0, int n, int fr, int g,
                                               bf16x8 (&f)[HD / 32]) {
    int r = row0 + fr;
    r = r < n ? r : n - 1;
#pragma unroll
    for (int c = 0; c < HD / 32; ++c) f[c] = *(const bf16x8*)(base + (int64_t)r * ts + 8 * (g + 4 * c));
}

template <int HD>
__device__ __forceinline__ void stage_tile(char* lds, int row0, int n, int fr, int g, const bf16x8 (&f)[HD / 32]) {
    const int r = row0 + fr;
#pragma unroll
    for (int c = 0; c < HD / 32; ++c) {
        bf16x8 v = f[c];
        if (r >= n) v = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        *(bf16x8*)(lds + SmallCfg<HD>::off(r, 16 * (g + 4 * c))) = v;
    }
}

// transposed operand: 16 (head-dim) x 32 (token) fragment for k-step ks of tile image `lds`, with the
// in-step token order (4g..4g+3 of the even tile, then 4g..4g+3 of the odd tile) that matches an
// accumulator pair used as the other operand.
template <int HD>
__device__ __forceinline__ bf16x8 tr_frag(const char* lds, int ks, int dt, int g, int qq, int p) {
    const int r0 = 32 * ks + 4 * g + qq, r1 = r0 + 16;
    const bf16x4 lo = tr_read4(lds + SmallCfg<HD>::off(r0, dt * 32 + 8 * p));
    const bf16x4 hi = tr_read4(lds + SmallCfg<HD>::off(r1, dt * 32 + 8 * p));
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

__device__ __forceinline__ bf16x8 pack_pair(const f32x4& a, const f32x4& b) {
    return bf16x8{(bf16)a[0], (bf16)a[1], (bf16)a[2], (bf16)a[3], (bf16)b[0], (bf16)b[1], (bf16)b[2], (bf16)b[3]};
}

__device__ __forceinline__ float group_max4(float v) {  // over the 4 lanes that share lane & 15
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float group_sum4(float v) {
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}

#define LOG2E 1.4426950408889634f
#define LN2 0.6931471805599453f

