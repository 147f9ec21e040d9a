// bf16 MFMA attention for long sequences (BASELINE config H: 1024 / 1025 tokens, head_dim 64):
// tiled, online-softmax ("flash") forward and a two-kernel backward, no N x N tensor anywhere.
//
// Same register algebra as the short-sequence kernels (attention_mfma.hip): scores are computed
// transposed (keys on accumulator rows, the query on the lane) so the softmax of a query lives in one
// lane's registers plus two shuffles, and the probabilities ARE the next MFMA's operand.
//
//   forward   one workgroup = 4 waves x 32 queries of one (batch, head); 64-key K/V tiles are staged
//             cooperatively (coalesced 16-byte loads -> registers -> XOR-swizzled LDS image, double
//             buffered: the next tile's loads are in flight under the current tile's MFMAs); running
//             max / sum per query, O rescaled per tile; LSE saved.
//   backward  dQ kernel : the forward loop with S^T, dP^T = V dO^T, dS^T = P^T o (dP^T - delta),
//                         dQ^T += K^T dS^T (K read transposed from the same LDS image).  delta: the textbook
//                         rowsum(dO o O) uses the forward's bf16-ROUNDED O, which is inconsistent with the fp32 P o dP it is
//                         subtracted from - on near-uniform attention (dP ~ delta for every key) that bias dominates dS
//                         (measured at 1024 keys, 4 + 4 blocks: d(kv) 83 % off where PyTorch's bf16 autocast is 7 % off).
//                         The loop therefore runs with that estimate, accumulates the EXACT delta = sum_j P o dP of its own
//                         P and dP beside it (one FMA per score) together with B^T = K^T P^T (one more MFMA product on the
//                         K^T fragment already loaded), and finishes dQ -= (delta - estimate) * B: algebraically the softmax
//                         backward with the exact delta, in one pass.  The exact delta is what the second kernel reads.
//             dKV kernel: one workgroup = 4 waves x 32 keys; loops over 64-query tiles of Q / dO / lse /
//                         delta; S = Q K^T and dP = dO V^T put the query on accumulator rows, so
//                         dV^T += dO^T P and dK^T += Q^T dS contract over accumulator rows again.
//             Both kernels accumulate in registers across the loop: no atomics, bit-reproducible.
#include <type_traits>
#include "attention_tiles.h"

#define FL_TILE 64   // keys (or queries) per LDS tile
#ifndef FL_W
#define FL_W 2       // 16-row tiles per wave
#endif

template <int HD> struct FlashCfg {
    using C = SmallCfg<HD>;
    static constexpr int TILE_BYTES = FL_TILE * C::ROW_BYTES;
    static constexpr int CHUNKS_PER_ROW = HD / 8;
    static constexpr int PASSES = FL_TILE * CHUNKS_PER_ROW / 256;   // 16-byte chunks per thread and tile
};

// This thread's 16-byte chunks of a 64-row tile: what does not change from tile to tile (the loops below are VALU-bound - PMC:
// 8 vector instructions per MFMA in the forward - and the staging arithmetic redone per tile was a third of them: row clamp as
// compare + select, 64-bit row * stride, the swizzle, a select per stored dword).
template <int HD> struct TileMap {
    int row[FlashCfg<HD>::PASSES];   // row inside the tile
    int col2[FlashCfg<HD>::PASSES];  // column, in bytes
    int loff[FlashCfg<HD>::PASSES];  // byte offset inside the swizzled LDS image
    __device__ __forceinline__ TileMap(int tid) {
#pragma unroll
        for (int c = 0; c < FlashCfg<HD>::PASSES; ++c) {
            const int idx = tid + 256 * c;
            const int ch = idx % FlashCfg<HD>::CHUNKS_PER_ROW;
            row[c] = idx / FlashCfg<HD>::CHUNKS_PER_ROW;
            col2[c] = ch * 16;
            loff[c] = SmallCfg<HD>::off(row[c], ch * 16);
        }
    }
};
// cooperative tile copy, global -> registers.  Rows past n - 1 re-read row n - 1 (finite values; regs_to_tile zeroes them): the
// clamp is one v_min against a wave-uniform bound and the address is a 64-bit uniform tile base + a 32-bit per-lane byte offset
// (attention_fwd_flash / attention_bwd_flash refuse token strides of 2^24 elements or more) ...
template <int HD>
__device__ __forceinline__ void tile_to_regs(const bf16* base, int64_t ts, int row0, int n, const TileMap<HD>& m, bf16x8 (&r)[FlashCfg<HD>::PASSES]) {
    const char* tb = (const char*)(base + (int64_t)row0 * ts);   // wave-uniform
    const int last = n - 1 - row0;                               // >= 0: a tile is only staged when it holds a valid row
    const unsigned ts2 = (unsigned)ts * 2u;
#pragma unroll
    for (int c = 0; c < FlashCfg<HD>::PASSES; ++c) {
        const unsigned rr = (unsigned)(m.row[c] < last ? m.row[c] : last);
        r[c] = *(const bf16x8*)(tb + (rr * ts2 + (unsigned)m.col2[c]));
    }
}
// ... and registers -> LDS image (rows >= n zeroed; only the ragged last tile pays for the selects)
template <int HD>
__device__ __forceinline__ void regs_to_tile(char* lds, int row0, int n, const TileMap<HD>& m, const bf16x8 (&r)[FlashCfg<HD>::PASSES]) {
    if (row0 + FL_TILE > n) {   // wave-uniform
        asm volatile("" ::: "memory");   // keep this a branch: if-converted it costs a select per stored dword on EVERY tile
#pragma unroll
        for (int c = 0; c < FlashCfg<HD>::PASSES; ++c) {
            bf16x8 v = r[c];
            if (row0 + m.row[c] >= n) v = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
            *(bf16x8*)(lds + m.loff[c]) = v;
        }
    } else {
#pragma unroll
        for (int c = 0; c < FlashCfg<HD>::PASSES; ++c) *(bf16x8*)(lds + m.loff[c]) = r[c];
    }
}
// MFMA row fragment (row fr of 16-row tile t, k chunk g + 4c) out of an LDS image
template <int HD>
__device__ __forceinline__ void lds_row_frags(const char* lds, int t, int fr, int g, bf16x8 (&f)[HD / 32]) {
#pragma unroll
    for (int c = 0; c < HD / 32; ++c) f[c] = *(const bf16x8*)(lds + SmallCfg<HD>::off(16 * t + fr, 16 * (g + 4 * c)));
}

// The file is compiled twice (Makefile): FLASH_PART 1 = forward, built with -mllvm -amdgpu-mfma-vgpr-form
// (the O accumulators stay in VGPRs: no AGPR<->VGPR moves around the rescale, forward +9.6 % at config H);
// FLASH_PART 2 = backward, built without it (the flag costs the backward kernels ~2 %).  0 = everything.
#ifndef FLASH_PART
#define FLASH_PART 0
#endif

#if FLASH_PART != 2
// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
template <int HD>
__global__ void __launch_bounds__(256)
attn_fwd_flash_kernel(AttnArgs a) {
    using C = SmallCfg<HD>;
    using F = FlashCfg<HD>;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 stages][K tile | V tile]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, g = lane >> 4, qq = fr >> 2, p = fr & 3;
    const int64_t b = blockIdx.z;
    const int h = blockIdx.y;
    const int nq = (int)a.nq, nk = (int)a.nk;
    const int q0 = blockIdx.x * (64 * FL_W) + wave * (16 * FL_W);
    const int64_t hoff = (int64_t)h * HD;
    const bf16* qb = (const bf16*)a.q + b * a.q_bs + hoff;
    const int64_t bkv = a.kv_index ? a.kv_index[b] : b;
    const bf16* kb = (const bf16*)a.k + bkv * a.k_bs + hoff;
    const bf16* vb = (const bf16*)a.v + bkv * a.v_bs + hoff;
    const float sc = a.scale * LOG2E;

    bf16x8 qf[FL_W][C::KCH];
    f32x4 o[FL_W][C::DT];
    float m[FL_W], l[FL_W];
#pragma unroll
    for (int w = 0; w < FL_W; ++w) {
        load_row_frags<HD>(qb, a.q_ts, q0 + 16 * w, nq, fr, g, qf[w]);
        m[w] = -INFINITY;
        l[w] = 0.f;
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt) o[w][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int ntiles = (nk + FL_TILE - 1) / FL_TILE;
    bf16x8 kr[F::PASSES], vr[F::PASSES];
    const TileMap<HD> mk(tid);
    const TileMap<HD>& mv = mk;
    tile_to_regs<HD>(kb, a.k_ts, 0, nk, mk, kr);
    tile_to_regs<HD>(vb, a.v_ts, 0, nk, mv, vr);
    regs_to_tile<HD>(smem, 0, nk, mk, kr);
    regs_to_tile<HD>(smem + F::TILE_BYTES, 0, nk, mv, vr);
    __syncthreads();
    // One tile = one call of `iteration`.  The ragged last tile is PEELED out of the loop (second instantiation below) instead of
    // being a branch inside it: with two bodies merging inside the loop hipcc copied all 32 O accumulators at every back-edge.
    auto iteration = [&](int t, auto ragged_tag) {
        const char* ks = smem + (t & 1) * 2 * F::TILE_BYTES;
        const char* vs = ks + F::TILE_BYTES;
        if (t + 1 < ntiles) {   // next tile: loads fly under this tile's MFMAs
            tile_to_regs<HD>(kb, a.k_ts, (t + 1) * FL_TILE, nk, mk, kr);
            tile_to_regs<HD>(vb, a.v_ts, (t + 1) * FL_TILE, nk, mv, vr);
        }
        bf16x8 kf[4][C::KCH];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) lds_row_frags<HD>(ks, kt, fr, g, kf[kt]);
        // The loop is VALU-bound (softmax work per score > MFMA time per score at hd 64), so every VALU instruction
        // counts: the key mask exists only in a ragged last tile, the scale rides in the exp2 argument's FMA, exp2 is
        // the raw v_exp_f32 (a probability below 2^-126 is 0 either way) and O is rescaled only when some lane's
        // running maximum moved.
        {
        constexpr bool RAGGED = decltype(ragged_tag)::value;
        f32x4 st[FL_W][4];
#pragma unroll
        for (int w = 0; w < FL_W; ++w) {
            float tmax = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c = 0; c < C::KCH; ++c) s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[kt][c], qf[w][c], s, 0, 0, 0);
                if constexpr (RAGGED) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) s[e] = (t * FL_TILE + 16 * kt + 4 * g + e) < nk ? s[e] : -INFINITY;
                }
                tmax = fmaxf(fmaxf(tmax, fmaxf(s[0], s[1])), fmaxf(s[2], s[3]));
                st[w][kt] = s;
            }
            tmax = group_max4(tmax) * sc;          // sc > 0: the maximum commutes with the scale
            // Lazy reference maximum: m[w] only moves when some query's tile maximum exceeds it by more than 2^8
            // (probabilities then stay below 256 - exact in fp32 sums, harmless in bf16 operands); the softmax and the
            // LSE are mathematically unchanged, and the O rescale (accumulator -> VGPR -> accumulator) becomes rare.
            if (__any(tmax > m[w] + 8.f)) {
                asm volatile("" ::: "memory");   // keep this a branch
                const float mn = fmaxf(m[w], tmax);
                const float alpha = __builtin_amdgcn_exp2f(m[w] - mn);
                l[w] *= alpha;
                m[w] = mn;
#pragma unroll
                for (int dt = 0; dt < C::DT; ++dt)
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[w][dt][e] *= alpha;
            }
            const float mref = m[w];
            float ls = 0.f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float pe = __builtin_amdgcn_exp2f(fmaf(st[w][kt][e], sc, -mref));
                    st[w][kt][e] = pe;
                    ls += pe;
                }
            l[w] += group_sum4(ls);
        }
        // O^T += V^T P^T: each transposed V fragment is read once and feeds every query tile of the wave
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt)
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                const bf16x8 vt = tr_frag<HD>(vs, k2, dt, g, qq, p);
#pragma unroll
                for (int w = 0; w < FL_W; ++w)
                    o[w][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vt, pack_pair(st[w][2 * k2], st[w][2 * k2 + 1]), o[w][dt], 0, 0, 0);
            }
        }
        if (t + 1 < ntiles) {
            char* nks = smem + ((t + 1) & 1) * 2 * F::TILE_BYTES;
            regs_to_tile<HD>(nks, (t + 1) * FL_TILE, nk, mk, kr);
            regs_to_tile<HD>(nks + F::TILE_BYTES, (t + 1) * FL_TILE, nk, mv, vr);
        }
        __syncthreads();
    };
    const int nfull = nk / FL_TILE;     // tiles without a masked key
    for (int t = 0; t < nfull; ++t) iteration(t, std::false_type{});
    if (nfull < ntiles) iteration(nfull, std::true_type{});
#pragma unroll
    for (int w = 0; w < FL_W; ++w) {
        const int q = q0 + 16 * w + fr;
        if (q < nq) {
            const float inv = 1.f / l[w];
#pragma unroll
            for (int dt = 0; dt < C::DT; ++dt) {
                const bf16x4 ov = {(bf16)(o[w][dt][0] * inv), (bf16)(o[w][dt][1] * inv), (bf16)(o[w][dt][2] * inv), (bf16)(o[w][dt][3] * inv)};
                *(bf16x4*)((bf16*)a.o + b * a.o_bs + (int64_t)q * a.o_ts + hoff + dt * 16 + 4 * g) = ov;
            }
            if (g == 0) a.lse[(b * a.heads + h) * a.nq + q] = (m[w] + log2f(l[w])) * LN2;
        }
    }
}

#endif  // forward

#if FLASH_PART != 1
// ------------------------------------------------------------------------------------------------
// backward 1/2: dQ (and delta)
// ------------------------------------------------------------------------------------------------
template <int HD>
__global__ void __launch_bounds__(256, 2)   // 2 waves per SIMD: <= 256 VGPRs
attn_bwd_dq_flash_kernel(AttnArgs a) {
    using C = SmallCfg<HD>;
    using F = FlashCfg<HD>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, g = lane >> 4, qq = fr >> 2, p = fr & 3;
    const int64_t b = blockIdx.z;
    const int h = blockIdx.y;
    const int nq = (int)a.nq, nk = (int)a.nk;
    const int q0 = blockIdx.x * (64 * FL_W) + wave * (16 * FL_W);
    const int64_t hoff = (int64_t)h * HD;
    const bf16* qb = (const bf16*)a.q + b * a.q_bs + hoff;
    const bf16* kb = (const bf16*)a.k + b * a.k_bs + hoff;
    const bf16* vb = (const bf16*)a.v + b * a.v_bs + hoff;
    const bf16* ob = (const bf16*)a.o + b * a.o_bs + hoff;
    const bf16* dob = (const bf16*)a.d_o + b * a.o_bs + hoff;
    const float sc = a.scale * LOG2E;

    bf16x8 qf[FL_W][C::KCH], dof[FL_W][C::KCH];
    f32x4 dq[FL_W][C::DT], bq[FL_W][C::DT];   // dQ^T with the estimated delta; B^T = K^T P^T for the exact-delta correction
    float lse2[FL_W], dl[FL_W], dex[FL_W];    // dl: delta estimated from the saved (bf16) O; dex: this lane's part of sum_j P o dP
#pragma unroll
    for (int w = 0; w < FL_W; ++w) {
        load_row_frags<HD>(qb, a.q_ts, q0 + 16 * w, nq, fr, g, qf[w]);
        load_row_frags<HD>(dob, a.o_ts, q0 + 16 * w, nq, fr, g, dof[w]);
        bf16x8 of[C::KCH];
        load_row_frags<HD>(ob, a.o_ts, q0 + 16 * w, nq, fr, g, of);
        float d = 0.f;
#pragma unroll
        for (int c = 0; c < C::KCH; ++c)
#pragma unroll
            for (int e = 0; e < 8; ++e) d = fmaf((float)of[c][e], (float)dof[w][c][e], d);
        dl[w] = group_sum4(d);
        dex[w] = 0.f;
        const int q = q0 + 16 * w + fr;
        const int qc = q < nq ? q : nq - 1;
        lse2[w] = a.lse[(b * a.heads + h) * a.nq + qc] * LOG2E;
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt) {
            dq[w][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
            bq[w][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    const int ntiles = (nk + FL_TILE - 1) / FL_TILE;
    bf16x8 kr[F::PASSES], vr[F::PASSES];
    const TileMap<HD> mk(tid);
    const TileMap<HD>& mv = mk;
    tile_to_regs<HD>(kb, a.k_ts, 0, nk, mk, kr);
    tile_to_regs<HD>(vb, a.v_ts, 0, nk, mv, vr);
    regs_to_tile<HD>(smem, 0, nk, mk, kr);
    regs_to_tile<HD>(smem + F::TILE_BYTES, 0, nk, mv, vr);
    __syncthreads();
    auto iteration = [&](int t, auto ragged_tag) {   // the ragged last tile is peeled out of the loop (see the forward kernel)
        const char* ks = smem + (t & 1) * 2 * F::TILE_BYTES;
        const char* vs = ks + F::TILE_BYTES;
        if (t + 1 < ntiles) {
            tile_to_regs<HD>(kb, a.k_ts, (t + 1) * FL_TILE, nk, mk, kr);
            tile_to_regs<HD>(vb, a.v_ts, (t + 1) * FL_TILE, nk, mv, vr);
        }
        bf16x8 kf[4][C::KCH], vf[4][C::KCH];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            lds_row_frags<HD>(ks, kt, fr, g, kf[kt]);
            lds_row_frags<HD>(vs, kt, fr, g, vf[kt]);
        }
        {   // VALU-bound like the forward loop: same instruction diet
        constexpr bool RAGGED = decltype(ragged_tag)::value;
        f32x4 ds[FL_W][4];
        bf16x8 pp[FL_W][2];     // the probabilities themselves, packed for the B^T product
#pragma unroll
        for (int w = 0; w < FL_W; ++w) {
            f32x4 pk[2];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c = 0; c < C::KCH; ++c) {
                    s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[kt][c], qf[w][c], s, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[kt][c], dof[w][c], dp, 0, 0, 0);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float pe = __builtin_amdgcn_exp2f(fmaf(s[e], sc, -lse2[w]));
                    if constexpr (RAGGED) pe = (t * FL_TILE + 16 * kt + 4 * g + e) < nk ? pe : 0.f;
                    ds[w][kt][e] = pe * (dp[e] - dl[w]);
                    dex[w] = fmaf(pe, dp[e], dex[w]);
                    pk[kt & 1][e] = pe;
                }
                if (kt & 1) pp[w][kt >> 1] = pack_pair(pk[0], pk[1]);
            }
        }
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt)
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                const bf16x8 kt_ = tr_frag<HD>(ks, k2, dt, g, qq, p);   // one transposed K fragment feeds every query tile, both products
#pragma unroll
                for (int w = 0; w < FL_W; ++w) {
                    dq[w][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt_, pack_pair(ds[w][2 * k2], ds[w][2 * k2 + 1]), dq[w][dt], 0, 0, 0);
                    bq[w][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kt_, pp[w][k2], bq[w][dt], 0, 0, 0);
                }
            }
        }
        if (t + 1 < ntiles) {
            char* nks = smem + ((t + 1) & 1) * 2 * F::TILE_BYTES;
            regs_to_tile<HD>(nks, (t + 1) * FL_TILE, nk, mk, kr);
            regs_to_tile<HD>(nks + F::TILE_BYTES, (t + 1) * FL_TILE, nk, mv, vr);
        }
        __syncthreads();
    };
    const int nfull = nk / FL_TILE;
    for (int t = 0; t < nfull; ++t) iteration(t, std::false_type{});
    if (nfull < ntiles) iteration(nfull, std::true_type{});
#pragma unroll
    for (int w = 0; w < FL_W; ++w) {
        const int q = q0 + 16 * w + fr;
        // exact delta of this query = sum over ALL keys of P o dP (the four lane groups hold disjoint keys), and the correction
        //   dQ = sum_j P (dP - delta) K = [sum_j P (dP - estimate) K] - (delta - estimate) [sum_j P K]
        const float dexact = group_sum4(dex[w]);
        const float corr = dexact - dl[w];
        if (q < nq) {
            if (g == 0) a.delta[(b * a.heads + h) * a.nq + q] = dexact;
#pragma unroll
            for (int dt = 0; dt < C::DT; ++dt) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (dq[w][dt][e] - corr * bq[w][dt][e]) * a.scale;
                const bf16x4 ov = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
                *(bf16x4*)((bf16*)a.dq + b * a.dq_bs + (int64_t)q * a.dq_ts + hoff + dt * 16 + 4 * g) = ov;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// backward 2/2: dK, dV
// ------------------------------------------------------------------------------------------------
template <int HD>
__global__ void __launch_bounds__(256, 2)   // 2 waves per SIMD: <= 256 VGPRs
attn_bwd_dkv_flash_kernel(AttnArgs a) {
    using C = SmallCfg<HD>;
    using F = FlashCfg<HD>;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 stages][Q tile | dO tile | lse 64 f32 | delta 64 f32]
    constexpr int STAGE_BYTES = 2 * F::TILE_BYTES + 2 * FL_TILE * 4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, g = lane >> 4, qq = fr >> 2, p = fr & 3;
    const int64_t b = blockIdx.z;
    const int h = blockIdx.y;
    const int nq = (int)a.nq, nk = (int)a.nk;
    const int k0 = blockIdx.x * (64 * FL_W) + wave * (16 * FL_W);
    const int64_t hoff = (int64_t)h * HD;
    const bf16* qb = (const bf16*)a.q + b * a.q_bs + hoff;
    const bf16* kb = (const bf16*)a.k + b * a.k_bs + hoff;
    const bf16* vb = (const bf16*)a.v + b * a.v_bs + hoff;
    const bf16* dob = (const bf16*)a.d_o + b * a.o_bs + hoff;
    const float* lse_g = a.lse + (b * a.heads + h) * a.nq;
    const float* del_g = a.delta + (b * a.heads + h) * a.nq;
    const float sc = a.scale * LOG2E;

    bf16x8 kf[FL_W][C::KCH], vf[FL_W][C::KCH];
    f32x4 dk[FL_W][C::DT], dv[FL_W][C::DT];
#pragma unroll
    for (int w = 0; w < FL_W; ++w) {
        load_row_frags<HD>(kb, a.k_ts, k0 + 16 * w, nk, fr, g, kf[w]);
        load_row_frags<HD>(vb, a.v_ts, k0 + 16 * w, nk, fr, g, vf[w]);
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt) {
            dk[w][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
            dv[w][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    const int ntiles = (nq + FL_TILE - 1) / FL_TILE;
    bf16x8 qr[F::PASSES], dor[F::PASSES];
    const TileMap<HD> mq(tid);
    const TileMap<HD>& mdo = mq;
    float statr = 0.f;
    auto load_stats = [&](int t) {
        if (tid < 2 * FL_TILE) {
            int q = t * FL_TILE + (tid & (FL_TILE - 1));
            q = q < nq ? q : nq - 1;
            statr = tid < FL_TILE ? lse_g[q] * LOG2E : del_g[q];
        }
    };
    auto store_stats = [&](char* stage) {
        if (tid < 2 * FL_TILE) ((float*)(stage + 2 * F::TILE_BYTES))[tid] = statr;
    };
    tile_to_regs<HD>(qb, a.q_ts, 0, nq, mq, qr);
    tile_to_regs<HD>(dob, a.o_ts, 0, nq, mdo, dor);
    load_stats(0);
    regs_to_tile<HD>(smem, 0, nq, mq, qr);
    regs_to_tile<HD>(smem + F::TILE_BYTES, 0, nq, mdo, dor);
    store_stats(smem);
    __syncthreads();
    auto iteration = [&](int t, auto ragged_tag) {   // the ragged last tile is peeled out of the loop (see the forward kernel)
        const char* qs = smem + (t & 1) * STAGE_BYTES;
        const char* dos = qs + F::TILE_BYTES;
        const float* lse_s = (const float*)(qs + 2 * F::TILE_BYTES);
        const float* del_s = lse_s + FL_TILE;
        if (t + 1 < ntiles) {
            tile_to_regs<HD>(qb, a.q_ts, (t + 1) * FL_TILE, nq, mq, qr);
            tile_to_regs<HD>(dob, a.o_ts, (t + 1) * FL_TILE, nq, mdo, dor);
            load_stats(t + 1);
        }
        {   // VALU-bound: see the forward kernel
        constexpr bool RAGGED = decltype(ragged_tag)::value;
        f32x4 pr[FL_W][4], ds[FL_W][4];
#pragma unroll
        for (int qt = 0; qt < 4; ++qt) {
            bf16x8 qfr[C::KCH], dofr[C::KCH];   // one LDS read of the query-tile fragments serves every key tile of the wave
            lds_row_frags<HD>(qs, qt, fr, g, qfr);
            lds_row_frags<HD>(dos, qt, fr, g, dofr);
            const f32x4 l4 = *(const f32x4*)(lse_s + 16 * qt + 4 * g);
            const f32x4 d4 = *(const f32x4*)(del_s + 16 * qt + 4 * g);
#pragma unroll
            for (int w = 0; w < FL_W; ++w) {
                f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int c = 0; c < C::KCH; ++c) {
                    s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qfr[c], kf[w][c], s, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dofr[c], vf[w][c], dp, 0, 0, 0);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float pe = __builtin_amdgcn_exp2f(fmaf(s[e], sc, -l4[e]));
                    if constexpr (RAGGED) pe = (t * FL_TILE + 16 * qt + 4 * g + e) < nq ? pe : 0.f;
                    pr[w][qt][e] = pe;
                    ds[w][qt][e] = pe * (dp[e] - d4[e]);
                }
            }
        }
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt)
#pragma unroll
            for (int q2 = 0; q2 < 2; ++q2) {
                const bf16x8 dot_ = tr_frag<HD>(dos, q2, dt, g, qq, p), qt_ = tr_frag<HD>(qs, q2, dt, g, qq, p);
#pragma unroll
                for (int w = 0; w < FL_W; ++w) {
                    dv[w][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dot_, pack_pair(pr[w][2 * q2], pr[w][2 * q2 + 1]), dv[w][dt], 0, 0, 0);
                    dk[w][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qt_, pack_pair(ds[w][2 * q2], ds[w][2 * q2 + 1]), dk[w][dt], 0, 0, 0);
                }
            }
        }
        if (t + 1 < ntiles) {
            char* nst = smem + ((t + 1) & 1) * STAGE_BYTES;
            regs_to_tile<HD>(nst, (t + 1) * FL_TILE, nq, mq, qr);
            regs_to_tile<HD>(nst + F::TILE_BYTES, (t + 1) * FL_TILE, nq, mdo, dor);
            store_stats(nst);
        }
        __syncthreads();
    };
    const int nfull = nq / FL_TILE;
    for (int t = 0; t < nfull; ++t) iteration(t, std::false_type{});
    if (nfull < ntiles) iteration(nfull, std::true_type{});
#pragma unroll
    for (int w = 0; w < FL_W; ++w) {
        const int key = k0 + 16 * w + fr;
        if (key < nk) {
#pragma unroll
            for (int dt = 0; dt < C::DT; ++dt) {
                const bf16x4 vv = {(bf16)dv[w][dt][0], (bf16)dv[w][dt][1], (bf16)dv[w][dt][2], (bf16)dv[w][dt][3]};
                const bf16x4 kk = {(bf16)(dk[w][dt][0] * a.scale), (bf16)(dk[w][dt][1] * a.scale), (bf16)(dk[w][dt][2] * a.scale), (bf16)(dk[w][dt][3] * a.scale)};
                *(bf16x4*)((bf16*)a.dv + b * a.dv_bs + (int64_t)key * a.dv_ts + hoff + dt * 16 + 4 * g) = vv;
                *(bf16x4*)((bf16*)a.dk + b * a.dk_bs + (int64_t)key * a.dk_ts + hoff + dt * 16 + 4 * g) = kk;
            }
        }
    }
}

#endif  // backward

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
#if FLASH_PART != 2
template <int HD>
static int launch_flash_fwd(const AttnArgs& a, hipStream_t s) {
    dim3 grid((unsigned)((a.nq + 64 * FL_W - 1) / (64 * FL_W)), (unsigned)a.heads, (unsigned)a.batch);
    hipLaunchKernelGGL((attn_fwd_flash_kernel<HD>), grid, dim3(256), 4 * FlashCfg<HD>::TILE_BYTES, s, a);
    return vited_check_launch();
}

#endif

#if FLASH_PART != 1
template <int HD>
static int launch_flash_bwd(const AttnArgs& a, hipStream_t s) {
    dim3 gq((unsigned)((a.nq + 64 * FL_W - 1) / (64 * FL_W)), (unsigned)a.heads, (unsigned)a.batch);
    dim3 gk((unsigned)((a.nk + 64 * FL_W - 1) / (64 * FL_W)), (unsigned)a.heads, (unsigned)a.batch);
    hipLaunchKernelGGL((attn_bwd_dq_flash_kernel<HD>), gq, dim3(256), 4 * FlashCfg<HD>::TILE_BYTES, s, a);
    hipLaunchKernelGGL((attn_bwd_dkv_flash_kernel<HD>), gk, dim3(256), 2 * (2 * FlashCfg<HD>::TILE_BYTES + 2 * FL_TILE * 4), s, a);
    return vited_check_launch();
}

#endif

#if FLASH_PART != 2
// the staging code addresses a tile as a 64-bit base + 32-bit byte offsets: 64 rows x token stride x 2 B must fit
static inline bool flash_strides_ok(const AttnArgs& a) {
    constexpr int64_t LIM = (int64_t)1 << 24;
    return a.q_ts < LIM && a.k_ts < LIM && a.v_ts < LIM && a.o_ts < LIM;
}

int attention_fwd_flash(const AttnArgs& a, hipStream_t s) {
    if (!flash_strides_ok(a)) return VITED_ERR_UNSUPPORTED;
    if (a.head_dim == 32) return launch_flash_fwd<32>(a, s);
    if (a.head_dim == 64) return launch_flash_fwd<64>(a, s);
    return VITED_ERR_UNSUPPORTED;
}

#endif

#if FLASH_PART != 1
int attention_bwd_flash(const AttnArgs& a, hipStream_t s) {
    if (a.q_ts >= ((int64_t)1 << 24) || a.k_ts >= ((int64_t)1 << 24) || a.v_ts >= ((int64_t)1 << 24) || a.o_ts >= ((int64_t)1 << 24))
        return VITED_ERR_UNSUPPORTED;
    if (a.head_dim == 32) return launch_flash_bwd<32>(a, s);
    if (a.head_dim == 64) return launch_flash_bwd<64>(a, s);
    return VITED_ERR_UNSUPPORTED;
}
#endif
