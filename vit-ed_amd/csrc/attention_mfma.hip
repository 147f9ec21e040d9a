// bf16 MFMA attention for short sequences (Nq, Nk <= 80 tokens: BASELINE config A has 64 / 65),
// head_dim 32 or 64.  One 64-lane wave owns one (batch, head); a workgroup's 4 waves take 4
// adjacent heads of the same batch element so that their 64/128-byte head slices of each token row
// share cache lines.  HBM-bound by design (the scores never leave registers):
//
//   forward :  S^T = K Q^T   (v_mfma_f32_16x16x32_bf16, keys on accumulator rows, queries on lanes)
//              softmax over keys = over a lane's registers + 2 cross-lane shuffles (no LDS)
//              O^T = V^T P^T : P^T is already the B operand in registers (a consistent permutation
//              of the key index inside each 32-key MFMA step makes the accumulator layout legal);
//              V^T comes from a wave-private, XOR-swizzled LDS image read with ds_read_b64_tr_b16.
//   backward:  both orientations of the scores are recomputed from Q, K and the saved LSE
//              (S^T for dQ, S for dK / dV) so every product contracts over an accumulator-row
//              index; Q, K, dO are staged once in LDS for the transposed operand reads.
//
// Rows beyond Nq / Nk are clamped on load (finite data), masked to -inf / 0 in the softmax and never
// stored.
#include "attention_tiles.h"

template <int HD, int NKT>
__global__ void __launch_bounds__(256)
attn_fwd_small_kernel(AttnArgs a) {
    using C = SmallCfg<HD>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fr = lane & 15, g = lane >> 4, qq = fr >> 2, p = fr & 3;
    const int64_t b = blockIdx.x;
    int h = blockIdx.y * 4 + wave;
    const bool live = h < a.heads;
    h = live ? h : a.heads - 1;
    const int nq = (int)a.nq, nk = (int)a.nk;
    const bf16* qb = (const bf16*)a.q + b * a.q_bs + (int64_t)h * HD;
    const int64_t bkv = a.kv_index ? a.kv_index[b] : b;
    const bf16* kb = (const bf16*)a.k + bkv * a.k_bs + (int64_t)h * HD;
    const bf16* vb = (const bf16*)a.v + bkv * a.v_bs + (int64_t)h * HD;
    char* vs = smem + wave * C::TILE_BYTES;
    char* os = smem + 4 * C::TILE_BYTES + wave * 16 * C::ROW_BYTES;   // output scratch: one 16-row tile per wave

    bf16x8 kf[NKT][C::KCH], qnext[C::KCH];
    load_row_frags<HD>(qb, a.q_ts, 0, nq, fr, g, qnext);   // query tiles are prefetched one iteration ahead
#pragma unroll
    for (int t = 0; t < NKT; ++t) {
        load_row_frags<HD>(kb, a.k_ts, 16 * t, nk, fr, g, kf[t]);
        bf16x8 vf[C::KCH];
        load_row_frags<HD>(vb, a.v_ts, 16 * t, nk, fr, g, vf);
        stage_tile<HD>(vs, 16 * t, nk, fr, g, vf);
    }
    if (NKT & 1) {  // the odd last k-step reads a phantom tile: keep it finite (zeros)
        const bf16x8 z[C::KCH] = {};
        stage_tile<HD>(vs, 16 * NKT, 0, fr, g, z);
    }
    // no barrier: the V image is wave-private and DS operations of one wave execute in order (the empty asm only
    // stops the COMPILER from moving the transposed reads above the staging stores)
    asm volatile("" ::: "memory");
    const float sc = a.scale * LOG2E;
    const int nqt = (nq + 15) >> 4;
    for (int i = 0; i < nqt; ++i) {
        bf16x8 qf[C::KCH];
#pragma unroll
        for (int c = 0; c < C::KCH; ++c) qf[c] = qnext[c];
        if (i + 1 < nqt) load_row_frags<HD>(qb, a.q_ts, 16 * (i + 1), nq, fr, g, qnext);
        f32x4 st[NKT + 1];
        float m = -INFINITY;
#pragma unroll
        for (int t = 0; t < NKT; ++t) {
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < C::KCH; ++c) s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[t][c], qf[c], s, 0, 0, 0);
            if (t == NKT - 1) {   // only the last key tile can be ragged (NKT = ceil(nk / 16))
#pragma unroll
                for (int e = 0; e < 4; ++e) s[e] = (16 * t + 4 * g + e) < nk ? s[e] : -INFINITY;
            }
            m = fmaxf(fmaxf(m, fmaxf(s[0], s[1])), fmaxf(s[2], s[3]));
            st[t] = s;
        }
        st[NKT] = f32x4{0.f, 0.f, 0.f, 0.f};
        m = group_max4(m) * sc;   // sc > 0: the maximum commutes with the scale, which then rides in the exp2 FMA
        float l = 0.f;
#pragma unroll
        for (int t = 0; t < NKT; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float pe = __builtin_amdgcn_exp2f(fmaf(st[t][e], sc, -m));   // raw v_exp_f32: p < 2^-126 is 0 either way
                st[t][e] = pe;
                l += pe;
            }
        l = group_sum4(l);
        const float inv = 1.f / l;
        const int q = 16 * i + fr;
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt) {
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < (NKT + 1) / 2; ++ks) {
                const bf16x8 vt = tr_frag<HD>(vs, ks, dt, g, qq, p);
                o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vt, pack_pair(st[2 * ks], st[2 * ks + 1]), o, 0, 0, 0);
            }
            // the 16 x HD output tile leaves through a wave-private LDS scratch as ONE 16-byte store per lane
            // (16 token rows x HD * 2 bytes per instruction; 8-byte stores of 32-byte pieces cost ~10 % of the kernel)
            const bf16x4 ov = {(bf16)(o[0] * inv), (bf16)(o[1] * inv), (bf16)(o[2] * inv), (bf16)(o[3] * inv)};
            *(bf16x4*)(os + fr * C::ROW_BYTES + (dt * 16 + 4 * g) * 2) = ov;
        }
        asm volatile("" ::: "memory");
        {
            constexpr int CPR = C::ROW_BYTES / 16;            // 16-byte chunks per row: 4 (hd 32) or 8 (hd 64)
#pragma unroll
            for (int j = 0; j < 16 * CPR / 64; ++j) {
                const int c = lane + 64 * j;
                const int row = c / CPR, ch = c % CPR;
                const bf16x8 v = *(const bf16x8*)(os + row * C::ROW_BYTES + ch * 16);
                if (live && 16 * i + row < nq)
                    *(bf16x8*)((bf16*)a.o + b * a.o_bs + (int64_t)(16 * i + row) * a.o_ts + (int64_t)h * HD + ch * 8) = v;
            }
        }
        asm volatile("" ::: "memory");
        if (live && g == 0 && q < nq) a.lse[(b * a.heads + h) * a.nq + q] = (m + log2f(l)) * LN2;
    }
}

// ------------------------------------------------------------------------------------------------
// backward, Nq and Nk <= 80.  LDS per wave: K, Q, dO images (transposed-operand reads) + lse/delta.
//   phase T (per query tile i; keys on accumulator rows, queries on lanes):
//       S^T = K Q_i^T, dP^T = V dO_i^T, dS^T = P^T o (dP^T - delta) -> dQ_i^T = K^T dS^T  (tr-read K)
//   phase N (per key tile t; queries on accumulator rows, keys on lanes):
//       S = Q K_t^T, dP = dO V_t^T -> dV_t^T = dO^T P (tr-read dO), dK_t^T = Q^T dS (tr-read Q)
// ------------------------------------------------------------------------------------------------
// Gradient tiles of the workgroup's 4 heads leave through a workgroup-shared LDS region laid out like the destination
// rows ([row][4 heads x HD] bf16, 256 B per row at hd 32): one 16-byte store per lane covers 4 token rows x 256 B of full
// cache lines.  (8-byte stores of 32-byte row pieces made the stores 55 % of this kernel; per-head 64-byte rows 45 %.)
template <int HD, int NT>
__device__ __forceinline__ void store_rows4(const char* region, bf16* dst_head0, int64_t ts, int n, int live_heads, int tid) {
    constexpr int ROW = 4 * HD * 2, CPR = ROW / 16;   // bytes per staged row, 16-byte chunks per row
#pragma unroll
    for (int j = 0; j < NT * 16 * CPR / 256; ++j) {
        const int c = tid + 256 * j;
        const int row = c / CPR, ch = c % CPR;
        const bf16x8 v = *(const bf16x8*)(region + row * ROW + ch * 16);
        if (row < n && ch / (CPR / 4) < live_heads) *(bf16x8*)(dst_head0 + (int64_t)row * ts + ch * 8) = v;
    }
}

template <int HD, int NQT, int NKT> struct BwdSmallLds {
    static constexpr int MAXT = NQT > NKT ? NQT : NKT;
    static constexpr int IMG_TILES = (MAXT + 1) & ~1;                 // an odd count gets a zero phantom tile
    static constexpr int IMG_BYTES = IMG_TILES * 16 * HD * 2;
    static constexpr int STAT_FLOATS = IMG_TILES * 16;
    static constexpr int WAVE_BYTES = 2 * IMG_BYTES + 2 * STAT_FLOATS * 4;
    // workgroup layout: [slot 0 x 4 waves][slot 1 x 4 waves][stats x 4 waves]; the four slot-0 (or slot-1) images together
    // are also the workgroup-shared staging region of store_rows4 (IMG_TILES * 4 KB >= NT * 16 rows * 256 B)
    static constexpr int SLOT1_BASE = 4 * IMG_BYTES, STATS_BASE = 8 * IMG_BYTES;
};

template <int HD, int NQT, int NKT>
#ifndef ATTN_BWD_WAVES_PER_EU
#define ATTN_BWD_WAVES_PER_EU 3
#endif
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(ATTN_BWD_WAVES_PER_EU)))   // <= 168 VGPRs: 3 workgroups per CU (4 would spill)
attn_bwd_small_kernel(AttnArgs a) {
    using C = SmallCfg<HD>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using L = BwdSmallLds<HD, NQT, NKT>;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fr = lane & 15, g = lane >> 4, qq = fr >> 2, p = fr & 3;
    // head group fastest: the workgroups that write the other head slices of the same token rows run at the same time
    const int64_t b = blockIdx.y;
    const int hg = blockIdx.x;
    int h = hg * 4 + wave;
    const bool live = h < a.heads;
    h = live ? h : a.heads - 1;
    const int nq = (int)a.nq, nk = (int)a.nk;
    const int64_t hoff = (int64_t)h * HD;
    const bf16* qb = (const bf16*)a.q + b * a.q_bs + hoff;
    const bf16* kb = (const bf16*)a.k + b * a.k_bs + hoff;
    const bf16* vb = (const bf16*)a.v + b * a.v_bs + hoff;
    const bf16* dob = (const bf16*)a.d_o + b * a.o_bs + hoff;
    // wave-private LDS: image slot 0 holds K during phase T and Q during phase N, slot 1 holds dO.  Two slots
    // instead of three is what lets a third workgroup fit on a CU; everything is written and read by the same
    // wave (DS operations of one wave execute in order), so no workgroup barrier is needed anywhere.
    char* k_s = smem + wave * L::IMG_BYTES;
    char* q_s = k_s;
    char* do_s = smem + L::SLOT1_BASE + wave * L::IMG_BYTES;
    float* lse_s = (float*)(smem + L::STATS_BASE) + wave * 2 * L::STAT_FLOATS;
    float* del_s = lse_s + L::STAT_FLOATS;
    char* out0 = smem;                    // workgroup-shared staging regions (the four slot-0 / slot-1 images)
    char* out1 = smem + L::SLOT1_BASE;
    const int tid = threadIdx.x;
    const int live_heads = a.heads - hg * 4;   // >= 1; heads past it are clamped duplicates, never stored
    const int64_t hoff0 = (int64_t)hg * 4 * HD;
    constexpr int OROW = 4 * HD * 2;      // bytes per staged output row
    const float sc = a.scale * LOG2E;

    // every global load of the (batch, head) item is issued before the first use: one exposed memory latency per
    // wave instead of one per query tile (the staging / delta code below used to sit between the loads)
    bf16x8 kf[NKT][C::KCH], vf[NKT][C::KCH], qf[NQT][C::KCH], dof[NQT][C::KCH];
    float lse_r[NQT];
#pragma unroll
    for (int t = 0; t < NKT; ++t) {
        load_row_frags<HD>(kb, a.k_ts, 16 * t, nk, fr, g, kf[t]);
        load_row_frags<HD>(vb, a.v_ts, 16 * t, nk, fr, g, vf[t]);
    }
#pragma unroll
    for (int i = 0; i < NQT; ++i) {
        load_row_frags<HD>(qb, a.q_ts, 16 * i, nq, fr, g, qf[i]);
        load_row_frags<HD>(dob, a.o_ts, 16 * i, nq, fr, g, dof[i]);
        const int q = 16 * i + fr;
        lse_r[i] = a.lse[(b * a.heads + h) * a.nq + (q < nq ? q : nq - 1)];
    }
    asm volatile("" ::: "memory");   // keep the staging stores below the loads
#pragma unroll
    for (int t = 0; t < NKT; ++t) stage_tile<HD>(k_s, 16 * t, nk, fr, g, kf[t]);
#pragma unroll
    for (int i = 0; i < NQT; ++i) {
        const int q = 16 * i + fr;
        if (g == 0) lse_s[q] = lse_r[i] * LOG2E;   // lse in log2 units
    }
    {
        const bf16x8 z[C::KCH] = {};
        if (NKT & 1) stage_tile<HD>(k_s, 16 * NKT, 0, fr, g, z);
    }
    asm volatile("" ::: "memory");   // compiler-only fence (see the forward kernel)

    // ---- phase T: dQ -------------------------------------------------------------------------------
#pragma unroll
    for (int i = 0; i < NQT; ++i) {
        const int q = 16 * i + fr;
        const float lse2 = lse_s[q];
        f32x4 ds[NKT + 1], dpv[NKT];
        // delta[q] = sum_j P[q][j] dP[q][j], taken from the SAME fp32 P and dP the softmax backward uses (the whole key row
        // of a query is in this wave: 4 lane groups x NKT tiles x 4).  The textbook rowsum(dO o O) equals it only in exact
        // arithmetic: with O rounded to bf16 the mismatch is a per-row bias eps_q that dQ = sum_j P (dP - delta) K picks up
        // along the mean key direction - on near-uniform attention (cancellation-dominated dS) that bias was 16x the error
        // of PyTorch's bf16 autocast on d(cross_attn.q) (tests/diag_bf16_gradients.py).  It also saves reading O.
        float dl = 0.f;
#pragma unroll
        for (int t = 0; t < NKT; ++t) {
            f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < C::KCH; ++c) {
                s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[t][c], qf[i][c], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[t][c], dof[i][c], dp, 0, 0, 0);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float pe = __builtin_amdgcn_exp2f(fmaf(s[e], sc, -lse2));
                if (t == NKT - 1) pe = (16 * t + 4 * g + e) < nk ? pe : 0.f;   // only the last key tile can be ragged
                ds[t][e] = pe;
                dl = fmaf(pe, dp[e], dl);
            }
            dpv[t] = dp;
        }
        dl = group_sum4(dl);
        if (g == 0) del_s[q] = dl;          // phase N reads it back (same wave: DS operations execute in order)
#pragma unroll
        for (int t = 0; t < NKT; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) ds[t][e] *= dpv[t][e] - dl;
        ds[NKT] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < (NKT + 1) / 2; ++ks)
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag<HD>(k_s, ks, dt, g, qq, p),
                                                              pack_pair(ds[2 * ks], ds[2 * ks + 1]), acc, 0, 0, 0);
            // gradient tiles leave through LDS: 8-byte stores of 32-byte row pieces made the stores 55 % of the kernel
            const bf16x4 ov = {(bf16)(acc[0] * a.scale), (bf16)(acc[1] * a.scale), (bf16)(acc[2] * a.scale), (bf16)(acc[3] * a.scale)};
            *(bf16x4*)(out1 + q * OROW + wave * (HD * 2) + (dt * 16 + 4 * g) * 2) = ov;   // the slot-1 images are free until phase N
        }
    }
    __syncthreads();
    store_rows4<HD, NQT>(out1, (bf16*)a.dq + b * a.dq_bs + hoff0, a.dq_ts, nq, live_heads, tid);
    __syncthreads();   // staging region read out: the waves may overwrite their slot-1 images

    // ---- phase N: dK, dV ---------------------------------------------------------------------------
    asm volatile("" ::: "memory");   // every transposed read of the K image is issued before slot 0 is overwritten
#pragma unroll
    for (int i = 0; i < NQT; ++i) {
        stage_tile<HD>(q_s, 16 * i, nq, fr, g, qf[i]);     // slot 0: K image -> Q image
        stage_tile<HD>(do_s, 16 * i, nq, fr, g, dof[i]);   // slot 1: dQ staging -> dO image
    }
    if (NQT & 1) {
        const bf16x8 z[C::KCH] = {};
        stage_tile<HD>(q_s, 16 * NQT, 0, fr, g, z);
        stage_tile<HD>(do_s, 16 * NQT, 0, fr, g, z);
    }
    asm volatile("" ::: "memory");
    bf16x4 dvo[NKT][C::DT], dko[NKT][C::DT];
#pragma unroll
    for (int t = 0; t < NKT; ++t) {
        f32x4 pr[NQT + 1], ds[NQT + 1];
#pragma unroll
        for (int i = 0; i < NQT; ++i) {
            f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < C::KCH; ++c) {
                s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[i][c], kf[t][c], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dof[i][c], vf[t][c], dp, 0, 0, 0);
            }
            const f32x4 l4 = *(const f32x4*)(lse_s + 16 * i + 4 * g);
            const f32x4 d4 = *(const f32x4*)(del_s + 16 * i + 4 * g);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float pe = __builtin_amdgcn_exp2f(fmaf(s[e], sc, -l4[e]));
                if (i == NQT - 1) pe = (16 * i + 4 * g + e) < nq ? pe : 0.f;   // only the last query tile can be ragged
                pr[i][e] = pe;
                ds[i][e] = pe * (dp[e] - d4[e]);
            }
        }
        pr[NQT] = f32x4{0.f, 0.f, 0.f, 0.f};
        ds[NQT] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt) {
            f32x4 av = {0.f, 0.f, 0.f, 0.f}, ak = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int qs = 0; qs < (NQT + 1) / 2; ++qs) {
                av = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag<HD>(do_s, qs, dt, g, qq, p),
                                                             pack_pair(pr[2 * qs], pr[2 * qs + 1]), av, 0, 0, 0);
                ak = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag<HD>(q_s, qs, dt, g, qq, p),
                                                             pack_pair(ds[2 * qs], ds[2 * qs + 1]), ak, 0, 0, 0);
            }
            dvo[t][dt] = bf16x4{(bf16)av[0], (bf16)av[1], (bf16)av[2], (bf16)av[3]};
            dko[t][dt] = bf16x4{(bf16)(ak[0] * a.scale), (bf16)(ak[1] * a.scale), (bf16)(ak[2] * a.scale), (bf16)(ak[3] * a.scale)};
        }
        __builtin_amdgcn_sched_barrier(0);   // one key tile at a time: interleaving the unrolled tiles costs 60+ VGPRs
    }
    // every wave's images are dead after this barrier: dV rows go out through the slot-0 region, dK rows through slot 1
    __syncthreads();
#pragma unroll
    for (int t = 0; t < NKT; ++t)
#pragma unroll
        for (int dt = 0; dt < C::DT; ++dt) {
            *(bf16x4*)(out0 + (16 * t + fr) * OROW + wave * (HD * 2) + (dt * 16 + 4 * g) * 2) = dvo[t][dt];
            *(bf16x4*)(out1 + (16 * t + fr) * OROW + wave * (HD * 2) + (dt * 16 + 4 * g) * 2) = dko[t][dt];
        }
    __syncthreads();
    store_rows4<HD, NKT>(out0, (bf16*)a.dv + b * a.dv_bs + hoff0, a.dv_ts, nk, live_heads, tid);
    store_rows4<HD, NKT>(out1, (bf16*)a.dk + b * a.dk_bs + hoff0, a.dk_ts, nk, live_heads, tid);
}

static bool small_ok(const AttnArgs& a, bool backward) {
    if (a.nk > 16 * SM_MAX_TILES) return false;
    if (backward) return a.head_dim == 32 && a.nq <= 16 * SM_MAX_TILES;
    return true;
}

bool attention_mfma_supported(const AttnArgs& a, bool backward) {
    auto al = [](const void* p, int n) { return ((uintptr_t)p % n) == 0; };
    if (a.head_dim != 32 && a.head_dim != 64) return false;
    if (a.nq > (1 << 20) || a.nk > (1 << 20)) return false;
    if (a.q_ts % 8 || a.k_ts % 8 || a.v_ts % 8 || a.q_bs % 8 || a.k_bs % 8 || a.v_bs % 8 || a.o_ts % 8 || a.o_bs % 8) return false;
    if (!al(a.q, 16) || !al(a.k, 16) || !al(a.v, 16) || !al(a.o, 16)) return false;
    if (backward) {
        if (a.dq_ts % 8 || a.dk_ts % 8 || a.dv_ts % 8 || a.dq_bs % 8 || a.dk_bs % 8 || a.dv_bs % 8) return false;
        if (!al(a.d_o, 16) || !al(a.dq, 16) || !al(a.dk, 16) || !al(a.dv, 16)) return false;   // 16-byte gradient stores
    }
    return true;
}

template <int HD>
static int launch_fwd_small(const AttnArgs& a, hipStream_t s) {
    const int nkt = (int)((a.nk + 15) / 16);
    dim3 grid((unsigned)a.batch, (unsigned)((a.heads + 3) / 4));
    const size_t lds = 4 * SmallCfg<HD>::TILE_BYTES + 4 * 16 * SmallCfg<HD>::ROW_BYTES;
#define L(N) hipLaunchKernelGGL((attn_fwd_small_kernel<HD, N>), grid, dim3(256), lds, s, a)
    switch (nkt) {
        case 1: L(1); break;
        case 2: L(2); break;
        case 3: L(3); break;
        case 4: L(4); break;
        case 5: L(5); break;
        default: return VITED_ERR_UNSUPPORTED;
    }
#undef L
    return vited_check_launch();
}

int attention_fwd_mfma(const AttnArgs& a, hipStream_t s) {
    if (!small_ok(a, false)) return attention_fwd_flash(a, s);      // long sequences: tiled online softmax
    if (a.head_dim == 32) return launch_fwd_small<32>(a, s);
    if (a.head_dim == 64) return launch_fwd_small<64>(a, s);
    return VITED_ERR_UNSUPPORTED;
}

template <int NQT, int NKT>
static void launch_bwd_small_one(const AttnArgs& a, dim3 grid, hipStream_t s) {
    auto kernel = attn_bwd_small_kernel<32, NQT, NKT>;
    constexpr size_t lds = 4 * BwdSmallLds<32, NQT, NKT>::WAVE_BYTES;
    hipLaunchKernelGGL(kernel, grid, dim3(256), lds, s, a);
}

template <int NQT>
static int launch_bwd_small32(const AttnArgs& a, int nkt, hipStream_t s) {
    dim3 grid((unsigned)((a.heads + 3) / 4), (unsigned)a.batch);
#define L(N) launch_bwd_small_one<NQT, N>(a, grid, s)
    switch (nkt) {
        case 1: L(1); break;
        case 2: L(2); break;
        case 3: L(3); break;
        case 4: L(4); break;
        case 5: L(5); break;
        default: return VITED_ERR_UNSUPPORTED;
    }
#undef L
    return vited_check_launch();
}

int attention_bwd_mfma(const AttnArgs& a, hipStream_t s) {
    if (!small_ok(a, true)) return attention_bwd_flash(a, s);
    const int nqt = (int)((a.nq + 15) / 16), nkt = (int)((a.nk + 15) / 16);
    switch (nqt) {
        case 1: return launch_bwd_small32<1>(a, nkt, s);
        case 2: return launch_bwd_small32<2>(a, nkt, s);
        case 3: return launch_bwd_small32<3>(a, nkt, s);
        case 4: return launch_bwd_small32<4>(a, nkt, s);
        case 5: return launch_bwd_small32<5>(a, nkt, s);
        default: return VITED_ERR_UNSUPPORTED;
    }
}
