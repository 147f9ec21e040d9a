// Fused MLP branch of a ViT-ED block for gfx950 (CDNA4), bf16 MFMA:
//
//     y = x + fc2(gelu(fc1(LayerNorm(x))))            (vision_transformer.py:126 / :271 with timm Mlp, :115, :259)
//
// in ONE kernel for embed dim 384 / hidden 1536 (both shipped configs): the LayerNorm output, the 1536-wide hidden
// activation and the fc2 partial sums never round-trip through HBM.  What leaves the chip is y and - only when the
// caller will run a backward - the tensors the (unfused) backward kernels read: mean / rstd, h = LN(x) (bf16), u = gelu(z)
// and gd = gelu'(z) (bf16).  HBM bytes per token: 1,536 in + 1,536 out (+ 768 + 6,144 saved) against 15,360 for the
// LayerNorm -> fc1+GELU -> fc2+residual kernel sequence.
//
// Decomposition ("token-stationary, transposed"): a workgroup = 4 waves = 128 token rows, ONE workgroup per CU, each wave
// alone on its SIMD with the whole 512-entry register file and its own 32 tokens for the whole kernel:
//   * LN(x) of the wave's 32 tokens is computed in registers and kept there as 24 MFMA operand fragments (B operand of
//     z^T = W1 . h^T): the activation never touches LDS;
//   * the hidden dimension is walked in 12 chunks of 128: z^T[128 hidden][32 tokens] accumulates over K = 384
//     (192 v_mfma_f32_16x16x32_bf16 per wave), GELU runs on the accumulators, and - because the products are issued
//     transposed - the bf16-packed accumulators ARE the B operand of the second product
//     y^T[384][32 tokens] += W2[:, chunk] . u^T (another 192 MFMAs): the hidden activation never touches LDS either
//     (cdna_hip_programming.md section 3, "an accumulator tile as the next MFMA's operand": the k order inside a 32-deep step
//     is permuted, so the W2 fragment is read as two 8-byte halves in the same permuted order);
//   * only the WEIGHTS stream: 2.36 MB of bf16 W1 / W2 per workgroup (L2-resident: every workgroup reads the same bytes
//     in the same order) through a 3-slot ring of 32 KB LDS slots filled by LDS-DMA (global_load_lds_dwordx4, issued from
//     inline asm so that hipcc's waitcnt bookkeeping does not drain it), two slots in flight while one is read,
//     one raw s_barrier per slot.  Weight bytes through the CU per FLOP: 1/128 (the 128 x 128 tile GEMMs: 1/64).
// Per 32 KB slot a wave issues 64 MFMAs (1,024 matrix cycles) against 32 KB of LDS fragment reads.
#include <mutex>

#include "common.h"

#define MF_D 384
#define MF_H 1536
#define MF_BM 128
#define MF_PIECE 16384                 // [128 rows][64 bf16] image, 128-byte rows, 16-byte chunks XOR-swizzled by (row >> 1) & 7
#define MF_SUPER (2 * MF_PIECE)
#define MF_SLOTS 3
#define MF_SCR_LD 272                  // bytes per row of the wave-private transposition scratch (256 + 16: 2-way writes, aligned reads)
#define MF_SCR_BYTES (32 * MF_SCR_LD)
#define MF_RING_BYTES (MF_SLOTS * MF_SUPER)
#define MF_B1_OFF (MF_RING_BYTES + 4 * MF_SCR_BYTES)
#define MF_B2_OFF (MF_B1_OFF + MF_H * 4)
#define MF_LDS_BYTES (MF_B2_OFF + MF_D * 4)
#define MF_NSUPER 72                   // 12 chunks x (3 W1 + 3 W2) super-pieces

struct MlpArgs {
    const float* x; int64_t ldx;
    const float* gamma; const float* beta;
    const bf16* w1; const float* b1;   // [1536][384], [1536]
    const bf16* w2; const float* b2;   // [384][1536], [384]
    float* y; int64_t ldy;
    bf16* h; bf16* gd; bf16* u;        // saved for backward: [M][384], [M][1536], [M][1536] (dense rows)
    float* mean; float* rstd;
    int64_t M; float eps;
};

// one 1-KiB LDS-DMA: lane l's 16 bytes (at base + voff) land at lds_dst + 16 l.  Scalar base + 32-bit lane offset: the lane offsets are
// loop-invariant (8 VGPRs for the whole kernel).  M0 is compiler-reserved: written and restored in the same statement.
__device__ __forceinline__ void mf_dma16(const void* base, unsigned voff, unsigned lds_dst) {
#ifdef MF_ABL_NODMA
    return;
#endif
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(base), "s"(lds_dst) : "memory");
}

// super-piece q = 6 c + idx: idx 0..2 = W1 rows [128 c, +128) x k [128 idx, +128); idx 3..5 = W2 rows [128 (idx-3), +128) x k [128 c, +128)
// off1 / off2: this lane's byte offsets of its 4 rows (wave * 32 + 8 i + lane / 8, swizzled 16-byte chunk) in a W1 / W2 piece
__device__ __forceinline__ void mf_issue(const MlpArgs& a, int c, int idx, int slot, int wave, const unsigned (&off1)[4], const unsigned (&off2)[4]) {
    const unsigned dst0 = (unsigned)slot * MF_SUPER;
    const bool first = idx < 3;
    const bf16* base = first ? a.w1 + (int64_t)c * 128 * MF_D + idx * 128 : a.w2 + (int64_t)(idx - 3) * 128 * MF_H + c * 128;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
            mf_dma16(base + j * 64, first ? off1[i] : off2[i], dst0 + j * MF_PIECE + (wave * 32 + i * 8) * 128);
}

// own quarter of the slot landed (at most the 8 DMAs of the next super-piece stay in flight), every LDS read returned,
// then the workgroup rendezvous: all quarters landed, everybody is done with the previous slot
#define MF_SYNC(VM) asm volatile("s_waitcnt vmcnt(" #VM ") lgkmcnt(0)\n\ts_barrier" ::: "memory")

template <bool SAVE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
mlp_fwd_fused_kernel(MlpArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int64_t m0w = (int64_t)blockIdx.x * MF_BM + wave * 32;
    char* scr = smem + MF_RING_BYTES + wave * MF_SCR_BYTES;
    float* b1s = (float*)(smem + MF_B1_OFF);
    float* b2s = (float*)(smem + MF_B2_OFF);

    // ---- weights start streaming before anything else ------------------------------------------------------------
    unsigned off1[4], off2[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = wave * 32 + i * 8 + (lane >> 3);
        const int cs = (lane & 7) ^ ((r >> 1) & 7);
        off1[i] = (unsigned)(r * MF_D + cs * 8) * 2;
        off2[i] = (unsigned)(r * MF_H + cs * 8) * 2;
    }
    mf_issue(a, 0, 0, 0, wave, off1, off2);
    mf_issue(a, 0, 1, 1, wave, off1, off2);
    for (int i = threadIdx.x; i < MF_H / 4; i += 256) ((f32x4*)b1s)[i] = ((const f32x4*)a.b1)[i];
    if (threadIdx.x < MF_D / 4) ((f32x4*)b2s)[threadIdx.x] = ((const f32x4*)a.b2)[threadIdx.x];

    // ---- LayerNorm of the wave's 32 tokens, in registers; lane (fr, fq) holds k = 32 ks + 8 fq .. + 8 of token fr -----
    bf16x8 hfrag[2][12];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        // rows past M are clamped to row M - 1: such a lane computes exactly row M - 1's values and every store below writes them to
        // row M - 1 again (a benign duplicate), so no store needs a predicate (predicated stores cost exec-mask branches and spills)
        int64_t r = m0w + tt * 16 + fr;
        r = r < a.M ? r : a.M - 1;
        const float* xr = a.x + r * a.ldx + 8 * fq;
        f32x4 v[12][2];
        float s = 0.f;
#pragma unroll
        for (int ks = 0; ks < 12; ++ks) {
            v[ks][0] = *(const f32x4*)(xr + 32 * ks);
            v[ks][1] = *(const f32x4*)(xr + 32 * ks + 4);
        }
#pragma unroll
        for (int ks = 0; ks < 12; ++ks)
            s += ((v[ks][0][0] + v[ks][0][1]) + (v[ks][0][2] + v[ks][0][3])) + ((v[ks][1][0] + v[ks][1][1]) + (v[ks][1][2] + v[ks][1][3]));
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        const float mu = s * (1.0f / MF_D);
        float q = 0.f;
#pragma unroll
        for (int ks = 0; ks < 12; ++ks)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float d = v[ks][hh][e] - mu;
                    q = fmaf(d, d, q);
                }
        q += __shfl_xor(q, 16, 64);
        q += __shfl_xor(q, 32, 64);
        const float rs = rsqrtf(q * (1.0f / MF_D) + a.eps);
#pragma unroll
        for (int ks = 0; ks < 12; ++ks) {
            const f32x4 g0 = *(const f32x4*)(a.gamma + 32 * ks + 8 * fq), g1 = *(const f32x4*)(a.gamma + 32 * ks + 8 * fq + 4);
            const f32x4 be0 = *(const f32x4*)(a.beta + 32 * ks + 8 * fq), be1 = *(const f32x4*)(a.beta + 32 * ks + 8 * fq + 4);
            bf16x8 hv;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                hv[e] = (bf16)((v[ks][0][e] - mu) * rs * g0[e] + be0[e]);
                hv[4 + e] = (bf16)((v[ks][1][e] - mu) * rs * g1[e] + be1[e]);
            }
            hfrag[tt][ks] = hv;
            if (SAVE) *(bf16x8*)(a.h + r * MF_D + 32 * ks + 8 * fq) = hv;
        }
        if (SAVE && fq == 0) {
            a.mean[r] = mu;
            a.rstd[r] = rs;
        }
    }

    f32x4 outT[24][2];
#pragma unroll
    for (int i = 0; i < 24; ++i) {
        outT[i][0] = f32x4{0.f, 0.f, 0.f, 0.f};
        outT[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // Lane parts of the fragment addresses inside a piece.  fc1's hidden rows are taken in a PERMUTED order inside every group of
    // 32: tile 2 s holds hidden 32 s + 8 (i >> 2) + (i & 3) in its row i, tile 2 s + 1 the same + 4.  A lane's accumulators of the
    // tile pair are then 8 CONSECUTIVE hidden units (32 s + 8 fq .. + 7) = the k order of one 32-deep step of fc2, so the packed
    // accumulators are fc2's B operand AND the matching W2 fragment is 16 contiguous bytes of a W2 row: one ds_read_b128 (with
    // fc1's rows in natural order it was two 8-byte halves 32 bytes apart, which hipcc fused across fragments into
    // ds_read2st64_b64 + 4 v_mov per fragment: 275 moves per 128-wide chunk).
    const int sw = (fr >> 1) & 7;
    // k-step 1 of a piece is chunk 4 + fq instead of fq: the same address with bit 6 flipped (the swizzle is an XOR below bit 7), so only
    // the k-step-0 addresses are kept in registers (the kernel sits at the 512-register limit)
    const int a2_k0 = fr * 128 + ((fq ^ sw) << 4);            // W2 fragment (row fr of a 16-row group), k-step 0 of the piece (chunk fq)
    int a1_k0[2];                                             // W1 fragment of an even / odd tile: permuted row
#pragma unroll
    for (int par = 0; par < 2; ++par) {
        const int row = 8 * (fr >> 2) + 4 * par + (fr & 3);
        const int swr = (row >> 1) & 7;                       // the 32-row group offset is a multiple of 16: it does not enter the swizzle
        a1_k0[par] = row * 128 + ((fq ^ swr) << 4);
    }

    for (int c = 0; c < 12; ++c) {
        f32x4 zT[8][2];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            zT[i][0] = f32x4{0.f, 0.f, 0.f, 0.f};
            zT[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        // ---- z^T[128 hidden][32 tokens] = W1[chunk] . h^T : three 32 KB slots of W1 -------------------------------
#pragma unroll
        for (int idx = 0; idx < 3; ++idx) {
            MF_SYNC(8);
            {   // refill the slot everybody just left with the super-piece two ahead
                const int n = idx + 2;
                mf_issue(a, c, n, n % 3, wave, off1, off2);               // n = 2, 3, 4: same chunk; slot of q = 6 c + n is n % 3
            }
            const char* sl = smem + (idx % 3) * MF_SUPER;
            // four groups of 8 fragments (one 32-deep k-step each), double-buffered in registers: the reads of group g + 1 are in
            // flight under the 16 MFMAs of group g (left to itself hipcc reuses ONE fragment register and exposes the LDS
            // latency in front of every pair of MFMAs: 6x slower)
            bf16x8 wa[8], wb[8];
#define MF_LOAD1(BUF, G) _Pragma("unroll") for (int ht = 0; ht < 8; ++ht) \
                BUF[ht] = *(const bf16x8*)(sl + ((G) >> 1) * MF_PIECE + (a1_k0[ht & 1] ^ (((G) & 1) << 6)) + (ht >> 1) * 4096)
#ifdef MF_ABL_NOMFMA
#define MF_MMA1(BUF, G) _Pragma("unroll") for (int ht = 0; ht < 8; ++ht) { asm volatile("" :: "v"(BUF[ht]), "v"(hfrag[0][4 * idx + (G)]), "v"(hfrag[1][4 * idx + (G)])); }
#else
#define MF_MMA1(BUF, G) _Pragma("unroll") for (int ht = 0; ht < 8; ++ht) { \
                zT[ht][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BUF[ht], hfrag[0][4 * idx + (G)], zT[ht][0], 0, 0, 0); \
                zT[ht][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BUF[ht], hfrag[1][4 * idx + (G)], zT[ht][1], 0, 0, 0); }
#endif
            MF_LOAD1(wa, 0);
            MF_LOAD1(wb, 1);
            __builtin_amdgcn_sched_barrier(0);
            MF_MMA1(wa, 0);
            __builtin_amdgcn_sched_barrier(0);
            MF_LOAD1(wa, 2);
            __builtin_amdgcn_sched_barrier(0);
            MF_MMA1(wb, 1);
            __builtin_amdgcn_sched_barrier(0);
            MF_LOAD1(wb, 3);
            __builtin_amdgcn_sched_barrier(0);
            MF_MMA1(wa, 2);
            __builtin_amdgcn_sched_barrier(0);
            MF_MMA1(wb, 3);
#undef MF_LOAD1
#undef MF_MMA1
        }
        // ---- bias, GELU and its derivative on the accumulators; pack u^T as the B operand of the second product --------
        // lane (fr, fq) holds hidden 32 (ht >> 1) + 8 fq + 4 (ht & 1) + e (e = 0..3) of tokens fr, 16 + fr (permuted rows, see above)
        bf16x8 ufrag[4][2];
#pragma unroll
        for (int ht = 0; ht < 8; ++ht) {
            const int hcol = 32 * (ht >> 1) + 8 * fq + 4 * (ht & 1);      // this lane's first hidden unit of the tile, inside the chunk
            const f32x4 bb = *(const f32x4*)(b1s + c * 128 + hcol);
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                bf16x4 uu, gg;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float z = zT[ht][tt][e] + bb[e];
#ifdef MF_ABL_NOGELU
                    uu[e] = (bf16)z;
                    gg[e] = (bf16)z;
#else
                    float cdf, ex;
                    gelu_parts_fast(z, cdf, ex);
                    uu[e] = (bf16)(z * cdf);
                    gg[e] = (bf16)fmaf(z * 0.39894228040143268f, ex, cdf);
#endif
                }
                // gd rows leave through the wave-private scratch as whole 256-byte row segments (staged right away: nothing is held)
                if (SAVE) *(bf16x4*)(scr + (tt * 16 + fr) * MF_SCR_LD + hcol * 2) = gg;
                // k order of fc2's 32-deep step s = ht / 2: element j < 4 <- tile 2 s, j >= 4 <- tile 2 s + 1: hidden 32 s + 8 fq + j
                if (ht & 1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) ufrag[ht >> 1][tt][4 + e] = uu[e];
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) ufrag[ht >> 1][tt][e] = uu[e];
                }
            }
            __builtin_amdgcn_sched_barrier(0);   // one hidden tile at a time: interleaving the unrolled tiles costs registers (spills)
        }
        if (SAVE) {
#pragma unroll
            for (int which = 0; which < 2; ++which) {
                if (which == 1) {
#pragma unroll
                    for (int ht = 0; ht < 8; ++ht)
#pragma unroll
                        for (int tt = 0; tt < 2; ++tt) {
                            const bf16x8 uf = ufrag[ht >> 1][tt];
                            const bf16x4 val = (ht & 1) ? bf16x4{uf[4], uf[5], uf[6], uf[7]} : bf16x4{uf[0], uf[1], uf[2], uf[3]};
                            *(bf16x4*)(scr + (tt * 16 + fr) * MF_SCR_LD + (32 * (ht >> 1) + 8 * fq + 4 * (ht & 1)) * 2) = val;
                        }
                }
                asm volatile("" ::: "memory");   // wave-private LDS, a wave's DS operations execute in order: compiler fence only
                bf16* dst = (which == 0 ? a.gd : a.u) + c * 128 + fr * 8;
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const bf16x8 val = *(const bf16x8*)(scr + (it * 4 + fq) * MF_SCR_LD + fr * 16);
                    int64_t r = m0w + it * 4 + fq;
                    r = r < a.M ? r : a.M - 1;
                    *(bf16x8*)(dst + r * MF_H) = val;
                }
                asm volatile("" ::: "memory");
            }
        }
        // ---- y^T[384][32 tokens] += W2[:, chunk] . u^T : three 32 KB slots of W2 (128 output rows each) -----------------
#pragma unroll
        for (int idx = 3; idx < 6; ++idx) {
            const int q = 6 * c + idx;
            if (q == MF_NSUPER - 1) MF_SYNC(0); else MF_SYNC(8);
            if (q + 2 < MF_NSUPER) {
                const int n = idx + 2;                               // 5 -> same chunk; 6, 7 -> idx 0, 1 of the next chunk
                mf_issue(a, n >= 6 ? c + 1 : c, n >= 6 ? n - 6 : n, n % 3, wave, off1, off2);
            }
            const char* sl = smem + (idx % 3) * MF_SUPER;
            bf16x8 wa[8], wb[8];
#define MF_LOAD2(BUF, G) _Pragma("unroll") for (int nt = 0; nt < 8; ++nt) \
                BUF[nt] = *(const bf16x8*)(sl + ((G) >> 1) * MF_PIECE + (a2_k0 ^ (((G) & 1) << 6)) + nt * 2048)
#ifdef MF_ABL_NOMFMA
#define MF_MMA2(BUF, G) _Pragma("unroll") for (int nt = 0; nt < 8; ++nt) { asm volatile("" :: "v"(BUF[nt]), "v"(ufrag[G][0]), "v"(ufrag[G][1])); }
#else
#define MF_MMA2(BUF, G) _Pragma("unroll") for (int nt = 0; nt < 8; ++nt) { \
                outT[(idx - 3) * 8 + nt][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BUF[nt], ufrag[G][0], outT[(idx - 3) * 8 + nt][0], 0, 0, 0); \
                outT[(idx - 3) * 8 + nt][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BUF[nt], ufrag[G][1], outT[(idx - 3) * 8 + nt][1], 0, 0, 0); }
#endif
            MF_LOAD2(wa, 0);
            MF_LOAD2(wb, 1);
            __builtin_amdgcn_sched_barrier(0);
            MF_MMA2(wa, 0);
            __builtin_amdgcn_sched_barrier(0);
            MF_LOAD2(wa, 2);
            __builtin_amdgcn_sched_barrier(0);
            MF_MMA2(wb, 1);
            __builtin_amdgcn_sched_barrier(0);
            MF_LOAD2(wb, 3);
            __builtin_amdgcn_sched_barrier(0);
            MF_MMA2(wa, 2);
            __builtin_amdgcn_sched_barrier(0);
            MF_MMA2(wb, 3);
#undef MF_LOAD2
#undef MF_MMA2
        }
    }

    // ---- epilogue: y = x + acc + b2, 64 columns at a time through the scratch (whole 256-byte fp32 row segments) ---------
#pragma unroll
    for (int grp = 0; grp < 6; ++grp) {
#pragma unroll
        for (int n4 = 0; n4 < 4; ++n4)
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
                *(f32x4*)(scr + (tt * 16 + fr) * MF_SCR_LD + (n4 * 16 + 4 * fq) * 4) = outT[grp * 4 + n4][tt];
        asm volatile("" ::: "memory");
        const f32x4 bb = *(const f32x4*)(b2s + grp * 64 + fr * 4);
        f32x4 xin[8];
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            int64_t r = m0w + it * 4 + fq;
            r = r < a.M ? r : a.M - 1;
            xin[it] = *(const f32x4*)(a.x + r * a.ldx + grp * 64 + fr * 4);
        }
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int row = it * 4 + fq;
            const f32x4 acc = *(const f32x4*)(scr + row * MF_SCR_LD + fr * 16);
            int64_t r = m0w + row;
            r = r < a.M ? r : a.M - 1;
            *(f32x4*)(a.y + r * a.ldy + grp * 64 + fr * 4) = f32x4{xin[it][0] + acc[0] + bb[0], xin[it][1] + acc[1] + bb[1],
                                                                                 xin[it][2] + acc[2] + bb[2], xin[it][3] + acc[3] + bb[3]};
        }
        asm volatile("" ::: "memory");
    }
}

extern "C" int vited_mlp_fwd(const float* x, int64_t ldx, const float* gamma, const float* beta, const void* w1, const float* b1,
                             const void* w2, const float* b2, float* y, int64_t ldy, void* h, void* gd, void* u, float* mean,
                             float* rstd, int64_t rows, int64_t dim, int64_t hidden, float eps, void* stream) {
    if (!x || !gamma || !beta || !w1 || !b1 || !w2 || !b2 || !y || rows <= 0) return VITED_ERR_BAD_ARG;
    if (dim != MF_D || hidden != MF_H) return VITED_ERR_UNSUPPORTED;
    if (ldx < dim || ldy < dim || (ldx & 3) || (ldy & 3)) return VITED_ERR_BAD_ARG;
    const bool save = h || gd || u || mean || rstd;
    if (save && !(h && gd && u && mean && rstd)) return VITED_ERR_BAD_ARG;
    const uintptr_t al = (uintptr_t)x | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)w1 | (uintptr_t)b1 | (uintptr_t)w2 | (uintptr_t)b2 |
                         (uintptr_t)y | (uintptr_t)h | (uintptr_t)gd | (uintptr_t)u;
    if (al & 15) return VITED_ERR_BAD_ARG;
    if (rows > ((int64_t)1 << 30)) return VITED_ERR_UNSUPPORTED;
    MlpArgs a;
    a.x = x; a.ldx = ldx; a.gamma = gamma; a.beta = beta;
    a.w1 = (const bf16*)w1; a.b1 = b1; a.w2 = (const bf16*)w2; a.b2 = b2;
    a.y = y; a.ldy = ldy; a.h = (bf16*)h; a.gd = (bf16*)gd; a.u = (bf16*)u; a.mean = mean; a.rstd = rstd;
    a.M = rows; a.eps = eps;
    static std::once_flag once;      // dynamic LDS above 64 KB needs the opt-in once per kernel (forward thread or autograd thread)
    static bool attr_ok = false;
    std::call_once(once, [&] {
        attr_ok = hipFuncSetAttribute((const void*)mlp_fwd_fused_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, MF_LDS_BYTES) == hipSuccess &&
                  hipFuncSetAttribute((const void*)mlp_fwd_fused_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, MF_LDS_BYTES) == hipSuccess;
    });
    if (!attr_ok) return VITED_ERR_LAUNCH;
    const unsigned grid = (unsigned)ceil_div64(rows, MF_BM);
    if (save) hipLaunchKernelGGL(mlp_fwd_fused_kernel<true>, dim3(grid), dim3(256), MF_LDS_BYTES, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(mlp_fwd_fused_kernel<false>, dim3(grid), dim3(256), MF_LDS_BYTES, (hipStream_t)stream, a);
    return vited_check_launch();
}
