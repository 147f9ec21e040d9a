// Row-complete bf16 MFMA GEMM for the 384-wide residual stream, with LayerNorm fused into the epilogue.
//
//   MODE_FWD  y = residual + A[M,K] . W[384,K]^T + bias   (fp32)        - proj / cross-proj / fc2 + residual
//             h = LayerNorm(y; gamma, beta, eps) (bf16), mean, rstd       - the NEXT sub-block's norm (models/vision_transformer.py:
//                                                                           124-127, 268-272: x = x + f(norm(x)) chains)
//   MODE_BWD  dh = dY[M,K] . Wt[384,K]^T                                  - dX of a Linear that consumed a LayerNorm's output
//             dx = dx_in + LN'(dh; x, mean, rstd, gamma) (fp32, + bf16 copy), dgamma / dbeta partial column sums
//
// Why: LayerNorm needs whole rows, and the 128 x 128 tile kernels (gemm_mfma.hip) split the 384 columns over three workgroups,
// so every LayerNorm was its own streaming pass over the fp32 residual stream (196 launches, 4.9 ms of a 30 ms step, round 2).
// Here one workgroup owns BM complete rows.
//
// ONE 8-wave workgroup per CU; the W panel is staged once per BM rows (1/96 B per FLOP at BM = 128 against 1/64 for a 128 x 128 tile).
// Main loop (third form; the schedule is described where it is written, below): K = 64 tiles staged as full 128-byte lines into
// three A buffers + two W buffers, MFMA clusters and load slots separated by raw s_barriers with the two wave groups one slot
// apart, counted vmcnt.  History, all measured at K = 1536, M = 65,536 (profiles/row_probe.py):
//   1. 64 / 80-row tiles, two 4-wave workgroups per CU, a two-stage image: 500 TF/s in the K loop - slower than the two kernels
//      it replaced for K >= 768;
//   2. 128 / 144-row tiles, a 4-slot ring of K = 32 stages (16 rows x 64 B LDS-DMA pieces), one barrier per step, fragment
//      double buffer, waves w / w + 4 in opposite DMA / MFMA order: 820 TF/s in the K loop, kernel 138 us;
//   3. this one: 1.0 - 1.1 PF/s in the K loop, kernel 117 - 125 us.
// Waves sit side by side: wave w owns all BM rows x columns 48 w .. 48 w + 47 (MT x 3 accumulators of v_mfma_f32_16x16x32_bf16).
//
// The product is issued transposed (W fragment as the A operand): a lane then holds 4 CONSECUTIVE columns of one row and the
// tile goes to an fp32 LDS scratch (48 - 80 rows x 388 floats per pass, reusing the staging buffers) as 16-byte stores; from there the
// epilogue is the LayerNorm kernels' own row loop (layernorm.hip): a 32-lane half-wave owns a row, 3 float4 per lane, all global
// traffic in full lines, row statistics by DPP row sums + one swizzle, the pass's row operands requested before the dump.  dh never reaches HBM and is never rounded to bf16.
//
// BM = 96 / 128 / 144 / 160 (MT = 6 / 8 / 9 / 10), chosen per M so that the tiles fill the chip's 256 CUs with the smallest makespan
// (row_mt below: 65,536 rows = 512 x 128; 66,560 = 463 x 144; 73,800 = 462 x 160; 24,576 = 256 x 96).
#include <mutex>
#include <type_traits>

#include "gemm_kernels.h"
#include "gemm_lds.h"

#define ROW_N 384
#define ROW_LD 388      // floats per scratch row: 16-byte aligned rows; 388 mod 32 = 4 spreads the 8-lane groups of ds_write_b128 over the banks

enum { ROW_MODE_FWD = 0, ROW_MODE_BWD = 1 };

struct RowArgs {
    const bf16* A; int64_t lda;
    int seg_k; int64_t seg_stride;     // the contraction dim may be cut into segments of seg_k columns, segment j starting at A + j * seg_stride
                                       // (one [M, seg_k] tensor per decoder block: vited_linear_layernorm_bwd_segmented); seg_k = K: one tensor
    const bf16* W; int64_t ldw;
    int64_t M; int K;
    // forward
    const float* bias; const float* residual; int64_t ldr;
    float* y; int64_t ldy;
    const float* gamma; const float* beta; float eps;
    bf16* h; int64_t ldh; float* mean; float* rstd;
    // backward
    const float* x; int64_t ldx; const float* mean_in; const float* rstd_in;
    const float* dx_in; int64_t ldxi; float* dx; int64_t lddx; bf16* dx_lp; int64_t ldlp;
    float* partial;     // [tiles][2][384]
};

template <int MT> struct RowCfg {
    static_assert(MT == 6 || MT == 8 || MT == 9 || MT == 10, "tile heights: 96, 128, 144 or 160 rows");
    static constexpr int BM = 16 * MT;
    static constexpr int A_BYTES = BM * 128;                             // one K = 64 tile of the A panel: 12 - 20 KB
    static constexpr int W_BYTES = ROW_N * 128;                          // ... of the whole W panel: 48 KB
    static constexpr int W_BASE = 3 * A_BYTES;
    static constexpr int LDS_BYTES = 3 * A_BYTES + 2 * W_BYTES;          // 135,168 ... 159,744
    static constexpr int P0 = (MT + 1) / 2;                              // row tiles of a K-tile's first phase
    static constexpr int A_PIECES = 2 * MT;                              // 8-row LDS-DMA pieces of an A tile: group 0's wave u takes u, u + 4, ...
    static constexpr int A_J = (A_PIECES + 3) / 4;                       // ... at most this many
    static constexpr int PASS_SUB = MT == 8 ? 4 : MT == 10 ? 5 : 3;      // 16-row sub-tiles per epilogue pass
    static constexpr int PASS_ROWS = 16 * PASS_SUB;                      // 48 / 64 / 80
    static constexpr int NPASS = MT / PASS_SUB;                          // 2 / 3
    static constexpr int ROWS_PER_HALF = PASS_ROWS / 16;                 // rows of a pass per half-wave (16 half-waves)
    // all of a pass's row operands in flight at once where the registers allow it (the backward row carries 26 values per lane)
    static constexpr bool EPI_ALL_ROWS = MT <= 9;
    static_assert(MT % PASS_SUB == 0, "passes cover the tile");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS");
    static_assert(LDS_BYTES >= PASS_ROWS * ROW_LD * 4, "epilogue scratch must fit the staging buffers");
    static_assert(LDS_BYTES >= 16 * 2 * ROW_N * 4, "column-partial combine must fit the staging buffers");
};

__device__ __forceinline__ float row_half_sum(float v) { return half_wave_sum(v); }   // over the 32 lanes of a half-wave (common.h)

// outputs are written once and read by a later kernel: streaming stores (see gemm_nt_epilogue.h for the measurements)
#define ROW_STORE(ptr, val) __builtin_nontemporal_store(val, ptr)
// ... except the bf16 outputs (h, dx_lp): the very next kernel's operand, a normal store (step 25.38 -> 25.19 ms)
#define ROW_STORE_LP(ptr, val) (*(ptr) = (val))
// ... and its row operands (residual, x, incoming gradient) are read exactly once: streaming loads (step 26.09 -> 25.9 ms)
#define ROW_LOAD(ptr) __builtin_nontemporal_load(ptr)

struct RowFwdIn { f32x4 res[3]; };
struct RowBwdIn { f32x4 x[3], din[3]; float mu, rs; };

template <int MODE, int MT>
__global__ void __launch_bounds__(512)
gemm_row_kernel(const RowArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using R = RowCfg<MT>;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t m0 = (int64_t)blockIdx.x * R::BM;

    f32x4 acc[MT][3];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- main loop: K = 64 tiles, phases gated by workgroup barriers, the two wave groups one slot apart -------------------------
    // LDS: three A buffers (BM rows x 128 B, tile t in buffer t % 3) + two W buffers (384 rows x 128 B, tile t in buffer t & 1).
    // Every LDS-DMA piece is 8 rows x 128 B: FULL cache lines (16 rows x 64 B pieces cost the vector-memory path twice the line
    // look-ups for the same bytes: staging alone ran 1.6 x faster with full lines).  Lane (r8, c8) lands at
    // row r8, 16-byte slot c8 of its piece and fetches source chunk c8 ^ swz(row), swz(row) = (row >> 1) & 7 = 4 (piece & 1) + (r8 >> 1).
    //
    // A wave's K-tile is two PHASES (row tiles 0 .. P0-1, then P0 .. MT-1; 3 column tiles x 2 k-halves each), and each phase is two
    // SLOTS separated by workgroup barriers: L = fragment reads (ds_read_b128) + this slot's share of the LDS-DMA, M = the MFMAs.
    // Group 1 (waves 4-7) runs one slot behind group 0 (waves 0-3), and waves w / w + 4 share a SIMD: in every slot each SIMD has
    // one wave in its MFMA cluster and one wave loading (the 8-phase schedule of the CDNA4 guide, section 5, on a 128 x 384 tile).
    // Global slot s = 4 t + k for group 0, one later for group 1.  Issue schedule (16 pieces = 4 per wave of the issuing group per
    // L slot; W tile = parts 0..2 of 128 rows):
    //     slot 4t   (g0): W(t+1) part 1      slot 4t+1 (g1): W(t+1) part 2      slot 4t+2 (g0): A(t+2)      slot 4t+3 (g1): W(t+2) part 0
    // WAR: W buffer t & 1 was last read in slot 4t+1 (g1's phase 0; retired by its lgkmcnt(0) in slot 4t+2, whose barrier certifies
    // it) -> rewritten from slot 4t+3; A buffer (t+2) % 3 = (t-1) % 3 was last read in slot 4t-1 -> rewritten in slot 4t+2.
    // RAW: every wave waits (counted vmcnt) for its own pieces of tile t+1 before the barrier that ends slot 4t+3; group 0 reads
    // tile t+1 in slot 4t+4.  The youngest part of W(t+1) then had two slots to land (it is L2-resident), A(t+1) six.
    const int grp = wave >> 2, u = wave & 3;
    const int nt = a.K / 64;
    const int r8 = lane >> 3, c8 = lane & 7;
    const int cs_even = (c8 ^ (r8 >> 1)) * 8, cs_odd = (c8 ^ (r8 >> 1) ^ 4) * 8;
    // group 0 stages A: wave u takes pieces u, u + 4, u + 8, ... of the tile's 2 MT (piece parity = u & 1); W parts likewise
    const int cs_u = (u & 1) ? cs_odd : cs_even;
    const int a_per = (R::A_PIECES - u + 3) / 4;          // 3, 4 or 5 pieces for this wave (wave-uniform)
    const bf16* pa[R::A_J];
#pragma unroll
    for (int j = 0; j < R::A_J; ++j) {
        int64_t ar = m0 + 8 * (4 * j + u) + r8;
#ifdef ROW_DBG_A_RESIDENT                    // timing experiment: every workgroup stages the FIRST tile's rows (A served by L2, not HBM)
        ar -= m0;
#endif
        ar = ar < a.M ? ar : a.M - 1;        // rows past M are staged from the last row and never stored
        pa[j] = a.A + ar * a.lda + cs_u;
    }
    const bf16* pw = a.W + (int64_t)(8 * u + r8) * a.ldw + cs_u;
    auto issue_a = [&](int t) {
#ifdef ROW_DBG_NO_DMA
        return;
#endif
        if (t >= nt) return;
        char* dst = smem + (t % 3) * R::A_BYTES;
        const int k0 = t * 64;
        const int seg = k0 / a.seg_k;                                         // scalar: which [M, seg_k] tensor this K-tile reads
        const int64_t ka = (int64_t)seg * a.seg_stride + (k0 - seg * a.seg_k);
#pragma unroll
        for (int j = 0; j < R::A_J; ++j)
            if (4 * j + 3 < R::A_PIECES || j < a_per) glds16_asm(pa[j] + ka, dst + (4 * j + u) * 1024);
    };
    auto issue_w = [&](int t, int part) {
#ifdef ROW_DBG_NO_DMA
        return;
#endif
        if (t >= nt) return;
        char* dst = smem + R::W_BASE + (t & 1) * R::W_BYTES + (16 * part + u) * 1024;
        const int64_t off = (int64_t)(128 * part) * a.ldw + t * 64;
#pragma unroll
        for (int j = 0; j < 4; ++j) glds16_asm(pw + off + (int64_t)(32 * j) * a.ldw, dst + 4 * j * 1024);
    };
    // fragments: row (16 i + fr) of A / row (48 wave + 16 j + fr) of W, k-half kk -> 16-byte chunk 4 kk + fq; swz(row) = (fr >> 1) & 7
    const int fr = lane & 15, fq = lane >> 4;
    const int loff0 = fr * 128 + ((fq ^ ((fr >> 1) & 7)) << 4), loff1 = fr * 128 + (((4 + fq) ^ ((fr >> 1) & 7)) << 4);
    bf16x8 fa[R::P0][2] = {}, fb[3][2] = {};
    auto read_b = [&](int t) {
#ifdef ROW_DBG_NO_READS
        return;
#endif
        const char* wb = smem + R::W_BASE + (t & 1) * R::W_BYTES + wave * 48 * 128;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            fb[j][0] = *(const bf16x8*)(wb + j * 2048 + loff0);
            fb[j][1] = *(const bf16x8*)(wb + j * 2048 + loff1);
        }
    };
    auto read_a = [&](int t, auto i0_tag, auto n_tag) {
        constexpr int I0 = decltype(i0_tag)::value, NR = decltype(n_tag)::value;
#ifdef ROW_DBG_NO_READS
        return;
#endif
        const char* ab = smem + (t % 3) * R::A_BYTES + I0 * 2048;
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            fa[i][0] = *(const bf16x8*)(ab + i * 2048 + loff0);
            fa[i][1] = *(const bf16x8*)(ab + i * 2048 + loff1);
        }
    };
    auto mfmas = [&](auto i0_tag, auto n_tag) {
        constexpr int I0 = decltype(i0_tag)::value, NR = decltype(n_tag)::value;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < NR; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j)   // transposed product: lane holds row (16 i + fr), columns 48 wave + 16 j + 4 fq + (0..3)
#ifdef ROW_DBG_NO_MFMA
                    asm volatile("" ::"v"(fb[j][kk]), "v"(fa[i][kk]));
#else
                    acc[I0 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j][kk], fa[i][kk], acc[I0 + i][j], 0, 0, 0);
#endif
        __builtin_amdgcn_s_setprio(0);
    };
    auto barrier = [&]() {
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    // this wave's pieces of every tile up to `upto` have landed; `younger` = whether one later group of its pieces stays in flight
    // (wait and barrier are ONE asm statement: as two, hipcc moved the wait a slot earlier - above the previous barrier and the MFMA
    // cluster, where it stalls on pieces issued two slots before)
    auto certify_barrier = [&](bool younger) {
        __builtin_amdgcn_sched_barrier(0);
        if (!younger) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        else if (grp == 0 && a_per == 5) asm volatile("s_waitcnt vmcnt(5)\n\ts_barrier" ::: "memory");
        else if (grp == 0 && a_per == 3) asm volatile("s_waitcnt vmcnt(3)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    // The fragment reads of an L slot retire INSIDE it (under the partner group's MFMAs): the M slot then opens with its first MFMA
    // instead of the reads' latency (compute-only loop 60 -> ?? us at K = 1536).
    auto landed = [&]() {
#ifndef ROW_DBG_LATE_LGKM
        __builtin_amdgcn_s_waitcnt(0xc07f);       // lgkmcnt(0)
#endif
    };
    using I0 = std::integral_constant<int, 0>;
    using IP0 = std::integral_constant<int, R::P0>;
    using IP1 = std::integral_constant<int, MT - R::P0>;
    if (grp == 0) {
        issue_a(0);
        issue_w(0, 1);
        issue_a(1);
        certify_barrier(nt > 1);             // A(0), W(0) part 1 landed; A(1) may still fly
        for (int t = 0; t < nt; ++t) {
            read_b(t);                                        // slot 4t: L of phase 0
            read_a(t, I0{}, IP0{});
            issue_w(t + 1, 1);
            landed();
            barrier();
            mfmas(I0{}, IP0{});                               // slot 4t+1: M of phase 0
            barrier();
            read_a(t, IP0{}, IP1{});                          // slot 4t+2: L of phase 1
            issue_a(t + 2);
            landed();
            barrier();
            mfmas(IP0{}, IP1{});                              // slot 4t+3: M of phase 1
            certify_barrier(t + 2 < nt);                      // A(t+1), W(t+1) part 1 landed; A(t+2) may still fly
        }
        barrier();                                            // group 1's last M slot
    } else {
        issue_w(0, 0);
        issue_w(0, 2);
        issue_w(1, 0);
        certify_barrier(nt > 1);             // W(0) parts 0 and 2 landed; W(1) part 0 may still fly
        barrier();                                            // one slot behind group 0
        for (int t = 0; t < nt; ++t) {
            read_b(t);                                        // slot 4t+1
            read_a(t, I0{}, IP0{});
            issue_w(t + 1, 2);
            landed();
            barrier();
            mfmas(I0{}, IP0{});                               // slot 4t+2
            barrier();
            read_a(t, IP0{}, IP1{});                          // slot 4t+3
            issue_w(t + 2, 0);
            landed();
            certify_barrier(t + 2 < nt);                      // W(t+1) parts 0 and 2 landed; W(t+2) part 0 may still fly
            mfmas(IP0{}, IP1{});                              // slot 4t+4
            barrier();
        }
    }
#ifdef ROW_DBG_NO_EPI                        // timing experiment: main loop only
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) asm volatile("" ::"v"(acc[i][j]));
    return;
#endif

    // ---- epilogue: PASS_ROWS rows per pass through the fp32 scratch, then the LayerNorm row loop (half-wave per row)
    float* sc = (float*)smem;
    const int hl = lane & 31;
    const int hw = wave * 2 + (lane >> 5);          // half-wave 0..15: rows hw, hw + 16, ... of a pass
    const float inv_d = 1.0f / ROW_N;
    f32x4 gm[3], bt[3], bs[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int c = (j * 32 + hl) * 4;
        gm[j] = a.gamma ? *(const f32x4*)(a.gamma + c) : f32x4{1.f, 1.f, 1.f, 1.f};
        bt[j] = (MODE == ROW_MODE_FWD && a.beta) ? *(const f32x4*)(a.beta + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        bs[j] = (MODE == ROW_MODE_FWD && a.bias) ? *(const f32x4*)(a.bias + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    f32x4 dg[3], db[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        dg[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        db[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    auto load_fwd = [&](RowFwdIn& in, int64_t m) {
        const bool ok = m < a.M;
#pragma unroll
        for (int j = 0; j < 3; ++j)
            in.res[j] = ok ? ROW_LOAD((const f32x4*)(a.residual + m * a.ldr + (j * 32 + hl) * 4)) : f32x4{0.f, 0.f, 0.f, 0.f};
    };
    auto load_bwd = [&](RowBwdIn& in, int64_t m) {
        const bool ok = m < a.M;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int c = (j * 32 + hl) * 4;
            in.x[j] = ok ? ROW_LOAD((const f32x4*)(a.x + m * a.ldx + c)) : f32x4{0.f, 0.f, 0.f, 0.f};
            in.din[j] = (ok && a.dx_in) ? ROW_LOAD((const f32x4*)(a.dx_in + m * a.ldxi + c)) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        in.mu = ok ? a.mean_in[m] : 0.f;
        in.rs = ok ? a.rstd_in[m] : 0.f;
    };
    auto row_fwd = [&](const RowFwdIn& in, int r, int64_t m) {
        if (m >= a.M) return;
        f32x4 v[3];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int c = (j * 32 + hl) * 4;
            v[j] = *(const f32x4*)(sc + r * ROW_LD + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[j][e] += bs[j][e];
                v[j][e] += in.res[j][e];
            }
            ROW_STORE((f32x4*)(a.y + m * a.ldy + c), v[j]);
            s += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
        }
        if (!a.h) return;
        const float mu = row_half_sum(s) * inv_d;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = v[j][e] - mu;
                q = fmaf(d, d, q);
            }
        const float rs = rsqrtf(row_half_sum(q) * inv_d + a.eps);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int c = (j * 32 + hl) * 4;
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (bf16)((v[j][e] - mu) * rs * gm[j][e] + bt[j][e]);
            ROW_STORE_LP((bf16x4*)(a.h + m * a.ldh + c), o);
        }
        if (hl == 0) {
            a.mean[m] = mu;
            a.rstd[m] = rs;
        }
    };
    auto row_bwd = [&](const RowBwdIn& in, int r, int64_t m) {
        if (m >= a.M) return;
        f32x4 xh[3], gg[3];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int c = (j * 32 + hl) * 4;
            const f32x4 d = *(const f32x4*)(sc + r * ROW_LD + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                xh[j][e] = (in.x[j][e] - in.mu) * in.rs;
                gg[j][e] = d[e] * gm[j][e];
                s1 += gg[j][e];
                s2 = fmaf(gg[j][e], xh[j][e], s2);
                dg[j][e] = fmaf(d[e], xh[j][e], dg[j][e]);
                db[j][e] += d[e];
            }
        }
        const float c1 = row_half_sum(s1) * inv_d, c2 = row_half_sum(s2) * inv_d;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int c = (j * 32 + hl) * 4;
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = in.rs * (gg[j][e] - c1 - xh[j][e] * c2) + in.din[j][e];
            ROW_STORE((f32x4*)(a.dx + m * a.lddx + c), v);
            if (a.dx_lp) ROW_STORE_LP((bf16x4*)(a.dx_lp + m * a.ldlp + c), (bf16x4{(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]}));
        }
    };

#pragma unroll
    for (int p = 0; p < R::NPASS; ++p) {
        const int64_t mp = m0 + p * R::PASS_ROWS;
        // ALL of this half-wave's rows of the pass are requested before the dump and its two barriers where the registers allow
        // (a half-wave is one of only 16 per CU: with one row - 1.5 to 3 KB - in flight each the epilogue was 2-3 % slower);
        // the 160-row tile keeps one row ahead
        constexpr int AHEAD = R::EPI_ALL_ROWS ? R::ROWS_PER_HALF : 1;
        RowFwdIn fin[AHEAD + (R::EPI_ALL_ROWS ? 0 : 1)];
        RowBwdIn bin[AHEAD + (R::EPI_ALL_ROWS ? 0 : 1)];
#pragma unroll
        for (int k = 0; k < AHEAD; ++k) {
            if (MODE == ROW_MODE_FWD) load_fwd(fin[k], mp + hw + 16 * k);
            else load_bwd(bin[k], mp + hw + 16 * k);
        }
        __syncthreads();       // every wave is done with the LDS (main loop's last stage / the previous pass's rows)
#pragma unroll
        for (int ii = 0; ii < R::PASS_SUB; ++ii) {
            const int i = R::PASS_SUB * p + ii;
#pragma unroll
            for (int j = 0; j < 3; ++j) *(f32x4*)(sc + (ii * 16 + fr) * ROW_LD + wave * 48 + j * 16 + fq * 4) = acc[i][j];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < R::ROWS_PER_HALF; ++k) {
            const int r = hw + 16 * k;
            if constexpr (R::EPI_ALL_ROWS) {
                if (MODE == ROW_MODE_FWD) row_fwd(fin[k], r, mp + r);
                else row_bwd(bin[k], r, mp + r);
            } else {
                const bool more = k + 1 < R::ROWS_PER_HALF;
                if (MODE == ROW_MODE_FWD) {
                    if (more) load_fwd(fin[(k + 1) & 1], mp + r + 16);
                    row_fwd(fin[k & 1], r, mp + r);
                } else {
                    if (more) load_bwd(bin[(k + 1) & 1], mp + r + 16);
                    row_bwd(bin[k & 1], r, mp + r);
                }
            }
        }
    }

    if (MODE == ROW_MODE_BWD) {
        // the 16 half-waves' column partials -> one [2][384] row per workgroup (summed over workgroups by ln_bwd_finish)
        __syncthreads();
        float* my = sc + hw * 2 * ROW_N;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int c = (j * 32 + hl) * 4;
            *(f32x4*)(my + c) = dg[j];
            *(f32x4*)(my + ROW_N + c) = db[j];
        }
        __syncthreads();
        float* out = a.partial + (int64_t)blockIdx.x * 2 * ROW_N;
        for (int c = threadIdx.x; c < 2 * ROW_N; c += 512) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < 16; ++w) s += sc[w * 2 * ROW_N + c];
            out[c] = s;
        }
    }
}

// ------------------------------------------------------------------------------------------------
static inline int row_mt(int64_t M) {
    // rounds of 256 workgroups (one per CU) x rows per tile: the smallest makespan wins, the taller tile on a tie (the W panel
    // is staged once per tile).  65,536 rows = 512 x 128 (two rounds); 66,560 = 463 x 144 (two rounds, where 520 x 128 needs three);
    // 73,800 (config H's decoder, 72 pairs) = 462 x 160 (two rounds of 160 rows against three of 128); 24,576 = 256 x 96 (one).
    static const int heights[] = {10, 9, 8, 6};
    int best = 8;
    int64_t cost = -1;
    for (int mt : heights) {
        const int64_t c = ceil_div64(ceil_div64(M, 16 * mt), 256) * mt;
        if (cost < 0 || c < cost) {
            cost = c;
            best = mt;
        }
    }
    return best;
}

static inline int64_t row_tiles(int64_t M) { return ceil_div64(M, 16 * row_mt(M)); }

static bool row_shape_ok(int64_t M, int64_t N, int64_t K) { return N == ROW_N && K >= 64 && K % 64 == 0 && K <= (1 << 20) && M >= 1 && M < ((int64_t)1 << 31); }

extern "C" int vited_linear_layernorm_supported(int64_t M, int64_t N, int64_t K) { return row_shape_ok(M, N, K) ? 1 : 0; }

template <int MODE, int MT>
static int row_launch_mt(const RowArgs& a, unsigned tiles, hipStream_t s) {
    auto kernel = gemm_row_kernel<MODE, MT>;
    static std::once_flag once;          // dynamic LDS above 64 KB needs the opt-in, once per kernel instance
    static hipError_t status = hipSuccess;
    std::call_once(once, [&] { status = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, RowCfg<MT>::LDS_BYTES); });
    if (status != hipSuccess) return VITED_ERR_LAUNCH;
    hipLaunchKernelGGL(kernel, dim3(tiles), dim3(512), RowCfg<MT>::LDS_BYTES, s, a);
    return vited_check_launch();
}

template <int MODE>
static int row_launch(const RowArgs& a, hipStream_t s) {
    const unsigned tiles = (unsigned)row_tiles(a.M);
    switch (row_mt(a.M)) {
        case 6: return row_launch_mt<MODE, 6>(a, tiles, s);
        case 9: return row_launch_mt<MODE, 9>(a, tiles, s);
        case 10: return row_launch_mt<MODE, 10>(a, tiles, s);
        default: return row_launch_mt<MODE, 8>(a, tiles, s);
    }
}

static inline bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

extern "C" int vited_linear_residual_layernorm_fwd(const void* a, int64_t lda, const void* w, int64_t ldw, const float* bias,
                                                   const float* residual, int64_t ldr, float* y, int64_t ldy, const float* gamma,
                                                   const float* beta, float eps, void* h, int64_t ldh, float* mean, float* rstd,
                                                   int64_t M, int64_t N, int64_t K, void* stream) {
    if (!a || !w || !residual || !y || M <= 0 || N <= 0 || K <= 0 || lda < K || ldw < K || ldr < N || ldy < N) return VITED_ERR_BAD_ARG;
    if (h && (!gamma || !beta || !mean || !rstd || ldh < N)) return VITED_ERR_BAD_ARG;
    if (!row_shape_ok(M, N, K)) return VITED_ERR_UNSUPPORTED;
    if ((lda & 7) || (ldw & 7) || (ldr & 3) || (ldy & 3) || (h && (ldh & 3))) return VITED_ERR_UNSUPPORTED;
    if (!al16(a) || !al16(w) || !al16(residual) || !al16(y) || !al16(bias) || !al16(gamma) || !al16(beta) || ((uintptr_t)h & 7))
        return VITED_ERR_BAD_ARG;
    RowArgs r = {};
    r.A = (const bf16*)a; r.lda = lda; r.seg_k = (int)K; r.seg_stride = 0; r.W = (const bf16*)w; r.ldw = ldw; r.M = M; r.K = (int)K;
    r.bias = bias; r.residual = residual; r.ldr = ldr; r.y = y; r.ldy = ldy;
    r.gamma = gamma; r.beta = beta; r.eps = eps; r.h = (bf16*)h; r.ldh = ldh; r.mean = mean; r.rstd = rstd;
    return row_launch<ROW_MODE_FWD>(r, (hipStream_t)stream);
}

// layernorm.hip
int ln_bwd_finish(const float* partial, int nparts, int dim, float* dgamma, float* dbeta, int accumulate, hipStream_t s);

extern "C" int64_t vited_linear_layernorm_bwd_partial_rows(int64_t M) { return M >= 1 ? row_tiles(M) : 0; }

extern "C" int64_t vited_linear_layernorm_bwd_workspace_bytes(int64_t M, int64_t N) {
    return N == ROW_N && M >= 1 ? row_tiles(M) * 2 * ROW_N * (int64_t)sizeof(float) : 0;
}

static int linear_layernorm_bwd_impl(const void* dy, int64_t lddy, int64_t seg_k, int64_t seg_stride, const void* wt, int64_t ldwt,
                                    const float* x, int64_t ldx, const float* gamma, const float* mean, const float* rstd,
                                    const float* dx_in, int64_t dx_in_ld, float* dx_out, int64_t dx_out_ld, void* dx_lp, int64_t dx_lp_ld,
                                    float* dgamma, float* dbeta, int accumulate, int64_t M, int64_t N, int64_t K, float* workspace,
                                    int64_t workspace_bytes, void* stream) {
    if (!dy || !wt || !x || !gamma || !mean || !rstd || !dx_out || M <= 0 || N <= 0 || K <= 0) return VITED_ERR_BAD_ARG;
    if ((dgamma == nullptr) != (dbeta == nullptr)) return VITED_ERR_BAD_ARG;
    if (seg_k <= 0 || K % seg_k || lddy < seg_k || ldwt < K || ldx < N || dx_out_ld < N || (dx_in && dx_in_ld < N) || (dx_lp && dx_lp_ld < N))
        return VITED_ERR_BAD_ARG;
    if (!row_shape_ok(M, N, K) || seg_k % 64 || (seg_stride & 7)) return VITED_ERR_UNSUPPORTED;
    if ((lddy & 7) || (ldwt & 7) || (ldx & 3) || (dx_out_ld & 3) || (dx_in && (dx_in_ld & 3)) || (dx_lp && (dx_lp_ld & 3))) return VITED_ERR_UNSUPPORTED;
    if (!al16(dy) || !al16(wt) || !al16(x) || !al16(gamma) || !al16(dx_in) || !al16(dx_out) || ((uintptr_t)dx_lp & 7)) return VITED_ERR_BAD_ARG;
    if (!workspace || workspace_bytes < vited_linear_layernorm_bwd_workspace_bytes(M, N)) return VITED_ERR_WORKSPACE;
    RowArgs r = {};
    r.A = (const bf16*)dy; r.lda = lddy; r.seg_k = (int)seg_k; r.seg_stride = seg_stride; r.W = (const bf16*)wt; r.ldw = ldwt; r.M = M; r.K = (int)K;
    r.gamma = gamma; r.x = x; r.ldx = ldx; r.mean_in = mean; r.rstd_in = rstd;
    r.dx_in = dx_in; r.ldxi = dx_in_ld; r.dx = dx_out; r.lddx = dx_out_ld; r.dx_lp = (bf16*)dx_lp; r.ldlp = dx_lp_ld;
    r.partial = workspace;
    const int rc = row_launch<ROW_MODE_BWD>(r, (hipStream_t)stream);
    if (rc != VITED_OK || !dgamma) return rc;     // dgamma == null: the caller keeps the partials and finishes several LayerNorms at once
    return ln_bwd_finish(workspace, (int)row_tiles(M), ROW_N, dgamma, dbeta, accumulate, (hipStream_t)stream);
}

extern "C" int vited_linear_layernorm_bwd(const void* dy, int64_t lddy, const void* wt, int64_t ldwt, const float* x, int64_t ldx,
                                          const float* gamma, const float* mean, const float* rstd, const float* dx_in,
                                          int64_t dx_in_ld, float* dx_out, int64_t dx_out_ld, void* dx_lp, int64_t dx_lp_ld,
                                          float* dgamma, float* dbeta, int accumulate, int64_t M, int64_t N, int64_t K,
                                          float* workspace, int64_t workspace_bytes, void* stream) {
    return linear_layernorm_bwd_impl(dy, lddy, K, 0, wt, ldwt, x, ldx, gamma, mean, rstd, dx_in, dx_in_ld, dx_out, dx_out_ld, dx_lp, dx_lp_ld,
                                     dgamma, dbeta, accumulate, M, N, K, workspace, workspace_bytes, stream);
}

// dy = `segments` tensors [M, seg_k] (row stride lddy) laid out seg_stride ELEMENTS apart: column block j of the contraction
// comes from tensor j (the d(kv) of decoder block j, each contiguous for its own attention backward).  wt is [N, segments * seg_k].
extern "C" int vited_linear_layernorm_bwd_segmented(const void* dy, int64_t lddy, int64_t seg_k, int64_t seg_stride, int64_t segments,
                                                    const void* wt, int64_t ldwt, const float* x, int64_t ldx, const float* gamma,
                                                    const float* mean, const float* rstd, const float* dx_in, int64_t dx_in_ld,
                                                    float* dx_out, int64_t dx_out_ld, void* dx_lp, int64_t dx_lp_ld, float* dgamma,
                                                    float* dbeta, int accumulate, int64_t M, int64_t N, float* workspace,
                                                    int64_t workspace_bytes, void* stream) {
    if (segments <= 0) return VITED_ERR_BAD_ARG;
    return linear_layernorm_bwd_impl(dy, lddy, seg_k, seg_stride, wt, ldwt, x, ldx, gamma, mean, rstd, dx_in, dx_in_ld, dx_out, dx_out_ld, dx_lp,
                                     dx_lp_ld, dgamma, dbeta, accumulate, M, N, segments * seg_k, workspace, workspace_bytes, stream);
}
