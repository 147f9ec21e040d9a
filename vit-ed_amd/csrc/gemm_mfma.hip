// bf16 MFMA GEMM kernels for gfx950 (CDNA4), the contraction engine of the ViT-ED path.
//
//  gemm_nt_mfma_kernel : out = epilogue(A[M,K] . B[N,K]^T)        (Linear fwd, and dX via the W^T shadow)
//                        128 x 128 tile / 4 waves (2 x 2, each 64 x 64 = 4 x 4 v_mfma_f32_16x16x32_bf16 accumulators), or
//                        256 x 128 / 8 waves for the plain-store K = 384, N >= 768 shapes; two LDS stages (builtin LDS-DMA,
//                        one barrier per K-step) or a three-stage ring (inline-asm LDS-DMA, counted vmcnt, raw s_barrier)
//  gemm_tn_mfma_kernel : dW[N,K] = sum_m dY[m,N]^T X[m,K] (+ dbias[N] = sum_m dY[m,N]) on a 128 x 128 tile, split over M
//                        into fp32 slabs that a small pass sums (deterministic); K % 384 != 0 or M < 4,096 only
//  gemm_tn_wide_kernel : the same product on a 128 n x 384 k tile, 8 waves (2 x 4, each 64 x 96), one workgroup per CU,
//                        4-slot ring with three stages in flight: every dW of the embed-384 models
//  (the row-complete N = 384 kernels with fused LayerNorm epilogues live in gemm_row.hip)
//
// Operands are staged global -> LDS with 16-byte global_load_lds (LDS-DMA, no VGPR round trip).  LDS images are XOR-swizzled
// on the SOURCE address (the DMA destination is lane-linear) with the matching XOR on the read, so the
// ds_read_b128 / ds_read_b64_tr_b16 fragment reads are bank-conflict free.
// The NT epilogue is staged through a wave-private LDS scratch so every global access is a 16-byte,
// row-contiguous one; its operands (residual rows, saved pre-activation, bias) are fetched after the K loop,
// one 16-row sub-tile ahead (held across the loop they cost a workgroup per CU).
// Timing-ablation hooks (NT_DBG_* / TN_DBG_* / TW_DBG_* / NT_TIMELINE / NT_STAGGER_US) and the VITED_NT_* / VITED_TN_*
// environment overrides compile only into experiment builds: make VARIANT=x EXTRA="-DVITED_TUNING -D...".
#include <stdlib.h>

#include <mutex>

#include "gemm_kernels.h"
#include "gemm_nt_epilogue.h"
#include "gemm_lds.h"

// ---- NT stage: A tile [128 rows][BKT bf16] then B tile, each wave DMAs 32 rows of both ---------------
template <int BKT, int WM, bool ASM_DMA = false>
__device__ __forceinline__ void nt_stage_load(const bf16* __restrict__ A, int64_t lda, const bf16* __restrict__ B,
                                              int64_t ldb, int64_t m0, int64_t n0, int64_t M, int64_t N, int64_t k0,
                                              char* stage, int wave, int lane) {
    using C = NtCfg<BKT, WM>;
    const int rsub = lane / C::CHUNKS, cp = lane % C::CHUNKS;
#pragma unroll
    for (int i = 0; i < 32 / C::ROWS_PER_DMA; ++i) {          // A: 32 rows per wave
        const int r = wave * 32 + i * C::ROWS_PER_DMA + rsub;
        const int c = cp ^ C::swz(r);
        int64_t gm = m0 + r;
        gm = gm < M ? gm : M - 1;
        const bf16* asrc = A + gm * lda + k0 + c * 8;
        if constexpr (ASM_DMA) glds16_asm(asrc, stage + (wave * 32 + i * C::ROWS_PER_DMA) * C::ROW_BYTES);
        else glds16(asrc, stage + (wave * 32 + i * C::ROWS_PER_DMA) * C::ROW_BYTES);
    }
    constexpr int RB = BN / (2 * WM);                         // B: 128 rows over all waves
#pragma unroll
    for (int i = 0; i < RB / C::ROWS_PER_DMA; ++i) {
        const int r = wave * RB + i * C::ROWS_PER_DMA + rsub;
        const int c = cp ^ C::swz(r);
        int64_t gn = n0 + r;
        gn = gn < N ? gn : N - 1;
        const bf16* bsrc = B + gn * ldb + k0 + c * 8;
        if constexpr (ASM_DMA) glds16_asm(bsrc, stage + C::A_BYTES + (wave * RB + i * C::ROWS_PER_DMA) * C::ROW_BYTES);
        else glds16(bsrc, stage + C::A_BYTES + (wave * RB + i * C::ROWS_PER_DMA) * C::ROW_BYTES);
    }
}

#ifdef NT_TIMELINE
// diagnostic build only (make VARIANT=tl EXTRA=-DNT_TIMELINE): per-workgroup time stamps of the NT kernel
__device__ unsigned long long nt_timeline[16384 * 4];
extern "C" int vited_debug_timeline(void* dst, int64_t bytes) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(nt_timeline), bytes, 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : 3;
}
#endif

// STAGES = 2: two LDS stages, the next stage's DMA in flight under the current stage's MFMAs (builtin LDS-DMA, __syncthreads).
// STAGES >= 3: a ring with STAGES - 1 stages in flight across the barrier (inline-asm LDS-DMA, counted vmcnt, raw s_barrier).
template <int EPI, int BKT, int WM, int STAGES = 2>
__global__ void __launch_bounds__(128 * WM)
gemm_nt_mfma_kernel(const bf16* __restrict__ A, int64_t lda, const bf16* __restrict__ B, int64_t ldb, int64_t M, int64_t N,
                    int64_t K, int tiles_n, int ntiles, EpiParams ep) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int tile = xcd_remap(blockIdx.x, ntiles);
    const int64_t m0 = (int64_t)(tile / tiles_n) * (64 * WM), n0 = (int64_t)(tile % tiles_n) * BN;

#ifdef NT_STAGGER_US
    // Workgroups b, b + 256, b + 512, ... share a CU (8 XCDs x 32 CUs are dealt round-robin).  Started together and doing identical
    // work they stay in lockstep: the whole chip stages operands (L2 -> LDS bound), then the whole chip stores (HBM bound), and the two
    // phases ADD (timeline: profiles/nt_timeline.py).  Delaying the k-th resident workgroup of a CU by k slices of the tile time
    // de-phases them once; identical tile times keep them de-phased.
    if (blockIdx.x < 256 * NT_STAGGER_SLOTS) {
        const int slot = blockIdx.x >> 8;
        for (int i = 0; i < slot * NT_STAGGER_US; ++i) __builtin_amdgcn_s_sleep(32);   // ~1 us per iteration
    }
#endif
#ifdef NT_TIMELINE
    const unsigned long long tl0 = __builtin_amdgcn_s_memrealtime();
#endif
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    using C = NtCfg<BKT, WM>;
    constexpr bool RING = STAGES >= 3;
    constexpr int PER = 32 / C::ROWS_PER_DMA + (BN / (2 * WM)) / C::ROWS_PER_DMA;   // DMA instructions per stage per wave
    const int nk = (int)(K / BKT);
    nt_stage_load<BKT, WM, RING>(A, lda, B, ldb, m0, n0, M, N, 0, smem, wave, lane);
    if constexpr (RING) {
#pragma unroll
        for (int pre = 1; pre < STAGES - 1; ++pre)
            if (pre < nk) nt_stage_load<BKT, WM, RING>(A, lda, B, ldb, m0, n0, M, N, (int64_t)pre * BKT, smem + pre * C::STAGE_BYTES, wave, lane);
    }
    const int fr = lane & 15, fq = lane >> 4;
    const int64_t mtile = m0 + wr * 64, ntile = n0 + wc * 64;
    EpiPrefetch<EPI> pf;   // epilogue operands are fetched after the K loop, one 16-row sub-tile ahead: held across the
                           // loop their 64 VGPRs cost a workgroup per CU (measured: -11 % proj+res, -7 % dz, -17 % fc2+res)
    for (int t = 0; t < nk; ++t) {
        if constexpr (RING) {
            const int ahead = nk - 1 - t;                    // stages issued after stage t
            if (ahead >= STAGES - 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((STAGES - 2) * PER) : "memory");
            else if (ahead == 4 && STAGES > 6) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(4 * PER) : "memory");
            else if (ahead == 3 && STAGES > 5) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(3 * PER) : "memory");
            else if (ahead == 2 && STAGES > 4) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * PER) : "memory");
            else if (ahead == 1 && STAGES > 3) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(PER) : "memory");
            else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();   // stage t landed for every wave; everyone is done reading stage t-1
            asm volatile("" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();  // stage t landed for every wave; everyone is done reading stage t-1
        }
        if (t + STAGES - 1 < nk)
            nt_stage_load<BKT, WM, RING>(A, lda, B, ldb, m0, n0, M, N, (int64_t)(t + STAGES - 1) * BKT,
                                         smem + ((t + STAGES - 1) % STAGES) * C::STAGE_BYTES, wave, lane);
        const char* sa = smem + (t % STAGES) * C::STAGE_BYTES;
        const char* sb = sa + C::A_BYTES;
        // All operand fragments of the stage are requested first, then the MFMAs run behind counted lgkmcnt waits:
        // left to itself the compiler keeps ~6 fragments live and waits lgkmcnt(0) four times per K-step, exposing the
        // LDS latency in front of every group of 8 MFMAs.
        bf16x8 af[BKT / 32][4], bf_[BKT / 32][4];
#pragma unroll
        for (int kk = 0; kk < BKT / 32; ++kk) {
#pragma unroll
            for (int i = 0; i < 4; ++i) af[kk][i] = *(const bf16x8*)(sa + C::off(wr * 64 + i * 16 + fr, kk * 4 + fq));
#pragma unroll
            for (int j = 0; j < 4; ++j) bf_[kk][j] = *(const bf16x8*)(sb + C::off(wc * 64 + j * 16 + fr, kk * 4 + fq));
        }
        // measured (interleaved A/B): -4..5 % on the K >= 1152 shapes, +1.5 % on the BK = 32 ones, so only BK = 64 pins it
        if constexpr (BKT == 64) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < BKT / 32; ++kk) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
#ifdef NT_DBG_NO_MFMA
                    asm volatile("" ::"v"(af[kk][i]), "v"(bf_[kk][j]));
#else
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[kk][i], bf_[kk][j], acc[i][j], 0, 0, 0);
#endif
                }
        }
    }
#ifdef NT_DBG_NO_EPI
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(acc[i][j]));
    return;
#endif
#ifdef NT_TIMELINE
    const unsigned long long tl1 = __builtin_amdgcn_s_memrealtime();
#endif
    // ---- epilogue: accumulators -> per-wave LDS scratch -> full-line row segments (gemm_nt_epilogue.h)
    epilogue_prefetch_bias<EPI>(ep, pf, ntile, N, lane);
    epilogue_prefetch_subtile<EPI>(ep, pf, 0, mtile, ntile, M, N, lane);
    __syncthreads();
    float* sc = (float*)(smem + wave * SCRATCH_BYTES);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (i + 1 < 4) epilogue_prefetch_subtile<EPI>(ep, pf, i + 1, mtile, ntile, M, N, lane);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) sc[(fq * 4 + e) * SCRATCH_LD + j * 16 + fr] = acc[i][j][e];
        // the scratch is wave-private and a wave's DS operations execute in order: only the compiler needs a fence
        asm volatile("" ::: "memory");
        epilogue_subtile<EPI>(ep, pf, sc, i, mtile, ntile, M, N, lane);
        asm volatile("" ::: "memory");
    }
#ifdef NT_TIMELINE
    if (threadIdx.x == 0 && blockIdx.x < 16384) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long tl2 = __builtin_amdgcn_s_memrealtime();
        const unsigned hw = __builtin_amdgcn_s_getreg(63492), xcc = __builtin_amdgcn_s_getreg(6164);
        nt_timeline[blockIdx.x * 4 + 0] = tl0;
        nt_timeline[blockIdx.x * 4 + 1] = tl1;
        nt_timeline[blockIdx.x * 4 + 2] = tl2;
        nt_timeline[blockIdx.x * 4 + 3] = ((unsigned long long)xcc << 32) | hw;
    }
#endif
}

bool gemm_nt_mfma_supported(const void* A, int64_t lda, const void* B, int64_t ldb, int64_t M, int64_t N, int64_t K,
                            int epilogue, const EpiParams& ep) {
    auto al16 = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    if (K % 64 || N % 16 || lda % 8 || ldb % 8 || ep.ldo % 8) return false;
    if (!al16(A) || !al16(B) || !al16(ep.out)) return false;
    if (ep.bias && !al16(ep.bias)) return false;
    if ((epilogue == VITED_EPI_GELU || epilogue == VITED_EPI_GELU_GRAD) && !al16(ep.out2)) return false;
    if ((epilogue == VITED_EPI_MUL_GELU_GRAD || epilogue == VITED_EPI_MUL) && !al16(ep.aux)) return false;
    if (epilogue == VITED_EPI_RESIDUAL && !al16(ep.residual)) return false;
    if (ceil_div64(M, 128) * ceil_div64(N, BN) > (1 << 30)) return false;
    if (M >= ((int64_t)1 << 31) || ep.rows_per_batch >= ((int64_t)1 << 31)) return false;   // the row remap divides in 32 bits
    return true;
}

template <int EPI, int BKT, int WM, int STAGES = 2>
static int launch_nt(const bf16* a, int64_t lda, const bf16* b, int64_t ldb, int64_t M, int64_t N, int64_t K, const EpiParams& ep,
                      hipStream_t s) {
    using C = NtCfg<BKT, WM>;
    const int tiles_n = (int)ceil_div64(N, BN);
    const int ntiles = (int)ceil_div64(M, C::BM) * tiles_n;
    auto kernel = gemm_nt_mfma_kernel<EPI, BKT, WM, STAGES>;
    constexpr int LDS = STAGES * C::STAGE_BYTES;
    if (LDS > 65536) {   // dynamic LDS above 64 KB needs the opt-in: once per kernel instance, from whichever thread launches first
        static std::once_flag once;
        static hipError_t status = hipSuccess;
        std::call_once(once, [&] { status = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS); });
        if (status != hipSuccess) return VITED_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(kernel, dim3(ntiles), dim3(128 * WM), LDS, s, a, lda, b, ldb, M, N, K, tiles_n, ntiles, ep);
    return VITED_OK;
}

template <int EPI>
static int dispatch_nt(const bf16* a, int64_t lda, const bf16* b, int64_t ldb, int64_t M, int64_t N, int64_t K, const EpiParams& ep,
                       hipStream_t s) {
    // BK = 32 (4 workgroups / CU) pays only when the grid is large: at N = 384 (1,536 tiles) BK = 64 is 9-14 % faster
    bool shallow = K <= 512 && N >= 768;
    // measured (M = 65536): the 256-row tile wins 5-10 % on plain-store K = 384 GEMMs with
    // N >= 768 (qkv, kv, fc1) and loses on the register-heavier epilogues, so it is used only there
    // (round 3, after the streaming stores: the taller tile now also wins on fc1 + GELU' (-8 %) and on dz = . * gelu' (-17 %))
    constexpr bool tall_epi = EPI == VITED_EPI_STORE || EPI == VITED_EPI_MUL || EPI == VITED_EPI_GELU_GRAD;
    bool tall = tall_epi && shallow && N >= 768 && M >= 8192;
#ifdef VITED_TUNING   // experiment builds only: VITED_NT_BK = 32 | 64, VITED_NT_WM = 2 | 4, VITED_NT_STAGES = 3
    static const int force_bk = getenv("VITED_NT_BK") ? atoi(getenv("VITED_NT_BK")) : 0;
    static const int force_wm = getenv("VITED_NT_WM") ? atoi(getenv("VITED_NT_WM")) : 0;
    static const int force_st = getenv("VITED_NT_STAGES") ? atoi(getenv("VITED_NT_STAGES")) : 0;
    if (force_bk) shallow = force_bk == 32;
    if (force_wm) tall = force_wm == 4;
    if (force_st == 3) {
        if (shallow && tall) return launch_nt<EPI, 32, 4, 3>(a, lda, b, ldb, M, N, K, ep, s);
        if (shallow) return launch_nt<EPI, 32, 2, 3>(a, lda, b, ldb, M, N, K, ep, s);
        if (tall) return launch_nt<EPI, 64, 4, 3>(a, lda, b, ldb, M, N, K, ep, s);
        return launch_nt<EPI, 64, 2, 3>(a, lda, b, ldb, M, N, K, ep, s);
    }
#endif
    // the three-stage asm-DMA ring (72 KB, 2 workgroups per CU, two stages in flight each) pays on the plain-store 256 x 128 tiles
    // only: qkv 82 -> 78 us, kv 56 -> 54; every other variant loses occupancy to it (BK = 64: 96 KB = one workgroup per CU, +50 %)
    if (shallow && tall && EPI == VITED_EPI_STORE) return launch_nt<EPI, 32, 4, 3>(a, lda, b, ldb, M, N, K, ep, s);
    if (shallow && tall) return launch_nt<EPI, 32, 4>(a, lda, b, ldb, M, N, K, ep, s);
    if (shallow) return launch_nt<EPI, 32, 2>(a, lda, b, ldb, M, N, K, ep, s);
    if (tall) return launch_nt<EPI, 64, 4>(a, lda, b, ldb, M, N, K, ep, s);
    return launch_nt<EPI, 64, 2>(a, lda, b, ldb, M, N, K, ep, s);
}

int gemm_nt_mfma(const void* A, int64_t lda, const void* B, int64_t ldb, int64_t M, int64_t N, int64_t K, int epilogue,
                 const EpiParams& ep, hipStream_t s) {
    const bf16* a = (const bf16*)A;
    const bf16* b = (const bf16*)B;
    int rc;
    switch (epilogue) {
        case VITED_EPI_STORE: rc = dispatch_nt<VITED_EPI_STORE>(a, lda, b, ldb, M, N, K, ep, s); break;
        case VITED_EPI_GELU: rc = dispatch_nt<VITED_EPI_GELU>(a, lda, b, ldb, M, N, K, ep, s); break;
        case VITED_EPI_RESIDUAL: rc = dispatch_nt<VITED_EPI_RESIDUAL>(a, lda, b, ldb, M, N, K, ep, s); break;
        case VITED_EPI_MUL_GELU_GRAD: rc = dispatch_nt<VITED_EPI_MUL_GELU_GRAD>(a, lda, b, ldb, M, N, K, ep, s); break;
        case VITED_EPI_STORE_F32: rc = dispatch_nt<VITED_EPI_STORE_F32>(a, lda, b, ldb, M, N, K, ep, s); break;
        case VITED_EPI_MUL: rc = dispatch_nt<VITED_EPI_MUL>(a, lda, b, ldb, M, N, K, ep, s); break;
        case VITED_EPI_GELU_GRAD: rc = dispatch_nt<VITED_EPI_GELU_GRAD>(a, lda, b, ldb, M, N, K, ep, s); break;
        default: return VITED_ERR_BAD_ARG;
    }
    return rc != VITED_OK ? rc : vited_check_launch();
}

// ================================================================================================
// TN (weight gradient): stage = dY tile [64 m][128 n] + X tile [64 m][128 k], 256-byte rows.
// Image (b) of the guide's dual-use layouts: chunk ch of row r lives at ch ^ (((r&3)<<2)|((r>>2)&3));
// both MFMA operands are read column-wise with ds_read_b64_tr_b16.
// The product is issued transposed (X fragment as the A operand) so a lane holds 4 consecutive k of one n row and
// the slab leaves as 16-byte stores.  The bias gradient rides along: the workgroups of the first k-tile column
// sum their dY fragments with v_dot2_f32_bf16 against packed ones (VALU, co-issued under the MFMAs).
// ================================================================================================
// Two geometries of the same kernel (template parameters TM = m-rows per stage, STAGES = ring depth):
//   <64, 2>  two 32 KB stages, 2 workgroups per CU, one stage in flight per workgroup (64 KB per CU); builtin LDS-DMA
//   <32, 3>  three 16 KB stages, 3 workgroups per CU, TWO stages in flight per workgroup across the barrier (96 KB per CU):
//            counted vmcnt + raw s_barrier, LDS-DMA from inline asm.  11-12 % faster on the 77-GFLOP shapes (dW fc1 / fc2:
//            140 -> 123 us at M = 65,536), +-4 % on the others, so only those take it (gemm_tn_mfma_geometry).
#ifndef TN_FORCE_TM          // -DTN_FORCE_TM=32 -DTN_FORCE_STAGES=3: one geometry for every shape (experiments)
#define TN_FORCE_TM 0
#define TN_FORCE_STAGES 0
#endif

__device__ __forceinline__ int tn_f(int r) { return ((r & 3) << 2) | ((r >> 2) & 3); }

__device__ __forceinline__ bf16x4 tr_read(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)p);
}

template <int TM, bool ASM_DMA>
__device__ __forceinline__ void tn_stage_load(const bf16* __restrict__ dY, int64_t lddy, const bf16* __restrict__ X,
                                              int64_t ldx, int64_t mrow0, int64_t mlast, int64_t n0, int64_t N,
                                              int64_t kc0, int64_t K, char* stage, int wave, int lane) {
    constexpr int T_TILE_BYTES = TM * 128 * 2;
    const int rsub = lane >> 4, cp = lane & 15;
#pragma unroll
    for (int i = 0; i < TM / 16; ++i) {
        const int r = wave * (TM / 4) + i * 4 + rsub;
        const int ch = cp ^ tn_f(r);
        int64_t gm = mrow0 + r;
        gm = gm <= mlast ? gm : mlast;  // clamped rows are zeroed in LDS after landing
        int64_t cn = n0 + ch * 8;
        cn = cn <= N - 8 ? cn : N - 8;
        int64_t ck = kc0 + ch * 8;
        ck = ck <= K - 8 ? ck : K - 8;
        if constexpr (ASM_DMA) {
            glds16_asm(dY + gm * lddy + cn, stage + (wave * (TM / 4) + i * 4) * 256);
            glds16_asm(X + gm * ldx + ck, stage + T_TILE_BYTES + (wave * (TM / 4) + i * 4) * 256);
        } else {
            glds16(dY + gm * lddy + cn, stage + (wave * (TM / 4) + i * 4) * 256);
            glds16(X + gm * ldx + ck, stage + T_TILE_BYTES + (wave * (TM / 4) + i * 4) * 256);
        }
    }
}

template <bool BIAS, int TM, int TN_STAGES>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3)))   // 3 workgroups/CU: <= 168 VGPRs
gemm_tn_mfma_kernel(const bf16* __restrict__ dY, int64_t lddy, const bf16* __restrict__ X, int64_t ldx, int64_t M, int64_t N,
                    int64_t K, int64_t rows_per_split, int tiles_k, float* __restrict__ out, float* __restrict__ bias_out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int T_TILE_BYTES = TM * 128 * 2, T_STAGE_BYTES = 2 * T_TILE_BYTES;
    constexpr bool RING = TN_STAGES >= 3;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    // All tiles of one M-split read the same dY / X rows: keep them on ONE XCD (its L2 then serves the
    // re-reads) by giving each XCD a contiguous run of the (split-major, tile-minor) order.
    const int ntile = gridDim.x;
    const int lin = xcd_remap(blockIdx.x + blockIdx.y * ntile, ntile * gridDim.y);
    const int split = lin / ntile, tile_id = lin - split * ntile;
    const int64_t n0 = (int64_t)(tile_id / tiles_k) * 128, kc0 = (int64_t)(tile_id % tiles_k) * 128;
    const int64_t mb = (int64_t)split * rows_per_split;
    int64_t me = mb + rows_per_split;
    me = me < M ? me : M;

    f32x4 acc[4][4];
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};   // dbias partials: this lane's 8 m-rows of dY column (lane & 15) of block i
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    const int nsteps = me > mb ? (int)((me - mb + TM - 1) / TM) : 0;
    if (nsteps > 0) tn_stage_load<TM, RING>(dY, lddy, X, ldx, mb, me - 1, n0, N, kc0, K, smem, wave, lane);
    if constexpr (RING) {
#pragma unroll
        for (int pre = 1; pre < TN_STAGES - 1; ++pre)
            if (nsteps > pre) tn_stage_load<TM, RING>(dY, lddy, X, ldx, mb + pre * TM, me - 1, n0, N, kc0, K, smem + pre * T_STAGE_BYTES, wave, lane);
    }
    const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    for (int t = 0; t < nsteps; ++t) {
        // S-slot ring (S >= 3): stages t + 1 .. t + S - 2 stay in flight across the barrier (counted vmcnt, raw s_barrier; the DMA is
        // issued from inline asm, see glds16_asm)
        if constexpr (RING) {
            constexpr int PER = 2 * (TM / 16);          // DMA instructions per stage per wave
            const int ahead = nsteps - 1 - t;            // stages issued after stage t
            if (ahead >= TN_STAGES - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((TN_STAGES - 2) * PER) : "memory");
            else if (ahead == 2 && TN_STAGES > 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER) : "memory");
            else if (ahead == 1 && TN_STAGES > 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        char* st = smem + (t % TN_STAGES) * T_STAGE_BYTES;
        const int64_t mrow0 = mb + (int64_t)t * TM;
        if (mrow0 + TM > me) {  // ragged last stage: zero the rows this lane's DMA clamped
            const int rsub = lane >> 4, cp = lane & 15;
#pragma unroll
            for (int i = 0; i < TM / 16; ++i) {
                const int r = wave * (TM / 4) + i * 4 + rsub;
                if (mrow0 + r >= me) {
                    *(f32x4*)(st + r * 256 + cp * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
                    *(f32x4*)(st + T_TILE_BYTES + r * 256 + cp * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
        }
        if constexpr (RING) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();   // stage t landed for every wave; everyone is done reading stage t - 1
            asm volatile("" ::: "memory");
        } else {
            __syncthreads();
        }
        if (t + TN_STAGES - 1 < nsteps)
            tn_stage_load<TM, RING>(dY, lddy, X, ldx, mrow0 + (TN_STAGES - 1) * TM, me - 1, n0, N, kc0, K,
                                    smem + ((t + TN_STAGES - 1) % TN_STAGES) * T_STAGE_BYTES, wave, lane);
        const char* sy = st;
        const char* sx = st + T_TILE_BYTES;
#ifdef TN_DBG_DMA_ONLY
        continue;
#endif
#pragma unroll
        for (int ms = 0; ms < TM / 32; ++ms) {
            bf16x8 af[4], bf_[4];
            const int r_lo = ms * 32 + 8 * g + q, r_hi = r_lo + 4;
            const int f_lo = tn_f(r_lo), f_hi = tn_f(r_hi);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ch = ((wr * 64 + i * 16) >> 3) + (p >> 1);
                const bf16x4 lo = tr_read(sy + r_lo * 256 + ((ch ^ f_lo) << 4) + 8 * (p & 1));
                const bf16x4 hi = tr_read(sy + r_hi * 256 + ((ch ^ f_hi) << 4) + 8 * (p & 1));
                af[i] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ch = ((wc * 64 + j * 16) >> 3) + (p >> 1);
                const bf16x4 lo = tr_read(sx + r_lo * 256 + ((ch ^ f_lo) << 4) + 8 * (p & 1));
                const bf16x4 hi = tr_read(sx + r_hi * 256 + ((ch ^ f_hi) << 4) + 8 * (p & 1));
                bf_[j] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
#ifdef TN_DBG_NO_MFMA
                    asm volatile("" ::"v"(af[i]), "v"(bf_[j]));
#else
                    // transposed product (X fragment as the A operand): a lane then holds 4 CONSECUTIVE k of one n row,
                    // so the slab leaves as 16-byte stores (dword stores of the n-major layout cost 6 % of the kernel)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf_[j], af[i], acc[i][j], 0, 0, 0);
#endif
                }
            if (BIAS && kc0 == 0) {
                // column sums of dY on the VALU (v_dot2_f32_bf16 against packed ones), branch-free so the MFMA
                // schedule is untouched: an MFMA-by-ones variant and a dealt (conditional) one both cost 70-100 VGPRs
                typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
                const bf16x2_t one2 = {(__bf16)1.0f, (__bf16)1.0f};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        bsum[i] = __builtin_amdgcn_fdot2_f32_bf16(bf16x2_t{af[i][2 * e], af[i][2 * e + 1]}, one2, bsum[i], false);
                }
            }
        }
    }
    float* o = out + (int64_t)split * N * K;
    const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // acc[i][j]: row n = ... + fr, columns k = ... + 4 fq + (0..3); K % 8 == 0 makes the 4 columns all-or-nothing
            const int64_t n = n0 + wr * 64 + i * 16 + fr;
            const int64_t k = kc0 + wc * 64 + j * 16 + fq * 4;
#ifdef TN_DBG_NO_STORE
            asm volatile("" ::"v"(acc[i][j]));
#else
            if (n < N && k < K) *(f32x4*)(o + n * K + k) = acc[i][j];
#endif
        }
    }
    if constexpr (BIAS) {
        if (kc0 == 0 && wc == 0) {   // every k-tile / wave column holds the same sums: one of them stores
            float* bo = bias_out + (int64_t)split * N;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float v = bsum[i];
                v += __shfl_xor(v, 16);
                v += __shfl_xor(v, 32);
                const int64_t n = n0 + wr * 64 + i * 16 + fr;
                if (fq == 0 && n < N) bo[n] = v;
            }
        }
    }
}

// ================================================================================================
// Wide TN: 128 n x 384 k output tile, 512 threads = 8 waves as 2 (n) x 4 (k), each wave 64 n x 96 k = 4 x 6 accumulators.
// Why: the 128 x 128 kernel above stages 512 B of operands per m-row per 16,384 MACs and every dY / X column panel is
// re-read by 3 / 12 other tiles (HBM fetch measured at 2.8 x the algorithmic bytes on dW fc1); this tile stages 1,024 B
// per 49,152 MACs (1.5 x fewer bytes through the ~70 GB/s-per-CU L2 -> LDS path that bounds these kernels), reads dY
// exactly once, issues 10 transposed fragment reads per 24 MFMAs instead of 8 per 16 ... and X's 768-byte rows are
// shared by the n-tiles of one m-split on ONE XCD.  One workgroup per CU: a 4-slot ring of 32-row stages
// (stage = [32 m][128 n] of dY + 3 panels [32 m][128 k] of X = 32 KB, same swizzled 256-byte-row image as above),
// three stages (96 KB) in flight per CU across the barrier - counted vmcnt, raw s_barrier, LDS-DMA from inline asm.
// Taken when K is a multiple of 384 (every dW of the embed-384 models: qkv / kv / proj / fc1 / fc2).
// ================================================================================================
#define TW_STAGES 4   // a power of two: slot = stage & 3
#define TW_PANEL_BYTES (32 * 256)
#define TW_STAGE_BYTES (4 * TW_PANEL_BYTES)
#define TW_PER 4   // LDS-DMA instructions per stage per wave

// This wave's four 1-KiB pieces of one stage: rows rbase + 4 i + (lane >> 4) of its panel.  src[i] is the lane's source pointer of
// piece i for the stage (advanced by the caller, 32 rows per stage); the ragged last stage re-derives it with the row clamped.
__device__ __forceinline__ void tw_stage_issue(const bf16* const (&src)[4], char* stage, int panel, int rbase) {
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16_asm(src[i], stage + panel * TW_PANEL_BYTES + (rbase + i * 4) * 256);
}

struct TwFrags {
    bf16x8 a[4], b[6];
};

__device__ __forceinline__ void tw_read_frags(TwFrags& f, const char* st, const int (&ya_lo)[4], const int (&ya_hi)[4],
                                              const int (&xb_lo)[6], const int (&xb_hi)[6]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const bf16x4 lo = tr_read(st + ya_lo[i]);
        const bf16x4 hi = tr_read(st + ya_hi[i]);
        f.a[i] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const bf16x4 lo = tr_read(st + xb_lo[j]);
        const bf16x4 hi = tr_read(st + xb_hi[j]);
        f.b[j] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
}

// One launch may carry several weight-gradient products (the dW of every Linear of one transformer block, queued by
// functions.py and flushed at the end of the block's backward): the chip's 256 workgroup slots are then shared by ~36-48 output
// tiles instead of 3-12, so each product is split over 5-7 row ranges instead of 21-85 and writes that many fewer fp32 slabs
// (round 2: 9 GB of slab traffic per step, as much as the operands for the N = K = 384 products).
#define TW_MAX_PROBLEMS 40      // 40 x 88 B of descriptors ride in the kernel arguments (4 KB limit)
struct TwProblem {
    const bf16* dY;
    const bf16* X;
    int64_t lddy, ldx, M, rows_per_split;
    int64_t slab_off, bias_off;     // float offsets of this product inside ONE split's slab block / bias block
    int N, K, tiles_k, tile0;       // tile0 = index of its first output tile in the launch
    int has_bias, pad_;
};
struct TwBatch {
    TwProblem p[TW_MAX_PROBLEMS];
    int64_t slab_stride, bias_stride;   // floats per split
    int count, total_tiles;
};

template <bool BIAS>
__global__ void __launch_bounds__(512)
gemm_tn_wide_kernel(const TwBatch batch, float* __restrict__ out_base, float* __restrict__ bias_base) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int ntile = batch.total_tiles;
    const int lin = xcd_remap(blockIdx.x + blockIdx.y * ntile, ntile * gridDim.y);   // one m-split's tiles share an XCD
    const int split = lin / ntile;
    int tile_id = lin - split * ntile;
    int pi = 0;
    for (int i = 1; i < batch.count; ++i)
        if (batch.p[i].tile0 <= tile_id) pi = i;
    const TwProblem& pr = batch.p[pi];
    tile_id -= pr.tile0;
    const bf16* __restrict__ dY = pr.dY;
    const bf16* __restrict__ X = pr.X;
    const int64_t lddy = pr.lddy, ldx = pr.ldx, M = pr.M, N = pr.N, K = pr.K, rows_per_split = pr.rows_per_split;
    const int tiles_k = pr.tiles_k;
    const bool with_bias = BIAS && pr.has_bias;
    float* __restrict__ out = out_base + pr.slab_off;
    float* __restrict__ bias_out = bias_base + pr.bias_off;
    const int64_t n0 = (int64_t)(tile_id / tiles_k) * 128, kc0 = (int64_t)(tile_id % tiles_k) * 384;
    const int64_t mb = (int64_t)split * rows_per_split;
    int64_t me = mb + rows_per_split;
    me = me < M ? me : M;
    const int nsteps = me > mb ? (int)((me - mb + 31) / 32) : 0;

    f32x4 acc[4][6];
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 6; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // ---- loader role of this wave: panel (wave >> 1) (0 = dY, 1..3 = X), rows rbase .. rbase + 15 of every stage
    const int panel = wave >> 1, rbase = (wave & 1) * 16;
    const int rsub = lane >> 4, cp = lane & 15;
    const bf16* const pbase = panel == 0 ? dY : X;
    const int64_t ld = panel == 0 ? lddy : ldx;
    const int64_t col0 = panel == 0 ? n0 : kc0 + (panel - 1) * 128;
    const int64_t colmax = panel == 0 ? N - 8 : col0 + 120;
    int64_t coff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int64_t c = col0 + ((cp ^ tn_f(rbase + i * 4 + rsub)) << 3);
        coff[i] = c <= colmax ? c : colmax;
    }
    // source pointers of stage 0 (rows unclamped); a full stage s reads 32 s rows further down: two VALU adds per piece, the
    // 64-bit row multiply and the clamp stay off the loop (only the ragged last stage re-derives its rows)
    const bf16* p0[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) p0[i] = pbase + (mb + rbase + i * 4 + rsub) * ld + coff[i];
    const int64_t stage_stride = 32 * ld;
    const int rows_here = (int)(me - mb);            // rows of this split: 32-bit scalar arithmetic from here on
    auto issue = [&](int stage_idx) {   // stage_idx < nsteps
        char* dst = smem + (stage_idx & (TW_STAGES - 1)) * TW_STAGE_BYTES;
        if ((stage_idx + 1) * 32 <= rows_here) {   // a real (scalar) branch: each arm issues its own DMA
            const int64_t adv = (int64_t)stage_idx * stage_stride;    // wave-uniform
            const bf16* src[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) src[i] = p0[i] + adv;
            tw_stage_issue(src, dst, panel, rbase);
        } else {
            const bf16* src[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int r = stage_idx * 32 + rbase + i * 4 + rsub;
                r = r < rows_here ? r : rows_here - 1;          // clamped rows are zeroed in LDS after landing
                src[i] = pbase + (mb + r) * ld + coff[i];
            }
            tw_stage_issue(src, dst, panel, rbase);
        }
    };
    // wait until this wave's pieces of stage `idx` have landed: the stages issued after it stay in flight
    auto wait_landed = [&](int idx) {
        const int later = (nsteps - 1 < idx + TW_STAGES - 2 ? nsteps - 1 : idx + TW_STAGES - 2) - idx;   // stages idx+1 .. issued so far
        if (later >= 3 && TW_STAGES > 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * TW_PER) : "memory");
        else if (later >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * TW_PER) : "memory");
        else if (later == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(TW_PER) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if ((idx + 1) * 32 > rows_here) {   // ragged last stage: zero the rows this lane's DMA clamped
            char* st = smem + (idx & (TW_STAGES - 1)) * TW_STAGE_BYTES;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = rbase + i * 4 + rsub;
                if (idx * 32 + r >= rows_here) *(f32x4*)(st + panel * TW_PANEL_BYTES + r * 256 + cp * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
    };

    // ---- fragment offsets inside a stage (rows r_lo = 8g + q and r_lo + 4 of the 32-row step, as in the kernel above)
    const int g = lane >> 4, q = (lane & 15) >> 2, p = lane & 3;
    const int r_lo = 8 * g + q, r_hi = r_lo + 4;
    const int f_lo = tn_f(r_lo), f_hi = tn_f(r_hi);
    int ya_lo[4], ya_hi[4], xb_lo[6], xb_hi[6];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ch = ((wr * 64 + i * 16) >> 3) + (p >> 1);
        ya_lo[i] = r_lo * 256 + ((ch ^ f_lo) << 4) + 8 * (p & 1);
        ya_hi[i] = r_hi * 256 + ((ch ^ f_hi) << 4) + 8 * (p & 1);
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int col = wc * 96 + j * 16;
        const int ch = ((col & 127) >> 3) + (p >> 1);
        const int base = (1 + (col >> 7)) * TW_PANEL_BYTES;
        xb_lo[j] = base + r_lo * 256 + ((ch ^ f_lo) << 4) + 8 * (p & 1);
        xb_hi[j] = base + r_hi * 256 + ((ch ^ f_hi) << 4) + 8 * (p & 1);
    }

    // Pipeline (round 3: the slot schedule of gemm_row.hip).  A step = one 32-row stage = two SLOTS separated by workgroup barriers:
    // L = the step's 20 transposed fragment reads + this wave's 4 LDS-DMA pieces of stage t + 3, M = its 24 MFMAs.  Group 1 (waves
    // 4-7, the lower 64 n-rows' partners on each SIMD) runs one slot behind group 0, so every SIMD always has one wave in its MFMA
    // cluster and one wave loading.  Stage t is read in slots 2t (group 0) and 2t + 1 (group 1), both retired inside their slot:
    // its ring slot is rewritten (stage t + 4) from slot 2t + 2; every wave certifies its pieces of stage t + 1 (counted vmcnt,
    // two later stages stay in flight) before the barrier that ends slot 2t + 1.
    auto slot_l = [&](int t, TwFrags& f) {
#ifndef TW_DBG_NO_READS
        tw_read_frags(f, smem + (t & (TW_STAGES - 1)) * TW_STAGE_BYTES, ya_lo, ya_hi, xb_lo, xb_hi);
#endif
#ifndef TW_DBG_NO_DMA
        if (t + TW_STAGES - 1 < nsteps) issue(t + TW_STAGES - 1);      // into the slot stage t - 1 left a slot ago
#endif
        __builtin_amdgcn_s_waitcnt(0xc07f);       // lgkmcnt(0): the reads retire inside the slot, under the partner group's MFMAs
    };
    auto slot_m = [&](const TwFrags& cur) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int j = 0; j < 6; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) {  // transposed product (X fragment as the A operand): a lane holds 4 consecutive k of one n row
#ifdef TW_DBG_NO_MFMA
                asm volatile("" ::"v"(cur.b[j]), "v"(cur.a[i]));
#else
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur.b[j], cur.a[i], acc[i][j], 0, 0, 0);
#endif
            }
        __builtin_amdgcn_s_setprio(0);
        if (with_bias && kc0 == 0) {
            typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
            const bf16x2_t one2 = {(__bf16)1.0f, (__bf16)1.0f};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    bsum[i] = __builtin_amdgcn_fdot2_f32_bf16(bf16x2_t{cur.a[i][2 * e], cur.a[i][2 * e + 1]}, one2, bsum[i], false);
            }
        }
    };
    auto barrier = [&]() {
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    // this wave's pieces of stage idx have landed (stages idx + 1 .. issued so far stay in flight), ragged rows zeroed; then the barrier
    auto certify_barrier = [&](int idx) {
        __builtin_amdgcn_sched_barrier(0);
        if (idx < nsteps) {
            wait_landed(idx);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the zero fill of a ragged stage
        }
        barrier();
    };

    if (nsteps > 0) {
#pragma unroll
        for (int pre = 0; pre < TW_STAGES - 1; ++pre)
            if (nsteps > pre) issue(pre);
        certify_barrier(0);
        TwFrags f;
        if (wr == 0) {
            for (int t = 0; t < nsteps; ++t) {
                slot_l(t, f);                   // slot 2t
                barrier();
                slot_m(f);                      // slot 2t + 1
                certify_barrier(t + 1);
            }
            barrier();                          // group 1's last M slot
        } else {
            barrier();                          // one slot behind group 0
            for (int t = 0; t < nsteps; ++t) {
                slot_l(t, f);                   // slot 2t + 1
                certify_barrier(t + 1);
                slot_m(f);                      // slot 2t + 2
                barrier();
            }
        }
    }

    float* o = out + (int64_t)split * batch.slab_stride;
    const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t n = n0 + wr * 64 + i * 16 + fr;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int64_t k = kc0 + wc * 96 + j * 16 + fq * 4;
            if (n < N) *(f32x4*)(o + n * K + k) = acc[i][j];   // plain: the slab sum reads it back at once (nt here: +0.08 ms)
        }
    }
    if constexpr (BIAS) {
        if (with_bias && kc0 == 0 && wc == 0) {
            float* bo = bias_out + (int64_t)split * batch.bias_stride;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float v = bsum[i];
                v += __shfl_xor(v, 16);
                v += __shfl_xor(v, 32);
                const int64_t n = n0 + wr * 64 + i * 16 + fr;
                if (fq == 0 && n < N) bo[n] = v;
            }
        }
    }
}

bool gemm_tn_mfma_supported(const void* dY, int64_t lddy, const void* X, int64_t ldx, int64_t M, int64_t N, int64_t K) {
    auto al16 = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    return N % 8 == 0 && K % 8 == 0 && N >= 8 && K >= 8 && lddy % 8 == 0 && ldx % 8 == 0 && al16(dY) && al16(X) && M >= 1;
}

// Geometry per shape: the three-stage ring pays on the largest shapes only (measured, see the comment above the kernel).
static inline bool tn_use_ring(int64_t M, int64_t N, int64_t K) {
    if (TN_FORCE_TM) return TN_FORCE_STAGES >= 3;
#ifdef VITED_TUNING
    static const char* env = getenv("VITED_TN_RING");          // tuning override: 0 = never, 1 = always
    if (env) return atoi(env) != 0;
#endif
    return N * K >= 1536 * 384 && M >= 16384;
}

// Splits over M: as many as fill the chip's resident workgroup slots (2 per CU at 64 KB of LDS, 3 per CU at 48 KB)
// in ONE round - one workgroup more than a round costs a whole extra round.
// The wide tile pays once a split is long enough to amortise its ring fill and its 192 KB slab (M >= 4,096 rows).
static inline bool tn_use_wide(int64_t M, int64_t N, int64_t K) {
    if (TN_FORCE_TM) return false;
    if (K % 384 != 0) return false;
#ifdef VITED_TUNING
    static const char* env = getenv("VITED_TN_WIDE");           // tuning override: 0 = never, 1 = whenever the shape allows
    if (env) return atoi(env) != 0;
#endif
    return M >= 4096;
}

int64_t gemm_tn_mfma_splits(int64_t M, int64_t N, int64_t K) {
    if (tn_use_wide(M, N, K)) {   // one workgroup per CU, one round
        const int64_t tiles = ceil_div64(N, 128) * (K / 384);
        int64_t slots = 256;
#ifdef VITED_TUNING
        static const int64_t slots_env = getenv("VITED_TN_WIDE_SLOTS") ? atoi(getenv("VITED_TN_WIDE_SLOTS")) : 0;
        if (slots_env) slots = slots_env;
#endif
        int64_t s = slots / tiles;
        const int64_t max_s = ceil_div64(M, 256);
        if (s > max_s) s = max_s;
        if (s < 1) s = 1;
        const int64_t rps = ceil_div64(ceil_div64(M, s), 32) * 32;
        return ceil_div64(M, rps);
    }
    const bool ring = tn_use_ring(M, N, K);
    const int tm = TN_FORCE_TM ? TN_FORCE_TM : (ring ? 32 : 64);
    const int64_t tiles = ceil_div64(N, 128) * ceil_div64(K, 128);
    int64_t slots = tm == 64 ? 512 : 768;
#ifdef VITED_TUNING
    static const int64_t slots_env = getenv("VITED_TN_SLOTS") ? atoi(getenv("VITED_TN_SLOTS")) : 0;
    if (slots_env) slots = slots_env;
#endif
    int64_t s = slots / tiles;
    const int64_t max_s = ceil_div64(M, 512);
    if (s > max_s) s = max_s;
    if (s < 1) s = 1;
    const int64_t rps = ceil_div64(ceil_div64(M, s), tm) * tm;
    return ceil_div64(M, rps);
}

template <int TM, int STAGES>
static int launch_tn(const void* dY, int64_t lddy, const void* X, int64_t ldx, int64_t M, int64_t N, int64_t K, int64_t splits,
                      float* out, float* bias_out, hipStream_t s) {
    constexpr int LDS = STAGES * 2 * TM * 128 * 2;
    const int tiles_k = (int)ceil_div64(K, 128);
    const int tiles = (int)ceil_div64(N, 128) * tiles_k;
    const int64_t rps = ceil_div64(ceil_div64(M, splits), TM) * TM;
    if (LDS > 65536) {   // dynamic LDS above 64 KB needs the opt-in (once per kernel)
        static std::once_flag once;
        static hipError_t status = hipSuccess;
        std::call_once(once, [&] {
            status = hipFuncSetAttribute((const void*)gemm_tn_mfma_kernel<true, TM, STAGES>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
            if (status == hipSuccess)
                status = hipFuncSetAttribute((const void*)gemm_tn_mfma_kernel<false, TM, STAGES>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        });
        if (status != hipSuccess) return VITED_ERR_LAUNCH;
    }
    if (bias_out)
        hipLaunchKernelGGL((gemm_tn_mfma_kernel<true, TM, STAGES>), dim3(tiles, (unsigned)splits), dim3(256), LDS, s,
                           (const bf16*)dY, lddy, (const bf16*)X, ldx, M, N, K, rps, tiles_k, out, bias_out);
    else
        hipLaunchKernelGGL((gemm_tn_mfma_kernel<false, TM, STAGES>), dim3(tiles, (unsigned)splits), dim3(256), LDS, s,
                           (const bf16*)dY, lddy, (const bf16*)X, ldx, M, N, K, rps, tiles_k, out, bias_out);
    return VITED_OK;
}

static int launch_tn_wide_batch(const TwBatch& b, int64_t splits, bool any_bias, float* out, float* bias_out, hipStream_t s) {
    constexpr int LDS = TW_STAGES * TW_STAGE_BYTES;
    static std::once_flag once;
    static hipError_t status = hipSuccess;
    std::call_once(once, [&] {
        status = hipFuncSetAttribute((const void*)gemm_tn_wide_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        if (status == hipSuccess)
            status = hipFuncSetAttribute((const void*)gemm_tn_wide_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    });
    if (status != hipSuccess) return VITED_ERR_LAUNCH;
    if (any_bias)
        hipLaunchKernelGGL((gemm_tn_wide_kernel<true>), dim3(b.total_tiles, (unsigned)splits), dim3(512), LDS, s, b, out, bias_out);
    else
        hipLaunchKernelGGL((gemm_tn_wide_kernel<false>), dim3(b.total_tiles, (unsigned)splits), dim3(512), LDS, s, b, out, bias_out);
    return VITED_OK;
}

static int launch_tn_wide(const void* dY, int64_t lddy, const void* X, int64_t ldx, int64_t M, int64_t N, int64_t K, int64_t splits,
                          float* out, float* bias_out, hipStream_t s) {
    TwBatch b = {};
    TwProblem& p = b.p[0];
    p.dY = (const bf16*)dY; p.X = (const bf16*)X; p.lddy = lddy; p.ldx = ldx; p.M = M;
    p.rows_per_split = ceil_div64(ceil_div64(M, splits), 32) * 32;
    p.N = (int)N; p.K = (int)K; p.tiles_k = (int)(K / 384); p.tile0 = 0; p.has_bias = bias_out != nullptr;
    b.count = 1;
    b.total_tiles = (int)ceil_div64(N, 128) * p.tiles_k;
    b.slab_stride = N * K;
    b.bias_stride = N;
    return launch_tn_wide_batch(b, splits, bias_out != nullptr, out, bias_out, s);
}

// ---- several products in one launch (see TwBatch) ------------------------------------------------------------------------
bool gemm_tn_batch_supported(int count, const int64_t* M, const int64_t* N, const int64_t* K) {
    if (count < 1 || count > TW_MAX_PROBLEMS) return false;
    for (int i = 0; i < count; ++i)
        if (!tn_use_wide(M[i], N[i], K[i]) || N[i] % 8 || N[i] > (1 << 20) || K[i] > (1 << 20)) return false;
    return true;
}

static inline int64_t tn_batch_splits(int count, const int64_t* M, const int64_t* N, const int64_t* K) {
    int64_t tiles = 0, mmin = M[0];
    for (int i = 0; i < count; ++i) {
        tiles += ceil_div64(N[i], 128) * (K[i] / 384);
        mmin = M[i] < mmin ? M[i] : mmin;
    }
    int64_t s = 256 / tiles;                      // one workgroup per CU, one round
    const int64_t max_s = ceil_div64(mmin, 256);
    if (s > max_s) s = max_s;
    return s < 1 ? 1 : s;
}

void gemm_tn_batch_layout(int count, const int64_t* M, const int64_t* N, const int64_t* K, int64_t* splits, int64_t* slab_stride,
                          int64_t* bias_stride) {
    *splits = tn_batch_splits(count, M, N, K);
    int64_t w = 0, b = 0;
    for (int i = 0; i < count; ++i) {
        w += N[i] * K[i];
        b += N[i];
    }
    *slab_stride = w;
    *bias_stride = b;
}

int gemm_tn_batch(int count, const void* const* dY, const int64_t* lddy, const void* const* X, const int64_t* ldx, const int64_t* M,
                  const int64_t* N, const int64_t* K, const int* has_bias, float* slabs, float* bias_slabs, hipStream_t s) {
    TwBatch b = {};
    int64_t splits;
    gemm_tn_batch_layout(count, M, N, K, &splits, &b.slab_stride, &b.bias_stride);
    int tile = 0;
    int64_t woff = 0, boff = 0;
    bool any_bias = false;
    for (int i = 0; i < count; ++i) {
        TwProblem& p = b.p[i];
        if (!gemm_tn_mfma_supported(dY[i], lddy[i], X[i], ldx[i], M[i], N[i], K[i])) return VITED_ERR_UNSUPPORTED;
        p.dY = (const bf16*)dY[i]; p.X = (const bf16*)X[i]; p.lddy = lddy[i]; p.ldx = ldx[i]; p.M = M[i];
        p.rows_per_split = ceil_div64(ceil_div64(M[i], splits), 32) * 32;
        p.N = (int)N[i]; p.K = (int)K[i]; p.tiles_k = (int)(K[i] / 384); p.tile0 = tile; p.has_bias = has_bias[i];
        p.slab_off = woff; p.bias_off = boff;
        tile += (int)ceil_div64(N[i], 128) * p.tiles_k;
        woff += N[i] * K[i];
        boff += N[i];
        any_bias = any_bias || has_bias[i];
    }
    b.count = count;
    b.total_tiles = tile;
    const int rc = launch_tn_wide_batch(b, splits, any_bias, slabs, bias_slabs, s);
    return rc != VITED_OK ? rc : vited_check_launch();
}

int gemm_tn_mfma(const void* dY, int64_t lddy, const void* X, int64_t ldx, int64_t M, int64_t N, int64_t K, int64_t splits,
                 float* out, float* bias_out, hipStream_t s) {
    int rc;
    if (tn_use_wide(M, N, K)) rc = launch_tn_wide(dY, lddy, X, ldx, M, N, K, splits, out, bias_out, s);
#if TN_FORCE_TM
    else rc = launch_tn<TN_FORCE_TM, TN_FORCE_STAGES>(dY, lddy, X, ldx, M, N, K, splits, out, bias_out, s);
#else
    else if (tn_use_ring(M, N, K)) rc = launch_tn<32, 3>(dY, lddy, X, ldx, M, N, K, splits, out, bias_out, s);
    else rc = launch_tn<64, 2>(dY, lddy, X, ldx, M, N, K, splits, out, bias_out, s);
#endif
    return rc != VITED_OK ? rc : vited_check_launch();
}
