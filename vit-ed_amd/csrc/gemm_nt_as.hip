// Activation-stationary persistent bf16 MFMA GEMM for the embed-dim contractions of the ViT-ED path:
//     out = epilogue(A[M, 384] . W[N, 384]^T),   M ~ 65k token rows, N = 384 .. 1536.
//
// The one-tile-per-workgroup kernel (gemm_mfma.hip) is latency-bound here: a 128 x 128 tile lives
// for 6 K-steps and every tile starts with a cold fetch.  This kernel turns the loop nest inside out:
//   * a workgroup (4 waves = one per SIMD with the whole 512-register file, 1 per CU) is persistent and owns a contiguous range of
//     (256-row m-tile, 64-column n-tile) units; ranges differ by at most one unit, so the grid is
//     balanced for any M (66560 = 260 m-tiles included);
//   * each wave keeps its 64 A rows x 384 K in REGISTERS (48 MFMA row fragments straight from HBM;
//     an A row is read once per n-range and never touches LDS);
//   * the weights stream L2 -> LDS through a ring of [64 n][64 k] slices (8 KB each) with the LDS-DMA
//     running DIST slices ahead - across unit boundaries and epilogues - under counted
//     s_waitcnt vmcnt(N) and raw s_barrier, so the stream never drains (W < 1.2 MB, hot in every L2);
//   * operand reads are software-pipelined one half-slice ahead of the MFMAs that consume them
//     (two register sets F0/F1), with the per-slice barrier placed BETWEEN the two halves, so LDS
//     latency and the barrier hide under 16 MFMAs per wave; one operand fragment feeds 4 MFMAs (4 row
//     sub-tiles per wave), so LDS traffic is a quarter of a KB per MFMA;
//   * the MFMA is issued transposed (W fragment as the A operand, activation fragment as B), which
//     leaves 4 consecutive output columns of one row in each lane: the epilogue is a plain 8-byte
//     (bf16) / 16-byte (fp32) store per accumulator, no LDS round trip; bias comes from an LDS copy so
//     no ordinary VMEM load (which would make hipcc drain the DMA queue) exists in the steady state.
// Epilogues: STORE, GELU (pre-activation + activation), STORE_F32.  K must be 384 (6 slices),
// M % 256 == 0, N % 64 == 0; everything else stays on gemm_nt_mfma.
#include "gemm_kernels.h"

namespace as {

constexpr int NKS = 6;                 // K = 384 = 6 slices of 64
// RW = 16-row sub-tiles per wave.  RW = 4: 256-row workgroup, whole register file, 1 workgroup per CU (the first
// version).  RW = 2: 128-row workgroup, <= 256 VGPRs, 2 workgroups per CU - one workgroup's stores and stream
// hiccups hide under the other's MFMAs, at the price of streaming W once per 128 instead of 256 rows.
template <int RW> struct Geo {
    static constexpr int NSLOT = RW == 4 ? 16 : 8;   // ring slots
    static constexpr int DIST = RW == 4 ? 7 : 5;     // slices in flight ahead of the consumer
    static constexpr int ROWS = 64 * RW;             // rows per workgroup
    static_assert(DIST >= 2 && DIST <= NKS + 1 && NSLOT >= DIST + 2, "ring geometry");
};
constexpr int BN = 64;                 // unit width (columns)
constexpr int SLICE_BYTES = BN * 64 * 2;
constexpr int MAX_N = 2048;            // bias staged in LDS

__device__ __forceinline__ void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
__device__ __forceinline__ int off(int r, int c) { return r * 128 + ((c ^ ((r >> 1) & 7)) << 4); }
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void block_barrier() { asm volatile("s_barrier" ::: "memory"); }

// one [64 n][64 k] weight slice: 8 x 1 KiB DMAs, 2 per wave (8 rows each)
// The stream position (which n-tile / k-slice comes next, which ring slot it goes to) is kept in
// incrementally updated scalars: no integer division in the steady-state loop.
struct WStream {
    const bf16* row[2];   // this lane's two source rows of n-tile 0, k-slice 0 (swizzled chunk folded in)
    int64_t tile_stride;  // BN * ldw elements
    int j, k, slot;       // next slice to issue: n-tile, k-slice, ring slot
};
template <int NSLOT>
__device__ __forceinline__ void issue_next_slice(WStream& ws, char* ring, int tiles_n, int wave) {
    const int64_t o = (int64_t)ws.j * ws.tile_stride + ws.k * 64;
    char* dst = ring + ws.slot * SLICE_BYTES + wave * 16 * 128;
    glds16(ws.row[0] + o, dst);
    glds16(ws.row[1] + o, dst + 8 * 128);
    if (++ws.k == NKS) { ws.k = 0; if (++ws.j == tiles_n) ws.j = 0; }
    if (++ws.slot == NSLOT) ws.slot = 0;
}

template <int RW> struct AFrags {
    bf16x8 f[NKS][2][RW];  // [k slice][kk][16-row sub-tile]
};
struct WFrags {
    bf16x8 f[4];          // one half-slice (32 k): 4 n-tiles of 16
};

template <int RW>
__device__ __forceinline__ void load_a(const bf16* __restrict__ A, int64_t lda, int64_t row0, int fr, int fq, int ks, AFrags<RW>& a) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < RW; ++i) a.f[ks][kk][i] = *(const bf16x8*)(A + (row0 + i * 16 + fr) * lda + ks * 64 + kk * 32 + fq * 8);
}

__device__ __forceinline__ void read_w(const char* slice, int kk, int fr, int fq, WFrags& w) {
#pragma unroll
    for (int j = 0; j < 4; ++j) w.f[j] = *(const bf16x8*)(slice + off(j * 16 + fr, kk * 4 + fq));
}

// D[n_local = 4 fq + e][m_local = fr] += W_frag . A_frag^T  (transposed product, see header)
template <int RW>
__device__ __forceinline__ void mma_half(f32x4 (&acc)[RW][4], const WFrags& w, const bf16x8 (&af)[RW]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#ifdef AS_DBG_NO_MFMA
        asm volatile("" ::"v"(w.f[j]));
#else
#pragma unroll
        for (int i = 0; i < RW; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w.f[j], af[i], acc[i][j], 0, 0, 0);
#endif
    }
}

template <int EPI, int RW> struct OutTraits {
    static constexpr int NST = 4 * RW * (EPI == VITED_EPI_GELU ? 2 : 1);  // store instructions per wave and unit
};

// lane (fr, fq) holds out[m = mrow0 + 16 i + fr][n = n0 + 16 j + 4 fq + (0..3)] in acc[i][j] (bias already
// inside: the accumulators start from it).  Stores are buffer stores: a per-lane byte offset that never
// changes plus a scalar offset per (unit, i, j) - no address arithmetic on the vector pipe.
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
struct OutDesc {
    __amdgpu_buffer_rsrc_t out, out2;
    int lane_off;   // bytes: (fr * ldo + 4 fq) * elem
    int ldo_bytes;  // row pitch in bytes
};

template <int EPI, int RW>
__device__ __forceinline__ void epilogue_unit(const OutDesc& od, const f32x4 (&acc)[RW][4], int64_t mrow0, int64_t n0) {
    constexpr int ES = EPI == VITED_EPI_STORE_F32 ? 4 : 2;
    const int base = __builtin_amdgcn_readfirstlane((int)(mrow0 * od.ldo_bytes + n0 * ES));
#pragma unroll
    for (int i = 0; i < RW; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int soff = base + i * 16 * od.ldo_bytes + j * 16 * ES;
            const f32x4 v = acc[i][j];
            if constexpr (EPI == VITED_EPI_STORE_F32) {
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), od.out, od.lane_off, soff, 0);
            } else {
                const bf16x4 z = {(bf16)v[0], (bf16)v[1], (bf16)v[2], (bf16)v[3]};
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, z), od.out, od.lane_off, soff, 0);
                if constexpr (EPI == VITED_EPI_GELU) {
                    const bf16x4 u = {(bf16)gelu_fast(v[0]), (bf16)gelu_fast(v[1]), (bf16)gelu_fast(v[2]), (bf16)gelu_fast(v[3])};
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, u), od.out2, od.lane_off, soff, 0);
                }
            }
        }
}

// One (m-tile, n-tile) unit = 6 ring slices.  On entry F0 holds the operand fragments of (slice s, kk 0).
// RELOAD: the next unit starts a new m-tile, so each A slice is re-fetched for it as soon as its last
// use here has issued.
template <int EPI, int RW, bool RELOAD>
__device__ __forceinline__ void run_unit(const bf16* __restrict__ A, int64_t lda, const bf16* __restrict__ W, int64_t ldw, char* ring,
                                         const float* bias_lds, const OutDesc& od, AFrags<RW>& a, WFrags& F0, WFrags& F1, int& s, int s_end,
                                         WStream& ws, int& cur_slot, int tiles_n, bool first_unit, int64_t mrow0, int64_t next_mrow0,
                                         int64_t n0, int wave, int lane) {
    constexpr int NSLOT = Geo<RW>::NSLOT, DIST = Geo<RW>::DIST;
    const int fr = lane & 15, fq = lane >> 4;
    f32x4 acc[RW][4];   // start from the bias of this lane's 4 columns (zeros were staged when there is none)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f32x4 b = *(const f32x4*)(bias_lds + n0 + 16 * j + 4 * fq);
#pragma unroll
        for (int i = 0; i < RW; ++i) acc[i][j] = b;
    }
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks, ++s) {
        const char* cur = ring + cur_slot * SLICE_BYTES;
        read_w(cur, 1, fr, fq, F1);                       // second half of slice s, one phase ahead
        mma_half<RW>(acc, F0, a.f[ks][0]);                    // first half (fragments read one phase ago)
        if (s + 1 < s_end) {
            // Slice s+1 has landed once only VMEM ops issued after its 2 DMAs may be outstanding: the DMAs
            // of slices s+2 .. s+DIST-1 (2 each) and, when the previous unit's epilogue was issued inside
            // that window (ks <= DIST-2), its NST stores.  A reloads only add younger ops (safe side).
            constexpr int AHEAD = 2 * (DIST - 2);
            constexpr int WITH_ST = AHEAD + OutTraits<EPI, RW>::NST > 63 ? 63 : AHEAD + OutTraits<EPI, RW>::NST;  // 6-bit counter
            if (s + DIST - 1 < s_end) {
                if (!first_unit && ks <= DIST - 2) wait_vmcnt<WITH_ST>(); else wait_vmcnt<AHEAD>();
            } else {
                wait_vmcnt<0>();                          // tail of the stream
            }
            block_barrier();   // slice s+1 complete in LDS for every wave; slot of slice s+DIST-NSLOT is free
            if (s + DIST < s_end) issue_next_slice<NSLOT>(ws, ring, tiles_n, wave);
            if (++cur_slot == NSLOT) cur_slot = 0;
            read_w(ring + cur_slot * SLICE_BYTES, 0, fr, fq, F0);            // first half of slice s+1
        }
        mma_half<RW>(acc, F1, a.f[ks][1]);
        if constexpr (RELOAD) load_a<RW>(A, lda, next_mrow0, fr, fq, ks, a);
    }
#ifdef AS_DBG_NO_EPI
#pragma unroll
    for (int i = 0; i < RW; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(acc[i][j]));
    return;
#endif
    epilogue_unit<EPI, RW>(od, acc, mrow0, n0);
}

template <int EPI, int RW>
__global__ void __launch_bounds__(256, RW == 4 ? 1 : 2)
gemm_nt_as_kernel(const bf16* __restrict__ A, int64_t lda, const bf16* __restrict__ W, int64_t ldw, int64_t N, int tiles_m,
                  int tiles_n, EpiParams ep) {
    constexpr int NSLOT = Geo<RW>::NSLOT, DIST = Geo<RW>::DIST, ROWS = Geo<RW>::ROWS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ring = smem;
    float* bias_lds = (float*)(smem + NSLOT * SLICE_BYTES);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int64_t units = (int64_t)tiles_m * tiles_n;
    const int64_t u0 = units * blockIdx.x / gridDim.x, u1 = units * (blockIdx.x + 1) / gridDim.x;
    if (u1 <= u0) return;
    const int n_units = (int)(u1 - u0);
    const int s_end = n_units * NKS;
    const int u0_j = (int)(u0 % tiles_n);

    for (int i = threadIdx.x; i < N; i += 256) bias_lds[i] = ep.bias ? ep.bias[i] : 0.f;
    OutDesc od;
    {
        constexpr int ES = EPI == VITED_EPI_STORE_F32 ? 4 : 2;
        const int64_t bytes = (int64_t)tiles_m * ROWS * ep.ldo * ES;
        od.out = __builtin_amdgcn_make_buffer_rsrc(ep.out, 0, (int)bytes, 0x00020000);
        od.out2 = __builtin_amdgcn_make_buffer_rsrc(EPI == VITED_EPI_GELU ? ep.out2 : ep.out, 0, (int)bytes, 0x00020000);
        od.lane_off = (int)((fr * ep.ldo + 4 * fq) * ES);
        od.ldo_bytes = (int)(ep.ldo * ES);
    }
    AFrags<RW> a;
    {
        const int64_t mrow0 = (u0 / tiles_n) * ROWS + wave * 16 * RW;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) load_a<RW>(A, lda, mrow0, fr, fq, ks, a);
    }
    WStream ws;
    {
        const int rsub = lane >> 3, cp = lane & 7;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = wave * 16 + i * 8 + rsub;
            ws.row[i] = W + (int64_t)r * ldw + (cp ^ ((r >> 1) & 7)) * 8;
        }
        ws.tile_stride = (int64_t)BN * ldw;
        ws.j = u0_j;
        ws.k = 0;
        ws.slot = 0;
    }
    // fill the ring: slices 0 .. DIST-1
#pragma unroll
    for (int d = 0; d < DIST; ++d)
        if (d < s_end) issue_next_slice<NSLOT>(ws, ring, tiles_n, wave);
    int cur_slot = 0;
    __syncthreads();   // bias copy + first slices visible (this waits for everything issued so far)
    WFrags F0, F1;
    read_w(ring, 0, fr, fq, F0);
    int s = 0;
    int64_t mt = u0 / tiles_n;
    int j = u0_j;
    for (int iu = 0; iu < n_units; ++iu, ++j) {
        if (j == tiles_n) { j = 0; ++mt; }
        const int64_t mrow0 = mt * ROWS + wave * 16 * RW;
        const bool reload = (j == tiles_n - 1) && (iu + 1 < n_units);
        if (reload)
            run_unit<EPI, RW, true>(A, lda, W, ldw, ring, bias_lds, od, a, F0, F1, s, s_end, ws, cur_slot, tiles_n, iu == 0, mrow0, mrow0 + ROWS, (int64_t)j * BN, wave, lane);
        else
            run_unit<EPI, RW, false>(A, lda, W, ldw, ring, bias_lds, od, a, F0, F1, s, s_end, ws, cur_slot, tiles_n, iu == 0, mrow0, 0, (int64_t)j * BN, wave, lane);
    }
}

}  // namespace as

static int as_cu_count() {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        n = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                ? prop.multiProcessorCount : 256;
    }
    return n;
}

bool gemm_nt_as_supported(const void* A, int64_t lda, const void* B, int64_t ldb, int64_t M, int64_t N, int64_t K, int epilogue,
                          const EpiParams& ep) {
    if (epilogue != VITED_EPI_STORE && epilogue != VITED_EPI_GELU && epilogue != VITED_EPI_STORE_F32) return false;
    if (!gemm_nt_mfma_supported(A, lda, B, ldb, M, N, K, epilogue, ep)) return false;
    if (K != 64 * as::NKS || M % 128 || N % as::BN || N > as::MAX_N) return false;
    if (M / 128 > (1 << 22)) return false;
    if (M * ep.ldo * 4 >= (int64_t)1 << 31) return false;   // 32-bit buffer offsets
    return true;
}

template <int EPI, int RW>
static int launch_as_rw(const bf16* a, int64_t lda, const bf16* b, int64_t ldb, int64_t M, int64_t N, const EpiParams& ep, hipStream_t s) {
    using G = as::Geo<RW>;
    const int tiles_m = (int)(M / G::ROWS), tiles_n = (int)(N / as::BN);
    int64_t grid = (int64_t)as_cu_count() * (RW == 4 ? 1 : 2);
    if (grid > (int64_t)tiles_m * tiles_n) grid = (int64_t)tiles_m * tiles_n;
    const size_t lds = (size_t)G::NSLOT * as::SLICE_BYTES + as::MAX_N * sizeof(float);
    auto kernel = as::gemm_nt_as_kernel<EPI, RW>;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(256), lds, s, a, lda, b, ldb, N, tiles_m, tiles_n, ep);
    return vited_check_launch();
}

template <int EPI>
static int launch_as(const bf16* a, int64_t lda, const bf16* b, int64_t ldb, int64_t M, int64_t N, const EpiParams& ep, hipStream_t s) {
    static const int force_rw = getenv("VITED_AS_RW") ? atoi(getenv("VITED_AS_RW")) : 0;   // tuning override: 2 | 4
    const bool wide = force_rw ? force_rw == 4 : false;
    if (wide && M % 256 == 0) return launch_as_rw<EPI, 4>(a, lda, b, ldb, M, N, ep, s);
    return launch_as_rw<EPI, 2>(a, lda, b, ldb, M, N, ep, s);
}

int gemm_nt_as(const void* A, int64_t lda, const void* B, int64_t ldb, int64_t M, int64_t N, int64_t K, int epilogue,
               const EpiParams& ep, hipStream_t s) {
    const bf16* a = (const bf16*)A;
    const bf16* b = (const bf16*)B;
    switch (epilogue) {
        case VITED_EPI_STORE: return launch_as<VITED_EPI_STORE>(a, lda, b, ldb, M, N, ep, s);
        case VITED_EPI_GELU: return launch_as<VITED_EPI_GELU>(a, lda, b, ldb, M, N, ep, s);
        case VITED_EPI_STORE_F32: return launch_as<VITED_EPI_STORE_F32>(a, lda, b, ldb, M, N, ep, s);
        default: return VITED_ERR_UNSUPPORTED;
    }
}
