// LDS staging helpers shared by the bf16 MFMA GEMM kernels (gemm_mfma.hip, gemm_row.hip): the XOR-swizzled stage image,
// the LDS-DMA wrappers and the XCD-aware tile order.
#pragma once
#include "common.h"

#define BN 128
// K-step depth per LDS stage: 64 (2 stages = 64 KB, 2 workgroups per CU) for long contractions,
// 32 (2 stages = 32 KB, 4 workgroups per CU) for K <= 512 where a tile's life is only a few steps and
// occupancy, not pipeline depth, is what hides HBM latency.
// WM = waves along M (each wave computes 64 x 64): 2 -> 128 x 128 tile / 256 threads, 4 -> 256 x 128 tile /
// 512 threads.  The taller tile stages 25 % fewer operand bytes per FLOP through the LDS-DMA path, which is
// what bounds these kernels.
template <int BKT, int WM = 2> struct NtCfg {
    static constexpr int BM = 64 * WM;
    static constexpr int ROW_BYTES = BKT * 2;
    static constexpr int A_BYTES = BM * ROW_BYTES;
    static constexpr int STAGE_BYTES = A_BYTES + BN * ROW_BYTES;
    static constexpr int ROWS_PER_DMA = 1024 / ROW_BYTES;   // one global_load_lds wave-instruction = 1 KiB
    static constexpr int CHUNKS = ROW_BYTES / 16;
    // chunk c of row r lives at c ^ swz(r): every ds_read_b128 lane group then covers all 64 banks
    __device__ static __forceinline__ int swz(int r) {
        return BKT == 64 ? ((r >> 1) & 7) : ((0x78 >> (2 * ((r >> 2) & 3))) & 3);
    }
    __device__ static __forceinline__ int off(int r, int c) { return r * ROW_BYTES + ((c ^ swz(r)) << 4); }
};

__device__ __forceinline__ void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
// The same LDS-DMA issued from inline asm: invisible to hipcc's waitcnt bookkeeping, which otherwise drains a ring that is more
// than one stage deep (it puts s_waitcnt vmcnt(0) in front of the first ds_read whenever it cannot prove the stages apart - the
// TN_STAGES == 3 build of round 1 measured exactly that drain, not the ring).  The caller counts vmcnt itself.
__device__ __forceinline__ void glds16_asm(const void* g, void* l) {
    unsigned keep;
    // low 32 bits of the flat address of an LDS object = its LDS byte offset (the aperture sits in the high half); the proper
    // generic -> local cast costs a null check per call (s_cmp_lg_u64 + s_cselect + the aperture load: 5 scalar instructions per piece)
    const unsigned dst = (unsigned)(uintptr_t)l;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(__builtin_amdgcn_readfirstlane(dst)) : "memory");
}

// bijective XCD-aware remap: blocks that share an XCD (bid % 8) get a contiguous run of tiles
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

