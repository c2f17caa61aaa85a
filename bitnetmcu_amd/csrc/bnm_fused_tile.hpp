// Image-tile transport shared by the fused whole-model kernels (and the diagnostic stream kernels): LDS-DMA pieces,
// counted vmcnt waits, tile geometry.  gfx950 only.
#pragma once
#include "bnm_device.hpp"

BNM_DEVICE i32x16 zero16() {
    i32x16 z;
#pragma unroll
    for (int i = 0; i < 16; i++) z[i] = 0;
    return z;
}


// 8 x 1 KiB LDS-DMA pieces of one 32-image tile.  p[t] wave-uniform base pointers, v[t] per-lane byte
// offsets, lds wave-uniform LDS byte address of the tile buffer.  The DMA destination is
// M0 + lane*16 (lane-linear); the swizzle lives in v[].  hipcc neither counts these loads nor waits for
// them: the caller retires them with bnm_wait_vmcnt<N>().
// NT: non-temporal policy (the image stream is read exactly once).  WAITLDS: first retire this wave's own
// outstanding ds_reads (s_waitcnt lgkmcnt(0)) — needed when the destination buffer was being read just before.
#define BNM_DMA8(NTS, PRE)                                                                                           \
    asm volatile(PRE "s_nop 4\n\t"                                                                                    \
                 "s_mov_b32 %0, m0\n\t"                                                                                \
                 "s_mov_b32 m0, %1\n\t"                                                                                \
                 "s_nop 0\n\t"                                                                                         \
                 "global_load_lds_dwordx4 %10, %2" NTS "\n\t"                                                          \
                 "s_add_u32 m0, m0, 0x400\n\t"                                                                         \
                 "s_nop 0\n\t"                                                                                         \
                 "global_load_lds_dwordx4 %11, %3" NTS "\n\t"                                                          \
                 "s_add_u32 m0, m0, 0x400\n\t"                                                                         \
                 "s_nop 0\n\t"                                                                                         \
                 "global_load_lds_dwordx4 %12, %4" NTS "\n\t"                                                          \
                 "s_add_u32 m0, m0, 0x400\n\t"                                                                         \
                 "s_nop 0\n\t"                                                                                         \
                 "global_load_lds_dwordx4 %13, %5" NTS "\n\t"                                                          \
                 "s_add_u32 m0, m0, 0x400\n\t"                                                                         \
                 "s_nop 0\n\t"                                                                                         \
                 "global_load_lds_dwordx4 %14, %6" NTS "\n\t"                                                          \
                 "s_add_u32 m0, m0, 0x400\n\t"                                                                         \
                 "s_nop 0\n\t"                                                                                         \
                 "global_load_lds_dwordx4 %15, %7" NTS "\n\t"                                                          \
                 "s_add_u32 m0, m0, 0x400\n\t"                                                                         \
                 "s_nop 0\n\t"                                                                                         \
                 "global_load_lds_dwordx4 %16, %8" NTS "\n\t"                                                          \
                 "s_add_u32 m0, m0, 0x400\n\t"                                                                         \
                 "s_nop 0\n\t"                                                                                         \
                 "global_load_lds_dwordx4 %17, %9" NTS "\n\t"                                                          \
                 "s_mov_b32 m0, %0"                                                                                    \
                 : "=&s"(keep)                                                                                         \
                 : "s"(lds), "s"(p0), "s"(p1), "s"(p2), "s"(p3), "s"(p4), "s"(p5), "s"(p6), "s"(p7), "v"(v0), "v"(v1), \
                   "v"(v2), "v"(v3), "v"(v4), "v"(v5), "v"(v6), "v"(v7)                                                \
                 : "memory", "scc")

template <bool NT = false, bool WAITLDS = false>
BNM_DEVICE void lds_dma_tile8(uint32_t lds, const int8_t *p0, const int8_t *p1, const int8_t *p2, const int8_t *p3,
                              const int8_t *p4, const int8_t *p5, const int8_t *p6, const int8_t *p7, uint32_t v0,
                              uint32_t v1, uint32_t v2, uint32_t v3, uint32_t v4, uint32_t v5, uint32_t v6,
                              uint32_t v7) {
    uint32_t keep;
    if constexpr (NT && WAITLDS) BNM_DMA8(" nt", "s_waitcnt lgkmcnt(0)\n\t");
    else if constexpr (NT) BNM_DMA8(" nt", "");
    else if constexpr (WAITLDS) BNM_DMA8("", "s_waitcnt lgkmcnt(0)\n\t");
    else BNM_DMA8("", "");
}

template <int N>
BNM_DEVICE void bnm_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

constexpr int FUSED_TILE_BYTES = 8192;    // 32 images x 256 B
constexpr int FUSED_WPB = 4;              // waves per workgroup; two workgroups per CU (LDS 64 KiB each)

// whole 32-image tile, rows contiguous: two base pointers + instruction offsets instead of eight pointers.
// The instruction offset of an LDS-DMA load is added to BOTH the global and the LDS address, so pieces 0..3 and
// 4..7 need M0 set only once each.
#define BNM_DMA8_LINEAR(POL)                                                                       \
    asm volatile("s_waitcnt lgkmcnt(0)\n\t"                                                        \
                 "s_nop 4\n\t"                                                                      \
                 "s_mov_b32 %0, m0\n\t"                                                             \
                 "s_mov_b32 m0, %1\n\t"                                                             \
                 "s_nop 0\n\t"                                                                      \
                 "global_load_lds_dwordx4 %4, %2" POL "\n\t"                                        \
                 "global_load_lds_dwordx4 %5, %2 offset:1024" POL "\n\t"                            \
                 "global_load_lds_dwordx4 %6, %2 offset:2048" POL "\n\t"                            \
                 "global_load_lds_dwordx4 %7, %2 offset:3072" POL "\n\t"                            \
                 "s_add_u32 m0, m0, 0x1000\n\t"                                                     \
                 "s_nop 0\n\t"                                                                      \
                 "global_load_lds_dwordx4 %4, %3" POL "\n\t"                                        \
                 "global_load_lds_dwordx4 %5, %3 offset:1024" POL "\n\t"                            \
                 "global_load_lds_dwordx4 %6, %3 offset:2048" POL "\n\t"                            \
                 "global_load_lds_dwordx4 %7, %3 offset:3072" POL "\n\t"                            \
                 "s_mov_b32 m0, %0"                                                                  \
                 : "=&s"(keep)                                                                       \
                 : "s"(lds), "s"(lo), "s"(hi), "v"(v0), "v"(v1), "v"(v2), "v"(v3)                    \
                 : "memory", "scc")
// Cache policy: nt (non-temporal).  Round 1 also measured sc1 nt, sc0 sc1 nt, sc1 and no hint on the same box
// (profiles/r01/r01n_cache_policy_experiment.log): all within the run-to-run spread, none better than nt.
BNM_DEVICE void lds_dma_tile8_linear(uint32_t lds, const int8_t *lo, const int8_t *hi, uint32_t v0, uint32_t v1,
                                     uint32_t v2, uint32_t v3) {
    uint32_t keep;
    BNM_DMA8_LINEAR(" nt");
}

