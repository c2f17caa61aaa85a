// Shared device-side definitions for the gfx950 kernels (wave64, MFMA i8 32x32x32, 160 KiB LDS).
#pragma once
#include <hip/hip_runtime.h>
#include <climits>
#include <cstdint>
#include <type_traits>
#include "bnm_kernels.h"

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define BNM_DEVICE __device__ __forceinline__
// compile-time counted loop: the body receives std::integral_constant<int, I>, so every array index derived
// from I is a constant and the arrays stay in registers
template <int B, int E, class F>
BNM_DEVICE void static_for(F &&f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}


BNM_DEVICE uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// number of compute units of the current device (cached); grids are sized as multiples of it
inline int bnm_num_cus() {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) cus = p.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    return cus;
}

// =================================================================================================
// Weight codecs (exportquant.py:104-187 packs, BitNetMCU_inference.c:96-201 unpacks).
// decode_weight() is the single device-side statement of all codecs; both the unpack kernel and the
// bit-serial layer kernel go through it.
// =================================================================================================
BNM_DEVICE int codec_field_bits(int bpw) {
    return bpw == 1 ? 1 : bpw == 2 ? 2 : (bpw == 4 || bpw == 12 || bpw == 20) ? 4 : bpw == 16 ? 8 : 0;
}

// field f (fb bits, already right-aligned) -> integer weight
BNM_DEVICE int decode_field(int bpw, uint32_t f) {
    switch (bpw) {
        case 1: return f ? 1 : -1;                                          // :96-104
        case 2: return ((f & 2u) ? -1 : 1) * (int)(1u + 2u * (f & 1u));     // :105-115
        case 4: return ((f & 8u) ? -1 : 1) * (int)(2u * (f & 7u) + 1u);     // :156-168
        case 12: return (int)(f ^ 8u) - 8;                                  // :169-178
        case 16: return (int)(int8_t)f;                                     // :179-188
        case 20: return ((f & 8u) ? -1 : 1) * (int)(1u << (f & 7u));        // :190-201
    }
    return 0;
}

// trit t (0..9) of a 16-bit ternary chunk (:116-136): multiply-by-3 pops digits MSB first
BNM_DEVICE int ternary_trit(uint32_t chunk, uint32_t t) {
    uint32_t digit = 0;
    for (uint32_t i = 0; i <= t; i++) {
        chunk *= 3u;
        digit = chunk >> 16;
        chunk &= 0xFFFFu;
    }
    return digit == 0 ? 1 : (digit == 1 ? -1 : 0);
}

BNM_DEVICE int decode_weight(const void *packed, int bpw, uint32_t n_input, uint32_t row, uint32_t k) {
    if (bpw == 64) {
        uint32_t chunk = ((const uint16_t *)packed)[row * (n_input / 10u) + k / 10u];
        return ternary_trit(chunk, k % 10u);
    }
    int fb = codec_field_bits(bpw);
    if (!fb) return 0;
    uint32_t per_word = 32u / (uint32_t)fb;
    uint32_t words_per_row = (n_input + per_word - 1u) / per_word;
    uint32_t word = ((const uint32_t *)packed)[row * words_per_row + k / per_word];
    uint32_t f = (word >> (32u - (uint32_t)fb * (k % per_word + 1u))) & ((1u << fb) - 1u);
    return decode_field(bpw, f);
}


// =================================================================================================
// Device-wide work counter for persistent kernels, on the SCALAR unit.
// Why: the SIMD arbiter favours its oldest wave, so waves with equal fixed shares finish one after the other and the end of a
// launch runs at one wave per SIMD ("work distribution").  Handing the work out from a counter keeps every wave
// busy until the work runs out.  s_atomic_add ... glc returns the counter's previous value in a scalar register: no VGPR, no
// EXEC change, no entry in the vmcnt queue that the LDS-DMA kernels count by hand.  profiles/probes/s_atomic_probe.hip: on
// gfx950 it is coherent across the whole device (512 workgroups x 8 waves x 64 takes on one word: no duplicate, none lost).
// One word serves ~88 M takes/s, eight words ~430 M/s, a word per wave 6 G/s at 341 ns per take (probes/s_atomic_rate.hip), so the
// fused kernels split their counter eight ways (wave w takes from word w mod 8: every word is shared by waves of all CUs).
// The take is split into issue and wait so that its round trip (0.3-2 us) runs under the caller's arithmetic; in between the
// result register must not move, so both statements name the SAME fixed register (s95; the streamed ternary kernels use the
// same technique for their weight buffers).  The compiler's own lgkmcnt waits merely become conservative while the take is
// outstanding (the counter is shared with LDS operations).  The counter words are zero at every launch (work_block_leave_*).
// =================================================================================================
BNM_DEVICE void work_take_issue(uint32_t &r, uint32_t *counter, uint32_t amount) {
    asm volatile("s_mov_b32 %0, %2\n\ts_atomic_add %0, %1, 0x0 glc" : "=&{s95}"(r) : "s"(counter), "s"(amount) : "memory");
}
BNM_DEVICE void work_take_wait(uint32_t &r) { asm volatile("s_waitcnt lgkmcnt(0)" : "+{s95}"(r)::"memory"); }

// =================================================================================================
// A launch's counter BLOCK (BNM_WORK_BLOCK_WORDS device words owned by the launch's stream, bnm_capi_ctx.cpp): the counter words
// at block[16 k], k < 8, and at block[BNM_WORK_EXIT_WORD] the number of waves that have left the kernel.  A block is all zero
// between launches: nobody zeroes it ahead of a launch (that was a hipMemsetAsync per launch: a second dispatch, 2 us of a
// launch-bound call's 6-9) - the LAST wave to leave puts it back.  A leaving wave first retires every take it still has in
// flight (counter operations of one wave reach the memory side in no particular order), then counts itself out; the wave that
// finds everybody else gone swaps zeros into the nine words.  Launches that share a block are ordered by their stream.
// _s: kernels whose takes are scalar atomics (fused FC kernels); _v: kernels whose takes are vector atomics of lane 0.
// =================================================================================================
BNM_DEVICE void work_block_leave_s(uint32_t *block, uint32_t total_waves) {
    uint32_t old;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\t"
                 "s_mov_b32 %0, 1\n\t"
                 "s_atomic_add %0, %1, %2 glc\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&s"(old) : "s"(block), "n"(4 * BNM_WORK_EXIT_WORD) : "memory");
    if (old + 1u == total_waves) {
        uint32_t z0, z1, z2;
        asm volatile("s_mov_b32 %0, 0\n\ts_mov_b32 %1, 0\n\ts_mov_b32 %2, 0\n\t"
                     "s_atomic_swap %0, %3, 0x0 glc\n\ts_atomic_swap %1, %3, 0x40 glc\n\ts_atomic_swap %2, %3, 0x80 glc\n\t"
                     "s_waitcnt lgkmcnt(0)\n\t"
                     "s_mov_b32 %0, 0\n\ts_mov_b32 %1, 0\n\ts_mov_b32 %2, 0\n\t"
                     "s_atomic_swap %0, %3, 0xc0 glc\n\ts_atomic_swap %1, %3, 0x100 glc\n\ts_atomic_swap %2, %3, 0x140 glc\n\t"
                     "s_waitcnt lgkmcnt(0)\n\t"
                     "s_mov_b32 %0, 0\n\ts_mov_b32 %1, 0\n\ts_mov_b32 %2, 0\n\t"
                     "s_atomic_swap %0, %3, 0x180 glc\n\ts_atomic_swap %1, %3, 0x1c0 glc\n\ts_atomic_swap %2, %3, %4 glc\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&s"(z0), "=&s"(z1), "=&s"(z2) : "s"(block), "n"(4 * BNM_WORK_EXIT_WORD) : "memory");
    }
}
BNM_DEVICE void work_block_leave_v(uint32_t *block, uint32_t total_waves) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const uint32_t lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    uint32_t old = 0;
    if (lane == 0) old = __hip_atomic_fetch_add(block + BNM_WORK_EXIT_WORD, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    old = (uint32_t)__builtin_amdgcn_readfirstlane((int)old);
    if (old + 1u == total_waves && lane < 9u)
        (void)__hip_atomic_exchange(block + 16u * lane, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
