// Resident single-wave inference kernel behind the drop-in `Inference()` symbol (opt-in: BNM_PERSISTENT=1).
// gfx950 only.
//
// The reference's harness calls `Inference(int8_t *input)` once per image, synchronously, 10,000 times (test_inference.py:136-168;
// BitNetMCU_MNIST_dll.c:24-26).  A kernel launch per call costs the launch itself: ~20 us of the call's 22, against 26 us for the
// reference's CPU code.  Here ONE wave stays resident and a call is a message:
//   host  -> writes the image into a page-locked mailbox: five 64-byte lines of fifteen image dwords + the call's sequence number
//            as the line's LAST dword.  A line is read atomically over PCIe and x86 stores become visible in program order, so a
//            line whose tag carries the number also carries its fifteen dwords: the kernel polls all five lines with ONE pair of
//            loads and the image is in its registers the moment the last tag matches - one PCIe round trip, no second read;
//   wave  -> image dwords to row 0 of its LDS tile, the FC stack of the generic kernel on the tile (weights resident in LDS since
//            the kernel started), the class id and the sequence number back to the mailbox in ONE 32-bit store;
//   host  -> spins on that word.
// The wave leaves by itself when no call arrived for `idle_ticks` of the 100 MHz wall clock (a few milliseconds: a
// hipDeviceSynchronize() anywhere in the process waits no longer than that) or when the host raises the mailbox's quit word; the
// host notices a kernel that has left (hipStreamQuery on the slow path of its spin) and starts another one, which finds the pending
// call in the mailbox.  Serves the models the fused float-input kernel serves (256-byte rows, tile classes 2 / 4 / 6).
#pragma once
#include "bnm_fused_generic_kernel.hpp"

// mailbox layout, in dwords (the host side - bnm_capi_host.cpp - uses the same constants)
constexpr uint32_t BNM_BOX_LINES = 5;         // request: dwords 16 L + k, k < 15 = image dword 15 L + k; dword 16 L + 15 = the tag
constexpr uint32_t BNM_BOX_QUIT = 80;         // nonzero: leave
constexpr uint32_t BNM_BOX_RESPONSE = 96;     // (sequence number << 8) | class id
constexpr uint32_t BNM_BOX_STAMPS = 112;      // after the answer: what the call took inside the wave, tags seen -> answer stored: [100 MHz wall-clock ticks, shader clocks]
constexpr uint32_t BNM_BOX_DWORDS = 128;

template <int MMAX, int SP, bool DBL>
__global__ __launch_bounds__(64) void persistent_inference_kernel(const i32x4 *__restrict__ frags, BnmGenericDesc d, uint32_t *box,
                                                                  uint32_t seq0, uint64_t idle_ticks) {
    constexpr int KT0 = 8, T = 1;
    using G = RowGeom<256>;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int lane = threadIdx.x & 63;
    for (uint32_t o = (uint32_t)lane * 16u; o < d.w_bytes; o += 64u * 16u) *(i32x4 *)(smem + o) = frags[o >> 4];
    const uint32_t tile_off = d.w_bytes;
    for (uint32_t o = (uint32_t)lane * 16u; o < (uint32_t)G::TILE; o += 64u * 16u) *(i32x4 *)(smem + tile_off + o) = i32x4{0, 0, 0, 0};
    __syncthreads();
    const uint32_t l16 = 16u * (uint32_t)lane;
    const int j = lane & 31, h = lane >> 5;
    const uint32_t rd = tile_off + (uint32_t)j * 256u + 16u * ((uint32_t)h ^ G::mask((uint32_t)j));
    // where this lane's polled dwords belong in row 0 of the tile (mask(0) = 0: row 0 is stored unswizzled); tags and pad go nowhere
    const uint32_t k = (uint32_t)lane & 15u;
    const bool payload_a = k < 15u, payload_b = (uint32_t)lane < 4u;
    const uint32_t dst_a = tile_off + 4u * (15u * ((uint32_t)lane >> 4) + k), dst_b = tile_off + 4u * (60u + (uint32_t)lane);
    const uint32_t M1 = d.M[0], M2 = d.M[1], M3 = d.M[2], M4 = d.M[3];
    const bool few_classes = d.n_classes <= 16u;
    const bool uniform = M2 == M1 && (M4 ? (M3 == M1 && M4 == 1u) : M3 == 1u) && M1 + 1u >= (uint32_t)MMAX;
    uint32_t last = seq0;
    uint64_t t0 = wall_clock64();
    for (;;) {
        const uint32_t want = last == 0xFFFFFFu ? 1u : last + 1u;      // 24-bit sequence numbers, 0 never used
        // ---- poll: lines 0..3 in `a` (lane = dword), line 4 and the control line in `b` (lanes 0..31) ------------------------------
        const uint32_t a = __hip_atomic_load(box + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const uint32_t b = __hip_atomic_load(box + 64 + (lane & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const bool tags = __builtin_amdgcn_ballot_w64(k != 15u || a == want) == ~0ull &&
                          (uint32_t)__builtin_amdgcn_readlane((int)b, 15) == want;
        if (!tags) {
            if (__builtin_amdgcn_readlane((int)b, 16) != 0 || wall_clock64() - t0 > idle_ticks) break;
            __builtin_amdgcn_s_sleep(1);
            continue;
        }
        const uint64_t w0 = wall_clock64(), c0 = clock64();
        if (payload_a) *(uint32_t *)(smem + dst_a) = a;
        if (payload_b) *(uint32_t *)(smem + dst_b) = b;
        // ---- the FC stack on the tile (the generic kernel's code; image 0 is the call's, rows 1..31 are zero) -----------------------
        i32x4 act[T][MMAX];
        uint32_t nc = d.n_classes;
        asm volatile("" : "+v"(nc));
        uint32_t cls[T] = {0};
        bool done = false;
        static_for<1, MMAX + 1>([&](auto MI) {
            constexpr int mt = decltype(MI)::value;
            if (M1 == (uint32_t)mt) {
                i32x16 acc[T][mt];
                i32x4 b0[T][KT0];
#pragma unroll
                for (int s = 0; s < KT0; s++) b0[0][s] = *(const i32x4 *)(smem + (rd ^ (32u * (uint32_t)s)));
                mma_l1<mt, KT0, 0, KT0, SP, true, T>(smem + (d.frag_off[0] + l16), b0, acc);
                relunorm_pack<mt, DBL, MMAX>(acc[0], act[0], h);
                if constexpr (mt >= MMAX - 1) {
                    if (uniform) {
                        {
#ifdef BNM_DIAG_TIMING
                        PhaseStamps st{};
#endif
                        uniform_tail<mt, MMAX, SP, DBL, T>(smem, l16, d, act, h, j, lane, cls, nullptr, nullptr, 0ull, 1ull, nc, few_classes BNM_ST_ARG);
                        }
                        done = true;
                    }
                }
            }
        });
        if (!done) {
            hidden_layer<MMAX, SP, DBL, T>(smem, l16, d.frag_off[1], M2, M1, act, h);
            uint32_t m_last = M3, k_last = M2, off_last = d.frag_off[2];
            if (M4) {
                hidden_layer<MMAX, SP, DBL, T>(smem, l16, d.frag_off[2], M3, M2, act, h);
                m_last = M4; k_last = M3; off_last = d.frag_off[3];
            }
            final_layer<MMAX, SP, T>(smem, l16, off_last, m_last, k_last, act, h, j, lane, cls, nullptr, nullptr, 0ull, 1ull, nc, few_classes);
        }
        // ---- the answer: one 32-bit store (lane 0 holds image 0's class id) ------------------------------------------------------
        if (lane == 0) __hip_atomic_store(box + BNM_BOX_RESPONSE, (want << 8) | (cls[0] & 0xFFu), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        last = want;
        t0 = wall_clock64();
        if (lane == 0) {      // (behind the answer: the host is already on its way)
            box[BNM_BOX_STAMPS] = (uint32_t)(t0 - w0);
            box[BNM_BOX_STAMPS + 1] = (uint32_t)(clock64() - c0);
        }
    }
}

// ---- launcher of one tile class (in the class's float-input translation unit: bnm_fused_f32_m{2,4,6}.hip) ---------------------------
#define BNM_PERSIST_LAUNCHER(NAME, MMAX)                                                                                       \
    hipError_t NAME(uint32_t sp, bool dbl, bool launch, unsigned lds, hipStream_t s, const void *frags, const BnmGenericDesc &d, \
                    uint32_t *box, uint32_t seq0, uint64_t idle_ticks) {                                                         \
        typedef void (*fn_t)(const i32x4 *, BnmGenericDesc, uint32_t *, uint32_t, uint64_t);                                    \
        fn_t fn = nullptr;                                                                                                     \
        if (sp == 1 && dbl) fn = persistent_inference_kernel<MMAX, 1, true>;                                                   \
        else if (sp == 1) fn = persistent_inference_kernel<MMAX, 1, false>;                                                    \
        else if (sp == 2 && !dbl) fn = persistent_inference_kernel<MMAX, 2, false>;                                            \
        if (!fn) return hipErrorInvalidValue;                                                                                  \
        if (!launch) return hipSuccess;                                                                                        \
        if (hipError_t err = bnm_generic_allow_big_lds((const void *)fn); err != hipSuccess) return err;                       \
        fn<<<dim3(1), dim3(64), lds, s>>>((const i32x4 *)frags, d, box, seq0, idle_ticks);                                     \
        return hipGetLastError();                                                                                              \
    }
