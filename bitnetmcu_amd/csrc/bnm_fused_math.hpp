// Per-tile arithmetic shared by the fused whole-model kernels: ReLUNorm on MFMA accumulators (packed straight into the
// next layer's B operand), first-maximum argmax, logits store.  gfx950 only.  Reference semantics:
// BitNetMCU_inference.c:23-72 (ReLUNorm).
#pragma once
#include "bnm_device.hpp"

// max(x, x of lane ^ 32): swapping the upper half of one copy with the lower half of another leaves
// {x_lo, x_lo} and {x_hi, x_hi}, whose maximum is the answer in every lane — no select on the half index
BNM_DEVICE int max_with_partner32(int x) {
    auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
    return max((int)r[0], (int)r[1]);
}

// clamp to [0, hi] in ONE instruction.  hipcc only forms v_med3_i32 from min(max(x, lo), hi) when it can prove
// lo <= hi (constants); with a run-time hi it emits v_max + v_min.
BNM_DEVICE int clamp0_med3(int x, int hi) {
    int r;
    asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(x), "v"(hi));
    return r;
}

// 16 clamped values -> 4 dwords, byte b of dword q = c[4q+b] >> s.  One SDWA shift per value writes its result
// byte straight into place (dst_sel:BYTE_b, dst_unused:UNUSED_PRESERVE), so no separate pack instructions.
// Same-register writes are 4 instructions apart and a trailing s_nop covers the dst_sel forwarding hazard that
// hipcc cannot see inside an asm statement.
BNM_DEVICE i32x4 sdwa_shift_pack16(const int (&c)[16], int s) {
    int d0, d1, d2, d3;
#define SD(dst, src, sel, unused) \
    "v_lshrrev_b32_sdwa " dst ", %4, " src " dst_sel:" sel " dst_unused:" unused " src0_sel:DWORD src1_sel:DWORD\n\t"
    asm(SD("%0", "%5", "BYTE_0", "UNUSED_PAD") SD("%1", "%9", "BYTE_0", "UNUSED_PAD")
        SD("%2", "%13", "BYTE_0", "UNUSED_PAD") SD("%3", "%17", "BYTE_0", "UNUSED_PAD")
        SD("%0", "%6", "BYTE_1", "UNUSED_PRESERVE") SD("%1", "%10", "BYTE_1", "UNUSED_PRESERVE")
        SD("%2", "%14", "BYTE_1", "UNUSED_PRESERVE") SD("%3", "%18", "BYTE_1", "UNUSED_PRESERVE")
        SD("%0", "%7", "BYTE_2", "UNUSED_PRESERVE") SD("%1", "%11", "BYTE_2", "UNUSED_PRESERVE")
        SD("%2", "%15", "BYTE_2", "UNUSED_PRESERVE") SD("%3", "%19", "BYTE_2", "UNUSED_PRESERVE")
        SD("%0", "%8", "BYTE_3", "UNUSED_PRESERVE") SD("%1", "%12", "BYTE_3", "UNUSED_PRESERVE")
        SD("%2", "%16", "BYTE_3", "UNUSED_PRESERVE") SD("%3", "%20", "BYTE_3", "UNUSED_PRESERVE")
        "s_nop 0"
        : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3)
        : "v"(s), "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]), "v"(c[4]), "v"(c[5]), "v"(c[6]), "v"(c[7]), "v"(c[8]),
          "v"(c[9]), "v"(c[10]), "v"(c[11]), "v"(c[12]), "v"(c[13]), "v"(c[14]), "v"(c[15]));
#undef SD
    i32x4 r = {d0, d1, d2, d3};
    return r;
}

// ReLUNorm (BitNetMCU_inference.c:23-72) on MT x 16 accumulator values per lane (+ the partner lane's),
// result packed as the next layer's B operand: packed[m][q] byte b = row 32m + 8q + 4h + b.
// Rows >= n_output are zero weights => value 0: they can only raise a negative maximum to 0, in which case
// every output is 0 either way.
//
// DBL = false: accumulators hold the layer sums x.   out = clamp((x + r) >> s, 0, 127), 3 VALU per value.
// DBL = true : this layer's weight fragments were built DOUBLED, accumulators hold 2x (exact).  With
//   s = bitlength(max(2x) >> 8) (= the reference's shift, from max(x) >> 7) and y = clamp(2x, 0, 255*2^s - 1) >> s
//   (0..254, one v_med3 + one SDWA shift that also packs), the rounded result is
//   (x + 2^(s-1)) >> s = (2x + 2^s) >> (s+1) = (y + 1) >> 1, which v_lerp_u8 computes for 4 bytes at once;
//   y <= 254 makes the "clip 128 to 127" case (:62-66) fall out.  2.25 VALU per value, bit-exact.
template <int MT, bool DBL, int NP = MT>
BNM_DEVICE void relunorm_pack(const i32x16 (&acc)[MT], i32x4 (&packed)[NP], int h) {
    static_assert(NP >= MT, "output array too short");
    int mx = acc[0][0];
#pragma unroll
    for (int m = 0; m < MT; m++)
#pragma unroll
        for (int r = 0; r < 16; r++) mx = max(mx, acc[m][r]);
    mx = max(max_with_partner32(mx), 0);
    if constexpr (DBL) {
        // shift = bitlength(mx >> 8) = bitlength(mx | 255) - 8: no zero test needed (mx >= 0)
        int sh = 24 - __builtin_clz((uint32_t)mx | 255u);
        int hi = (255 << sh) - 1;
#pragma unroll
        for (int m = 0; m < MT; m++) {
            int c[16];
#pragma unroll
            for (int r = 0; r < 16; r++) c[r] = clamp0_med3(acc[m][r], hi);
            i32x4 y = sdwa_shift_pack16(c, sh);
#pragma unroll
            for (int q = 0; q < 4; q++) packed[m][q] = (int)__builtin_amdgcn_lerp((uint32_t)y[q], 0u, 0x01010101u);
        }
    } else {
        // plain sums: out = min(127, (x + 2^(s-1)) >> s) for x >= 0, else 0 — add, v_med3 to [0, 128*2^s - 1], SDWA
        // shift straight into the packed byte: 3 VALU per value
        int sh = 25 - __builtin_clz((uint32_t)mx | 127u);     // bitlength(mx >> 7)
        int rnd = (1 << sh) >> 1;
        int hi = (128 << sh) - 1;
#pragma unroll
        for (int m = 0; m < MT; m++) {
            int c[16];
#pragma unroll
            for (int r = 0; r < 16; r++) c[r] = clamp0_med3(acc[m][r] + rnd, hi);
            packed[m] = sdwa_shift_pack16(c, sh);
        }
    }
}

// ---- relunorm_pack in SLICES ----------------------------------------------------------------------------------------
// The same arithmetic as relunorm_pack, cut into chunks of 8-9 VALU instructions with a call slot(k), k = 0, 1, ... behind each
// chunk (k is a std::integral_constant).  A kernel whose wave has its SIMD to itself (bnm_fused_regw.hip) issues one MFMA of an
// INDEPENDENT chain per slot and pins the order with sched_barrier: an MFMA occupies the matrix core for 32 clocks = 8 VALU
// issues, and a wave issues in order - two MFMAs back to back would stall its VALU work, a long VALU run leaves the matrix core
// idle.  Slots: MT (maximum, one per tile) + 1 (shift) + 4 MT (clamp / shift / pack, four values of each output dword's byte
// lane b per chunk) = 5 MT + 1.
template <int MT>
constexpr int relunorm_slots() { return 5 * MT + 1; }

// four values -> byte B of four dwords (sdwa_shift_pack16's instruction, one byte lane); consecutive chunks write the same
// registers 8+ instructions apart; the s_nop that covers the dst_sel forwarding hazard stands behind the last lane only
#define BNM_SD4(SEL, UNUSED, NOP)                                                                                                      \
    asm("v_lshrrev_b32_sdwa %0, %4, %5 dst_sel:" SEL " dst_unused:" UNUSED " src0_sel:DWORD src1_sel:DWORD\n\t"                        \
        "v_lshrrev_b32_sdwa %1, %4, %6 dst_sel:" SEL " dst_unused:" UNUSED " src0_sel:DWORD src1_sel:DWORD\n\t"                        \
        "v_lshrrev_b32_sdwa %2, %4, %7 dst_sel:" SEL " dst_unused:" UNUSED " src0_sel:DWORD src1_sel:DWORD\n\t"                        \
        "v_lshrrev_b32_sdwa %3, %4, %8 dst_sel:" SEL " dst_unused:" UNUSED " src0_sel:DWORD src1_sel:DWORD" NOP                           \
        : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(s), "v"(c0), "v"(c1), "v"(c2), "v"(c3))
template <int B>
BNM_DEVICE void sdwa_shift_pack4(int &d0, int &d1, int &d2, int &d3, int c0, int c1, int c2, int c3, int s) {
    if constexpr (B == 0) {
        asm("v_lshrrev_b32_sdwa %0, %4, %5 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n\t"
            "v_lshrrev_b32_sdwa %1, %4, %6 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n\t"
            "v_lshrrev_b32_sdwa %2, %4, %7 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n\t"
            "v_lshrrev_b32_sdwa %3, %4, %8 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD"
            : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3) : "v"(s), "v"(c0), "v"(c1), "v"(c2), "v"(c3));
    } else if constexpr (B == 1) BNM_SD4("BYTE_1", "UNUSED_PRESERVE", "");
    else if constexpr (B == 2) BNM_SD4("BYTE_2", "UNUSED_PRESERVE", "");
    else BNM_SD4("BYTE_3", "UNUSED_PRESERVE", "\n\ts_nop 0");      // (the byte lanes' only reader follows this one)
}
#undef BNM_SD4

template <int MT, bool DBL, int NP, class SLOT>
BNM_DEVICE void relunorm_pack_sliced(const i32x16 (&acc)[MT], i32x4 (&packed)[NP], int h, SLOT &&slot) {
    static_assert(NP >= MT, "output array too short");
    int mx = acc[0][0];
    static_for<0, MT>([&](auto M_) {
        constexpr int m = decltype(M_)::value;
#pragma unroll
        for (int r = 0; r < 16; r++) mx = max(mx, acc[m][r]);
        slot(std::integral_constant<int, m>{});
    });
    mx = max(max_with_partner32(mx), 0);
    // DBL: accumulators hold 2x; shift = bitlength(mx >> 8), clamp to [0, 255 * 2^sh - 1], >> sh, then (y + 1) >> 1 by v_lerp_u8
    // else: shift = bitlength(mx >> 7), (x + 2^(sh-1)) clamped to [0, 128 * 2^sh - 1], >> sh            (see relunorm_pack)
    const int sh = DBL ? 24 - __builtin_clz((uint32_t)mx | 255u) : 25 - __builtin_clz((uint32_t)mx | 127u);
    const int rnd = DBL ? 0 : (1 << sh) >> 1;
    const int hi = DBL ? (255 << sh) - 1 : (128 << sh) - 1;
    slot(std::integral_constant<int, MT>{});
    static_for<0, MT>([&](auto M_) {
        constexpr int m = decltype(M_)::value;
        int d0, d1, d2, d3;
        static_for<0, 4>([&](auto B_) {
            constexpr int b = decltype(B_)::value;
            int c[4];
#pragma unroll
            for (int q = 0; q < 4; q++) c[q] = clamp0_med3(DBL ? acc[m][4 * q + b] : acc[m][4 * q + b] + rnd, hi);
            sdwa_shift_pack4<b>(d0, d1, d2, d3, c[0], c[1], c[2], c[3], sh);
            if constexpr (b == 3) {
                if constexpr (DBL) {
                    packed[m][0] = (int)__builtin_amdgcn_lerp((uint32_t)d0, 0u, 0x01010101u);
                    packed[m][1] = (int)__builtin_amdgcn_lerp((uint32_t)d1, 0u, 0x01010101u);
                    packed[m][2] = (int)__builtin_amdgcn_lerp((uint32_t)d2, 0u, 0x01010101u);
                    packed[m][3] = (int)__builtin_amdgcn_lerp((uint32_t)d3, 0u, 0x01010101u);
                } else {
                    packed[m][0] = d0; packed[m][1] = d1; packed[m][2] = d2; packed[m][3] = d3;
                }
            }
            slot(std::integral_constant<int, MT + 1 + 4 * m + b>{});
        });
    });
}

// first strict maximum over the class rows (ReLUNorm's return value, :25-37).  key = value*256 + (255 - row):
// the largest key is the largest value and, among equals, the smallest row.  |value| < 2^23 for every layer that
// can be last (K <= 128, |act| <= 127, |w| <= 128).
// No run-time row masks: the fragment builder fills the last layer's padding rows (row >= n_classes) with weight
// -128 on every real input column, so a padding row's sum is -128 * sum(act) <= every real row's sum (act >= 0,
// w >= -128) and on a tie the real row, having the smaller index, wins.  NC8 > 0 states at compile time that
// n_classes <= 8 * NC8, so accumulator registers holding only rows >= 8 * NC8 are not looked at at all
// (10 classes: 8 of 16 registers); NC8 == 0 looks at every register.  1.5 VALU per register examined.
template <int MT, int NC8>
BNM_DEVICE uint32_t argmax_rows(const i32x16 (&acc)[MT], int h) {
    constexpr int G = NC8 > 0 ? NC8 : 4 * MT;
    int best = INT_MIN;
    // the low byte 255 - row is assembled per 32-row tile: 31 - (row in tile) is an inline constant of the per-value
    // v_lshl_or_b32, the tile's 224 - 32 m is added once to the tile's maximum.  (Written as 255 - row per value, every row
    // needs its own constant above 64 in a register: dozens of registers in the kernels that serve up to 256 classes.)
#pragma unroll
    for (int m = 0; m < MT; m++) {
        if (4 * m >= G) continue;
        int bm = INT_MIN;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            if (4 * m + (r >> 2) >= G) continue;
            const uint32_t rb = (r & 3) + 8u * (r >> 2);          // row within the tile for the h = 0 half; h = 1: +4
            bm = max(bm, (int)(((uint32_t)acc[m][r] << 8) | (31u - rb)));
        }
        best = max(best, bm + (224 - 32 * m));
    }
    best = max_with_partner32(best - 4 * h);
    return 255u - ((uint32_t)best & 255u);
}

// A lane holds, per tile and per group q of four accumulator registers, the four CONSECUTIVE rows 32m + 8q + 4h .. +3 of its
// image: they leave as one 16-byte store (8 bytes when exactly two of them are classes, e.g. rows 8-9 of 10), element-wise only
// for other ragged ends.  (Sixteen predicated dword stores per tile cost the logits configuration ~10 % of its time.)
template <int MT>
BNM_DEVICE void store_logits(const i32x16 (&acc)[MT], int32_t *dst, int h, uint32_t n_classes) {
    // (a logits row starts at 4 n_classes bytes x the image index: dword-aligned only, which is all a global b64 / b128 store needs)
    typedef int v4a4 __attribute__((ext_vector_type(4), aligned(4)));
    typedef int v2a4 __attribute__((ext_vector_type(2), aligned(4)));
#pragma unroll
    for (int m = 0; m < MT; m++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t row = 32u * m + 8u * q + 4u * (uint32_t)h;
            if (row + 4u <= n_classes) {
                *(v4a4 *)(dst + row) = v4a4{acc[m][4 * q], acc[m][4 * q + 1], acc[m][4 * q + 2], acc[m][4 * q + 3]};
            } else if (row + 2u == n_classes) {
                *(v2a4 *)(dst + row) = v2a4{acc[m][4 * q], acc[m][4 * q + 1]};
            } else {
#pragma unroll
                for (int e = 0; e < 4; e++)
                    if (row + (uint32_t)e < n_classes) dst[row + e] = acc[m][4 * q + e];
            }
        }
}

// A WHOLE tile's logits (32 valid images, n_classes <= 16) through a per-wave LDS staging area laid out as the tile is in memory
// (image-major, n_classes dwords per image): the tile then leaves as contiguous 16 B/lane stores, 1 KiB per instruction and every
// 128-byte line written whole by one instruction - store_logits' 16-byte pieces at a 4 n_classes-byte stride fill a line from two
// instructions, which only a write-back cache merges.  `stage` = 32 * 16 dwords of LDS owned by the wave (LDS executes a wave's
// instructions in order: the reads below see the writes above, and the next tile's writes come after these reads).
// MODE 0: nontemporal stores, 1: plain stores.
template <int MT, int NC8, int MODE>
BNM_DEVICE void store_logits_tile(const i32x16 (&acc)[MT], int32_t *stage, int32_t *tile_dst, int j, int h, int lane, uint32_t n_classes) {
    int32_t *const mine = stage + (uint32_t)j * n_classes + 4u * (uint32_t)h;
#pragma unroll
    for (int m = 0; m < MT; m++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            if (NC8 > 0 && 4 * m + q >= NC8) continue;      // rows 32m + 8q .. +7 lie beyond the classes
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const uint32_t row = 32u * m + 8u * q + 4u * (uint32_t)h + (uint32_t)e;
                if (row < n_classes) mine[32 * m + 8 * q + e] = acc[m][4 * q + e];
            }
        }
    const uint32_t chunks = 8u * n_classes;                   // 16-byte pieces of the tile: 32 images x 4 n_classes bytes
#pragma unroll
    for (int r = 0; r < 2; r++) {
        const uint32_t c = (uint32_t)lane + 64u * r;
        if (c < chunks) {
            const i32x4 v = *(const i32x4 *)(stage + 4u * c);
            if (MODE == 0) __builtin_nontemporal_store(v, (i32x4 *)(tile_dst + 4u * c));
            else *(i32x4 *)(tile_dst + 4u * c) = v;
        }
    }
}

