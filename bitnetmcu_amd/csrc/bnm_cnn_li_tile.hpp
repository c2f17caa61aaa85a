// Helpers of the lane = image CNN front end shared by cnn_li_kernel (bnm_cnn_li.hip: act rows to HBM, the FC tail as its own
// launch) and cnn_li_fused_kernel (bnm_cnn_li_fused.hip: the FC tail inside the same wave).  The per-tile arithmetic itself - the
// three Toeplitz convolutions + pools of one 32-image tile over all channels - is the statement sequence in
// bnm_cnn_li_tile_body.inc, included textually by both kernels (as a function it cost cnn_li_kernel, which sits at its 128
// registers, a spill).  See bnm_cnn_li.hip for the formulation.  gfx950 only.
#pragma once
#include "bnm_fused_math.hpp"

namespace {

constexpr int LI_WAVES = 16;         // up to four waves per SIMD (128 VGPRs); fewer when the records of a wide model fill the LDS

BNM_DEVICE i32x16 mfma0(const i32x4 &a, const i32x4 &b) { return __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, i32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, 0, 0, 0); }
BNM_DEVICE i32x16 mfma(const i32x4 &a, const i32x4 &b, const i32x16 &c) { return __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0); }

typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2v __attribute__((ext_vector_type(2)));
// relu of two 15-bit values as one packed pair: [max(a, 0) | max(b, 0) << 16]  (v_cvt_pk_i16_i32 + v_pk_max_i16)
BNM_DEVICE uint32_t relu_pair16(int a, int b) {
    const s16x2 p = __builtin_amdgcn_cvt_pk_i16(a, b), z = {0, 0};
    const s16x2 r = __builtin_elementwise_max(p, z);
    return __builtin_bit_cast(uint32_t, r);
}

// Two pairs of conv1 sums -> two packed int16 pairs, xa = [a0 >> 4 | (a1 >> 4) << 16], xb likewise: ONE SDWA shift per value writes its
// low 16 bits (the shifted sums fit 15) straight into its half of the pair - no v_cvt_pk_i16_i32.  Inline asm, so three things hipcc
// would do for its own instructions are done by hand: (1) the statement is ordered BEHIND a compiler-visible reader of the same MFMA
// result (`after` is that reader's output, an unused input here): hipcc has then placed the MFMA -> VALU wait states, and the MFMA's
// write-back - its dead rows included - is over before an output register of this statement can be written; (2) the two writes of one
// register are an instruction apart and (3) an s_nop closes the statement (dst_sel forwarding, as in sdwa_shift_pack16).
// profiles/check_mfma_hazards.py (a static test) walks the disassembly for the MFMAs issued AFTER the reader.
BNM_DEVICE void shr4_pack_pairs2(int a0, int a1, int b0, int b1, uint32_t after, uint32_t &xa, uint32_t &xb) {
    asm("v_ashrrev_i32_sdwa %0, %6, %2 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n\t"
        "v_ashrrev_i32_sdwa %1, %6, %4 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n\t"
        "v_ashrrev_i32_sdwa %0, %6, %3 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
        "v_ashrrev_i32_sdwa %1, %6, %5 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
        "s_nop 0"
        : "=&v"(xa), "=&v"(xb) : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "s"(4), "v"(after));
}
// The anchor pair of a row pair: a0 >> 4 is a compiler-visible shift (hipcc places the MFMA -> VALU wait states in front of it), the
// second value joins it through one SDWA shift into the upper half: two instructions for the pair.
BNM_DEVICE uint32_t shr4_pack_anchor(int a0, int a1) {
    int t = a0 >> 4;
    asm("v_ashrrev_i32_sdwa %0, %2, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
        "s_nop 0"
        : "+v"(t) : "v"(a1), "s"(4));
    return (uint32_t)t;
}
BNM_DEVICE uint32_t relu_pk16(uint32_t p) {
    const s16x2 z = {0, 0};
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, p), z));
}

// Byte planes of pooled values for conv3's operand.  A pooled value arrives as V = 16 x (its 24-bit ReLU'd sum) = (P << 8) | low
// bits: plane p of P is byte p + 1 of V, and v_perm_b32 gathers bytes of two registers - 7 instructions for the three planes of
// four values (two pair gathers per pair, one merge per plane).  Compiler-visible on purpose: an earlier inline-asm version (SDWA
// byte writes) had its outputs allocated to the DEAD rows of an MFMA result still in flight - the padding quad no one reads - and
// hipcc places no hazard wait in front of inline asm: the late MFMA write-back then zeroed a plane-2 dword, about one image in
// 50,000 at full-range weights and only at four waves per SIMD.
struct PlaneQuad { int p0, p1, p2; };
BNM_DEVICE PlaneQuad plane_quad(int A, int B, int C, int D) {
    const uint32_t ab01 = __builtin_amdgcn_perm((uint32_t)B, (uint32_t)A, 0x06020501u);      // [A.1, B.1, A.2, B.2]
    const uint32_t cd01 = __builtin_amdgcn_perm((uint32_t)D, (uint32_t)C, 0x06020501u);
    const uint32_t ab2 = __builtin_amdgcn_perm((uint32_t)B, (uint32_t)A, 0x0c0c0703u);       // [A.3, B.3, 0, 0]
    const uint32_t cd2 = __builtin_amdgcn_perm((uint32_t)D, (uint32_t)C, 0x0c0c0703u);
    return PlaneQuad{(int)__builtin_amdgcn_perm(cd01, ab01, 0x05040100u), (int)__builtin_amdgcn_perm(cd01, ab01, 0x07060302u),
                     (int)__builtin_amdgcn_perm(cd2, ab2, 0x05040100u)};
}
BNM_DEVICE PlaneQuad plane_pair(int A, int B) {      // two values: bytes 0, 1 of the dword, the rest zero
    const uint32_t ab01 = __builtin_amdgcn_perm((uint32_t)B, (uint32_t)A, 0x06020501u);
    return PlaneQuad{(int)__builtin_amdgcn_perm(0u, ab01, 0x0c0c0100u), (int)__builtin_amdgcn_perm(0u, ab01, 0x0c0c0302u),
                     (int)__builtin_amdgcn_perm((uint32_t)B, (uint32_t)A, 0x0c0c0703u)};
}

}  // namespace

// Lane values derived afresh per tile, twice, from an opaque copy of tid: hoisted out of the tile loop, as the compiler would,
// those values are live across the channel loop, and at four waves per SIMD (128 VGPRs) that is what spilled.
// Expects tid, wave, C, li_records in scope; defines lane, j, h, rec, rec_k.
#define LI_LANE_VALUES                                                                       \
    uint32_t lane_ = tid;                                                                    \
    asm volatile("" : "+v"(lane_));                                                          \
    const int lane = (int)(lane_ & 63u), j = lane & 31, h = lane >> 5;                       \
    uint16_t *const rec = (uint16_t *)(li_records + wave * C * 160u) + lane;                 \
    uint8_t *const rec_k = li_records + wave * C * 160u + C * 128u + j;
