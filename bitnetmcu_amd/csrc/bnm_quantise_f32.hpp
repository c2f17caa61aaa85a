// Float32 -> int8 input quantisation, the per-value arithmetic shared by the fused float-input kernels (bnm_fused_f32_kernel.hpp: FC
// models; bnm_cnn_li_fused.hip: the one-kernel CNN).  The reference does it in Python in front of every Inference() call
// (test_inference.py:140-141, the same two lines at BitNetMCU.py:435-436):
//     scale = 127.0 / max(max|x|, 1e-5);  q = clip(round_half_even(x * scale), -128, 127)        all in float32.
// q = low byte of (x * scale + 1.5 * 2^23): the multiply rounds to float32 as numpy's does, the add's ulp is 1, so it rounds that
// product to the nearest integer, ties to even, as np.round does; an empty asm statement between the two keeps hipcc from contracting
// them into one fma (a single rounding would differ from numpy's two).  |x * scale| <= 127.00001, so the clip never acts.
//
// Non-finite images (out of the reference's contract).  numpy on x86 turns an image that holds a NaN or an
// infinity into ALL ZEROS (the NaN reaches the scale, or the scale is 127 / inf = 0 and inf * 0 is NaN; NaN casts to 0).  The
// kernels do the same and COUNT such images (bnm_ctx_float_nonfinite): every rounding add of a finite image lies in
// [1.5 * 2^23 - 127, 1.5 * 2^23 + 127] = bit patterns 0x4B3FFF81 .. 0x4B40007F, a NaN's bit pattern lies above as an unsigned
// integer - and an image with an infinity (scale 0) or a NaN yields a NaN product - so the unsigned maximum of an image's rounding
// adds says whether the image was finite: two v_max3_u32 per four values.
#pragma once
#include "bnm_device.hpp"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// max(|a|, |b|, |c|, |d|) as a bit pattern (a non-negative float): two VALU, no canonicalisation of the inputs
BNM_DEVICE uint32_t absmax4_bits(const f32x4 &v) {
    uint32_t m;
    asm("v_max3_f32 %0, |%1|, |%2|, |%3|\n\tv_max_f32_e64 %0, %0, |%4|" : "=&v"(m) : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));
    return m;
}

BNM_DEVICE uint32_t umax(uint32_t a, uint32_t b) { return a > b ? a : b; }

// the scale of an image whose max|x| (a non-negative float's bit pattern) is m: max(m, 1e-5f) on the bit patterns, then the IEEE division
BNM_DEVICE float quantise_scale(uint32_t m) { return __fdiv_rn(127.0f, __uint_as_float(umax(m, 0x3727C5ACu))); }

// four floats -> four int8 in one dword (byte b = value b); `worst` collects the unsigned maximum of the rounding adds' bit patterns
// (finite images stay at or below BNM_QUANT_FINITE_MAX, see the header comment)
constexpr uint32_t BNM_QUANT_FINITE_MAX = 0x4B40007Fu;
BNM_DEVICE uint32_t quantise4(const f32x4 &v, float scale, uint32_t &worst) {
    uint32_t q[4];
#pragma unroll
    for (int b = 0; b < 4; b++) {
        float p = __fmul_rn(v[b], scale);
        asm("" : "+v"(p));                     // no fma: the product is rounded to float32 first (see the header comment)
        q[b] = __float_as_uint(__fadd_rn(p, 12582912.0f));
    }
    worst = umax(umax(umax(umax(q[0], q[1]), q[2]), q[3]), worst);      // (two v_max3_u32)
    const uint32_t lo = __builtin_amdgcn_perm(q[1], q[0], 0x0c0c0400u), hi = __builtin_amdgcn_perm(q[3], q[2], 0x04000c0cu);
    return lo | hi;
}

// the four values of a lane belong to an image that occupies the WHOLE wave (one float4 per lane): its bytes, all zero when any lane
// met a non-finite value; `count` counts such images - a VECTOR register holding the same number in every lane (the fused kernels'
// scalar registers are spoken for: a scalar counter made hipcc move an outstanding work-counter take out of its register)
// `counted`: the image is one of the call's (rows past the end of a ragged tile re-read the last image: zeroed alike, not counted)
BNM_DEVICE uint32_t quantise4_image(const f32x4 &v, float scale, uint32_t &count, bool counted = true) {
    uint32_t worst = 0;
    const uint32_t q = quantise4(v, scale, worst);
    const uint32_t keep = __builtin_amdgcn_ballot_w64(worst > BNM_QUANT_FINITE_MAX) != 0ull ? 0u : ~0u;
    uint32_t c = count;
    asm("" : "+v"(c));
    count = c + (counted ? 1u & ~keep : 0u);
    return q & keep;
}

// count > 0: one atomic per wave at the end of a kernel (non-finite images are out of contract: the add practically never happens)
BNM_DEVICE void report_nonfinite(unsigned long long *counter, uint32_t count) {
    const uint32_t lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    if (counter && count && lane == 0) atomicAdd(counter, (unsigned long long)count);
}

}  // namespace
