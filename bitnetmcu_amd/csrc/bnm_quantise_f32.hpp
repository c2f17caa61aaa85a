// Float32 -> int8 input quantisation, the per-value arithmetic shared by the fused float-input kernels (bnm_fused_f32_kernel.hpp: FC
// models; bnm_cnn_li_fused.hip: the one-kernel CNN).  The reference does it in Python in front of every Inference() call
// (test_inference.py:140-141, the same two lines at BitNetMCU.py:435-436):
//     scale = 127.0 / max(max|x|, 1e-5);  q = clip(round_half_even(x * scale), -128, 127)        all in float32.
// q = low byte of (x * scale + 1.5 * 2^23): the multiply rounds to float32 as numpy's does, the add's ulp is 1, so it rounds that
// product to the nearest integer, ties to even, as np.round does; an empty asm statement between the two keeps hipcc from contracting
// them into one fma (a single rounding would differ from numpy's two).  |x * scale| <= 127.00001, so the clip never acts.
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

// four floats -> four int8 in one dword (byte b = value b)
BNM_DEVICE uint32_t quantise4(const f32x4 &v, float scale) {
    uint32_t q[4];
#pragma unroll
    for (int b = 0; b < 4; b++) {
        float p = __fmul_rn(v[b], scale);
        asm("" : "+v"(p));                     // no fma: the product is rounded to float32 first (see the header comment)
        q[b] = __float_as_uint(__fadd_rn(p, 12582912.0f));
    }
    const uint32_t lo = __builtin_amdgcn_perm(q[1], q[0], 0x0c0c0400u), hi = __builtin_amdgcn_perm(q[3], q[2], 0x04000c0cu);
    return lo | hi;
}

}  // namespace
