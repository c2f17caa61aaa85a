// The reference's weight_quant (BitNetMCU.py:130-177) as device functions, shared by the per-layer QAT op (bnm_qat.hip) and the
// whole-model fused forward (bnm_qat_model.hip).
#pragma once
#include "bnm_device.hpp"
#include "../../include/bitnetmcu_hip.h"

namespace {

// weight_quant's scale (BitNetMCU.py:136-148)
BNM_DEVICE float qat_weight_scale(int qt, float s, float mean_abs) {
    switch (qt) {
        case BNM_QAT_NONE: return 1.0f;
        case BNM_QAT_FP130: return __fdiv_rn(128.0f, s);
        case BNM_QAT_NF4: return __fdiv_rn(1.0f, s);
        case BNM_QAT_TERNARY: return __fdiv_rn(1.0f, fmaxf(mean_abs, 1e-5f));
        case BNM_QAT_BINARY:
        case BNM_QAT_BINARYSYM: return __fdiv_rn(1.0f, s);            // 2^(1-1) / s
        case BNM_QAT_2BITSYM: return __fdiv_rn(2.0f, s);
        case BNM_QAT_4BIT:
        case BNM_QAT_4BITSYM: return __fdiv_rn(8.0f, s);
        case BNM_QAT_5BITSYM: return __fdiv_rn(16.0f, s);
        case BNM_QAT_8BIT: return __fdiv_rn(128.0f, s);
    }
    return 1.0f;
}

BNM_DEVICE float sign_of(float v) { return v > 0.0f ? 1.0f : (v < 0.0f ? -1.0f : 0.0f); }

// int8 matrix path: factor f such that f * level is an integer in [-128, 127] for every level of the type; 0 = not eligible
// (FP130 reaches +-128 -> +128 does not fit; 4bit's levels carry +0.01; NF4's are not dyadic; None has no levels)
__host__ __device__ inline int qat_i8_factor(int qt) {
    switch (qt) {
        case BNM_QAT_BINARY:
        case BNM_QAT_BINARYSYM:
        case BNM_QAT_TERNARY:
        case BNM_QAT_8BIT: return 1;
        case BNM_QAT_2BITSYM:
        case BNM_QAT_4BITSYM:
        case BNM_QAT_5BITSYM: return 2;
    }
    return 0;
}

// weight_quant's level for one weight (BitNetMCU.py:150-177).  Explicit __fmul_rn/__fsub_rn: the reference rounds
// after the multiply, so the compiler must not contract w*scale - 0.5 into one fma.
BNM_DEVICE float qat_weight_level(int qt, float w, float sc, float mean_w) {
    const float ws = __fmul_rn(w, sc);
    switch (qt) {
        case BNM_QAT_NONE: return w;
        case BNM_QAT_TERNARY: return fminf(fmaxf(rintf(ws), -1.0f), 1.0f);
        case BNM_QAT_BINARY: return sign_of(__fsub_rn(w, mean_w));
        case BNM_QAT_BINARYSYM: return sign_of(w);
        case BNM_QAT_2BITSYM: return __fadd_rn(fminf(fmaxf(rintf(__fsub_rn(ws, 0.5f)), -2.0f), 1.0f), 0.5f);
        case BNM_QAT_4BIT: return __fadd_rn(fminf(fmaxf(rintf(__fsub_rn(ws, 0.01f)), -8.0f), 7.0f), 0.01f);
        case BNM_QAT_4BITSYM: return __fadd_rn(fminf(fmaxf(rintf(__fsub_rn(ws, 0.5f)), -8.0f), 7.0f), 0.5f);
        case BNM_QAT_5BITSYM: return __fadd_rn(fminf(fmaxf(rintf(__fsub_rn(ws, 0.5f)), -16.0f), 15.0f), 0.5f);
        case BNM_QAT_8BIT: return fminf(fmaxf(rintf(ws), -128.0f), 127.0f);
        case BNM_QAT_FP130: {
            // e = floor(log2|w*scale|) clamped to [0,7]; log2(0) = -inf clamps to 0 and sign(0) = 0 gives level 0
            float e = fminf(fmaxf(floorf(log2f(fabsf(ws))), 0.0f), 7.0f);
            return sign_of(w) * exp2f(e);
        }
        case BNM_QAT_NF4: {
            const float lv[16] = {-1.0f, -0.6962f, -0.5251f, -0.3949f, -0.2844f, -0.1848f, -0.0911f, 0.0f,
                                  0.0796f, 0.1609f, 0.2461f, 0.3379f, 0.4407f, 0.5626f, 0.723f, 1.0f};
            int best = 0;
            float bd = fabsf(__fsub_rn(ws, lv[0]));
#pragma unroll
            for (int i = 1; i < 16; i++) {
                float dd = fabsf(__fsub_rn(ws, lv[i]));
                if (dd < bd) { bd = dd; best = i; }   // argmin keeps the first minimum
            }
            return lv[best];
        }
    }
    return 0.0f;
}
}  // namespace
