// Fused float-input FC kernel, tile class 2 (layers of up to 64 outputs): see bnm_fused_f32_kernel.hpp.  A whole tile's floats in
// flight per wave (four groups, 128 landing registers) at two waves per SIMD; the two-group form for A/B measurements.
#include "bnm_fused_f32_kernel.hpp"
BNM_F32_LAUNCHER(bnmk_f32_launch_m2_g4, 2, 4, 2)
BNM_F32_LAUNCHER(bnmk_f32_launch_m2_g2, 2, 2, 2)
#include "bnm_persist_kernel.hpp"
BNM_PERSIST_LAUNCHER(bnmk_persist_launch_m2, 2)
