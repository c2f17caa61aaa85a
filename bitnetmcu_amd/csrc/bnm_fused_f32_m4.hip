// Fused float-input FC kernel, tile class 4 (layers of up to 128 outputs): see bnm_fused_f32_kernel.hpp.  Two groups of 8 images
// (64 landing registers) in flight per wave beside the 4-tile accumulators, two waves per SIMD.
#include "bnm_fused_f32_kernel.hpp"
BNM_F32_LAUNCHER(bnmk_f32_launch_m4_g2, 4, 2, 2)
#include "bnm_persist_kernel.hpp"
BNM_PERSIST_LAUNCHER(bnmk_persist_launch_m4, 4)
