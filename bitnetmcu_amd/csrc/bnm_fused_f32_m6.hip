// Fused float-input FC kernel, tile class 6 (layers of up to 192 outputs: the reference's documented 12 KB binary model, 160-160-160): see
// bnm_fused_f32_kernel.hpp.  ONE group of 8 images (32 landing registers) in flight per wave beside the 6-tile accumulators (two groups spill), two waves per SIMD.
#include "bnm_fused_f32_kernel.hpp"
BNM_F32_LAUNCHER(bnmk_f32_launch_m6_g1, 6, 1, 2)
#include "bnm_persist_kernel.hpp"
BNM_PERSIST_LAUNCHER(bnmk_persist_launch_m6, 6)
