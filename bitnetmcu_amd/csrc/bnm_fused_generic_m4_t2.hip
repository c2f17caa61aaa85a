// Generic fused FC kernel, tile class 4 (layers of up to 128 outputs), TWO image tiles per wave (every fragment read from LDS feeds
// two MFMAs), input rows of 256 bytes, two waves per SIMD: see bnm_fused_generic_kernel.hpp.  Selected with variant 8 (A/B).
#include "bnm_fused_generic_kernel.hpp"
BNM_GENERIC_LAUNCHER_BEGIN(bnmk_generic_launch_m4_t2)
BNM_GENERIC_PICK(4, 8, 2)
BNM_GENERIC_LAUNCHER_END
