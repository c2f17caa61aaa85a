// Generic fused FC kernel, tile class 2 (layers of up to 64 outputs), two image tiles per wave: see bnm_fused_generic_kernel.hpp.
#include "bnm_fused_generic_kernel.hpp"
BNM_GENERIC_LAUNCHER_T2(bnmk_generic_launch_m2_t2, 2)
