// Generic fused FC kernel, tile class 4 (layers of up to 128 outputs), one image tile per wave: see bnm_fused_generic_kernel.hpp.
#include "bnm_fused_generic_kernel.hpp"
BNM_GENERIC_LAUNCHER_T1(bnmk_generic_launch_m4, 4)
