// Generic fused FC kernel, tile class 8 (layers of up to 256 outputs), one image tile per wave, input rows of 512 bytes:
// see bnm_fused_generic_kernel.hpp.
#include "bnm_fused_generic_kernel.hpp"
BNM_GENERIC_LAUNCHER_T1_K(bnmk_generic_launch_m8_k16, 8, 16)
