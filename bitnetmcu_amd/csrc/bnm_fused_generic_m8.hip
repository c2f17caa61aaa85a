// Generic fused FC kernel, tile class 8 (layers of up to 256 outputs): see bnm_fused_generic_kernel.hpp.
#include "bnm_fused_generic_kernel.hpp"
BNM_GENERIC_LAUNCHER(bnmk_generic_launch_m8, 8)
