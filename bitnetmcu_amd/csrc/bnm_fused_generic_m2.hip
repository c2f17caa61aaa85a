// Generic fused FC kernel, tile class 2 (layers of up to 64 outputs): see bnm_fused_generic_kernel.hpp.
#include "bnm_fused_generic_kernel.hpp"
BNM_GENERIC_LAUNCHER(bnmk_generic_launch_m2, 2)
