// Generic fused FC kernel, tile class 6 (layers of up to 192 outputs: the reference's 12 KB binary model, 160-160-160), one image
// tile per wave, input rows of 256 bytes: see bnm_fused_generic_kernel.hpp.  The class exists because it fits 256 VGPRs - two
// waves per SIMD - where the 8-tile class needs one wave's whole register file.
#include "bnm_fused_generic_kernel.hpp"
BNM_GENERIC_LAUNCHER_T1_K(bnmk_generic_launch_m6_k8, 6, 8)
