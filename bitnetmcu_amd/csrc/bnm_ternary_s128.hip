// ternary_stream_kernel<1, 128, H2, H3, ..>: the family members whose first hidden layer is 128 wide (bnm_ternary_kernel.hpp)
#include "bnm_ternary_kernel.hpp"
BNM_TERN_UNIT(128)
