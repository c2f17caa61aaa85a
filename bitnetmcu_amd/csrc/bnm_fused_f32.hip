// Fused float-input whole-model FC kernel: planning and dispatch (the kernel lives in bnm_fused_f32_kernel.hpp and is instantiated
// per tile class in bnm_fused_f32_m{2,4,6}.hip).  It runs on the generic kernel's descriptor and fragment image (weights in LDS).
#include "bnm_device.hpp"
#include "bnm_kernels.h"

#define DECL(NAME)                                                                                                             \
    hipError_t NAME(uint32_t sp, bool dbl, unsigned blocks, unsigned threads, unsigned lds, hipStream_t s, const float *x, uint64_t n, \
                    const void *frags, const BnmGenericDesc &d, uint32_t *cls, int32_t *logits, uint32_t *counter, uint32_t batch, \
                    unsigned long long *nonfinite)
DECL(bnmk_f32_launch_m2_g4);
DECL(bnmk_f32_launch_m2_g2);
DECL(bnmk_f32_launch_m4_g2);
DECL(bnmk_f32_launch_m6_g1);
#undef DECL
#define DECLP(NAME)                                                                                                          \
    hipError_t NAME(uint32_t sp, bool dbl, bool launch, unsigned lds, hipStream_t s, const void *frags, const BnmGenericDesc &d, \
                    uint32_t *box, uint32_t seq0, uint64_t idle_ticks)
DECLP(bnmk_persist_launch_m2);
DECLP(bnmk_persist_launch_m4);
DECLP(bnmk_persist_launch_m6);
#undef DECLP

namespace {
constexpr uint32_t kLdsBytes = 160u * 1024u;
constexpr uint32_t kWavesPerSimd = 2;      // every instantiation is compiled for two waves per SIMD (256 VGPRs)
typedef hipError_t (*launch_fn)(uint32_t, bool, unsigned, unsigned, unsigned, hipStream_t, const float *, uint64_t, const void *,
                                const BnmGenericDesc &, uint32_t *, int32_t *, uint32_t *, uint32_t, unsigned long long *);
launch_fn launcher_of(uint32_t mmax, int groups) {
    if (mmax == 2) return groups == 2 ? bnmk_f32_launch_m2_g2 : bnmk_f32_launch_m2_g4;
    if (mmax == 4) return (groups == 0 || groups == 2) ? bnmk_f32_launch_m4_g2 : nullptr;
    if (mmax == 6) return (groups == 0 || groups == 1) ? bnmk_f32_launch_m6_g1 : nullptr;      // (the 6-tile accumulators leave room for one group)
    return nullptr;
}
// waves per workgroup (one workgroup per CU): an 8 KiB int8 tile buffer (+ 2 KiB logits staging) per wave beside the weights
uint32_t f32_waves(const BnmGenericDesc &d, bool stage) {
    const uint32_t per_wave = 8192u + (stage ? 2048u : 0u);
    if (d.w_bytes + per_wave > kLdsBytes) return 0;
    const uint32_t w = (kLdsBytes - d.w_bytes) / per_wave;
    return w < 4u * kWavesPerSimd ? w : 4u * kWavesPerSimd;
}
}  // namespace

// groups: landing groups in flight per wave (0 = the library's choice: 4 in the 2-tile class, 2 in the 4-tile class, 1 in the 6-tile class)
bool bnmk_fused_f32_supported(const BnmGenericDesc &d, bool dbl, int groups) {
    if (d.KT0 != 8 || (d.sp == 2 && dbl) || (d.sp != 1 && d.sp != 2)) return false;
    launch_fn f = launcher_of(d.mmax, groups);
    if (!f || f(d.sp, dbl, 0, 0, 0, nullptr, nullptr, 0, nullptr, d, nullptr, nullptr, nullptr, 1, nullptr) != hipSuccess) return false;
    return f32_waves(d, false) >= 4u;      // below one wave per SIMD the two-kernel path is the faster one
}

hipError_t bnmk_fused_f32(const BnmGenericDesc &d_in, bool dbl, int groups, int grid_blocks, const float *x, uint64_t n, const void *frags,
                          uint32_t *cls, int32_t *logits, uint32_t *counter, uint32_t batch, unsigned long long *nonfinite, hipStream_t s) {
    if (!bnmk_fused_f32_supported(d_in, dbl, groups)) return hipErrorInvalidValue;
    if (!n) return hipSuccess;
    if (n >= (1ull << 36) || !counter) return hipErrorInvalidValue;      // 32-bit tile indices in the kernel
    BnmGenericDesc d = d_in;
    const bool want_stage = logits != nullptr && d.n_classes <= 16u;
    const uint32_t waves = f32_waves(d, false);
    d.stage = (want_stage && f32_waves(d, true) == waves) ? 1u : 0u;
    if (!batch) batch = 4u;
    if (batch > 0xFFFFu) batch = 0xFFFFu;
    const uint32_t lds = d.w_bytes + waves * (8192u + (d.stage ? 2048u : 0u));
    const uint64_t n_units = (n + 31ull) / 32ull;
    const uint64_t want = (n_units + (uint64_t)waves * batch - 1) / ((uint64_t)waves * batch);
    const uint64_t cap = grid_blocks > 0 ? (uint64_t)grid_blocks : (uint64_t)bnm_num_cus();   // one workgroup per CU
    const uint64_t blocks = want < cap ? want : cap;
    const uint32_t words = ((blocks * waves) & 7ull) == 0ull ? 8u : 1u;
    return launcher_of(d.mmax, groups)(d.sp, dbl, (unsigned)blocks, 64u * waves, lds, s, x, n, frags, d, cls, logits, counter,
                                       batch | (words << 16), nonfinite);
}

// ---- the resident single-wave kernel behind Inference() (bnm_persist_kernel.hpp): the models the float-input kernel serves --------
namespace {
typedef hipError_t (*persist_fn)(uint32_t, bool, bool, unsigned, hipStream_t, const void *, const BnmGenericDesc &, uint32_t *, uint32_t, uint64_t);
persist_fn persist_of(uint32_t mmax) {
    return mmax == 2 ? bnmk_persist_launch_m2 : mmax == 4 ? bnmk_persist_launch_m4 : mmax == 6 ? bnmk_persist_launch_m6 : nullptr;
}
}  // namespace

bool bnmk_persistent_supported(const BnmGenericDesc &d, bool dbl) {
    if (d.KT0 != 8 || (d.sp == 2 && dbl) || (d.sp != 1 && d.sp != 2) || d.n_classes > 256u || d.w_bytes + 8192u > kLdsBytes) return false;
    persist_fn f = persist_of(d.mmax);
    return f && f(d.sp, dbl, false, 0, nullptr, nullptr, d, nullptr, 0, 0) == hipSuccess;
}

hipError_t bnmk_persistent_launch(const BnmGenericDesc &d, bool dbl, const void *frags, uint32_t *box, uint32_t seq0, uint64_t idle_ticks,
                                  hipStream_t s) {
    if (!bnmk_persistent_supported(d, dbl) || !box) return hipErrorInvalidValue;
    return persist_of(d.mmax)(d.sp, dbl, true, d.w_bytes + 8192u, s, frags, d, box, seq0, idle_ticks);
}
