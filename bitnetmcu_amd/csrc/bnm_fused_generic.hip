// Generic fused whole-model FC kernel: planning and dispatch (the kernels live in bnm_fused_generic_kernel.hpp and are
// instantiated per tile class in bnm_fused_generic_m{2,4,8}.hip).
#include "bnm_device.hpp"
#include "bnm_kernels.h"

#define DECL(NAME)                                                                                                       \
    hipError_t NAME(uint32_t kt0, uint32_t sp, bool dbl, unsigned blocks, unsigned threads, unsigned lds, hipStream_t s,   \
                    const int8_t *images, uint64_t n, const void *frags, const BnmGenericDesc &d, uint32_t *cls, int32_t *logits, \
                    uint32_t *counter, uint32_t batch)
DECL(bnmk_generic_launch_m2);
DECL(bnmk_generic_launch_m4);
DECL(bnmk_generic_launch_m8);
#undef DECL

namespace {
constexpr uint32_t kLdsBytes = 160u * 1024u;
typedef hipError_t (*launch_fn)(uint32_t, uint32_t, bool, unsigned, unsigned, unsigned, hipStream_t, const int8_t *, uint64_t,
                                const void *, const BnmGenericDesc &, uint32_t *, int32_t *, uint32_t *, uint32_t);
struct ClassInfo {
    launch_fn launch;
};
bool class_of(uint32_t mmax, ClassInfo &c) {
    switch (mmax) {
        case 2: c = {bnmk_generic_launch_m2}; return true;
        case 4: c = {bnmk_generic_launch_m4}; return true;
        case 8: c = {bnmk_generic_launch_m8}; return true;
    }
    return false;
}
// waves per workgroup: as many as the register budget of the class and the LDS left beside the weights allow
uint32_t generic_waves(const ClassInfo &, const BnmGenericDesc &d) {
    const uint32_t tile = 1024u * d.KT0;
    if (d.w_bytes + 4u * tile > kLdsBytes) return 0;
    uint32_t w = (kLdsBytes - d.w_bytes) / tile;
    const uint32_t wps = (uint32_t)bnmk_generic_wps((int)d.mmax, (int)d.KT0, (int)d.sp);
    if (w > 4u * wps) w = 4u * wps;
    return w & ~3u;
}
}  // namespace

// Fill the derived fields of a descriptor from the model's real tile counts m_real[0..3] (m_real[3] == 0: three layers):
// the tile class, the tile counts the kernel runs (8-tile class: rounded up to even), the padded K-step counts and the
// layout of the fragment image.  Returns false when a layer is wider than 8 tiles.
bool bnmk_generic_plan(BnmGenericDesc &d, const uint32_t m_real[4]) {
    uint32_t mm = 0;
    for (int i = 0; i < 4; i++) mm = m_real[i] > mm ? m_real[i] : mm;
    if (mm == 0 || mm > 8) return false;
    d.mmax = mm <= 2 ? 2 : mm <= 4 ? 4 : 8;
    uint32_t kt = d.KT0, bytes = 0;
    for (int i = 0; i < 4; i++) {
        d.M[i] = d.mmax == 8 ? (m_real[i] + 1u) & ~1u : m_real[i];
        d.KTP[i] = i == 0 ? d.KT0 : kt;
        d.frag_off[i] = bytes;
        bytes += d.M[i] * d.KTP[i] * d.sp * 1024u;
        // the next layer's K-steps: this layer's tiles, padded to half or all of the class's maximum
        kt = d.mmax == 2 ? 2 : (d.M[i] <= d.mmax / 2 ? d.mmax / 2 : d.mmax);
    }
    d.w_bytes = bytes;
    return true;
}

bool bnmk_generic_supported(const BnmGenericDesc &d, bool dbl) {
    if (d.M[0] == 0 || d.M[1] == 0 || d.M[2] == 0 || (d.sp != 1 && d.sp != 2) || d.n_classes == 0 || d.n_classes > 256) return false;
    if (d.sp == 2 && dbl) return false;
    ClassInfo c;
    if (!class_of(d.mmax, c)) return false;
    if (c.launch(d.KT0, d.sp, dbl, 0, 0, 0, nullptr, nullptr, 0, nullptr, d, nullptr, nullptr, nullptr, 1) != hipSuccess) return false;
    return generic_waves(c, d) >= 4;
}

hipError_t bnmk_fused_generic(const BnmGenericDesc &d, bool dbl, int grid_blocks, const int8_t *images, uint64_t n,
                              const void *frags, uint32_t *cls, int32_t *logits, uint32_t *counter, uint32_t batch, hipStream_t s) {
    ClassInfo c;
    if (!class_of(d.mmax, c)) return hipErrorInvalidValue;
    const uint32_t waves = generic_waves(c, d);
    if (waves < 4) return hipErrorInvalidValue;
    if (!n) return hipSuccess;
    if (n >= (1ull << 36) || !counter) return hipErrorInvalidValue;      // 32-bit tile indices in the kernel
    // one word serves ~88 M takes per second device-wide, eight ~430 M/s (profiles/r02/s_atomic_rate_r02.log): with the counter
    // split eight ways batches of 4 tiles (3.1 M tiles per 1e8 images -> 0.8 M takes per launch) stay far below it
    if (!batch) batch = 4;
    if (batch > 0xFFFFu) batch = 0xFFFFu;
    const uint32_t lds = d.w_bytes + waves * 1024u * d.KT0;
    const uint64_t n_tiles = (n + 31ull) / 32ull;
    uint64_t want = (n_tiles + (uint64_t)waves * batch - 1) / ((uint64_t)waves * batch);
    uint64_t cap = grid_blocks > 0 ? (uint64_t)grid_blocks : (uint64_t)bnm_num_cus();   // one workgroup per CU
    const uint64_t blocks = want < cap ? want : cap;
    const uint32_t words = ((blocks * waves) & 7ull) == 0ull ? 8u : 1u;
    return c.launch(d.KT0, d.sp, dbl, (unsigned)blocks, 64u * waves, lds, s, images, n, frags, d, cls, logits, counter,
                    batch | (words << 16));
}
