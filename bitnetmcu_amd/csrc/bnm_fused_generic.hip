// Generic fused whole-model FC kernel: planning and dispatch (the kernels live in bnm_fused_generic_kernel.hpp and are
// instantiated per tile class and tiles-per-wave in bnm_fused_generic_m{2,4}.hip, _m2_t2.hip and _m8_k{2,4,8,16}.hip).
#include <cstdlib>
#include "bnm_device.hpp"
#include "bnm_kernels.h"

#define DECL(NAME)                                                                                                       \
    hipError_t NAME(uint32_t kt0, uint32_t sp, bool dbl, unsigned blocks, unsigned threads, unsigned lds, hipStream_t s,   \
                    const int8_t *images, uint64_t n, const void *frags, const BnmGenericDesc &d, uint32_t *cls, int32_t *logits, \
                    uint32_t *counter, uint32_t batch)
DECL(bnmk_generic_launch_m2);
DECL(bnmk_generic_launch_m4);
DECL(bnmk_generic_launch_m6_k8);
DECL(bnmk_generic_launch_m8_k2);
DECL(bnmk_generic_launch_m8_k4);
DECL(bnmk_generic_launch_m8_k8);
DECL(bnmk_generic_launch_m8_k16);
DECL(bnmk_generic_launch_m2_t2);
DECL(bnmk_generic_launch_m4_t2);
#undef DECL

namespace {
constexpr uint32_t kLdsBytes = 160u * 1024u;
typedef hipError_t (*launch_fn)(uint32_t, uint32_t, bool, unsigned, unsigned, unsigned, hipStream_t, const int8_t *, uint64_t,
                                const void *, const BnmGenericDesc &, uint32_t *, int32_t *, uint32_t *, uint32_t);
launch_fn launcher_of(uint32_t mmax, int tiles, uint32_t kt0) {
    if (mmax == 8 && tiles != 2)
        return kt0 == 2 ? bnmk_generic_launch_m8_k2 : kt0 == 4 ? bnmk_generic_launch_m8_k4 : kt0 == 8 ? bnmk_generic_launch_m8_k8
               : kt0 == 16 ? bnmk_generic_launch_m8_k16 : nullptr;
    if (mmax == 6) return (tiles != 2 && kt0 == 8) ? bnmk_generic_launch_m6_k8 : nullptr;
    switch (mmax) {
        case 2: return tiles == 2 ? bnmk_generic_launch_m2_t2 : bnmk_generic_launch_m2;
        case 4: return tiles == 2 ? (kt0 == 8 ? bnmk_generic_launch_m4_t2 : nullptr) : bnmk_generic_launch_m4;
    }
    return nullptr;
}
bool instantiated(const BnmGenericDesc &d, bool dbl, int tiles) {
    launch_fn f = launcher_of(d.mmax, tiles, d.KT0);
    return f && f(d.KT0, d.sp, dbl, 0, 0, 0, nullptr, nullptr, 0, nullptr, d, nullptr, nullptr, nullptr, 1) == hipSuccess;
}
// waves per workgroup with `tiles` tile buffers (and, if asked, a 2 KiB logits staging area) per wave: as many as the register
// budget of the instantiation and the LDS left beside the weights allow.  The waves take their work from the device-wide
// counter, so any count works - it need not be a multiple of the four SIMDs.
uint32_t generic_waves(const BnmGenericDesc &d, int tiles, bool stage) {
    const uint32_t per_wave = 1024u * d.KT0 * (uint32_t)tiles + (stage ? 2048u : 0u);
    if (d.w_bytes + per_wave > kLdsBytes) return 0;
    uint32_t w = (kLdsBytes - d.w_bytes) / per_wave;
    const uint32_t cap = 4u * (uint32_t)bnmk_generic_wps((int)d.mmax, (int)d.KT0, (int)d.sp, tiles);
    return w < cap ? w : cap;
}
}  // namespace

// Fill the derived fields of a descriptor from the model's tile counts m_real[0..3] (m_real[3] == 0: three layers): the tile
// class and the layout of the fragment image - exact tile counts and K-steps, nothing padded.  Returns false when a layer is
// wider than 8 tiles.
bool bnmk_generic_plan(BnmGenericDesc &d, const uint32_t m_real[4]) {
    uint32_t mm = 0;
    for (int i = 0; i < 4; i++) mm = m_real[i] > mm ? m_real[i] : mm;
    if (mm == 0 || mm > 8) return false;
    // (the 6-tile class is instantiated for 256-byte rows - FC models - only)
    d.mmax = mm <= 2 ? 2 : mm <= 4 ? 4 : (mm <= 6 && d.KT0 == 8) ? 6 : 8;
    uint32_t kt = d.KT0, bytes = 0;
    for (int i = 0; i < 4; i++) {
        d.M[i] = m_real[i];
        d.KTP[i] = kt;
        d.frag_off[i] = bytes;
        bytes += d.M[i] * d.KTP[i] * d.sp * 1024u;
        kt = d.M[i];      // the next layer's K-steps: this layer's tiles
    }
    d.w_bytes = bytes;
    d.stage = 0;
    return true;
}

// Tiles per wave.  Two tiles per wave (2-tile class only) halve the LDS traffic of the weight fragments - one read feeds two MFMAs -
// but the two tiles' MFMAs then finish together and both ReLUNorms follow with nothing of the wave's own to overlap them, and
// half as many waves fit beside the weights: measured on the headline model, same process, interleaved (profiles/generic_ab.py,
// profiles/r03/generic_ab_r03c.log): 4.46 ms against 4.35 ms with one tile per wave (specialised dual kernel 4.11, plain read of the
// images 3.92).  The library therefore runs one tile per wave; two remain selectable (variant 8) for measurements.
int bnmk_generic_tiles(const BnmGenericDesc &d, bool dbl, int tiles, bool logits) {
    const bool stage = logits && d.n_classes <= 16u;
    auto ok = [&](int t) { return instantiated(d, dbl, t) && (generic_waves(d, t, stage) >= 1u || generic_waves(d, t, false) >= 1u); };
    if (tiles == 1 || tiles == 2) return ok(tiles) ? tiles : 0;
    if (ok(1)) return 1;
    return ok(2) ? 2 : 0;
}

// waves per CU the default launch of this model would have (0: the model does not fit): the caller warns below one per SIMD
uint32_t bnmk_generic_resident_waves(const BnmGenericDesc &d, bool dbl) {
    const int t = bnmk_generic_tiles(d, dbl, 0, false);
    return t ? generic_waves(d, t, false) : 0u;
}

bool bnmk_generic_supported(const BnmGenericDesc &d, bool dbl) {
    if (d.M[0] == 0 || d.M[1] == 0 || d.M[2] == 0 || (d.sp != 1 && d.sp != 2) || d.n_classes == 0 || d.n_classes > 256) return false;
    if (d.sp == 2 && dbl) return false;
    return bnmk_generic_tiles(d, dbl, 0, false) != 0;
}

hipError_t bnmk_fused_generic(const BnmGenericDesc &d_in, bool dbl, int tiles, int grid_blocks, const int8_t *images, uint64_t n,
                              const void *frags, uint32_t *cls, int32_t *logits, uint32_t *counter, uint32_t batch, hipStream_t s) {
    const int T = bnmk_generic_tiles(d_in, dbl, tiles, logits != nullptr);
    if (!T) return hipErrorInvalidValue;
    launch_fn launch = launcher_of(d_in.mmax, T, d_in.KT0);
    if (!n) return hipSuccess;
    if (n >= (1ull << 36) || !counter) return hipErrorInvalidValue;      // 32-bit tile indices in the kernel
    BnmGenericDesc d = d_in;
    // logits of a whole tile leave through a per-wave LDS staging area as whole-line nontemporal stores when the classes fit it
    // (<= 16) and reserving it costs no wave; otherwise piecewise from the accumulators
    const bool want_stage = logits != nullptr && d.n_classes <= 16u;
    uint32_t waves = generic_waves(d, T, false);
    d.stage = (want_stage && generic_waves(d, T, true) == waves) ? 1u : 0u;
    if (!waves) return hipErrorInvalidValue;
#ifdef BNM_DIAG      // diagnostic libraries only: fewer waves per workgroup (profiles/r05/r05_generic_phases.py)
    if (const char *e = getenv("BNM_GENERIC_WAVES")) { uint32_t w = (uint32_t)atoi(e); if (w && w < waves) waves = w; }
#endif
    // one word serves ~88 M takes per second device-wide, eight ~430 M/s (profiles/r02/s_atomic_rate_r02.log): with the counter
    // split eight ways batches of 4 tiles (3.1 M tiles per 1e8 images -> 0.8 M takes per launch) stay far below it
    if (!batch) batch = T == 2 ? 2u : 4u;
    if (batch > 0xFFFFu) batch = 0xFFFFu;
    const uint32_t lds = d.w_bytes + waves * (1024u * d.KT0 * (uint32_t)T + (d.stage ? 2048u : 0u));
    const uint64_t n_units = ((n + 31ull) / 32ull + (uint64_t)(T - 1)) / (uint64_t)T;
    uint64_t want = (n_units + (uint64_t)waves * batch - 1) / ((uint64_t)waves * batch);
    uint64_t cap = grid_blocks > 0 ? (uint64_t)grid_blocks : (uint64_t)bnm_num_cus();   // one workgroup per CU
    const uint64_t blocks = want < cap ? want : cap;
    const uint32_t words = ((blocks * waves) & 7ull) == 0ull ? 8u : 1u;
    return launch(d.KT0, d.sp, dbl, (unsigned)blocks, 64u * waves, lds, s, images, n, frags, d, cls, logits, counter,
                  batch | (words << 16));
}
