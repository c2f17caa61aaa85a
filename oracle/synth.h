/*
 * oracle/synth.h — TEST INFRASTRUCTURE (see oracle/README.md).
 *
 * Host-C statement of the counter-based synthetic 16x16 int8 image generator defined in
 * SURVEY.md §8(d).  The reference has no generator of its own (it reads MNIST through
 * torchvision, test_inference.py:82-91, which is not available offline), so this is the
 * workload definition shared by the CPU baseline and the GPU bench.  The HIP statement is
 * synth_fill_kernel in bitnetmcu_amd/csrc/bnm_support.hip; tests compare the two
 * byte-for-byte.
 *
 *   word  w of image i (w = 0..31, 8 bytes each):  x  = splitmix64(seed + 32*i + w)
 *                                                  x2 = splitmix64(x)
 *   byte  k of that word (k = 0..7):               b  = (x  >> 8k) & 0xFF
 *                                                  b2 = (x2 >> 8k) & 0xFF
 *   Dist-U (dist 0): v = (int8) b                       uniform over [-128,127]
 *   Dist-M (dist 1): v = b < 169 ? -20 : (b2 % 148) - 20   66 % background -20, else U[-20,127]
 *                    (statistics of the 10 images in BitNetMCU_MNIST_test_data.h:1-190)
 */
#ifndef BNM_ORACLE_SYNTH_H
#define BNM_ORACLE_SYNTH_H
#include <stdint.h>

#define BNM_SEED_DIST_U 0xB17E7001ull
#define BNM_SEED_DIST_M 0xB17E7002ull

static inline uint64_t orc_splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

/* Fill `count` images starting at global image index `first` (256 bytes each). */
static inline void orc_synth_images(uint64_t seed, int dist, uint64_t first, uint64_t count,
                                    int8_t *out) {
    for (uint64_t n = 0; n < count; n++) {
        uint64_t i = first + n;
        for (uint32_t w = 0; w < 32; w++) {
            uint64_t x = orc_splitmix64(seed + 32ull * i + w);
            uint64_t x2 = orc_splitmix64(x);
            for (uint32_t k = 0; k < 8; k++) {
                uint32_t b = (uint32_t)(x >> (8 * k)) & 0xFFu;
                uint32_t b2 = (uint32_t)(x2 >> (8 * k)) & 0xFFu;
                int v;
                if (dist == 0) v = (int8_t)b;
                else v = (b < 169u) ? -20 : (int)(b2 % 148u) - 20;
                out[n * 256 + w * 8 + k] = (int8_t)v;
            }
        }
    }
}
#endif
