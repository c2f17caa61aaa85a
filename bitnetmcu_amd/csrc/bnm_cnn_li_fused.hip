// CNN, ONE kernel per call: the lane = image front end (bnm_cnn_li.hip: three Toeplitz convolutions + pools per channel on the
// matrix cores, ReLUNorm records in LDS) with the model's FC tail run by the SAME wave on the same 32 images as soon as their
// records are complete - BitMnistInference's CNN schedule (BitNetMCU_MNIST_dll.c:48-91) from the image bytes to the class id
// without the 4 C act bytes ever leaving the CU.  gfx950 (CDNA4 / MI355X) only.
//
// cnn_li_kernel writes the int8 act rows to HBM and a second kernel re-reads them: 256 + 4C + 4C + 4 bytes per image and three
// launches per 2^22 images; here a call is one launch and moves 256 + 4 bytes per image.  What changes against cnn_li_kernel:
//   * the final ReLUNorm pass builds the tail's layer-1 B operand directly: lane (j, h) needs bytes 32 s + 16 h .. + 15 of image
//     j's act row for K-step s, i.e. the channels 8 s + 4 h .. + 3 - so the two lane halves normalise DIFFERENT channels (each
//     reads both halves' records of its channels from LDS: no v_permlane32_swap, half the iterations per lane) and a K-step's
//     operand is complete after four channels per lane; its MFMAs are issued at once (one operand quad live at a time);
//   * the tail's weight fragments - the generic kernel's fragment image (bnm_fused_generic_kernel.hpp: K-step major, nothing
//     padded) - are read from global memory (L2-resident: 32 KiB per tile at 256-96-64-10, beside the 384 KiB of convolution
//     fragments a tile already reads there); the records fill the LDS at 64 channels and four waves per SIMD;
//   * layers 2.. and the classifier are the generic kernel's code (hidden_layer / final_layer), logits through the wave's own
//     record area as staging once the records have been consumed.
// Serves models whose act row fits 256 bytes (C <= 64), whose FC layers are at most 96 wide (three 32-row tiles: the register
// budget of four waves per SIMD) and hold no FP1.3.0 +128 (one weight plane); everything else runs cnn_li_kernel + a tail launch.
#include <mutex>
#include "bnm_cnn_li_tile.hpp"
#include "bnm_fused_generic_kernel.hpp"
#include "bnm_quantise_f32.hpp"

namespace {
constexpr int LI_TAIL_MMAX = 3;
constexpr int LI_WAVES_F32 = 12;
}

// FLT: `images` is float32 [n][256] and the input quantisation runs in front of the operands (bnm_cnn_li_tile_body.inc) - float
// images to class ids in one kernel for the CNN models as bnm_fused_f32_kernel.hpp does for the FC ones.
// (the float form is compiled for three waves per SIMD: the quantisation in front of the channel loop does not fit the 128 registers
// the int8 form sits at without spilling operands that live across the loop)
// P2: conv3's third operand plane is in the kernel (bnm_cnn_li_tables says whether a model's weights can reach it).
template <bool DBL, bool FLT, bool P2>
__global__ __launch_bounds__(64 * (FLT ? LI_WAVES_F32 : LI_WAVES)) void cnn_li_fused_kernel(const int8_t *__restrict__ images, uint32_t n, const i32x4 *__restrict__ frags,
                                                                     const int *__restrict__ bias, uint32_t C, const char *__restrict__ tail_frags,
                                                                     BnmGenericDesc d, uint32_t *__restrict__ cls_out, int32_t *__restrict__ logits_out,
                                                                     uint32_t *__restrict__ counter, uint32_t grab) {
    constexpr int MMAX = LI_TAIL_MMAX;
    constexpr bool LI_PLANE2 = P2;
    extern __shared__ __attribute__((aligned(16))) uint8_t li_records[];      // per wave: [C][64] uint16 {f0 >> k, f1 >> k} then [C][32] uint8 k (one per image)
    const uint32_t tid = threadIdx.x;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6)), nwaves = blockDim.x >> 6;
    const uint32_t n_tiles = (n + 31u) >> 5;
    const uint32_t total_waves = gridDim.x * nwaves, wave_id = blockIdx.x * nwaves + wave;
    uint32_t tile = wave_id * grab, left = grab - 1u;      // a wave's first batch is static, later ones come from the counter

    while (tile < n_tiles) {
        int mx = 0;
        {
        LI_LANE_VALUES
        if constexpr (FLT) {
#define LI_FLOAT_IMAGES
#include "bnm_cnn_li_tile_body.inc"
#undef LI_FLOAT_IMAGES
        } else {
#include "bnm_cnn_li_tile_body.inc"
        }
        }
        // ---- ReLUNorm over the image's 4 C features (BitNetMCU_inference.c:23-72) straight into the FC tail's layer-1 operands
        LI_LANE_VALUES
        (void)rec; (void)rec_k;      // (the records are read through image-relative pointers here)
        const int s_all = 25 - __builtin_clz((uint32_t)mx | 127u);      // bitlength(mx >> 7)
        const uint16_t *const rec_img = (const uint16_t *)(li_records + wave * C * 160u) + j;      // record (c, half) of image j: [64 c + 32 half]
        const uint8_t *const k_img = li_records + wave * C * 160u + C * 128u + j;
        const uint32_t l16 = 16u * (uint32_t)lane;
        // K-step s of the act row = channels 8 s .. 8 s + 7; this lane's half of it: channels 8 s + 4 h + q, q = 0..3, one dword each
        auto operand = [&](int s) {
            i32x4 b;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const uint32_t c = 8u * (uint32_t)s + 4u * (uint32_t)h + (uint32_t)q, cc = c < C ? c : 0u;
                const uint32_t w0 = rec_img[cc * 64u], w1 = rec_img[cc * 64u + 32u];
                // (f + (1 << e >> 1)) >> e == (((2 f) >> e) + 1) >> 1 for every e >= 0 (f < 256); from e = 10 on the result is 0
                const uint32_t e = (uint32_t)min(s_all - (int)k_img[cc * 32u], 15);
                const u16x2v ee = {(unsigned short)e, (unsigned short)e}, one = {1, 1}, top = {127, 127};
                u16x2v v0 = __builtin_bit_cast(u16x2v, __builtin_amdgcn_perm(0u, w0, 0x0c010c00u) << 1);      // [2 f0, 2 f1] of half 0
                u16x2v v1 = __builtin_bit_cast(u16x2v, __builtin_amdgcn_perm(0u, w1, 0x0c010c00u) << 1);      // ... of half 1
                v0 = __builtin_elementwise_min((u16x2v)(((v0 >> ee) + one) >> one), top);
                v1 = __builtin_elementwise_min((u16x2v)(((v1 >> ee) + one) >> one), top);
                // act bytes of a channel: [window 0, 1, 2, 3] = [o0 of half 0, o0 of half 1, o1 of half 0, o1 of half 1]
                const uint32_t word = __builtin_bit_cast(uint32_t, v0) | (__builtin_bit_cast(uint32_t, v1) << 8);
                b[q] = c < C ? (int)word : 0;
            }
            return b;
        };
        i32x4 act[1][MMAX];
#pragma unroll
        for (int m = 0; m < MMAX; m++) act[0][m] = i32x4{0, 0, 0, 0};
        const uint32_t M1 = d.M[0], M2 = d.M[1], M3 = d.M[2], M4 = d.M[3], KT = d.KT0;
        static_for<1, MMAX + 1>([&](auto MI) {
            constexpr int mt = decltype(MI)::value;
            if (M1 == (uint32_t)mt) {
                i32x16 acc[mt];
                const char *a = tail_frags + (d.frag_off[0] + l16);
                static_for<0, 8>([&](auto SI) {
                    constexpr int s = decltype(SI)::value;
                    if ((uint32_t)s < KT) {
                        const i32x4 b = operand(s);
#pragma unroll
                        for (int m = 0; m < mt; m++) {
                            const i32x4 w = *(const i32x4 *)(a + (s * mt + m) * 1024);
                            acc[m] = s == 0 ? mfma0(w, b) : mfma(w, b, acc[m]);
                        }
                    }
                });
                relunorm_pack<mt, DBL, MMAX>(acc, act[0], h);
            }
        });
        // the records have been consumed: the wave's record area stages a tile's logits (2 KiB) when it is that large
        const uint32_t nc = d.n_classes;
        int32_t *const stage = (logits_out != nullptr && C * 160u >= 2048u && nc <= 16u) ? (int32_t *)(li_records + wave * C * 160u) : nullptr;
        uint32_t cls[1] = {0};
        hidden_layer<MMAX, 1, DBL, 1>(tail_frags, l16, d.frag_off[1], M2, M1, act, h);
        uint32_t m_last = M3, k_last = M2, off_last = d.frag_off[2];
        if (M4) {
            hidden_layer<MMAX, 1, DBL, 1>(tail_frags, l16, d.frag_off[2], M3, M2, act, h);
            m_last = M4; k_last = M3; off_last = d.frag_off[3];
        }
        const uint64_t first_img = (uint64_t)tile << 5;
        final_layer<MMAX, 1, 1>(tail_frags, l16, off_last, m_last, k_last, act, h, j, lane, cls, logits_out, stage, first_img, (uint64_t)n, nc, nc <= 16u);
        const uint32_t img_out = (tile << 5) + (uint32_t)j;
        if (img_out < n && h == 0) __builtin_nontemporal_store(cls[0], cls_out + img_out);
        // ---- next tile
        if (left) { tile += 1u; left -= 1u; }
        else {
            uint32_t t = 0;
            if (lane == 0) t = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
            tile = (total_waves + t) * grab;
            left = grab - 1u;
        }
    }
    work_block_leave_v(counter, total_waves);
}

// the fused kernel serves this (front end, tail) pair: act row of at most 256 bytes in the tail's fragment image, FC layers of at
// most LI_TAIL_MMAX tiles, one weight plane, and a front end the lane = image kernel itself serves
bool bnmk_cnn_li_fused_supported(uint32_t C, const BnmGenericDesc &d) {
    if (!bnmk_cnn_li_waves(C) || 4u * C > 256u || d.KT0 > 8u || d.KT0 * 32u < 4u * C || d.sp != 1u) return false;
    for (int i = 0; i < 4; i++)
        if (d.M[i] > (uint32_t)LI_TAIL_MMAX) return false;
    return d.M[0] && d.M[1] && d.M[2] && d.n_classes && d.n_classes <= 256u;
}

hipError_t bnmk_cnn_li_fused(const void *images, bool float_images, uint64_t n, const void *frags, const int *bias, uint32_t C, bool plane2, const void *tail_frags,
                             const BnmGenericDesc &d, bool dbl, uint32_t *cls, int32_t *logits, uint32_t *counter, uint32_t grab, hipStream_t s) {
    if (!n) return hipSuccess;
    uint32_t waves = bnmk_cnn_li_waves(C);
    if (float_images && waves > (uint32_t)LI_WAVES_F32) waves = (uint32_t)LI_WAVES_F32;
    if (!bnmk_cnn_li_fused_supported(C, d) || !counter || !cls || n >= (1ull << 31)) return hipErrorInvalidValue;
    if (!grab) grab = 1;
    typedef void (*fn_t)(const int8_t *, uint32_t, const i32x4 *, const int *, uint32_t, const char *, BnmGenericDesc, uint32_t *, int32_t *, uint32_t *, uint32_t);
    static const fn_t table[8] = {cnn_li_fused_kernel<false, false, false>, cnn_li_fused_kernel<true, false, false>, cnn_li_fused_kernel<false, true, false>,
                                  cnn_li_fused_kernel<true, true, false>,   cnn_li_fused_kernel<false, false, true>, cnn_li_fused_kernel<true, false, true>,
                                  cnn_li_fused_kernel<false, true, true>,   cnn_li_fused_kernel<true, true, true>};
    const int which = (plane2 ? 4 : 0) + (float_images ? 2 : 0) + (dbl ? 1 : 0);
    const fn_t fn = table[which];
    static std::mutex mu;
    static bool allowed[8][64] = {};
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
    {
        std::lock_guard<std::mutex> g(mu);
        if (dev < 64 && !allowed[which][dev]) {
            if (hipError_t e = hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); e != hipSuccess) return e;
            allowed[which][dev] = true;
        }
    }
    // (launch shape as bnmk_cnn_front_li: a call with fewer tiles than the chip has wave slots spreads them over the CUs first)
    const uint64_t tiles = (n + 31) / 32, cap = (uint64_t)bnm_num_cus();
    uint32_t waves_now = (uint32_t)((tiles + cap - 1) / cap);
    waves_now = waves_now < 1u ? 1u : waves_now > waves ? waves : waves_now;
    const uint64_t per_block = (uint64_t)waves_now * grab;
    uint64_t blocks = (tiles + per_block - 1) / per_block;
    if (blocks > cap) blocks = cap;
    fn<<<dim3((unsigned)blocks), dim3(64 * waves_now), waves_now * C * 160u, s>>>((const int8_t *)images, (uint32_t)n, (const i32x4 *)frags, bias, C, (const char *)tail_frags, d,
                                                                                 cls, logits, counter, grab);
    return hipGetLastError();
}
