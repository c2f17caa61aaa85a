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
constexpr int LI_WAVES_PIPE = 12;
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
                                                                     uint32_t *__restrict__ counter, uint32_t grab, unsigned long long *__restrict__ nonfinite) {
#include "bnm_cnn_li_fused_body.inc"
}

// The pipelined form (bnm_cnn_li_tile_body_pipe.inc): three waves per SIMD, conv1's ReLU + packing through v_cvt_pk_i16_i32 - for
// models whose conv1 sums stay below 2^16 (`sums16` of bnm_cnn_li_tables).  Same parameters, same results.
template <bool DBL, bool FLT, bool P2>
__global__ __launch_bounds__(64 * LI_WAVES_PIPE) void cnn_li_fused_pipe_kernel(const int8_t *__restrict__ images, uint32_t n, const i32x4 *__restrict__ frags,
                                                                     const int *__restrict__ bias, uint32_t C, const char *__restrict__ tail_frags,
                                                                     BnmGenericDesc d, uint32_t *__restrict__ cls_out, int32_t *__restrict__ logits_out,
                                                                     uint32_t *__restrict__ counter, uint32_t grab, unsigned long long *__restrict__ nonfinite) {
#define LI_TILE_BODY "bnm_cnn_li_tile_body_pipe.inc"
#include "bnm_cnn_li_fused_body.inc"
}

// the fused kernel serves this (front end, tail) pair: act row of at most 256 bytes in the tail's fragment image, FC layers of at
// most LI_TAIL_MMAX tiles, one weight plane, and a front end the lane = image kernel itself serves
bool bnmk_cnn_li_fused_supported(uint32_t C, const BnmGenericDesc &d) {
    if (!bnmk_cnn_li_waves(C) || 4u * C > 256u || d.KT0 > 8u || d.KT0 * 32u < 4u * C || d.sp != 1u) return false;
    for (int i = 0; i < 4; i++)
        if (d.M[i] > (uint32_t)LI_TAIL_MMAX) return false;
    return d.M[0] && d.M[1] && d.M[2] && d.n_classes && d.n_classes <= 256u;
}

hipError_t bnmk_cnn_li_fused(const void *images, bool float_images, uint64_t n, const void *frags, const int *bias, uint32_t C, bool plane2, bool pipe,
                             const void *tail_frags, const BnmGenericDesc &d, bool dbl, uint32_t *cls, int32_t *logits, uint32_t *counter, uint32_t grab,
                             unsigned long long *nonfinite, hipStream_t s) {
    if (!n) return hipSuccess;
    uint32_t waves = bnmk_cnn_li_waves(C);
    const uint32_t cap_waves = pipe ? (uint32_t)LI_WAVES_PIPE : float_images ? (uint32_t)LI_WAVES_F32 : (uint32_t)LI_WAVES;
    if (waves > cap_waves) waves = cap_waves;
    if (!bnmk_cnn_li_fused_supported(C, d) || !counter || !cls || n >= (1ull << 31)) return hipErrorInvalidValue;
    if (!grab) grab = 1;
    typedef void (*fn_t)(const int8_t *, uint32_t, const i32x4 *, const int *, uint32_t, const char *, BnmGenericDesc, uint32_t *, int32_t *, uint32_t *, uint32_t,
                         unsigned long long *);
    static const fn_t table[16] = {cnn_li_fused_kernel<false, false, false>, cnn_li_fused_kernel<true, false, false>, cnn_li_fused_kernel<false, true, false>,
                                   cnn_li_fused_kernel<true, true, false>,   cnn_li_fused_kernel<false, false, true>, cnn_li_fused_kernel<true, false, true>,
                                   cnn_li_fused_kernel<false, true, true>,   cnn_li_fused_kernel<true, true, true>,
                                   cnn_li_fused_pipe_kernel<false, false, false>, cnn_li_fused_pipe_kernel<true, false, false>, cnn_li_fused_pipe_kernel<false, true, false>,
                                   cnn_li_fused_pipe_kernel<true, true, false>,   cnn_li_fused_pipe_kernel<false, false, true>, cnn_li_fused_pipe_kernel<true, false, true>,
                                   cnn_li_fused_pipe_kernel<false, true, true>,   cnn_li_fused_pipe_kernel<true, true, true>};
    const int which = (pipe ? 8 : 0) + (plane2 ? 4 : 0) + (float_images ? 2 : 0) + (dbl ? 1 : 0);
    const fn_t fn = table[which];
    static std::mutex mu;
    static bool allowed[16][64] = {};
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
                                                                                 cls, logits, counter, grab, nonfinite);
    return hipGetLastError();
}
