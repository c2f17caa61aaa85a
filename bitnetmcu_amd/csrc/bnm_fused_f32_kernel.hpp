// Fused float-input whole-model FC kernel: float32 images -> per-image input quantisation -> the FC stack -> class ids
// (and logits) in ONE kernel.  gfx950 (CDNA4 / MI355X) only.
//
// The step in front of the path is the reference's Python (test_inference.py:140-141, same formula BitNetMCU.py:435-436):
//     scale = 127.0 / max(max|x|, 1e-5);  q = clip(round_half_even(x * scale), -128, 127)        all in float32,
// followed by BitMnistInference (BitNetMCU_MNIST_dll.c:95-121).  Two kernels with an int8 round trip through HBM move
// 1024 + 256 + 256 + 4 bytes per image; this one moves 1024 + 4.
//
// Everything behind the layer-1 B operands is the generic kernel's (bnm_fused_generic_kernel.hpp: weights in LDS, run-time
// widths, one 32-image tile per wave and iteration).  What is new is how a tile reaches LDS:
//   * a tile's 32 KiB of floats land in VGPRs, not LDS: one nontemporal global_load_dwordx4 per image (lane l holds floats
//     4l .. 4l+3: 1 KiB contiguous per instruction), in groups of 8 images, NG groups (NG x 32 registers) in flight per wave.
//     The slot a group vacates is refilled at once with a later group of this tile or of the wave's next tile, so NG x 8 KiB per
//     wave stay in flight through the tile's arithmetic; hipcc counts these loads itself (plain loads, no LDS-DMA);
//   * per group: |x| maxima per lane (v_max3_f32 with abs modifiers), then a MERGED wave reduction of the eight maxima -
//     v_permlane32_swap + max folds two registers into one (lanes < 32 keep one image, lanes >= 32 the other),
//     v_permlane16_swap + max does the same per 16-lane row, four v_max_u32_dpp row_ror finish inside the rows (the maxima are
//     non-negative floats: their bit patterns order as unsigned integers) - 20 VALU for eight images instead of 48;
//     ONE IEEE division for the eight scales, which go to scalar registers with v_readlane;
//   * q = low byte of (x * scale + 1.5 * 2^23): the multiply rounds to float32 as numpy's does, the add's ulp is 1, so it
//     rounds that product to the nearest integer, ties to even, as np.round does (an empty asm statement between the two keeps
//     hipcc from contracting them into one fma: a single rounding would differ from numpy's two).  |x * scale| <= 127.00001,
//     so the clip never acts.  Four such bytes are packed with v_perm_b32 and leave as ONE ds_write_b32 per image into the
//     wave's int8 tile buffer, in the swizzled layout the generic kernel's B-operand reads expect.
// Non-finite inputs (out of contract): an image that holds a NaN or an infinity quantises to all zeros, numpy's result on x86, and is
// counted (bnm_quantise_f32.hpp; bnm_ctx_float_nonfinite), as in bnm_quantize_input_device.
#pragma once
#include "bnm_fused_generic_kernel.hpp"
#include "bnm_quantise_f32.hpp"

namespace {

// lanes < 32: max over {l, l + 32} of a;  lanes >= 32: the same of b
BNM_DEVICE uint32_t fold32(uint32_t a, uint32_t b) {
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    return umax((uint32_t)r[0], (uint32_t)r[1]);
}
// even 16-lane rows: max over the row pair of a;  odd rows: of b
BNM_DEVICE uint32_t fold16(uint32_t a, uint32_t b) {
    auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    return umax((uint32_t)r[0], (uint32_t)r[1]);
}
// maximum over a 16-lane row, in every lane of the row (v_max_u32_dpp row_ror:8 / 4 / 2 / 1)
BNM_DEVICE uint32_t rowmax16(uint32_t v) {
    v = umax(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, false));
    v = umax(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xf, 0xf, false));
    v = umax(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x122, 0xf, 0xf, false));
    v = umax(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x121, 0xf, 0xf, false));
    return v;
}

// Quantisation scales of eight images whose floats sit one image per register quadruple, lane-linear.  Returns them as wave-uniform
// values (scalar registers).  Which lane holds which image after the folds: fold32 pairs (0,1) (2,3) (4,5) (6,7) -> lower half the
// even image; fold16 pairs the results (c0,c1) (c2,c3) -> rows 0..3 of e0 hold images 0, 2, 1, 3, of e1 images 4, 6, 5, 7; the
// select puts e1 into the upper eight lanes of every row, so image r is read from lane kLane[r].
BNM_DEVICE void group_scales(const f32x4 (&v)[8], float (&scale)[8]) {
    uint32_t m[8];
#pragma unroll
    for (int r = 0; r < 8; r++) m[r] = absmax4_bits(v[r]);
    const uint32_t c0 = fold32(m[0], m[1]), c1 = fold32(m[2], m[3]), c2 = fold32(m[4], m[5]), c3 = fold32(m[6], m[7]);
    const uint32_t e0 = rowmax16(fold16(c0, c1)), e1 = rowmax16(fold16(c2, c3));
    const uint32_t lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const uint32_t e = (lane & 8u) ? e1 : e0;
    const float s = quantise_scale(e);      // max(m, 1e-5f), then the IEEE division numpy performs
    constexpr int kLane[8] = {0, 32, 16, 48, 8, 40, 24, 56};
#pragma unroll
    for (int r = 0; r < 8; r++) scale[r] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(s), kLane[r]));
}

}  // namespace

// NG: groups of 8 images in flight per wave (1, 2 or 4: 32, 64 or 128 landing registers).  WPS: waves per SIMD the register budget is
// compiled for.  KT0 = 8 (rows of 256 values), one tile per wave and iteration.
template <int MMAX, int SP, bool DBL, int NG, int WPS>
__global__ __launch_bounds__(256 * WPS) void fused_fc_f32_kernel(const float *__restrict__ x, uint64_t n,
                                                                 const i32x4 *__restrict__ frags, BnmGenericDesc d,
                                                                 uint32_t *__restrict__ cls_out, int32_t *__restrict__ logits_out,
                                                                 uint32_t *__restrict__ counter, uint32_t batch_arg,
                                                                 unsigned long long *__restrict__ nonfinite) {
    constexpr int KT0 = 8, T = 1;
    uint32_t bad_images = 0;
    using G = RowGeom<256>;
    static_assert(NG == 1 || NG == 2 || NG == 4, "a tile's four groups must map to fixed landing slots");
    const uint32_t batch = batch_arg & 0xFFFFu;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t nwaves = blockDim.x >> 6;

    // ---- weights: fragment image global -> LDS, once per workgroup -------------------------------------
    for (uint32_t o = threadIdx.x * 16u; o < d.w_bytes; o += blockDim.x * 16u) *(i32x4 *)(smem + o) = frags[o >> 4];
    __syncthreads();

    const uint32_t tile_off = d.w_bytes + wave * (uint32_t)G::TILE;                    // this wave's int8 tile buffer
    int32_t *const stage = d.stage ? (int32_t *)(smem + d.w_bytes + nwaves * (uint32_t)G::TILE + wave * 2048u) : nullptr;
    const uint32_t lane16 = 16u * (uint32_t)lane;
    // B operand of K-step s: image j, global slot 2s+h -> LDS slot (2s+h) ^ (j & 15)
    const uint32_t rd_off = tile_off + (uint32_t)(lane & 31) * 256u + 16u * ((uint32_t)(lane >> 5) ^ G::mask((uint32_t)(lane & 31)));
    // the quantised dword of lane l (values 4l .. 4l+3 of row R) belongs at byte 4l of the row: slot l >> 2, stored at
    // slot (l >> 2) ^ (R & 15) -> byte offset (4l ^ 16 (R & 15)) + 256 R of the tile
    const uint32_t wr_off = tile_off + 4u * (uint32_t)lane;

    const uint32_t n_units = (uint32_t)((n + 31ull) >> 5);       // the launcher refuses n >= 2^36
    const float *const xl = x + 4u * (uint32_t)lane;

    f32x4 land[NG][8];
#pragma unroll
    for (int s = 0; s < NG; s++)
#pragma unroll
        for (int r = 0; r < 8; r++) land[s][r] = f32x4{0.f, 0.f, 0.f, 0.f};
    // images 8g .. 8g+7 of unit u -> landing slot g % NG; rows past the last image re-read it (never out of bounds)
    auto load_group = [&](uint32_t u, int g, f32x4(&dst)[8]) {
        const uint64_t first = (uint64_t)u * 32ull + (uint64_t)(8 * g);
        if (first + 8ull <= n) {
            const float *p = xl + first * 256ull;
#pragma unroll
            for (int r = 0; r < 8; r++) dst[r] = __builtin_nontemporal_load((const f32x4 *)(p + 256 * r));
        } else {
#pragma unroll
            for (int r = 0; r < 8; r++) {
                uint64_t img = first + (uint64_t)r;
                img = img < n ? img : n - 1ull;
                dst[r] = __builtin_nontemporal_load((const f32x4 *)(xl + img * 256ull));
            }
        }
    };

    // ---- units in batches of `batch` consecutive ones: a wave's first batch is static, later ones come from the device-wide
    // counter (as in the generic kernel).  The loop runs ONE unit ahead: `next` is known at the top of an iteration, because its
    // loads start inside it; the take that decides next's successor is issued at the top and retired behind the quantisation
    // phase, which waits on no LDS result (an outstanding scalar atomic makes the compiler's lgkmcnt waits conservative).
    const uint32_t total_waves = gridDim.x * nwaves, wave_id = blockIdx.x * nwaves + wave;
    const uint32_t words = batch_arg >> 16, wshift = (uint32_t)__builtin_ctz(words | 0x100u);
    const uint32_t my_word = wave_id & (words - 1u), first_dyn = total_waves >> wshift;
    uint32_t taken = 0;
    auto batch_first = [&](uint32_t t) { return (((first_dyn + t) << wshift) + my_word) * batch; };
    uint32_t unit = wave_id * batch, next, next_left;
    if (batch > 1u) {
        next = unit + 1u;
        next_left = batch - 2u;
    } else {
        work_take_issue(taken, counter + 16u * my_word, 1u);
        work_take_wait(taken);
        next = batch_first(taken);
        next_left = 0u;
    }
    if (unit < n_units) {
        static_for<0, NG>([&](auto GI) { load_group(unit, decltype(GI)::value, land[decltype(GI)::value]); });
    }

    const uint32_t M1 = d.M[0], M2 = d.M[1], M3 = d.M[2], M4 = d.M[3];
    const bool few_classes = d.n_classes <= 16u;
    const bool uniform = M2 == M1 && (M4 ? (M3 == M1 && M4 == 1u) : M3 == 1u) && M1 + 1u >= (uint32_t)MMAX;
    while (unit < n_units) {
        const bool take = next_left == 0u;
        if (take) work_take_issue(taken, counter + 16u * my_word, 1u);
        uint32_t lv = (uint32_t)lane, rd = rd_off, l16 = lane16, wr = wr_off;
        asm volatile("" : "+v"(lv), "+v"(rd), "+v"(l16), "+v"(wr));
        const int j = (int)(lv & 31u), h = (int)(lv >> 5);
        // ---- quantisation: four groups of 8 images, registers -> int8 rows of the tile buffer ---------------------------
        static_for<0, 4>([&](auto GI) {
            constexpr int g = decltype(GI)::value, slot = g % NG;
            float scale[8];
            group_scales(land[slot], scale);
            uint32_t q[8];
            uint32_t worst = 0;      // over the group's eight images: two v_max3_u32 per register (bnm_quantise_f32.hpp)
#pragma unroll
            for (int r = 0; r < 8; r++) q[r] = quantise4(land[slot][r], scale[r], worst);
            if (__builtin_amdgcn_ballot_w64(worst > BNM_QUANT_FINITE_MAX) != 0ull) {
                // out of contract, practically never: some image of the group holds a NaN or an infinity - find which, zero and count it
#pragma unroll
                for (int r = 0; r < 8; r++) q[r] = quantise4_image(land[slot][r], scale[r], bad_images, (uint64_t)unit * 32ull + (uint64_t)(8 * g + r) < n);
            }
            // the slot is free: the tile's group g + NG, or group g + NG - 4 of the wave's next unit
            if constexpr (g + NG < 4) load_group(unit, g + NG, land[slot]);
            else if (next < n_units) load_group(next, g + NG - 4, land[slot]);
#pragma unroll
            for (int r = 0; r < 8; r++) {
                constexpr int R0 = 8 * g;
                *(uint32_t *)(smem + ((wr ^ (16u * (uint32_t)((R0 + r) & 15))) + 256u * (uint32_t)(R0 + r))) = q[r];
            }
        });
        uint32_t nn = next + 1u, nn_left = next_left - 1u;
        if (take) {
            work_take_wait(taken);
            nn = batch_first(taken);
            nn_left = batch - 1u;
        }
        // ---- the FC stack on the tile (the generic kernel's code) ---------------------------------------------------------
        i32x4 act[T][MMAX];
        const uint64_t first_img = (uint64_t)unit * 32ull;
        uint32_t nc = d.n_classes;
        asm volatile("" : "+v"(nc));
        uint32_t cls[T] = {0};
        bool done = false;
        static_for<1, MMAX + 1>([&](auto MI) {
            constexpr int mt = decltype(MI)::value;
            if (M1 == (uint32_t)mt) {
                i32x16 acc[T][mt];
                i32x4 b0[T][KT0];
#pragma unroll
                for (int s = 0; s < KT0; s++) b0[0][s] = *(const i32x4 *)(smem + (rd ^ (32u * (uint32_t)s)));
                mma_l1<mt, KT0, 0, KT0, SP, true, T>(smem + (d.frag_off[0] + l16), b0, acc);
                relunorm_pack<mt, DBL, MMAX>(acc[0], act[0], h);
                if constexpr (mt >= MMAX - 1) {
                    if (uniform) {
                        {
#ifdef BNM_DIAG_TIMING
                        PhaseStamps st{};      // (the diagnostic build stamps the generic kernel only)
#endif
                        uniform_tail<mt, MMAX, SP, DBL, T>(smem, l16, d, act, h, j, lane, cls, logits_out, stage, first_img, n, nc, few_classes BNM_ST_ARG);
                        }
                        done = true;
                    }
                }
            }
        });
        if (!done) {
            hidden_layer<MMAX, SP, DBL, T>(smem, l16, d.frag_off[1], M2, M1, act, h);
            uint32_t m_last = M3, k_last = M2, off_last = d.frag_off[2];
            if (M4) {
                hidden_layer<MMAX, SP, DBL, T>(smem, l16, d.frag_off[2], M3, M2, act, h);
                m_last = M4; k_last = M3; off_last = d.frag_off[3];
            }
            final_layer<MMAX, SP, T>(smem, l16, off_last, m_last, k_last, act, h, j, lane, cls, logits_out, stage, first_img, n, nc, few_classes);
        }
        // every class word is written exactly once; the store is younger than every load the next iteration waits for first
        const uint64_t img = first_img + (uint64_t)(lv & 31u);
        if (img < n && lv < 32u) __builtin_nontemporal_store(cls[0], cls_out + img);
        unit = next;
        next = nn;
        next_left = nn_left;
    }
    report_nonfinite(nonfinite, bad_images);
    work_block_leave_s(counter, total_waves);   // the last wave to leave puts the counter block back to all-zero
}

// ---- launcher of one tile class (its own translation unit: bnm_fused_f32_m{2,4}.hip) ----------------------------------------
#define BNM_F32_LAUNCHER(NAME, MMAX, NG, WPS)                                                                                 \
    hipError_t NAME(uint32_t sp, bool dbl, unsigned blocks, unsigned threads, unsigned lds, hipStream_t s, const float *x,    \
                    uint64_t n, const void *frags, const BnmGenericDesc &d, uint32_t *cls, int32_t *logits, uint32_t *counter, \
                    uint32_t batch, unsigned long long *nonfinite) {                                                          \
        typedef void (*fn_t)(const float *, uint64_t, const i32x4 *, BnmGenericDesc, uint32_t *, int32_t *, uint32_t *, uint32_t, \
                             unsigned long long *);                                                                          \
        fn_t fn = nullptr;                                                                                                    \
        if (sp == 1 && dbl) fn = fused_fc_f32_kernel<MMAX, 1, true, NG, WPS>;                                                 \
        else if (sp == 1) fn = fused_fc_f32_kernel<MMAX, 1, false, NG, WPS>;                                                  \
        else if (sp == 2 && !dbl) fn = fused_fc_f32_kernel<MMAX, 2, false, NG, WPS>;                                          \
        if (!fn) return hipErrorInvalidValue;                                                                                 \
        if (!blocks) return hipSuccess;                                                                                       \
        if (hipError_t err = bnm_generic_allow_big_lds((const void *)fn); err != hipSuccess) return err;                      \
        fn<<<dim3(blocks), dim3(threads), lds, s>>>(x, n, (const i32x4 *)frags, d, cls, logits, counter, batch, nonfinite);   \
        return hipGetLastError();                                                                                             \
    }
