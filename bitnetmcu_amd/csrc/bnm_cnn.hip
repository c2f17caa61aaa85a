// CNN front end: three depthwise 3x3 stages + two pools per channel, fused ReLUNorm.
// gfx950 (CDNA4 / MI355X) only; see DESIGN.md for layouts and rooflines.  Reference semantics:
// BitNetMCU_inference.c:23-72 (ReLUNorm), :88-208 (processfclayer), :238-277 (conv), :300-322 (pool);
// schedule BitNetMCU_MNIST_dll.c:48-121.
#include "bnm_device.hpp"

// =================================================================================================
// CNN front end (BitNetMCU_MNIST_dll.c:66-80), batched.
// Mapping: one wavefront = one image, one lane = one channel.  The image is wave-uniform, so its pixels
// are scalar operands (s_load + s_bfe on the scalar unit); each lane keeps its channel's 27 int8 weights
// in VGPRs and streams the three depthwise stages row by row in registers (3 conv1 rows, 2 conv2 rows,
// the 6x6 pooled plane), never materialising a 16x16 int32 plane.  All products fit the 24-bit
// multiplier: |conv1 in| <= 128, |conv2 in| <= 9*128*128>>4 = 9216, |conv3 in| <= 9*128*9216>>4 = 663552
// < 2^23 (needs n_shift >= 4, the only value the reference uses), so every MAC is one v_mad_i32_i24.
// The ReLUNorm over all 4*C pooled values (:80) is fused: per-lane max, wave max, shift, pack 4 bytes.
// =================================================================================================
// hipcc turns a 9-tap "__mul24 + add" chain into 9 v_mul_i32_i24 + 4 v_add3 (13 issues); one fused multiply-add per
// tap is 9.  Same for the packed-dot chain, where it emits v_mov 0 + v_dot4c.  Pin the instruction choice.
BNM_DEVICE int mul24(int a, int b) {
    int r;
    asm("v_mul_i32_i24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
BNM_DEVICE int mad24(int a, int b, int c) {
    int r;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// w: per-lane packed int8x4 (VGPR), p: wave-uniform packed int8x4 (SGPR)
BNM_DEVICE int dot4_su(int w, int p) {
    int r;
    asm("v_dot4_i32_i8 %0, %1, %2, 0" : "=v"(r) : "v"(w), "s"(p));
    return r;
}
BNM_DEVICE int dot4_su(int w, int p, int acc) {
    int r;
    asm("v_dot4_i32_i8 %0, %1, %2, %3" : "=v"(r) : "v"(w), "s"(p), "v"(acc));
    return r;
}
// last dot of a chain: gfx940+ needs 3 wait states between a DOT write and a different VALU reading the result
// (LLVM GCNHazardRecognizer DotWriteDifferentVALURead); hipcc cannot see the opcode inside an asm statement and
// pads only one state, so the pad lives in the string.  Dot -> same-opcode dot through src2 needs none.
BNM_DEVICE int dot4_su_last(int w, int p, int acc) {
    int r;
    asm("v_dot4_i32_i8 %0, %1, %2, %3\n\ts_nop 2" : "=v"(r) : "v"(w), "s"(p), "v"(acc));
    return r;
}

// c0: first channel handled by this launch (lane -> channel c0 + lane).  FUSE: C <= 64, the whole
// feature vector lives in one wave and ReLUNorm is fused; otherwise the int32 features are written and
// relunorm_kernel runs afterwards.
// two int16 lanes packed in an int, for v_dot2_i32_i16 (compiler-visible builtin: hipcc pads the DOT hazards itself)
typedef short i16x2 __attribute__((ext_vector_type(2)));
BNM_DEVICE int dot2_i16(int a, int b, int acc) {
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(i16x2, a), __builtin_bit_cast(i16x2, b), acc, false);
}

template <bool FUSE>
__global__ __launch_bounds__(256) void cnn_front_kernel(const int8_t *__restrict__ images, uint64_t n,
                                                        const int8_t *__restrict__ w1, const int8_t *__restrict__ w2,
                                                        const int8_t *__restrict__ w3, uint32_t C, uint32_t c0,
                                                        uint32_t n_shift, int8_t *__restrict__ acts,
                                                        uint32_t acts_stride, int32_t *__restrict__ feat) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave0 = (uint64_t)blockIdx.x * 4u + (uint64_t)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint64_t nwaves = (uint64_t)gridDim.x * 4u;
    const uint32_t c = c0 + (uint32_t)lane;
    const bool live = c < C;

    // conv1 weights as three packed rows (w0,w1,w2,0) for v_dot4_i32_i8; conv2 weights as int16 pairs (w0,w1), (w2,0) per
    // kernel row for v_dot2_i32_i16 (stage-2 inputs are <= 9216, 14 bits); conv3 weights as 24-bit mad operands
    int wk[3], k2[9], k3[9], w01[3], w2z[3];
#pragma unroll
    for (int dy = 0; dy < 3; dy++) {
        uint32_t w = 0;
#pragma unroll
        for (int dx = 0; dx < 3; dx++) w |= (uint32_t)(uint8_t)(live ? w1[9u * c + 3 * dy + dx] : (int8_t)0) << (8 * dx);
        wk[dy] = (int)w;
    }
#pragma unroll
    for (int t = 0; t < 9; t++) {
        k2[t] = live ? (int)w2[9u * c + t] : 0;
        k3[t] = live ? (int)w3[9u * c + t] : 0;
    }
#pragma unroll
    for (int dy = 0; dy < 3; dy++) {
        w01[dy] = (int)(((uint32_t)k2[3 * dy] & 0xFFFFu) | ((uint32_t)k2[3 * dy + 1] << 16));
        w2z[dy] = (int)((uint32_t)k2[3 * dy + 2] & 0xFFFFu);
    }

    for (uint64_t img = wave0; img < n; img += nwaves) {
        const uint32_t *__restrict__ iw = (const uint32_t *)(images + img * 256ull);   // wave-uniform
        int f[4];
        {
            // Stage 1 reads the int8 image: for output column x the three pixels x..x+2 of an image row are one packed
            // scalar (s_lshr_b64 of two image dwords on the scalar unit), so a kernel row is ONE v_dot4_i32_i8 with the
            // lane's packed weights: 3 dots per output instead of 9 multiply-adds.
            // ReLU and the shift commute with max-pooling (both monotonic), so stages that feed a pool are pooled
            // first: max(a,b,c,d,0) >> n == max over the window of (max(v,0) >> n).
            int pk[3][14];      // rolling packed pixel triples of three image rows (uniform -> SGPRs)
            int r1[3][14];      // rolling conv1 rows (after ReLU and shift)
            int pr[3][14];      // the same rows as int16 pairs (r1[x], r1[x+1]) — operands of the stage-2 dots
            int r2[2][12];      // raw conv2 sums of a row pair feeding the first pool
            int p1[6][6];       // pooled 6x6 plane
            auto load_row = [&](auto Y) {
                constexpr int y = decltype(Y)::value;
                const uint32_t d0 = iw[4 * y], d1 = iw[4 * y + 1], d2 = iw[4 * y + 2], d3 = iw[4 * y + 3];
                static_for<0, 14>([&](auto X) {
                    constexpr int x = decltype(X)::value;
                    const uint32_t lo = x / 4 == 0 ? d0 : x / 4 == 1 ? d1 : x / 4 == 2 ? d2 : d3;
                    const uint32_t hi = x / 4 == 0 ? d1 : x / 4 == 1 ? d2 : x / 4 == 2 ? d3 : 0u;
                    const uint64_t pair = ((uint64_t)hi << 32) | lo;
                    pk[y % 3][x] = __builtin_amdgcn_readfirstlane((int)(uint32_t)(pair >> (8 * (x % 4))));
                });
            };
            load_row(std::integral_constant<int, 0>{});
            load_row(std::integral_constant<int, 1>{});
            static_for<0, 14>([&](auto Y1) {
                constexpr int y1 = decltype(Y1)::value;
                load_row(std::integral_constant<int, y1 + 2>{});
                static_for<0, 14>([&](auto X) {
                    constexpr int x = decltype(X)::value;
                    int s = dot4_su(wk[0], pk[y1 % 3][x]);
                    s = dot4_su(wk[1], pk[(y1 + 1) % 3][x], s);
                    s = dot4_su_last(wk[2], pk[(y1 + 2) % 3][x], s);
                    r1[y1 % 3][x] = max(s, 0) >> n_shift;
                });
                static_for<0, 14>([&](auto X) {
                    constexpr int x = decltype(X)::value;
                    pr[y1 % 3][x] = x < 13 ? (int)((uint32_t)r1[y1 % 3][x] | ((uint32_t)r1[y1 % 3][x + 1] << 16)) : r1[y1 % 3][13];
                });
                if constexpr (y1 >= 2) {
                    constexpr int y2 = y1 - 2;
                    static_for<0, 12>([&](auto X) {
                        constexpr int x = decltype(X)::value;
                        // 3 kernel rows x { (x, x+1) . (w0, w1)  +  (x+2, x+3) . (w2, 0) }: 6 dots instead of 9 multiply-adds
                        // (the chain starts from a plain 24-bit multiply so that hipcc's accumulate-in-place v_dot2c needs
                        // no v_mov 0 to seed it)
                        int s = __mul24(k2[2], r1[y2 % 3][x + 2]);
                        s = dot2_i16(pr[y2 % 3][x], w01[0], s);
                        static_for<1, 3>([&](auto DY) {
                            constexpr int dy = decltype(DY)::value;
                            s = dot2_i16(pr[(y2 + dy) % 3][x], w01[dy], s);
                            s = dot2_i16(pr[(y2 + dy) % 3][x + 2], w2z[dy], s);
                        });
                        r2[y2 & 1][x] = s;
                    });
                    if constexpr (y2 & 1) {
                        static_for<0, 6>([&](auto X) {
                            constexpr int x = decltype(X)::value;
                            int m = max(max(r2[0][2 * x], r2[0][2 * x + 1]), r2[1][2 * x]);
                            p1[y2 >> 1][x] = max(max(m, r2[1][2 * x + 1]), 0) >> n_shift;
                        });
                    }
                }
            });
            int o3[4][4];
#pragma unroll
            for (int y = 0; y < 4; y++)
#pragma unroll
                for (int x = 0; x < 4; x++) {
                    int s = mul24(k3[0], p1[y][x]);
#pragma unroll
                    for (int t = 1; t < 9; t++) s = mad24(k3[t], p1[y + t / 3][x + t % 3], s);
                    o3[y][x] = s;
                }
#pragma unroll
            for (int y = 0; y < 2; y++)
#pragma unroll
                for (int x = 0; x < 2; x++) {
                    int m = max(max(o3[2 * y][2 * x], o3[2 * y][2 * x + 1]), o3[2 * y + 1][2 * x]);
                    f[2 * y + x] = max(max(m, o3[2 * y + 1][2 * x + 1]), 0) >> n_shift;
                }
        }
        if (feat && live) {
            i32x4 v = {f[0], f[1], f[2], f[3]};
            *(i32x4 *)(feat + img * (4ull * C) + 4ull * c) = v;
        }
        if constexpr (FUSE) {
            // fused ReLUNorm over the 4*C features (values are >= 0 after ReLU; idle lanes contribute 0)
            int mx = 0;
#pragma unroll
            for (int t = 0; t < 4; t++) mx = max(mx, live ? f[t] : 0);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) mx = max(mx, __shfl_xor(mx, off));
            uint32_t tt = (uint32_t)mx >> 7;
            int sh = tt ? 32 - __builtin_clz(tt) : 0;
            int rnd = (1 << sh) >> 1;
            if (live) {
                uint32_t d = 0;
#pragma unroll
                for (int t = 0; t < 4; t++) d |= (uint32_t)min((f[t] + rnd) >> sh, 127) << (8 * t);
                *(uint32_t *)(acts + img * (uint64_t)acts_stride + 4ull * c) = d;
            }
        }
    }
}

// =================================================================================================
// CNN front end, round 2: conv1 on the matrix cores.
//
// Stage 1 is an int8 GEMM that all channels share — [positions x 9 taps] . [9 taps x channels] — so it goes to
// v_mfma_i32_32x32x32_i8 with the image patches as the A operand (rows = positions, K = taps) and the conv1 weights as the
// B operand (columns = channels).  The D layout hands lane (j, h) column j (= a channel) and 16 of a tile's 32 rows, and the
// row -> position map is chosen so that the two lane halves own two BANDS of the image:
//   half h = 0: conv1 window rows 0..7 = image-space conv1 rows 0..7   (feeds conv2 rows 0..5, pooled rows 0..2)
//   half h = 1: the image processed UPSIDE DOWN: window rows 0..7 = conv1 rows 13..6 (conv2 rows 11..6, pooled rows 5..3)
// 8 rows x 14 columns = 112 positions = exactly 7 tiles x 16 rows per half, D register r of tile t <-> window position
// 16t + r.  The overlap of the two bands (conv1 rows 6, 7) is recomputed by the otherwise idle matrix pipe.  With the lower
// band flipped and its kernels flipped with it (per-lane weight registers, see the table below) both halves execute the
// SAME instruction stream on band-local coordinates, including the one exchange the split needs: conv3 row "b" of each
// band misses one kernel row's contribution, which the partner lane computes from its own last pooled row and sends over
// (one ds_bpermute per value, 4 values).  A wave = one image; a lane = channel j of block 0 (channels 0..31) and then
// of block 1 (32..63), same registers.
// VALU per image: 588 v_dot4 + 588 ReLU/shift/pack of stage 1 become 448 (shift-and-pack by SDWA, ReLU on int16 pairs,
// odd pairs by v_alignbit); stages 2-3 keep their instruction count.  The A operand costs no VALU: three
// global_load_dword per tile with per-lane byte offsets that are the same for every image.
// Per-channel weight table (built on the host, bnm_cnn_weight_table): [band][channel][20 dwords] =
//   [0..3]  conv1 B operand: byte 4*dy+dx = w1[dy][dx], all zero for band 1 (= K-slots 16..31 of the MFMA)
//   [4..9]  conv2, per kernel row dy' (flipped for band 1): int16 pairs (w0, w1), (w2, 0)
//   [10..18] conv3 w3[dy'][dx] as ints;  [19] padding
// =================================================================================================
constexpr int CNN_WTAB_DWORDS = 20;

typedef short i16x2v __attribute__((ext_vector_type(2)));
BNM_DEVICE int pk_relu_i16(int pair) {     // max(pair, 0) on both int16 halves: one v_pk_max_i16
    i16x2v v = __builtin_bit_cast(i16x2v, pair), z = {0, 0};
    return __builtin_bit_cast(int, __builtin_elementwise_max(v, z));
}
// 16 accumulators -> 8 int16 pairs (acc[2k] >> s, acc[2k+1] >> s), arithmetic shift, low 16 bits kept: one SDWA shift per
// value writes its half-word in place.  The eight low halves first, then the eight high halves (same destination 8
// instructions later), trailing s_nop for the dst_sel forwarding hazard hipcc cannot see inside an asm statement.
BNM_DEVICE void sdwa_ashr_pack8(const i32x16 &a, int s, int (&d)[8]) {
    // The accumulators come straight from an MFMA and hipcc's hazard recognizer does not look inside an asm statement: a
    // 16-pass XDL write needs 18 wait states before a VALU read, so the block pads them itself (the other wave of the SIMD
    // issues meanwhile).
    asm("s_nop 15\n\ts_nop 3\n\t"
        "v_ashrrev_i32_sdwa %0, %8, %9 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n\t"
        "v_ashrrev_i32_sdwa %1, %8, %11 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n\t"
        "v_ashrrev_i32_sdwa %2, %8, %13 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n\t"
        "v_ashrrev_i32_sdwa %3, %8, %15 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n\t"
        "v_ashrrev_i32_sdwa %4, %8, %17 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n\t"
        "v_ashrrev_i32_sdwa %5, %8, %19 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n\t"
        "v_ashrrev_i32_sdwa %6, %8, %21 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n\t"
        "v_ashrrev_i32_sdwa %7, %8, %23 dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n\t"
        "v_ashrrev_i32_sdwa %0, %8, %10 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
        "v_ashrrev_i32_sdwa %1, %8, %12 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
        "v_ashrrev_i32_sdwa %2, %8, %14 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
        "v_ashrrev_i32_sdwa %3, %8, %16 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
        "v_ashrrev_i32_sdwa %4, %8, %18 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
        "v_ashrrev_i32_sdwa %5, %8, %20 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
        "v_ashrrev_i32_sdwa %6, %8, %22 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
        "v_ashrrev_i32_sdwa %7, %8, %24 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
        "s_nop 1"
        : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5]), "=&v"(d[6]), "=&v"(d[7])
        : "s"(s), "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(a[8]), "v"(a[9]),
          "v"(a[10]), "v"(a[11]), "v"(a[12]), "v"(a[13]), "v"(a[14]), "v"(a[15]));
}

// v_dot2_i32_i16 in its VOP3P form (src2 may be the constant 0): hipcc only emits the accumulate-in-place v_dot2c, which costs
// a v_mov 0 to start every chain.  A chain uses one opcode throughout, so no wait states are needed inside it; the last link
// pads the DOT-write -> other-VALU-read hazard itself (see dot4_su_last).
BNM_DEVICE int dot2_first(int a, int b) {
    int r;
    asm("v_dot2_i32_i16 %0, %1, %2, 0" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
BNM_DEVICE int dot2_next(int a, int b, int acc) {
    int r;
    asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(acc));
    return r;
}
BNM_DEVICE int dot2_last(int a, int b, int acc) {
    int r;
    asm("v_dot2_i32_i16 %0, %1, %2, %3\n\ts_nop 2" : "=v"(r) : "v"(a), "v"(b), "v"(acc));
    return r;
}

// maximum of a non-negative value over the wave's 64 lanes, as a wave-uniform scalar: four DPP steps leave every row of 16
// lanes holding its row maximum (quad xor 1, quad xor 2, half-row mirror, row mirror), four v_readlane + scalar max join the
// rows.  (The six dependent ds_bpermute round trips of __shfl_xor cost several hundred cycles of waiting per image.)
BNM_DEVICE int wave_max_nonneg(int v) {
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true));     // quad_perm [1,0,3,2]
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true));     // quad_perm [2,3,0,1]
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true));    // row_half_mirror
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true));    // row_mirror
    const int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
    const int c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    return max(max(a, b), max(c, d));
}

// PAIR mode: lanes 0-15 / 32-47 hold one image, lanes 16-31 / 48-63 another: the maximum of the lane's own image
BNM_DEVICE int pair_max_nonneg(int v, bool second) {
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true));
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true));
    const int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
    const int c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    return second ? max(b, d) : max(a, c);
}

// SAFE: the patch loads of image row 15, column 13 reach one byte past the image; for the LAST image of the caller's buffer
// that byte may not exist.  The launcher runs that one image through the SAFE instantiation (clamped address, bytes shifted
// into place: 2 extra VALU per load) and every other image through the plain one — no run-time test in the hot loop.
#ifdef BNM_DIAG_TIMING
// diagnostic build only (build.py --diag-timing; profiles/cnn_wait_timing.py): per wave {loop cycles, cycles waiting for the
// item's head loads, for the patch tiles, for the partner exchange, items, start stamp, XCC_ID register, HW_ID register}
__device__ uint64_t *g_cnn_rec = nullptr;
hipError_t bnmk_diag_cnn_set_record(uint64_t *d_rec) { return hipMemcpyToSymbol(HIP_SYMBOL(g_cnn_rec), &d_rec, sizeof(d_rec)); }
#define CNN_STAMP() __builtin_readcyclecounter()
#endif
// MODE 0: an item is one image x 32 channels (one or two items per image), `n` counts images.
// MODE 1 (PAIR, models with <= 16 channels, or the last <= 16 channels of a segmented model): an item is TWO consecutive images,
// `n` counts pairs.  Lane columns 0..15 are the channels
// of the pair's first image, columns 16..31 the same channels of its second image: the A operand's K-slots 0..15 (lane half
// 0) carry the first image's patch, K-slots 16..31 (lane half 1) the second image's, and the weight table holds a channel's
// conv1 weights in K-slots 0..15 for columns 0..15 and in K-slots 16..31 for columns 16..31 (zero elsewhere): the MFMA
// computes both images' conv1 at the price of one.  Everything after it is per lane and does not care which image a lane
// belongs to; the ReLUNorm maximum and the output addresses are per column group.
// MODE 2 (TRI, models with 33..48 channels, FUSE only): a unit is again a pair of images, done as THREE items - channels 0..31 of
// the first image, channels 0..31 of the second, and channels 32..47 of both in the pair layout - instead of two items per
// image with the second one half empty.
// MODE 3 (GEN, 65..256 channels, FUSE only): a unit is a pair of images done as nbf items of the first image (its whole
// 32-channel blocks), nbf of the second and - when the channel count leaves <= 16 channels beyond a multiple of 32 - one item
// in the pair layout; the pooled outputs wait in LDS (8.5 KiB per wave) instead of registers for the fused ReLUNorm over
// the 4 C features of each image, so no int32 features travel through HBM and no separate ReLUNorm kernel runs.
template <bool FUSE, bool SAFE, int MODE = 0>
__global__ __launch_bounds__(256, SAFE ? 2 : 4) void cnn_front_mfma_kernel(const int8_t *__restrict__ images, uint64_t n,
                                                             const int *__restrict__ wtab, uint32_t C_pad, uint32_t C, uint32_t c0,
                                                             uint32_t n_shift, int8_t *__restrict__ acts, uint32_t acts_stride,
                                                             int32_t *__restrict__ feat, uint32_t feat_stride, uint32_t *__restrict__ counter,
                                                             uint32_t grab) {
    // (C: one past the last channel this launch computes - the model's channel count, or the end of a channel segment when the
    // launcher splits the channels; feat_stride: ints per image in `feat`)
    const int lane = threadIdx.x & 63;
    const int j = lane & 31, h = lane >> 5;
    const uint32_t wave_id = blockIdx.x * 4u + (uint32_t)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t nwaves = gridDim.x * 4u;
    const uint32_t n32 = (uint32_t)n;           // the launcher refuses n >= 2^31
    constexpr bool PAIR = MODE == 1, TRI = MODE == 2, GEN = MODE == 3;
    static_assert(!(MODE && SAFE), "the two-image modes have no SAFE instantiation (the launcher keeps the last images out of them)");
    static_assert(!(TRI || GEN) || FUSE, "the three-item and the general mode serve whole models");
    // GEN: first channel of the pair segment (== C: none; the rule of cnn_pair_segment_start) and whole blocks per image
    uint32_t gen_pair0 = C;
    if (GEN) {
        const uint32_t R = C & 63u;
        if (R != 0u && R <= 16u) gen_pair0 = C - R;
        else if (R > 32u && R <= 48u) gen_pair0 = C - R + 32u;
    }
    const uint32_t nbf = (gen_pair0 + 31u) / 32u;
    const bool gen_pair = gen_pair0 < C;
    const uint32_t nblk = PAIR ? 1u : TRI ? 3u : GEN ? 2u * nbf + (gen_pair ? 1u : 0u) : ((C - c0) > 32u ? 2u : 1u);
    constexpr uint32_t IMG_BYTES = MODE ? 512u : 256u;      // bytes of a work unit's image(s); `n` counts units

    // A-operand addressing (the same for every image): A row i of tile t is window position 16t + q of band beta
    // (band 1 rows compute conv1 at image row 13 - wr with the kernel the right way up: only the ORDER in which the window
    // presents rows to stages 2-3 is reversed, and those stages' kernels are reversed with it in the weight table)
    const int ia = lane & 31, beta = (ia >> 2) & 1, q = (ia & 3) + 4 * (ia >> 3);
    // byte offset of the TOP patch row's four pixels (the other two rows are 16 and 32 bytes further): at most 13 * 16 + 13,
    // so the seven tiles' offsets travel as bytes of two registers and cost one v_bfe_u32 per tile instead of seven registers
    uint32_t o1p[2] = {0u, 0u};
#pragma unroll
    for (int t = 0; t < 7; t++) {
        const int p = 16 * t + q, wr = p / 14, x = p % 14;
        const int g = beta ? 13 - wr : wr;
        o1p[t >> 2] |= (uint32_t)(16 * g + x) << (8 * (t & 3));
    }
    // (extracted behind an opaque copy, or hipcc hoists the seven extractions out of the item loop - seven registers again)
    uint32_t pair_off = PAIR ? (uint32_t)h << 8 : 0u;      // lane half 1 reads the pair's second image (TRI: set per item)
    auto o1 = [&](int t) -> uint32_t {
        uint32_t w = o1p[t >> 2];
        asm volatile("" : "+v"(w));
        const uint32_t o = (w >> (8 * (t & 3))) & 0xFFu;
        if constexpr (MODE != 0) return o | pair_off;
        else return o;
    };
    const int vshift = (int)n_shift;
    const int partner = (lane ^ 32) << 2;        // ds_bpermute address of the lane that owns the other band of this channel

    // ---- work items: (image, 32-channel block), block 0 then block 1 of each image.  Images are handed out in batches of
    // `grab` consecutive images from ONE device-wide counter (the first batch of every wave is static).  Why not a fixed
    // share per wave: the SIMD's arbiter favours its oldest wave, so with equal shares the four waves of a SIMD finish one
    // after the other (measured, profiles/cnn_wait_timing.py: the fastest wave of a launch took 3.9 M cycles, the slowest
    // 8.3 M for the same 512 items) and the last quarter of the launch runs at one wave per SIMD with every latency exposed.
    // With the counter every wave works until the images run out.  counter == nullptr: fixed shares (single-image launches).
    // A operand of a tile: bytes 0-3, 4-7, 8-11 of the lane's 16 K-bytes = image rows g, g+1, g+2 at columns x..x+3
    // (K-slots 12..15 meet zero weights); three L1-resident dword loads, reloaded per channel block.
    auto load_A = [&](const int8_t *ip, uint32_t o) -> i32x4 {
        int va, vb, vc;
        if constexpr (!SAFE) {
            // buffer loads: wave-uniform descriptor of this image (rebuilt per item on the scalar unit) + the lane's 32-bit
            // offset.  (As plain pointer arithmetic hipcc formed a 64-bit lane address per tile - one more VALU per tile and
            // seven register PAIRS carried across the item loop.)  512 bytes are in range: the one-byte overrun of the last
            // row belongs to the next image, which exists for every image this instantiation sees.
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)ip, 0, 256 + IMG_BYTES, 0x00020000);
            va = __builtin_amdgcn_raw_buffer_load_b32(rs, (int)o, 0, 0);
            vb = __builtin_amdgcn_raw_buffer_load_b32(rs, (int)(o + 16u), 0, 0);
            vc = __builtin_amdgcn_raw_buffer_load_b32(rs, (int)(o + 32u), 0, 0);
        } else {      // offsets 253..255 (last image row, x = 13): read the dword at 252 and shift the bytes down
            auto ld = [&](uint32_t oo) {
                const uint32_t oc = oo > 252u ? 252u : oo;
                return (int)((uint32_t)(*(const int *)(ip + oc)) >> (8u * (oo - oc)));
            };
            va = ld(o); vb = ld(o + 16u); vc = ld(o + 32u);
        }
        return i32x4{va, vb, vc, vc};
    };
    // what an item needs before its first MFMA: its block's 20 weight dwords and its first A tile.  (Fetching them one item
    // ahead was tried: hipcc waits for loop-carried loads on the back edge, and the extra live registers cost the third wave.)
    struct Head {
        i32x4 q0, q1, q2, q3, q4, a0;
    };
    // the weight table through a wave-uniform descriptor + one 32-bit lane offset (the channel block is a scalar offset)
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void *)wtab, 0, (int)(2u * C_pad * CNN_WTAB_DWORDS * 4u), 0x00020000);
    const int wvoff = (int)(((uint32_t)h * C_pad + c0 + (uint32_t)j) * (CNN_WTAB_DWORDS * 4u));
    auto fetch = [&](const int8_t *ip0, uint32_t bb) -> Head {
        Head hd;
        hd.q0 = __builtin_amdgcn_raw_buffer_load_b128(wrs, wvoff, (int)(bb * 32u * CNN_WTAB_DWORDS * 4u), 0);
        hd.q1 = __builtin_amdgcn_raw_buffer_load_b128(wrs, wvoff + 16, (int)(bb * 32u * CNN_WTAB_DWORDS * 4u), 0);
        hd.q2 = __builtin_amdgcn_raw_buffer_load_b128(wrs, wvoff + 32, (int)(bb * 32u * CNN_WTAB_DWORDS * 4u), 0);
        hd.q3 = __builtin_amdgcn_raw_buffer_load_b128(wrs, wvoff + 48, (int)(bb * 32u * CNN_WTAB_DWORDS * 4u), 0);
        hd.q4 = __builtin_amdgcn_raw_buffer_load_b128(wrs, wvoff + 64, (int)(bb * 32u * CNN_WTAB_DWORDS * 4u), 0);
        hd.a0 = load_A(ip0, o1(0));
        return hd;
    };
    // landing zone of the next-image touch: an LDS-DMA load has no register destination to keep reserved
    __shared__ int s_touch[4][64];
    const uint32_t touch_lds = (uint32_t)(uintptr_t)&s_touch[__builtin_amdgcn_readfirstlane(threadIdx.x >> 6)][0];
    int fo[TRI ? 3 : 2][2] = {};              // the lane's pooled outputs: [block][t & 1], t = 2h + (t & 1)
    constexpr int GEN_BLOCKS = 17;            // GEN keeps them in LDS: up to 8 + 8 + 1 items per image pair (34 KiB per workgroup)
    __shared__ int s_fo[GEN ? 4 : 1][GEN ? GEN_BLOCKS : 1][2][64];
    int(*const fo_lds)[2][64] = s_fo[GEN ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : 0];
    int pend_sa = 0, pend_sb = 0;             // GEN: ReLUNorm shifts of the pending pair's two images
    // The act bytes of an image are stored at the START of the next item, not at the end of their own: vmcnt counts stores,
    // and the wait hipcc places on the loop's back edge would otherwise park the wave for the write acknowledgement of a
    // store it has just issued (measured: as long as the whole arithmetic of an image).  Deferred, whatever that wait
    // covers is thousands of cycles old.
    const int8_t *pend_row = acts;            // wave-uniform: the pending image's act row
    uint32_t pend_v = 0;                      // block 0's two bytes | block 1's two bytes << 16
    uint32_t pend_w = 0;                      // TRI: block 2's two bytes
    bool pend = false;
    // stores go through a wave-uniform descriptor of the image's act row (4 C valid bytes) + the lane's constant offset: the
    // range check drops the lanes of channels >= C (and the whole second block of a <= 32-channel model), no lane addresses
    // (lane-constant offsets that are needed once per image are re-derived from the lane id behind an opaque copy: hoisted
    // out of the item loop they would each hold a register for the whole kernel)
    auto lane_now = []() -> int {
        int l;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
        return l;
    };
    // (PAIR: `row` = bytes from the first image's output row to the second's; lanes of channels >= C get an offset out of range)
    auto store_off = [&](uint32_t row) -> int {
        const int l = lane_now();
        if constexpr (PAIR) {
            const int ch = l & 15;
            return c0 + (uint32_t)ch < C ? ((l >> 4) & 1) * (int)row + 4 * (int)c0 + 4 * ch + 2 * (l >> 5) : 0x1FFFFF00;
        } else {
            return 4 * (int)c0 + 4 * (l & 31) + 2 * (l >> 5);
        }
    };
    auto flush_pending = [&]() {
        if (!pend) return;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)pend_row, 0, (int)((MODE ? acts_stride : 0u) + 4u * C), 0x00020000);
        if constexpr (GEN) {
            // the pooled outputs still sit in LDS (the next unit writes its own only at the end of its first item)
            const int l = lane_now();
            const bool second = (l & 16) != 0;
            auto norm2 = [](int a, int b2, int sh) -> short {
                const int rnd = (1 << sh) >> 1;
                return (short)((uint32_t)min((a + rnd) >> sh, 127) | ((uint32_t)min((b2 + rnd) >> sh, 127) << 8));
            };
            for (uint32_t bq = 0; bq < nbf; bq++) {
                const uint32_t ch = 32u * bq + (uint32_t)(l & 31);
                const int o = ch < C ? (int)(4u * ch) + 2 * (l >> 5) : 0x1FFFFF00;
                __builtin_amdgcn_raw_buffer_store_b16(norm2(fo_lds[bq][0][l], fo_lds[bq][1][l], pend_sa), rs, o, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b16(norm2(fo_lds[nbf + bq][0][l], fo_lds[nbf + bq][1][l], pend_sb), rs, o, (int)acts_stride, 0);
            }
            if (gen_pair) {
                const uint32_t ch = gen_pair0 + (uint32_t)(l & 15);
                const int o = ch < C ? (second ? (int)acts_stride : 0) + (int)(4u * ch) + 2 * (l >> 5) : 0x1FFFFF00;
                __builtin_amdgcn_raw_buffer_store_b16(norm2(fo_lds[2u * nbf][0][l], fo_lds[2u * nbf][1][l], second ? pend_sb : pend_sa), rs, o, 0, 0);
            }
            pend = false;
            return;
        }
        if constexpr (TRI) {
            // blocks 0 / 1: channel l & 31 of the first / second image; block 2: channel 32 + (l & 15) of the image of the column group
            const int l = lane_now();
            const int o01 = 4 * (l & 31) + 2 * (l >> 5);
            const int o2 = 32u + (uint32_t)(l & 15) < C ? ((l >> 4) & 1) * (int)acts_stride + 128 + 4 * (l & 15) + 2 * (l >> 5) : 0x1FFFFF00;
            __builtin_amdgcn_raw_buffer_store_b16((short)(pend_v & 0xFFFFu), rs, o01, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b16((short)(pend_v >> 16), rs, o01, (int)acts_stride, 0);
            __builtin_amdgcn_raw_buffer_store_b16((short)pend_w, rs, o2, 0, 0);
            pend = false;
            return;
        }
        const int st_off = store_off(acts_stride);
        __builtin_amdgcn_raw_buffer_store_b16((short)(pend_v & 0xFFFFu), rs, st_off, 0, 0);
        if constexpr (!PAIR) __builtin_amdgcn_raw_buffer_store_b16((short)(pend_v >> 16), rs, st_off, 128, 0);     // block 1: scalar offset
        pend = false;
    };
#ifdef BNM_DIAG_TIMING
    uint64_t t_head = 0, t_tile = 0, t_xchg = 0;
    const uint64_t t_start = CNN_STAMP();
#endif
    uint32_t cur_batch = wave_id * grab, unit = 0, blk = 0;      // unit: image (pair) within the batch; blk: its item
#ifdef BNM_DIAG_TIMING
    uint32_t n_items = 0;
#endif
    int nxt_v = 0;                            // lane 0: the counter before this wave's add = first image of its next batch - nwaves * grab
    for (;;) {
        const uint32_t img = cur_batch + unit;
        if (img >= n32) break;                // batches are handed out in increasing order: nothing is left for this wave
        // (TRI / GEN: which image of the pair this item reads, and whether it is the item in the pair layout)
        const bool pair_item = TRI ? blk == 2u : GEN ? blk == 2u * nbf : false;
        const bool second_img = TRI ? blk == 1u : GEN ? (blk >= nbf && !pair_item) : false;
        const int8_t *__restrict__ ip = images + (uint64_t)img * IMG_BYTES + (second_img ? 256u : 0u);       // wave-uniform
        if constexpr (TRI || GEN) pair_off = pair_item ? ((uint32_t)lane_now() & 32u) << 3 : 0u;
        // the batch after this one is requested at the start of this one; the answer is needed `grab` images later
        if (unit == 0 && blk == 0 && counter != nullptr && lane_now() == 0)
            nxt_v = (int)__hip_atomic_fetch_add(counter, grab, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // Touch the wave's NEXT image (64 lanes x 4 B = its 256 bytes) so that its patch loads hit the cache instead of waiting
        // a microsecond for HBM.  The data is never used: an LDS-DMA load drops it into a per-wave landing zone.
        if (blk == 0) {
            uint32_t ni = img + 1u;
            if (unit + 1u == grab)                   // last image of the batch: the next one opens the next batch
                ni = counter != nullptr ? (grab > 1u ? nwaves * grab + (uint32_t)__builtin_amdgcn_readfirstlane(nxt_v) : n32)
                                        : cur_batch + nwaves * grab;
            if (ni < n32) {
                const int8_t *np = images + (uint64_t)ni * IMG_BYTES;
                uint32_t keep;
                const int lane4 = lane_now() * 4;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, %3\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(lane4), "s"(touch_lds), "s"(np) : "memory");
                if constexpr (MODE != 0)
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, %3 offset:256\n\ts_mov_b32 m0, %0"
                                 : "=&s"(keep) : "v"(lane4), "s"(touch_lds), "s"(np) : "memory");
            }
        }
        const Head cur = fetch(ip, TRI ? (blk == 2u ? 1u : 0u) : GEN ? (pair_item ? nbf : second_img ? blk - nbf : blk) : blk);
#ifdef BNM_DIAG_TIMING
        {
            const uint64_t t0 = CNN_STAMP();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            t_head += CNN_STAMP() - t0;
        }
#endif
        const i32x4 wB = cur.q0, wq1 = cur.q1, wq2 = cur.q2, wq3 = cur.q3, wq4 = cur.q4;
        const int w01[3] = {wq1[0], wq1[2], wq2[0]}, w2z[3] = {wq1[1], wq1[3], wq2[1]};
        const int k3[9] = {wq2[2], wq2[3], wq3[0], wq3[1], wq3[2], wq3[3], wq4[0], wq4[1], wq4[2]};
        // ---- stage 1: 7 MFMAs, each followed by the shift / pack / ReLU of its 16 window positions -------------------
        // conv1 window as int16 pairs E[k] = (r[2k], r[2k+1]), raster order, 7 pairs per row.  conv2 at an even column x reads
        // (r[x], r[x+1]) . (w0, w1) + (r[x+2], r[x+3]) . (w2, 0); at an odd column the SAME two registers hold
        // (r[x-1], r[x]) and (r[x+1], r[x+2]), so it is (.) . (0, w0) + (.) . (w1, w2): two more weight pairs per kernel row
        // (6 VALU per item) instead of a second, odd-aligned copy of the window (56 v_alignbit and 56 registers per item).
        int E[56];
        int wz0[3], w12[3];
#pragma unroll
        for (int dy = 0; dy < 3; dy++) {
            wz0[dy] = (int)((uint32_t)w01[dy] << 16);
            w12[dy] = (int)__builtin_amdgcn_alignbit((uint32_t)w2z[dy], (uint32_t)w01[dy], 16);
        }
        int hm[6];                 // even conv2 rows, already reduced to max(sum[2x], sum[2x+1], 0): 6 live values, not 12 raw sums
        int p1[3][6];              // pooled rows (band-local)
        i32x4 Anext = cur.a0;
        static_for<0, 7>([&](auto T) {
            constexpr int t = decltype(T)::value;
            i32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; r++) acc[r] = 0;
#ifdef BNM_DIAG_TIMING
            if constexpr (t > 0) {
                const uint64_t t0 = CNN_STAMP();
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                t_tile += CNN_STAMP() - t0;
            }
#endif
            acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(Anext, wB, acc, 0, 0, 0);
            // (a scheduling fence per tile: left alone hipcc hoists all seven tiles' patch loads to the top of the item, 18 live
            // registers that push the kernel over the 168 of three waves per SIMD)
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (t < 6) Anext = load_A(ip, o1(t + 1));
            // (the previous image's act bytes go out here, BEHIND all of this item's patch loads in program order: no wait
            // for a patch load covers them, and the back-edge wait finds them ~500 VALU instructions old)
            if constexpr (t == 6) { if (blk == 0) flush_pending(); }
            int d[8];
            sdwa_ashr_pack8(acc, vshift, d);
#pragma unroll
            for (int kk = 0; kk < 8; kk++) E[8 * t + kk] = pk_relu_i16(d[kk]);
            // ---- stage 2: the conv2 window rows whose three conv1 rows now exist ---------------------------------------
            static_for<0, 6>([&](auto Y) {
                constexpr int y = decltype(Y)::value;
                constexpr int ready = (7 * (y + 3) - 1) / 8;                          // tile that produced E[7 (y + 2) + 6]
                if constexpr (ready == t) {
                    // pool-before-ReLU, two columns at a time: an even row leaves max(s0, s1, 0), the odd row below it
                    // max(that, s0, s1) >> shift (3 VALU per pooled value, as before, with 6 values carried instead of 12)
                    static_for<0, 6>([&](auto XP) {
                        constexpr int xp = decltype(XP)::value;
                        int sm[2];
                        static_for<0, 2>([&](auto XL) {
                            constexpr int x = 2 * xp + decltype(XL)::value;
                            int sum = 0;
                            static_for<0, 3>([&](auto DY) {
                                constexpr int dy = decltype(DY)::value;
                                constexpr int base = 7 * (y + dy);              // first pair of conv1 window row y + dy
                                const int pa = E[base + x / 2], pb = E[base + x / 2 + 1];
                                const int wa = (x & 1) ? wz0[dy] : w01[dy], wb = (x & 1) ? w12[dy] : w2z[dy];
                                sum = dy == 0 ? dot2_first(pa, wa) : dot2_next(pa, wa, sum);
                                sum = dy == 2 ? dot2_last(pb, wb, sum) : dot2_next(pb, wb, sum);
                            });
                            sm[x & 1] = sum;
                        });
                        if constexpr (y & 1) p1[y >> 1][xp] = max(max(hm[xp], sm[0]), sm[1]) >> n_shift;
                        else hm[xp] = max(max(sm[0], sm[1]), 0);
                    });
                }
            });
        });
        // ---- stage 3 on band-local rows: conv3 row a complete, row b misses kernel row 2 over the PARTNER's pooled
        // row 2, which the partner computes (its `send`) while this lane computes the partner's missing term -------------
        int oa3[4], ob3[4], snd[4];
#pragma unroll
        for (int x = 0; x < 4; x++) {
            int sa = mul24(k3[0], p1[0][x]);
#pragma unroll
            for (int tt = 1; tt < 9; tt++) sa = mad24(k3[tt], p1[tt / 3][x + tt % 3], sa);
            int sb = mul24(k3[0], p1[1][x]);
#pragma unroll
            for (int tt = 1; tt < 6; tt++) sb = mad24(k3[tt], p1[1 + tt / 3][x + tt % 3], sb);
            int sc = mul24(k3[0], p1[2][x]);
            sc = mad24(k3[1], p1[2][x + 1], sc);
            sc = mad24(k3[2], p1[2][x + 2], sc);
            oa3[x] = sa; ob3[x] = sb; snd[x] = sc;
        }
#ifdef BNM_DIAG_TIMING
        const uint64_t tx0 = CNN_STAMP();
#endif
#pragma unroll
        for (int x = 0; x < 4; x++) ob3[x] += __builtin_amdgcn_ds_bpermute(partner, snd[x]);
#ifdef BNM_DIAG_TIMING
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        t_xchg += CNN_STAMP() - tx0;
#endif
#pragma unroll
        for (int x = 0; x < 2; x++) {
            int m = max(max(oa3[2 * x], oa3[2 * x + 1]), ob3[2 * x]);
            const int v = max(max(m, ob3[2 * x + 1]), 0) >> n_shift;
            if constexpr (GEN) {
                fo_lds[blk][x][lane_now()] = v;
            } else {
                fo[0][x] = blk == 0 ? v : fo[0][x];
                fo[1][x] = blk == 1 ? v : fo[1][x];
                if constexpr (TRI) fo[2][x] = blk == 2 ? v : fo[2][x];
            }
        }
        if (blk + 1 == nblk) {
            // ---- outputs: lane (j, h), block b: channel c0 + 32 b + j, values t = 2h, 2h + 1 ---------------------------------
            if (feat) {
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)(feat + (uint64_t)img * ((MODE ? 2ull : 1ull) * feat_stride)), 0,
                                                                                   (int)((MODE ? 4u * feat_stride : 0u) + 16u * C), 0x00020000);
                typedef int i32x2 __attribute__((ext_vector_type(2)));
                if constexpr (GEN) {
                    const int l = lane_now();
                    for (uint32_t bq = 0; bq < nbf; bq++) {
                        const uint32_t ch = 32u * bq + (uint32_t)(l & 31);
                        const int o = ch < C ? (int)(16u * ch) + 8 * (l >> 5) : 0x7FFFFC00;
                        __builtin_amdgcn_raw_buffer_store_b64(i32x2{fo_lds[bq][0][l], fo_lds[bq][1][l]}, rs, o, 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b64(i32x2{fo_lds[nbf + bq][0][l], fo_lds[nbf + bq][1][l]}, rs, o, (int)(4u * feat_stride), 0);
                    }
                    if (gen_pair) {
                        const uint32_t ch = gen_pair0 + (uint32_t)(l & 15);
                        const int o = ch < C ? ((l >> 4) & 1) * 4 * (int)feat_stride + (int)(16u * ch) + 8 * (l >> 5) : 0x7FFFFC00;
                        __builtin_amdgcn_raw_buffer_store_b64(i32x2{fo_lds[2u * nbf][0][l], fo_lds[2u * nbf][1][l]}, rs, o, 0, 0);
                    }
                } else if constexpr (TRI) {
                    const int l = lane_now();
                    const int o01 = 16 * (l & 31) + 8 * (l >> 5);
                    const int o2 = 32u + (uint32_t)(l & 15) < C ? ((l >> 4) & 1) * 4 * (int)feat_stride + 512 + 16 * (l & 15) + 8 * (l >> 5) : 0x7FFFFC00;
                    __builtin_amdgcn_raw_buffer_store_b64(i32x2{fo[0][0], fo[0][1]}, rs, o01, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b64(i32x2{fo[1][0], fo[1][1]}, rs, o01, (int)(4u * feat_stride), 0);
                    __builtin_amdgcn_raw_buffer_store_b64(i32x2{fo[2][0], fo[2][1]}, rs, o2, 0, 0);
                } else {
                    const int f_off = 4 * store_off(feat_stride);
                    __builtin_amdgcn_raw_buffer_store_b64(i32x2{fo[0][0], fo[0][1]}, rs, f_off, 0, 0);
                    if constexpr (!PAIR) __builtin_amdgcn_raw_buffer_store_b64(i32x2{fo[1][0], fo[1][1]}, rs, f_off, 512, 0);
                }
            }
            if constexpr (FUSE) {
                // fused ReLUNorm over the 4*C features (all >= 0; channels >= C have zero weights and contribute 0)
                auto norm2 = [](int a, int b2, int sh, int rnd) -> uint32_t {
                    return (uint32_t)min((a + rnd) >> sh, 127) | ((uint32_t)min((b2 + rnd) >> sh, 127) << 8);
                };
                auto shift_of = [](int mx) -> int {
                    const uint32_t tt = (uint32_t)mx >> 7;
                    return tt ? 32 - __builtin_clz(tt) : 0;
                };
                if constexpr (GEN) {
                    const int l = lane_now();
                    const bool second = (l & 16) != 0;
                    int va = 0, vb = 0;
                    for (uint32_t bq = 0; bq < nbf; bq++) {
                        va = max(va, max(fo_lds[bq][0][l], fo_lds[bq][1][l]));
                        vb = max(vb, max(fo_lds[nbf + bq][0][l], fo_lds[nbf + bq][1][l]));
                    }
                    if (gen_pair) {
                        const int m2 = max(fo_lds[2u * nbf][0][l], fo_lds[2u * nbf][1][l]);
                        va = max(va, second ? 0 : m2);
                        vb = max(vb, second ? m2 : 0);
                    }
                    pend_sa = shift_of(wave_max_nonneg(va));
                    pend_sb = shift_of(wave_max_nonneg(vb));
                } else if constexpr (TRI) {
                    // first image: block 0 + column group 0 of block 2; second image: block 1 + column group 1 of block 2
                    const bool second = (lane_now() & 16) != 0;
                    const int m2 = max(fo[2][0], fo[2][1]);
                    const int ma = wave_max_nonneg(max(max(fo[0][0], fo[0][1]), second ? 0 : m2));
                    const int mb = wave_max_nonneg(max(max(fo[1][0], fo[1][1]), second ? m2 : 0));
                    const int sa = shift_of(ma), sb = shift_of(mb);
                    const int s2 = second ? sb : sa;
                    pend_v = norm2(fo[0][0], fo[0][1], sa, (1 << sa) >> 1) | (norm2(fo[1][0], fo[1][1], sb, (1 << sb) >> 1) << 16);
                    pend_w = norm2(fo[2][0], fo[2][1], s2, (1 << s2) >> 1);
                } else {
                    const int mxl = max(max(fo[0][0], fo[0][1]), nblk == 2u ? max(fo[1][0], fo[1][1]) : 0);
                    int mx;
                    if constexpr (PAIR) mx = pair_max_nonneg(mxl, (lane_now() & 16) != 0);
                    else mx = wave_max_nonneg(mxl);
                    const int sh = shift_of(mx);
                    const int rnd = (1 << sh) >> 1;
                    pend_v = 0;
#pragma unroll
                    for (uint32_t bb = 0; bb < 2; bb++) pend_v |= norm2(fo[bb][0], fo[bb][1], sh, rnd) << (16 * bb);
                }
                pend_row = acts + (uint64_t)img * ((MODE ? 2ull : 1ull) * (uint64_t)acts_stride);
                pend = true;
            }
        }
        // ---- next item; after the batch's last one: the batch the counter handed out ------------------------------------------
#ifdef BNM_DIAG_TIMING
        n_items++;
#endif
        if (++blk == nblk) {
            blk = 0;
            if (++unit == grab) {
                unit = 0;
                cur_batch = counter != nullptr ? nwaves * grab + (uint32_t)__builtin_amdgcn_readfirstlane(nxt_v) : cur_batch + nwaves * grab;
            }
        }
    }
    flush_pending();
    if (counter != nullptr) work_block_leave_v(counter, nwaves);   // the last wave to leave zeroes the counter block
#ifdef BNM_DIAG_TIMING
    if (g_cnn_rec && lane_now() == 0) {
        uint64_t *rec = g_cnn_rec + 8ull * wave_id;
        rec[0] = CNN_STAMP() - t_start; rec[1] = t_head; rec[2] = t_tile; rec[3] = t_xchg; rec[4] = n_items;
        rec[5] = t_start; rec[6] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)); rec[7] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
    }
#endif
}

// ---- launchers -------------------------------------------------------------------------------------------------
// Channel segments of a model with C channels: whole groups of 64 (two 32-channel items per image), then the remainder R:
// R <= 16 one PAIR segment (two images per item), R <= 32 one 32-channel item, R <= 48 a 32-channel item + a PAIR segment,
// else two 32-channel items.  `pair0` = first channel of the PAIR segment (== C: none).
static uint32_t cnn_pair_segment_start(uint32_t C) {
    const uint32_t R = C % 64u;
    if (R == 0u) return C;
    if (R <= 16u) return C - R;
    if (R > 32u && R <= 48u) return C - R + 32u;
    return C;
}

// Host-side weight table for cnn_front_mfma_kernel (layout above).  w1/w2/w3: int8 [C][9] as in the header; out:
// 2 * C_pad * 20 ints with C_pad = C rounded up to 64 (surplus channels: zero weights).
// The PAIR segment's 32 table columns (the pair mode's layout, which the one-image instantiations read just as well): columns
// 16..31 repeat the segment's channels, and a column's conv1 weights sit in the K-half of its column group - K-slots 0..15
// (band 0's entry) for columns 0..15, K-slots 16..31 (band 1's entry) for columns 16..31.  With one image in both K-halves of
// the A operand the columns 16..31 merely recompute the segment's channels (their stores fall outside the valid output bytes).
void bnm_cnn_weight_table(const int8_t *w1, const int8_t *w2, const int8_t *w3, uint32_t C, int *out) {
    const uint32_t C_pad = (C + 63u) / 64u * 64u;
    for (uint32_t i = 0; i < 2u * C_pad * CNN_WTAB_DWORDS; i++) out[i] = 0;
    const uint32_t pair0 = cnn_pair_segment_start(C);
    for (uint32_t band = 0; band < 2; band++)
        for (uint32_t col = 0; col < (pair0 < C ? pair0 + 32u : C); col++) {
            const bool pair = col >= pair0;
            const uint32_t c = pair ? pair0 + ((col - pair0) & 15u) : col;
            if (c >= C) continue;
            int *e = out + ((size_t)band * C_pad + col) * CNN_WTAB_DWORDS;
            if (band == (pair ? (col - pair0) >> 4 : 0u)) {
                uint8_t bytes[16] = {0};
                for (int dy = 0; dy < 3; dy++)
                    for (int dx = 0; dx < 3; dx++) bytes[4 * dy + dx] = (uint8_t)w1[9u * c + 3 * dy + dx];
                for (int d = 0; d < 4; d++)
                    e[d] = (int)((uint32_t)bytes[4 * d] | ((uint32_t)bytes[4 * d + 1] << 8) | ((uint32_t)bytes[4 * d + 2] << 16) | ((uint32_t)bytes[4 * d + 3] << 24));
            }
            for (int dyp = 0; dyp < 3; dyp++) {
                const int dy = band ? 2 - dyp : dyp;        // the lower band sees the image upside down
                const int a0 = w2[9u * c + 3 * dy], a1 = w2[9u * c + 3 * dy + 1], a2 = w2[9u * c + 3 * dy + 2];
                e[4 + 2 * dyp] = (int)(((uint32_t)a0 & 0xFFFFu) | ((uint32_t)a1 << 16));
                e[5 + 2 * dyp] = (int)((uint32_t)a2 & 0xFFFFu);
                for (int dx = 0; dx < 3; dx++) e[10 + 3 * dyp + dx] = (int)w3[9u * c + 3 * dy + dx];
            }
        }
}

// acts: int8 [n][acts_stride] (always produced).  feat: int32 [n][4C]; optional when C <= 64, REQUIRED scratch when
// C > 64 (several channel groups: ReLUNorm then runs as its own kernel over the complete vector).
// wtab != nullptr: conv1 on the matrix cores (cnn_front_mfma_kernel, default); nullptr: the all-VALU kernel of round 1.
hipError_t bnmk_cnn_front(const int8_t *images, uint64_t n, const int8_t *w1, const int8_t *w2, const int8_t *w3, const int *wtab,
                          uint32_t C, uint32_t n_shift, int8_t *acts, uint32_t acts_stride, int32_t *feat, bool feat_is_output,
                          uint32_t *counter, uint32_t grab, hipStream_t s) {
    if (!n) return hipSuccess;
    if (C == 0 || C > 256 || n_shift < 4 || n_shift > 15 || acts_stride < 4u * C || (acts_stride & 3u) || n >= (1ull << 31))
        return hipErrorInvalidValue;
    uint64_t blocks = (n + 3) / 4;
    // persistent grid = what is resident: 4 workgroups of 4 waves per CU (both kernels fit 128 VGPRs)
    uint64_t cap = (uint64_t)bnm_num_cus() * 4ull;
    if (blocks > cap) blocks = cap;
    dim3 g((unsigned)blocks), b(256);
    const uint32_t C_pad = (C + 63u) / 64u * 64u;
    // MFMA kernel: images [0, n-1) through the plain instantiation, the last one through the SAFE one (see the kernel)
    const uint64_t n_main = n - 1;
    const uint64_t tail_off = n_main * 256ull;
    // MFMA kernel: batches of `grab` images from word 0 of the counter block (all zero between launches: the waves' first batches
    // are static, the counter hands out what follows); counter == nullptr or grab == 0: fixed shares of single images
    if (!counter || !grab) { counter = nullptr; grab = 1; }
    if (!wtab) {      // round 1's all-VALU kernel
        if (C <= 64) {
            cnn_front_kernel<true><<<g, b, 0, s>>>(images, n, w1, w2, w3, C, 0, n_shift, acts, acts_stride, feat);
            return hipGetLastError();
        }
        if (!feat) return hipErrorInvalidValue;
        for (uint32_t c0 = 0; c0 < C; c0 += 64) {
            cnn_front_kernel<false><<<g, b, 0, s>>>(images, n, w1, w2, w3, C, c0, n_shift, acts, acts_stride, feat);
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) return e;
        }
        return bnmk_relunorm(feat, 4u * C, acts, acts_stride, nullptr, n, s);
    }
    // PAIR segments: images [0, 2 * pairs) two per item, the last one or two through the SAFE one-image instantiation
    const uint64_t pairs = (n - 1) / 2, rest = n - 2 * pairs;
    uint64_t pb = (pairs + 3) / 4;
    if (pb > cap) pb = cap;
    const dim3 gp((unsigned)pb);
    const uint32_t pair0 = cnn_pair_segment_start(C);
    if (C <= 16) {
        if (pairs)
            cnn_front_mfma_kernel<true, false, 1><<<gp, b, 0, s>>>(images, pairs, wtab, C_pad, C, 0, n_shift, acts, acts_stride, feat, 4u * C,
                                                                    counter, grab);
        cnn_front_mfma_kernel<true, true><<<dim3(1), b, 0, s>>>(images + 2 * pairs * 256ull, rest, wtab, C_pad, C, 0, n_shift,
                                                              acts + 2 * pairs * (uint64_t)acts_stride, acts_stride,
                                                              feat ? feat + 2 * pairs * 4ull * C : nullptr, 4u * C, nullptr, 1);
        return hipGetLastError();
    }
    if (C > 32 && C <= 48) {
        // three items per image pair (the third: channels 32..47 of both images); the last one or two images one per item
        if (pairs)
            cnn_front_mfma_kernel<true, false, 2><<<gp, b, 0, s>>>(images, pairs, wtab, C_pad, C, 0, n_shift, acts, acts_stride, feat, 4u * C, counter, grab);
        cnn_front_mfma_kernel<true, true><<<dim3(1), b, 0, s>>>(images + 2 * pairs * 256ull, rest, wtab, C_pad, C, 0, n_shift,
                                                              acts + 2 * pairs * (uint64_t)acts_stride, acts_stride,
                                                              feat ? feat + 2 * pairs * 4ull * C : nullptr, 4u * C, nullptr, 1);
        return hipGetLastError();
    }
    if (C <= 64) {
        if (n_main)
            cnn_front_mfma_kernel<true, false><<<g, b, 0, s>>>(images, n_main, wtab, C_pad, C, 0, n_shift, acts, acts_stride, feat, 4u * C, counter, grab);
        cnn_front_mfma_kernel<true, true><<<dim3(1), b, 0, s>>>(images + tail_off, 1, wtab, C_pad, C, 0, n_shift,
                                                              acts + n_main * (uint64_t)acts_stride, acts_stride,
                                                              feat ? feat + n_main * 4ull * C : nullptr, 4u * C, nullptr, 1);
        return hipGetLastError();
    }
    // several channel segments: every segment writes its int32 features, ReLUNorm runs as its own kernel over the complete vector.
    // The launches of one call share the counter block: each one leaves it zeroed for the next (stream order).
    if (!feat) return hipErrorInvalidValue;
    auto segments = [&](const int8_t *im, uint64_t cnt, int8_t *ac, int32_t *ft) -> hipError_t {
        const uint64_t prs = (cnt - 1) / 2, rst = cnt - 2 * prs, main1 = cnt - 1;
        uint64_t pbs = (prs + 3) / 4, bl1 = (cnt + 3) / 4;
        if (pbs > cap) pbs = cap;
        if (bl1 > cap) bl1 = cap;
        for (uint32_t c0 = 0; c0 < C;) {
            const bool pair = c0 == pair0;
            const uint32_t c_end = pair ? C : (c0 + 64u <= pair0 ? c0 + 64u : (pair0 < C ? pair0 : C));
            if (pair) {
                if (prs)
                    cnn_front_mfma_kernel<false, false, 1><<<dim3((unsigned)pbs), b, 0, s>>>(im, prs, wtab, C_pad, c_end, c0, n_shift, ac, acts_stride, ft,
                                                                                          4u * C, counter, grab);
                cnn_front_mfma_kernel<false, true><<<dim3(1), b, 0, s>>>(im + 2 * prs * 256ull, rst, wtab, C_pad, c_end, c0, n_shift,
                                                                       ac + 2 * prs * (uint64_t)acts_stride, acts_stride, ft + 2 * prs * 4ull * C,
                                                                       4u * C, nullptr, 1);
            } else {
                if (main1)
                    cnn_front_mfma_kernel<false, false><<<dim3((unsigned)bl1), b, 0, s>>>(im, main1, wtab, C_pad, c_end, c0, n_shift, ac, acts_stride, ft,
                                                                                       4u * C, counter, grab);
                cnn_front_mfma_kernel<false, true><<<dim3(1), b, 0, s>>>(im + main1 * 256ull, 1, wtab, C_pad, c_end, c0, n_shift,
                                                                       ac + main1 * (uint64_t)acts_stride, acts_stride, ft + main1 * 4ull * C, 4u * C,
                                                                       nullptr, 1);
            }
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) return e;
            c0 = c_end;
        }
        return bnmk_relunorm(ft, 4u * C, ac, acts_stride, nullptr, cnt, s);
    };
    {
        // one fused launch over the image pairs (pooled outputs in LDS, fused ReLUNorm); the last one or two images take the
        // segment path with the feature scratch
        if (pairs)
            cnn_front_mfma_kernel<true, false, 3><<<gp, b, 0, s>>>(images, pairs, wtab, C_pad, C, 0, n_shift, acts, acts_stride,
                                                                 feat_is_output ? feat : nullptr, 4u * C, counter, grab);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        return segments(images + 2 * pairs * 256ull, rest, acts + 2 * pairs * (uint64_t)acts_stride,
                        feat_is_output ? feat + 2 * pairs * 4ull * C : feat);      // (scratch: the rest's rows sit at its start)
    }
}
