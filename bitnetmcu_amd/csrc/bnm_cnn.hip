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

// acts: int8 [n][4C] (always produced).  feat: int32 [n][4C]; optional when C <= 64, REQUIRED scratch when
// C > 64 (several channel groups: ReLUNorm then runs as its own kernel over the complete vector).
hipError_t bnmk_cnn_front(const int8_t *images, uint64_t n, const int8_t *w1, const int8_t *w2, const int8_t *w3,
                          uint32_t C, uint32_t n_shift, int8_t *acts, uint32_t acts_stride, int32_t *feat, hipStream_t s) {
    if (!n) return hipSuccess;
    if (C == 0 || C > 256 || n_shift < 4 || n_shift > 31 || acts_stride < 4u * C || (acts_stride & 3u)) return hipErrorInvalidValue;
    uint64_t blocks = (n + 3) / 4;
    uint64_t cap = (uint64_t)bnm_num_cus() * 4ull;
    if (blocks > cap) blocks = cap;
    dim3 g((unsigned)blocks), b(256);
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

