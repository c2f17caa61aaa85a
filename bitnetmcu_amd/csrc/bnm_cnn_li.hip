// CNN front end, lane = image formulation: conv 3x3 -> conv 3x3 -> pool -> conv 3x3 -> pool per channel
// (BitNetMCU_MNIST_dll.c:48-91; BitNetMCU_inference.c:238-277 processconv33ReLU, :300-322 processmaxpool22) and the ReLUNorm over
// an image's 4 C features (:23-72), ALL THREE convolutions on the matrix cores.  gfx950 (CDNA4 / MI355X) only.
//
// cnn_front_mfma_kernel (bnm_cnn.hip) puts a CHANNEL in every lane: conv1 is a GEMM there, conv2 / conv3 run on v_dot2 / 24-bit
// mads - 864 + 144 of its 1,602 VALU instructions per image - and channel counts that do not fill 32 lanes idle the rest.  Here a
// lane is an IMAGE: a wave owns a tile of 32 images (the MFMA's N dimension) and walks the channels; per channel every stage is
// a banded Toeplitz product  out[positions x images] = T_c[positions x inputs] . in[inputs x images]  on v_mfma_i32_32x32x32_i8:
//   * a 3x3 kernel is translation invariant, so with the K axis laid out in ROW PAIRS (two input rows per 32-slot K-step) every
//     output row pair of a stage uses the SAME two A fragments: 6 fragments = 6 KiB per channel for the three stages (L2-resident
//     table, tests/cnn_li_model.py builds the same matrices), 14 + 24 + 6 = 44 MFMAs per channel and tile = 88 per image at 64
//     channels (the channel kernel: 14) - 14 + 24 + 4 = 42 for models whose weights bound the pooled conv2 outputs below 2^16
//     (every CNN of the reference's zoo): conv3's third operand plane is then not in the kernel (template parameter P2);
//   * operands wider than int8 travel as int8 PLANES with separate accumulators (the plane weight 256 does not fit an int8
//     weight): conv1's 14-bit outputs as two planes, the pooled 20-bit conv2 outputs as three; the low planes are offset by
//     -128 (byte ^ 0x80, four bytes per v_xor) and the offset's contribution 128 sum(w) is a per-channel constant added behind
//     the pooling maximum;
//   * the D rows of every stage are ordered so that (a) a lane's 14 conv1 values are the 14 bytes of ITS half of conv2's operand,
//     (b) a D-register quad is one 2x2 pooling window (the maximum is in-lane) and (c) a lane's pooled values are bytes of ITS
//     half of conv3's operand: no LDS, no cross-lane traffic between the stages;
//   * the ReLUNorm over all 4 C features stays fused without holding 2 C int32 per lane: per channel a lane writes {f0 >> k, f1 >> k}
//     to LDS and the image one byte k = max(bitlength(mx >> 7) - 1, 0) from the IMAGE's running maximum mx (both lane halves: 160
//     bytes per channel and wave) - at most 8 significant bits are kept, which is exact for the final shift s >= k + 1
//     ((f + (1 << s >> 1)) >> s == ((f >> k) + (1 << (s-k) >> 1)) >> (s-k)).
// Any channel count costs exactly its channels (no idle lanes at 24 or 48 channels).  VALU per channel and tile: 447 in the loop
// body (conv1 epilogue 7 x 33 - one SDWA shift per value writes its half of a packed int16 pair, round 5 -, conv2 6 x 21 + the plane
// split 32 + 10, conv3 and the record ~ 48) + the final pass, 44 MFMAs.
// 125 VGPRs: four waves per SIMD - a lone wave issues VALU at half rate, and the compiler's MFMA -> VALU wait states (190 per
// channel) need other waves to fill them.
// Work: tiles of 32 images, one per take, from the launch's counter block (word 0; bnm_device.hpp, work_block_leave_v).
#include <mutex>
#include "bnm_cnn_li_tile.hpp"

// frags: [C][6] fragments of 1 KiB (stage 1 K-steps 0, 1; stage 2; stage 3), lane-linear; bias: [C][2] = {128 sum(w2), 32896 sum(w3)}
// acts: int8 [n][acts_stride], 4 C bytes written per image.  Dynamic LDS: waves x C x 160 bytes (the ReLUNorm records).
// P2: conv3's third operand plane (pooled conv2 values of 2^16 and more) is in the kernel; false for models whose weights rule such values out
template <bool P2>
__global__ __launch_bounds__(64 * LI_WAVES) void cnn_li_kernel(const int8_t *__restrict__ images, uint32_t n, const i32x4 *__restrict__ frags,
                                                                   const int *__restrict__ bias, uint32_t C, int8_t *__restrict__ acts,
                                                                   uint32_t acts_stride, uint32_t *__restrict__ counter, uint32_t grab) {
    extern __shared__ __attribute__((aligned(16))) uint8_t li_records[];      // per wave: [C][64] uint16 {f0 >> k, f1 >> k} then [C][32] uint8 k (one per image)
    constexpr bool LI_PLANE2 = P2;
    const uint32_t tid = threadIdx.x;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6)), nwaves = blockDim.x >> 6;
    const uint32_t n_tiles = (n + 31u) >> 5;
    const uint32_t total_waves = gridDim.x * nwaves, wave_id = blockIdx.x * nwaves + wave;
    uint32_t tile = wave_id * grab, left = grab - 1u;      // a wave's first batch is static, later ones come from the counter

    while (tile < n_tiles) {
        int mx = 0;
        {
        LI_LANE_VALUES
#include "bnm_cnn_li_tile_body.inc"
        }
        // ---- ReLUNorm over the image's 4 C features (BitNetMCU_inference.c:23-72): the image's maximum, one shift
        LI_LANE_VALUES
        const int s_all = (mx >> 7) == 0 ? 0 : 32 - __builtin_clz((uint32_t)(mx >> 7));
        const uint32_t img_out = (tile << 5) + (uint32_t)j;
        const bool valid = img_out < n;
        int8_t *row = acts + (uint64_t)(valid ? img_out : n - 1u) * acts_stride;
        for (uint32_t c = 0; c < C; c++) {
            const uint32_t w = rec[c * 64u];
            // both features of the record at once, as 16-bit halves: (f + (1 << d >> 1)) >> d == (((2 f) >> d) + 1) >> 1 for every
            // d >= 0 (f < 256); a packed shift takes four bits of its amount, and from d = 10 on the result is 0 anyway
            const uint32_t d = (uint32_t)min(s_all - (int)rec_k[c * 32u], 15);
            const u16x2v dd = {(unsigned short)d, (unsigned short)d}, one = {1, 1}, top = {127, 127};
            u16x2v v = __builtin_bit_cast(u16x2v, __builtin_amdgcn_perm(0u, w, 0x0c010c00u) << 1);      // [2 f0, 2 f1]
            v = __builtin_elementwise_min((u16x2v)(((v >> dd) + one) >> one), top);
            // act bytes of channel c: [window 0, 1, 2, 3] = [o0 of half 0, o0 of half 1, o1 of half 0, o1 of half 1]
            const int x = (int)(__builtin_bit_cast(uint32_t, v) << (8 * h));
            auto both = __builtin_amdgcn_permlane32_swap(x, x, false, false);
            const int word = (int)both[0] | (int)both[1];
            if (valid && h == 0) *(int *)(row + 4u * c) = word;
        }
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

// ---- host side: the per-channel Toeplitz fragments (tests/cnn_li_model.py states the same matrices in numpy) -----------------
namespace {
inline int li_d_row(int i, int h) { return (i & 3) + 8 * (i >> 2) + 4 * h; }
inline int li_conv3_slot(int prow, int pcol) { return 32 * (prow >> 2) + 16 * (pcol & 1) + 3 * (prow & 3) + (pcol >> 1); }
}  // namespace

// frag_out: C * 6 KiB, bias_out: 2 C ints.  w1 / w2 / w3: [C][9] int8 kernels (row-major 3x3) of conv1 / conv2 / conv3.
// Returns whether conv3 needs its third operand plane: false when the weights bound EVERY pooled conv2 output below 2^16 - conv1's
// sums reach at most 127 sum(w1+) + 128 sum(|w1-|) (inputs in -128 .. 127), its outputs m1 = that >> 4 (BitNetMCU_inference.c:261-271:
// ReLU, then the shift); conv2 sees inputs in 0 .. m1, so its outputs reach at most (m1 sum(w2+)) >> 4, and pooling takes a maximum.
bool bnm_cnn_li_tables(const int8_t *w1, const int8_t *w2, const int8_t *w3, uint32_t C, int8_t *frag_out, int *bias_out, bool *sums16) {
    bool plane2 = false, s16 = true;
    for (uint32_t c = 0; c < C; c++) {
        {
            int64_t p1 = 0, n1 = 0, p2 = 0;
            for (int k = 0; k < 9; k++) {
                const int a = w1[9 * c + k], b = w2[9 * c + k];
                if (a > 0) p1 += a; else n1 -= a;
                if (b > 0) p2 += b;
            }
            const int64_t m1 = (127 * p1 + 128 * n1) >> 4, m2 = (m1 * p2) >> 4;
            plane2 = plane2 || m2 >= 65536;
            s16 = s16 && 127 * p1 + 128 * n1 <= 65535;      // conv1's largest possible sum (inputs -128 .. 127)
        }
        int8_t A[3][32][64] = {};
        const int8_t *k1 = w1 + 9 * c, *k2 = w2 + 9 * c, *k3 = w3 + 9 * c;
        for (int h = 0; h < 2; h++) {
            for (int i = 0; i < 14; i++)            // stage 1: D register i of half h = conv1 position (row 2r + h, col i)
                for (int dy = 0; dy < 3; dy++)
                    for (int dx = 0; dx < 3; dx++) {
                        const int row = h + dy;
                        A[0][li_d_row(i, h)][32 * (row >> 1) + 16 * (row & 1) + i + dx] = k1[3 * dy + dx];
                    }
            for (int t = 0; t < 3; t++)             // stage 2: quad t of half h = pooling window 2t + h of row pair r2
                for (int e = 0; e < 4; e++)
                    for (int dy = 0; dy < 3; dy++)
                        for (int dx = 0; dx < 3; dx++) {
                            const int row = (e >> 1) + dy, col = 2 * (2 * t + h) + (e & 1) + dx;
                            A[1][li_d_row(4 * t + e, h)][32 * (row >> 1) + 16 * (row & 1) + col] = k2[3 * dy + dx];
                        }
            for (int t = 0; t < 2; t++)             // stage 3: quad t of half h = pooling window (t, h) of the 4x4 output
                for (int e = 0; e < 4; e++)
                    for (int dy = 0; dy < 3; dy++)
                        for (int dx = 0; dx < 3; dx++)
                            A[2][li_d_row(4 * t + e, h)][li_conv3_slot(2 * t + (e >> 1) + dy, 2 * h + (e & 1) + dx)] = k3[3 * dy + dx];
        }
        for (int st = 0; st < 3; st++)
            for (int s = 0; s < 2; s++)
                for (int lane = 0; lane < 64; lane++)
                    for (int b = 0; b < 16; b++)
                        frag_out[(((size_t)c * 6 + (size_t)st * 2 + s) * 64 + lane) * 16 + b] = A[st][lane & 31][32 * s + 16 * (lane >> 5) + b];
        int s2 = 0, s3 = 0;
        for (int k = 0; k < 9; k++) { s2 += k2[k]; s3 += k3[k]; }
        bias_out[2 * c] = 128 * s2;
        bias_out[2 * c + 1] = (128 + 32768) * s3;
    }
    if (sums16) *sums16 = s16;
    return plane2;
}

// waves per workgroup (one workgroup per CU): the records take C x 160 bytes of LDS per wave; 0 = the kernel does not serve C
uint32_t bnmk_cnn_li_waves(uint32_t C) {
    if (C == 0) return 0;
    const uint32_t w = (160u * 1024u) / (C * 160u);
    return w >= (uint32_t)LI_WAVES ? (uint32_t)LI_WAVES : (w >= 6u ? w : 0u);      // fewer than six waves: the channel kernel serves the model
}

hipError_t bnmk_cnn_front_li(const int8_t *images, uint64_t n, const void *frags, const int *bias, uint32_t C, bool plane2, int8_t *acts,
                             uint32_t acts_stride, uint32_t *counter, uint32_t grab, hipStream_t s) {
    if (!n) return hipSuccess;
    const uint32_t waves = bnmk_cnn_li_waves(C);
    if (!waves || !counter || acts_stride < 4u * C || (acts_stride & 3u) || n >= (1ull << 31)) return hipErrorInvalidValue;
    if (!grab) grab = 1;
    auto fn = plane2 ? cnn_li_kernel<true> : cnn_li_kernel<false>;
    static std::mutex mu;
    static bool allowed[2][64] = {};
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
    {
        std::lock_guard<std::mutex> g(mu);
        if (dev < 64 && !allowed[plane2][dev]) {
            if (hipError_t e = hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); e != hipSuccess) return e;
            allowed[plane2][dev] = true;
        }
    }
    // a call with fewer tiles than the chip has wave slots spreads them over the CUs first: a wave's walk over the channels takes the
    // same time alone or beside one other wave on its SIMD (a lone wave issues VALU at half rate), and twice as long beside three
    const uint64_t tiles = (n + 31) / 32, cap = (uint64_t)bnm_num_cus();
    uint32_t waves_now = (uint32_t)((tiles + cap - 1) / cap);
    waves_now = waves_now < 1u ? 1u : waves_now > waves ? waves : waves_now;
    const uint64_t per_block = (uint64_t)waves_now * grab;
    uint64_t blocks = (tiles + per_block - 1) / per_block;
    if (blocks > cap) blocks = cap;
    fn<<<dim3((unsigned)blocks), dim3(64 * waves_now), waves_now * C * 160u, s>>>(images, (uint32_t)n, (const i32x4 *)frags, bias, C, acts,
                                                                                  acts_stride, counter, grab);
    return hipGetLastError();
}
