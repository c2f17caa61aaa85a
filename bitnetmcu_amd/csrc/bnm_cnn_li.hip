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
//     channels (the channel kernel: 14);
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
// Any channel count costs exactly its channels (no idle lanes at 24 or 48 channels).  VALU per channel and tile: 496 in the loop
// body (conv1 epilogue 7 x 40, conv2 6 x 21 + the plane split 32 + 10, conv3 and the record ~ 48) + 13 in the final pass, 44 MFMAs.
// 125 VGPRs: four waves per SIMD - a lone wave issues VALU at half rate, and the compiler's MFMA -> VALU wait states (190 per
// channel) need other waves to fill them: 4.4 .. 4.5e8 inferences/s at 64 channels against the channel kernel's 3.4e8.
// Work: tiles of 32 images, one per take, from the launch's counter block (word 0; bnm_device.hpp, work_block_leave_v).
#include <mutex>
#include "bnm_fused_math.hpp"

namespace {

constexpr int LI_WAVES = 16;         // up to four waves per SIMD (128 VGPRs); fewer when the records of a wide model fill the LDS

BNM_DEVICE i32x16 mfma0(const i32x4 &a, const i32x4 &b) { return __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, i32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, 0, 0, 0); }
BNM_DEVICE i32x16 mfma(const i32x4 &a, const i32x4 &b, const i32x16 &c) { return __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0); }

typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2v __attribute__((ext_vector_type(2)));
// relu of two 15-bit values as one packed pair: [max(a, 0) | max(b, 0) << 16]  (v_cvt_pk_i16_i32 + v_pk_max_i16)
BNM_DEVICE uint32_t relu_pair16(int a, int b) {
    const s16x2 p = __builtin_amdgcn_cvt_pk_i16(a, b), z = {0, 0};
    const s16x2 r = __builtin_elementwise_max(p, z);
    return __builtin_bit_cast(uint32_t, r);
}

// Byte planes of pooled values for conv3's operand.  A pooled value arrives as V = 16 x (its 24-bit ReLU'd sum) = (P << 8) | low
// bits: plane p of P is byte p + 1 of V, and v_perm_b32 gathers bytes of two registers - 7 instructions for the three planes of
// four values (two pair gathers per pair, one merge per plane).  Compiler-visible on purpose: an earlier inline-asm version (SDWA
// byte writes) had its outputs allocated to the DEAD rows of an MFMA result still in flight - the padding quad no one reads - and
// hipcc places no hazard wait in front of inline asm: the late MFMA write-back then zeroed a plane-2 dword, about one image in
// 50,000 at full-range weights and only at four waves per SIMD.
struct PlaneQuad { int p0, p1, p2; };
BNM_DEVICE PlaneQuad plane_quad(int A, int B, int C, int D) {
    const uint32_t ab01 = __builtin_amdgcn_perm((uint32_t)B, (uint32_t)A, 0x06020501u);      // [A.1, B.1, A.2, B.2]
    const uint32_t cd01 = __builtin_amdgcn_perm((uint32_t)D, (uint32_t)C, 0x06020501u);
    const uint32_t ab2 = __builtin_amdgcn_perm((uint32_t)B, (uint32_t)A, 0x0c0c0703u);       // [A.3, B.3, 0, 0]
    const uint32_t cd2 = __builtin_amdgcn_perm((uint32_t)D, (uint32_t)C, 0x0c0c0703u);
    return PlaneQuad{(int)__builtin_amdgcn_perm(cd01, ab01, 0x05040100u), (int)__builtin_amdgcn_perm(cd01, ab01, 0x07060302u),
                     (int)__builtin_amdgcn_perm(cd2, ab2, 0x05040100u)};
}
BNM_DEVICE PlaneQuad plane_pair(int A, int B) {      // two values: bytes 0, 1 of the dword, the rest zero
    const uint32_t ab01 = __builtin_amdgcn_perm((uint32_t)B, (uint32_t)A, 0x06020501u);
    return PlaneQuad{(int)__builtin_amdgcn_perm(0u, ab01, 0x0c0c0100u), (int)__builtin_amdgcn_perm(0u, ab01, 0x0c0c0302u),
                     (int)__builtin_amdgcn_perm((uint32_t)B, (uint32_t)A, 0x0c0c0703u)};
}

}  // namespace

// frags: [C][6] fragments of 1 KiB (stage 1 K-steps 0, 1; stage 2; stage 3), lane-linear; bias: [C][2] = {128 sum(w2), 32896 sum(w3)}
// acts: int8 [n][acts_stride], 4 C bytes written per image.  Dynamic LDS: waves x C x 160 bytes (the ReLUNorm records).
__global__ __launch_bounds__(64 * LI_WAVES) void cnn_li_kernel(const int8_t *__restrict__ images, uint32_t n, const i32x4 *__restrict__ frags,
                                                                   const int *__restrict__ bias, uint32_t C, int8_t *__restrict__ acts,
                                                                   uint32_t acts_stride, uint32_t *__restrict__ counter, uint32_t grab) {
    extern __shared__ __attribute__((aligned(16))) uint8_t li_records[];      // per wave: [C][64] uint16 {f0 >> k, f1 >> k} then [C][32] uint8 k (one per image)
    const uint32_t tid = threadIdx.x;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6)), nwaves = blockDim.x >> 6;
    // Everything derived from the lane id (the fragment / record / image addresses) is derived AFRESH per tile, twice, from an opaque
    // copy of tid: hoisted out of the tile loop, as the compiler would, those values are live across the channel loop, and at four
    // waves per SIMD (128 VGPRs) that is what spilled.
#define LI_LANE_VALUES                                                                       \
    uint32_t lane_ = tid;                                                                    \
    asm volatile("" : "+v"(lane_));                                                          \
    const int lane = (int)(lane_ & 63u), j = lane & 31, h = lane >> 5;                       \
    uint16_t *const rec = (uint16_t *)(li_records + wave * C * 160u) + lane;                 \
    uint8_t *const rec_k = li_records + wave * C * 160u + C * 128u + j;

    const uint32_t n_tiles = (n + 31u) >> 5;
    const uint32_t total_waves = gridDim.x * nwaves, wave_id = blockIdx.x * nwaves + wave;
    uint32_t tile = wave_id * grab, left = grab - 1u;      // a wave's first batch is static, later ones come from the counter

    while (tile < n_tiles) {
        int mx = 0;
        {
        LI_LANE_VALUES
        // ---- the tile's 32 images: lane (j, h) keeps bytes 32 s + 16 h .. + 15 of image j for s = 0..7 (K-step s = rows 2s, 2s+1)
        i32x4 b1[8];
        {
            uint32_t img = (tile << 5) + (uint32_t)j;
            if (img >= n) img = n - 1u;      // ragged last tile: rows past the end re-read the last image (their stores are masked)
            const int8_t *p = images + (uint64_t)img * 256u + 16 * h;
#pragma unroll
            for (int s = 0; s < 8; s++) b1[s] = __builtin_nontemporal_load((const i32x4 *)(p + 32 * s));
        }
        for (uint32_t c = 0; c < C; c++) {
            const i32x4 *fc = frags + (uint64_t)c * 6u * 64u + lane;
            const i32x4 a1a = fc[0], a1b = fc[64], a2a = fc[128], a2b = fc[192], a3a = fc[256], a3b = fc[320];
            const int bias2 = bias[2 * c], bias3 = bias[2 * c + 1];
            i32x4 lo[7], hi[7];                 // conv2's operands: the two planes of conv1's outputs, K-step r = conv1 rows 2r, 2r+1
            int P[18];                          // 256 x the pooled conv2 outputs of this lane (+ 4 low bits): P[3 r2 + t] = window 2t + h of pooled row r2
            static_for<0, 8>([&](auto R_) {
                constexpr int r = decltype(R_)::value;
                if constexpr (r < 7) {
                    // ---- stage 1, row pair r: relu(sum) >> 4 of this lane's conv1 row 2r + h, 14 values -> bytes of lo[r] / hi[r]
                    const i32x16 d = mfma(a1b, b1[r + 1], mfma0(a1a, b1[r]));
                    // w = sum >> 4 fits 15 bits: ReLU on packed int16 pairs; a pair's bytes are [lo(2k), hi(2k), lo(2k+1), hi(2k+1)], and
                    // one v_perm_b32 gathers four values' low (high) bytes into a dword of the low (high) plane: 40 VALU per 14 values
                    uint32_t x[8];
#pragma unroll
                    for (int k = 0; k < 7; k++) x[k] = relu_pair16(d[2 * k] >> 4, d[2 * k + 1] >> 4);
                    x[7] = 0;
                    i32x4 l, g;
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        l[q] = (int)__builtin_amdgcn_perm(x[2 * q + 1], x[2 * q], 0x06040200u);
                        g[q] = (int)__builtin_amdgcn_perm(x[2 * q + 1], x[2 * q], 0x07050301u);
                    }
                    lo[r] = l ^ 0x80808080;
                    hi[r] = g;
                }
                if constexpr (r >= 1 && r <= 6) {
                    // ---- stage 2, row pair r2 = r - 1 (needs conv1 row pairs r - 1 and r): pooled, ReLU'd, >> 4
                    constexpr int r2 = r - 1;
                    const i32x16 dl = mfma(a2b, lo[r2 + 1], mfma0(a2a, lo[r2]));
                    const i32x16 dh = mfma(a2b, hi[r2 + 1], mfma0(a2a, hi[r2]));
#pragma unroll
                    for (int t = 0; t < 3; t++) {
                        int s[4];
#pragma unroll
                        for (int e = 0; e < 4; e++) s[e] = dl[4 * t + e] + (dh[4 * t + e] << 8);
                        int m = max(max(s[0], s[1]), s[2]);
                        m = max(max(m, s[3]), -bias2);                       // relu(max + bias) = max(max, -bias) + bias
                        P[3 * r2 + t] = (m + bias2) << 4;                      // (v_add_lshl_u32: the stage's >> 4 turns into "bytes 1..3")
                    }
                }
            });
            // conv3's operands: three byte planes of the 20-bit pooled values; K-step 0 = pooled rows 0..3 (bytes 0..11 of this lane's
            // half), K-step 1 = rows 4, 5 (bytes 0..5); unused bytes meet zero weights
            i32x4 pl[3][2];
            {
                const PlaneQuad q0 = plane_quad(P[0], P[1], P[2], P[3]), q1 = plane_quad(P[4], P[5], P[6], P[7]),
                                q2 = plane_quad(P[8], P[9], P[10], P[11]), q3 = plane_quad(P[12], P[13], P[14], P[15]),
                                q4 = plane_pair(P[16], P[17]);
                pl[0][0] = i32x4{q0.p0, q1.p0, q2.p0, 0} ^ 0x80808080;
                pl[0][1] = i32x4{q3.p0, q4.p0, 0, 0} ^ 0x80808080;
                pl[1][0] = i32x4{q0.p1, q1.p1, q2.p1, 0} ^ 0x80808080;
                pl[1][1] = i32x4{q3.p1, q4.p1, 0, 0} ^ 0x80808080;
                pl[2][0] = i32x4{q0.p2, q1.p2, q2.p2, 0};
                pl[2][1] = i32x4{q3.p2, q4.p2, 0, 0};
            }
            // ---- stage 3: 4x4 outputs -> 2x2 pooled features; this lane holds windows u = h (quad 0) and u = 2 + h (quad 1)
            int f[2];
            {
                // (planes one after the other into eight partial sums: three accumulators live at once cost 16 registers more than the
                // kernel has at four waves per SIMD)
                const i32x16 d0 = mfma(a3b, pl[0][1], mfma0(a3a, pl[0][0]));
                const i32x16 d1 = mfma(a3b, pl[1][1], mfma0(a3a, pl[1][0]));
                int s[8];
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    s[i] = d0[i] + (d1[i] << 8);
                    // (an opaque value: left to itself hipcc reassociates the three-plane sum into two shifts + one v_add3 per value;
                    // two v_lshl_add_u32 do it.  The statement holds no instruction - nothing a late MFMA write-back could meet.)
                    asm volatile("" : "+v"(s[i]));
                }
                const i32x16 d2 = mfma(a3b, pl[2][1], mfma0(a3a, pl[2][0]));
#pragma unroll
                for (int i = 0; i < 8; i++) s[i] += d2[i] << 16;
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    int m = max(max(s[4 * t], s[4 * t + 1]), s[4 * t + 2]);
                    m = max(max(m, s[4 * t + 3]), -bias3);
                    f[t] = (m + bias3) >> 4;
                }
            }
            // ---- the ReLUNorm record of (channel, lane): 8 significant bits of each feature under the running maximum
            // (mx: the IMAGE's running maximum - both lane halves - so that the two lanes of an image share one k byte: 160 bytes of
            // records per channel and wave, which is what lets 16 waves = four per SIMD fit the LDS at 64 channels)
            mx = max(mx, max_with_partner32(max(f[0], f[1])));
            const int shv = (mx >> 7) == 0 ? 0 : 32 - __builtin_clz((uint32_t)(mx >> 7));      // bitlength(mx >> 7)
            const int k = max(shv - 1, 0);
            rec[c * 64u] = (uint16_t)((uint32_t)(f[0] >> k) | ((uint32_t)(f[1] >> k) << 8));
            rec_k[c * 32u] = (uint8_t)k;      // (both lanes of the image write the same byte)
        }
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
#undef LI_LANE_VALUES
}

// ---- host side: the per-channel Toeplitz fragments (tests/cnn_li_model.py states the same matrices in numpy) -----------------
namespace {
inline int li_d_row(int i, int h) { return (i & 3) + 8 * (i >> 2) + 4 * h; }
inline int li_conv3_slot(int prow, int pcol) { return 32 * (prow >> 2) + 16 * (pcol & 1) + 3 * (prow & 3) + (pcol >> 1); }
}  // namespace

// frag_out: C * 6 KiB, bias_out: 2 C ints.  w1 / w2 / w3: [C][9] int8 kernels (row-major 3x3) of conv1 / conv2 / conv3.
void bnm_cnn_li_tables(const int8_t *w1, const int8_t *w2, const int8_t *w3, uint32_t C, int8_t *frag_out, int *bias_out) {
    for (uint32_t c = 0; c < C; c++) {
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
}

// waves per workgroup (one workgroup per CU): the records take C x 160 bytes of LDS per wave; 0 = the kernel does not serve C
uint32_t bnmk_cnn_li_waves(uint32_t C) {
    if (C == 0) return 0;
    const uint32_t w = (160u * 1024u) / (C * 160u);
    return w >= (uint32_t)LI_WAVES ? (uint32_t)LI_WAVES : (w >= 6u ? w : 0u);      // fewer than six waves: the channel kernel serves the model
}

hipError_t bnmk_cnn_front_li(const int8_t *images, uint64_t n, const void *frags, const int *bias, uint32_t C, int8_t *acts,
                             uint32_t acts_stride, uint32_t *counter, uint32_t grab, hipStream_t s) {
    if (!n) return hipSuccess;
    const uint32_t waves = bnmk_cnn_li_waves(C);
    if (!waves || !counter || acts_stride < 4u * C || (acts_stride & 3u) || n >= (1ull << 31)) return hipErrorInvalidValue;
    if (!grab) grab = 1;
    static std::mutex mu;
    static bool allowed[64] = {};
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
    {
        std::lock_guard<std::mutex> g(mu);
        if (dev < 64 && !allowed[dev]) {
            if (hipError_t e = hipFuncSetAttribute((const void *)cnn_li_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); e != hipSuccess) return e;
            allowed[dev] = true;
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
    cnn_li_kernel<<<dim3((unsigned)blocks), dim3(64 * waves_now), waves_now * C * 160u, s>>>(images, (uint32_t)n, (const i32x4 *)frags, bias, C, acts,
                                                                                  acts_stride, counter, grab);
    return hipGetLastError();
}
