// SURVEY.md §8(f) row 4, the convolution front of the reference's CNNMNIST (models.py:109-119: the model trainingparameters.yaml
// names) in ONE kernel: float32 16x16 images in, the 4 C float32 features behind Flatten out - BitConv2d(1 -> C, 3x3) -> ReLU ->
// BitConv2d(C -> C, 3x3, depthwise) -> ReLU -> MaxPool 2x2 -> BitConv2d(C -> C, 3x3, depthwise) -> ReLU -> MaxPool 2x2, every
// BitConv2d with NormType 'None' (BitNetMCU.py:285-305: activation_quant per image ROW of each plane - the maximum over the last
// dimension, :125-127 -, weight_quant, F.conv2d).  Layer by layer (bnm_qat.hip: one workgroup per image and channel, every plane
// through HBM) the front runs at 3-7 x 10^6 images/s and is 99.9 % of the model's forward pass.  gfx950 only.  Floating point:
// parity with the reference module within the tolerances tests/test_gpu_qat_cnn.py states, not bit-exact.
//
//   qat_cnn_prep_kernel    one workgroup per layer, once per call: the taps w_int / w_scale (the straight-through forward value) of
//                          all channels as floats [layer][channel][9]
//   qat_cnn_front_kernel   a LANE owns two channels of one image - a wave IPW images of C / 2 channel pairs - and walks the whole
//                          front for them in registers, both channels side by side in the halves of v_pk_fma_f32:
//     * the wave's images arrive as one float4 per lane and image (a row of 16 = four lanes: the row maximum is two DPP steps),
//       are quantised with the reference's own float32 operations (127 / max, round, / scale) and parked in LDS, from where every
//       lane of the image reads the rows back as broadcasts, two rows ahead of their use;
//     * conv1's taps are (w_a, w_b) pairs against a splat of the input value (op_sel picks the half: no extra move), the depthwise
//       layers' operands are (x_a, x_b) pairs: nine v_pk_fma_f32 per pair of outputs, 3,204 per lane and image;
//     * rows are produced in order and consumed at once - conv1 row r completes the three-row window of conv2 row r - 2, two of
//       those make a pooled row, three pooled rows a conv3 row - so a lane holds three rows of each stage, never a plane; the
//       loops are unrolled completely (every index is a constant: the windows are registers);
//     * activation_quant of a row is per lane: the row's maximum m with v_max3_f32, u = clamp(y / m, 0, 1) (v_pk_mul_f32 with the
//       clamp modifier: the ReLU costs nothing), q = rne(127 u) as (127 u + 1.5 * 2^23) - 1.5 * 2^23, x_quant = q m / 127;
//     * a lane's eight features are 32 consecutive bytes of the image's row of 4 C floats.
// Bound: VALU issue (~4,900 instructions per lane and image, two thirds of them the packed multiply-adds; measured: 0.84 busy); HBM
// sees 1 KiB + 16 C bytes per image.
#include "bnm_qat_math.hpp"
#include "bnm_quantise_f32.hpp"
#include <atomic>

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr float QC_MAGIC = 12582912.0f;      // 1.5 * 2^23

BNM_DEVICE f32x2 splat(float v) { return f32x2{v, v}; }

// ReLU + activation_quant of one row of W pairs (channel a in .x, channel b in .y): in place, y -> x_int / x_scale.
// With m = max(row maximum of relu(y), 1e-5):  x_int = rne(relu(y) 127 / m) = rne(127 u),  u = clamp(y / m, 0, 1)  - the clamp is the
// ReLU (and free: the multiplication's own output modifier); x_int / x_scale = x_int m / 127.  v_rcp_f32 and two multiplications where
// the reference divides: a rounding of the scale (see the file header on what that can do).
template <int W>
BNM_DEVICE void relu_quant_row(f32x2 (&row)[W]) {
    float ma = 1e-5f, mb = 1e-5f;
#pragma unroll
    for (int c = 0; c + 1 < W; c += 2) {
        ma = __builtin_fmaxf(__builtin_fmaxf(ma, row[c][0]), row[c + 1][0]);      // (v_max3_f32)
        mb = __builtin_fmaxf(__builtin_fmaxf(mb, row[c][1]), row[c + 1][1]);
    }
    static_assert(W % 2 == 0, "even rows");
    // (the reciprocals pass through an ordinary multiplication before the hand-written instruction reads them: hipcc does not see into
    // inline assembly when it places the wait state a transcendental result needs in front of its first VALU reader)
    f32x2 one = splat(1.0f);
    asm volatile("" : "+v"(one));
    const f32x2 rm = f32x2{__builtin_amdgcn_rcpf(ma), __builtin_amdgcn_rcpf(mb)} * one;
    const f32x2 inv = f32x2{ma, mb} * splat(1.0f / 127.0f);
    const f32x2 magic = splat(QC_MAGIC), c127 = splat(127.0f);
#pragma unroll
    for (int c = 0; c < W; c++) {
        f32x2 u;
        asm("v_pk_mul_f32 %0, %1, %2 clamp" : "=v"(u) : "v"(row[c]), "v"(rm));
        const f32x2 t = __builtin_elementwise_fma(u, c127, magic);
        row[c] = (t - magic) * inv;
    }
}

// one output row of a depthwise 3x3 convolution from three input rows of pairs
template <int WOUT>
BNM_DEVICE void conv_row_pairs(const f32x2 *r0, const f32x2 *r1, const f32x2 *r2, const f32x2 (&w)[9], f32x2 (&out)[WOUT]) {
#pragma unroll
    for (int c = 0; c < WOUT; c++) out[c] = f32x2{0.0f, 0.0f};
#pragma unroll
    for (int dx = 0; dx < 3; dx++)
#pragma unroll
        for (int c = 0; c < WOUT; c++) out[c] = __builtin_elementwise_fma(r0[c + dx], w[dx], out[c]);
#pragma unroll
    for (int dx = 0; dx < 3; dx++)
#pragma unroll
        for (int c = 0; c < WOUT; c++) out[c] = __builtin_elementwise_fma(r1[c + dx], w[3 + dx], out[c]);
#pragma unroll
    for (int dx = 0; dx < 3; dx++)
#pragma unroll
        for (int c = 0; c < WOUT; c++) out[c] = __builtin_elementwise_fma(r2[c + dx], w[6 + dx], out[c]);
}

// 2x2 max pool of two rows of 2 WOUT pairs (the ReLU is the quantiser's, or the caller's)
template <int WOUT>
BNM_DEVICE void pool_rows(const f32x2 *top, const f32x2 *bottom, f32x2 (&out)[WOUT]) {
#pragma unroll
    for (int c = 0; c < WOUT; c++) {
        out[c][0] = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(top[2 * c][0], top[2 * c + 1][0]), bottom[2 * c][0]), bottom[2 * c + 1][0]);
        out[c][1] = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(top[2 * c][1], top[2 * c + 1][1]), bottom[2 * c][1]), bottom[2 * c + 1][1]);
    }
}

}  // namespace

// ---- taps: w_int / w_scale of every channel, [layer][channel][9] floats ------------------------------------------------------------
struct QatCnnPrepArgs {
    const float *w[3];
    const float *s[3];
    int qt[3];
};

__global__ __launch_bounds__(1024) void qat_cnn_prep_kernel(QatCnnPrepArgs a, uint32_t channels, float *__restrict__ taps) {
    __shared__ double sa[16], sw[16];
    __shared__ float stats[2];
    const uint32_t l = blockIdx.x, count = channels * 9u;
    const float *w = a.w[l];
    const int qt = a.qt[l];
    double x = 0.0, y = 0.0;      // mean |w| and mean w of the whole tensor (Ternary's scale, Binary's offset), fixed reduction order
    for (uint32_t i = threadIdx.x; i < count; i += 1024u) {
        const float v = w[i];
        x += fabsf(v);
        y += v;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        x += __shfl_xor(x, off);
        y += __shfl_xor(y, off);
    }
    if ((threadIdx.x & 63) == 0) { sa[threadIdx.x >> 6] = x; sw[threadIdx.x >> 6] = y; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double ta = 0.0, tb = 0.0;
        for (int i = 0; i < 16; i++) { ta += sa[i]; tb += sw[i]; }
        stats[0] = (float)(ta / (double)count);
        stats[1] = (float)(tb / (double)count);
    }
    __syncthreads();
    const float sc = qat_weight_scale(qt, a.s[l][0], stats[0]);
    for (uint32_t i = threadIdx.x; i < count; i += 1024u) {
        const float u = qat_weight_level(qt, w[i], sc, stats[1]);
        taps[l * count + i] = qt == BNM_QAT_NONE ? u : __fdiv_rn(u, sc);
    }
}

// ---- the front -----------------------------------------------------------------------------------------------------------------------
// one row of W pairs -> a saved plane in CHANNELS-LAST order ([n][rows][W][C] floats): the pair (channel 2 p, channel 2 p + 1) of one
// position is 8 consecutive bytes, the C / 2 lanes of an image write 4 C consecutive bytes per position - no repacking, whole lines
template <int W>
BNM_DEVICE void store_plane_row(const f32x2 (&row)[W], float *at, uint32_t channels) {
#pragma unroll
    for (int c = 0; c < W; c++) __builtin_nontemporal_store(row[c], (f32x2 *)(at + (size_t)c * channels));
}

// IPW images per wave (a power of two, IPW * C / 2 <= 64 lanes); four waves per workgroup, each with two IPW KiB staging buffers in LDS.
// SAVE (the training form): the three convolutions' outputs BEFORE their ReLU leave as well, channels-last - y1 [n][14][14][C],
// y2 [n][12][12][C], y3 [n][4][4][C] (torch.channels_last tensors of shape [n, C, k, k]) -: with x and the weights all a backward pass
// needs (ReLU masks, pooling arguments and every layer's input are functions of them).  356 C floats per image: bound by those writes.
template <int IPW, bool SAVE>
__global__ __launch_bounds__(256) void qat_cnn_front_kernel(const float *__restrict__ x, uint64_t n, const float *__restrict__ taps,
                                                            uint32_t channels, float *__restrict__ features, float *__restrict__ y1_out,
                                                            float *__restrict__ y2_out, float *__restrict__ y3_out) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t pairs = channels >> 1;
    uint32_t slot = lane / pairs;
    const bool active = slot < (uint32_t)IPW;
    slot = active ? slot : 0u;
    const uint32_t p = active ? lane - slot * pairs : 0u;
    char *const stage = smem + wave * (2u * IPW * 1024u);

    // this lane's taps: (channel 2 p, channel 2 p + 1) pairs
    f32x2 w1[9], w2[9], w3[9];
    {
        const float *t = taps + (size_t)(2u * p) * 9u;
        const size_t layer = (size_t)channels * 9u;
#pragma unroll
        for (int k = 0; k < 9; k++) {
            w1[k] = f32x2{t[k], t[9 + k]};
            w2[k] = f32x2{t[layer + k], t[layer + 9 + k]};
            w3[k] = f32x2{t[2 * layer + k], t[2 * layer + 9 + k]};
        }
    }

    const uint64_t groups = (n + (uint64_t)IPW - 1ull) / (uint64_t)IPW;
    const uint64_t total_waves = (uint64_t)gridDim.x * (blockDim.x >> 6), wave_id = (uint64_t)blockIdx.x * (blockDim.x >> 6) + wave;

    f32x4 land[IPW];
    auto load_group = [&](uint64_t g) {
#pragma unroll
        for (int i = 0; i < IPW; i++) {
            uint64_t img = g * (uint64_t)IPW + (uint64_t)i;
            img = img < n ? img : n - 1ull;
            land[i] = __builtin_nontemporal_load((const f32x4 *)(x + img * 256ull + 4u * lane));
        }
    };
    // conv1's activation_quant (NormType 'None'): per row of 16 = four lanes, the reference's float32 operations one by one
    auto quantise_group = [&](char *buf) {
#pragma unroll
        for (int i = 0; i < IPW; i++) {
            const f32x4 v = land[i];
            uint32_t m = absmax4_bits(v);
            m = umax(m, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0xB1, 0xf, 0xf, false));      // quad_perm [1,0,3,2]
            m = umax(m, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)m, 0x4E, 0xf, 0xf, false));      // quad_perm [2,3,0,1]
            const float scale = __fdiv_rn(127.0f, __builtin_fmaxf(__uint_as_float(m), 1e-5f));
            f32x4 q;
#pragma unroll
            for (int b = 0; b < 4; b++) q[b] = __fdiv_rn(__builtin_rintf(__fmul_rn(v[b], scale)), scale);
            *(f32x4 *)(buf + i * 1024 + 16u * lane) = q;
        }
    };

    uint64_t g = wave_id;
    uint32_t cur = 0;
    if (g < groups) {
        load_group(g);
        quantise_group(stage);
    }
    for (; g < groups; g += total_waves) {
        const uint64_t gn = g + total_waves;
        if (gn < groups) load_group(gn);
        const char *img = stage + cur * (IPW * 1024u) + slot * 1024u;
        const uint64_t image = g * (uint64_t)IPW + (uint64_t)slot;
        const bool mine = active && image < n;
        const uint64_t saved = image < n ? image : 0ull;      // (the image's index in the saved planes)

        // rolling windows: three input rows, three rows of conv1's quantised outputs, two conv2 rows, three pooled rows, two conv3 rows
        f32x4 xin[3][4];
        f32x2 q1[3][14], c2[2][12], q2[3][6], c3[2][4], out[2][2];
#pragma unroll
        for (int r = 0; r < 2; r++)
#pragma unroll
            for (int k = 0; k < 4; k++) xin[r][k] = *(const f32x4 *)(img + 64 * r + 16 * k);
        static_for<0, 14>([&](auto RI) {
            constexpr int r = decltype(RI)::value;
            // input row r + 2 is asked for two rows ahead of its first use (a lone wave per SIMD has nobody to hide an LDS round trip
            // behind; the scheduling barrier keeps hipcc from sinking the reads next to their use, which it otherwise does)
#pragma unroll
            for (int k = 0; k < 4; k++) xin[(r + 2) % 3][k] = *(const f32x4 *)(img + 64 * (r + 2) + 16 * k);
            __builtin_amdgcn_sched_barrier(0);
            // ---- conv1 row r: taps are pairs, the input value is a splat ----
            f32x2 (&y1)[14] = q1[r % 3];
#pragma unroll
            for (int c = 0; c < 14; c++) y1[c] = f32x2{0.0f, 0.0f};
#pragma unroll
            for (int dy = 0; dy < 3; dy++) {
                const f32x4 (&row)[4] = xin[(r + dy) % 3];
#pragma unroll
                for (int dx = 0; dx < 3; dx++)
#pragma unroll
                    for (int c = 0; c < 14; c++) y1[c] = __builtin_elementwise_fma(splat(row[(c + dx) >> 2][(c + dx) & 3]), w1[3 * dy + dx], y1[c]);
            }
            if constexpr (SAVE) {
                if (mine) store_plane_row<14>(y1, y1_out + ((saved * 14ull + r) * 14ull) * channels + 2u * p, channels);
            }
            relu_quant_row<14>(y1);
            if constexpr (r >= 2) {
                // ---- conv2 row r2 = r - 2 ----
                constexpr int r2 = r - 2;
                conv_row_pairs<12>(q1[r2 % 3], q1[(r2 + 1) % 3], q1[(r2 + 2) % 3], w2, c2[r2 & 1]);
                if constexpr (SAVE) {
                    if (mine) store_plane_row<12>(c2[r2 & 1], y2_out + ((saved * 12ull + r2) * 12ull) * channels + 2u * p, channels);
                }
                if constexpr (r2 & 1) {
                    // ---- pooled row k (ReLU inside its quantiser), conv3's activation_quant ----
                    constexpr int k = r2 >> 1;
                    pool_rows<6>(c2[0], c2[1], q2[k % 3]);
                    relu_quant_row<6>(q2[k % 3]);
                    if constexpr (k >= 2) {
                        constexpr int m = k - 2;
                        conv_row_pairs<4>(q2[m % 3], q2[(m + 1) % 3], q2[(m + 2) % 3], w3, c3[m & 1]);
                        if constexpr (SAVE) {
                            if (mine) store_plane_row<4>(c3[m & 1], y3_out + ((saved * 4ull + m) * 4ull) * channels + 2u * p, channels);
                        }
                        if constexpr (m & 1) {
                            pool_rows<2>(c3[0], c3[1], out[m >> 1]);
#pragma unroll
                            for (int c = 0; c < 2; c++) out[m >> 1][c] = f32x2{__builtin_fmaxf(out[m >> 1][c][0], 0.0f), __builtin_fmaxf(out[m >> 1][c][1], 0.0f)};
                        }
                    }
                }
            }
        });
        // the next group's images (loaded at the top of this iteration) are quantised BEFORE this group's features are stored: loads
        // and stores retire through one counter in order, and a wait for the loads behind the stores would sit out the stores' way to HBM
        if (gn < groups) quantise_group(stage + (cur ^ 1u) * (IPW * 1024u));
        __builtin_amdgcn_sched_barrier(0);
        // ---- features: channel a's 2 x 2, then channel b's: 32 consecutive bytes of the image's row ----
        if (mine) {
            float *f = features + image * (uint64_t)(4u * channels) + 8u * p;
            __builtin_nontemporal_store(f32x4{out[0][0][0], out[0][1][0], out[1][0][0], out[1][1][0]}, (f32x4 *)f);
            __builtin_nontemporal_store(f32x4{out[0][0][1], out[0][1][1], out[1][0][1], out[1][1][1]}, (f32x4 *)(f + 4));
        }
        cur ^= 1u;
    }
}

// ---- host side ---------------------------------------------------------------------------------------------------------------------
bool bnmk_qat_cnn_front_supported(uint32_t channels, const uint32_t *s_count, const int *quant_types) {
    if (channels < 16u || channels > 128u || (channels & 1u)) return false;
    for (int l = 0; l < 3; l++) {
        if (s_count[l] != 1u) return false;      // per-tensor clipping scalars (the conv layers' PerOutput form is per kernel row: layer by layer)
        if (quant_types[l] <= BNM_QAT_NONE || quant_types[l] > BNM_QAT_8BIT) return false;      // ('None' skips activation_quant as well, BitNetMCU.py:294-295)
    }
    return true;
}

size_t bnmk_qat_cnn_front_workspace_bytes(uint32_t channels) { return (size_t)3u * channels * 9u * sizeof(float); }

template <int IPW, bool SAVE>
static hipError_t qat_cnn_front_launch(const float *x, uint64_t n, const float *taps, uint32_t channels, float *features, float *y1, float *y2,
                                       float *y3, hipStream_t st) {
    const uint64_t groups = (n + IPW - 1) / IPW;
    const size_t lds = 4u * 2u * IPW * 1024u;
    // persistent waves: as many workgroups as are resident at once (one per CU at this kernel's ~300 registers; asked once per
    // device, not assumed)
    static std::atomic<int> resident[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    int per_cu = resident[dev].load(std::memory_order_relaxed);
    if (per_cu < 1) {
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, qat_cnn_front_kernel<IPW, SAVE>, 256, lds) != hipSuccess || per_cu < 1) per_cu = 1;
        resident[dev].store(per_cu, std::memory_order_relaxed);
    }
    uint64_t blocks = (groups + 3u) / 4u;
    const uint64_t cap = (uint64_t)bnm_num_cus() * (uint64_t)per_cu;
    if (blocks > cap) blocks = cap;
    qat_cnn_front_kernel<IPW, SAVE><<<dim3((unsigned)blocks), dim3(256), lds, st>>>(x, n, taps, channels, features, y1, y2, y3);
    return hipGetLastError();
}

template <bool SAVE>
static hipError_t qat_cnn_front_go(const float *x, uint64_t n, const float *taps, uint32_t channels, float *features, float *y1, float *y2, float *y3,
                                   hipStream_t st) {
    const uint32_t pairs = channels / 2u, fit = 64u / pairs;
    if (fit >= 8u) return qat_cnn_front_launch<8, SAVE>(x, n, taps, channels, features, y1, y2, y3, st);
    if (fit >= 4u) return qat_cnn_front_launch<4, SAVE>(x, n, taps, channels, features, y1, y2, y3, st);
    if (fit >= 2u) return qat_cnn_front_launch<2, SAVE>(x, n, taps, channels, features, y1, y2, y3, st);
    return qat_cnn_front_launch<1, SAVE>(x, n, taps, channels, features, y1, y2, y3, st);
}

hipError_t bnmk_qat_cnn_front_forward(const float *x, uint64_t n, uint32_t channels, const float *const *w, const float *const *s,
                                      const int *quant_types, float *features, float *y1, float *y2, float *y3, void *workspace,
                                      hipStream_t st) {
    QatCnnPrepArgs a{};
    for (int l = 0; l < 3; l++) {
        a.w[l] = w[l];
        a.s[l] = s[l];
        a.qt[l] = quant_types[l];
    }
    float *taps = (float *)workspace;
    qat_cnn_prep_kernel<<<dim3(3), dim3(1024), 0, st>>>(a, channels, taps);
    if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
    if (n == 0) return hipSuccess;
    // the training form: all three planes or none (the C ABI checks)
    return y1 ? qat_cnn_front_go<true>(x, n, taps, channels, features, y1, y2, y3, st)
              : qat_cnn_front_go<false>(x, n, taps, channels, features, nullptr, nullptr, nullptr, st);
}
