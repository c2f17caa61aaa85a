// SURVEY.md §8(f) row 4, whole model: the forward pass of the reference's fully connected QAT network (models.py:56-90 FCMNIST, and
// the FC stack behind CNNMNIST's Flatten, :120-135) - per layer BitLinear.forward (BitNetMCU.py:214-235: Normalize ->
// activation_quant -> weight_quant -> F.linear) with ReLU between the layers - in ONE kernel per batch: float32 rows of 256 values
// in, float32 logits out, nothing in between goes through HBM (optionally the hidden activations, which are what a backward pass
// needs, are written once, as whole rows staged through LDS).  gfx950 only.  Floating point: parity with the reference module is within the tolerances
// tests/test_gpu_qat_model.py states, not bit-exact.
//
//   qat_model_prep_kernel     16 workgroups per layer, once per call: weight_quant of the layer's float weights (the level of every
//                             weight, BitNetMCU.py:150-177, x 2 for the half-integer types: an int8) written as the A-operand
//                             fragments of v_mfma_i32_32x32x32_i8, the reciprocal weight scales, optionally w_int / w_scale (the
//                             straight-through backward's operand); zeroes the work counter of the launch behind it
//   qat_fc_model_fwd_kernel   persistent waves, a wave owns a tile of 32 rows at a time:
//     * the tile's 32 KiB of floats land in VGPRs (one nontemporal global_load_dwordx4 per row: 1 KiB contiguous per instruction),
//       8 rows per group, NG groups in flight; per group a MERGED wave reduction gives every row's sum of squares (or of absolute
//       values: NormType 'Lin') and max|x| (v_permlane32_swap / v_permlane16_swap / DPP row_ror: ~45 VALU for eight rows);
//     * Normalize + activation_quant of layer 1 in registers: t = x * (1 / den), q = rne(t * scale) as the low byte of
//       (t * scale + 1.5 * 2^23), scale = 127 / max(max|x| / den, 1e-5); four bytes per lane and row go to the wave's int8 tile in
//       LDS (swizzled 16-byte slots) and come back as B operands: lane (j, h) holds bytes 16 h .. 16 h + 15 of K-step s of row j;
//     * every layer is Y^T[32 outputs x 32 rows] tiles on the int8 matrix cores with the weight fragments read from LDS; the
//       int32 sums are exact, y = sum / factor / x_scale / w_scale is applied as ONE multiplication per output (per-tensor
//       clipping scalar) or two (per-output);
//     * between layers nothing moves: the D fragment leaves each lane 16 of a tile's 32 outputs FOR ITS OWN ROW
//       (output (r & 3) + 8 (r >> 2) + 4 h of register r), so ReLU, the row's sum of squares and maximum are per-lane loops plus
//       one v_permlane32_swap, and the 16 quantised bytes ARE the next layer's B operand of K-step = this tile (the next
//       layer's fragments are written with their K columns in that order);
//     * the logits of a tile are 32 x n_classes consecutive floats: staged through the wave's LDS tile and written as 16-byte
//       lanes.
// What differs from the reference's arithmetic, all at the level of one float32 rounding: x / den is x * (1 / den); the sums of
// squares are added in another order; the integer GEMM is exact where F.linear rounds.  A value that lands within ~1e-5 of a
// rounding tie can therefore quantise one step away from the reference's.
// A row whose layer input is all zero has den = 0: the reference divides 0 / 0 and the row's logits are NaN; so are they here.
// NormType 'LayerNorm' (NORM 2): the row's mean first (one more merged reduction; hidden layers: one more pass over the lane's values,
// the last tile's padding columns masked out), then everything above on d = x - mean; its epsilon is absolute, so the row constant
// `a` enters the denominator (row_scalars_layernorm) - and keeps every row finite.  Compiled for one wave per SIMD.
#include "bnm_qat_math.hpp"
#include "bnm_quantise_f32.hpp"
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <set>
#include <utility>

namespace {

constexpr int QM_MAX_LAYERS = BNM_QAT_MODEL_MAX_LAYERS;
constexpr uint32_t QM_PREP_SPLIT = 16;  // workgroups per layer of the weight preparation (each reduces the tensor's statistics itself)
constexpr uint32_t QM_BATCH = 4;       // consecutive 32-row tiles per take from the work counter
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct QatModelDesc {
    uint32_t n_layers;                 // 2 .. 4
    uint32_t M[QM_MAX_LAYERS];         // 32-row output tiles per layer
    uint32_t KS[QM_MAX_LAYERS];        // K-steps of 32 per layer (8 for layer 0, else the previous layer's M)
    uint32_t width[QM_MAX_LAYERS];     // true output widths
    uint32_t frag_off[QM_MAX_LAYERS];  // byte offsets inside the weight image
    uint32_t winv_off[QM_MAX_LAYERS];  // reciprocal weight scales (floats, M x 32 per layer), byte offsets inside the image
    float inv_factor[QM_MAX_LAYERS];   // 1 / (int8 level factor)
    float inv_width[QM_MAX_LAYERS];    // 1 / (true output width): the mean of Normalize over the NEXT layer's inputs
    uint32_t image_bytes;              // multiple of 16
    uint32_t hidden_stride;            // floats per row of the hidden-activation output (sum of the hidden widths)
    uint32_t hidden_off[QM_MAX_LAYERS];
};

struct QatPrepArgs {
    const float *w[QM_MAX_LAYERS];
    const float *s[QM_MAX_LAYERS];
    float *w_deq[QM_MAX_LAYERS];
    uint32_t s_count[QM_MAX_LAYERS];
    int qt[QM_MAX_LAYERS];
    uint32_t d_in[QM_MAX_LAYERS];
    // fewer than 256 inputs (rows zero-padded to 256 floats by the caller): the model kernel's Normalize of layer 1 averages over 256
    // values, the reference over d - its denominator comes out sqrt(d / 256) ('Lin': d / 256) too small, the layer's outputs that much
    // too large: folded into layer 1's reciprocal weight scales here (everything behind is the kernel's as it stands)
    float in_fix;
};

// lanes < 32: a over {l, l + 32}; lanes >= 32: b
BNM_DEVICE float fold32_add(float a, float b) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float((uint32_t)r[0]) + __uint_as_float((uint32_t)r[1]);
}
BNM_DEVICE float fold16_add(float a, float b) {
    auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float((uint32_t)r[0]) + __uint_as_float((uint32_t)r[1]);
}
BNM_DEVICE float dpp_ror(float v, int ctrl8421) {
    switch (ctrl8421) {
        case 8: return __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0x128, 0xf, 0xf, false));
        case 4: return __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0x124, 0xf, 0xf, false));
        case 2: return __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0x122, 0xf, 0xf, false));
    }
    return __uint_as_float((uint32_t)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v), 0x121, 0xf, 0xf, false));
}
BNM_DEVICE float rowsum16(float v) {
    v += dpp_ror(v, 8);
    v += dpp_ror(v, 4);
    v += dpp_ror(v, 2);
    v += dpp_ror(v, 1);
    return v;
}
BNM_DEVICE uint32_t fold32_max(uint32_t a, uint32_t b) {
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    return umax((uint32_t)r[0], (uint32_t)r[1]);
}
BNM_DEVICE uint32_t fold16_max(uint32_t a, uint32_t b) {
    auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    return umax((uint32_t)r[0], (uint32_t)r[1]);
}
BNM_DEVICE uint32_t rowmax16u(uint32_t v) {
    v = umax(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, false));
    v = umax(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xf, 0xf, false));
    v = umax(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x122, 0xf, 0xf, false));
    v = umax(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x121, 0xf, 0xf, false));
    return v;
}
// lanes l and l ^ 32 combined, in both (v_permlane32_swap of v with itself: {(v.lo, v.lo), (v.hi, v.hi)})
BNM_DEVICE float halves_sum(float v) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float((uint32_t)r[0]) + __uint_as_float((uint32_t)r[1]);
}
BNM_DEVICE float halves_max(float v) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float((uint32_t)r[0]), __uint_as_float((uint32_t)r[1]));
}

// Per-row scalars of Normalize + activation_quant (BitNetMCU.py:237-246, :125-127) from the row's sum (of squares: NORM 0 'RMS'; of
// absolute values: 1 'Lin') and its max |x|:
//     x_norm = x / den,  scale = 127 / max|x_norm| = 127 den / max|x|,  x_int = rne(x_norm scale) = rne(x c) with c = 127 / max|x|
// (max|x_norm| >= 1 for both norms, so the reference's clamp at 1e-5 never acts on a row that is not all zero).  c does not depend
// on den; scale = c den.  v_rcp_f32 / v_sqrt_f32 (1 ulp) - see the header comment on what an ulp of a scale can do.  An all-zero
// row: c = inf, scale = inf * 0 = NaN, which reaches the row's logits as the reference's 0 / 0 does.
template <int NORM>
BNM_DEVICE void row_scalars(float sum, float mx, float inv_width, float &c, float &scale) {
    const float mean = sum * inv_width;
    const float den = NORM == 0 ? __builtin_amdgcn_sqrtf(mean) : mean;
    c = 127.0f * __builtin_amdgcn_rcpf(mx);
    scale = c * den;
}

// NormType 'LayerNorm' (BitNetMCU.py:253-257): x_norm = (x - mean) / sqrt(var + 1e-5).  Called with the sum of squares and the maximum of
// d = x - mean IN UNITS OF 1 / a (layer 1: a = 1; a hidden layer's outputs are y = a u): den = sqrt(a^2 sumsq / width + 1e-5) - the
// epsilon is absolute, so a does not cancel here -, scale = 127 / max(a max|d| / den, 1e-5) and x_int = rne(d c) with c = a scale / den.
// A constant row (d = 0) is not NaN: scale = 127 / 1e-5, every x_int 0.
BNM_DEVICE void row_scalars_layernorm(float sumsq, float mx, float inv_width, float a, float &c, float &scale) {
    const float den = __builtin_amdgcn_sqrtf(a * a * sumsq * inv_width + 1e-5f);
    const float inv_den = __builtin_amdgcn_rcpf(den);
    scale = 127.0f * __builtin_amdgcn_rcpf(fmaxf(a * mx * inv_den, 1e-5f));
    c = a * scale * inv_den;
}

// rne(v c) of four values as four int8 in one dword (byte b = value b): v c + 1.5 * 2^23 as two v_pk_fma_f32 (the sum's ulp is 1, so
// the fma rounds the exact product to the nearest integer, ties to even), the four low bytes gathered by v_perm_b32
// (magic = {1.5 * 2^23, 1.5 * 2^23} in a VGPR pair of the caller: as a scalar-register pair the compiler places its undefined high half on
// the register an outstanding work-counter take returns into - harmless, but profiles/check_inflight_sgprs.py rightly refuses to reason
// about that)
BNM_DEVICE uint32_t quantise4_pk(f32x2 lo, f32x2 hi, float c, f32x2 magic) {
    const f32x2 cc = {c, c};
    const f32x2 a = __builtin_elementwise_fma(lo, cc, magic), b = __builtin_elementwise_fma(hi, cc, magic);
    const uint32_t p = __builtin_amdgcn_perm(__float_as_uint(a[1]), __float_as_uint(a[0]), 0x0c0c0400u);
    const uint32_t q = __builtin_amdgcn_perm(__float_as_uint(b[1]), __float_as_uint(b[0]), 0x04000c0cu);
    return p | q;
}

}  // namespace

// ---- weight preparation: QM_PREP_SPLIT workgroups per layer ------------------------------------------------------------------------------
// Fragment order of layer l: [m][s][lane][16 bytes]; lane (i, h) of tile m, K-step s holds row 32 m + i and the 16 K columns
//   layer 0: 32 s + 16 h + r                       (the LDS tile's natural byte order)
//   layer l > 0: 32 s + (r & 3) + 8 (r >> 2) + 4 h    (the D-fragment order the previous layer's outputs are packed in)
__global__ __launch_bounds__(1024) void qat_model_prep_kernel(QatPrepArgs a, QatModelDesc d, char *__restrict__ image,
                                                              uint32_t *__restrict__ counter) {
    __shared__ double sa[16], sw[16];
    __shared__ float stats[2];
    const uint32_t l = blockIdx.x, part = blockIdx.y, tid = part * 1024u + threadIdx.x, stride = 1024u * QM_PREP_SPLIT;
    if (l == 0 && part == 0 && threadIdx.x < 144u) counter[threadIdx.x] = 0u;      // eight counter words, 64 bytes apart
    const float *w = a.w[l];
    const uint32_t k = d.width[l], din = a.d_in[l];
    const int qt = a.qt[l];
    const uint64_t count = (uint64_t)k * din;
    // mean |w| and mean w of the whole tensor (Ternary's scale, Binary's offset), fixed reduction order
    double x = 0.0, y = 0.0;
    if (qt == BNM_QAT_TERNARY || qt == BNM_QAT_BINARY)
        for (uint64_t i = threadIdx.x; i < count; i += 1024) {
            float v = w[i];
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
    const float mean_abs = stats[0], mean_w = stats[1];
    const float factor = (float)qat_i8_factor(qt);
    const uint32_t Mt = d.M[l], Ks = d.KS[l];
    int8_t *frag = (int8_t *)(image + d.frag_off[l]);
    for (uint32_t o = tid; o < Mt * Ks * 1024u; o += stride) {
        const uint32_t r = o & 15u, lane = (o >> 4) & 63u, ms = o >> 10, s = ms % Ks, m = ms / Ks;
        const uint32_t i = lane & 31u, h = lane >> 5;
        const uint32_t row = 32u * m + i;
        const uint32_t col = 32u * s + (l == 0 ? 16u * h + r : (r & 3u) + 8u * (r >> 2) + 4u * h);
        int8_t b = 0;
        if (row < k && col < din) {
            const float sc = qat_weight_scale(qt, a.s[l][a.s_count[l] > 1 ? row : 0], mean_abs);
            b = (int8_t)(int)(factor * qat_weight_level(qt, w[(uint64_t)row * din + col], sc, mean_w));      // exact: an integer in [-128, 127]
        }
        frag[o] = b;
    }
    float *winv = (float *)(image + d.winv_off[l]);
    const float fix = l == 0 ? a.in_fix : 1.0f;
    for (uint32_t row = tid; row < Mt * 32u; row += stride)
        winv[row] = row < k ? __fdiv_rn(1.0f, qat_weight_scale(qt, a.s[l][a.s_count[l] > 1 ? row : 0], mean_abs)) * fix : 0.0f;
    if (a.w_deq[l])
        for (uint64_t i = tid; i < count; i += stride) {
            const float sc = qat_weight_scale(qt, a.s[l][a.s_count[l] > 1 ? (uint32_t)(i / din) : 0], mean_abs);
            a.w_deq[l][i] = __fdiv_rn(qat_weight_level(qt, w[i], sc, mean_w), sc);       // w_int / w_scale, the STE forward value
        }
}

// ---- the model ----------------------------------------------------------------------------------------------------------------
// MH: most 32-row tiles of any layer.  NORM 0 RMS / 1 Lin / 2 LayerNorm.  PEROUT: per-output clipping scalars (a multiplication per output more).
// HID: the hidden activations are written.  NG: 8-row landing groups in flight per wave.  WPS: waves per SIMD the register budget is
// compiled for.
//
// The arithmetic per layer, in the units the kernel keeps.  The integer sums of a row are acc_o; the layer's outputs are
//     y_o = acc_o * a * [winv_o],     a = 1 / (factor x_scale [w_scale])  > 0        (winv_o = 1 / w_scale_o: PEROUT only)
// ReLU, Normalize and activation_quant commute with the positive row constant a, so with u_o = relu(acc_o [* winv_o]):
//     x_int(next) = rne(u_o * 127 / max u),     x_scale(next) = 127 sqrt(mean u^2) / max u       ('Lin': mean u)
// and a is needed only where y itself leaves the kernel (the logits, the optional hidden activations).
template <int MH, int NORM, bool PEROUT, bool HID, int NG, int WPS>
__global__ __launch_bounds__(256 * WPS) void qat_fc_model_fwd_kernel(const float *__restrict__ x, uint64_t n,
                                                                     const i32x4 *__restrict__ image, QatModelDesc d,
                                                                     float *__restrict__ logits, float *__restrict__ hidden,
                                                                     uint32_t n_classes, uint32_t *__restrict__ counter, uint32_t batch) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t nwaves = blockDim.x >> 6;
    for (uint32_t o = threadIdx.x * 16u; o < d.image_bytes; o += blockDim.x * 16u) *(i32x4 *)(smem + o) = image[o >> 4];
    __syncthreads();

    const uint32_t tile_off = d.image_bytes + wave * 8192u;       // this wave's int8 tile (32 rows x 256 bytes), later its logits stage
    const int j = lane & 31, h = lane >> 5;
    const uint32_t lane16 = 16u * (uint32_t)lane;
    const uint32_t rd_off = tile_off + (uint32_t)j * 256u + 16u * ((uint32_t)h ^ ((uint32_t)j & 15u));
    const uint32_t wr_off = tile_off + 4u * (uint32_t)lane;
    // which of a group's eight rows this lane's reduced statistics belong to (see the fold comment below); lanes with (lane & 7) == 0 report them
    const uint32_t my_row = (((uint32_t)lane >> 5) & 1u) | ((((uint32_t)lane >> 4) & 1u) << 1) | ((((uint32_t)lane >> 3) & 1u) << 2);
    float *const xs = (float *)(smem + d.image_bytes + nwaves * 8192u) + wave * 32u;      // layer-1 activation scales of the tile's rows

    const uint32_t n_units = (uint32_t)((n + 31ull) >> 5);
    const float *const xl = x + 4u * (uint32_t)lane;
    const uint32_t total_waves = gridDim.x * nwaves, wave_id = blockIdx.x * nwaves + wave;

    f32x2 magic = {12582912.0f, 12582912.0f};      // 1.5 * 2^23: the rounding add of quantise4_pk
    asm volatile("" : "+v"(magic));
    f32x4 land[NG][8];
    auto load_group = [&](uint32_t u, int g, f32x4(&dst)[8]) {
        const uint64_t first = (uint64_t)u * 32ull + (uint64_t)(8 * g);
        if (first + 8ull <= n) {
            const float *p = xl + first * 256ull;
#pragma unroll
            for (int r = 0; r < 8; r++) dst[r] = __builtin_nontemporal_load((const f32x4 *)(p + 256 * r));
        } else {
#pragma unroll
            for (int r = 0; r < 8; r++) {
                uint64_t row = first + (uint64_t)r;
                row = row < n ? row : n - 1ull;
                dst[r] = __builtin_nontemporal_load((const f32x4 *)(xl + row * 256ull));
            }
        }
    };

    // units in batches of `batch` consecutive ones (1 .. QM_BATCH: the launcher picks it so that a wave takes at least ~8 times - a call
    // of 10^6 rows is 15 tiles per wave, and whole batches of four leave a quarter of the waves idle at the end): a wave's first batch
    // is static, later ones come from a device-wide counter (eight words, wave w takes from word w mod 8; zeroed by the prep kernel).  The loop runs one unit ahead because `next`'s loads start
    // inside the current iteration; the take that decides next's successor is issued at the top and retired behind the quantisation.
    const uint32_t my_word = wave_id & 7u, first_dyn = (total_waves + 7u) >> 3;
    uint32_t taken = 0;
    auto batch_first = [&](uint32_t t) { return (((first_dyn + t) << 3) + my_word) * batch; };
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
    if (unit < n_units) static_for<0, NG>([&](auto GI) { load_group(unit, decltype(GI)::value, land[decltype(GI)::value]); });

    while (unit < n_units) {
        const bool take = next_left == 0u;
        if (take) work_take_issue(taken, counter + 16u * my_word, 1u);
        // ---- layer 1's Normalize + activation_quant: four groups of 8 rows, registers -> int8 rows of the LDS tile ------------
        static_for<0, 4>([&](auto GI) {
            constexpr int g = decltype(GI)::value, slot = g % NG;
            constexpr int kLane[8] = {0, 32, 16, 48, 8, 40, 24, 56};
            float sum[8];
            uint32_t mx[8];
            if constexpr (NORM == 2) {
                // LayerNorm: the rows' means first (the same merged reduction on plain sums), subtracted in place: land holds d = x - mean
#pragma unroll
                for (int r = 0; r < 8; r++) {
                    const f32x4 &v = land[slot][r];
                    sum[r] = (v[0] + v[1]) + (v[2] + v[3]);
                }
                const float t0 = fold32_add(sum[0], sum[1]), t1 = fold32_add(sum[2], sum[3]), t2 = fold32_add(sum[4], sum[5]), t3 = fold32_add(sum[6], sum[7]);
                const float g0 = rowsum16(fold16_add(t0, t1)), g1 = rowsum16(fold16_add(t2, t3));
                const float mean = ((lane & 8) ? g1 : g0) * (1.0f / 256.0f);
#pragma unroll
                for (int r = 0; r < 8; r++) {
                    const float mr = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mean), kLane[r]));
                    f32x4 &v = land[slot][r];
                    v[0] -= mr; v[1] -= mr; v[2] -= mr; v[3] -= mr;
                }
            }
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const f32x4 &v = land[slot][r];
                if constexpr (NORM == 0 || NORM == 2) {
                    const f32x2 lo = {v[0], v[1]}, hi = {v[2], v[3]};
                    const f32x2 sq = __builtin_elementwise_fma(hi, hi, lo * lo);
                    sum[r] = sq[0] + sq[1];
                } else {
                    sum[r] = (fabsf(v[0]) + fabsf(v[1])) + (fabsf(v[2]) + fabsf(v[3]));
                }
                mx[r] = absmax4_bits(v);
            }
            // fold32 pairs rows (0,1) (2,3) (4,5) (6,7); fold16 pairs the results; the select puts the second quadruple into the upper
            // eight lanes of every 16-lane row: lane L holds row ((L>>5)&1) | ((L>>4)&1)<<1 | ((L>>3)&1)<<2
            const float s0 = fold32_add(sum[0], sum[1]), s1 = fold32_add(sum[2], sum[3]), s2 = fold32_add(sum[4], sum[5]), s3 = fold32_add(sum[6], sum[7]);
            const float e0 = rowsum16(fold16_add(s0, s1)), e1 = rowsum16(fold16_add(s2, s3));
            const uint32_t c0 = fold32_max(mx[0], mx[1]), c1 = fold32_max(mx[2], mx[3]), c2 = fold32_max(mx[4], mx[5]), c3 = fold32_max(mx[6], mx[7]);
            const uint32_t f0 = rowmax16u(fold16_max(c0, c1)), f1 = rowmax16u(fold16_max(c2, c3));
            const float row_sum = (lane & 8) ? e1 : e0;
            const float row_max = __uint_as_float((lane & 8) ? f1 : f0);
            float cq, scale;
            if constexpr (NORM == 2) row_scalars_layernorm(row_sum, row_max, 1.0f / 256.0f, 1.0f, cq, scale);
            else row_scalars<NORM>(row_sum, row_max, 1.0f / 256.0f, cq, scale);
            if ((lane & 7) == 0) xs[8 * g + (int)my_row] = scale;
            uint32_t q[8];
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const float c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cq), kLane[r]));
                const f32x4 &v = land[slot][r];
                q[r] = quantise4_pk(f32x2{v[0], v[1]}, f32x2{v[2], v[3]}, c, magic);
            }
            if constexpr (g + NG < 4) load_group(unit, g + NG, land[slot]);
            else if (next < n_units) load_group(next, g + NG - 4, land[slot]);
#pragma unroll
            for (int r = 0; r < 8; r++) {
                constexpr int R0 = 8 * g;
                *(uint32_t *)(smem + ((wr_off ^ (16u * (uint32_t)((R0 + r) & 15))) + 256u * (uint32_t)(R0 + r))) = q[r];
            }
        });
        uint32_t nn = next + 1u, nn_left = next_left - 1u;
        if (take) {
            work_take_wait(taken);
            nn = batch_first(taken);
            nn_left = batch - 1u;
        }

        // ---- the layers -----------------------------------------------------------------------------------------------------
        float u[MH][16];          // relu(acc [* winv]) of the layer in hand
        i32x4 act[MH];            // its quantised form: the next layer's B operands
        float x_scale = xs[j];
        // a of layer l (see the comment above the kernel)
        auto out_scale = [&](uint32_t l) {
            float a = d.inv_factor[l] * __builtin_amdgcn_rcpf(x_scale);
            if constexpr (!PEROUT) a *= *(const float *)(smem + d.winv_off[l]);
            return a;
        };
        // the D fragment of tile m of layer l as floats (PEROUT: times the outputs' reciprocal weight scales)
        auto to_float = [&](uint32_t l, int m, const i32x16 &acc, float(&out)[16]) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                f32x4 wi = {1.0f, 1.0f, 1.0f, 1.0f};
                if constexpr (PEROUT) wi = *(const f32x4 *)(smem + d.winv_off[l] + 4u * (32u * (uint32_t)m + 8u * (uint32_t)q + 4u * (uint32_t)h));
#pragma unroll
                for (int b = 0; b < 4; b++) out[4 * q + b] = PEROUT ? (float)acc[4 * q + b] * wi[b] : (float)acc[4 * q + b];
            }
        };
        // ReLU + Normalize + activation_quant of layer l's outputs (held in u as acc [* winv]) -> the next layer's B operands; `a` only
        // where the activations are written out
        auto relu_norm_quant = [&](uint32_t l, float a) {
            f32x2 sum2 = {0.0f, 0.0f};
            float mx = 0.0f;
#pragma unroll
            for (int m = 0; m < MH; m++)
                if ((uint32_t)m < d.M[l]) {
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const f32x2 v = {fmaxf(u[m][r], 0.0f), fmaxf(u[m][r + 1], 0.0f)};
                        u[m][r] = v[0];
                        u[m][r + 1] = v[1];
                        if constexpr (NORM == 0) sum2 = __builtin_elementwise_fma(v, v, sum2);
                        else sum2 += v;
                        mx = __builtin_fmaxf(__builtin_fmaxf(mx, v[0]), v[1]);      // (v_max3_f32)
                    }
                }
            if constexpr (HID) {
                // the layer's outputs y = u a leave as whole rows: two tiles (64 outputs: 256 bytes per row) at a time through the wave's
                // LDS tile - free here: its int8 rows were consumed by layer 0's operand reads - in the int8 tile's own geometry
                // (32 rows x sixteen 16-byte slots, slot s of row r stored at s ^ (r & 15): conflict-free both ways), then 16 lanes
                // write a row's 256 consecutive bytes, four rows per instruction
                const uint64_t first_row = (uint64_t)unit * 32ull;
                const uint32_t width = d.width[l], hoff = d.hidden_off[l], hstride = d.hidden_stride;
                const bool aligned = ((hoff | hstride) & 3u) == 0u && (((uintptr_t)hidden) & 15u) == 0u;
#pragma unroll
                for (int mp = 0; mp < (MH + 1) / 2; mp++)
                    if ((uint32_t)(2 * mp) < d.M[l]) {
#pragma unroll
                        for (int mm = 0; mm < 2; mm++) {
                            const int m = 2 * mp + mm;
                            if (m < MH && (uint32_t)m < d.M[l]) {
#pragma unroll
                                for (int q = 0; q < 4; q++) {
                                    const uint32_t slot = 8u * (uint32_t)mm + 2u * (uint32_t)q + (uint32_t)h;
                                    const f32x4 v = {u[m][4 * q] * a, u[m][4 * q + 1] * a, u[m][4 * q + 2] * a, u[m][4 * q + 3] * a};
                                    *(f32x4 *)(smem + tile_off + 256u * (uint32_t)j + 16u * (slot ^ ((uint32_t)j & 15u))) = v;
                                }
                            }
                        }
                        const uint32_t col = 64u * (uint32_t)mp + 4u * ((uint32_t)lane & 15u);      // this lane's four outputs of the layer
#pragma unroll
                        for (int it = 0; it < 8; it++) {
                            const uint32_t r = 4u * (uint32_t)it + ((uint32_t)lane >> 4);
                            const f32x4 v = *(const f32x4 *)(smem + tile_off + 256u * r + 16u * (((uint32_t)lane & 15u) ^ (r & 15u)));
                            if (first_row + r < n && col < width) {
                                float *hp = hidden + (first_row + r) * hstride + hoff + col;
                                if (aligned && col + 4u <= width) {
                                    *(f32x4 *)hp = v;
                                } else {
#pragma unroll
                                    for (int e = 0; e < 4; e++)
                                        if (col + (uint32_t)e < width) hp[e] = v[e];
                                }
                            }
                        }
                    }
            }
            float sum = halves_sum(sum2[0] + sum2[1]);
            float cq;
            if constexpr (NORM == 2) {
                // LayerNorm: d = u - mean over the layer's REAL outputs (the padding columns of the last tile hold u = 0, not d = 0:
                // masked out), then the sum of squares and the maximum of d
                const float mean = sum * d.inv_width[l];
                const uint32_t width = d.width[l];
                f32x2 sq2 = {0.0f, 0.0f};
                mx = 0.0f;
#pragma unroll
                for (int m = 0; m < MH; m++)
                    if ((uint32_t)m < d.M[l]) {
#pragma unroll
                        for (int r = 0; r < 16; r += 2) {
                            const uint32_t col = 32u * (uint32_t)m + (uint32_t)(r & 3) + 8u * (uint32_t)(r >> 2) + 4u * (uint32_t)h;      // (r even: col + 1 is r + 1's)
                            const f32x2 v = {col < width ? u[m][r] - mean : 0.0f, col + 1u < width ? u[m][r + 1] - mean : 0.0f};
                            u[m][r] = v[0];
                            u[m][r + 1] = v[1];
                            sq2 = __builtin_elementwise_fma(v, v, sq2);
                            mx = __builtin_fmaxf(__builtin_fmaxf(mx, __builtin_fabsf(v[0])), __builtin_fabsf(v[1]));
                        }
                    }
                sum = halves_sum(sq2[0] + sq2[1]);
                mx = halves_max(mx);
                row_scalars_layernorm(sum, mx, d.inv_width[l], a, cq, x_scale);
            } else {
                mx = halves_max(mx);
                row_scalars<NORM>(sum, mx, d.inv_width[l], cq, x_scale);
            }
#pragma unroll
            for (int m = 0; m < MH; m++)
                if ((uint32_t)m < d.M[l]) {
#pragma unroll
                    for (int q = 0; q < 4; q++)
                        act[m][q] = (int)quantise4_pk(f32x2{u[m][4 * q], u[m][4 * q + 1]}, f32x2{u[m][4 * q + 2], u[m][4 * q + 3]}, cq, magic);
                }
        };
        // layer 0: B operands from the LDS tile (MH = 4: re-read per output tile - 64 registers of outputs leave no room to hold them)
        {
            i32x4 b0[MH <= 2 ? 8 : 1];
            if constexpr (MH <= 2) {
#pragma unroll
                for (int s = 0; s < 8; s++) b0[s] = *(const i32x4 *)(smem + (rd_off ^ (32u * (uint32_t)s)));
            }
#pragma unroll
            for (int m = 0; m < MH; m++)
                if ((uint32_t)m < d.M[0]) {
                    i32x16 acc = {0};
                    const char *fp = smem + d.frag_off[0] + (uint32_t)m * 8u * 1024u + lane16;
#pragma unroll
                    for (int s = 0; s < 8; s++) {
                        if constexpr (MH <= 2) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(*(const i32x4 *)(fp + 1024 * s), b0[s], acc, 0, 0, 0);
                        else acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(*(const i32x4 *)(fp + 1024 * s), *(const i32x4 *)(smem + (rd_off ^ (32u * (uint32_t)s))), acc, 0, 0, 0);
                    }
                    to_float(0, m, acc, u[m]);
                }
        }
        auto layer = [&](uint32_t l) {
            const uint32_t ks = d.KS[l];
#pragma unroll
            for (int m = 0; m < MH; m++)
                if ((uint32_t)m < d.M[l]) {
                    i32x16 acc = {0};
                    const char *fp = smem + d.frag_off[l] + (uint32_t)m * ks * 1024u + lane16;
#pragma unroll
                    for (int s = 0; s < MH; s++)
                        if ((uint32_t)s < ks) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(*(const i32x4 *)(fp + 1024 * s), act[s], acc, 0, 0, 0);
                    to_float(l, m, acc, u[m]);
                }
        };
        relu_norm_quant(0, (HID || NORM == 2) ? out_scale(0) : 0.0f);
        for (uint32_t l = 1; l + 1u < d.n_layers; l++) {
            const float a = (HID || NORM == 2) ? out_scale(l) : 0.0f;      // (x_scale is the layer's INPUT scale here)
            layer(l);
            relu_norm_quant(l, a);
        }
        const float a_last = out_scale(d.n_layers - 1u);
        layer(d.n_layers - 1u);
        // ---- logits: 32 x n_classes consecutive floats, staged through the wave's tile ------------------------------------------
        {
            float *stage = (float *)(smem + tile_off);
#pragma unroll
            for (int m = 0; m < (MH < 2 ? MH : 2); m++)
                if ((uint32_t)m < d.M[d.n_layers - 1u]) {
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const uint32_t c = 32u * (uint32_t)m + (uint32_t)(r & 3) + 8u * (uint32_t)(r >> 2) + 4u * (uint32_t)h;
                        if (c < n_classes) stage[(uint32_t)j * n_classes + c] = u[m][r] * a_last;
                    }
                }
            const uint64_t first = (uint64_t)unit * 32ull;
            const uint32_t rows = (uint32_t)(n - first < 32ull ? n - first : 32ull), count = rows * n_classes;
            float *out = logits + first * n_classes;
            // (first * n_classes * 4 bytes is a multiple of 128: whole 16-byte lanes, then the tail)
            for (uint32_t o = 4u * (uint32_t)lane; o + 4u <= count; o += 256u)
                __builtin_nontemporal_store(*(const f32x4 *)(stage + o), (f32x4 *)(out + o));
            if ((uint32_t)lane < (count & 3u)) out[(count & ~3u) + (uint32_t)lane] = stage[(count & ~3u) + (uint32_t)lane];
        }
        unit = next;
        next = nn;
        next_left = nn_left;
    }
}

// ---- host side ----------------------------------------------------------------------------------------------------------------
namespace {

struct QatModelPlan {
    QatModelDesc d;
    uint32_t mh;
    size_t workspace_bytes;     // image + counter block
};

bool qat_model_plan(uint32_t n_layers, const uint32_t *widths, QatModelPlan &p) {
    if (n_layers < 2 || n_layers > (uint32_t)QM_MAX_LAYERS || widths[0] == 0u || widths[0] > 256u) return false;
    QatModelDesc &d = p.d;
    d = QatModelDesc{};
    d.n_layers = n_layers;
    uint32_t off = 0, mh = 1, hoff = 0;
    for (uint32_t l = 0; l < n_layers; l++) {
        const uint32_t w = widths[l + 1];
        if (w == 0 || w > 192u) return false;
        if (l + 1 == n_layers && w > 64u) return false;
        d.width[l] = w;
        d.inv_width[l] = 1.0f / (float)w;
        d.M[l] = (w + 31u) / 32u;
        d.KS[l] = l == 0 ? 8u : d.M[l - 1];
        d.frag_off[l] = off;
        off += d.M[l] * d.KS[l] * 1024u;
        mh = d.M[l] > mh ? d.M[l] : mh;
        if (l + 1 < n_layers) {
            d.hidden_off[l] = hoff;
            hoff += w;
        }
    }
    for (uint32_t l = 0; l < n_layers; l++) {
        d.winv_off[l] = off;
        off += d.M[l] * 32u * 4u;
    }
    d.image_bytes = (off + 1023u) & ~1023u;      // the waves' tiles behind it are addressed with XOR swizzles: 1 KiB aligned
    d.hidden_stride = hoff;
    p.mh = mh <= 2 ? 2 : mh <= 4 ? 4 : 6;
    p.workspace_bytes = (size_t)d.image_bytes + 1024u;      // + the work counter block
    return true;
}

// dynamic LDS beyond 64 KiB must be allowed per kernel and device, once (any host thread may be the first)
hipError_t qat_allow_big_lds(const void *fn) {
    static std::mutex mu;
    static std::set<std::pair<const void *, int>> done;
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
    std::lock_guard<std::mutex> g(mu);
    if (done.count({fn, dev})) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) done.insert({fn, dev});
    return e;
}

template <int MH, int NORM, bool PEROUT, bool HID, int NG, int WPS>
hipError_t qat_model_launch_as(const QatModelDesc &d, const float *x, uint64_t n, const char *image, float *logits, float *hidden,
                               uint32_t n_classes, uint32_t *counter, hipStream_t st) {
    auto fn = qat_fc_model_fwd_kernel<MH, NORM, PEROUT, HID, NG, WPS>;
    // (the widest stacks of the six-tile class - four layers of ~192 - leave room for two waves' tiles beside the weight image, not four)
    unsigned threads = 256 * WPS, nwaves = threads / 64;
    size_t lds = (size_t)d.image_bytes + (size_t)nwaves * 8192u + (size_t)nwaves * 128u;
    if (lds > 160u * 1024u && WPS == 1) {
        threads = 128;
        nwaves = 2;
        lds = (size_t)d.image_bytes + (size_t)nwaves * 8192u + (size_t)nwaves * 128u;
    }
    if (lds > 160u * 1024u) return hipErrorInvalidValue;
    if (hipError_t e = qat_allow_big_lds((const void *)fn); e != hipSuccess) return e;
    const uint64_t units = (n + 31ull) >> 5;
    if (units >= (1ull << 31)) return hipErrorInvalidValue;      // 32-bit tile indices in the kernel
    // resident workgroups per CU: one of eight waves, or (one wave per SIMD) two of four where the LDS holds two
    const uint64_t per_cu = (WPS == 1 && 2u * lds <= 160u * 1024u) ? 2u : 1u;
    const uint64_t cap = (uint64_t)bnm_num_cus() * per_cu;
    uint64_t batch = units / (cap * nwaves * 8u);      // tiles per take: at least ~8 takes per wave
    batch = batch < 1u ? 1u : batch > QM_BATCH ? QM_BATCH : batch;
    uint64_t blocks = (units + (uint64_t)nwaves * batch - 1) / ((uint64_t)nwaves * batch);
    if (blocks > cap) blocks = cap & ~(uint64_t)(8u / nwaves - 1u);      // capped grids: waves a multiple of 8 (the counter's eight words; nwaves 2, 4 or 8)
    fn<<<dim3((unsigned)blocks), dim3(threads), lds, st>>>(x, n, (const i32x4 *)image, d, logits, hidden, n_classes, counter, (uint32_t)batch);
    return hipGetLastError();
}

template <int MH, int NORM, bool PEROUT, bool HID>
hipError_t qat_model_launch(const QatModelDesc &d, const float *x, uint64_t n, const char *image, float *logits, float *hidden,
                            uint32_t n_classes, uint32_t *counter, hipStream_t st) {
#ifdef BNM_QAT_MODEL_TUNE
    // measurement builds only: BNM_QAT_TUNE=<ng><wps> picks another landing depth / occupancy for the flagship instantiation
    if constexpr (MH == 2 && NORM == 0 && !PEROUT && !HID) {
        const char *t = getenv("BNM_QAT_TUNE");
        if (t && !strcmp(t, "42")) return qat_model_launch_as<MH, NORM, PEROUT, HID, 4, 2>(d, x, n, image, logits, hidden, n_classes, counter, st);
        if (t && !strcmp(t, "41")) return qat_model_launch_as<MH, NORM, PEROUT, HID, 4, 1>(d, x, n, image, logits, hidden, n_classes, counter, st);
        if (t && !strcmp(t, "21")) return qat_model_launch_as<MH, NORM, PEROUT, HID, 2, 1>(d, x, n, image, logits, hidden, n_classes, counter, st);
    }
#endif
    // (the four-tile class with the hidden activations written, LayerNorm's second pass over the values, and the six-tile class
    // need more than 256 registers: one wave per SIMD there)
    return qat_model_launch_as<MH, NORM, PEROUT, HID, 2, ((MH == 4 && HID) || NORM == 2 || MH == 6) ? 1 : 2>(d, x, n, image, logits, hidden, n_classes, counter, st);
}

}  // namespace

size_t bnmk_qat_model_workspace_bytes(uint32_t n_layers, const uint32_t *widths) {
    QatModelPlan p;
    return qat_model_plan(n_layers, widths, p) ? p.workspace_bytes : 0;
}

bool bnmk_qat_model_supported(uint32_t n_layers, const uint32_t *widths, const int *quant_types, int norm_type) {
    QatModelPlan p;
    if (!qat_model_plan(n_layers, widths, p)) return false;
    for (uint32_t l = 0; l < n_layers; l++)
        if (!qat_i8_factor(quant_types[l])) return false;
    // (LayerNorm subtracts the row's mean from the padding columns as well: 256 inputs only)
    return norm_type == BNM_QAT_NORM_RMS || norm_type == BNM_QAT_NORM_LIN || (norm_type == BNM_QAT_NORM_LAYERNORM && widths[0] == 256u);
}

hipError_t bnmk_qat_model_forward(const float *x, uint64_t n, uint32_t n_layers, const uint32_t *widths, const float *const *w,
                                  const float *const *s, const uint32_t *s_count, const int *quant_types, int norm_type,
                                  float *logits, float *hidden, float *const *w_deq, void *workspace, hipStream_t st) {
    QatModelPlan p;
    if (!qat_model_plan(n_layers, widths, p)) return hipErrorInvalidValue;
    QatPrepArgs a{};
    bool perout = false;
    for (uint32_t l = 0; l < n_layers; l++) {
        a.w[l] = w[l];
        a.s[l] = s[l];
        a.s_count[l] = s_count[l];
        a.qt[l] = quant_types[l];
        a.d_in[l] = widths[l];
        a.w_deq[l] = w_deq ? w_deq[l] : nullptr;
        p.d.inv_factor[l] = 1.0f / (float)qat_i8_factor(quant_types[l]);
        perout = perout || s_count[l] > 1;
    }
    a.in_fix = widths[0] == 256u ? 1.0f : norm_type == BNM_QAT_NORM_LIN ? (float)widths[0] / 256.0f : sqrtf((float)widths[0] / 256.0f);
    char *image = (char *)workspace;
    uint32_t *counter = (uint32_t *)(image + p.d.image_bytes);
    qat_model_prep_kernel<<<dim3(n_layers, QM_PREP_SPLIT), dim3(1024), 0, st>>>(a, p.d, image, counter);
    if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
    if (n == 0) return hipSuccess;
    const uint32_t nc = widths[n_layers];
    const int nt = norm_type == BNM_QAT_NORM_RMS ? 0 : norm_type == BNM_QAT_NORM_LIN ? 1 : 2;
#define QM_GO(MH, NORM, PO)                                                                                   \
    return hidden ? qat_model_launch<MH, NORM, PO, true>(p.d, x, n, image, logits, hidden, nc, counter, st) \
                  : qat_model_launch<MH, NORM, PO, false>(p.d, x, n, image, logits, hidden, nc, counter, st)
    if (p.mh == 2) {
        if (nt == 0) { if (perout) QM_GO(2, 0, true); else QM_GO(2, 0, false); }
        else if (nt == 1) { if (perout) QM_GO(2, 1, true); else QM_GO(2, 1, false); }
        else { if (perout) QM_GO(2, 2, true); else QM_GO(2, 2, false); }
    } else if (p.mh == 4) {
        if (nt == 0) { if (perout) QM_GO(4, 0, true); else QM_GO(4, 0, false); }
        else if (nt == 1) { if (perout) QM_GO(4, 1, true); else QM_GO(4, 1, false); }
        else { if (perout) QM_GO(4, 2, true); else QM_GO(4, 2, false); }
    } else {
        if (nt == 0) { if (perout) QM_GO(6, 0, true); else QM_GO(6, 0, false); }
        else if (nt == 1) { if (perout) QM_GO(6, 1, true); else QM_GO(6, 1, false); }
        else { if (perout) QM_GO(6, 2, true); else QM_GO(6, 2, false); }
    }
#undef QM_GO
}
