// the forward pass of the reference's quantisation-aware-training layer BitLinear
// (BitNetMCU.py:198-235: Normalize -> activation_quant -> weight_quant -> F.linear) as fused gfx950 kernels.
// Floating point; parity is against PyTorch fp32 within a stated tolerance (tests/test_gpu_qat.py), not bit-exact.
//
// Formulation.  The reference multiplies x_int / x_scale by w_int / w_scale in fp32.  x_int are integers in
// [-128, 127] and the weight levels w_int are small integers or half-integers for every QuantType except '4bit'
// (+0.01) and 'NF4', so the products and their fp32 sums over d <= 1024 are EXACT on the fp32 matrix cores
// (|sum| < 2^24); the two scales are applied once per output.  The result is therefore at least as accurate as
// the reference's own fp32 GEMM, and differs from it only by that GEMM's rounding.
//
//   qat_weight_stats_kernel   mean|w|, mean(w) of the whole tensor (Ternary's scale, Binary's offset)
//   qat_weight_quant_kernel   w [k][d] -> levels u, stored transposed uT [dpad][kpad] (coalesced A operands),
//                             and the per-output scale w_scale [kpad]
//   qat_batch_stats_kernel    per-feature mean and sqrt(var + 1e-5) over the batch (NormType BatchNorm only)
//   qat_bitlinear_fwd_kernel  32 rows per workgroup: normalise + quantise each row into an LDS tile (one
//                             wavefront per row), then Y^T[32 outs x 32 rows] tiles on the matrix cores:
//                             v_mfma_i32_32x32x32_i8 when the weight levels (x2 for half-integer types) fit int8 - Binary,
//                             BinarySym, Ternary, 2bitsym, 4bitsym, 5bitsym, 8bit: the integer sums are the same exact values
//                             the fp32 path produces, 16 K-steps per instruction instead of 1 - else v_mfma_f32_32x32x2_f32
#include "bnm_qat_math.hpp"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int QAT_ROWS = 32;    // rows of x per workgroup (= MFMA N)
constexpr int QAT_WAVES = 4;

BNM_DEVICE float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
BNM_DEVICE float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    return v;
}

}  // namespace

// stats[0] = mean|w|, stats[1] = mean(w); one workgroup, fixed reduction order (deterministic)
__global__ __launch_bounds__(1024) void qat_weight_stats_kernel(const float *__restrict__ w, uint64_t count,
                                                                float *__restrict__ stats) {
    __shared__ double sa[16], sw[16];
    double a = 0.0, b = 0.0;
    for (uint64_t i = threadIdx.x; i < count; i += 1024) {
        float v = w[i];
        a += fabsf(v);
        b += v;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        a += __shfl_xor(a, off);
        b += __shfl_xor(b, off);
    }
    if ((threadIdx.x & 63) == 0) { sa[threadIdx.x >> 6] = a; sw[threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double ta = 0.0, tb = 0.0;
        for (int i = 0; i < 16; i++) { ta += sa[i]; tb += sw[i]; }
        stats[0] = (float)(ta / (double)count);
        stats[1] = (float)(tb / (double)count);
    }
}

// one thread per weight; uT and w_scale are zero-filled by the launcher first (padding rows/columns stay 0)
__global__ __launch_bounds__(256) void qat_weight_quant_kernel(const float *__restrict__ w, uint32_t k, uint32_t d,
                                                               const float *__restrict__ s, uint32_t s_count, int qt,
                                                               const float *__restrict__ stats, uint32_t kpad,
                                                               float *__restrict__ uT, float *__restrict__ w_scale,
                                                               float *__restrict__ w_deq_out, int8_t *__restrict__ u8,
                                                               uint32_t d8) {
    const uint64_t idx = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (idx >= (uint64_t)k * d) return;
    const uint32_t row = (uint32_t)(idx / d), col = (uint32_t)(idx % d);
    const float sc = qat_weight_scale(qt, s[s_count > 1 ? row : 0], stats[0]);
    const float u = qat_weight_level(qt, w[idx], sc, stats[1]);
    uT[(uint64_t)col * kpad + row] = u;
    if (u8) u8[(uint64_t)row * d8 + col] = (int8_t)(int)((float)qat_i8_factor(qt) * u);      // exact: an integer in [-128, 127]
    if (w_deq_out) w_deq_out[idx] = qt == BNM_QAT_NONE ? u : __fdiv_rn(u, sc);   // w_int / w_scale, the STE forward value
    if (col == 0) w_scale[row] = sc;
}

// BatchNorm statistics over the batch dimension (BitNetMCU.py:247-250): one thread per feature, rows are walked in
// order so a wavefront reads 256 contiguous bytes per row
__global__ __launch_bounds__(256) void qat_batch_stats_kernel(const float *__restrict__ x, uint64_t n, uint32_t d,
                                                              float *__restrict__ mean, float *__restrict__ den) {
    const uint32_t c = blockIdx.x * 256u + threadIdx.x;
    if (c >= d) return;
    double sum = 0.0;
    for (uint64_t r = 0; r < n; r++) sum += x[r * d + c];
    const float m = (float)(sum / (double)n);
    double var = 0.0;
    for (uint64_t r = 0; r < n; r++) {
        double t = (double)x[r * d + c] - (double)m;
        var += t * t;
    }
    mean[c] = m;
    den[c] = sqrtf((float)(var / (double)n) + 1e-5f);
}

// dynamic LDS: q tile [32][dpad + 1] floats (odd row stride: conflict-free column reads) + 32 row scales
__global__ __launch_bounds__(64 * QAT_WAVES) void qat_bitlinear_fwd_kernel(
    const float *__restrict__ x, uint64_t n, uint32_t d, uint32_t dpad, const float *__restrict__ uT,
    const float *__restrict__ w_scale, uint32_t k, uint32_t kpad, int qt, int nt, const float *__restrict__ bn_mean,
    const float *__restrict__ bn_den, float *__restrict__ y, float *__restrict__ x_int_out,
    float *__restrict__ x_scale_out, const int8_t *__restrict__ u8, uint32_t d8) {
    extern __shared__ float lds[];
    const uint32_t rs = dpad + 1u;
    float *q = lds;
    float *xs = lds + QAT_ROWS * rs;
    // int8 path: the quantised activations once more as bytes, rows of d8 + 16 (K padded to 32 with zeros; +16: bank spread)
    const uint32_t rs8 = d8 + 16u;
    int8_t *q8 = (int8_t *)(xs + QAT_ROWS);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t row0 = (uint64_t)blockIdx.x * QAT_ROWS;

    // ---- phase 1: Normalize (:237-262) + activation_quant (:119-128), one wavefront per row ------------------
    for (int r = wave; r < QAT_ROWS; r += QAT_WAVES) {
        float *qr = q + r * rs;
        const uint64_t row = row0 + (uint64_t)r;
        if (u8)
            for (uint32_t c = lane; c < rs8 / 4u; c += 64) ((int *)(q8 + r * rs8))[c] = 0;        // zero K padding (and dead rows)
        if (row >= n) {
            for (uint32_t c = lane; c < rs; c += 64) qr[c] = 0.0f;
            if (lane == 0) xs[r] = 1.0f;
            continue;
        }
        const float *xr = x + row * d;
        float a = 0.0f, b = 0.0f;
        for (uint32_t c = lane; c < d; c += 64) {
            float v = xr[c];
            qr[c] = v;
            a += nt == BNM_QAT_NORM_RMS ? v * v : (nt == BNM_QAT_NORM_LIN ? fabsf(v) : v);
        }
        if (lane == 0 && (d & 1u)) qr[d] = 0.0f;       // K padding to an even length
        a = wave_sum(a);
        float sub = 0.0f, den = 1.0f;
        if (nt == BNM_QAT_NORM_RMS) den = sqrtf(a / (float)d);
        else if (nt == BNM_QAT_NORM_LIN) den = a / (float)d;
        else if (nt == BNM_QAT_NORM_LAYERNORM) {
            sub = a / (float)d;
            for (uint32_t c = lane; c < d; c += 64) {
                float t = qr[c] - sub;
                b += t * t;
            }
            den = sqrtf(wave_sum(b) / (float)d + 1e-5f);
        }
        float mx = 0.0f;
        for (uint32_t c = lane; c < d; c += 64) {
            float v = qr[c];
            if (nt == BNM_QAT_NORM_BATCHNORM) v = __fdiv_rn(__fsub_rn(v, bn_mean[c]), bn_den[c]);
            else if (nt == BNM_QAT_NORM_LAYERNORM) v = __fdiv_rn(__fsub_rn(v, sub), den);
            else if (nt != BNM_QAT_NORM_NONE) v = __fdiv_rn(v, den);
            qr[c] = v;
            mx = fmaxf(mx, fabsf(v));
        }
        if (qt == BNM_QAT_NONE) {
            if (lane == 0) xs[r] = 1.0f;
            continue;
        }
        mx = wave_max(mx);
        const float sc = __fdiv_rn(127.0f, fmaxf(mx, 1e-5f));
        for (uint32_t c = lane; c < d; c += 64) {
            float v = fminf(fmaxf(rintf(__fmul_rn(qr[c], sc)), -128.0f), 127.0f);
            qr[c] = v;
            if (u8) q8[r * rs8 + c] = (int8_t)(int)v;
            if (x_int_out) x_int_out[row * d + c] = v;
        }
        if (lane == 0) {
            xs[r] = sc;
            if (x_scale_out) x_scale_out[row] = sc;
        }
    }
    __syncthreads();

    // ---- phase 2: Y^T tile = U[32 outs x K] * Q^T[K x 32 rows] on the fp32 matrix cores ----------------------
    const int i = lane & 31, h = lane >> 5;
    const uint32_t mtiles = kpad / 32u;
    const float inv_f = u8 ? 1.0f / (float)qat_i8_factor(qt) : 1.0f;
    for (uint32_t m = wave; m < mtiles; m += QAT_WAVES) {
        f32x16 acc = {0};
        if (u8) {
            // int8 matrix cores: A = 16 weight bytes of output 32m + i at K = 32s + 16h .., B = 16 activation bytes of row i.
            // The int32 sums are exact, |sum| <= 1024 * 128 * 127 < 2^24, so the float conversion and the division by the
            // factor are exact too: bit-identical to the fp32 path's sums.
            i32x16 ai;
#pragma unroll
            for (int r = 0; r < 16; r++) ai[r] = 0;
            const int8_t *ap8 = u8 + (uint64_t)(32u * m + (uint32_t)i) * d8 + 16u * (uint32_t)h;
            const int8_t *bp8 = q8 + (uint32_t)i * rs8 + 16u * (uint32_t)h;
            for (uint32_t s8 = 0; s8 < d8 / 32u; s8++)
                ai = __builtin_amdgcn_mfma_i32_32x32x32_i8(*(const i32x4 *)(ap8 + 32u * s8), *(const i32x4 *)(bp8 + 32u * s8), ai, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 16; r++) acc[r] = (float)ai[r] * inv_f;
        } else {
            const float *ap = uT + (uint64_t)h * kpad + 32u * m + i;   // A[out i][k = 2*s + h]
            const float *bp = q + i * rs + h;                          // B[k = 2*s + h][row i]
            for (uint32_t s2 = 0; s2 < dpad / 2u; s2++)
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[(uint64_t)2u * s2 * kpad], bp[2u * s2], acc, 0, 0, 0);
        }
        const uint64_t row = row0 + (uint64_t)i;
        if (row >= n) continue;
        const float xsc = xs[i];
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const uint32_t out = 32u * m + (r & 3) + 8u * (r >> 2) + 4u * h;
            if (out < k) {
                float v = acc[r];
                if (qt != BNM_QAT_NONE) v = __fdiv_rn(__fdiv_rn(v, xsc), w_scale[out]);
                y[row * k + out] = v;
            }
        }
    }
}

// BitConv2d forward (BitNetMCU.py:264-322): any group structure (groups = 1, depthwise, anything between), any stride, kh x kw
// kernels, symmetric zero padding.  One workgroup per (image, group):
//   phase 1  the group's cin/groups input planes -> LDS, each inside a zero border; Normalize ('RMS' over each plane, :306-308,
//            or 'None'); activation_quant per image ROW of each plane (max over the last dimension, :125-127): the LDS planes
//            hold x_int / x_scale
//   phase 2  every thread produces outputs  sum_{plane, taps} plane[...] * (w_int / w_scale)  for the group's output channels
// dynamic LDS: cig x (h + 2p) x (w + 2p) planes + cog x cig x kh x kw dequantised taps
__global__ __launch_bounds__(256) void qat_bitconv2d_fwd_kernel(const float *__restrict__ x, uint32_t cin, uint32_t h,
                                                                uint32_t wd, const float *__restrict__ uT,
                                                                const float *__restrict__ w_scale, uint32_t cout,
                                                                uint32_t kpad, uint32_t kh, uint32_t kw, uint32_t pad,
                                                                uint32_t stride, uint32_t groups, int qt, int nt,
                                                                float *__restrict__ y) {
    extern __shared__ float lds[];
    __shared__ float red[4];
    const uint32_t img = blockIdx.x / groups, g = blockIdx.x % groups;
    const uint32_t cig = cin / groups, cog = cout / groups;     // input / output channels of one group
    const uint32_t hp = h + 2u * pad, wp = wd + 2u * pad;
    const uint32_t ho = (hp - kh) / stride + 1u, wo = (wp - kw) / stride + 1u;
    const uint32_t d = cig * kh * kw;                 // one output channel's weights: [cig][kh][kw]
    float *planes = lds;
    float *taps = lds + cig * hp * wp;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

    for (uint32_t i = threadIdx.x; i < cig * hp * wp; i += 256u) planes[i] = 0.0f;
    for (uint32_t i = threadIdx.x; i < cog * d; i += 256u) {
        const uint32_t co = g * cog + i / d, t = i % d;
        const float u = uT[(uint64_t)t * kpad + co];
        taps[i] = qt == BNM_QAT_NONE ? u : __fdiv_rn(u, w_scale[co]);
    }
    for (uint32_t c = 0; c < cig; c++) {
        const float *xp = x + ((uint64_t)img * cin + g * cig + c) * h * wd;
        float *plane = planes + c * hp * wp;
        float den = 1.0f;
        __syncthreads();                              // the zero fill (c = 0) / the previous plane's use of red[]
        if (nt == BNM_QAT_NORM_RMS) {
            float a = 0.0f;
            for (uint32_t i = threadIdx.x; i < h * wd; i += 256u) a += xp[i] * xp[i];
            a = wave_sum(a);
            if (lane == 0) red[wave] = a;
            __syncthreads();
            den = sqrtf((red[0] + red[1] + red[2] + red[3]) / (float)(h * wd));
        }
        for (uint32_t r = wave; r < h; r += 4u) {
            float *pr = plane + (r + pad) * wp + pad;
            float mx = 0.0f;
            for (uint32_t cc = lane; cc < wd; cc += 64u) {
                float v = xp[r * wd + cc];
                if (nt == BNM_QAT_NORM_RMS) v = __fdiv_rn(v, den);
                pr[cc] = v;
                mx = fmaxf(mx, fabsf(v));
            }
            if (qt == BNM_QAT_NONE) continue;
            mx = wave_max(mx);
            const float sc = __fdiv_rn(127.0f, fmaxf(mx, 1e-5f));
            for (uint32_t cc = lane; cc < wd; cc += 64u)
                pr[cc] = __fdiv_rn(fminf(fmaxf(rintf(__fmul_rn(pr[cc], sc)), -128.0f), 127.0f), sc);
        }
    }
    __syncthreads();
    const uint32_t per = ho * wo;
    for (uint32_t i = threadIdx.x; i < cog * per; i += 256u) {
        const uint32_t cl = i / per, oy = (i % per) / wo, ox = i % wo;
        const float *t = taps + cl * d;
        float acc = 0.0f;
        for (uint32_t c = 0; c < cig; c++) {
            const float *plane = planes + c * hp * wp + (oy * stride) * wp + ox * stride;
            for (uint32_t dy = 0; dy < kh; dy++)
                for (uint32_t dx = 0; dx < kw; dx++) acc = fmaf(plane[dy * wp + dx], t[(c * kh + dy) * kw + dx], acc);
        }
        y[(((uint64_t)img * cout + g * cog + cl) * ho + oy) * wo + ox] = acc;
    }
}

size_t bnmk_qat_bitconv2d_lds_bytes(uint32_t cin, uint32_t h, uint32_t wd, uint32_t cout, uint32_t kh, uint32_t kw, uint32_t pad,
                                    uint32_t groups) {
    const size_t cig = cin / groups, cog = cout / groups;
    return (cig * (size_t)(h + 2u * pad) * (wd + 2u * pad) + cog * cig * kh * kw) * sizeof(float);
}

hipError_t bnmk_qat_bitconv2d_forward(const float *x, uint64_t n, uint32_t cin, uint32_t h, uint32_t wd, const float *w,
                                      uint32_t cout, uint32_t kh, uint32_t kw, uint32_t pad, uint32_t stride, uint32_t groups,
                                      const float *s, int qt, int nt, float *y, float *workspace, hipStream_t st) {
    const uint32_t d = (cin / groups) * kh * kw, dpad = (d + 1u) & ~1u, kpad = (cout + 31u) & ~31u;
    float *uT = workspace;
    float *w_scale = uT + (size_t)dpad * kpad;
    float *stats = w_scale + kpad;
    hipError_t e = hipMemsetAsync(workspace, 0, ((size_t)dpad * kpad + kpad + 4u) * sizeof(float), st);
    if (e != hipSuccess) return e;
    if (n == 0) return hipSuccess;
    const uint64_t cnt = (uint64_t)cout * d;
    if (qt == BNM_QAT_TERNARY || qt == BNM_QAT_BINARY) qat_weight_stats_kernel<<<dim3(1), dim3(1024), 0, st>>>(w, cnt, stats);
    qat_weight_quant_kernel<<<dim3((unsigned)((cnt + 255u) / 256u)), dim3(256), 0, st>>>(w, cout, d, s, 1u, qt, stats, kpad, uT,
                                                                                   w_scale, nullptr, nullptr, 0u);
    const size_t lds_bytes = bnmk_qat_bitconv2d_lds_bytes(cin, h, wd, cout, kh, kw, pad, groups);
    if (lds_bytes > 64u * 1024u) {
        e = hipFuncSetAttribute((const void *)qat_bitconv2d_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
    }
    qat_bitconv2d_fwd_kernel<<<dim3((unsigned)(n * groups)), dim3(256), lds_bytes, st>>>(x, cin, h, wd, uT, w_scale, cout, kpad, kh, kw,
                                                                                        pad, stride, groups, qt, nt, y);
    return hipGetLastError();
}

// float part: uT [dpad][kpad], w_scale [kpad], stats [4], BatchNorm mean / den [2 d]; then the int8 weight rows [kpad][d8]
size_t bnmk_qat_workspace_bytes(uint32_t d, uint32_t k) {
    const size_t dpad = (d + 1u) & ~1u, kpad = (k + 31u) & ~31u, d8 = ((size_t)d + 31u) & ~(size_t)31u;
    return (dpad * kpad + kpad + 4u + 2u * (size_t)d) * sizeof(float) + kpad * d8;
}

hipError_t bnmk_qat_bitlinear_forward(const float *x, uint64_t n, uint32_t d, const float *w, uint32_t k, const float *s,
                                      uint32_t s_count, int qt, int nt, float *y, float *workspace, float *x_int_out,
                                      float *x_scale_out, float *w_deq_out, hipStream_t st) {
    const uint32_t dpad = (d + 1u) & ~1u, kpad = (k + 31u) & ~31u;
    float *uT = workspace;
    float *w_scale = uT + (size_t)dpad * kpad;
    float *stats = w_scale + kpad;
    float *bn_mean = stats + 4, *bn_den = bn_mean + d;
    const uint32_t d8 = (d + 31u) & ~31u;
    // int8 matrix path for the types that allow it - unless the extra byte tile no longer fits beside the float tile (d > ~900)
    const bool i8_fits = ((size_t)QAT_ROWS * (dpad + 1u) + QAT_ROWS) * sizeof(float) + (size_t)QAT_ROWS * (d8 + 16u) <= 160u * 1024u;
    int8_t *u8 = (qat_i8_factor(qt) && i8_fits) ? (int8_t *)(bn_den + d) : nullptr;
    hipError_t e = hipMemsetAsync(workspace, 0, ((size_t)dpad * kpad + kpad + 4u) * sizeof(float), st);
    if (e != hipSuccess) return e;
    if (u8 && (e = hipMemsetAsync(u8, 0, (size_t)kpad * d8, st)) != hipSuccess) return e;
    if (n == 0) return hipSuccess;
    if (qt == BNM_QAT_TERNARY || qt == BNM_QAT_BINARY)
        qat_weight_stats_kernel<<<dim3(1), dim3(1024), 0, st>>>(w, (uint64_t)k * d, stats);
    const uint64_t cnt = (uint64_t)k * d;
    qat_weight_quant_kernel<<<dim3((unsigned)((cnt + 255u) / 256u)), dim3(256), 0, st>>>(w, k, d, s, s_count, qt, stats, kpad, uT,
                                                                                   w_scale, w_deq_out, u8, d8);
    if (nt == BNM_QAT_NORM_BATCHNORM)
        qat_batch_stats_kernel<<<dim3((d + 255u) / 256u), dim3(256), 0, st>>>(x, n, d, bn_mean, bn_den);
    const size_t lds_bytes = ((size_t)QAT_ROWS * (dpad + 1u) + QAT_ROWS) * sizeof(float) + (u8 ? (size_t)QAT_ROWS * (d8 + 16u) : 0u);
    if (lds_bytes > 64u * 1024u) {
        e = hipFuncSetAttribute((const void *)qat_bitlinear_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
    }
    const uint64_t blocks = (n + QAT_ROWS - 1) / QAT_ROWS;
    qat_bitlinear_fwd_kernel<<<dim3((unsigned)blocks), dim3(64 * QAT_WAVES), lds_bytes, st>>>(
        x, n, d, dpad, uT, w_scale, k, kpad, qt, nt, bn_mean, bn_den, y, x_int_out, x_scale_out, u8, d8);
    return hipGetLastError();
}
