// Layer-wise ALU kernels behind the reference's per-function symbols (bit-serial unpack + wave-shuffle reduction).
// gfx950 (CDNA4 / MI355X) only; see DESIGN.md for layouts and rooflines.  Reference semantics:
// BitNetMCU_inference.c:23-72 (ReLUNorm), :88-208 (processfclayer), :238-277 (conv), :300-322 (pool);
// schedule BitNetMCU_MNIST_dll.c:48-121.
#include "bnm_device.hpp"

// =================================================================================================
// Layer-wise ALU kernels: the reference's own structure (one call per layer), north-star style:
// a wavefront owns one output neuron, the packed weight row and the int8 activation vectors are staged
// in LDS, every lane unpacks its own weight word(s) with shifts/masks, partial int32 dot products are
// reduced with wave shuffles.  Used behind the processfclayer/ReLUNorm symbols, for codecs/shapes outside
// the fused table, and as an independent cross-check of the MFMA path.
// =================================================================================================
constexpr int LW_IMGS = 8;      // images per workgroup pass
constexpr int LW_MAXIN = 1024;  // activations per vector that relunorm_kernel holds in registers (longer vectors: relunorm_big_kernel)
// activations per LDS pass of the FC kernel: a multiple of every codec's inputs-per-element (32, 16, 8, 4 per 32-bit word,
// 10 per ternary chunk), so a pass always starts on an element boundary; longer rows take several passes
constexpr int LW_KCHUNK = 960;

__global__ __launch_bounds__(256) void fc_layer_bitserial_kernel(const int8_t *__restrict__ act, uint32_t act_stride,
                                                                 const void *__restrict__ packed, int bpw,
                                                                 uint32_t n_input, uint32_t n_output,
                                                                 int32_t *__restrict__ out, uint64_t batch) {
    __shared__ __attribute__((aligned(16))) int8_t s_act[LW_IMGS][LW_KCHUNK + 16];
    __shared__ __attribute__((aligned(16))) uint32_t s_w[4][LW_KCHUNK / 4 + 4];   // 4 neuron rows, one pass each
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const uint32_t row = blockIdx.x * 4u + (uint32_t)wave;
    const int fb = codec_field_bits(bpw);
    const uint32_t per_word = fb ? 32u / (uint32_t)fb : 10u;
    // elements of the packed row: 32-bit words, or 16-bit chunks for ternary
    const uint32_t row_elems = bpw == 64 ? n_input / 10u : (fb ? (n_input + per_word - 1u) / per_word : 0u);
    const uint32_t pass_elems = (uint32_t)LW_KCHUNK / per_word;
    const bool known = bpw == 64 || fb != 0;
    const uint32_t n_pass = known && row_elems ? (row_elems + pass_elems - 1u) / pass_elems : 1u;

    for (uint64_t base = (uint64_t)blockIdx.y * LW_IMGS; base < batch; base += (uint64_t)gridDim.y * LW_IMGS) {
        const uint32_t nimg = (uint32_t)((batch - base) < LW_IMGS ? (batch - base) : LW_IMGS);
        int32_t sum[LW_IMGS];
#pragma unroll
        for (int im = 0; im < LW_IMGS; im++) sum[im] = 0;
        for (uint32_t pass = 0; pass < n_pass; pass++) {
            const uint32_t e0 = pass * pass_elems, k0 = pass * (uint32_t)LW_KCHUNK;
            const uint32_t ne = row_elems - e0 < pass_elems ? row_elems - e0 : pass_elems;
            __syncthreads();
            // stage this wave's slice of its packed weight row ...
            if (row < n_output && known) {
                if (bpw == 64) {
                    const uint16_t *src = (const uint16_t *)packed + (size_t)row * row_elems + e0;
                    for (uint32_t i = lane; i < ne; i += 64) s_w[wave][i] = src[i];
                } else {
                    const uint32_t *src = (const uint32_t *)packed + (size_t)row * row_elems + e0;
                    for (uint32_t i = lane; i < ne; i += 64) s_w[wave][i] = src[i];
                }
            }
            // ... and the same slice of up to LW_IMGS activation vectors (only bytes < act_stride exist: ternary pads are never read)
            const uint32_t kn = act_stride > k0 ? (act_stride - k0 < (uint32_t)LW_KCHUNK ? act_stride - k0 : (uint32_t)LW_KCHUNK) : 0u;
            for (uint32_t i = threadIdx.x; i < nimg * kn; i += blockDim.x) {
                uint32_t im = i / kn, k = i % kn;
                s_act[im][k] = act[(base + im) * act_stride + k0 + k];
            }
            __syncthreads();
            if (row >= n_output || !known) continue;
            for (uint32_t e = lane; e < ne; e += 64) {
                uint32_t word = s_w[wave][e];
                for (uint32_t f = 0; f < per_word; f++) {
                    uint32_t k = e * per_word + f;           // within the pass
                    int w;
                    if (bpw == 64) {
                        word *= 3u;                       // BitNetMCU_inference.c:121-134
                        uint32_t digit = word >> 16;
                        word &= 0xFFFFu;
                        w = digit == 0 ? 1 : (digit == 1 ? -1 : 0);
                    } else {
                        uint32_t field = (word >> (32u - (uint32_t)fb * (f + 1u))) & ((1u << fb) - 1u);
                        w = decode_field(bpw, field);
                    }
                    if (w != 0 && k < kn) {
#pragma unroll
                        for (int im = 0; im < LW_IMGS; im++) sum[im] += w * (int)s_act[im][k];
                    }
                }
            }
        }
        if (row >= n_output) continue;
#pragma unroll
        for (int im = 0; im < LW_IMGS; im++) {
            int v = sum[im];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
            if (lane == 0 && (uint32_t)im < nimg) out[(base + im) * n_output + row] = v;
        }
    }
}

hipError_t bnmk_fc_layer(const int8_t *act, uint32_t act_stride, const void *packed, int32_t bpw, uint32_t n_input,
                         uint32_t n_output, int32_t *out, uint64_t batch, hipStream_t s) {
    if (!batch || !n_output) return hipSuccess;
    uint64_t gy = (batch + LW_IMGS - 1) / LW_IMGS;
    if (gy > 8192) gy = 8192;
    fc_layer_bitserial_kernel<<<dim3((n_output + 3u) / 4u, (unsigned)gy), dim3(256), 0, s>>>(act, act_stride, packed, bpw,
                                                                                            n_input, n_output, out, batch);
    return hipGetLastError();
}

// =================================================================================================
// One FC layer of a batch on the matrix cores: out[img][neuron] = sum_k act[img][k] * w[neuron][k], int8 x int8 -> int32,
// any layer width, any input length.  The layer-wise path of models that fit no fused kernel (a layer wider than 256
// outputs, input rows longer than 512 bytes, fragments beyond LDS): every codec the C engine decodes unpacks to int8 rows
// (FP1.3.0's +128: a second plane), so processfclayer (BitNetMCU_inference.c:88-208) is an integer GEMM whatever the width.
// A wave computes a 32-image x (MT x 32)-neuron block with v_mfma_i32_32x32x32_i8: B operand = 16 activation bytes of image
// j = lane & 31 straight from its row, A operand = 16 weight bytes of neuron row i straight from the unpacked rows (L2-resident),
// natural K order on both sides; every activation load feeds MT MFMAs.  Rows and activations are padded to 32-byte K-steps:
// the rows' padding is zero, so whatever the activation bytes behind a short row are (the next image's, scratch) is harmless.
// Not tuned to a roofline - it only has to keep shapes outside the fused kernels off the bit-serial kernel (~20x slower).
// =================================================================================================
template <int MT>
__global__ __launch_bounds__(256) void fc_layer_mfma_kernel(const int8_t *__restrict__ act, uint32_t act_stride,
                                                            const int8_t *__restrict__ rows_lo, const int8_t *__restrict__ rows_hi,
                                                            uint32_t row_stride, uint32_t kt, uint32_t n_output,
                                                            int32_t *__restrict__ out, uint64_t batch) {
    const int lane = threadIdx.x & 63;
    const int j = lane & 31, h = lane >> 5;
    const uint64_t tile = (uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6);      // 32-image tile
    const uint32_t m0 = blockIdx.y * (uint32_t)MT;                               // first 32-neuron tile of this wave
    if (tile * 32ull >= batch) return;
    uint64_t img = tile * 32ull + (uint64_t)j;
    const bool live = img < batch;
    if (!live) img = batch - 1ull;
    const int8_t *ap = act + img * (uint64_t)act_stride + 16 * h;
    i32x16 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; m++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[m][r] = 0;
    for (uint32_t s = 0; s < kt; s++) {
        const i32x4 b = *(const i32x4 *)(ap + 32u * s);
#pragma unroll
        for (int m = 0; m < MT; m++) {
            const size_t wo = (size_t)(32u * (m0 + (uint32_t)m) + (uint32_t)j) * row_stride + 32u * s + 16u * (uint32_t)h;
            acc[m] = __builtin_amdgcn_mfma_i32_32x32x32_i8(*(const i32x4 *)(rows_lo + wo), b, acc[m], 0, 0, 0);
            if (rows_hi != nullptr) acc[m] = __builtin_amdgcn_mfma_i32_32x32x32_i8(*(const i32x4 *)(rows_hi + wo), b, acc[m], 0, 0, 0);
        }
    }
    if (!live) return;
    // lane (j, h), register 4q + e of tile m: neuron 32 (m0 + m) + 8 q + 4 h + e of image j - four consecutive outputs per quad
    typedef int v4a4 __attribute__((ext_vector_type(4), aligned(4)));
    int32_t *dst = out + img * (uint64_t)n_output;
#pragma unroll
    for (int m = 0; m < MT; m++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t row = 32u * (m0 + (uint32_t)m) + 8u * q + 4u * (uint32_t)h;
            if (row + 4u <= n_output) {
                *(v4a4 *)(dst + row) = v4a4{acc[m][4 * q], acc[m][4 * q + 1], acc[m][4 * q + 2], acc[m][4 * q + 3]};
            } else {
#pragma unroll
                for (int e = 0; e < 4; e++)
                    if (row + (uint32_t)e < n_output) dst[row + e] = acc[m][4 * q + e];
            }
        }
}

// rows_lo / rows_hi: unpacked int8 rows [round32(n_output)][row_stride] (rows past n_output zero; rows_hi nullptr unless the
// layer holds FP1.3.0's +128), row_stride a multiple of 32 >= the layer's real inputs; act rows 16-byte aligned with at least
// row_stride readable bytes behind the last row's start.
hipError_t bnmk_fc_layer_mfma(const int8_t *act, uint32_t act_stride, const int8_t *rows_lo, const int8_t *rows_hi, uint32_t row_stride,
                              uint32_t n_output, int32_t *out, uint64_t batch, hipStream_t s) {
    if (!batch || !n_output) return hipSuccess;
    if ((row_stride & 31u) || (act_stride & 15u) || ((uintptr_t)act & 15u)) return hipErrorInvalidValue;
    const uint32_t tiles = (n_output + 31u) / 32u, kt = row_stride / 32u;
    const uint64_t gx = (batch + 127ull) / 128ull;           // 4 waves = 4 image tiles per workgroup
    if (gx > 0x7fffffffull) return hipErrorInvalidValue;
    if (tiles >= 4 && tiles % 4 == 0)
        fc_layer_mfma_kernel<4><<<dim3((unsigned)gx, tiles / 4u), dim3(256), 0, s>>>(act, act_stride, rows_lo, rows_hi, row_stride, kt, n_output, out, batch);
    else if (tiles % 2 == 0)
        fc_layer_mfma_kernel<2><<<dim3((unsigned)gx, tiles / 2u), dim3(256), 0, s>>>(act, act_stride, rows_lo, rows_hi, row_stride, kt, n_output, out, batch);
    else
        fc_layer_mfma_kernel<1><<<dim3((unsigned)gx, tiles), dim3(256), 0, s>>>(act, act_stride, rows_lo, rows_hi, row_stride, kt, n_output, out, batch);
    return hipGetLastError();
}

// ReLUNorm, one wavefront per vector.  All inputs are read before any output is written, so `out` may
// alias `in` exactly as BitNetMCU_MNIST_dll.c:80 uses it.
__global__ __launch_bounds__(256) void relunorm_kernel(const int32_t *in, uint32_t n, int8_t *out, uint32_t out_stride,
                                                       uint32_t *argmax, uint64_t batch) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave0 = (uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6);
    for (uint64_t v = wave0; v < batch; v += (uint64_t)gridDim.x * 4u) {
        const int32_t *src = in + v * n;
        int32_t x[LW_MAXIN / 64];
        int bv = -INT_MAX;
        uint32_t bi = 255;
#pragma unroll
        for (int t = 0; t < LW_MAXIN / 64; t++) {
            uint32_t i = (uint32_t)lane + 64u * t;
            x[t] = i < n ? src[i] : INT_MIN;
            if (i < n && x[t] > bv) { bv = x[t]; bi = i; }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            int pv = __shfl_xor(bv, off);
            uint32_t pi = (uint32_t)__shfl_xor((int)bi, off);
            if (pv > bv || (pv == bv && pi < bi)) { bv = pv; bi = pi; }
        }
        int mx = max(bv, 0);
        uint32_t tt = (uint32_t)mx >> 7;
        int sh = tt ? 32 - __builtin_clz(tt) : 0;
        int rnd = (1 << sh) >> 1;
        int8_t *dst = out + v * out_stride;
#pragma unroll
        for (int t = 0; t < LW_MAXIN / 64; t++) {
            uint32_t i = (uint32_t)lane + 64u * t;
            if (i < n) {
                // (x + rounding wraps in 32 bits exactly as the compiled C does for inputs within 2^23 of INT32_MAX - no layer sum
                // gets there; written on unsigned operands so that hipcc may not assume the sum stays non-negative)
                int q = x[t] < 0 ? 0 : min((int)((uint32_t)x[t] + (uint32_t)rnd) >> sh, 127);
                dst[i] = (int8_t)q;
            }
        }
        if (argmax && lane == 0) argmax[v] = bi;
    }
}

// Vectors longer than LW_MAXIN: one workgroup per vector, two passes over global memory (maximum + first position, then the
// outputs).  `out` must not alias `in` here (every caller passes its own device buffers).
__global__ __launch_bounds__(256) void relunorm_big_kernel(const int32_t *__restrict__ in, uint32_t n, int8_t *__restrict__ out,
                                                           uint32_t out_stride, uint32_t *__restrict__ argmax, uint64_t batch) {
    __shared__ int s_v[4];
    __shared__ uint32_t s_i[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (uint64_t v = blockIdx.x; v < batch; v += gridDim.x) {
        const int32_t *src = in + v * n;
        int bv = -INT_MAX;
        uint32_t bi = 255;                  // ReLUNorm's "no maximum found" value (:25-37)
        for (uint32_t i = threadIdx.x; i < n; i += blockDim.x)
            if (src[i] > bv) { bv = src[i]; bi = i; }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            int pv = __shfl_xor(bv, off);
            uint32_t pi = (uint32_t)__shfl_xor((int)bi, off);
            if (pv > bv || (pv == bv && pi < bi)) { bv = pv; bi = pi; }
        }
        __syncthreads();
        if (lane == 0) { s_v[wave] = bv; s_i[wave] = bi; }
        __syncthreads();
#pragma unroll
        for (int w = 0; w < 4; w++)
            if (s_v[w] > bv || (s_v[w] == bv && s_i[w] < bi)) { bv = s_v[w]; bi = s_i[w]; }
        const int mx = max(bv, 0);
        const uint32_t tt = (uint32_t)mx >> 7;
        const int sh = tt ? 32 - __builtin_clz(tt) : 0;
        const int rnd = (1 << sh) >> 1;
        int8_t *dst = out + v * out_stride;
        for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
            const int x = src[i];
            dst[i] = (int8_t)(x < 0 ? 0 : min((int)((uint32_t)x + (uint32_t)rnd) >> sh, 127));
        }
        if (argmax && threadIdx.x == 0) argmax[v] = bi;
    }
}

hipError_t bnmk_relunorm(const int32_t *in, uint32_t n, int8_t *out, uint32_t out_stride, uint32_t *argmax,
                         uint64_t batch, hipStream_t s) {
    if (!batch || !n) return hipSuccess;
    if (n > LW_MAXIN) {
        if ((const void *)in == (const void *)out) return hipErrorInvalidValue;
        relunorm_big_kernel<<<dim3((unsigned)(batch < 4096 ? batch : 4096)), dim3(256), 0, s>>>(in, n, out, out_stride, argmax, batch);
        return hipGetLastError();
    }
    uint64_t blocks = (batch + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    relunorm_kernel<<<dim3((unsigned)blocks), dim3(256), 0, s>>>(in, n, out, out_stride, argmax, batch);
    return hipGetLastError();
}

// Single-channel 3x3 conv + ReLU + shift, and 2x2 max pool (the symbol-level entry points).  One
// workgroup; the whole input plane is read into LDS first so that output may alias input.
__global__ __launch_bounds__(256) void conv33_kernel(const int32_t *in, const int8_t *w, uint32_t xy, uint32_t n_shift,
                                                     int32_t *out) {
    __shared__ int32_t plane[64 * 64];
    for (uint32_t i = threadIdx.x; i < xy * xy; i += blockDim.x) plane[i] = in[i];
    int wk[9];
#pragma unroll
    for (int t = 0; t < 9; t++) wk[t] = w[t];
    __syncthreads();
    uint32_t o = xy - 2u;
    for (uint32_t i = threadIdx.x; i < o * o; i += blockDim.x) {
        uint32_t y = i / o, x = i % o;
        int s = 0;
#pragma unroll
        for (int dy = 0; dy < 3; dy++)
#pragma unroll
            for (int dx = 0; dx < 3; dx++) s += wk[3 * dy + dx] * plane[(y + dy) * xy + x + dx];
        out[i] = s < 0 ? 0 : (s >> n_shift);
    }
}

__global__ __launch_bounds__(256) void maxpool22_kernel(const int32_t *in, uint32_t xy, int32_t *out) {
    __shared__ int32_t plane[64 * 64];
    for (uint32_t i = threadIdx.x; i < xy * xy; i += blockDim.x) plane[i] = in[i];
    __syncthreads();
    uint32_t o = xy / 2u;
    for (uint32_t i = threadIdx.x; i < o * o; i += blockDim.x) {
        uint32_t y = i / o, x = i % o;
        const int32_t *p = plane + 2u * y * xy + 2u * x;
        out[i] = max(max(p[0], p[1]), max(p[xy], p[xy + 1]));
    }
}

// planes wider than 64: straight from global memory, many workgroups; `out` must not alias `in` (the host symbols copy the
// caller's buffers into device buffers of their own, so the reference's in-place use still works at the symbol level)
__global__ __launch_bounds__(256) void conv33_big_kernel(const int32_t *__restrict__ in, const int8_t *__restrict__ w, uint32_t xy,
                                                         uint32_t n_shift, int32_t *__restrict__ out) {
    int wk[9];
#pragma unroll
    for (int t = 0; t < 9; t++) wk[t] = w[t];
    const uint64_t o = xy - 2u, total = o * o;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t y = i / o, x = i % o;
        int s = 0;
#pragma unroll
        for (int dy = 0; dy < 3; dy++)
#pragma unroll
            for (int dx = 0; dx < 3; dx++) s += wk[3 * dy + dx] * in[(y + dy) * xy + x + dx];
        out[i] = s < 0 ? 0 : (s >> n_shift);
    }
}
__global__ __launch_bounds__(256) void maxpool22_big_kernel(const int32_t *__restrict__ in, uint32_t xy, int32_t *__restrict__ out) {
    const uint64_t o = xy / 2u, total = o * o;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t y = i / o, x = i % o;
        const int32_t *p = in + 2u * y * xy + 2u * x;
        out[i] = max(max(p[0], p[1]), max(p[xy], p[xy + 1]));
    }
}

hipError_t bnmk_conv33(const int32_t *in, const int8_t *w, uint32_t xy, uint32_t n_shift, int32_t *out, hipStream_t s) {
    if (xy < 3) return hipErrorInvalidValue;
    if (xy > 64) {
        if (in == out) return hipErrorInvalidValue;
        const uint64_t total = (uint64_t)(xy - 2u) * (xy - 2u), blocks = (total + 255) / 256;
        conv33_big_kernel<<<dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, s>>>(in, w, xy, n_shift, out);
    } else {
        conv33_kernel<<<dim3(1), dim3(256), 0, s>>>(in, w, xy, n_shift, out);
    }
    return hipGetLastError();
}
hipError_t bnmk_maxpool22(const int32_t *in, uint32_t xy, int32_t *out, hipStream_t s) {
    if (xy < 2) return hipErrorInvalidValue;
    if (xy > 64) {
        if (in == out) return hipErrorInvalidValue;
        const uint64_t total = (uint64_t)(xy / 2u) * (xy / 2u), blocks = (total + 255) / 256;
        maxpool22_big_kernel<<<dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, s>>>(in, xy, out);
    } else {
        maxpool22_kernel<<<dim3(1), dim3(256), 0, s>>>(in, xy, out);
    }
    return hipGetLastError();
}

