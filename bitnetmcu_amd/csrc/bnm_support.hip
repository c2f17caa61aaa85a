// Support kernels: synthetic workload, class digest, GPU weight unpack, MFMA fragment builder, input quantisation.
// gfx950 (CDNA4 / MI355X) only; see DESIGN.md for layouts and rooflines.  Reference semantics:
// BitNetMCU_inference.c:23-72 (ReLUNorm), :88-208 (processfclayer), :238-277 (conv), :300-322 (pool);
// schedule BitNetMCU_MNIST_dll.c:48-121.
#include "bnm_device.hpp"
#include "bnm_quantise_f32.hpp"

// =================================================================================================
// Synthetic workload (host statement: oracle/synth.h)
// =================================================================================================
BNM_DEVICE uint64_t synth_word(uint64_t seed, int dist, uint64_t image, uint32_t w) {
    uint64_t x = splitmix64(seed + 32ull * image + w);
    if (dist == 0) return x;
    uint64_t x2 = splitmix64(x);
    uint64_t out = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        uint32_t b = (uint32_t)(x >> (8 * k)) & 0xFFu;
        uint32_t b2 = (uint32_t)(x2 >> (8 * k)) & 0xFFu;
        int v = (b < 169u) ? -20 : (int)(b2 % 148u) - 20;
        out |= (uint64_t)(uint8_t)v << (8 * k);
    }
    return out;
}

// one thread = 16 bytes (two 8-byte words) of one image; a wave writes 1 KiB contiguous
__global__ __launch_bounds__(256) void synth_fill_kernel(int8_t *dst, uint64_t first, uint64_t count,
                                                         uint64_t seed, int dist) {
    const uint64_t total = count * 16ull;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (uint64_t)gridDim.x * blockDim.x) {
        uint64_t img = i >> 4;
        uint32_t wp = (uint32_t)(i & 15u) * 2u;
        uint64_t a = synth_word(seed, dist, first + img, wp);
        uint64_t b = synth_word(seed, dist, first + img, wp + 1u);
        u32x4 v = {(uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32)};
        *(u32x4 *)(dst + i * 16ull) = v;
    }
}

hipError_t bnmk_synth_fill(int8_t *d, uint64_t first, uint64_t count, uint64_t seed, int dist, hipStream_t s) {
    if (count == 0) return hipSuccess;
    uint64_t blocks = (count * 16ull + 255ull) / 256ull;
    if (blocks > 256ull * 32ull) blocks = 256ull * 32ull;
    synth_fill_kernel<<<dim3((unsigned)blocks), dim3(256), 0, s>>>(d, first, count, seed, dist);
    return hipGetLastError();
}

// digest[0] += sum splitmix64((first+i)*64 + cls[i]); digest[1+c] += count(cls == c)
__global__ __launch_bounds__(256) void class_digest_kernel(const uint32_t *cls, uint64_t first, uint64_t n,
                                                           unsigned long long *out, uint32_t n_bins) {
    __shared__ unsigned long long sh[65];
    for (uint32_t i = threadIdx.x; i < 65; i += blockDim.x) sh[i] = 0;
    __syncthreads();
    unsigned long long acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t c = cls[i];
        acc += splitmix64((first + i) * 64ull + c);
        if (c < n_bins && c < 64u) atomicAdd(&sh[1 + c], 1ull);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if ((threadIdx.x & 63) == 0) atomicAdd(&sh[0], acc);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < 1 + n_bins && i < 65; i += blockDim.x)
        if (sh[i]) atomicAdd(&out[i], sh[i]);
}

hipError_t bnmk_class_digest(const uint32_t *cls, uint64_t first, uint64_t n, uint64_t *out, uint32_t n_bins,
                             hipStream_t s) {
    if (n == 0) return hipSuccess;
    uint64_t blocks = (n + 255ull) / 256ull;
    if (blocks > 2048) blocks = 2048;
    class_digest_kernel<<<dim3((unsigned)blocks), dim3(256), 0, s>>>(cls, first, n, (unsigned long long *)out, n_bins);
    return hipGetLastError();
}

// ---- unpack kernel: packed words -> int8 rows.  One thread = 4 consecutive k of one row. ----------
__global__ __launch_bounds__(256) void unpack_rows_kernel(const void *packed, int bpw, uint32_t n_input,
                                                          uint32_t n_real, uint32_t n_output, int8_t *lo,
                                                          int8_t *hi, uint32_t stride) {
    uint32_t quads = stride / 4u;
    uint32_t total = n_output * quads;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        uint32_t row = i / quads, q = i % quads;
        uint32_t plo = 0, phi = 0;
#pragma unroll
        for (uint32_t b = 0; b < 4; b++) {
            uint32_t k = 4u * q + b;
            int w = (k < n_real) ? decode_weight(packed, bpw, n_input, row, k) : 0;
            int l = w, h = 0;
            if (w == 128) { l = 64; h = 64; }   // only FP1.3.0's +2^7 does not fit int8 (-128 does)
            plo |= (uint32_t)(uint8_t)(int8_t)l << (8u * b);
            phi |= (uint32_t)(uint8_t)(int8_t)h << (8u * b);
        }
        *(uint32_t *)(lo + (size_t)row * stride + 4u * q) = plo;
        if (hi) *(uint32_t *)(hi + (size_t)row * stride + 4u * q) = phi;
    }
}

hipError_t bnmk_unpack_rows(const void *packed, int32_t bpw, uint32_t n_input, uint32_t n_real, uint32_t n_output,
                            int8_t *lo, int8_t *hi, uint32_t stride, hipStream_t s) {
    uint32_t total = n_output * (stride / 4u);
    if (!total) return hipSuccess;
    unpack_rows_kernel<<<dim3((total + 255u) / 256u), dim3(256), 0, s>>>(packed, bpw, n_input, n_real, n_output, lo, hi,
                                                                         stride);
    return hipGetLastError();
}

// ---- fragment builder: int8 rows -> A operands of v_mfma_i32_32x32x32_i8 -----------------------
// Fragment (m,s): lane l = (i = l&31, h = l>>5) holds 16 bytes = weights of output row 32m+i for the 16
// K indices this lane-half owns in K-step s.  The K index of byte t follows the B operand it will meet:
//   kmap 0 (layer fed by a raw image row):          k = 32s + 16h + t
//   kmap 1 (layer fed by the previous layer's packed ReLUNorm output, see relunorm_pack()):
//                                                   k = 32s + 8(t>>2) + 4h + (t&3)
// scale: 1, or 2 for hidden layers of the "doubled" kernels (see relunorm_pack<MT, true>).
__global__ __launch_bounds__(256) void build_fragments_kernel(const int8_t *rows, uint32_t stride, uint32_t n_output,
                                                              uint32_t n_real, uint32_t MT, uint32_t KT, int kmap,
                                                              int scale, int pad_row_weight, uint32_t *dst, uint32_t frag_dwords) {
    uint32_t total = MT * KT * 64u * 4u;   // dwords
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        uint32_t j = i & 3u, lane = (i >> 2) & 63u, frag = i >> 8;
        uint32_t s = frag % KT, m = frag / KT;
        uint32_t row = 32u * m + (lane & 31u), h = lane >> 5;
        uint32_t v = 0;
#pragma unroll
        for (uint32_t b = 0; b < 4; b++) {
            uint32_t k = kmap == 0 ? 32u * s + 16u * h + 4u * j + b : 32u * s + 8u * j + 4u * h + b;
            int w = k >= n_real ? 0 : row < n_output ? (int)rows[(size_t)row * stride + k] * scale : pad_row_weight;
            v |= (uint32_t)(uint8_t)(int8_t)w << (8u * b);
        }
        // fragment (m, s) at (m * KT + s) * frag_dwords; within it lane-linear
        dst[(size_t)frag * frag_dwords + (i & 255u)] = v;
    }
}

hipError_t bnmk_build_fragments(const int8_t *rows, uint32_t stride, uint32_t n_output, uint32_t n_real, uint32_t MT,
                                uint32_t KT, int kmap, int scale, int pad_row_weight, void *dst, uint32_t frag_stride, hipStream_t s) {
    uint32_t total = MT * KT * 256u;
    if (!total) return hipSuccess;
    if (frag_stride < 1024u || (frag_stride & 1023u)) return hipErrorInvalidValue;
    build_fragments_kernel<<<dim3((total + 255u) / 256u), dim3(256), 0, s>>>(rows, stride, n_output, n_real, MT, KT, kmap,
                                                                             scale, pad_row_weight, (uint32_t *)dst, frag_stride / 4u);
    return hipGetLastError();
}

// =================================================================================================
// Input quantisation: the step immediately before the path, which the reference does in
// Python for every image (test_inference.py:140-141, same formula BitNetMCU.py:435-436):
//     scale = 127.0 / max(max|x|, 1e-5);  q = clip(round_half_even(x * scale), -128, 127)   all in float32.
// One wavefront per image (256 floats = one float4 per lane); IEEE float32 divide / multiply and the round-to-nearest-even of a
// float32 add (bnm_quantise_f32.hpp), so the result is bit-identical to numpy's float32 arithmetic.
// =================================================================================================
__global__ __launch_bounds__(256) void quantize_input_kernel(const float *__restrict__ x, uint64_t n, int8_t *__restrict__ out,
                                                            unsigned long long *__restrict__ nonfinite) {
    const int lane = threadIdx.x & 63;
    uint32_t bad = 0;
    for (uint64_t img = (uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6); img < n; img += (uint64_t)gridDim.x * 4u) {
        const f32x4 v = __builtin_nontemporal_load((const f32x4 *)(x + img * 256ull + 4u * lane));     // every byte is touched once
        uint32_t m = absmax4_bits(v);      // (non-negative floats order as unsigned integers)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = umax(m, (uint32_t)__shfl_xor((int)m, off));
        // the same per-value arithmetic as the fused float-input kernels (bnm_quantise_f32.hpp): identical bytes for EVERY input -
        // an image with a NaN or an infinity becomes all zeros (numpy's result on x86) and is counted
        __builtin_nontemporal_store(quantise4_image(v, quantise_scale(m), bad), (uint32_t *)(out + img * 256ull + 4u * lane));
    }
    report_nonfinite(nonfinite, bad);
}

hipError_t bnmk_quantize_input(const float *x, uint64_t n, int8_t *out, unsigned long long *nonfinite, hipStream_t s) {
    if (!n) return hipSuccess;
    uint64_t blocks = (n + 3) / 4;
    if (blocks > 8192) blocks = 8192;
    quantize_input_kernel<<<dim3((unsigned)blocks), dim3(256), 0, s>>>(x, n, out, nonfinite);
    return hipGetLastError();
}


// ---- the box's plain read rate ------------------------------------------------------------------------------------
// Plain nontemporal 16 B/lane loads of a resident buffer, grid-stride, four loads in flight per lane, XOR-folded; nothing is
// written (one conditional dword that practically never fires keeps the loads alive).  bench.py times this over the SAME
// image set next to the inference kernels: an HBM-bound kernel's distance from its binding roofline is its time relative
// to just reading its input on THIS box (boxes differ by 10 % in what they stream).
__global__ __launch_bounds__(256) void stream_read_kernel(const u32x4 *__restrict__ src, uint64_t n16, uint32_t *__restrict__ sink) {
    uint32_t acc = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        const u32x4 a = __builtin_nontemporal_load(src + i), b = __builtin_nontemporal_load(src + i + stride);
        const u32x4 c = __builtin_nontemporal_load(src + i + 2 * stride), d = __builtin_nontemporal_load(src + i + 3 * stride);
        acc ^= a[0] ^ a[1] ^ a[2] ^ a[3] ^ b[0] ^ b[1] ^ b[2] ^ b[3] ^ c[0] ^ c[1] ^ c[2] ^ c[3] ^ d[0] ^ d[1] ^ d[2] ^ d[3];
    }
    for (; i < n16; i += stride) {
        const u32x4 a = __builtin_nontemporal_load(src + i);
        acc ^= a[0] ^ a[1] ^ a[2] ^ a[3];
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

// ---- the box's mixed read / write stream rate ----------------------------------------------------------------------
// What the inference kernels' memory traffic costs with NO arithmetic: a wave reads a tile of 32 rows of 256 bytes (8 nontemporal
// 16 B/lane loads, as the fused kernels do) and writes the tile's 32 x out_bytes contiguous result bytes (a fold of what it
// read) as nontemporal 16 B/lane stores - 44 bytes per row for ids + ten int32 logits.  Tiles grid-stride over the waves.
// bench.py times it next to the logits row: an HBM stream that mixes 13 % writes into its reads is slower per byte than a pure
// read, and this is by how much on the box at hand.
template <int BATCH, int STORE>
__global__ __launch_bounds__(256) void stream_rw_kernel(const u32x4 *__restrict__ src, uint64_t n_tiles, u32x4 *__restrict__ dst,
                                                        uint32_t out16_per_tile) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t waves = (uint64_t)gridDim.x * (blockDim.x >> 6);
    // a wave takes BATCH consecutive tiles at a time: all their reads first, then all their (contiguous) result bytes
    for (uint64_t t = ((uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * BATCH; t < n_tiles; t += waves * BATCH) {
        u32x4 a[BATCH];
#pragma unroll
        for (int j = 0; j < BATCH; j++) {
            const uint64_t tj = t + j < n_tiles ? t + j : n_tiles - 1;
            const u32x4 *p = src + tj * 512u + lane;
            a[j] = __builtin_nontemporal_load(p);
#pragma unroll
            for (int k = 1; k < 8; k++) a[j] ^= __builtin_nontemporal_load(p + 64 * k);
        }
#pragma unroll
        for (int j = 0; j < BATCH; j++) {
            if (t + j >= n_tiles) break;
            u32x4 *q = dst + (t + j) * out16_per_tile;
            for (uint32_t i = lane; i < out16_per_tile; i += 64u) {
                if constexpr (STORE == 0) __builtin_nontemporal_store(a[j], q + i);
                else q[i] = a[j];
            }
        }
    }
}

// mode: tiles per batch (1, 2, 4, 8; 0 = 2, what the dual-tile kernel does) + 16 for plain instead of nontemporal stores
// + 32 x (waves per SIMD - 2) (default two: CUs x 2 blocks of 256 threads... see the launch)
hipError_t bnmk_stream_rw(const void *d_src, uint64_t n_rows, void *d_dst, uint32_t out_bytes_per_row, uint32_t mode, hipStream_t s) {
    const uint64_t tiles = n_rows / 32ull;      // (whole tiles only: a rate probe)
    if (!tiles) return hipSuccess;
    const uint32_t batch = (mode & 15u) ? (mode & 15u) : 2u, plain = (mode >> 4) & 1u, wps = ((mode >> 5) & 7u) + 2u;
    const dim3 grid((unsigned)bnm_num_cus() * wps), block(256);
    const u32x4 *src = (const u32x4 *)d_src;
    u32x4 *dst = (u32x4 *)d_dst;
    const uint32_t o16 = out_bytes_per_row * 2u;
#define BNM_RW(B, S) stream_rw_kernel<B, S><<<grid, block, 0, s>>>(src, tiles, dst, o16)
    switch (batch * 2u + plain) {
    case 2: BNM_RW(1, 0); break;
    case 3: BNM_RW(1, 1); break;
    case 4: BNM_RW(2, 0); break;
    case 5: BNM_RW(2, 1); break;
    case 8: BNM_RW(4, 0); break;
    case 9: BNM_RW(4, 1); break;
    case 16: BNM_RW(8, 0); break;
    case 17: BNM_RW(8, 1); break;
    default: return hipErrorInvalidValue;
    }
#undef BNM_RW
    return hipGetLastError();
}

hipError_t bnmk_stream_read(const void *d_src, uint64_t bytes, uint32_t *d_sink, hipStream_t s) {
    if (bytes < 16) return hipSuccess;
    stream_read_kernel<<<dim3((unsigned)bnm_num_cus() * 8u), dim3(256), 0, s>>>((const u32x4 *)d_src, bytes / 16ull, d_sink);
    return hipGetLastError();
}
