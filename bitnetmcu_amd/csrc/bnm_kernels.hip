// gfx950 (CDNA4 / MI355X) kernels for the BitNetMCU inference path.  Written for wave64, MFMA
// i8 32x32x32, 160 KiB LDS; no portability layer.  See DESIGN.md for the data layout and the
// roofline of each kernel.  Reference semantics: BitNetMCU_inference.c:23-72 (ReLUNorm),
// :88-208 (processfclayer), :238-277 (conv), :300-322 (pool); schedule
// BitNetMCU_MNIST_dll.c:48-121.
#include "bnm_kernels.h"
#include <climits>
#include <type_traits>

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define BNM_DEVICE __device__ __forceinline__

// compile-time counted loop: the body receives std::integral_constant<int, I>, so every array index derived
// from I is a constant and the arrays stay in registers
template <int B, int E, class F>
BNM_DEVICE void static_for(F &&f) {
    if constexpr (B < E) {
        f(std::integral_constant<int, B>{});
        static_for<B + 1, E>(f);
    }
}

// =================================================================================================
// Synthetic workload (host statement: oracle/synth.h)
// =================================================================================================
BNM_DEVICE uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

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

// =================================================================================================
// Weight codecs (exportquant.py:104-187 packs, BitNetMCU_inference.c:96-201 unpacks).
// decode_weight() is the single device-side statement of all codecs; both the unpack kernel and the
// bit-serial layer kernel go through it.
// =================================================================================================
BNM_DEVICE int codec_field_bits(int bpw) {
    return bpw == 1 ? 1 : bpw == 2 ? 2 : (bpw == 4 || bpw == 12 || bpw == 20) ? 4 : bpw == 16 ? 8 : 0;
}

// field f (fb bits, already right-aligned) -> integer weight
BNM_DEVICE int decode_field(int bpw, uint32_t f) {
    switch (bpw) {
        case 1: return f ? 1 : -1;                                          // :96-104
        case 2: return ((f & 2u) ? -1 : 1) * (int)(1u + 2u * (f & 1u));     // :105-115
        case 4: return ((f & 8u) ? -1 : 1) * (int)(2u * (f & 7u) + 1u);     // :156-168
        case 12: return (int)(f ^ 8u) - 8;                                  // :169-178
        case 16: return (int)(int8_t)f;                                     // :179-188
        case 20: return ((f & 8u) ? -1 : 1) * (int)(1u << (f & 7u));        // :190-201
    }
    return 0;
}

// trit t (0..9) of a 16-bit ternary chunk (:116-136): multiply-by-3 pops digits MSB first
BNM_DEVICE int ternary_trit(uint32_t chunk, uint32_t t) {
    uint32_t digit = 0;
    for (uint32_t i = 0; i <= t; i++) {
        chunk *= 3u;
        digit = chunk >> 16;
        chunk &= 0xFFFFu;
    }
    return digit == 0 ? 1 : (digit == 1 ? -1 : 0);
}

BNM_DEVICE int decode_weight(const void *packed, int bpw, uint32_t n_input, uint32_t row, uint32_t k) {
    if (bpw == 64) {
        uint32_t chunk = ((const uint16_t *)packed)[row * (n_input / 10u) + k / 10u];
        return ternary_trit(chunk, k % 10u);
    }
    int fb = codec_field_bits(bpw);
    if (!fb) return 0;
    uint32_t per_word = 32u / (uint32_t)fb;
    uint32_t words_per_row = (n_input + per_word - 1u) / per_word;
    uint32_t word = ((const uint32_t *)packed)[row * words_per_row + k / per_word];
    uint32_t f = (word >> (32u - (uint32_t)fb * (k % per_word + 1u))) & ((1u << fb) - 1u);
    return decode_field(bpw, f);
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
                                                              int scale, uint32_t *dst) {
    uint32_t total = MT * KT * 64u * 4u;   // dwords
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        uint32_t j = i & 3u, lane = (i >> 2) & 63u, frag = i >> 8;
        uint32_t s = frag % KT, m = frag / KT;
        uint32_t row = 32u * m + (lane & 31u), h = lane >> 5;
        uint32_t v = 0;
#pragma unroll
        for (uint32_t b = 0; b < 4; b++) {
            uint32_t k = kmap == 0 ? 32u * s + 16u * h + 4u * j + b : 32u * s + 8u * j + 4u * h + b;
            int w = (row < n_output && k < n_real) ? (int)rows[(size_t)row * stride + k] * scale : 0;
            v |= (uint32_t)(uint8_t)(int8_t)w << (8u * b);
        }
        dst[i] = v;
    }
}

hipError_t bnmk_build_fragments(const int8_t *rows, uint32_t stride, uint32_t n_output, uint32_t n_real, uint32_t MT,
                                uint32_t KT, int kmap, int scale, void *dst, hipStream_t s) {
    uint32_t total = MT * KT * 256u;
    if (!total) return hipSuccess;
    build_fragments_kernel<<<dim3((total + 255u) / 256u), dim3(256), 0, s>>>(rows, stride, n_output, n_real, MT, KT, kmap,
                                                                             scale, (uint32_t *)dst);
    return hipGetLastError();
}

// =================================================================================================
// Fused whole-model FC kernel.
//
// Formulation (per wave, per tile of 32 images):  Y^T[neurons x images] = W[neurons x K] * X^T[K x images]
// on v_mfma_i32_32x32x32_i8.  A = weight fragments (unpacked once per model, held in VGPRs for the whole
// persistent loop), B = activations: B-lane (j = lane&31, h = lane>>5) holds 16 K-bytes of image j.  The D
// layout gives lane (j,h) rows (r&3)+8(r>>2)+4h of image j, i.e. every lane owns half of its OWN image's
// outputs, so ReLUNorm's max is a per-lane reduction plus one v_permlane32_swap, and the normalised int8
// bytes packed 4 regs -> 1 dword are directly the next layer's B operand (the next layer's A fragments
// were built with the matching K permutation, kmap 1).  No LDS or cross-lane traffic between layers.
//
// Image tile load, variant 1: 8 x global_load_lds_dwordx4 (1 KiB contiguous each) into a per-wave
// double-buffered LDS tile, XOR-swizzled on the SOURCE side so that the ds_read_b128 B-operand reads are
// bank-conflict free; the next tile's DMA is issued before the current tile's math and retired with a
// counted s_waitcnt vmcnt(8).  Variant 0: direct global->VGPR loads in operand layout (any row length).
// =================================================================================================
template <int MT, int KS>
struct AFrags {
    i32x4 a[MT][KS];
    BNM_DEVICE void load(const i32x4 *base, int lane) {
#pragma unroll
        for (int m = 0; m < MT; m++)
#pragma unroll
            for (int s = 0; s < KS; s++) a[m][s] = base[(m * KS + s) * 64 + lane];
    }
};

BNM_DEVICE i32x16 zero16() {
    i32x16 z;
#pragma unroll
    for (int i = 0; i < 16; i++) z[i] = 0;
    return z;
}

template <int MT, int KT, bool SPLIT>
BNM_DEVICE void layer_mma(const AFrags<MT, KT *(SPLIT ? 2 : 1)> &A, const i32x4 (&b)[KT], i32x16 (&acc)[MT]) {
#pragma unroll
    for (int m = 0; m < MT; m++) acc[m] = zero16();
#pragma unroll
    for (int s = 0; s < KT; s++)
#pragma unroll
        for (int m = 0; m < MT; m++) acc[m] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A.a[m][s], b[s], acc[m], 0, 0, 0);
    if constexpr (SPLIT) {
#pragma unroll
        for (int s = 0; s < KT; s++)
#pragma unroll
            for (int m = 0; m < MT; m++)
                acc[m] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A.a[m][KT + s], b[s], acc[m], 0, 0, 0);
    }
}

// value of the partner lane (lane ^ 32)
BNM_DEVICE int partner32(int x, int h) {
    auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
    return h ? r[0] : r[1];
}

// clamp to [0, hi] in ONE instruction.  hipcc only forms v_med3_i32 from min(max(x, lo), hi) when it can prove
// lo <= hi (constants); with a run-time hi it emits v_max + v_min.
BNM_DEVICE int clamp0_med3(int x, int hi) {
    int r;
    asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(x), "v"(hi));
    return r;
}

// 16 clamped values -> 4 dwords, byte b of dword q = c[4q+b] >> s.  One SDWA shift per value writes its result
// byte straight into place (dst_sel:BYTE_b, dst_unused:UNUSED_PRESERVE), so no separate pack instructions.
// Same-register writes are 4 instructions apart and a trailing s_nop covers the dst_sel forwarding hazard that
// hipcc cannot see inside an asm statement.
BNM_DEVICE i32x4 sdwa_shift_pack16(const int (&c)[16], int s) {
    int d0, d1, d2, d3;
#define SD(dst, src, sel, unused) \
    "v_lshrrev_b32_sdwa " dst ", %4, " src " dst_sel:" sel " dst_unused:" unused " src0_sel:DWORD src1_sel:DWORD\n\t"
    asm(SD("%0", "%5", "BYTE_0", "UNUSED_PAD") SD("%1", "%9", "BYTE_0", "UNUSED_PAD")
        SD("%2", "%13", "BYTE_0", "UNUSED_PAD") SD("%3", "%17", "BYTE_0", "UNUSED_PAD")
        SD("%0", "%6", "BYTE_1", "UNUSED_PRESERVE") SD("%1", "%10", "BYTE_1", "UNUSED_PRESERVE")
        SD("%2", "%14", "BYTE_1", "UNUSED_PRESERVE") SD("%3", "%18", "BYTE_1", "UNUSED_PRESERVE")
        SD("%0", "%7", "BYTE_2", "UNUSED_PRESERVE") SD("%1", "%11", "BYTE_2", "UNUSED_PRESERVE")
        SD("%2", "%15", "BYTE_2", "UNUSED_PRESERVE") SD("%3", "%19", "BYTE_2", "UNUSED_PRESERVE")
        SD("%0", "%8", "BYTE_3", "UNUSED_PRESERVE") SD("%1", "%12", "BYTE_3", "UNUSED_PRESERVE")
        SD("%2", "%16", "BYTE_3", "UNUSED_PRESERVE") SD("%3", "%20", "BYTE_3", "UNUSED_PRESERVE")
        "s_nop 0"
        : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3)
        : "v"(s), "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]), "v"(c[4]), "v"(c[5]), "v"(c[6]), "v"(c[7]), "v"(c[8]),
          "v"(c[9]), "v"(c[10]), "v"(c[11]), "v"(c[12]), "v"(c[13]), "v"(c[14]), "v"(c[15]));
#undef SD
    i32x4 r = {d0, d1, d2, d3};
    return r;
}

// ReLUNorm (BitNetMCU_inference.c:23-72) on MT x 16 accumulator values per lane (+ the partner lane's),
// result packed as the next layer's B operand: packed[m][q] byte b = row 32m + 8q + 4h + b.
// Rows >= n_output are zero weights => value 0: they can only raise a negative maximum to 0, in which case
// every output is 0 either way.
//
// DBL = false: accumulators hold the layer sums x.   out = clamp((x + r) >> s, 0, 127), 4 VALU per value.
// DBL = true : this layer's weight fragments were built DOUBLED, accumulators hold 2x (exact).  With
//   s = bitlength(max(2x) >> 8) (= the reference's shift, from max(x) >> 7) and y = clamp(2x, 0, 255*2^s - 1) >> s
//   (0..254, one v_med3 + one SDWA shift that also packs), the rounded result is
//   (x + 2^(s-1)) >> s = (2x + 2^s) >> (s+1) = (y + 1) >> 1, which v_lerp_u8 computes for 4 bytes at once;
//   y <= 254 makes the "clip 128 to 127" case (:62-66) fall out.  2.25 VALU per value, bit-exact.
template <int MT, bool DBL>
BNM_DEVICE void relunorm_pack(const i32x16 (&acc)[MT], i32x4 (&packed)[MT], int h) {
    int mx = acc[0][0];
#pragma unroll
    for (int m = 0; m < MT; m++)
#pragma unroll
        for (int r = 0; r < 16; r++) mx = max(mx, acc[m][r]);
    mx = max(mx, partner32(mx, h));
    mx = max(mx, 0);
    if constexpr (DBL) {
        uint32_t t = (uint32_t)mx >> 8;
        int sh = t ? 32 - __builtin_clz(t) : 0;
        int hi = (255 << sh) - 1;
#pragma unroll
        for (int m = 0; m < MT; m++) {
            int c[16];
#pragma unroll
            for (int r = 0; r < 16; r++) c[r] = clamp0_med3(acc[m][r], hi);
            i32x4 y = sdwa_shift_pack16(c, sh);
#pragma unroll
            for (int q = 0; q < 4; q++) packed[m][q] = (int)__builtin_amdgcn_lerp((uint32_t)y[q], 0u, 0x01010101u);
        }
    } else {
        uint32_t t = (uint32_t)mx >> 7;
        int sh = t ? 32 - __builtin_clz(t) : 0;
        int rnd = (1 << sh) >> 1;
#pragma unroll
        for (int m = 0; m < MT; m++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                uint32_t d = 0;
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    int v = (acc[m][4 * q + b] + rnd) >> sh;
                    v = min(max(v, 0), 127);
                    d |= (uint32_t)v << (8 * b);
                }
                packed[m][q] = (int)d;
            }
    }
}

// first strict maximum over rows < n_classes (ReLUNorm's return value, :25-37).  key = value*256 + (255 - row):
// the largest key is the largest value and, among equals, the smallest row.  |value| < 2^23 for every layer that
// can be last (K <= 128, |act| <= 127, |w| <= 128).  Registers whose rows are all >= n_classes are skipped by
// wave-uniform branches.
template <int MT>
BNM_DEVICE uint32_t argmax_rows(const i32x16 (&acc)[MT], int h, uint32_t n_classes) {
    int best = INT_MIN;
#pragma unroll
    for (int m = 0; m < MT; m++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const uint32_t rowbase = 32u * m + (r & 3) + 8u * (r >> 2);   // row of the h = 0 half; h = 1: +4
            if (rowbase < n_classes) {
                int key = (int)(((uint32_t)acc[m][r] << 8) + (255u - rowbase));
                if (rowbase + 4u >= n_classes) key = h ? INT_MIN : key;
                best = max(best, key);
            }
        }
    best = best == INT_MIN ? INT_MIN : best - 4 * h;
    best = max(best, partner32(best, h));
    return 255u - ((uint32_t)best & 255u);
}

template <int MT>
BNM_DEVICE void store_logits(const i32x16 (&acc)[MT], int32_t *dst, int h, uint32_t n_classes) {
#pragma unroll
    for (int m = 0; m < MT; m++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            uint32_t row = 32u * m + (r & 3) + 8u * (r >> 2) + 4u * h;
            if (row < n_classes) dst[row] = acc[m][r];
        }
}

// 8 x 1 KiB LDS-DMA pieces of one 32-image tile.  p[t] wave-uniform base pointers, v[t] per-lane byte
// offsets, lds wave-uniform LDS byte address of the tile buffer.  The DMA destination is
// M0 + lane*16 (lane-linear); the swizzle lives in v[].  hipcc neither counts these loads nor waits for
// them: the caller retires them with bnm_wait_vmcnt<N>().
// NT: non-temporal policy (the image stream is read exactly once).  WAITLDS: first retire this wave's own
// outstanding ds_reads (s_waitcnt lgkmcnt(0)) — needed when the destination buffer was being read just before.
#define BNM_DMA8(NTS, PRE)                                                                                           \
    asm volatile(PRE "s_nop 4\n\t"                                                                                    \
                 "s_mov_b32 %0, m0\n\t"                                                                                \
                 "s_mov_b32 m0, %1\n\t"                                                                                \
                 "s_nop 0\n\t"                                                                                         \
                 "global_load_lds_dwordx4 %10, %2" NTS "\n\t"                                                          \
                 "s_add_u32 m0, m0, 0x400\n\t"                                                                         \
                 "s_nop 0\n\t"                                                                                         \
                 "global_load_lds_dwordx4 %11, %3" NTS "\n\t"                                                          \
                 "s_add_u32 m0, m0, 0x400\n\t"                                                                         \
                 "s_nop 0\n\t"                                                                                         \
                 "global_load_lds_dwordx4 %12, %4" NTS "\n\t"                                                          \
                 "s_add_u32 m0, m0, 0x400\n\t"                                                                         \
                 "s_nop 0\n\t"                                                                                         \
                 "global_load_lds_dwordx4 %13, %5" NTS "\n\t"                                                          \
                 "s_add_u32 m0, m0, 0x400\n\t"                                                                         \
                 "s_nop 0\n\t"                                                                                         \
                 "global_load_lds_dwordx4 %14, %6" NTS "\n\t"                                                          \
                 "s_add_u32 m0, m0, 0x400\n\t"                                                                         \
                 "s_nop 0\n\t"                                                                                         \
                 "global_load_lds_dwordx4 %15, %7" NTS "\n\t"                                                          \
                 "s_add_u32 m0, m0, 0x400\n\t"                                                                         \
                 "s_nop 0\n\t"                                                                                         \
                 "global_load_lds_dwordx4 %16, %8" NTS "\n\t"                                                          \
                 "s_add_u32 m0, m0, 0x400\n\t"                                                                         \
                 "s_nop 0\n\t"                                                                                         \
                 "global_load_lds_dwordx4 %17, %9" NTS "\n\t"                                                          \
                 "s_mov_b32 m0, %0"                                                                                    \
                 : "=&s"(keep)                                                                                         \
                 : "s"(lds), "s"(p0), "s"(p1), "s"(p2), "s"(p3), "s"(p4), "s"(p5), "s"(p6), "s"(p7), "v"(v0), "v"(v1), \
                   "v"(v2), "v"(v3), "v"(v4), "v"(v5), "v"(v6), "v"(v7)                                                \
                 : "memory", "scc")

template <bool NT = false, bool WAITLDS = false>
BNM_DEVICE void lds_dma_tile8(uint32_t lds, const int8_t *p0, const int8_t *p1, const int8_t *p2, const int8_t *p3,
                              const int8_t *p4, const int8_t *p5, const int8_t *p6, const int8_t *p7, uint32_t v0,
                              uint32_t v1, uint32_t v2, uint32_t v3, uint32_t v4, uint32_t v5, uint32_t v6,
                              uint32_t v7) {
    uint32_t keep;
    if constexpr (NT && WAITLDS) BNM_DMA8(" nt", "s_waitcnt lgkmcnt(0)\n\t");
    else if constexpr (NT) BNM_DMA8(" nt", "");
    else if constexpr (WAITLDS) BNM_DMA8("", "s_waitcnt lgkmcnt(0)\n\t");
    else BNM_DMA8("", "");
}

template <int N>
BNM_DEVICE void bnm_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

constexpr int FUSED_TILE_BYTES = 8192;    // 32 images x 256 B
constexpr int FUSED_WPB = 4;              // waves per workgroup; two workgroups per CU (LDS 64 KiB each)

// Kernel variants = how the image tile reaches the B operands:
//   0  DIRECT     global -> VGPR loads in operand layout (any row length; CNN tails with 64/128/192-byte rows)
//   1  LDSDMA     8 x 1 KiB LDS-DMA pieces into a per-wave double buffer, next tile issued at the top of the iteration
//   2  LDSDMA2    as 1 with TWO tiles in flight per wave (default where available): the refill of the buffer a tile
//                 just vacated (tile k+2) is issued right AFTER tile k's layer-1 MFMAs, non-temporal.  With one tile
//                 in flight the kernel is bound by per-wave memory-level parallelism (8 KiB / latency x 2048 waves);
//                 issuing the second DMA in front of the MFMAs instead serialises the wave (profiles/r01).
// Tried and dropped in round 1 (tag r01-experiments-all-variants): 8-wave workgroups with staggered halves, a
// software-pipelined MFMA||VALU form, three waves per SIMD with weights in LDS, split half-tile refills.
enum { FUSED_DIRECT = 0, FUSED_LDSDMA = 1, FUSED_LDSDMA2 = 2 };

template <int KT0, int M1, int M2, int M3, int M4, bool SPLIT, bool DBL, int VARIANT>
__global__ __launch_bounds__(64 * FUSED_WPB, 2) void fused_fc_kernel(const int8_t *__restrict__ images, uint64_t n,
                                                                     const i32x4 *__restrict__ frags, uint32_t n_classes,
                                                                     uint32_t *__restrict__ cls_out,
                                                                     int32_t *__restrict__ logits_out, uint64_t src_wrap) {
    constexpr int SP = SPLIT ? 2 : 1;
    constexpr int ROW = 32 * KT0;
    constexpr bool LDSDMA = VARIANT != FUSED_DIRECT;
    constexpr bool TWO = VARIANT == FUSED_LDSDMA2;
    static_assert(!LDSDMA || KT0 == 8, "the LDS-DMA tile layout is for 256-byte rows");
    static_assert(!(SPLIT && DBL), "FP1.3.0 weights cannot be doubled in int8");
    __shared__ __attribute__((aligned(1024))) char smem[LDSDMA ? FUSED_WPB * 2 * FUSED_TILE_BYTES : 16];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;

    // weights: unpacked fragments -> registers, once
    AFrags<M1, KT0 * SP> A1;
    AFrags<M2, M1 * SP> A2;
    AFrags<M3, M2 * SP> A3;
    AFrags<(M4 > 0 ? M4 : 1), M3 * SP> A4;
    const i32x4 *fp = frags;
    A1.load(fp, lane);  fp += M1 * KT0 * SP * 64;
    A2.load(fp, lane);  fp += M2 * M1 * SP * 64;
    A3.load(fp, lane);  fp += M3 * M2 * SP * 64;
    if constexpr (M4 > 0) A4.load(fp, lane);

    const uint64_t n_tiles = (n + 31ull) >> 5;
    const uint64_t stride = (uint64_t)gridDim.x * FUSED_WPB;
    uint64_t tile = (uint64_t)blockIdx.x * FUSED_WPB + wave;

    // ---- LDS-DMA addressing ------------------------------------------------------------------------
    // LDS tile image: row r (image) at r*256, 16-byte slot c' holds global slot c = c' ^ (r & 15).
    // DMA piece t covers rows 4t..4t+3: lane l -> row 4t + (l>>4), slot l&15.
    uint32_t voff[4];
    uint32_t lds_wave = 0, rd_base = 0;
    if constexpr (LDSDMA) {
#pragma unroll
        for (int u = 0; u < 4; u++) voff[u] = (uint32_t)(lane >> 4) * 256u + 16u * ((uint32_t)(lane & 15) ^ (uint32_t)(lane >> 4) ^ (4u * u));
        lds_wave = (uint32_t)(size_t)(__attribute__((address_space(3))) char *)smem + (uint32_t)wave * 2u * FUSED_TILE_BYTES;
        // B-operand read of K-step s: row j, global slot 2s+h -> LDS slot (2s+h) ^ (j&15) = (2s) ^ (h ^ (j&15))
        rd_base = (uint32_t)wave * 2u * FUSED_TILE_BYTES + (uint32_t)j * 256u + 16u * ((uint32_t)h ^ (uint32_t)(j & 15));
    }

    auto dma_tile = [&](uint64_t t, int par) {
        // src_wrap != 0 (diagnostics only, BNM_DIAG_SRC_WRAP): read tile (t mod src_wrap) instead, so the source stays
        // cache-resident and the kernel's compute-side time can be measured without HBM in the way
        const int8_t *base = images + (src_wrap ? t % src_wrap : t) * (uint64_t)FUSED_TILE_BYTES;
        uint32_t lds = lds_wave + (uint32_t)par * FUSED_TILE_BYTES;
        uint64_t first = t << 5;
        if (first + 32ull <= n) {
            lds_dma_tile8<TWO, TWO>(lds, base, base + 1024, base + 2048, base + 3072, base + 4096, base + 5120, base + 6144,
                                    base + 7168, voff[0], voff[1], voff[2], voff[3], voff[0], voff[1], voff[2], voff[3]);
        } else {
            // ragged last tile: rows past the end re-read the last valid image (never out of bounds)
            uint32_t nv = (uint32_t)(n - first);
            uint32_t v[8];
#pragma unroll
            for (int tt = 0; tt < 8; tt++) {
                uint32_t r = 4u * tt + (uint32_t)(lane >> 4);
                uint32_t src = r < nv ? r : nv - 1u;
                v[tt] = src * 256u + 16u * ((uint32_t)(lane & 15) ^ (r & 15u));
            }
            lds_dma_tile8<TWO, TWO>(lds, base, base, base, base, base, base, base, base, v[0], v[1], v[2], v[3], v[4], v[5],
                                    v[6], v[7]);
        }
    };

    int par = 0;
    i32x4 bnext[KT0];
    auto direct_load = [&](uint64_t t, i32x4(&dst)[KT0]) {
        uint64_t img = (t << 5) + (uint64_t)j;
        if (img >= n) img = n - 1ull;
        const int8_t *p = images + img * (uint64_t)ROW + 16 * h;
#pragma unroll
        for (int s = 0; s < KT0; s++) dst[s] = __builtin_nontemporal_load((const i32x4 *)(p + 32 * s));
    };

    if (tile < n_tiles) {
        if constexpr (LDSDMA) dma_tile(tile, 0);
        else direct_load(tile, bnext);
    }
    if constexpr (TWO) {
        if (tile + stride < n_tiles) dma_tile(tile + stride, 1);
    }

    for (; tile < n_tiles; tile += stride) {
        const uint64_t next = tile + stride;
        i32x4 b0[KT0];
        i32x16 acc1[M1];
        if constexpr (LDSDMA) {
            if constexpr (!TWO) {
                if (next < n_tiles) {
                    dma_tile(next, par ^ 1);
                    bnm_wait_vmcnt<8>();
                } else {
                    bnm_wait_vmcnt<0>();
                }
            } else {
                // in flight: this tile (8 pieces) and, if it exists, the next one (8 pieces, issued an iteration ago)
                if (next < n_tiles) bnm_wait_vmcnt<8>();
                else bnm_wait_vmcnt<0>();
            }
            // rd_base carries the slot field (h ^ (j&15)) << 4 in bits 4..7 and nothing else below bit 8, so
            // XOR-ing 32*s (bits 5..7) selects slot (2s+h) ^ (j&15): one v_xor per K-step.
#pragma unroll
            for (int s = 0; s < KT0; s++)
                b0[s] = *(const i32x4 *)(smem + ((rd_base ^ (32u * s)) + (uint32_t)par * FUSED_TILE_BYTES));
            if constexpr (!TWO) par ^= 1;
        } else {
#pragma unroll
            for (int s = 0; s < KT0; s++) b0[s] = bnext[s];
            if (next < n_tiles) direct_load(next, bnext);
        }

        layer_mma<M1, KT0, SPLIT>(A1, b0, acc1);
        if constexpr (TWO) {
            // all 8 B fragments of this tile's buffer have been consumed (the DMA statement first retires the wave's
            // own ds_reads): refill it with the tile after next
            if (next + stride < n_tiles) dma_tile(next + stride, par);
            par ^= 1;
        }
        i32x4 p1[M1];
        relunorm_pack<M1, DBL>(acc1, p1, h);

        i32x16 acc2[M2];
        layer_mma<M2, M1, SPLIT>(A2, p1, acc2);
        i32x4 p2[M2];
        relunorm_pack<M2, DBL>(acc2, p2, h);

        i32x16 acc3[M3];
        layer_mma<M3, M2, SPLIT>(A3, p2, acc3);

        const uint64_t img = (tile << 5) + (uint64_t)j;
        uint32_t cls;
        if constexpr (M4 > 0) {
            i32x4 p3[M3];
            relunorm_pack<M3, DBL>(acc3, p3, h);
            i32x16 acc4[M4];
            layer_mma<M4, M3, SPLIT>(A4, p3, acc4);
            cls = argmax_rows<M4>(acc4, h, n_classes);
            if (logits_out && img < n) store_logits<M4>(acc4, logits_out + img * n_classes, h, n_classes);
        } else {
            cls = argmax_rows<M3>(acc3, h, n_classes);
            if (logits_out && img < n) store_logits<M3>(acc3, logits_out + img * n_classes, h, n_classes);
        }
        if (h == 0 && img < n) cls_out[img] = cls;
    }
}

// ---- dispatch table: model shapes of the reference zoo (+ ternary) -----------------------------
namespace {
typedef void (*fused_fn)(const int8_t *, uint64_t, const i32x4 *, uint32_t, uint32_t *, int32_t *, uint64_t);
struct FusedEntry {
    BnmFusedShape sh;
    int variant;
    fused_fn fn;
};
#define FUSED(KT0, M1, M2, M3, M4, SPLIT, DBL, VAR) \
    { {KT0, {M1, M2, M3, M4}, SPLIT, DBL}, VAR, fused_fc_kernel<KT0, M1, M2, M3, M4, SPLIT, DBL, VAR> }
const FusedEntry kFused[] = {
    // FC 256-64-64-64-10 4bitsym (BitNetMCU_model_fc.h, mcu/BitNetMCU_model_12k.h) — the headline shape
    FUSED(8, 2, 2, 2, 1, false, true, FUSED_LDSDMA2),
    FUSED(8, 2, 2, 2, 1, false, true, FUSED_LDSDMA),
    FUSED(8, 2, 2, 2, 1, false, true, FUSED_DIRECT),
    // same shape with codecs whose weights cannot be doubled in int8 (8-bit two's complement)
    FUSED(8, 2, 2, 2, 1, false, false, FUSED_LDSDMA),
    FUSED(8, 2, 2, 2, 1, false, false, FUSED_DIRECT),
    // same shape, FP1.3.0 weights (mcu/BitNetMCU_model_12k_FP130.h): +128 split over two A passes
    FUSED(8, 2, 2, 2, 1, true, false, FUSED_LDSDMA),
    FUSED(8, 2, 2, 2, 1, true, false, FUSED_DIRECT),
    // FC 256-16-16-10 2bitsym (mcu/BitNetMCU_model_1k.h)
    FUSED(8, 1, 1, 1, 0, false, true, FUSED_LDSDMA2),
    FUSED(8, 1, 1, 1, 0, false, true, FUSED_LDSDMA),
    FUSED(8, 1, 1, 1, 0, false, true, FUSED_DIRECT),
    // ternary FC 256-96-96-96-10 through the MFMA path (optional; config 3's product path is the ALU kernel)
    FUSED(8, 3, 3, 3, 1, false, true, FUSED_LDSDMA),
    FUSED(8, 3, 3, 3, 1, false, true, FUSED_DIRECT),
    // CNN FC tails: 4C-96-64-10 (cnn_64/48/32/16), 64-64-48-10 (cnn_16small), 256-96-64-37 (letters)
    FUSED(8, 3, 2, 1, 0, false, true, FUSED_LDSDMA),
    FUSED(8, 3, 2, 1, 0, false, true, FUSED_DIRECT),
    FUSED(6, 3, 2, 1, 0, false, true, FUSED_DIRECT),
    FUSED(4, 3, 2, 1, 0, false, true, FUSED_DIRECT),
    FUSED(2, 3, 2, 1, 0, false, true, FUSED_DIRECT),
    FUSED(2, 2, 2, 1, 0, false, true, FUSED_DIRECT),
    FUSED(8, 3, 2, 2, 0, false, true, FUSED_LDSDMA),
    FUSED(8, 3, 2, 2, 0, false, true, FUSED_DIRECT),
};
const FusedEntry *find_fused(const BnmFusedShape &sh, int variant) {
    for (const FusedEntry &e : kFused)
        if (e.sh == sh && e.variant == variant) return &e;
    return nullptr;
}
int g_num_cus = 0;
int num_cus() {
    if (!g_num_cus) {
        int dev = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) g_num_cus = p.multiProcessorCount;
        if (g_num_cus <= 0) g_num_cus = 256;
    }
    return g_num_cus;
}
}  // namespace

bool bnmk_fused_supported(const BnmFusedShape &sh, int variant) { return find_fused(sh, variant) != nullptr; }
// measured best first (profiles/r01)
int bnmk_fused_default_variant(const BnmFusedShape &sh) {
    return find_fused(sh, FUSED_LDSDMA2) ? FUSED_LDSDMA2 : find_fused(sh, FUSED_LDSDMA) ? FUSED_LDSDMA : FUSED_DIRECT;
}

hipError_t bnmk_fused_fc(const BnmFusedShape &sh, int variant, int grid_blocks, const BnmFusedArgs &a, hipStream_t s) {
    const FusedEntry *e = find_fused(sh, variant);
    if (!e) return hipErrorInvalidValue;
    if (a.n == 0) return hipSuccess;
    uint64_t n_tiles = (a.n + 31ull) / 32ull;
    uint64_t want = (n_tiles + FUSED_WPB - 1) / FUSED_WPB;
    // persistent grid: 8 resident waves per CU (2 workgroups x 4 waves; VGPRs and LDS allow no more)
    uint64_t cap = grid_blocks > 0 ? (uint64_t)grid_blocks : (uint64_t)num_cus() * 2ull;
    unsigned blocks = (unsigned)(want < cap ? want : cap);
    e->fn<<<dim3(blocks), dim3(64 * FUSED_WPB), 0, s>>>(a.images, a.n, (const i32x4 *)a.frags, a.n_classes, a.cls, a.logits, a.src_wrap);
    return hipGetLastError();
}

// =================================================================================================
// Layer-wise ALU kernels: the reference's own structure (one call per layer), north-star style:
// a wavefront owns one output neuron, the packed weight row and the int8 activation vectors are staged
// in LDS, every lane unpacks its own weight word(s) with shifts/masks, partial int32 dot products are
// reduced with wave shuffles.  Used behind the processfclayer/ReLUNorm symbols, for codecs/shapes outside
// the fused table, and as an independent cross-check of the MFMA path.
// =================================================================================================
constexpr int LW_IMGS = 8;      // images per workgroup pass
constexpr int LW_MAXIN = 1024;  // activations per vector

__global__ __launch_bounds__(256) void fc_layer_bitserial_kernel(const int8_t *__restrict__ act, uint32_t act_stride,
                                                                 const void *__restrict__ packed, int bpw,
                                                                 uint32_t n_input, uint32_t n_output,
                                                                 int32_t *__restrict__ out, uint64_t batch) {
    __shared__ __attribute__((aligned(16))) int8_t s_act[LW_IMGS][LW_MAXIN + 16];
    __shared__ __attribute__((aligned(16))) uint32_t s_w[4][LW_MAXIN / 4 + 4];   // 4 neuron rows, <= 1 KiB each
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const uint32_t row = blockIdx.x * 4u + (uint32_t)wave;
    const int fb = codec_field_bits(bpw);
    const uint32_t per_word = fb ? 32u / (uint32_t)fb : 10u;
    // elements of the packed row: 32-bit words, or 16-bit chunks for ternary
    const uint32_t row_elems = bpw == 64 ? n_input / 10u : (fb ? (n_input + per_word - 1u) / per_word : 0u);
    const bool known = bpw == 64 || fb != 0;

    // stage this wave's packed weight row
    if (row < n_output && known) {
        if (bpw == 64) {
            const uint16_t *src = (const uint16_t *)packed + (size_t)row * row_elems;
            for (uint32_t i = lane; i < row_elems; i += 64) s_w[wave][i] = src[i];
        } else {
            const uint32_t *src = (const uint32_t *)packed + (size_t)row * row_elems;
            for (uint32_t i = lane; i < row_elems; i += 64) s_w[wave][i] = src[i];
        }
    }

    for (uint64_t base = (uint64_t)blockIdx.y * LW_IMGS; base < batch; base += (uint64_t)gridDim.y * LW_IMGS) {
        __syncthreads();
        // stage up to LW_IMGS activation vectors (only bytes < n_input that exist: ternary pads are never read)
        const uint32_t nimg = (uint32_t)((batch - base) < LW_IMGS ? (batch - base) : LW_IMGS);
        for (uint32_t i = threadIdx.x; i < nimg * act_stride; i += blockDim.x) {
            uint32_t im = i / act_stride, k = i % act_stride;
            if (k < LW_MAXIN) s_act[im][k] = act[(base + im) * act_stride + k];
        }
        __syncthreads();
        if (row >= n_output) continue;
        int32_t sum[LW_IMGS];
#pragma unroll
        for (int im = 0; im < LW_IMGS; im++) sum[im] = 0;
        if (known) {
            for (uint32_t e = lane; e < row_elems; e += 64) {
                uint32_t word = s_w[wave][e];
                for (uint32_t f = 0; f < per_word; f++) {
                    uint32_t k = e * per_word + f;
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
                    if (w != 0 && k < act_stride) {
#pragma unroll
                        for (int im = 0; im < LW_IMGS; im++) sum[im] += w * (int)s_act[im][k];
                    }
                }
            }
        }
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
    if (n_input > LW_MAXIN + 15 || act_stride > LW_MAXIN) return hipErrorInvalidValue;
    uint64_t gy = (batch + LW_IMGS - 1) / LW_IMGS;
    if (gy > 8192) gy = 8192;
    fc_layer_bitserial_kernel<<<dim3((n_output + 3u) / 4u, (unsigned)gy), dim3(256), 0, s>>>(act, act_stride, packed, bpw,
                                                                                            n_input, n_output, out, batch);
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
                int q = x[t] < 0 ? 0 : min((x[t] + rnd) >> sh, 127);
                dst[i] = (int8_t)q;
            }
        }
        if (argmax && lane == 0) argmax[v] = bi;
    }
}

hipError_t bnmk_relunorm(const int32_t *in, uint32_t n, int8_t *out, uint32_t out_stride, uint32_t *argmax,
                         uint64_t batch, hipStream_t s) {
    if (!batch || !n) return hipSuccess;
    if (n > LW_MAXIN) return hipErrorInvalidValue;
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

hipError_t bnmk_conv33(const int32_t *in, const int8_t *w, uint32_t xy, uint32_t n_shift, int32_t *out, hipStream_t s) {
    if (xy < 3 || xy > 64) return hipErrorInvalidValue;
    conv33_kernel<<<dim3(1), dim3(256), 0, s>>>(in, w, xy, n_shift, out);
    return hipGetLastError();
}
hipError_t bnmk_maxpool22(const int32_t *in, uint32_t xy, int32_t *out, hipStream_t s) {
    if (xy < 2 || xy > 64) return hipErrorInvalidValue;
    maxpool22_kernel<<<dim3(1), dim3(256), 0, s>>>(in, xy, out);
    return hipGetLastError();
}

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
template <bool FUSE>
__global__ __launch_bounds__(256) void cnn_front_kernel(const int8_t *__restrict__ images, uint64_t n,
                                                        const int8_t *__restrict__ w1, const int8_t *__restrict__ w2,
                                                        const int8_t *__restrict__ w3, uint32_t C, uint32_t c0,
                                                        uint32_t n_shift, int8_t *__restrict__ acts,
                                                        int32_t *__restrict__ feat) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave0 = (uint64_t)blockIdx.x * 4u + (uint64_t)__builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint64_t nwaves = (uint64_t)gridDim.x * 4u;
    const uint32_t c = c0 + (uint32_t)lane;
    const bool live = c < C;

    // conv1 weights as three packed rows (w0,w1,w2,0) for v_dot4_i32_i8; conv2/conv3 weights as 24-bit mad operands
    int wk[3], k2[9], k3[9];
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
                if constexpr (y1 >= 2) {
                    constexpr int y2 = y1 - 2;
                    static_for<0, 12>([&](auto X) {
                        constexpr int x = decltype(X)::value;
                        int s = mul24(k2[0], r1[y2 % 3][x]);
                        static_for<1, 9>([&](auto T) {
                            constexpr int t = decltype(T)::value;
                            s = mad24(k2[t], r1[(y2 + t / 3) % 3][x + t % 3], s);
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
                *(uint32_t *)(acts + img * (4ull * C) + 4ull * c) = d;
            }
        }
    }
}

// acts: int8 [n][4C] (always produced).  feat: int32 [n][4C]; optional when C <= 64, REQUIRED scratch when
// C > 64 (several channel groups: ReLUNorm then runs as its own kernel over the complete vector).
hipError_t bnmk_cnn_front(const int8_t *images, uint64_t n, const int8_t *w1, const int8_t *w2, const int8_t *w3,
                          uint32_t C, uint32_t n_shift, int8_t *acts, int32_t *feat, hipStream_t s) {
    if (!n) return hipSuccess;
    if (C == 0 || C > 256 || n_shift < 4 || n_shift > 31) return hipErrorInvalidValue;
    uint64_t blocks = (n + 3) / 4;
    uint64_t cap = (uint64_t)num_cus() * 4ull;
    if (blocks > cap) blocks = cap;
    dim3 g((unsigned)blocks), b(256);
    if (C <= 64) {
        cnn_front_kernel<true><<<g, b, 0, s>>>(images, n, w1, w2, w3, C, 0, n_shift, acts, feat);
        return hipGetLastError();
    }
    if (!feat) return hipErrorInvalidValue;
    for (uint32_t c0 = 0; c0 < C; c0 += 64) {
        cnn_front_kernel<false><<<g, b, 0, s>>>(images, n, w1, w2, w3, C, c0, n_shift, acts, feat);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return bnmk_relunorm(feat, 4u * C, acts, 4u * C, nullptr, n, s);
}

// =================================================================================================
// Ternary whole-model kernel, ALU only (BASELINE config 3: "bit-unpack / sign-accumulate path, no MFMA").
// Mapping: one lane = one image.  The trits were unpacked once per model to int8 {-1,0,+1} rows by
// unpack_rows_kernel; a neuron's row is wave-uniform, so 4 trits at a time arrive as a scalar operand and
// v_dot4_i32_i8 adds/subtracts/skips 4 activations per issue.  The ReLUNorm maximum is per lane (no
// cross-lane traffic at all); layer outputs are parked in a lane-private LDS column between the two
// ReLUNorm passes because VGPRs cannot be indexed by the (runtime) neuron loop.
// =================================================================================================
// Hidden-layer sums are parked AFTER ReLU as uint16 pairs: max(sum, 0) <= 256*128 = 32768 fits 16 bits, negative
// sums become 0 exactly as ReLUNorm would make them, and the maximum is unchanged (an all-negative vector has
// maximum 0 and every output 0 either way).  Halves the LDS column: 12 KiB per wave -> 3 waves per SIMD.
template <int H>
BNM_DEVICE void tern_norm_pack(const uint32_t *col, int mx, int (&act)[H / 4]) {
    mx = max(mx, 0);
    uint32_t t = (uint32_t)mx >> 7;
    int sh = t ? 32 - __builtin_clz(t) : 0;
    int rnd = (1 << sh) >> 1;
#pragma unroll
    for (int q = 0; q < H / 4; q++) {
        uint32_t lo = col[(2 * q) * 64], hi = col[(2 * q + 1) * 64];   // neurons 4q,4q+1 | 4q+2,4q+3
        int v0 = min((int)((lo & 0xFFFFu) + rnd) >> sh, 127), v1 = min((int)((lo >> 16) + rnd) >> sh, 127);
        int v2 = min((int)((hi & 0xFFFFu) + rnd) >> sh, 127), v3 = min((int)((hi >> 16) + rnd) >> sh, 127);
        act[q] = v0 | (v1 << 8) | (v2 << 16) | (v3 << 24);
    }
}

// out rows [0,H) of one layer: acc = sum_q dot4(act[q], W[n][q]);  returns the running max
template <int KQ, int H>
BNM_DEVICE int tern_layer(const int (&act)[KQ], const int8_t *__restrict__ rows, uint32_t stride, uint32_t *col) {
    int mx = 0;
#pragma unroll 1
    for (int nn = 0; nn < H; nn += 4) {
        const int *__restrict__ w0 = (const int *)(rows + (size_t)(nn + 0) * stride);
        const int *__restrict__ w1 = (const int *)(rows + (size_t)(nn + 1) * stride);
        const int *__restrict__ w2 = (const int *)(rows + (size_t)(nn + 2) * stride);
        const int *__restrict__ w3 = (const int *)(rows + (size_t)(nn + 3) * stride);
        int a0 = 0, a1 = 0, a2 = 0, a3 = 0;
#pragma unroll
        for (int q = 0; q < KQ; q++) {
            a0 = __builtin_amdgcn_sdot4(act[q], w0[q], a0, false);
            a1 = __builtin_amdgcn_sdot4(act[q], w1[q], a1, false);
            a2 = __builtin_amdgcn_sdot4(act[q], w2[q], a2, false);
            a3 = __builtin_amdgcn_sdot4(act[q], w3[q], a3, false);
        }
        a0 = max(a0, 0); a1 = max(a1, 0); a2 = max(a2, 0); a3 = max(a3, 0);
        col[(nn / 2 + 0) * 64] = (uint32_t)a0 | ((uint32_t)a1 << 16);
        col[(nn / 2 + 1) * 64] = (uint32_t)a2 | ((uint32_t)a3 << 16);
        mx = max(max(mx, max(a0, a1)), max(a2, a3));
    }
    return mx;
}

template <int H1, int H2, int H3>
__global__ __launch_bounds__(64) void ternary_alu_kernel(const int8_t *__restrict__ images, uint64_t n,
                                                         const int8_t *__restrict__ r1, const int8_t *__restrict__ r2,
                                                         const int8_t *__restrict__ r3, const int8_t *__restrict__ r4,
                                                         uint32_t s1, uint32_t s2, uint32_t s3, uint32_t s4,
                                                         uint32_t n_classes, uint32_t *__restrict__ cls_out,
                                                         int32_t *__restrict__ logits_out) {
    constexpr int HM = H1 > H2 ? (H1 > H3 ? H1 : H3) : (H2 > H3 ? H2 : H3);
    __shared__ uint32_t s_col[HM / 2 * 64];
    const int lane = threadIdx.x;
    uint32_t *col = s_col + lane;
    for (uint64_t base = (uint64_t)blockIdx.x * 64ull; base < n; base += (uint64_t)gridDim.x * 64ull) {
        uint64_t img = base + (uint64_t)lane;
        const bool live = img < n;
        if (!live) img = n - 1ull;
        int x0[64];
        const i32x4 *p = (const i32x4 *)(images + img * 256ull);
#pragma unroll
        for (int q = 0; q < 16; q++) {
            i32x4 v = p[q];
            x0[4 * q + 0] = v[0]; x0[4 * q + 1] = v[1]; x0[4 * q + 2] = v[2]; x0[4 * q + 3] = v[3];
        }
        int a1[H1 / 4], a2[H2 / 4], a3[H3 / 4];
        int mx = tern_layer<64, H1>(x0, r1, s1, col);
        tern_norm_pack<H1>(col, mx, a1);
        mx = tern_layer<H1 / 4, H2>(a1, r2, s2, col);
        tern_norm_pack<H2>(col, mx, a2);
        mx = tern_layer<H2 / 4, H3>(a2, r3, s3, col);
        tern_norm_pack<H3>(col, mx, a3);
        // output layer: first strict maximum (ReLUNorm's return value)
        int bv = -INT_MAX;
        uint32_t bi = 255;
        for (uint32_t c = 0; c < n_classes; c++) {
            const int *__restrict__ w = (const int *)(r4 + (size_t)c * s4);
            int acc = 0;
#pragma unroll
            for (int q = 0; q < H3 / 4; q++) acc = __builtin_amdgcn_sdot4(a3[q], w[q], acc, false);
            if (acc > bv) { bv = acc; bi = c; }
            if (logits_out && live) logits_out[img * n_classes + c] = acc;
        }
        if (live) cls_out[img] = bi;
    }
}

hipError_t bnmk_ternary_alu(const BnmTernArgs &a, int grid_blocks, hipStream_t s) {
    if (!a.n) return hipSuccess;
    if (a.n_layers != 4 || a.n_in[0] != 256 || a.n_out[0] != 96 || a.n_out[1] != 96 || a.n_out[2] != 96 ||
        a.n_in[1] != 96 || a.n_in[2] != 96 || a.n_in[3] != 96)
        return hipErrorInvalidValue;
    uint64_t want = (a.n + 63ull) / 64ull;
    uint64_t cap = grid_blocks > 0 ? (uint64_t)grid_blocks : (uint64_t)num_cus() * 12ull;
    unsigned blocks = (unsigned)(want < cap ? want : cap);
    ternary_alu_kernel<96, 96, 96><<<dim3(blocks), dim3(64), 0, s>>>(a.images, a.n, a.rows[0], a.rows[1], a.rows[2],
                                                                       a.rows[3], a.stride[0], a.stride[1], a.stride[2],
                                                                       a.stride[3], a.n_out[3], a.cls, a.logits);
    return hipGetLastError();
}

// =================================================================================================
// Diagnostics: what the image stream alone costs.  mode 0: plain 16 B/lane global loads, grid-stride;
// mode 1 / 2: the fused kernel's own tile loop (variant LDSDMA / LDSDMA2) with the math replaced by one ds_read
// per tile.  Both write one dword per 32 images so the result cannot be optimised away.  Used by
// profiles/stream_ceiling.py to put the achieved GB/s of the real kernel next to the practical read ceiling.
// =================================================================================================
__global__ __launch_bounds__(256) void diag_stream_plain_kernel(const u32x4 *__restrict__ src, uint64_t n16,
                                                                uint32_t *__restrict__ out) {
    uint32_t acc = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        u32x4 a = __builtin_nontemporal_load(src + i), b = __builtin_nontemporal_load(src + i + stride);
        u32x4 c = __builtin_nontemporal_load(src + i + 2 * stride), d = __builtin_nontemporal_load(src + i + 3 * stride);
        acc ^= a[0] ^ a[1] ^ a[2] ^ a[3] ^ b[0] ^ b[1] ^ b[2] ^ b[3] ^ c[0] ^ c[1] ^ c[2] ^ c[3] ^ d[0] ^ d[1] ^ d[2] ^ d[3];
    }
    for (; i < n16; i += stride) {
        u32x4 a = __builtin_nontemporal_load(src + i);
        acc ^= a[0] ^ a[1] ^ a[2] ^ a[3];
    }
    if (acc == 0x12345678u) out[0] = acc;   // practically never: keeps the loads alive without a store stream
}

template <bool TWO>
__global__ __launch_bounds__(64 * FUSED_WPB, 2) void diag_stream_tiles_kernel(const int8_t *__restrict__ images, uint64_t n,
                                                                              uint32_t *__restrict__ out) {
    __shared__ __attribute__((aligned(1024))) char smem[FUSED_WPB * 2 * FUSED_TILE_BYTES];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t voff[4];
#pragma unroll
    for (int u = 0; u < 4; u++) voff[u] = (uint32_t)(lane >> 4) * 256u + 16u * ((uint32_t)(lane & 15) ^ (uint32_t)(lane >> 4) ^ (4u * u));
    const uint32_t lds_wave = (uint32_t)(size_t)(__attribute__((address_space(3))) char *)smem + (uint32_t)wave * 2u * FUSED_TILE_BYTES;
    const uint64_t n_tiles = n >> 5;   // whole tiles only
    const uint64_t stride = (uint64_t)gridDim.x * FUSED_WPB;
    uint64_t tile = (uint64_t)blockIdx.x * FUSED_WPB + wave;
    auto dma = [&](uint64_t t, int par) {
        const int8_t *base = images + t * (uint64_t)FUSED_TILE_BYTES;
        lds_dma_tile8<TWO, TWO>(lds_wave + (uint32_t)par * FUSED_TILE_BYTES, base, base + 1024, base + 2048, base + 3072,
                                base + 4096, base + 5120, base + 6144, base + 7168, voff[0], voff[1], voff[2], voff[3], voff[0],
                                voff[1], voff[2], voff[3]);
    };
    int par = 0;
    if (tile < n_tiles) dma(tile, 0);
    if (TWO && tile + stride < n_tiles) dma(tile + stride, 1);
    for (; tile < n_tiles; tile += stride) {
        const uint64_t next = tile + stride;
        if constexpr (!TWO) {
            if (next < n_tiles) { dma(next, par ^ 1); bnm_wait_vmcnt<8>(); } else { bnm_wait_vmcnt<0>(); }
        } else {
            if (next < n_tiles) bnm_wait_vmcnt<8>(); else bnm_wait_vmcnt<0>();
        }
        uint32_t v = *(const uint32_t *)(smem + (uint32_t)wave * 2u * FUSED_TILE_BYTES + (uint32_t)par * FUSED_TILE_BYTES + 128u * lane);
        if (TWO && next + stride < n_tiles) dma(next + stride, par);
        if (lane < 32) out[(tile << 5) + lane] = v;
        par ^= 1;
    }
}

hipError_t bnmk_diag_stream(const int8_t *images, uint64_t n, int mode, int grid_blocks, uint32_t *out, hipStream_t s) {
    if (!n) return hipSuccess;
    int cus = num_cus();
    if (mode == 0) {
        unsigned blocks = grid_blocks > 0 ? (unsigned)grid_blocks : (unsigned)cus * 8u;
        diag_stream_plain_kernel<<<dim3(blocks), dim3(256), 0, s>>>((const u32x4 *)images, n * 16ull, out);
    } else {
        unsigned blocks = grid_blocks > 0 ? (unsigned)grid_blocks : (unsigned)cus * 2u;
        if (mode == 1) diag_stream_tiles_kernel<false><<<dim3(blocks), dim3(256), 0, s>>>(images, n, out);
        else diag_stream_tiles_kernel<true><<<dim3(blocks), dim3(256), 0, s>>>(images, n, out);
    }
    return hipGetLastError();
}

// =================================================================================================
// Input quantisation: the step immediately before the path, which the reference does in
// Python for every image (test_inference.py:140-141, same formula BitNetMCU.py:435-436):
//     scale = 127.0 / max(max|x|, 1e-5);  q = clip(round_half_even(x * scale), -128, 127)   all in float32.
// One wavefront per image (256 floats = one float4 per lane); IEEE float32 divide/multiply and v_rndne_f32, so the
// result is bit-identical to numpy's float32 arithmetic.
// =================================================================================================
__global__ __launch_bounds__(256) void quantize_input_kernel(const float *__restrict__ x, uint64_t n, int8_t *__restrict__ out) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63;
    for (uint64_t img = (uint64_t)blockIdx.x * 4u + (threadIdx.x >> 6); img < n; img += (uint64_t)gridDim.x * 4u) {
        f32x4 v = *(const f32x4 *)(x + img * 256ull + 4u * lane);
        float m = fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3])));
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
        const float scale = __fdiv_rn(127.0f, fmaxf(m, 1e-5f));
        uint32_t d = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) {
            float r = rintf(__fmul_rn(v[b], scale));
            r = fminf(fmaxf(r, -128.0f), 127.0f);
            d |= (uint32_t)(uint8_t)(int8_t)(int)r << (8 * b);
        }
        *(uint32_t *)(out + img * 256ull + 4u * lane) = d;
    }
}

hipError_t bnmk_quantize_input(const float *x, uint64_t n, int8_t *out, hipStream_t s) {
    if (!n) return hipSuccess;
    uint64_t blocks = (n + 3) / 4;
    if (blocks > 8192) blocks = 8192;
    quantize_input_kernel<<<dim3((unsigned)blocks), dim3(256), 0, s>>>(x, n, out);
    return hipGetLastError();
}
