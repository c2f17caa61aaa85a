// Ternary whole-model kernels, ALU only (no MFMA): weight-stream builder, shape tables and dispatch.
// gfx950 (CDNA4 / MI355X) only; the kernels live in bnm_ternary_kernel.hpp.
#include "bnm_ternary_kernel.hpp"

// ---- weight stream of the streamed kernel -----------------------------------------------------------------------
// per layer: for each quad of neurons, for each slice of 8 activation dwords: [4 neurons][8 dwords]; after the last layer
// one more chunk = a copy of the first (the prefetch that follows the last class quad).  Rows past n_out are zero.
// A layer's K is padded to whole chunks (kq dwords; kq_real of them exist in the rows - 112 inputs = 28 dwords -> 32).
__global__ void tern_stream_kernel(const int8_t *__restrict__ rows, uint32_t stride, uint32_t n_out, uint32_t kq,
                                   uint32_t kq_real, uint32_t n_quads, int *__restrict__ out) {
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t nc = kq / 8u;
    if (idx >= n_quads * nc * 32u) return;
    const uint32_t chunk = idx / 32u, d = idx % 32u;
    const uint32_t neuron = 4u * (chunk / nc) + d / 8u, k = 8u * (chunk % nc) + d % 8u;
    out[idx] = (neuron < n_out && k < kq_real) ? ((const int *)(rows + (size_t)neuron * stride))[k] : 0;
}

namespace {
uint32_t tern_kq(uint32_t n_in) { return (n_in / 4u + 7u) / 8u * 8u; }
}  // namespace

uint32_t bnmk_ternary_stream_dwords(const uint32_t n_out[4]) {
    uint32_t dw = 0, kq = 64;
    for (int i = 0; i < 4; i++) {
        dw += ((n_out[i] + 3u) / 4u) * (kq / 8u) * 32u;
        kq = tern_kq(n_out[i]);
    }
    return dw + 32u;
}

hipError_t bnmk_ternary_stream_build(const BnmTernArgs &a, int *d_stream, hipStream_t s) {
    uint32_t off = 0, kq = 64, kq_real = 64;
    for (int i = 0; i < 4; i++) {
        const uint32_t nq = (a.n_out[i] + 3u) / 4u, dw = nq * (kq / 8u) * 32u;
        tern_stream_kernel<<<dim3((dw + 255u) / 256u), dim3(256), 0, s>>>(a.rows[i], a.stride[i], a.n_out[i], kq, kq_real, nq,
                                                                        d_stream + off);
        off += dw;
        kq = tern_kq(a.n_out[i]);
        kq_real = a.n_out[i] / 4u;
    }
    hipError_t e = hipMemcpyAsync(d_stream + off, d_stream, 128, hipMemcpyDeviceToDevice, s);
    return e != hipSuccess ? e : hipGetLastError();
}

// ---- shapes ------------------------------------------------------------------------------------------------------
// The kernels are instantiated per hidden-width triple (a lane keeps a layer's packed activations in H/4 registers and the
// neuron loop is unrolled over them).  The streamed kernel with one image per lane (variant 1, weights through scalar registers)
// exists for the whole family H1, H2 in {32, 64, 96, 128} x H3 = 16, 32 .. 128 (bnm_ternary_s*.hip) - among them the widths the
// reference documents for ternary models, 96-96-96 (BASELINE configs[2]) and the 12 KB family's 128-128-112
// (docs/documentation.md:169-183); two images per lane (variant 2, the default where it exists) for 96-96-96; round 1's plain ALU
// kernel (variant 0, kept for A/B measurements) for the four shapes of its table.  Ternary layers declare a padded input count
// (a multiple of 10, exportquant.py:132-137); the kernels read the REAL inputs only.
namespace {
typedef void (*tern_fn)(const int8_t *, uint64_t, const int8_t *, const int8_t *, const int8_t *, const int8_t *, uint32_t, uint32_t,
                        uint32_t, uint32_t, uint32_t, uint32_t *, int32_t *);
struct TernShape {
    uint32_t h[3];
    tern_fn fn;                 // variant 0: the plain ALU kernel
};
const TernShape kTernShapes[] = {
    {{96, 96, 96}, ternary_alu_kernel<96, 96, 96>},
    {{128, 128, 112}, ternary_alu_kernel<128, 128, 112>},
    {{64, 64, 64}, ternary_alu_kernel<64, 64, 64>},
    {{128, 128, 128}, ternary_alu_kernel<128, 128, 128>},
};
const TernShape *find_tern(const uint32_t n_out[4]) {
    for (const TernShape &t : kTernShapes)
        if (t.h[0] == n_out[0] && t.h[1] == n_out[1] && t.h[2] == n_out[2]) return &t;
    return nullptr;
}
tern_stream_fn find_stream1(const uint32_t n_out[4]) {
    switch (n_out[0]) {
        case 32: return bnmk_tern_stream1_h32(n_out[1], n_out[2]);
        case 64: return bnmk_tern_stream1_h64(n_out[1], n_out[2]);
        case 96: return bnmk_tern_stream1_h96(n_out[1], n_out[2]);
        case 128: return bnmk_tern_stream1_h128(n_out[1], n_out[2]);
    }
    return nullptr;
}
}  // namespace

bool bnmk_ternary_alu_supported(const uint32_t n_in[4], const uint32_t n_out[4]) {
    return n_in[0] == 256 && n_in[1] == n_out[0] && n_in[2] == n_out[1] && n_in[3] == n_out[2] && find_stream1(n_out) != nullptr;
}
// images_per_lane 1: the whole family; 2 (the default where it exists): 96-96-96; 0: round 1's plain kernel, four shapes
bool bnmk_ternary_stream_supported(const uint32_t n_out[4], int images_per_lane) {
    if (images_per_lane == 2) return n_out[0] == 96 && n_out[1] == 96 && n_out[2] == 96;
    if (images_per_lane == 0) return find_tern(n_out) != nullptr;
    return images_per_lane == 1 && find_stream1(n_out) != nullptr;
}

hipError_t bnmk_ternary_alu(const BnmTernArgs &a, int grid_blocks, hipStream_t s) {
    if (!a.n) return hipSuccess;
    if (a.n_layers != 4 || !bnmk_ternary_alu_supported(a.n_in, a.n_out)) return hipErrorInvalidValue;
    const bool stream = a.variant != 0;
    const int G = a.variant == 2 ? 2 : 1;
    if (!bnmk_ternary_stream_supported(a.n_out, stream ? G : 0)) return hipErrorInvalidValue;
    const uint64_t per = 64ull * (uint64_t)G;
    const uint64_t want = (a.n + per - 1ull) / per;
    // resident waves per CU: the streamed kernels 12 or 8 (one image per lane: tern_wpe waves per SIMD) / 8 (two per lane); the plain
    // kernel's LDS column is max(H) * 128 bytes per wave
    uint32_t hm = a.n_out[0] > a.n_out[1] ? a.n_out[0] : a.n_out[1];
    hm = hm > a.n_out[2] ? hm : a.n_out[2];
    const uint64_t wpc = stream ? (G == 2 ? 8ull : 4ull * (uint64_t)tern_wpe((int)a.n_out[0], (int)a.n_out[1], (int)a.n_out[2])) : (uint64_t)((160u * 1024u) / (hm * 128u) < 12u ? (160u * 1024u) / (hm * 128u) : 12u);
    const uint64_t cap = grid_blocks > 0 ? (uint64_t)grid_blocks : (uint64_t)bnm_num_cus() * wpc;
    const unsigned blocks = (unsigned)(want < cap ? want : cap);
    if (!stream) {
        find_tern(a.n_out)->fn<<<dim3(blocks), dim3(64), 0, s>>>(a.images, a.n, a.rows[0], a.rows[1], a.rows[2], a.rows[3], a.stride[0],
                                                                 a.stride[1], a.stride[2], a.stride[3], a.n_out[3], a.cls, a.logits);
    } else {
        if (!a.wstream || want >= (1ull << 32)) return hipErrorInvalidValue;
        if (G == 2)
            ternary_stream_kernel<2, 96, 96, 96, 20><<<dim3(blocks), dim3(64), 0, s>>>(a.images, a.n, a.wstream, a.n_out[3],
                                                                                         a.cls, a.logits, a.counter);
        else
            find_stream1(a.n_out)<<<dim3(blocks), dim3(64), 0, s>>>(a.images, a.n, a.wstream, a.n_out[3], a.cls, a.logits, a.counter);
    }
    return hipGetLastError();
}
