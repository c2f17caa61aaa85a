// Ternary whole-model kernel, ALU only (no MFMA).
// gfx950 (CDNA4 / MI355X) only; see DESIGN.md for layouts and rooflines.  Reference semantics:
// BitNetMCU_inference.c:23-72 (ReLUNorm), :88-208 (processfclayer), :238-277 (conv), :300-322 (pool);
// schedule BitNetMCU_MNIST_dll.c:48-121.
#include "bnm_device.hpp"

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
    uint64_t cap = grid_blocks > 0 ? (uint64_t)grid_blocks : (uint64_t)bnm_num_cus() * 12ull;
    unsigned blocks = (unsigned)(want < cap ? want : cap);
    ternary_alu_kernel<96, 96, 96><<<dim3(blocks), dim3(64), 0, s>>>(a.images, a.n, a.rows[0], a.rows[1], a.rows[2],
                                                                       a.rows[3], a.stride[0], a.stride[1], a.stride[2],
                                                                       a.stride[3], a.n_out[3], a.cls, a.logits);
    return hipGetLastError();
}

