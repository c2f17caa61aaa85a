// Ternary whole-model kernels, ALU only (no MFMA): the device code, shared by bnm_ternary.hip (dispatch, the plain kernel and the
// two-images-per-lane kernel) and bnm_ternary_s{32,64,96,128}.hip (the streamed one-image-per-lane kernel for every hidden-width
// triple of its family, one translation unit per first width so that they compile side by side).
#pragma once
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

// =================================================================================================
// Round 2: the streamed kernel.  Round 1's kernel (above) left the VALU 70 % busy: hipcc waits for every batch of
// scalar weight loads right after issuing it (s_waitcnt lgkmcnt(0) ahead of the first use), and the 44 KB of trit rows
// cycle through a 16 KB scalar cache, so each batch is an L2 round trip a wave sits out (SQ_WAIT_ANY 0.40 of wave time).
// Here
//  * the model's trits are laid out once per model as ONE linear stream of 128-byte chunks in exactly the order the
//    kernel consumes them (tern_stream_kernel): chunk = 4 neurons x 8 activation dwords = 32 SGPRs = 2 cache lines;
//  * the loads are inline asm hipcc neither counts nor waits for (guide 5.7): chunk i+1 is issued into the second SGPR
//    buffer, chunk i is consumed (32 x G v_dot4), then ONE s_waitcnt lgkmcnt(0) retires chunk i+1 (scalar loads return
//    out of order, so only "all" can be waited for) - a chunk's latency hides under the previous chunk's dots;
//  * a lane carries G images (G = 2 by default: 128 image VGPRs), so every weight dword feeds G dots: half the scalar
//    traffic per image and twice the VALU work behind each load;
//  * the next images are requested right after layer 1 (their registers are dead from there on) and land under layers 2-4;
//  * layer sums are parked as SATURATED int16 pairs (v_cvt_pk_i16_i32; the only sum that saturates is +32768, which
//    forces shift 9 and rounds to 64 either way) and ReLU / rounding add / shift / clip run on packed 16-bit pairs:
//    3.25 VALU per hidden value instead of 6.5.  With G = 2 a wave's LDS share (20 KiB at 2 waves per SIMD) holds 80 of
//    a layer's 96 sums per image; the last 16 stay in registers (those neuron quads are peeled, all indices constant).
// =================================================================================================
typedef int sx16 __attribute__((ext_vector_type(16)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));

struct TernW {
    sx16 lo, hi;   // one chunk: dword d = 8 * neuron-in-quad + activation dword in the K-slice
};
// The two chunk buffers live in FIXED scalar registers - A = s[36:67], B = s[68:99] - named in the constraints of the issue
// and of the wait.  With ordinary "s" operands hipcc is free to give the issue's output and the wait's tied operand
// different registers and to copy one into the other AHEAD of the wait, i.e. while the load is still in flight (it did, in
// the class loop, as soon as the kernel grew a little: wrong class ids).  Pinned, the value never moves between the two.
// issue (not wait for) the two cache lines of the chunk at p + 128 * OFF bytes
template <bool ISA, int OFF>
BNM_DEVICE void tern_issue(TernW &w, const int *p) {
    if constexpr (ISA)
        asm volatile("s_load_dwordx16 %0, %2, %3\n\ts_load_dwordx16 %1, %2, %4"
                     : "=&{s[36:51]}"(w.lo), "=&{s[52:67]}"(w.hi) : "s"(p), "i"(OFF * 128), "i"(OFF * 128 + 64));
    else
        asm volatile("s_load_dwordx16 %0, %2, %3\n\ts_load_dwordx16 %1, %2, %4"
                     : "=&{s[68:83]}"(w.lo), "=&{s[84:99]}"(w.hi) : "s"(p), "i"(OFF * 128), "i"(OFF * 128 + 64));
}
template <bool ISA>
BNM_DEVICE void tern_land(TernW &w) {
    if constexpr (ISA) asm volatile("s_waitcnt lgkmcnt(0)" : "+{s[36:51]}"(w.lo), "+{s[52:67]}"(w.hi));
    else asm volatile("s_waitcnt lgkmcnt(0)" : "+{s[68:83]}"(w.lo), "+{s[84:99]}"(w.hi));
}
// the same, also naming the running sums: pins the chunk's dots between the issue and this wait (left free, hipcc sinks
// a class quad's dots into the `class < n_classes` blocks behind all the waits and keeps every chunk live)
template <bool ISA, int G>
BNM_DEVICE void tern_land(TernW &w, int (&acc)[4][G]) {
    static_assert(G == 1 || G == 2, "one or two images per lane");
    if constexpr (G == 1) {
        if constexpr (ISA)
            asm volatile("s_waitcnt lgkmcnt(0)" : "+{s[36:51]}"(w.lo), "+{s[52:67]}"(w.hi), "+v"(acc[0][0]), "+v"(acc[1][0]), "+v"(acc[2][0]), "+v"(acc[3][0]));
        else
            asm volatile("s_waitcnt lgkmcnt(0)" : "+{s[68:83]}"(w.lo), "+{s[84:99]}"(w.hi), "+v"(acc[0][0]), "+v"(acc[1][0]), "+v"(acc[2][0]), "+v"(acc[3][0]));
    } else {
        if constexpr (ISA)
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+{s[36:51]}"(w.lo), "+{s[52:67]}"(w.hi), "+v"(acc[0][0]), "+v"(acc[1][0]), "+v"(acc[2][0]), "+v"(acc[3][0]),
                           "+v"(acc[0][1]), "+v"(acc[1][1]), "+v"(acc[2][1]), "+v"(acc[3][1]));
        else
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+{s[68:83]}"(w.lo), "+{s[84:99]}"(w.hi), "+v"(acc[0][0]), "+v"(acc[1][0]), "+v"(acc[2][0]), "+v"(acc[3][0]),
                           "+v"(acc[0][1]), "+v"(acc[1][1]), "+v"(acc[2][1]), "+v"(acc[3][1]));
    }
}

template <int D>
BNM_DEVICE int tern_wd(const TernW &w) {
    if constexpr (D < 16) return w.lo[D]; else return w.hi[D - 16];
}

// 32 x G dots of one chunk: neuron i of the quad, activation dwords X0 .. X0+7
template <int G, int NX, int X0>
BNM_DEVICE void tern_chunk(const int (&x)[G][NX], const TernW &w, int (&acc)[4][G]) {
    static_for<0, 8>([&](auto J) {
        constexpr int j = decltype(J)::value;
        static_for<0, 4>([&](auto I) {
            constexpr int i = decltype(I)::value;
#pragma unroll
            for (int g = 0; g < G; g++) acc[i][g] = __builtin_amdgcn_sdot4(x[g][X0 + j], tern_wd<8 * i + j>(w), acc[i][g], false);
        });
    });
}

// One neuron quad over KQ activation dwords = KQ/8 chunks.  On entry `a` holds the quad's first chunk (landed); on exit
// the buffer named by the return parity holds the NEXT chunk of the stream (landed): chunks alternate a, b, a, ...
// `p` points at the quad's first chunk; the chunk after the quad's last one is simply the next one in the stream.
template <int G, int NX, int KQ, bool AFIRST>
BNM_DEVICE void tern_quad(const int (&x)[G][NX], TernW &a, TernW &b, const int *p, int (&acc)[4][G]) {
    constexpr int NC = KQ / 8;
    static_for<0, NC>([&](auto C) {
        constexpr int c = decltype(C)::value;
        constexpr bool cur_a = ((c & 1) == 0) == AFIRST;
        TernW &cur = cur_a ? a : b;
        TernW &nxt = cur_a ? b : a;
        tern_issue<!cur_a, c + 1>(nxt, p);        // `a` is buffer A, `b` is buffer B: the next chunk goes to the one not in use
        __builtin_amdgcn_sched_barrier(0);
        tern_chunk<G, NX, 8 * c>(x, cur, acc);
        __builtin_amdgcn_sched_barrier(0);
        tern_land<!cur_a, G>(nxt, acc);
    });
}

// park one quad's sums of group g: saturated int16 pairs; running maximum from the 32-bit sums
BNM_DEVICE void tern_pairs(const int (&s)[4], int &mx, uint32_t &p01, uint32_t &p23) {
    s16x2 a = __builtin_amdgcn_cvt_pk_i16(s[0], s[1]), b = __builtin_amdgcn_cvt_pk_i16(s[2], s[3]);
    p01 = __builtin_bit_cast(uint32_t, a);
    p23 = __builtin_bit_cast(uint32_t, b);
    mx = max(mx, max(s[0], s[1]));
    mx = max(mx, max(s[2], s[3]));
}

// ReLUNorm (BitNetMCU_inference.c:23-72) of four parked sums -> four packed int8 activations
BNM_DEVICE int tern_norm4(uint32_t p01, uint32_t p23, uint32_t rnd2, uint32_t sh2) {
    const s16x2 z = {0, 0};
    const u16x2 c127 = {127, 127};
    s16x2 a = __builtin_bit_cast(s16x2, p01), b = __builtin_bit_cast(s16x2, p23);
    a = __builtin_elementwise_max(a, z);
    b = __builtin_elementwise_max(b, z);
    u16x2 ua = __builtin_bit_cast(u16x2, a) + __builtin_bit_cast(u16x2, rnd2);
    u16x2 ub = __builtin_bit_cast(u16x2, b) + __builtin_bit_cast(u16x2, rnd2);
    ua = ua >> __builtin_bit_cast(u16x2, sh2);
    ub = ub >> __builtin_bit_cast(u16x2, sh2);
    ua = __builtin_elementwise_min(ua, c127);
    ub = __builtin_elementwise_min(ub, c127);
    // bytes: [n0, n1, n2, n3] = low bytes of the four halves
    return (int)__builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, ub), __builtin_bit_cast(uint32_t, ua), 0x06040200u);
}

// NA >= H / 4: extent of the activation array (the classifier layer reads whole 8-dword chunks: 112 outputs = 28 dwords are
// padded to 32, the padding dwords are zero and meet zero weights)
template <int G, int H, int QL, int NA = H / 4>
BNM_DEVICE void tern_norm(const uint32_t *col, const uint32_t (&tail)[G][(H / 4 - QL) * 2 + 1], const int (&mx)[G],
                          int (&act)[G][NA]) {
    static_assert(NA >= H / 4, "activation array too short");
#pragma unroll
    for (int g = 0; g < G; g++) {
#pragma unroll
        for (int q = H / 4; q < NA; q++) act[g][q] = 0;
        uint32_t t = (uint32_t)mx[g] >> 7;                    // mx >= 0
        uint32_t sh = t ? 32u - (uint32_t)__builtin_clz(t) : 0u;
        uint32_t rnd = (1u << sh) >> 1;
        uint32_t rnd2 = rnd | (rnd << 16), sh2 = sh | (sh << 16);
        static_for<0, H / 4>([&](auto Q) {
            constexpr int q = decltype(Q)::value;
            uint32_t p01, p23;
            if constexpr (q < QL) {
                p01 = col[((g * QL + q) * 2 + 0) * 64];
                p23 = col[((g * QL + q) * 2 + 1) * 64];
            } else {
                p01 = tail[g][(q - QL) * 2 + 0];
                p23 = tail[g][(q - QL) * 2 + 1];
            }
            act[g][q] = tern_norm4(p01, p23, rnd2, sh2);
        });
    }
}

// One hidden layer: H neurons over KQ activation dwords.  Quads [0, QL) are a run-time loop whose sums go to the LDS
// column, quads [QL, H/4) are peeled and keep theirs in registers.  `a` holds the layer's first chunk on entry and the
// next layer's first chunk on exit (every layer here has an even number of chunks per loop step).
template <int G, int NX, int KQ, int H, int QL>
BNM_DEVICE void tern_layer_s(const int (&x)[G][NX], TernW &a, TernW &b, const int *&p, uint32_t *col,
                             uint32_t (&tail)[G][(H / 4 - QL) * 2 + 1], int (&mx)[G]) {
    constexpr int NC = KQ / 8;                 // chunks per quad
    constexpr int STEP = (NC & 1) ? 2 : 1;     // quads per loop step: an even number of chunks
    static_assert(QL % STEP == 0 && (H / 4 - QL) % STEP == 0, "quad split must keep the buffer parity");
#pragma unroll
    for (int g = 0; g < G; g++) mx[g] = 0;
    auto step = [&](auto park) {
        static_for<0, STEP>([&](auto S) {
            constexpr int s = decltype(S)::value;
            int acc[4][G];
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int g = 0; g < G; g++) acc[i][g] = 0;
            if constexpr ((s * NC) % 2 == 0) tern_quad<G, NX, KQ, true>(x, a, b, p + s * NC * 32, acc);
            else tern_quad<G, NX, KQ, false>(x, a, b, p + s * NC * 32, acc);
#pragma unroll
            for (int g = 0; g < G; g++) {
                int sums[4] = {acc[0][g], acc[1][g], acc[2][g], acc[3][g]};
                uint32_t p01, p23;
                tern_pairs(sums, mx[g], p01, p23);
                park(S, g, p01, p23);
            }
        });
        p += STEP * NC * 32;
    };
#pragma unroll 1
    for (int q = 0; q < QL; q += STEP)
        step([&](auto S, int g, uint32_t p01, uint32_t p23) {
            constexpr int s = decltype(S)::value;
            col[((g * QL + q + s) * 2 + 0) * 64] = p01;
            col[((g * QL + q + s) * 2 + 1) * 64] = p23;
        });
    static_for<0, (H / 4 - QL) / STEP>([&](auto T) {
        constexpr int t = decltype(T)::value;
        step([&](auto S, int g, uint32_t p01, uint32_t p23) {
            constexpr int s = decltype(S)::value;
            tail[g][(t * STEP + s) * 2 + 0] = p01;
            tail[g][(t * STEP + s) * 2 + 1] = p23;
        });
    });
}

// H1, H2: multiples of 32 (whole 8-dword chunks of activations for the next layer); H3: a multiple of 4, padded to KQ3 dwords for
// the classifier layer; QL <= H / 4 quads of every hidden layer park their sums in the LDS column, the rest in registers.
// WPE: waves per SIMD the register budget is compiled for (G = 2: 2; G = 1: 3 for 96-wide layers, 2 for 128-wide ones).
template <int G, int H1, int H2, int H3, int QL, int WPE = (G == 1 ? 3 : 2)>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
void ternary_stream_kernel(const int8_t *__restrict__ images, uint64_t n, const int *__restrict__ wstream,
                           uint32_t n_classes, uint32_t *__restrict__ cls_out, int32_t *__restrict__ logits_out,
                           uint32_t *__restrict__ counter) {
    static_assert(H1 % 32 == 0 && H2 % 32 == 0 && H3 % 4 == 0, "hidden widths: whole chunks for the next layer");
    static_assert(QL <= H1 / 4 && QL <= H2 / 4 && QL <= H3 / 4, "the LDS column holds QL quads of every hidden layer");
    constexpr int KQ3 = (H3 / 4 + 7) / 8 * 8;      // activation dwords the classifier layer reads (zero-padded)
    __shared__ uint32_t s_col[G * QL * 2 * 64];
    const int lane = threadIdx.x;
    uint32_t *col = s_col + lane;
    // Groups of 64 G images: a wave's first group is static, every later one comes from a device-wide counter (word 0 of the
    // launch's counter block, zero on entry).  With a fixed stride the two waves of a SIMD do not finish together - the arbiter favours the older one -
    // and the tail of the launch runs at one wave per SIMD; counter == nullptr keeps the fixed stride.
    const uint64_t stride = (uint64_t)gridDim.x * (64ull * G);
    uint64_t base = (uint64_t)blockIdx.x * (64ull * G);
    if (base >= n) {      // (the launcher starts no such wave; a wave that leaves must count itself out all the same)
        if (counter != nullptr) work_block_leave_v(counter, gridDim.x);
        return;
    }
    int nxt_v = 0;

    int x0[G][64];
    auto request = [&](uint64_t b) {
#pragma unroll
        for (int g = 0; g < G; g++) {
            uint64_t img = b + (uint64_t)(g * 64 + lane);
            img = img < n ? img : n - 1ull;
            const i32x4 *ptr = (const i32x4 *)(images + img * 256ull);
#pragma unroll
            for (int q = 0; q < 16; q++) {
                i32x4 v = __builtin_nontemporal_load(ptr + q);
                x0[g][4 * q + 0] = v[0]; x0[g][4 * q + 1] = v[1]; x0[g][4 * q + 2] = v[2]; x0[g][4 * q + 3] = v[3];
            }
        }
    };
    request(base);
    TernW wa, wb;
    tern_issue<true, 0>(wa, wstream);
    tern_land<true>(wa);
    const uint32_t nq4 = (n_classes + 3u) / 4u;
    while (base < n) {
        const int *p = wstream;
        if (counter != nullptr && lane == 0)
            nxt_v = (int)__hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t tail[G][(H1 / 4 - QL) * 2 + 1], tail2[G][(H2 / 4 - QL) * 2 + 1], tail3[G][(H3 / 4 - QL) * 2 + 1];
        int mx[G];
        int a1[G][H1 / 4], a2[G][H2 / 4], a3[G][KQ3];
        tern_layer_s<G, 64, 64, H1, QL>(x0, wa, wb, p, col, tail, mx);
        const uint64_t next_base = counter != nullptr
            ? ((uint64_t)gridDim.x + (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane(nxt_v)) * (64ull * G) : base + stride;
        if (next_base < n) request(next_base);      // lands under layers 2-4
        tern_norm<G, H1, QL>(col, tail, mx, a1);
        tern_layer_s<G, H1 / 4, H1 / 4, H2, QL>(a1, wa, wb, p, col, tail2, mx);
        tern_norm<G, H2, QL>(col, tail2, mx, a2);
        tern_layer_s<G, H2 / 4, H2 / 4, H3, QL>(a2, wa, wb, p, col, tail3, mx);
        tern_norm<G, H3, QL, KQ3>(col, tail3, mx, a3);
        // classifier layer (first strict maximum = ReLUNorm's return value): quads of classes, 3 chunks each; the stream
        // wraps to its first chunk after the last class quad, so `wa` is ready for the next images
        int bv[G];
        uint32_t bi[G];
#pragma unroll
        for (int g = 0; g < G; g++) { bv[g] = -INT_MAX; bi[g] = 255u; }
        auto classes = [&](auto AF, uint32_t q, const int *pq) {
            constexpr bool af = decltype(AF)::value;
            int acc[4][G];
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int g = 0; g < G; g++) acc[i][g] = 0;
            tern_quad<G, KQ3, KQ3, af>(a3, wa, wb, pq, acc);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint32_t c = 4u * q + (uint32_t)i;
                if (c < n_classes) {
#pragma unroll
                    for (int g = 0; g < G; g++) {
                        if (acc[i][g] > bv[g]) { bv[g] = acc[i][g]; bi[g] = c; }
                        const uint64_t img = base + (uint64_t)(g * 64 + lane);
                        if (logits_out && img < n) logits_out[img * n_classes + c] = acc[i][g];
                    }
                }
            }
        };
        constexpr int NC3 = KQ3 / 8;   // chunks per class quad
        // the stream ends with a copy of its first chunk, so the prefetch after the last class quad leaves the next
        // images' first chunk landed.  Odd chunk counts alternate the buffers per quad: after an odd number of quads the
        // chunk sits in wb (copied over: both buffers have landed); even counts start and end every quad in wa.
        if constexpr (NC3 % 2 == 1) {
            for (uint32_t q = 0; q < nq4; q += 2u) {
                classes(std::true_type{}, q, p);
                p += NC3 * 32;
                if (q + 1u < nq4) {
                    classes(std::false_type{}, q + 1u, p);
                    p += NC3 * 32;
                } else {
                    wa = wb;
                }
            }
        } else {
            for (uint32_t q = 0; q < nq4; q++) {
                classes(std::true_type{}, q, p);
                p += NC3 * 32;
            }
        }
#pragma unroll
        for (int g = 0; g < G; g++) {
            const uint64_t img = base + (uint64_t)(g * 64 + lane);
            if (img < n) cls_out[img] = bi[g];
        }
        base = next_base;
    }
    if (counter != nullptr) work_block_leave_v(counter, gridDim.x);   // the last wave to leave zeroes the counter block
}


// ---- the streamed one-image-per-lane kernel's family ------------------------------------------------------------------
// H1, H2 in {32, 64, 96, 128}, H3 a multiple of 16 up to 128 (the widths of the reference's ternary models are free parameters,
// models.py:62-84): 128 instantiations, one translation unit per H1.  QL, the quads of every hidden layer that park their sums in
// the LDS column, is as large as the narrowest layer allows, up to the 24 of the 96- and 128-wide shapes.
typedef void (*tern_stream_fn)(const int8_t *, uint64_t, const int *, uint32_t, uint32_t *, int32_t *, uint32_t *);
constexpr int tern_ql(int h1, int h2, int h3) {
    const int m = (h1 < h2 ? (h1 < h3 ? h1 : h3) : (h2 < h3 ? h2 : h3)) / 4;
    return m < 24 ? m : 24;
}
// waves per SIMD the register budget is compiled for: with a narrow layer in the model (QL < 16) the wide layers keep most of their
// sums in registers - three waves' 168 registers spill, two waves' 256 do not
constexpr int tern_wpe(int h1, int h2, int h3) { return tern_ql(h1, h2, h3) >= 16 ? 3 : 2; }
#define BNM_TERN_ROW(H1, H2)                                                                                                     \
    if (h2 == H2) {                                                                                                              \
        switch (h3) {                                                                                                            \
            case 16: return ternary_stream_kernel<1, H1, H2, 16, tern_ql(H1, H2, 16), tern_wpe(H1, H2, 16)>;                                            \
            case 32: return ternary_stream_kernel<1, H1, H2, 32, tern_ql(H1, H2, 32), tern_wpe(H1, H2, 32)>;                                            \
            case 48: return ternary_stream_kernel<1, H1, H2, 48, tern_ql(H1, H2, 48), tern_wpe(H1, H2, 48)>;                                            \
            case 64: return ternary_stream_kernel<1, H1, H2, 64, tern_ql(H1, H2, 64), tern_wpe(H1, H2, 64)>;                                            \
            case 80: return ternary_stream_kernel<1, H1, H2, 80, tern_ql(H1, H2, 80), tern_wpe(H1, H2, 80)>;                                            \
            case 96: return ternary_stream_kernel<1, H1, H2, 96, tern_ql(H1, H2, 96), tern_wpe(H1, H2, 96)>;                                            \
            case 112: return ternary_stream_kernel<1, H1, H2, 112, tern_ql(H1, H2, 112), tern_wpe(H1, H2, 112)>;                                         \
            case 128: return ternary_stream_kernel<1, H1, H2, 128, tern_ql(H1, H2, 128), tern_wpe(H1, H2, 128)>;                                         \
        }                                                                                                                        \
    }
#define BNM_TERN_UNIT(H1)                                                                                                        \
    tern_stream_fn bnmk_tern_stream1_h##H1(uint32_t h2, uint32_t h3) {                                                          \
        BNM_TERN_ROW(H1, 32) BNM_TERN_ROW(H1, 64) BNM_TERN_ROW(H1, 96) BNM_TERN_ROW(H1, 128)                                     \
        return nullptr;                                                                                                          \
    }
tern_stream_fn bnmk_tern_stream1_h32(uint32_t h2, uint32_t h3);
tern_stream_fn bnmk_tern_stream1_h64(uint32_t h2, uint32_t h3);
tern_stream_fn bnmk_tern_stream1_h96(uint32_t h2, uint32_t h3);
tern_stream_fn bnmk_tern_stream1_h128(uint32_t h2, uint32_t h3);
