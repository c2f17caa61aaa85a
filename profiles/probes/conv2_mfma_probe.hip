// Probe (round 4): the CNN's depthwise conv2 stage as a lane = image Toeplitz product on the matrix cores,
// timed against the v_dot2 stage of cnn_front_mfma_kernel that it would replace.  Standalone (no library): synthetic operands,
// no memory traffic in the timed loop - what is measured is the issue / pipe cost of each formulation at whatever clock the
// power cap leaves (the wall time and the shader-clock counter are both reported).
//
// Formulation measured (per channel, per tile of 32 images, images along the MFMA's N dimension):
//   input   conv1's outputs, 14 x 14 per channel and image, 14 bits after ReLU and >> 4 -> TWO int8 planes (low byte ^ 0x80, x >> 12),
//           K axis = 16 slots per input row (14 used), two input rows per 32-slot K-step: 7 K-steps per plane, 56 B-operand registers
//   output  12 x 12 positions in 6 ROW PAIRS; a row pair (2 x 12 = 24 of a tile's 32 D rows) reads input rows 2r .. 2r+3 = K-steps
//           r and r+1, and - the kernel being translation invariant - EVERY row pair uses the same two A fragments: a channel's
//           conv2 costs 2 KiB of LDS (64 channels: 128 KiB, resident) and 6 x 2 x 2 planes = 24 MFMAs per tile
//   D rows  ordered so that D-register quad q of a lane = the four positions of pooling window q (rows 4q .. 4q+3): the 2 x 2
//           maximum is in-lane.  Per row pair and lane: 12 combines (lo + (hi << 8): the plane weight 256 does not fit an int8
//           weight, so the planes have separate accumulators), 3 windows x (2 v_max3 incl. the ReLU's 0, 1 shift), and - PACK - 9
//           SDWA byte writes that split the pooled 20-bit values into conv3's three int8 planes
//   => per channel and tile: 24 MFMAs + 126 (180 with PACK) VALU, against the 432 v_dot2_i32_i16 (864 per image x 32 images / 64
//      lanes) that today's int16-pair formulation spends on the same 32 x 144 outputs - plus 90 for its pooling and shifts.
// Modes: 0 MFMA formulation, 1 the same with PACK, 2 the dot2 stage (432 v_dot2 + 90 VALU per channel-tile equivalent),
//        3 MFMAs only (24 per channel-tile), 4 the MFMA formulation's VALU only.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O2 -o conv2_mfma_probe profiles/probes/conv2_mfma_probe.hip && ./conv2_mfma_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef short i16x2 __attribute__((ext_vector_type(2)));

constexpr int CH = 64;             // channels
constexpr int WPS = 3;             // waves per SIMD

__device__ __forceinline__ int max3(int a, int b, int c) { return max(max(a, b), c); }

template <int MODE>
__global__ __launch_bounds__(256 * WPS) void probe(uint64_t *out, int tiles, int seed) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];      // [CH][2] A fragments of 1 KiB (synthetic weights)
    const int lane = threadIdx.x & 63;
    for (uint32_t o = threadIdx.x; o < CH * 2 * 64; o += blockDim.x) {
        i32x4 v = {(int)(o * 2654435761u) >> 3, (int)(o * 40503u + seed), (int)(o ^ (seed * 977)), (int)(o * 7919u)};
        ((i32x4 *)smem)[o] = v & 0x07070707;      // small weights, as the reference's conv kernels
    }
    __syncthreads();
    i32x4 lo[7], hi[7];
#pragma unroll
    for (int s = 0; s < 7; s++) {
        lo[s] = i32x4{seed * (s + 3) + lane, seed ^ (lane * 77), (seed + s) * 1234567, lane * 0x01010101 + s};
        hi[s] = i32x4{s + lane, lane ^ s, (seed + s) & 0x1f1f1f1f, (lane + s) & 0x0f0f0f0f} & 0x1f1f1f1f;
    }
    int sink = seed, shift = 4 + (seed & 1);
    i32x4 packed = {0, 0, 0, 0};
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int t = 0; t < tiles; t++) {
        for (int c = 0; c < CH; c++) {
            if (MODE == 2) {
                // today's stage: 432 dots of int16 pairs (weights as scalar operands in the real kernel) + its pooling / shifts
                int acc[12];
#pragma unroll
                for (int i = 0; i < 12; i++) acc[i] = sink + i;
                for (int k1 = 0; k1 < 36; k1++) {      // 36 x 12 = 432
#pragma unroll
                    for (int i = 0; i < 12; i++)
                        asm volatile("v_dot2_i32_i16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(lo[i % 7][i & 3]), "v"(hi[(i + 3) % 7][(i >> 2) & 3]));
                }
                for (int k1 = 0; k1 < 9; k1++) {       // 9 x 10 = 90
#pragma unroll
                    for (int i = 0; i < 10; i++) asm volatile("v_max3_i32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(acc[(i + 5) % 12]), "v"(lo[i % 7][1]));
                }
#pragma unroll
                for (int i = 0; i < 12; i++) sink ^= acc[i];
                continue;
            }
            const i32x4 a0 = *(const i32x4 *)(smem + (2 * c) * 1024 + 16 * lane), a1 = *(const i32x4 *)(smem + (2 * c + 1) * 1024 + 16 * lane);
#pragma unroll
            for (int r = 0; r < 6; r++) {
                i32x16 accl, acch;
                if (MODE != 4) {
                    i32x16 z;
#pragma unroll
                    for (int i = 0; i < 16; i++) z[i] = sink;      // (the bias 128 * sum(w): an initial accumulator)
                    accl = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, lo[r], z, 0, 0, 0);
                    acch = __builtin_amdgcn_mfma_i32_32x32x32_i8(a0, hi[r], z, 0, 0, 0);
                    accl = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, lo[r + 1], accl, 0, 0, 0);
                    acch = __builtin_amdgcn_mfma_i32_32x32x32_i8(a1, hi[r + 1], acch, 0, 0, 0);
                } else {
#pragma unroll
                    for (int i = 0; i < 16; i++) { accl[i] = sink + i; acch[i] = lo[r][i & 3] + i; }
                    asm volatile("" : "+v"(accl), "+v"(acch));
                }
                if (MODE == 3) {
                    sink ^= accl[0] ^ acch[5];
                    continue;
                }
#pragma unroll
                for (int q = 0; q < 3; q++) {      // three pooling windows per lane and row pair
                    int v[4];
#pragma unroll
                    for (int e = 0; e < 4; e++) v[e] = accl[4 * q + e] + (acch[4 * q + e] << 8);          // v_lshl_add_u32
                    int m = max3(v[0], v[1], v[2]);
                    m = max3(m, v[3], 0);                                                                // pool, ReLU
                    m >>= shift;
                    if (MODE == 1) {
                        // conv3's three int8 planes of the pooled value, byte (3 r + q) & 3 of three operand dwords
                        asm volatile("v_lshrrev_b32_sdwa %0, %3, %4 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n\t"
                                     "v_lshrrev_b32_sdwa %1, %3, %4 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_1 src1_sel:DWORD\n\t"
                                     "v_lshrrev_b32_sdwa %2, %3, %4 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2 src1_sel:DWORD"
                                     : "+v"(packed[0]), "+v"(packed[1]), "+v"(packed[2]) : "v"(q), "v"(m));
                    } else {
                        sink ^= m;
                    }
                }
            }
            if (MODE == 1) sink ^= packed[0] ^ packed[1] ^ packed[2];
            // next channel's input planes (in the real kernel: conv1's epilogue writes them); one cheap op keeps them opaque
            asm volatile("" : "+v"(lo[0]), "+v"(hi[0]), "+v"(lo[3]), "+v"(hi[5]));
        }
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    if (lane == 0) {
        out[2 * (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6))] = t1 - t0;
        out[2 * (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) + 1] = (uint64_t)sink;
    }
}

template <int MODE>
void run(uint64_t *d, const char *what, int tiles) {
    const int waves = 256 * 4 * WPS;
    hipFuncSetAttribute((const void *)probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, CH * 2 * 1024);
    probe<MODE><<<256, 256 * WPS, CH * 2 * 1024>>>(d, 4, 3);
    hipDeviceSynchronize();
    double best = 1e30, clocks = 0;
    for (int rep = 0; rep < 3; rep++) {
        const auto t0 = std::chrono::steady_clock::now();
        probe<MODE><<<256, 256 * WPS, CH * 2 * 1024>>>(d, tiles, 5 + rep);
        hipDeviceSynchronize();
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        uint64_t h[2];
        hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        if (s < best) { best = s; clocks = (double)h[0]; }
    }
    // a wave handles `tiles` tiles of 32 images x 64 channels per launch: images of the conv2 stage per second, whole chip
    const double images = (double)waves * tiles * 32.0;
    printf("mode %d  %-58s %8.3f ms  %9.4g images/s (conv2 stage only)  %7.0f clocks per channel-tile and wave  clock %.2f GHz\n", MODE, what,
           best * 1e3, images / best, clocks / ((double)tiles * CH), clocks / best / 1e9);
}

int main() {
    uint64_t *d;
    hipMalloc(&d, 16 * 256 * 4 * WPS + 64);
    const int tiles = 40;
    run<2>(d, "today: 432 v_dot2_i32_i16 + 90 VALU per channel-tile", tiles);
    run<0>(d, "lane = image: 24 MFMAs + 126 VALU per channel-tile", tiles);
    run<1>(d, "lane = image, + conv3 plane split (180 VALU)", tiles);
    run<3>(d, "the 24 MFMAs alone", tiles);
    run<4>(d, "the 126 VALU alone", tiles);
    hipFree(d);
    return 0;
}
