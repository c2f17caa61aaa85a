// Diagnostic kernels (read-ceiling, synthetic-load and pipe-overlap probes).  NOT part of the product library: compiled
// only into libbitnetmcu_hip_diag.so (bitnetmcu_amd/build.py --diag), used by profiles/*.py.
#include "bnm_fused_tile.hpp"

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

template <bool PLAIN>
__global__ void diag_stream_compute_kernel(const int8_t *__restrict__ images, uint64_t n, uint32_t *__restrict__ out);

hipError_t bnmk_diag_stream(const int8_t *images, uint64_t n, int mode, int grid_blocks, uint32_t *out, hipStream_t s) {
    if (!n) return hipSuccess;
    int cus = bnm_num_cus();
    if (mode == 0) {
        unsigned blocks = grid_blocks > 0 ? (unsigned)grid_blocks : (unsigned)cus * 8u;
        diag_stream_plain_kernel<<<dim3(blocks), dim3(256), 0, s>>>((const u32x4 *)images, n * 16ull, out);
    } else {
        unsigned blocks = grid_blocks > 0 ? (unsigned)grid_blocks : (unsigned)cus * 2u;
        if (mode == 1) diag_stream_tiles_kernel<false><<<dim3(blocks), dim3(256), 0, s>>>(images, n, out);
        else if (mode == 2) diag_stream_tiles_kernel<true><<<dim3(blocks), dim3(256), 0, s>>>(images, n, out);
        else if (mode == 3) diag_stream_compute_kernel<false><<<dim3(blocks), dim3(256), 0, s>>>(images, n, out);
        else diag_stream_compute_kernel<true><<<dim3(blocks), dim3(256), 0, s>>>(images, n, out);
    }
    return hipGetLastError();
}


// -------------------------------------------------------------------------------------------------
// Diagnostics, modes 3/4: the image stream under a SYNTHETIC compute load of the real kernel's size (26 MFMAs +
// ~400 dependent-ish VALU per 32-image tile, operands from registers), fed either by the LDS-DMA tile loop (mode 3)
// or by plain coalesced 16 B/lane loads into double-buffered VGPRs (mode 4).  Question for the next round: does the
// 15 % the real kernel loses against its own stream-only loop come from the LDS-DMA path under load, or from any
// load path under load?
// -------------------------------------------------------------------------------------------------
BNM_DEVICE void fake_tile_compute(i32x16 &acc0, i32x16 &acc1, int &v0, int &v1, int &v2, int &v3, const i32x4 &a, const i32x4 &b) {
#pragma unroll
    for (int i = 0; i < 13; i++) {
        acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(b, a, acc1, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 50; i++) {     // 8 VALU per round, four independent chains
        v0 = min(max(v0 + 3, 0), 0x7fffff) ^ v3;
        v1 = min(max(v1 + 5, 0), 0x7fffff) ^ v0;
        v2 = (v2 >> 1) + v1;
        v3 = (v3 << 1) ^ v2;
    }
}

template <bool PLAIN>
__global__ __launch_bounds__(64 * FUSED_WPB, 2) void diag_stream_compute_kernel(const int8_t *__restrict__ images, uint64_t n,
                                                                                uint32_t *__restrict__ out) {
    __shared__ __attribute__((aligned(1024))) char smem[PLAIN ? 16 : FUSED_WPB * 2 * FUSED_TILE_BYTES];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint64_t n_tiles = n >> 5;
    const uint64_t stride = (uint64_t)gridDim.x * FUSED_WPB;
    uint64_t tile = (uint64_t)blockIdx.x * FUSED_WPB + wave;
    i32x16 acc0 = zero16(), acc1 = zero16();
    int v0 = lane, v1 = lane * 3, v2 = lane * 5, v3 = lane * 7;
    i32x4 fa = {lane, lane + 1, lane + 2, lane + 3}, fb = {lane * 2, 1, 2, 3};
    if constexpr (PLAIN) {
        const i32x4 *src = (const i32x4 *)images;
        i32x4 cur[8], nxt[8];
        auto load = [&](uint64_t t, i32x4(&d)[8]) {
#pragma unroll
            for (int k = 0; k < 8; k++) d[k] = __builtin_nontemporal_load(src + t * 512 + k * 64 + lane);
        };
        if (tile < n_tiles) load(tile, nxt);
        for (; tile < n_tiles; tile += stride) {
#pragma unroll
            for (int k = 0; k < 8; k++) cur[k] = nxt[k];
            if (tile + stride < n_tiles) load(tile + stride, nxt);
            fb[0] ^= cur[0][0] ^ cur[1][1] ^ cur[2][2] ^ cur[3][3] ^ cur[4][0] ^ cur[5][1] ^ cur[6][2] ^ cur[7][3];
            fake_tile_compute(acc0, acc1, v0, v1, v2, v3, fa, fb);
            if (lane < 32) out[(tile << 5) + lane] = (uint32_t)(acc0[0] ^ acc1[1] ^ v0 ^ v1 ^ v2 ^ v3);
        }
    } else {
        uint32_t voff[4];
#pragma unroll
        for (int u = 0; u < 4; u++) voff[u] = (uint32_t)(lane >> 4) * 256u + 16u * ((uint32_t)(lane & 15) ^ (uint32_t)(lane >> 4) ^ (4u * u));
        const uint32_t lds_wave = (uint32_t)(size_t)(__attribute__((address_space(3))) char *)smem + (uint32_t)wave * 2u * FUSED_TILE_BYTES;
        auto dma = [&](uint64_t t, int par) {
            const int8_t *base = images + t * (uint64_t)FUSED_TILE_BYTES;
            lds_dma_tile8<true, true>(lds_wave + (uint32_t)par * FUSED_TILE_BYTES, base, base + 1024, base + 2048, base + 3072,
                                      base + 4096, base + 5120, base + 6144, base + 7168, voff[0], voff[1], voff[2], voff[3],
                                      voff[0], voff[1], voff[2], voff[3]);
        };
        int par = 0;
        if (tile < n_tiles) dma(tile, 0);
        if (tile + stride < n_tiles) dma(tile + stride, 1);
        for (; tile < n_tiles; tile += stride) {
            const uint64_t next = tile + stride;
            if (next < n_tiles) bnm_wait_vmcnt<8>(); else bnm_wait_vmcnt<0>();
            const i32x4 *rb = (const i32x4 *)(smem + (uint32_t)wave * 2u * FUSED_TILE_BYTES + (uint32_t)par * FUSED_TILE_BYTES);
            i32x4 c0 = rb[lane], c1 = rb[64 + lane], c2 = rb[128 + lane], c3 = rb[192 + lane];
            i32x4 c4 = rb[256 + lane], c5 = rb[320 + lane], c6 = rb[384 + lane], c7 = rb[448 + lane];
            fb[0] ^= c0[0] ^ c1[1] ^ c2[2] ^ c3[3] ^ c4[0] ^ c5[1] ^ c6[2] ^ c7[3];
            if (next + stride < n_tiles) dma(next + stride, par);
            fake_tile_compute(acc0, acc1, v0, v1, v2, v3, fa, fb);
            if (lane < 32) out[(tile << 5) + lane] = (uint32_t)(acc0[0] ^ acc1[1] ^ v0 ^ v1 ^ v2 ^ v3);
            par ^= 1;
        }
    }
}

// -------------------------------------------------------------------------------------------------
// Diagnostics, modes 5/6/7: no memory traffic at all.  Each wave repeats a tile-sized block of 26 MFMAs (mode 5),
// ~400 VALU (mode 6) or both (mode 7), two waves per SIMD as in the fused kernel.  T(7) ~ T(5) + T(6) means the matrix
// pipe and the VALU of one SIMD do not overlap for this instruction mix; T(7) ~ max means they do.
// -------------------------------------------------------------------------------------------------
template <bool DO_MFMA, bool DO_VALU>
__global__ __launch_bounds__(64 * FUSED_WPB, 2) void diag_pipes_kernel(uint64_t tiles_per_wave, uint32_t *__restrict__ out) {
    __shared__ char pad[FUSED_WPB * 2 * FUSED_TILE_BYTES];   // same LDS footprint -> same residency (2 workgroups per CU)
    const int lane = threadIdx.x & 63;
    i32x16 acc0 = zero16(), acc1 = zero16();
    int v0 = lane, v1 = lane * 3, v2 = lane * 5, v3 = lane * 7;
    i32x4 fa = {lane, lane + 1, lane + 2, lane + 3}, fb = {lane * 2, 1, 2, 3};
    for (uint64_t t = 0; t < tiles_per_wave; t++) {
        if constexpr (DO_MFMA) {
#pragma unroll
            for (int i = 0; i < 13; i++) {
                acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa, fb, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(fb, fa, acc1, 0, 0, 0);
            }
        }
        if constexpr (DO_VALU) {
#pragma unroll
            for (int i = 0; i < 40; i++) {     // 10 VALU per round, four chains
                v0 = min(max(v0 + 3, 0), 0x7fffff) ^ v3;
                v1 = min(max(v1 + 5, 0), 0x7fffff) ^ v0;
                v2 = (v2 >> 1) + v1;
                v3 = (v3 << 1) ^ v2;
            }
        }
    }
    if (pad[threadIdx.x] == 123 || (acc0[0] ^ acc1[1] ^ v0 ^ v1 ^ v2 ^ v3) == 0x5a5a5a5a) out[threadIdx.x] = 1;
}

hipError_t bnmk_diag_pipes(int mode, uint64_t tiles_per_wave, uint32_t *out, hipStream_t s) {
    unsigned blocks = (unsigned)bnm_num_cus() * 2u;
    if (mode == 5) diag_pipes_kernel<true, false><<<dim3(blocks), dim3(256), 0, s>>>(tiles_per_wave, out);
    else if (mode == 6) diag_pipes_kernel<false, true><<<dim3(blocks), dim3(256), 0, s>>>(tiles_per_wave, out);
    else diag_pipes_kernel<true, true><<<dim3(blocks), dim3(256), 0, s>>>(tiles_per_wave, out);
    return hipGetLastError();
}
