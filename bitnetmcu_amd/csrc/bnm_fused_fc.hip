// Fused whole-model FC kernels (int8 MFMA), register-resident weights, specialised per model shape.
// gfx950 (CDNA4 / MI355X) only; see DESIGN.md for layouts and rooflines.  Reference semantics:
// BitNetMCU_inference.c:23-72 (ReLUNorm), :88-208 (processfclayer), :238-277 (conv), :300-322 (pool);
// schedule BitNetMCU_MNIST_dll.c:48-121.
#include "bnm_fused_tile.hpp"
#include "bnm_fused_math.hpp"

// Measurement scaffolding of the diagnostic builds (build.py --diag / --diag-timing; profiles/wait_timing.py, slow_state_probe.py):
// compiled out of the product library.
#ifdef BNM_DIAG
#define DIAG_ONLY(...) __VA_ARGS__
#else
#define DIAG_ONLY(...)
#endif
#ifdef BNM_DIAG_TIMING
#define DIAG_TIMING(...) __VA_ARGS__      // shader-clock stamps around the kernel's two waits
#else
#define DIAG_TIMING(...)
#endif

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

// Kernel variants = how the image tile reaches the B operands:
//   0  DIRECT     global -> VGPR loads in operand layout (any row length; CNN tails with 64/128/192-byte rows)
//   1  LDSDMA     8 x 1 KiB non-temporal LDS-DMA pieces into a per-wave double buffer, next tile issued at the top of
//                 the iteration (non-temporal: +8..12 % on the 16-wide 1k model, neutral on 64-wide ones)
//   2  LDSDMA2    as 1 with TWO tiles in flight per wave (default where available): the refill of the buffer a tile
//                 just vacated (tile k+2) is issued right AFTER tile k's layer-1 MFMAs, non-temporal.  With one tile
//                 in flight the kernel is bound by per-wave memory-level parallelism (8 KiB / latency x 2048 waves);
//                 issuing the second DMA in front of the MFMAs instead serialises the wave (profiles/r01).
// Tried and dropped in round 1 (tag r01-experiments-all-variants): 8-wave workgroups with staggered halves, a
// software-pipelined MFMA||VALU form, three waves per SIMD with weights in LDS, split half-tile refills.
//   3  DUAL       two tiles per wave per iteration (fused_fc_dual_kernel), fixed stride per wave
//   5  DUAL_SHARED  the same loop in ONE 8-wave workgroup per CU whose waves take their pairs from a counter in LDS (4 is the
//                 generic kernel's id).  The SIMD arbiter favours the older of a SIMD's two waves: with a fixed stride the
//                 favoured waves finish after ~65 % of the launch and the rest of it runs at one wave per SIMD
//                 (profiles/r02/wait_timing_hbm_r02s_fixed_stride.json: 4.6 M .. 7.2 M cycles for the same 763 iterations).
//   6  DUAL_DEVWIDE  the dual-tile loop (two 4-wave workgroups per CU) with batches of pairs from ONE device-wide counter
//   9  REGW         3..5-tile shapes: weights resident in the register file (AccVGPRs) at one wave per SIMD, dual-tile loop,
//                 two pairs in flight (bnm_fused_regw.hip)
enum { FUSED_DIRECT = 0, FUSED_LDSDMA = 1, FUSED_LDSDMA2 = 2, FUSED_DUAL = 3, FUSED_DUAL_SHARED = 5, FUSED_DUAL_DEVWIDE = 6, FUSED_REGW = BNM_FUSED_REGW };

template <int KT0, int M1, int M2, int M3, int M4, bool SPLIT, bool DBL, int VARIANT, int NC8>
__global__ __launch_bounds__(64 * FUSED_WPB, 2) void fused_fc_kernel(const int8_t *__restrict__ images, uint64_t n,
                                                                     const i32x4 *__restrict__ frags, uint32_t n_classes,
                                                                     uint32_t *__restrict__ cls_out,
                                                                     int32_t *__restrict__ logits_out, uint64_t src_wrap,
                                                                     uint32_t *__restrict__ work, uint32_t *__restrict__ idle,
                                                                     uint32_t batch) {
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
#ifdef BNM_DIAG
        // diagnostic library only (build.py --diag, bnm_diag_set_src_wrap): read tile (t mod src_wrap) instead, so the
        // source stays cache-resident and the kernel's compute-side time can be measured without HBM in the way
        const int8_t *base = images + (src_wrap ? t % src_wrap : t) * (uint64_t)FUSED_TILE_BYTES;
#else
        const int8_t *base = images + t * (uint64_t)FUSED_TILE_BYTES;   // src_wrap is ignored by the product build
#endif
        uint32_t lds = lds_wave + (uint32_t)par * FUSED_TILE_BYTES;
        uint64_t first = t << 5;
        if (first + 32ull <= n) {
            lds_dma_tile8<true, TWO>(lds, base, base + 1024, base + 2048, base + 3072, base + 4096, base + 5120, base + 6144,
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
            lds_dma_tile8<true, TWO>(lds, base, base, base, base, base, base, base, base, v[0], v[1], v[2], v[3], v[4], v[5],
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

        // raised issue priority while the 16 layer-1 MFMAs go out: the partner wave on this SIMD is usually in a VALU
        // phase and loses nothing measurable (+0.5..1 %, profiles/r01); non-temporal DMA is worth another 2 %
        if constexpr (TWO) __builtin_amdgcn_s_setprio(1);
        layer_mma<M1, KT0, SPLIT>(A1, b0, acc1);
        if constexpr (TWO) __builtin_amdgcn_s_setprio(0);
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
            cls = argmax_rows<M4, NC8>(acc4, h);
            if (logits_out && img < n) store_logits<M4>(acc4, logits_out + img * n_classes, h, n_classes);
        } else {
            cls = argmax_rows<M3, NC8>(acc3, h);
            if (logits_out && img < n) store_logits<M3>(acc3, logits_out + img * n_classes, h, n_classes);
        }
        if (h == 0 && img < n) __builtin_nontemporal_store(cls, cls_out + img);
    }
}

// The dual kernel's logits of one pair (64 valid images): whole tiles through LDS (store_logits_tile) when a tile's rows fit the
// staging area, else piecewise.  Diagnostic library: `mode` 1 = plain whole-tile stores, 2 = the piecewise stores (round 2's first form).
#ifdef BNM_DIAG
#define LOGITS_DIAG_ARG , diag_logits_mode
#else
#define LOGITS_DIAG_ARG
#endif
template <int MT, int NC8>
__device__ __forceinline__ void store_logits_pair(const i32x16 (&a)[MT], const i32x16 (&b)[MT], int32_t *stage, int32_t *logits_out,
                                                  uint64_t pair, int j, int h, int lane, uint32_t n_classes
#ifdef BNM_DIAG
                                                  , uint32_t mode
#endif
) {
    int32_t *const tile_a = logits_out + (pair << 6) * n_classes, *const tile_b = tile_a + 32u * n_classes;
#ifdef BNM_DIAG
    if (mode == 1u && n_classes <= 16u) {
        store_logits_tile<MT, NC8, 1>(a, stage, tile_a, j, h, lane, n_classes);
        store_logits_tile<MT, NC8, 1>(b, stage, tile_b, j, h, lane, n_classes);
        return;
    }
    if (mode == 2u) {
        store_logits<MT>(a, tile_a + (uint32_t)j * n_classes, h, n_classes);
        store_logits<MT>(b, tile_b + (uint32_t)j * n_classes, h, n_classes);
        return;
    }
#endif
    if (n_classes <= 16u) {
        store_logits_tile<MT, NC8, 0>(a, stage, tile_a, j, h, lane, n_classes);
        store_logits_tile<MT, NC8, 0>(b, stage, tile_b, j, h, lane, n_classes);
    } else {
        store_logits<MT>(a, tile_a + (uint32_t)j * n_classes, h, n_classes);
        store_logits<MT>(b, tile_b + (uint32_t)j * n_classes, h, n_classes);
    }
}

// The dual kernel's deferred class-id store: nontemporal, executed under `mask` (all lanes or none) without a branch.
__device__ __forceinline__ void store_class_ids_masked(uint32_t *addr, uint32_t value, uint64_t mask) {
    uint64_t saved;
    asm volatile("s_and_saveexec_b64 %0, %1\n\tglobal_store_dword %2, %3, off nt\n\ts_mov_b64 exec, %0"
                 : "=&s"(saved) : "s"(mask), "v"(addr), "v"(value) : "memory", "scc");
}

#ifdef BNM_DIAG
// diagnostic library only: the class-id store in several flavours (see diag_store_mode in the kernel)
typedef uint32_t diag_u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t diag_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void diag_store(uint32_t mode, uint32_t *__restrict__ cls_out, uint64_t img_prev, uint64_t pair_prev,
                                           uint32_t cls_prev, uint32_t left, uint32_t batch, int lane, uint64_t mask) {
    switch (mode) {
    case 0: store_class_ids_masked(cls_out + img_prev, cls_prev, mask); break;   // the product's store
    case 1: cls_out[img_prev] = cls_prev; break;                                  // plain store (the product's until r02)
    case 2: asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(cls_out + img_prev), "v"(cls_prev) : "memory"); break;
    case 3: cls_out[img_prev & 0xffffull] = cls_prev; break;                      // same instruction, cache-resident destination
    case 4: ((uint8_t *)cls_out)[img_prev] = (uint8_t)cls_prev; break;            // same number of stores, a quarter of the bytes
    case 5:                                                                       // one store per BATCH of 2 / 4 pairs, same bytes
        if (left == 0u) {
            const uint64_t first = (pair_prev & ~(uint64_t)(batch - 1u)) << 6;
            if (batch == 2u) *(diag_u32x2 *)(cls_out + first + 2u * (uint32_t)lane) = diag_u32x2{cls_prev, cls_prev};
            else *(diag_u32x4 *)(cls_out + first + 4u * (uint32_t)lane) = diag_u32x4{cls_prev, cls_prev, cls_prev, cls_prev};
        }
        break;
    case 6: {                                                                     // through the scalar unit: 16 x s_store_dwordx4
        uint32_t *const dst = cls_out + (pair_prev << 6);
#pragma unroll
        for (int q = 0; q < 16; q++) {
            diag_u32x4 v = {(uint32_t)__builtin_amdgcn_readlane((int)cls_prev, 4 * q), (uint32_t)__builtin_amdgcn_readlane((int)cls_prev, 4 * q + 1),
                            (uint32_t)__builtin_amdgcn_readlane((int)cls_prev, 4 * q + 2), (uint32_t)__builtin_amdgcn_readlane((int)cls_prev, 4 * q + 3)};
            asm volatile("s_store_dwordx4 %0, %1, %2" ::"s"(v), "s"(dst), "n"(16 * q) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the data SGPRs are free again
        break;
    }
    default: break;                                                               // any other value (the scripts use 255): no store at all
    }
}
#endif

// ---- variant 3, DUAL: one wave carries TWO independent tiles (A, B) per iteration -----------------
// Same LDS budget as variant 2 (one 8 KiB buffer per tile slot, refilled right after that slot's layer-1 MFMAs), but
// the two tiles' layer chains are independent instruction streams inside one wave, and the loop body is ONE basic
// block (no ragged / last-iteration branches: the launcher hands this kernel whole 64-image pairs only and gives
// the remainder to variant 2; the refill after the last pair re-reads that pair), so the scheduler can place one
// tile's ReLUNorm VALU work between the other tile's MFMAs.  Weights stay in registers once for both tiles.
// LDS per wave: two 8 KiB tile slots + 2 KiB staging for a tile's logits (store_logits_tile) = 72 KiB per 4-wave workgroup, two per CU.
template <int M1, int M2, int M3, int M4, bool DBL, int NC8, int WPB = FUSED_WPB, bool DW = false>
__global__ __launch_bounds__(64 * WPB, 2) void fused_fc_dual_kernel(const int8_t *__restrict__ images, uint64_t n,
                                                                          const i32x4 *__restrict__ frags,
                                                                          uint32_t n_classes, uint32_t *__restrict__ cls_out,
                                                                          int32_t *__restrict__ logits_out,
                                                                          uint64_t src_wrap, uint32_t *__restrict__ work,
                                                                          uint32_t *__restrict__ idle, uint32_t batch_arg) {
    constexpr int KT0 = 8;
    const uint32_t batch = batch_arg & 0xFFFFu;
    constexpr bool SHARED = WPB == 8;          // one workgroup per CU, pairs handed out from s_next
    constexpr bool DEVWIDE = DW;   // 4-wave workgroups, batches of pairs from the device-wide counter work[0]
    __shared__ __attribute__((aligned(1024))) char smem[WPB * 2 * FUSED_TILE_BYTES];
    __shared__ __attribute__((aligned(16))) int32_t s_logits[WPB][32 * 16];      // store_logits_tile's staging area, 2 KiB per wave
    __shared__ uint32_t s_next;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;
    if constexpr (SHARED) {
        if (threadIdx.x == 0) s_next = 0;
        __syncthreads();
    }
    // the workgroup's m-th pair is pair m * gridDim.x + blockIdx.x; ONE lane performs the LDS atomic (with all 64 lanes on the
    // same word it occupied the LDS pipe for 64 cycles per iteration, a quarter of what the tile traffic itself needs)
    auto take = [&]() -> uint64_t {
        uint32_t m = 0;
        if (lane == 0) m = __hip_atomic_fetch_add(&s_next, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        m = (uint32_t)__builtin_amdgcn_readfirstlane((int)m);
        return (uint64_t)m * gridDim.x + blockIdx.x;
    };

    AFrags<M1, KT0> A1;
    AFrags<M2, M1> A2;
    AFrags<M3, M2> A3;
    AFrags<(M4 > 0 ? M4 : 1), M3> A4;
    const i32x4 *fp = frags;
    A1.load(fp, lane);  fp += M1 * KT0 * 64;
    A2.load(fp, lane);  fp += M2 * M1 * 64;
    A3.load(fp, lane);  fp += M3 * M2 * 64;
    if constexpr (M4 > 0) A4.load(fp, lane);

    const uint64_t n_pairs = n >> 6;            // the launcher guarantees n % 64 == 0
    const uint64_t stride = (uint64_t)gridDim.x * WPB;
    // DEVWIDE: batches of `batch` (>= 2) consecutive pairs; a wave's first batch is static, later ones come from work[0] on the
    // scalar unit (work_take_*).  The take is issued in EVERY iteration at the same place - behind the iteration's last LDS
    // read, so that no lgkmcnt wait of the iteration covers it - and the loop body stays one basic block: in the iteration
    // before a batch's last pair it adds 1 to its counter word, in all others it adds 0 to a word of the wave's own
    // (idle[16 wave id]); scalar selects pick the address, the amount and, one iteration later, the result.
    // The counter is split into `words` words (8 when the wave count allows it): wave w takes from word w % words, which hands
    // out the batches g = t * words + (w % words), t = 0, 1, ... - every word is shared by waves of ALL CUs, so the split keeps
    // the balance device-wide while one word only sees an eighth of the takes (one word serves ~88 M takes/s, eight ~430 M/s:
    // profiles/r02/s_atomic_rate_r02.log), which is what allows small batches.
    const uint32_t wave_id = blockIdx.x * WPB + (uint32_t)wave, total_waves = gridDim.x * WPB;
    // (the launcher passes words in the upper half of `batch` and first_dyn = total_waves / words needs no division: both
    // derived on the device they ended up on the vector unit, which the scalar asm operands below cannot take)
    // (no selects here either: `words == 8 ? 3 : 0` is materialised with v_cndmask)
    const uint32_t words = DEVWIDE ? batch_arg >> 16 : 1u, wshift = (uint32_t)__builtin_ctz(words | 0x100u);   // 8 -> 3, 1 -> 0
    const uint32_t my_word = wave_id & (words - 1u), first_dyn = total_waves >> wshift;
    // the real take happens in the iteration whose `left` equals `trigger`: 1 (the pair before a batch's last), or 0 for batches
    // of ONE pair, where every iteration takes (and the take for the second pair is issued ahead of the loop)
    const uint32_t trigger = min(batch - 1u, 1u);
    uint32_t left = DEVWIDE ? batch - 1u : 0u, taken = 0;
    uint64_t pair = SHARED ? take() : DEVWIDE ? (uint64_t)wave_id * batch : (uint64_t)blockIdx.x * WPB + wave;

    uint32_t voff[4];
#pragma unroll
    for (int u = 0; u < 4; u++) voff[u] = (uint32_t)(lane >> 4) * 256u + 16u * ((uint32_t)(lane & 15) ^ (uint32_t)(lane >> 4) ^ (4u * u));
    const uint32_t lds_wave = (uint32_t)(size_t)(__attribute__((address_space(3))) char *)smem + (uint32_t)wave * 2u * FUSED_TILE_BYTES;
    const uint32_t rd_base = (uint32_t)wave * 2u * FUSED_TILE_BYTES + (uint32_t)j * 256u + 16u * ((uint32_t)h ^ (uint32_t)(j & 15));

    // slot 0 holds tile 2p, slot 1 tile 2p+1.  Diagnostic library only: src_wrap (a power of two here) keeps the
    // source cache-resident; as a mask it costs one s_and and no branch.  The product build ignores the argument.
#ifdef BNM_DIAG
    // bits 56..63 of the argument select how the class ids are stored (profiles/slow_state_probe.py: what do the writes cost on
    // this box, and which property of them - their number, their bytes, the vector memory pipe - is it?).  TIMING ONLY: every mode
    // but 0 leaves the class buffer with wrong or partial contents.
    const uint32_t diag_store_mode = (uint32_t)(src_wrap >> 56), diag_logits_mode = (uint32_t)(src_wrap >> 48) & 0xffu;
    const uint64_t wrap_lo = src_wrap & ~(0xffffull << 48);
    uint64_t pair_prev = pair;
    const uint64_t wrap_mask = wrap_lo ? wrap_lo - 1ull : ~0ull;
#else
    constexpr uint64_t wrap_mask = ~0ull;
#endif
    auto dma_tile = [&](uint64_t t, int slot) {
        const int8_t *base = images + (t & wrap_mask) * (uint64_t)FUSED_TILE_BYTES;
        lds_dma_tile8_linear(lds_wave + (uint32_t)slot * FUSED_TILE_BYTES, base, base + 4096, voff[0], voff[1], voff[2], voff[3]);
    };
    auto read_tile = [&](int slot, i32x4(&b)[KT0]) {
#pragma unroll
        for (int s = 0; s < KT0; s++) b[s] = *(const i32x4 *)(smem + ((rd_base ^ (32u * s)) + (uint32_t)slot * FUSED_TILE_BYTES));
    };

    const bool any = pair < n_pairs;
    if (any) {
        dma_tile(2ull * pair, 0);
        dma_tile(2ull * pair + 1ull, 1);
    }
    if constexpr (DEVWIDE) {
        if (batch == 1u) work_take_issue(taken, work + 16u * my_word, 1u);
    }
    // Class ids are stored HALF AN ITERATION LATE (after the next pair's second wait).  vmcnt counts stores too, and
    // "<= 8 outstanding" retires everything older than the newest 8 operations: a store issued at the end of the
    // body would sit between the two refills and the next pair's second wait would stall on its write
    // acknowledgement (~1000 cycles per iteration while HBM is streaming; profiles/r01/conly_r01p...).  Deferred, the
    // only store either wait can cover is a whole iteration old.  The first iteration's deferred store runs with an
    // empty exec mask (see store_class_ids_masked).
    uint64_t img_prev = (pair << 6) + (uint64_t)lane;
    uint32_t cls_prev = 0;
    uint64_t store_mask = 0;      // exec mask of the deferred store: empty in a wave's first iteration
    DIAG_TIMING(uint64_t t_wait_a = 0, t_wait_b = 0, t_iters = 0; const uint64_t t_start = __builtin_readcyclecounter();)
    while (pair < n_pairs) {
        // the refill after the last pair re-reads that pair (keeps the wait counts constant and the body branch-free)
        uint64_t cand;
        if constexpr (DEVWIDE) {
            work_take_wait(taken);        // the take of the previous iteration (two thirds of an iteration old)
            // (32-bit arithmetic - the launcher refuses 2^31 pairs - so that this is two scalar selects, not a branch)
            // (a word counts its own BATCHES: every take adds 1; the static first batches are g = 0 .. total_waves - 1)
            const uint32_t c32 = left != 0u ? (uint32_t)pair + 1u : (((first_dyn + taken) << wshift) + my_word) * batch;
            cand = c32;
        } else {
            cand = SHARED ? take() : pair + stride;
        }
        const uint64_t next = cand < n_pairs ? cand : pair;
        // outstanding, oldest first: slot 0 (8 pieces), [the deferred store], slot 1 (8 pieces).
        // Loads retire in order among themselves, so "<= 8 left" implies slot 0 has landed.
        DIAG_TIMING(const uint64_t t0 = __builtin_readcyclecounter();)
        bnm_wait_vmcnt<8>();
        DIAG_TIMING(t_wait_a += __builtin_readcyclecounter() - t0; t_iters++;)
        i32x4 bA[KT0], bB[KT0];
        i32x16 a1A[M1], a1B[M1];
        read_tile(0, bA);
        layer_mma<M1, KT0, false>(A1, bA, a1A);
        dma_tile(2ull * next, 0);
        DIAG_TIMING(const uint64_t t2 = __builtin_readcyclecounter();)
        bnm_wait_vmcnt<8>();     // slot 1 is now the oldest load group
        DIAG_TIMING(t_wait_b += __builtin_readcyclecounter() - t2;)
#ifdef BNM_DIAG
        diag_store(diag_store_mode, cls_out, img_prev, pair_prev, cls_prev, left, batch, lane, store_mask);
#else
        // nontemporal: a plain store leaves its line dirty in L2 and the write-back lands between the reads whenever the stream
        // evicts it; 0.11-0.16 ms of 4.1 on the fast boxes, 0.5 ms on the slow ones (profiles/r02/store_modes_ab_r02*.log).
        // The first iteration has nothing to store yet: the store issues with an empty exec mask (one asm statement, so the loop
        // body stays one basic block).  It must not write a placeholder that the real value overwrites later: the latency path of
        // bnm_infer_host polls the class words in page-locked host memory and takes the first change of a word as its result
        // (bnm_capi_host.cpp, infer_host_small) - every word is written exactly once.
        store_class_ids_masked(cls_out + img_prev, cls_prev, store_mask);
#endif
        read_tile(1, bB);
        layer_mma<M1, KT0, false>(A1, bB, a1B);
        dma_tile(2ull * next + 1ull, 1);
        if constexpr (DEVWIDE) {
            // The statement names one accumulator register of each of tile B's layer-1 MFMA chains, so hipcc places it behind
            // the last of those MFMAs - i.e. behind the iteration's last LDS operand wait; nothing after that point touches
            // LDS until the next iteration, so no lgkmcnt wait sits on the take's round trip.
            uint32_t *const addr = left == trigger ? work + 16u * my_word : idle + 16u * wave_id;
            // (an opaque scalar 1: written as `left == 1u ? 1u : 0u` hipcc materialises the comparison on the vector unit and
            // the asm's scalar operand no longer is one)
            uint32_t one = 1u;
            asm volatile("" : "+s"(one));
            const uint32_t amount = left == trigger ? one : 0u;
            asm volatile("s_mov_b32 %0, %2\n\ts_atomic_add %0, %1, 0x0 glc"
                         : "=&{s95}"(taken) : "s"(addr), "s"(amount), "v"(a1B[0][15]), "v"(a1B[M1 - 1][15]) : "memory");
        }

        i32x4 p1A[M1], p1B[M1];
        relunorm_pack<M1, DBL>(a1A, p1A, h);
        i32x16 a2A[M2], a2B[M2];
        layer_mma<M2, M1, false>(A2, p1A, a2A);
        relunorm_pack<M1, DBL>(a1B, p1B, h);
        layer_mma<M2, M1, false>(A2, p1B, a2B);

        i32x4 p2A[M2], p2B[M2];
        relunorm_pack<M2, DBL>(a2A, p2A, h);
        i32x16 a3A[M3], a3B[M3];
        layer_mma<M3, M2, false>(A3, p2A, a3A);
        relunorm_pack<M2, DBL>(a2B, p2B, h);
        layer_mma<M3, M2, false>(A3, p2B, a3B);

        const uint64_t imgA = (pair << 6) + (uint64_t)j, imgB = imgA + 32ull;
        uint32_t clsA, clsB;
        if constexpr (M4 > 0) {
            i32x4 p3A[M3], p3B[M3];
            relunorm_pack<M3, DBL>(a3A, p3A, h);
            i32x16 a4A[M4], a4B[M4];
            layer_mma<M4, M3, false>(A4, p3A, a4A);
            relunorm_pack<M3, DBL>(a3B, p3B, h);
            layer_mma<M4, M3, false>(A4, p3B, a4B);
            clsA = argmax_rows<M4, NC8>(a4A, h);
            clsB = argmax_rows<M4, NC8>(a4B, h);
#ifndef BNM_DIAG_TIMING
            if (logits_out) store_logits_pair<M4, NC8>(a4A, a4B, s_logits[wave], logits_out, pair, j, h, lane, n_classes LOGITS_DIAG_ARG);
#endif
        } else {
            clsA = argmax_rows<M3, NC8>(a3A, h);
            clsB = argmax_rows<M3, NC8>(a3B, h);
#ifndef BNM_DIAG_TIMING
            if (logits_out) store_logits_pair<M3, NC8>(a3A, a3B, s_logits[wave], logits_out, pair, j, h, lane, n_classes LOGITS_DIAG_ARG);
#endif
        }
        // both halves of the wave hold the result: lanes 0..31 keep tile A's classes, lanes 32..63 tile B's —
        // one 256-byte store per pair, issued in the next iteration (or after the loop)
        img_prev = h ? imgB : imgA;
        cls_prev = h ? clsB : clsA;
        store_mask = ~0ull;
        DIAG_ONLY(pair_prev = pair;)
        pair = cand;
        if constexpr (DEVWIDE) left = left != 0u ? left - 1u : batch - 1u;
    }
    // the loop's final take is still in flight: retire it before its result register can be given to anything else
    if constexpr (DEVWIDE) work_take_wait(taken);
    if (any) __builtin_nontemporal_store(cls_prev, cls_out + img_prev);
    DIAG_ONLY(if (diag_store_mode == 6u) asm volatile("s_dcache_wb" ::: "memory");)
    bnm_wait_vmcnt<0>();   // LDS-DMA still in flight must not outlive the workgroup's LDS allocation
    // the launch's counter block goes back to all-zero with the last wave to leave
    if constexpr (DEVWIDE) work_block_leave_s(work, total_waves);
#ifdef BNM_DIAG_TIMING
    // the logits buffer is reused as the record array: 4 x uint64 per wave {loop cycles, wait A, wait B, iterations}
    if (logits_out && lane == 0) {
        uint64_t *rec = (uint64_t *)logits_out + 4ull * ((uint64_t)blockIdx.x * WPB + (uint64_t)wave);
        rec[0] = __builtin_readcyclecounter() - t_start;
        rec[1] = t_wait_a;
        rec[2] = t_wait_b;
        rec[3] = t_iters;
    }
#endif
}

// ---- dispatch table: model shapes of the reference zoo (+ ternary) -----------------------------
namespace {
typedef void (*fused_fn)(const int8_t *, uint64_t, const i32x4 *, uint32_t, uint32_t *, int32_t *, uint64_t, uint32_t *, uint32_t *, uint32_t);
struct FusedEntry {
    BnmFusedShape sh;
    int variant;
    fused_fn fn;
};
// NC8 = 0: any class count; NC8 = k: specialised for n_classes <= 8k (see argmax_rows)
#define FUSED(KT0, M1, M2, M3, M4, SPLIT, DBL, VAR, NC8) \
    { {KT0, {M1, M2, M3, M4}, SPLIT, DBL, NC8}, VAR, fused_fc_kernel<KT0, M1, M2, M3, M4, SPLIT, DBL, VAR, NC8> }
#define FUSED_ANY_AND_10(KT0, M1, M2, M3, M4, SPLIT, DBL, VAR) \
    FUSED(KT0, M1, M2, M3, M4, SPLIT, DBL, VAR, 2), FUSED(KT0, M1, M2, M3, M4, SPLIT, DBL, VAR, 0)
const FusedEntry kFused[] = {
    // FC 256-64-64-64-10 4bitsym (BitNetMCU_model_fc.h, mcu/BitNetMCU_model_12k.h) — the headline shape
    { {8, {2, 2, 2, 1}, false, true, 2}, FUSED_DUAL_DEVWIDE, fused_fc_dual_kernel<2, 2, 2, 1, true, 2, 4, true> },
    { {8, {2, 2, 2, 1}, false, true, 0}, FUSED_DUAL_DEVWIDE, fused_fc_dual_kernel<2, 2, 2, 1, true, 0, 4, true> },
    { {8, {2, 2, 2, 1}, false, true, 2}, FUSED_DUAL_SHARED, fused_fc_dual_kernel<2, 2, 2, 1, true, 2, 8> },
    { {8, {2, 2, 2, 1}, false, true, 0}, FUSED_DUAL_SHARED, fused_fc_dual_kernel<2, 2, 2, 1, true, 0, 8> },
    { {8, {2, 2, 2, 1}, false, true, 2}, FUSED_DUAL, fused_fc_dual_kernel<2, 2, 2, 1, true, 2> },
    { {8, {2, 2, 2, 1}, false, true, 0}, FUSED_DUAL, fused_fc_dual_kernel<2, 2, 2, 1, true, 0> },
    FUSED_ANY_AND_10(8, 2, 2, 2, 1, false, true, FUSED_LDSDMA2),
    FUSED_ANY_AND_10(8, 2, 2, 2, 1, false, true, FUSED_LDSDMA),
    FUSED(8, 2, 2, 2, 1, false, true, FUSED_DIRECT, 0),
    // same shape with codecs whose weights cannot be doubled in int8 (8-bit two's complement, FP1.3.0 without +128)
    { {8, {2, 2, 2, 1}, false, false, 2}, FUSED_DUAL_DEVWIDE, fused_fc_dual_kernel<2, 2, 2, 1, false, 2, 4, true> },
    { {8, {2, 2, 2, 1}, false, false, 0}, FUSED_DUAL_DEVWIDE, fused_fc_dual_kernel<2, 2, 2, 1, false, 0, 4, true> },
    { {8, {2, 2, 2, 1}, false, false, 2}, FUSED_DUAL_SHARED, fused_fc_dual_kernel<2, 2, 2, 1, false, 2, 8> },
    { {8, {2, 2, 2, 1}, false, false, 0}, FUSED_DUAL_SHARED, fused_fc_dual_kernel<2, 2, 2, 1, false, 0, 8> },
    { {8, {2, 2, 2, 1}, false, false, 2}, FUSED_DUAL, fused_fc_dual_kernel<2, 2, 2, 1, false, 2> },
    { {8, {2, 2, 2, 1}, false, false, 0}, FUSED_DUAL, fused_fc_dual_kernel<2, 2, 2, 1, false, 0> },
    FUSED_ANY_AND_10(8, 2, 2, 2, 1, false, false, FUSED_LDSDMA2),
    FUSED(8, 2, 2, 2, 1, false, false, FUSED_LDSDMA, 0),
    FUSED(8, 2, 2, 2, 1, false, false, FUSED_DIRECT, 0),
    // (FP1.3.0 models that really contain a +128 weight need a second weight plane: 2 x the A fragments do not fit the
    // register file without spilling, so those go to the generic kernel, whose weights live in LDS)
    // FC 256-16-16-10 2bitsym (mcu/BitNetMCU_model_1k.h)
    { {8, {1, 1, 1, 0}, false, true, 2}, FUSED_DUAL_DEVWIDE, fused_fc_dual_kernel<1, 1, 1, 0, true, 2, 4, true> },
    { {8, {1, 1, 1, 0}, false, true, 2}, FUSED_DUAL_SHARED, fused_fc_dual_kernel<1, 1, 1, 0, true, 2, 8> },
    { {8, {1, 1, 1, 0}, false, true, 2}, FUSED_DUAL, fused_fc_dual_kernel<1, 1, 1, 0, true, 2> },
    FUSED(8, 1, 1, 1, 0, false, true, FUSED_LDSDMA2, 0),
    FUSED_ANY_AND_10(8, 1, 1, 1, 0, false, true, FUSED_LDSDMA),
    FUSED(8, 1, 1, 1, 0, false, true, FUSED_DIRECT, 0),
    // (96-wide four-layer shapes, e.g. the ternary model through MFMA: 132 weight registers spill -> generic kernel)
    // CNN FC tails: 4C-96-64-10 (cnn_64/48/32/16), 64-64-48-10 (cnn_16small), 256-96-64-37 (letters)
    FUSED(8, 3, 2, 1, 0, false, true, FUSED_LDSDMA, 0),
    FUSED(8, 3, 2, 1, 0, false, true, FUSED_DIRECT, 0),
    FUSED(6, 3, 2, 1, 0, false, true, FUSED_DIRECT, 0),
    FUSED(4, 3, 2, 1, 0, false, true, FUSED_DIRECT, 0),
    FUSED(2, 3, 2, 1, 0, false, true, FUSED_DIRECT, 0),
    FUSED(2, 2, 2, 1, 0, false, true, FUSED_DIRECT, 0),
    // (256-96-64-37, the letters CNN's tail: spilled 3-4 registers here -> generic kernel)
};
// exact class-count specialisation first, then the any-count instantiation of the same shape
const FusedEntry *find_fused(const BnmFusedShape &sh, int variant) {
    for (int pass = 0; pass < 2; pass++) {
        BnmFusedShape want = sh;
        if (pass) want.nc8 = 0;
        for (const FusedEntry &e : kFused)
            if (e.sh == want && e.variant == variant) return &e;
    }
    return nullptr;
}
}  // namespace

bool bnmk_fused_supported(const BnmFusedShape &sh, int variant) {
    return variant == FUSED_REGW ? bnmk_regw_supported(sh) : find_fused(sh, variant) != nullptr;
}
// measured best first (profiles/r01): the dual-tile kernel wherever it is instantiated (64-wide four-layer shapes:
// 4.40-4.55 vs 4.63-4.70 ms per 1e8 images; the 16-wide 1k model: 4.29 vs 4.33 ms), then two tiles in flight for
// shapes whose tiles carry real work, else the plain one-ahead loop
int bnmk_fused_default_variant(const BnmFusedShape &sh) {
    // round 2: the dual-tile loop with the CU's waves sharing a work counter is 2.5-3 % ahead of the fixed stride on the
    // same box (profiles/r02/headline_ab_r02v_variants_3_4_5.log)
    // ... and batches of 2 pairs from the device-wide counter (split eight ways) 6-7.5 % ahead of the fixed stride
    // (same-process interleaved A/B, profiles/headline_ab.py: profiles/r02/headline_ab_r02x.json ... r02z4.json)
    if (find_fused(sh, FUSED_DUAL_DEVWIDE)) return FUSED_DUAL_DEVWIDE;
    if (find_fused(sh, FUSED_DUAL_SHARED)) return FUSED_DUAL_SHARED;
    if (find_fused(sh, FUSED_DUAL)) return FUSED_DUAL;
    if (sh.M[0] >= 2 && find_fused(sh, FUSED_LDSDMA2)) return FUSED_LDSDMA2;
    return find_fused(sh, FUSED_LDSDMA) ? FUSED_LDSDMA : FUSED_DIRECT;
}

hipError_t bnmk_fused_fc(const BnmFusedShape &sh, int variant, int grid_blocks, const BnmFusedArgs &a, hipStream_t s) {
    if (variant == FUSED_REGW) return bnmk_fused_regw(sh, grid_blocks, a, s);
    const FusedEntry *e = find_fused(sh, variant);
    if (!e) return hipErrorInvalidValue;
    if (a.n == 0) return hipSuccess;
    if (variant == FUSED_DUAL || variant == FUSED_DUAL_SHARED || variant == FUSED_DUAL_DEVWIDE) {
        // whole 64-image pairs go to the dual-tile kernel, the remainder (< 64 images) to variant 2
        const uint64_t n_main = a.n & ~63ull;
        if (variant == FUSED_DUAL_DEVWIDE && n_main) {
            // launch-bound sizes: when the fixed stride already gives every resident wave at most ONE pair there is nothing to
            // distribute - the fixed-stride kernel does the same work in one dispatch instead of two (no counter to zero):
            // 8.5 -> 4.6 us per call back to back (profiles/graph_replay.py)
            const uint64_t resident_waves = (grid_blocks > 0 ? (uint64_t)grid_blocks : (uint64_t)bnm_num_cus() * 2ull) * FUSED_WPB;
            if ((n_main >> 6) <= resident_waves && find_fused(sh, FUSED_DUAL)) return bnmk_fused_fc(sh, FUSED_DUAL, grid_blocks, a, s);
        }
        if (n_main) {
            // fixed stride: two 4-wave workgroups per CU; shared counter: ONE 8-wave workgroup per CU
            const uint64_t wpb = variant == FUSED_DUAL_SHARED ? 8 : FUSED_WPB;
            uint32_t batch = 1;
            if (variant == FUSED_DUAL_DEVWIDE) {
                // work[16 k], k < 8: the counter words (all zero between launches: the kernel's last wave puts them back, so
                // the launch is ONE dispatch); idle[16 w]: wave w's own word for the zero-adds
                if (!a.work || !a.idle || (n_main >> 6) >= (1ull << 31)) return hipErrorInvalidValue;
                batch = a.batch >= 1 ? a.batch : BNM_DUAL_DEFAULT_BATCH;
            }
            uint64_t want = ((n_main >> 6) + wpb * batch - 1) / (wpb * batch);
            uint64_t cap = grid_blocks > 0 ? (uint64_t)grid_blocks : (uint64_t)bnm_num_cus() * (variant == FUSED_DUAL_SHARED ? 1ull : 2ull);
            if (variant == FUSED_DUAL_DEVWIDE && cap * wpb > BNM_WORK_DUMMY_WAVES) cap = BNM_WORK_DUMMY_WAVES / wpb;
            const uint64_t blocks = want < cap ? want : cap;
            // variant 6: 8 counter words when the wave count is a multiple of 8 (else 1), in the upper half of the argument
            const uint32_t words = (variant == FUSED_DUAL_DEVWIDE && ((blocks * wpb) & 7ull) == 0ull) ? 8u : 1u;
            if (batch > 0xFFFFu) batch = 0xFFFFu;
            e->fn<<<dim3((unsigned)blocks), dim3((unsigned)(64 * wpb)), 0, s>>>(
                a.images, n_main, (const i32x4 *)a.frags, a.n_classes, a.cls, a.logits, a.src_wrap, a.work, a.idle, batch | (words << 16));
            hipError_t err = hipGetLastError();
            if (err != hipSuccess) return err;
        }
        if (a.n == n_main) return hipSuccess;
        BnmFusedArgs t = a;
        t.images = a.images + n_main * 256ull;
        t.n = a.n - n_main;
        t.cls = a.cls + n_main;
        t.logits = a.logits ? a.logits + n_main * a.n_classes : nullptr;
        t.src_wrap = 0;
        return bnmk_fused_fc(sh, FUSED_LDSDMA2, grid_blocks, t, s);
    }
    uint64_t n_tiles = (a.n + 31ull) / 32ull;
    uint64_t want = (n_tiles + FUSED_WPB - 1) / FUSED_WPB;
    // persistent grid: 8 resident waves per CU (2 workgroups x 4 waves; VGPRs and LDS allow no more)
    uint64_t cap = grid_blocks > 0 ? (uint64_t)grid_blocks : (uint64_t)bnm_num_cus() * 2ull;
    unsigned blocks = (unsigned)(want < cap ? want : cap);
    e->fn<<<dim3(blocks), dim3(64 * FUSED_WPB), 0, s>>>(a.images, a.n, (const i32x4 *)a.frags, a.n_classes, a.cls, a.logits, a.src_wrap,
                                                       nullptr, nullptr, 0);
    return hipGetLastError();
}

