// Generic fused whole-model FC kernel (int8 MFMA): ANY exporter-producible layer widths up to 256 per layer, 3 or 4 FC
// layers, any class count <= 256, input rows of 64/128/256/512 bytes, every int8-representable codec plus FP1.3.0's
// +128 (second weight plane).  gfx950 (CDNA4 / MI355X) only.  Reference semantics: BitNetMCU_inference.c:23-72
// (ReLUNorm), :88-208 (processfclayer); schedule BitNetMCU_MNIST_dll.c:48-121; widths are free parameters of the
// reference's model zoo (models.py:62-84; the documented 12 KB family docs/documentation.md:169-183).
//
// Same formulation as bnm_fused_fc.hip (Y^T = W * X^T on v_mfma_i32_32x32x32_i8, ReLUNorm output packed straight into
// the next layer's B operand).  What makes it shape-free:
//   * the weight fragments live in LDS (one copy per workgroup = per CU, staged once per launch) and are read as A
//     operands with lane-linear ds_read_b128 - no weight VGPRs, so no shape can spill;
//   * the layer widths are RUN-TIME values and nothing is padded: a layer of M 32-row tiles over K K-steps executes exactly
//     M*K MFMAs and its fragment image holds exactly M*K KiB.  Per layer the kernel switches (wave-uniformly) on the tile
//     count M - inside a case every accumulator index is a compile-time constant - and walks the K-steps in source order
//     with an early exit behind each one (K-steps are compile-time positions, so the packed activations stay in registers).
//     The fragments of a layer are stored K-step major ([plane][K-step][tile]): with M fixed inside a case, every fragment
//     address is the lane's base plus an immediate offset.  Only the upper bound MMAX of tiles per layer (2 / 4 / 8) is a
//     template parameter: it sizes the register arrays;
//   * T = 1 or 2 image tiles per wave per iteration.  With T = 2 every fragment read from LDS feeds TWO MFMAs (one per
//     tile): half the LDS traffic per image - at one fragment read per MFMA the matrix cores of a CU ask the LDS for exactly
//     its peak 128 B per clock - and two independent dependency chains inside a wave;
//   * the tile buffers are refilled with the wave's next unit by LDS-DMA as soon as the layer-1 B operands sit in registers,
//     so the load is in flight for the whole of the unit's arithmetic.
// LDS image of a tile: row r (image) at r*ROW, 16-byte slot c' holds global slot c = c' ^ mask(r) with
// mask(r) = (r >> (4-p)) & (S-1) for S = ROW/16 = 2^p slots per row (p <= 4) and r & 15 for p = 5: the ds_read_b128
// lane groups (MI355X_MICROARCH.md, LDS table) then hit 16 distinct 16-byte bank groups for every supported ROW.  The
// swizzle is applied on the SOURCE address of the DMA (its LDS destination is lane-linear).
#pragma once
#include <mutex>
#include <set>
#include <utility>
#include "bnm_fused_tile.hpp"
#include "bnm_fused_math.hpp"

// fragment reads in flight ahead of their MFMAs (a translation unit may set it before including this header)
#ifndef BNM_FRAG_DEPTH
#define BNM_FRAG_DEPTH 4
#endif

namespace {

// Diagnostic build only (build.py --diag-timing): shader-clock stamps at the phase boundaries of the uniform path, summed per wave
// and written where the caller's `logits` point (profiles/r05/r05_generic_phases.py).  Nothing of it exists in the product build.
#ifdef BNM_DIAG_TIMING
struct PhaseStamps {
    uint64_t last;
    uint32_t sum[12];
    BNM_DEVICE void start() { last = __builtin_amdgcn_s_memtime(); }
    BNM_DEVICE void tick(int k) {
        __builtin_amdgcn_sched_barrier(0);      // (a wave issues in order: the stamp is taken once everything before it has been issued)
        const uint64_t t = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_sched_barrier(0);
        sum[k] += (uint32_t)(t - last);
        last = t;
    }
};
#define BNM_ST_PARAM , PhaseStamps &st
#define BNM_ST_ARG , st
#define BNM_TICK(k) st.tick(k)
#else
#define BNM_ST_PARAM
#define BNM_ST_ARG
#define BNM_TICK(k)
#endif

// wave-uniform values that the compiler may nevertheless have placed in VGPRs: back to SGPRs for the "s" constraints
BNM_DEVICE uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
BNM_DEVICE const int8_t *uni(const int8_t *p) {
    uint64_t v = (uint64_t)p;
    return (const int8_t *)(((uint64_t)uni((uint32_t)(v >> 32)) << 32) | uni((uint32_t)v));
}
// up to four 1 KiB LDS-DMA pieces that share one M0 setting (the instruction offset is added to both addresses)
#define BNM_DMA_GROUP(BODY, ...)                                                                  \
    asm volatile("s_mov_b32 %0, m0\n\t"                                                           \
                 "s_mov_b32 m0, %1\n\t"                                                           \
                 "s_nop 0\n\t" BODY "s_mov_b32 m0, %0"                                            \
                 : "=&s"(keep)                                                                    \
                 : "s"(lds), "s"(base), __VA_ARGS__                                               \
                 : "memory")
BNM_DEVICE void dma_group4(uint32_t lds_, const int8_t *base_, uint32_t v0, uint32_t v1, uint32_t v2, uint32_t v3) {
    uint32_t keep;
    const uint32_t lds = uni(lds_);
    const int8_t *base = uni(base_);
    BNM_DMA_GROUP("global_load_lds_dwordx4 %3, %2 nt\n\t"
                  "global_load_lds_dwordx4 %4, %2 offset:1024 nt\n\t"
                  "global_load_lds_dwordx4 %5, %2 offset:2048 nt\n\t"
                  "global_load_lds_dwordx4 %6, %2 offset:3072 nt\n\t",
                  "v"(v0), "v"(v1), "v"(v2), "v"(v3));
}
BNM_DEVICE void dma_group2(uint32_t lds_, const int8_t *base_, uint32_t v0, uint32_t v1) {
    uint32_t keep;
    const uint32_t lds = uni(lds_);
    const int8_t *base = uni(base_);
    BNM_DMA_GROUP("global_load_lds_dwordx4 %3, %2 nt\n\t"
                  "global_load_lds_dwordx4 %4, %2 offset:1024 nt\n\t",
                  "v"(v0), "v"(v1));
}
BNM_DEVICE void dma_group1(uint32_t lds_, const int8_t *base_, uint32_t v0) {
    uint32_t keep;
    const uint32_t lds = uni(lds_);
    const int8_t *base = uni(base_);
    BNM_DMA_GROUP("global_load_lds_dwordx4 %3, %2 nt\n\t", "v"(v0));
}
// before overwriting a buffer: this wave's own ds_reads of it must have returned (hipcc only waits before the USE of
// a ds_read's result); the s_nop covers a VALU-written SGPR feeding M0
BNM_DEVICE void retire_lds_reads() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 4" ::: "memory"); }
// a point no memory operation may be moved across by the compiler (no instruction is emitted): keeps a fragment read that is
// issued ahead of a branch from being sunk into the block that uses it
BNM_DEVICE void pin_memory_order() { asm volatile("" ::: "memory"); }

template <int ROW>
struct RowGeom {
    static constexpr int SLOTS = ROW / 16;
    static constexpr int P = ROW == 64 ? 2 : ROW == 128 ? 3 : ROW == 256 ? 4 : 5;
    static_assert(ROW == 64 || ROW == 128 || ROW == 256 || ROW == 512, "row bytes must be 64, 128, 256 or 512");
    static constexpr int TILE = 32 * ROW;
    static constexpr int PIECES = TILE / 1024;
    static constexpr int ROWS_PER_PIECE = 1024 / ROW;
    // distinct per-piece source offsets: xmask() below has this period
    static constexpr int NV = ROW == 64 ? 1 : ROW == 128 ? 2 : ROW == 256 ? 4 : 8;
    BNM_DEVICE static uint32_t mask(uint32_t r) { return P == 5 ? (r & 15u) : ((r >> (4 - P)) & (uint32_t)(SLOTS - 1)); }
    // what piece t XORs into the lane's piece-relative source offset (see the derivation in DESIGN.md §4.1b)
    static constexpr uint32_t xmask(int t) { return P == 5 ? ((32u * t) & 0xF0u) : ((64u * t) & (uint32_t)(ROW - 16)); }
};

// ---- MFMA streams ------------------------------------------------------------------------------------------------
// Items are (plane p, K-step s, tile m), K-step outermost inside a plane so that consecutive MFMAs go to DIFFERENT
// accumulators (no dependent-accumulator stalls); every item reads ONE fragment and issues T MFMAs (one per image tile of the
// wave).  Fragment reads are issued DEPTH items ahead of their MFMAs in source order and the order is pinned with
// sched_group_barrier - left to itself hipcc put every ds_read directly in front of its MFMA and paid the full LDS latency per
// item.  Fragment (p, s, m) of a layer with MT tiles and KTOT K-steps sits at ((p*KTOT + s)*MT + m) KiB, lane-linear.

// NS compile-time K-steps S0 .. S0+NS-1 of KTOT: layer 1 (or a chunk of its K-steps), and every layer of the uniform fast path
template <int MT, int NS, int S0, int KTOT, int SP, bool INIT, int T, int NB = NS>
BNM_DEVICE void mma_l1(const char *a, const i32x4 (&b)[T][NB], i32x16 (&acc)[T][MT]) {
    static_assert(NS <= NB, "operand array too short");
    constexpr int N = SP * NS * MT;
    constexpr int DEPTH = N < BNM_FRAG_DEPTH ? N : BNM_FRAG_DEPTH;
    auto frag = [&](auto I) -> i32x4 {
        constexpr int i = decltype(I)::value;
        constexpr int pp = i / (NS * MT), ss = (i % (NS * MT)) / MT, mm = i % MT;
        return *(const i32x4 *)(a + ((pp * KTOT + S0 + ss) * MT + mm) * 1024);
    };
    __builtin_amdgcn_sched_barrier(0);      // the stream is its own scheduling region
    i32x4 ring[DEPTH];
    static_for<0, DEPTH>([&](auto I) { ring[decltype(I)::value] = frag(I); });
    static_for<0, N>([&](auto I) {
        constexpr int i = decltype(I)::value;
        constexpr int pp = i / (NS * MT), ss = (i % (NS * MT)) / MT, mm = i % MT;
#pragma unroll
        for (int t = 0; t < T; t++) {
            const i32x16 c = (INIT && pp == 0 && ss == 0) ? zero16() : acc[t][mm];
            acc[t][mm] = __builtin_amdgcn_mfma_i32_32x32x32_i8(ring[i % DEPTH], b[t][ss], c, 0, 0, 0);
        }
        if constexpr (i + DEPTH < N) ring[i % DEPTH] = frag(std::integral_constant<int, i + DEPTH>{});
    });
    // pin the issue order: DEPTH fragment reads, then { T MFMAs, one read } groups (0x100 = DS read, 0x008 = MFMA)
    __builtin_amdgcn_sched_group_barrier(0x100, DEPTH, 0);
    static_for<0, N>([&](auto I) {
        __builtin_amdgcn_sched_group_barrier(0x008, T, 0);
        if constexpr (decltype(I)::value + DEPTH < N) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    });
    __builtin_amdgcn_sched_barrier(0);
}

// layers 2..4, one plane: K (run-time, 1 .. MMAX) K-steps.  K-step S is a compile-time position (its B operands are the
// registers in[.][S]); behind every K-step the chain leaves when K is reached.  The ring is primed by the caller; reads run
// DEPTH items ahead across the exits (an exit drops at most DEPTH reads of fragments that exist but are not needed - or of
// whatever follows the layer's last fragment in LDS: the next layer's fragments or the tile buffers, never out of bounds).
template <int S, int MT, int T, int MMAX, int DEPTH, bool INIT>
BNM_DEVICE void hidden_steps(const char *a, uint32_t K, const i32x4 (&in)[T][MMAX], i32x16 (&acc)[T][MT], i32x4 (&ring)[DEPTH]) {
    __builtin_amdgcn_sched_barrier(0);
    static_for<0, MT>([&](auto MI) {
        constexpr int m = decltype(MI)::value, i = S * MT + m;
#pragma unroll
        for (int t = 0; t < T; t++) {
            const i32x16 c = (INIT && S == 0) ? zero16() : acc[t][m];
            acc[t][m] = __builtin_amdgcn_mfma_i32_32x32x32_i8(ring[i % DEPTH], in[t][S], c, 0, 0, 0);
        }
        if constexpr (i + DEPTH < MMAX * MT) ring[i % DEPTH] = *(const i32x4 *)(a + (i + DEPTH) * 1024);
    });
    static_for<0, MT>([&](auto MI) {
        __builtin_amdgcn_sched_group_barrier(0x008, T, 0);
        if constexpr (S * MT + decltype(MI)::value + DEPTH < MMAX * MT) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    });
    __builtin_amdgcn_sched_barrier(0);
    pin_memory_order();
    if constexpr (S + 1 < MMAX) {
        if (K > (uint32_t)(S + 1)) hidden_steps<S + 1, MT, T, MMAX, DEPTH, INIT>(a, K, in, acc, ring);
    }
}

template <int MT, int SP, int T, int MMAX>
BNM_DEVICE void hidden_mma(const char *a, uint32_t K, const i32x4 (&in)[T][MMAX], i32x16 (&acc)[T][MT]) {
    constexpr int DEPTH = MMAX * MT < BNM_FRAG_DEPTH ? MMAX * MT : BNM_FRAG_DEPTH;
    static_for<0, SP>([&](auto PI) {
        constexpr int p = decltype(PI)::value;
        const char *ap = a;
        if constexpr (p == 1) ap = a + K * (uint32_t)(MT * 1024);      // the second plane follows the first one's K*MT fragments
        i32x4 ring[DEPTH];
        static_for<0, DEPTH>([&](auto I) { ring[decltype(I)::value] = *(const i32x4 *)(ap + decltype(I)::value * 1024); });
        hidden_steps<0, MT, T, MMAX, DEPTH, p == 0>(ap, K, in, acc, ring);
    });
}

// hidden layer: MFMAs + ReLUNorm (M tiles, K K-steps, both run-time), IN PLACE: the packed activations act[.][0..K) are the B
// operands, act[.][0..M) receive the layer's outputs once the last MFMA has been issued.  (With separate in / out arrays the
// cases' outputs met in PHI nodes whose undefined inputs hipcc filled with copies of `in`: a dozen v_mov per layer and tile.)
template <int MMAX, int SP, bool DBL, int T>
BNM_DEVICE void hidden_layer(const char *smem, uint32_t lane16, uint32_t off, uint32_t M, uint32_t K, i32x4 (&act)[T][MMAX], int h) {
    static_for<1, MMAX + 1>([&](auto MI) {
        constexpr int mt = decltype(MI)::value;
        if (M == (uint32_t)mt) {
            i32x16 acc[T][mt];
            hidden_mma<mt, SP, T, MMAX>(smem + (off + lane16), K, act, acc);
#pragma unroll
            for (int t = 0; t < T; t++) relunorm_pack<mt, DBL, MMAX>(acc[t], act[t], h);
        }
    });
}

// first-maximum argmax of a classifier layer's accumulators (+ logits)
template <int MT, int T>
BNM_DEVICE void classify(const i32x16 (&acc)[T][MT], int h, int j, int lane, uint32_t (&cls)[T], int32_t *logits_out, int32_t *stage,
                         uint64_t first_img, uint64_t n, uint32_t n_classes, bool few_classes) {
    // up to 16 classes sit in the first two register groups of tile 0 (rows 0..15): the common case examines 8 registers
    // instead of 16 per tile (wave-uniform branch)
    if (few_classes) {
#pragma unroll
        for (int t = 0; t < T; t++) cls[t] = argmax_rows<MT, 2>(acc[t], h);
    } else {
#pragma unroll
        for (int t = 0; t < T; t++) cls[t] = argmax_rows<MT, 0>(acc[t], h);
    }
    if (logits_out) {
#pragma unroll
        for (int t = 0; t < T; t++) {
            const uint64_t tile_first = first_img + 32ull * (uint64_t)t;
            if (tile_first >= n) continue;
            int32_t *tile_dst = logits_out + tile_first * n_classes;
            if (stage != nullptr && tile_first + 32ull <= n) {
                // a whole tile through the staging area: contiguous nontemporal 16 B/lane stores, every line written whole
                store_logits_tile<MT, 0, 0>(acc[t], stage, tile_dst, j, h, lane, n_classes);
            } else if (tile_first + (uint64_t)j < n) {
                store_logits<MT>(acc[t], tile_dst + (uint32_t)j * n_classes, h, n_classes);
            }
        }
    }
}

// classifier layer, general path: MFMAs (M tiles, K K-steps, both run-time) + classify
template <int MMAX, int SP, int T>
BNM_DEVICE void final_layer(const char *smem, uint32_t lane16, uint32_t off, uint32_t M, uint32_t K, const i32x4 (&in)[T][MMAX], int h,
                            int j, int lane, uint32_t (&cls)[T], int32_t *logits_out, int32_t *stage, uint64_t first_img, uint64_t n,
                            uint32_t n_classes, bool few_classes) {
    static_for<1, MMAX + 1>([&](auto MI) {
        constexpr int mt = decltype(MI)::value;
        if (M == (uint32_t)mt) {
            i32x16 acc[T][mt];
            hidden_mma<mt, SP, T, MMAX>(smem + (off + lane16), K, in, acc);
            classify<mt, T>(acc, h, j, lane, cls, logits_out, stage, first_img, n, n_classes, few_classes);
        }
    });
}

// ---- the uniform fast path -----------------------------------------------------------------------------------------
// When every hidden layer has the same tile count MT (the shapes the reference documents: 64-64-64, 96-96-96, 128-128-112,
// 160-160-160) and the classifier fits one tile, everything behind layer 1 is ONE straight-line piece of code per MT: K-steps
// are compile-time counts (no exits), no case boundaries between the layers (no PHI copies of the packed activations: the
// general path spends 7-26 v_mov per layer on them), and hipcc schedules across layers as in the shape-specialised kernels.
template <int MT, int MMAX, int SP, bool DBL, int T>
BNM_DEVICE void uniform_layer(const char *a, i32x4 (&act)[T][MMAX], int h BNM_ST_PARAM, int k0 = 0) {
    i32x16 acc[T][MT];
    mma_l1<MT, MT, 0, MT, SP, true, T, MMAX>(a, act, acc);
    BNM_TICK(k0);
#pragma unroll
    for (int t = 0; t < T; t++) relunorm_pack<MT, DBL, MMAX>(acc[t], act[t], h);
    BNM_TICK(k0 + 1);
}
template <int MT, int MMAX, int SP, bool DBL, int T>
BNM_DEVICE void uniform_tail(const char *smem, uint32_t lane16, const BnmGenericDesc &d, i32x4 (&act)[T][MMAX], int h, int j, int lane,
                             uint32_t (&cls)[T], int32_t *logits_out, int32_t *stage, uint64_t first_img, uint64_t n, uint32_t n_classes,
                             bool few_classes BNM_ST_PARAM) {
    uniform_layer<MT, MMAX, SP, DBL, T>(smem + (d.frag_off[1] + lane16), act, h BNM_ST_ARG, 3);
    uint32_t off_last = d.frag_off[2];
    if (d.M[3]) {
        uniform_layer<MT, MMAX, SP, DBL, T>(smem + (d.frag_off[2] + lane16), act, h BNM_ST_ARG, 5);
        off_last = d.frag_off[3];
    }
    i32x16 acc[T][1];
    mma_l1<1, MT, 0, MT, SP, true, T, MMAX>(smem + (off_last + lane16), act, acc);
    BNM_TICK(7);
    classify<1, T>(acc, h, j, lane, cls, logits_out, stage, first_img, n, n_classes, few_classes);
    BNM_TICK(8);
}

}  // namespace

// WPS: waves per SIMD the register budget is compiled for (the workgroup holds up to 4*WPS waves, ONE workgroup per CU).
template <int MMAX, int KT0, int SP, bool DBL, int T, int WPS>
__global__ __launch_bounds__(256 * WPS) void fused_fc_generic_kernel(const int8_t *__restrict__ images, uint64_t n,
                                                                     const i32x4 *__restrict__ frags, BnmGenericDesc d,
                                                                     uint32_t *__restrict__ cls_out,
                                                                     int32_t *__restrict__ logits_out, uint32_t *__restrict__ counter,
                                                                     uint32_t batch_arg) {
    using G = RowGeom<32 * KT0>;
    const uint32_t batch = batch_arg & 0xFFFFu;
    constexpr int ROW = 32 * KT0;
    // layer-1 K-steps whose B operands are held in VGPRs at a time (x T tiles x 4 registers): all of a 256-byte row with one
    // tile per wave, half of it with two (the refill then starts after the first half's MFMAs)
    // (4-tile class with two tiles: two K-steps at a time - its two tiles' 128 accumulator registers leave room for no more)
    constexpr int KC = KT0 * T <= 8 ? KT0 : (MMAX >= 4 && T == 2) ? 2 : 8 / T;
    constexpr int UNIT = T * G::TILE;              // bytes of the T consecutive tiles a wave handles per iteration
    static_assert(T == 1 || T == 2, "one or two tiles per wave");
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t nwaves = blockDim.x >> 6;

    // ---- weights: fragment image global -> LDS, once per workgroup -------------------------------------
    for (uint32_t o = threadIdx.x * 16u; o < d.w_bytes; o += blockDim.x * 16u) *(i32x4 *)(smem + o) = frags[o >> 4];
    __syncthreads();

    const uint32_t tile_off = d.w_bytes + wave * (uint32_t)UNIT;                       // this wave's T tile buffers
    const uint32_t tile_lds = (uint32_t)(size_t)(__attribute__((address_space(3))) char *)smem + tile_off;
    // logits staging area of the wave (2 KiB behind all tile buffers) when the launcher reserved one
    int32_t *const stage = d.stage ? (int32_t *)(smem + d.w_bytes + nwaves * (uint32_t)UNIT + wave * 2048u) : nullptr;
    // Lane constants that stay in registers across the persistent loop: the lane id, the B-operand base, the fragment base and
    // the NV distinct DMA source offsets (DMA piece t, lane l: LDS byte 1024t + 16l = row t*RPP + rl, slot c'; source slot
    // c = c' ^ mask(row)).  The loop body works on opaque per-iteration copies of them: left to itself hipcc hoists dozens of
    // DERIVED operand / fragment addresses out of the loop and spills them.
    // B operand of K-step s: image j, global slot 2s+h -> LDS slot (2s+h) ^ mask(j): XOR 32*s into the byte offset rd_off
    const uint32_t lane16 = 16u * (uint32_t)lane;
    const uint32_t rd_off = tile_off + (uint32_t)(lane & 31) * (uint32_t)ROW + 16u * ((uint32_t)(lane >> 5) ^ G::mask((uint32_t)(lane & 31)));
    // rows of up to 256 bytes: the K-steps' B-operand addresses themselves stay in registers (no XOR per K-step and tile)
    // (not in the 4-tile class with two tiles: eight loop-carried registers it does not have)
    constexpr bool PRE_RD = KT0 <= 8 && !(MMAX >= 4 && T == 2);
    uint32_t rda[PRE_RD ? KT0 : 1];
    if constexpr (PRE_RD) {
#pragma unroll
        for (int s = 0; s < KT0; s++) rda[s] = rd_off ^ (32u * (uint32_t)s);
    }
    uint32_t voff[G::NV];
    {
        const uint32_t rl0 = (16u * (uint32_t)lane) / (uint32_t)ROW, cs0 = (uint32_t)lane & (uint32_t)(G::SLOTS - 1);
#pragma unroll
        for (int k = 0; k < G::NV; k++) voff[k] = (rl0 * (uint32_t)ROW + 16u * (cs0 ^ G::mask(rl0))) ^ G::xmask(k);
    }

    const uint32_t n_tiles = (uint32_t)((n + 31ull) >> 5);       // the launcher refuses n >= 2^36
    const uint32_t n_units = (n_tiles + (uint32_t)(T - 1)) / (uint32_t)T;
    auto dma_unit = [&](uint32_t u) {
        const int8_t *base = images + (uint64_t)u * (uint64_t)UNIT;
        const uint64_t first = (uint64_t)u * (uint64_t)(32 * T);
        if (first + (uint64_t)(32 * T) <= n) {
            constexpr int NP = T * G::PIECES;
            static_for<0, (NP + 3) / 4>([&](auto GI) {
                constexpr int t0 = 4 * decltype(GI)::value;
                if constexpr (NP - t0 >= 4)
                    dma_group4(tile_lds + 1024u * t0, base + 1024 * t0, voff[t0 % G::NV], voff[(t0 + 1) % G::NV], voff[(t0 + 2) % G::NV],
                               voff[(t0 + 3) % G::NV]);
                else if constexpr (NP - t0 == 2)
                    dma_group2(tile_lds + 1024u * t0, base + 1024 * t0, voff[t0 % G::NV], voff[(t0 + 1) % G::NV]);
                else
                    dma_group1(tile_lds + 1024u * t0, base + 1024 * t0, voff[t0 % G::NV]);
            });
        } else {
            // ragged end: rows past the last image re-read it (never out of bounds); a tile wholly past the end reads the last
            // image 32 times and stores nothing
            const uint32_t nv = (uint32_t)(n - first);                                 // valid rows from `first` on (1 .. 32 T - 1)
            uint32_t l = (uint32_t)lane;
            asm volatile("" : "+v"(l));
            const uint32_t rl = (16u * l) / (uint32_t)ROW, cs = l & (uint32_t)(G::SLOTS - 1);
            static_for<0, T * G::PIECES>([&](auto TI) {
                constexpr int tt = decltype(TI)::value;
                const uint32_t r = (uint32_t)(tt * G::ROWS_PER_PIECE) + rl;            // row within the unit
                const uint32_t src = r < nv ? r : nv - 1u;
                dma_group1(tile_lds + 1024u * tt, base, src * (uint32_t)ROW + 16u * (cs ^ G::mask(r & 31u)));
            });
        }
    };

    // ---- units in batches of `batch` consecutive ones: a wave's first batch is static, every later one comes from the
    // device-wide counter (work_take_*; the wave that runs out of batch asks while it still has one unit to go).
    // The counter is split into `words` words (upper half of the argument: 8 when the wave count is a multiple of 8, else 1):
    // wave w takes from word w % words, which hands out the batches g = t * words + w % words - every word is shared by waves
    // of all CUs, so the balance stays device-wide while a word sees an eighth of the takes (work distribution).
    const uint32_t total_waves = gridDim.x * nwaves, wave_id = blockIdx.x * nwaves + wave;
    const uint32_t words = batch_arg >> 16, wshift = (uint32_t)__builtin_ctz(words | 0x100u);
    const uint32_t my_word = wave_id & (words - 1u), first_dyn = total_waves >> wshift;
    uint32_t unit = wave_id * batch, left = batch - 1u;      // `left`: units of the batch after this one
    uint32_t taken = 0;
    if (unit < n_units) dma_unit(unit);

    const uint32_t M1 = d.M[0], M2 = d.M[1], M3 = d.M[2], M4 = d.M[3];
    const bool few_classes = d.n_classes <= 16u;
    // the uniform fast path (see uniform_tail): equal tile counts in all hidden layers, near the class's maximum, one classifier tile
    const bool uniform = M2 == M1 && (M4 ? (M3 == M1 && M4 == 1u) : M3 == 1u) && M1 + 1u >= (uint32_t)MMAX;
    // Class ids leave one iteration LATE, right behind the refill: the loop's top waits for vmcnt(0) (the tile), and a store
    // issued at the end of the body would make that wait sit on its write acknowledgement every iteration.  Every class word is
    // written exactly once (bnm_infer_host's latency path polls them), so the first iteration stores nothing.
    uint32_t cls_prev = 0, unit_prev = 0;
    bool pend = false;
    auto flush_cls = [&]() {
        if (!pend) return;
        // T = 2: lanes 0..31 hold tile 0's ids, lanes 32..63 tile 1's - one 256-byte store per unit; T = 1: the lower half stores
        uint32_t l = (uint32_t)lane;
        asm volatile("" : "+v"(l));
        const uint64_t img = (uint64_t)unit_prev * (uint64_t)(32 * T) + (uint64_t)(T == 2 ? l : (l & 31u));
        if (img < n && (T == 2 || l < 32u)) __builtin_nontemporal_store(cls_prev, cls_out + img);
    };
#ifdef BNM_DIAG_TIMING
    PhaseStamps st{};
    int32_t *const records = logits_out;
    logits_out = nullptr;
    uint32_t iters = 0;
    st.start();
#endif
    while (unit < n_units) {
        if (left == 0u) work_take_issue(taken, counter + 16u * my_word, 1u);
        BNM_TICK(9);                 // loop overhead behind the previous tile's last stamp
        bnm_wait_vmcnt<0>();
        BNM_TICK(0);                 // the wait for the tile
        uint32_t next_unit = unit + 1u, next_left = left - 1u;
        // per-iteration opaque copies of the three lane values the arithmetic starts from (one v_mov each): what is derived from
        // them inside the iteration cannot be hoisted out of the persistent loop
        uint32_t lv = (uint32_t)lane, rd = rd_off, l16 = lane16;
        asm volatile("" : "+v"(lv), "+v"(rd), "+v"(l16));
        const int j = (int)(lv & 31u), h = (int)(lv >> 5);
        i32x4 act[T][MMAX];       // packed layer outputs = the next layer's B operands (updated in place, layer by layer)
        const uint64_t first_img = (uint64_t)unit * (uint64_t)(32 * T);
        // per-iteration copy: keeps the (row < n_classes) predicates of every accumulator register of every case from being
        // hoisted out of the persistent loop as hundreds of live 64-bit masks
        uint32_t nc = d.n_classes;
        asm volatile("" : "+v"(nc));
        uint32_t cls[T];
#pragma unroll
        for (int t = 0; t < T; t++) cls[t] = 0;
        bool done = false;
        // ---- layer 1: B operands from the tile buffers, KC K-steps at a time; the buffers are refilled with the wave's
        // next unit as soon as the last operand has been read (the load is then in flight for the rest of the iteration)
        static_for<1, MMAX + 1>([&](auto MI) {
            constexpr int mt = decltype(MI)::value;
            if (M1 == (uint32_t)mt) {
                i32x16 acc[T][mt];
                static_for<0, KT0 / KC>([&](auto CI) {
                    constexpr int ch = decltype(CI)::value;
                    i32x4 b0[T][KC];
#pragma unroll
                    for (int t = 0; t < T; t++)
#pragma unroll
                        for (int s = 0; s < KC; s++)
                            if constexpr (PRE_RD) b0[t][s] = *(const i32x4 *)(smem + (rda[ch * KC + s] + (uint32_t)(t * G::TILE)));
                            else b0[t][s] = *(const i32x4 *)(smem + ((rd + (uint32_t)(t * G::TILE)) ^ (32u * (uint32_t)(ch * KC + s))));
                    if constexpr (ch == KT0 / KC - 1) {
                        retire_lds_reads();
                        if (left == 0u) {
                            work_take_wait(taken);
                            next_unit = (((first_dyn + taken) << wshift) + my_word) * batch;
                            next_left = batch - 1u;
                        }
                        if (next_unit < n_units) dma_unit(next_unit);
                        flush_cls();
                    }
                    mma_l1<mt, KC, ch * KC, KT0, SP, ch == 0, T>(smem + (d.frag_off[0] + l16), b0, acc);
                });
                BNM_TICK(1);
#pragma unroll
                for (int t = 0; t < T; t++) relunorm_pack<mt, DBL, MMAX>(acc[t], act[t], h);
                BNM_TICK(2);
                if constexpr (mt >= MMAX - 1) {
                    if (uniform) {      // all hidden layers have mt tiles, one classifier tile: straight-line code from here on
                        uniform_tail<mt, MMAX, SP, DBL, T>(smem, l16, d, act, h, j, lane, cls, logits_out, stage, first_img, n, nc, few_classes BNM_ST_ARG);
                        done = true;
                    }
                }
            }
        });
        if (!done) {
            hidden_layer<MMAX, SP, DBL, T>(smem, l16, d.frag_off[1], M2, M1, act, h);
            uint32_t m_last = M3, k_last = M2, off_last = d.frag_off[2];
            if (M4) {
                hidden_layer<MMAX, SP, DBL, T>(smem, l16, d.frag_off[2], M3, M2, act, h);
                m_last = M4; k_last = M3; off_last = d.frag_off[3];
            }
            final_layer<MMAX, SP, T>(smem, l16, off_last, m_last, k_last, act, h, j, lane, cls, logits_out, stage, first_img, n, nc, few_classes);
        }
        // both halves of the wave hold every tile's result: T = 2 keeps tile 0's ids in lanes 0..31 and tile 1's in lanes 32..63
        // (one 256-byte store per unit), T = 1 stores from the lower half
        if constexpr (T == 2) cls_prev = h ? cls[1] : cls[0];   // (h: this iteration's copy)
        else cls_prev = cls[0];
        unit_prev = unit;
        pend = true;
        unit = next_unit;
        left = next_left;
#ifdef BNM_DIAG_TIMING
        iters++;
#endif
    }
#ifdef BNM_DIAG_TIMING
    if (records != nullptr && lane == 0) {
        int32_t *r = records + 16u * wave_id;
        for (int k = 0; k < 12; k++) r[k] = (int32_t)st.sum[k];
        r[12] = (int32_t)iters;
        r[13] = (int32_t)(__builtin_amdgcn_s_getreg((31 << 11) | 4));
        r[14] = (int32_t)wave;
    }
#endif
    flush_cls();
    bnm_wait_vmcnt<0>();   // no LDS-DMA may outlive the workgroup's LDS allocation
    work_block_leave_s(counter, total_waves);   // the last wave to leave puts the counter block back to all-zero
}

// ---- per-class launcher: each (tile class, tiles per wave) pair is its own translation unit (bnm_fused_generic_m{2,4,8}[_t2].hip)
// so that they compile side by side.  Instantiations per row length: doubled hidden weights; plain; plain with FP1.3.0's second
// weight plane.  T = 2 is instantiated for rows of 128 and 256 bytes (32 / 64 B-operand registers per wave).
// hipFuncSetAttribute costs about a microsecond per call: launch-bound callers pay it once per kernel and device
inline hipError_t bnm_generic_allow_big_lds(const void *fn) {
    static std::mutex mu;
    static std::set<std::pair<const void *, int>> done;
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
    std::lock_guard<std::mutex> g(mu);
    if (done.count({fn, dev})) return hipSuccess;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) done.insert({fn, dev});
    return e;
}

#define BNM_GENERIC_PICK(MMAX, K0, T)                                                                                       \
    if (kt0 == K0) {                                                                                                        \
        if (sp == 1 && dbl) fn = fused_fc_generic_kernel<MMAX, K0, 1, true, T, bnmk_generic_wps(MMAX, K0, 1, T)>;           \
        else if (sp == 1) fn = fused_fc_generic_kernel<MMAX, K0, 1, false, T, bnmk_generic_wps(MMAX, K0, 1, T)>;            \
        else if (sp == 2 && !dbl) fn = fused_fc_generic_kernel<MMAX, K0, 2, false, T, bnmk_generic_wps(MMAX, K0, 2, T)>;    \
    }
#define BNM_GENERIC_LAUNCHER_BEGIN(NAME)                                                                                    \
    hipError_t NAME(uint32_t kt0, uint32_t sp, bool dbl, unsigned blocks, unsigned threads, unsigned lds, hipStream_t s,     \
                    const int8_t *images, uint64_t n, const void *frags, const BnmGenericDesc &d, uint32_t *cls,            \
                    int32_t *logits, uint32_t *counter, uint32_t batch) {                                                    \
        typedef void (*fn_t)(const int8_t *, uint64_t, const i32x4 *, BnmGenericDesc, uint32_t *, int32_t *, uint32_t *, uint32_t); \
        fn_t fn = nullptr;
#define BNM_GENERIC_LAUNCHER_END                                                                                             \
        if (!fn) return hipErrorInvalidValue;                                                                                \
        if (!blocks) return hipSuccess;   /* probe: is there an instantiation? */                                            \
        /* opt in to > 64 KiB of dynamic LDS: a per-device function attribute, set once per (kernel, device) */               \
        if (hipError_t err = bnm_generic_allow_big_lds((const void *)fn); err != hipSuccess) return err;                     \
        fn<<<dim3(blocks), dim3(threads), lds, s>>>(images, n, (const i32x4 *)frags, d, cls, logits, counter, batch);        \
        return hipGetLastError();                                                                                            \
    }
#define BNM_GENERIC_LAUNCHER_T1(NAME, MMAX)                                                                                  \
    BNM_GENERIC_LAUNCHER_BEGIN(NAME)                                                                                         \
    BNM_GENERIC_PICK(MMAX, 2, 1) BNM_GENERIC_PICK(MMAX, 4, 1) BNM_GENERIC_PICK(MMAX, 8, 1) BNM_GENERIC_PICK(MMAX, 16, 1)     \
    BNM_GENERIC_LAUNCHER_END
// one row length per translation unit (the 8-tile class: a single unit with all four took 6.5 minutes to compile)
#define BNM_GENERIC_LAUNCHER_T1_K(NAME, MMAX, K0)                                                                            \
    BNM_GENERIC_LAUNCHER_BEGIN(NAME)                                                                                         \
    BNM_GENERIC_PICK(MMAX, K0, 1)                                                                                            \
    BNM_GENERIC_LAUNCHER_END
#define BNM_GENERIC_LAUNCHER_T2(NAME, MMAX)                                                                                  \
    BNM_GENERIC_LAUNCHER_BEGIN(NAME)                                                                                         \
    BNM_GENERIC_PICK(MMAX, 4, 2) BNM_GENERIC_PICK(MMAX, 8, 2)                                                                \
    BNM_GENERIC_LAUNCHER_END
// (Two tiles per wave exist for the 2-tile class only.  A wave addresses at most 256 VGPRs; two tiles' accumulators of a 4-tile
// layer are 128 of them, of an 8-tile layer all 256 - what does not fit goes to AGPRs, which ReLUNorm can only read through a
// v_accvgpr_read per value, or to scratch: the 4-tile instantiation spilled 22,000 registers.)
