// Generic fused whole-model FC kernel (int8 MFMA): ANY exporter-producible layer widths up to 256 per layer, 3 or 4 FC
// layers, any class count <= 256, input rows of 64/128/256/512 bytes, every int8-representable codec plus FP1.3.0's
// +128 (second weight plane).  gfx950 (CDNA4 / MI355X) only.  Reference semantics: BitNetMCU_inference.c:23-72
// (ReLUNorm), :88-208 (processfclayer); schedule BitNetMCU_MNIST_dll.c:48-121; widths are free parameters of the
// reference's model zoo (models.py:62-84).
//
// Same formulation as bnm_fused_fc.hip (Y^T = W * X^T on v_mfma_i32_32x32x32_i8, ReLUNorm output packed straight into
// the next layer's B operand) with two differences that remove the per-shape instantiation table:
//   * the weight fragments live in LDS (one copy per workgroup = per CU, staged once per launch) and are read as A
//     operands with lane-linear ds_read_b128 — no weight VGPRs, so no shape can spill, and the layer widths are
//     RUN-TIME values: per layer the kernel switches once on the K-step count and branches (wave-uniformly) per 32-row
//     output tile; only the upper bound MMAX of tiles per layer is a compile-time parameter (2 / 4 / 8);
//   * one 32-image tile buffer per wave: the tile's B operands are read into VGPRs and the buffer is refilled with the
//     wave's next tile by LDS-DMA at once, so the load is in flight for the whole of the tile's arithmetic.
// LDS image of a tile: row r (image) at r*ROW, 16-byte slot c' holds global slot c = c' ^ mask(r) with
// mask(r) = (r >> (4-p)) & (S-1) for S = ROW/16 = 2^p slots per row (p <= 4) and r & 15 for p = 5: the ds_read_b128
// lane groups (MI355X_MICROARCH.md, LDS table) then hit 16 distinct 16-byte bank groups for every supported ROW.  The
// swizzle is applied on the SOURCE address of the DMA (its LDS destination is lane-linear).
#pragma once
#include "bnm_fused_tile.hpp"
#include "bnm_fused_math.hpp"

namespace {

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

template <int ROW>
struct RowGeom {
    static constexpr int SLOTS = ROW / 16;
    static constexpr int P = ROW == 64 ? 2 : ROW == 128 ? 3 : ROW == 256 ? 4 : 5;
    static_assert(ROW == 64 || ROW == 128 || ROW == 256 || ROW == 512, "row bytes must be 64, 128, 256 or 512");
    static constexpr int TILE = 32 * ROW;
    static constexpr int PIECES = TILE / 1024;
    static constexpr int ROWS_PER_PIECE = 1024 / ROW;
    BNM_DEVICE static uint32_t mask(uint32_t r) { return P == 5 ? (r & 15u) : ((r >> (4 - P)) & (uint32_t)(SLOTS - 1)); }
    // what piece t XORs into the lane's piece-relative source offset (see the derivation in DESIGN.md §4.1b)
    static constexpr uint32_t xmask(int t) { return P == 5 ? ((32u * t) & 0xF0u) : ((64u * t) & (uint32_t)(ROW - 16)); }
};

// ---- layer blocks --------------------------------------------------------------------------------------------
// A layer is executed by ONE straight-line block chosen by a wave-uniform switch on its run-time tile count M (and on the
// padded K-step count KT): inside a block every accumulator index is a compile-time constant and the accumulators do not
// outlive it — only the packed int8 outputs (4 registers per tile) cross block boundaries.  (A first version branched per
// tile around updates of one shared accumulator array: hipcc answered with accumulator copies and spills.)
// A fragment (m, part, s) of a layer sits at off + ((m*SP + part)*KT + s) KiB of the LDS weight image, lane-linear; K-steps
// past the previous layer's real tile count hold zero weights, so whatever the matching B registers contain is harmless.
// The MFMA stream of (part of) a layer: items (part p, K-step s, tile m), K-step outermost so that consecutive MFMAs go to
// DIFFERENT accumulators (no dependent-accumulator stalls), fragment reads issued DEPTH items ahead of their MFMA in source
// order — hipcc keeps that order, whereas left to itself it put every ds_read directly in front of its MFMA and paid the
// full LDS latency fifty times per tile.  The fragment of (m, p, K-step s0+s) sits at a + ((m*SP + p)*KSTRIDE + s)*1024.
template <int MT, int NS, int KSTRIDE, int SP, bool INIT, int NB>
BNM_DEVICE void mma_stream(const char *a, const i32x4 (&b)[NB], i32x16 (&acc)[MT]) {
    static_assert(NS <= NB, "operand array too short");
    constexpr int N = SP * NS * MT;
    constexpr int DEPTH = N < 4 ? N : 4;
    auto frag = [&](auto I) -> i32x4 {
        constexpr int i = decltype(I)::value;
        constexpr int pp = i / (NS * MT), ss = (i % (NS * MT)) / MT, mm = i % MT;
        return *(const i32x4 *)(a + ((mm * SP + pp) * KSTRIDE + ss) * 1024);
    };
    __builtin_amdgcn_sched_barrier(0);      // the stream is its own scheduling region
    i32x4 ring[DEPTH];
    static_for<0, DEPTH>([&](auto I) { ring[decltype(I)::value] = frag(I); });
    static_for<0, N>([&](auto I) {
        constexpr int i = decltype(I)::value;
        constexpr int pp = i / (NS * MT), ss = (i % (NS * MT)) / MT, mm = i % MT;
        const i32x16 c = (INIT && pp == 0 && ss == 0) ? zero16() : acc[mm];
        acc[mm] = __builtin_amdgcn_mfma_i32_32x32x32_i8(ring[i % DEPTH], b[ss], c, 0, 0, 0);
        if constexpr (i + DEPTH < N) ring[i % DEPTH] = frag(std::integral_constant<int, i + DEPTH>{});
    });
    // pin the issue order: DEPTH fragment reads, then { one MFMA, one read } pairs (0x100 = DS read, 0x008 = MFMA)
    __builtin_amdgcn_sched_group_barrier(0x100, DEPTH, 0);
    static_for<0, N>([&](auto I) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if constexpr (decltype(I)::value + DEPTH < N) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    });
    __builtin_amdgcn_sched_barrier(0);
}

template <int MT, int KT, int SP, int NB>
BNM_DEVICE void block_mma(const char *smem, uint32_t lane16, uint32_t off, const i32x4 (&b)[NB], i32x16 (&acc)[MT]) {
    mma_stream<MT, KT, KT, SP, true, NB>(smem + (off + lane16), b, acc);
}

// K-step count a layer's fragments are padded to, given the previous layer's tile count
template <int MMAX>
constexpr int kpad(int m_prev) { return MMAX == 2 ? 2 : (m_prev <= MMAX / 2 ? MMAX / 2 : MMAX); }

// hidden layer: MFMAs + ReLUNorm, in -> out
template <int MMAX, int SP, bool DBL>
BNM_DEVICE void hidden_layer(const char *smem, uint32_t lane16, uint32_t off, uint32_t M, uint32_t KTP, const i32x4 (&in)[MMAX],
                             i32x4 (&out)[MMAX], int h) {
    constexpr int MSTEP = MMAX == 8 ? 2 : 1;     // 8-tile class: tile counts are rounded up to even (zero fragments)
    static_for<1, MMAX / MSTEP + 1>([&](auto MI) {
        constexpr int mt = decltype(MI)::value * MSTEP;
        static_for<0, (MMAX == 2 ? 1 : 2)>([&](auto KI) {
            constexpr int kt = MMAX == 2 ? 2 : (decltype(KI)::value == 0 ? MMAX / 2 : MMAX);
            if (M == (uint32_t)mt && KTP == (uint32_t)kt) {
                i32x16 acc[mt];
                block_mma<mt, kt, SP, MMAX>(smem, lane16, off, in, acc);
                relunorm_pack<mt, DBL, MMAX>(acc, out, h);
#pragma unroll
                for (int m = mt; m < MMAX; m++) out[m] = i32x4{0, 0, 0, 0};     // K-steps past the real tiles: zero weights meet zeros
            }
        });
    });
}

// classifier layer: MFMAs + first-maximum argmax (+ logits)
template <int MMAX, int SP>
BNM_DEVICE uint32_t final_layer(const char *smem, uint32_t lane16, uint32_t off, uint32_t M, uint32_t KTP, const i32x4 (&in)[MMAX],
                                int h, int32_t *logits_row, uint32_t n_classes) {
    constexpr int MSTEP = MMAX == 8 ? 2 : 1;
    uint32_t cls = 0;
    static_for<1, MMAX / MSTEP + 1>([&](auto MI) {
        constexpr int mt = decltype(MI)::value * MSTEP;
        static_for<0, (MMAX == 2 ? 1 : 2)>([&](auto KI) {
            constexpr int kt = MMAX == 2 ? 2 : (decltype(KI)::value == 0 ? MMAX / 2 : MMAX);
            if (M == (uint32_t)mt && KTP == (uint32_t)kt) {
                i32x16 acc[mt];
                block_mma<mt, kt, SP, MMAX>(smem, lane16, off, in, acc);
                cls = argmax_rows<mt, 0>(acc, h);
                if (logits_row) store_logits<mt>(acc, logits_row, h, n_classes);
            }
        });
    });
    return cls;
}

}  // namespace

// WPS: waves per SIMD the register budget is compiled for (the workgroup holds up to 4*WPS waves, ONE workgroup per CU).
// The descriptor's M[] are the tile counts the fragment image was BUILT for (already rounded as the class requires).
template <int MMAX, int KT0, int SP, bool DBL, int WPS>
__global__ __launch_bounds__(256 * WPS) void fused_fc_generic_kernel(const int8_t *__restrict__ images, uint64_t n,
                                                                     const i32x4 *__restrict__ frags, BnmGenericDesc d,
                                                                     uint32_t *__restrict__ cls_out,
                                                                     int32_t *__restrict__ logits_out, uint32_t *__restrict__ counter,
                                                                     uint32_t batch_arg) {
    using G = RowGeom<32 * KT0>;
    const uint32_t batch = batch_arg & 0xFFFFu;
    constexpr int ROW = 32 * KT0;
    constexpr int KC = KT0 < 8 ? KT0 : 8;          // layer-1 K-steps held in VGPRs at a time
    constexpr int MSTEP = MMAX == 8 ? 2 : 1;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t nwaves = blockDim.x >> 6;

    // ---- weights: fragment image global -> LDS, once per workgroup -------------------------------------
    for (uint32_t o = threadIdx.x * 16u; o < d.w_bytes; o += blockDim.x * 16u) *(i32x4 *)(smem + o) = frags[o >> 4];
    __syncthreads();

    const uint32_t tile_off = d.w_bytes + wave * (uint32_t)G::TILE;                    // this wave's tile buffer
    const uint32_t tile_lds = (uint32_t)(size_t)(__attribute__((address_space(3))) char *)smem + tile_off;
    // DMA piece t, lane l: LDS byte 1024t + 16l = row t*RPP + rl, slot c'; source slot c = c' ^ mask(row)
    // B operand of K-step s: image j, global slot 2s+h -> LDS slot (2s+h) ^ mask(j): XOR 32*s into the byte offset (rd_off)

    const uint32_t n_tiles = (uint32_t)((n + 31ull) >> 5);       // the launcher refuses n >= 2^36
    // (the per-use copies below keep hipcc from hoisting dozens of derived per-piece / per-K-step address registers out of
    // the persistent loop, where they would only raise the register pressure of the arithmetic)
    auto dma_tile = [&](uint32_t t) {
        const int8_t *base = images + (uint64_t)t * (uint64_t)G::TILE;
        const uint64_t first = (uint64_t)t << 5;
        uint32_t l = (uint32_t)lane;
        asm volatile("" : "+v"(l));
        const uint32_t rl = (16u * l) / (uint32_t)ROW, cs = l & (uint32_t)(G::SLOTS - 1);
        if (first + 32ull <= n) {
            const uint32_t voff0 = rl * (uint32_t)ROW + 16u * (cs ^ G::mask(rl));
            static_for<0, (G::PIECES + 3) / 4>([&](auto GI) {
                constexpr int g = decltype(GI)::value, t0 = 4 * g;
                if constexpr (G::PIECES - t0 >= 4)
                    dma_group4(tile_lds + 1024u * t0, base + 1024 * t0, voff0 ^ G::xmask(t0), voff0 ^ G::xmask(t0 + 1),
                               voff0 ^ G::xmask(t0 + 2), voff0 ^ G::xmask(t0 + 3));
                else if constexpr (G::PIECES - t0 == 2)
                    dma_group2(tile_lds + 1024u * t0, base + 1024 * t0, voff0 ^ G::xmask(t0), voff0 ^ G::xmask(t0 + 1));
                else
                    dma_group1(tile_lds + 1024u * t0, base + 1024 * t0, voff0 ^ G::xmask(t0));
            });
        } else {
            // ragged last tile: rows past the end re-read the last valid image (never out of bounds)
            const uint32_t nv = (uint32_t)(n - first);
            static_for<0, G::PIECES>([&](auto TI) {
                constexpr int tt = decltype(TI)::value;
                const uint32_t r = (uint32_t)(tt * G::ROWS_PER_PIECE) + rl;
                const uint32_t src = r < nv ? r : nv - 1u;
                dma_group1(tile_lds + 1024u * tt, base, src * (uint32_t)ROW + 16u * (cs ^ G::mask(r)));
            });
        }
    };

    // ---- tiles in batches of `batch` consecutive ones: a wave's first batch is static, every later one comes from the
    // device-wide counter (work_take_*; the wave that runs out of batch asks while it still has one tile to go)
    // The counter is split into `words` words (upper half of the argument: 8 when the wave count is a multiple of 8, else 1):
    // wave w takes from word w % words, which hands out the batches g = t * words + w % words - every word is shared by waves
    // of all CUs, so the balance stays device-wide while a word sees an eighth of the takes (work distribution).
    const uint32_t total_waves = gridDim.x * nwaves, wave_id = blockIdx.x * nwaves + wave;
    const uint32_t words = batch_arg >> 16, wshift = (uint32_t)__builtin_ctz(words | 0x100u);
    const uint32_t my_word = wave_id & (words - 1u), first_dyn = total_waves >> wshift;
    uint32_t tile = wave_id * batch, left = batch - 1u;      // `left`: tiles of the batch after this one
    uint32_t taken = 0;
    if (tile < n_tiles) dma_tile(tile);

    const uint32_t M1 = d.M[0], M2 = d.M[1], M3 = d.M[2], M4 = d.M[3];
    const uint32_t K2 = d.KTP[1], K3 = d.KTP[2], K4 = d.KTP[3];
    while (tile < n_tiles) {
        if (left == 0u) work_take_issue(taken, counter + 16u * my_word, 1u);
        bnm_wait_vmcnt<0>();
        // every per-lane quantity of the iteration is re-derived from this copy of the lane id (a handful of VALU per tile):
        // nothing but the lane id itself stays live across iterations, and hipcc cannot hoist derived addresses
        uint32_t lv = (uint32_t)lane;
        asm volatile("" : "+v"(lv));
        uint32_t next_tile = tile + 1u, next_left = left - 1u;
        const int j = (int)(lv & 31u), h = (int)(lv >> 5);
        const uint32_t lane16 = 16u * lv;
        const uint32_t rd_off = tile_off + (uint32_t)j * (uint32_t)ROW + 16u * ((uint32_t)h ^ G::mask((uint32_t)j));
        i32x4 pa[MMAX], pb[MMAX];       // packed layer outputs = the next layer's B operands
        // ---- layer 1: B operands from the tile buffer, KC K-steps at a time; the buffer is refilled with the wave's
        // next tile as soon as its last operand has been read (the load is then in flight for the rest of the tile)
        static_for<1, MMAX / MSTEP + 1>([&](auto MI) {
            constexpr int mt = decltype(MI)::value * MSTEP;
            if (M1 == (uint32_t)mt) {
                i32x16 acc[mt];
                static_for<0, KT0 / KC>([&](auto CI) {
                    constexpr int ch = decltype(CI)::value;
                    i32x4 b0[KC];
#pragma unroll
                    for (int s = 0; s < KC; s++) b0[s] = *(const i32x4 *)(smem + (rd_off ^ (32u * (uint32_t)(ch * KC + s))));
                    if constexpr (ch == KT0 / KC - 1) {
                        retire_lds_reads();
                        if (left == 0u) {
                            work_take_wait(taken);
                            next_tile = (((first_dyn + taken) << wshift) + my_word) * batch;
                            next_left = batch - 1u;
                        }
                        if (next_tile < n_tiles) dma_tile(next_tile);
                    }
                    mma_stream<mt, KC, KT0, SP, ch == 0, KC>(smem + (d.frag_off[0] + (uint32_t)(ch * KC * 1024) + lane16), b0, acc);
                });
                relunorm_pack<mt, DBL, MMAX>(acc, pa, h);
#pragma unroll
                for (int m = mt; m < MMAX; m++) pa[m] = i32x4{0, 0, 0, 0};
            }
        });
        hidden_layer<MMAX, SP, DBL>(smem, lane16, d.frag_off[1], M2, K2, pa, pb, h);
        const uint64_t img = ((uint64_t)tile << 5) + (uint64_t)j;
        int32_t *lrow = (logits_out && img < n) ? logits_out + img * d.n_classes : nullptr;
        // per-iteration copy: keeps the (row < n_classes) predicates of every accumulator register of every block from being
        // hoisted out of the persistent loop as hundreds of live 64-bit masks
        uint32_t nc = d.n_classes;
        asm volatile("" : "+v"(nc));
        uint32_t cls;
        if (M4) {
            hidden_layer<MMAX, SP, DBL>(smem, lane16, d.frag_off[2], M3, K3, pb, pa, h);
            cls = final_layer<MMAX, SP>(smem, lane16, d.frag_off[3], M4, K4, pa, h, lrow, nc);
        } else {
            cls = final_layer<MMAX, SP>(smem, lane16, d.frag_off[2], M3, K3, pb, h, lrow, nc);
        }
#ifdef BNM_EXPERIMENT_PLAIN_CLASS_STORE
        if (h == 0 && img < n) cls_out[img] = cls;
#else
        if (h == 0 && img < n) __builtin_nontemporal_store(cls, cls_out + img);   // see bnm_fused_fc.hip's dual kernel
#endif
        tile = next_tile;
        left = next_left;
    }
    bnm_wait_vmcnt<0>();   // no LDS-DMA may outlive the workgroup's LDS allocation
    work_block_leave_s(counter, total_waves);   // the last wave to leave puts the counter block back to all-zero
}

// ---- per-class launcher: each tile class is its own translation unit (bnm_fused_generic_m{2,4,8}.hip) so that the
// classes compile side by side.  Instantiations per (class, row length): doubled hidden weights; plain; plain with
// FP1.3.0's second weight plane.
#define BNM_GENERIC_PICK(MMAX, K0)                                                                              \
    if (kt0 == K0) {                                                                                            \
        if (sp == 1 && dbl) fn = fused_fc_generic_kernel<MMAX, K0, 1, true, bnmk_generic_wps(MMAX, K0, 1)>;     \
        else if (sp == 1) fn = fused_fc_generic_kernel<MMAX, K0, 1, false, bnmk_generic_wps(MMAX, K0, 1)>;      \
        else if (sp == 2 && !dbl) fn = fused_fc_generic_kernel<MMAX, K0, 2, false, bnmk_generic_wps(MMAX, K0, 2)>; \
    }
#define BNM_GENERIC_LAUNCHER(NAME, MMAX)                                                                               \
    hipError_t NAME(uint32_t kt0, uint32_t sp, bool dbl, unsigned blocks, unsigned threads, unsigned lds, hipStream_t s,     \
                    const int8_t *images, uint64_t n, const void *frags, const BnmGenericDesc &d, uint32_t *cls,            \
                    int32_t *logits, uint32_t *counter, uint32_t batch) {                                                    \
        typedef void (*fn_t)(const int8_t *, uint64_t, const i32x4 *, BnmGenericDesc, uint32_t *, int32_t *, uint32_t *, uint32_t); \
        fn_t fn = nullptr;                                                                                                   \
        BNM_GENERIC_PICK(MMAX, 2) BNM_GENERIC_PICK(MMAX, 4) BNM_GENERIC_PICK(MMAX, 8) BNM_GENERIC_PICK(MMAX, 16)             \
        if (!fn) return hipErrorInvalidValue;                                                                                \
        if (!blocks) return hipSuccess;   /* probe: is there an instantiation? */                                            \
        /* opt in to > 64 KiB of dynamic LDS (a per-device function attribute; a host call of about a microsecond) */        \
        hipError_t err = hipFuncSetAttribute((const void *)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);       \
        if (err != hipSuccess) return err;                                                                                   \
        fn<<<dim3(blocks), dim3(threads), lds, s>>>(images, n, (const i32x4 *)frags, d, cls, logits, counter, batch);        \
        return hipGetLastError();                                                                                            \
    }
