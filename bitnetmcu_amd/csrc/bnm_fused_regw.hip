// Fused whole-model FC kernel for the 3- and 4-tile shapes (96-, 112-, 128-wide hidden layers: members of the reference's
// documented 12 KB family, docs/documentation.md:169-183, and the ternary 96-96-96 of BASELINE configs[2] on its MFMA path):
// weight fragments RESIDENT IN ACCVGPRS at ONE wave per SIMD.  gfx950 (CDNA4 / MI355X) only.  OPT-IN (fused variant 9): it is
// bit-exact and measured 0 .. 7 % SLOWER than the generic kernel - DESIGN.md 4.1c says why (a lone wave issues VALU instructions at
// half the SIMD's rate, profiles/probes/mfma_valu_overlap.hip) - and stays in the library as the measured alternative.
// Reference semantics: BitNetMCU_inference.c:23-72 (ReLUNorm), :88-208 (processfclayer); schedule BitNetMCU_MNIST_dll.c:95-121.
//
// The idea: the generic kernel (bnm_fused_generic_kernel.hpp) reads every A fragment from LDS - one 1 KiB
// ds_read_b128 per MFMA, which at four matrix cores per CU is the LDS's whole 128 B per clock - and two waves per SIMD (256
// registers each) cannot hold 45 fragments.  A wave that has its SIMD to itself owns all 512 registers of the unified file, 256
// architectural VGPRs + 256 AccVGPRs, and an MFMA reads its A / B operands from either half (MI355X_MICROARCH.md, register files).
// So: the fragments are loaded straight into AccVGPRs (RegFrags::load) and named as AccVGPR operands by every MFMA of the loop
// (mfma_areg); the accumulators, which ReLUNorm's VALU code reads, stay in architectural VGPRs (this translation unit is compiled
// with -amdgpu-mfma-vgpr-form).  All fragments where they fit (96-96-96-10: 45 = 180 registers; 112-96-96-10: 56), layers 1..RL in
// registers and the rest in LDS where they do not (128-128-112-10: 64 of 68).
//
// Loop: a wave carries TWO 32-image tiles per iteration, as fused_fc_dual_kernel does (bnm_fused_fc.hip) - but with a single wave
// per SIMD the overlap of one tile's ReLUNorm with the other tile's MFMAs is the only overlap there is, so it is arranged by hand
// and across iterations (see the kernel).
#include <mutex>
#include <set>
#include <utility>      // (std::pair: the per-(kernel, device) set of allow_big_lds)
#include "bnm_fused_tile.hpp"
#include "bnm_fused_math.hpp"

// Measurement scaffolding (profiles/r04_regw_timing.py builds a private copy of the library with -DBNM_REGW_TIMING): shader-clock
// stamps around the two tile waits of the loop body; compiled out of the product library.
#ifdef BNM_REGW_TIMING
#define RW_TIMING(...) __VA_ARGS__
#else
#define RW_TIMING(...)
#endif

namespace {

// A fragments of one layer: MT tiles x KS K-steps, (m, s) at fragment index m * KS + s of the layer's image
template <int MT, int KS>
struct RegFrags {       // held in AccVGPRs for the whole persistent loop
    i32x4 a[MT][KS];
    // Loaded STRAIGHT INTO AccVGPRs by an asm load ("=a"): a value that starts its life in an architectural VGPR (any load hipcc
    // emits itself) reaches an "a"-constrained asm operand through a COPY per use, and the copies that MachineLICM does not hoist
    // out of the loop keep their VGPR originals alive in it - 22 fragments = 88 VGPRs and as many v_accvgpr_write per iteration
    // in the first asm-MFMA build.  hipcc does not count these loads (cdna_hip_programming.md 5.7, form iii): the kernel retires
    // them with one s_waitcnt vmcnt(0) before anything reads a fragment.
    BNM_DEVICE void load(const i32x4 *base, int lane) {
        const i32x4 *p = base + lane;
#pragma unroll
        for (int m = 0; m < MT; m++)
#pragma unroll
            for (int s = 0; s < KS; s++)
                asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(a[m][s]) : "v"(p + (m * KS + s) * 64) : "memory");
    }
    BNM_DEVICE i32x4 get(int m, int s) const { return a[m][s]; }
};
template <int MT, int KS>
struct LdsFrags {       // read from the workgroup's LDS copy, lane-linear ds_read_b128 (conflict-free)
    const char *p;      // LDS address of the layer's image + 16 * lane
    BNM_DEVICE i32x4 get(int m, int s) const { return *(const i32x4 *)(p + (m * KS + s) * 1024); }
};

// One MFMA of the pinned loop body, A operand NAMED as an AccVGPR tuple ("a" constraint).  With the builtin hipcc's register
// allocator is free to give a weight fragment an AccVGPR for one stretch of the loop and architectural VGPRs for another, and it
// does: ~200 v_accvgpr_read / v_accvgpr_mov per pair of tiles (18 % of the loop's VALU instructions) shuffling loop-invariant
// weights around (profiles/r04/regw_pmc_r04c_builtin_mfma.md).  A value whose every use in the loop asks for an AccVGPR stays in one.
// hipcc neither schedules nor pads an asm statement (cdna_hip_programming.md 5.7), so the hazards are handled by construction:
//   * VALU write -> MFMA operand read (2 wait states): every statement opens with s_nop 1, whatever hipcc put in front of it;
//   * MFMA result -> VALU read (12 wait states for this 8-pass MFMA): the loop body's order is pinned with sched_barrier and an
//     accumulator's reader (the sliced ReLUNorm of the NEXT block, or the next iteration) is hundreds of instructions behind its
//     last MFMA; the one short distance - the classifier's sums into the argmax - is padded with 16 wait states (settle());
//   * MFMA -> MFMA on the same accumulator (SrcC = vDst, the accumulate chain): no wait states, as in hipcc's own code;
//   * the B operand is never an MFMA result; a fragment's AccVGPRs are written once, ahead of the loop.
template <bool INIT>
BNM_DEVICE void mfma_areg(i32x16 &acc, const i32x4 &w, const i32x4 &b) {
    if constexpr (INIT) asm volatile("s_nop 1\n\tv_mfma_i32_32x32x32_i8 %0, %1, %2, 0" : "=&v"(acc) : "a"(w), "v"(b));
    else asm volatile("s_nop 1\n\tv_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(acc) : "a"(w), "v"(b));
}
// fragment (m, s) of a layer x b: register-resident layers through mfma_areg, LDS-resident ones through the builtin
template <bool INIT, int MT, int KS>
BNM_DEVICE void mfma_frag(i32x16 &acc, const RegFrags<MT, KS> &A, int m, int s, const i32x4 &b) {
    mfma_areg<INIT>(acc, A.a[m][s], b);
}
template <bool INIT, int MT, int KS>
BNM_DEVICE void mfma_frag(i32x16 &acc, const LdsFrags<MT, KS> &A, int m, int s, const i32x4 &b) {
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(A.get(m, s), b, INIT ? zero16() : acc, 0, 0, 0);
}

// layer L (1-based) of a model with RL register-resident layers: its fragments' home
template <int L, int RL, int MT, int KS>
struct FragsOf {
    typedef typename std::conditional<(L <= RL), RegFrags<MT, KS>, LdsFrags<MT, KS>>::type type;
};

constexpr int RW_WPB = 4;     // one wave per SIMD, one workgroup per CU

}  // namespace

// Filler plan: first[b] .. first[b + 1] are the K-steps (groups of M1 MFMAs) of the NEXT pair's tile A that block b of the loop
// body issues beside its own MFMAs (see the kernel); block 0 carries the whole layer 1 of the current pair's tile B instead.
template <int NL> struct FillerPlan;
template <> struct FillerPlan<4> { static constexpr int first[7] = {0, 0, 2, 4, 6, 7, 8}; };   // blocks 1..5: 2 2 2 1 1
template <> struct FillerPlan<3> { static constexpr int first[6] = {0, 0, 2, 4, 6, 8}; };      // blocks 1..4: 2 2 2 2

// RL: layers 1..RL keep their fragments in registers, the others in LDS.
//
// The loop is software-pipelined BY HAND, across iterations, because a wave that owns its SIMD has nobody to hide its stalls.
// A wave issues in order: an MFMA occupies the matrix core for 32 clocks = 8 VALU issues, two MFMAs back to back stall the
// wave's VALU work, a long VALU run leaves the matrix core idle.  So the body is a sequence of BLOCKS; a block is one tile's
// sliced ReLUNorm (relunorm_pack_sliced: chunks of 8-9 VALU instructions) with ONE MFMA of an independent chain issued in
// the slot behind each chunk, order pinned with sched_barrier:
//   block 0   ReLUNorm of tile A's layer-1 sums (carried in a1A from the previous iteration)  |  layer 1 of tile B: 8 M1 MFMAs
//   block 1   ReLUNorm of tile B's layer 1                    |  layer 2 of tile A  + K-steps of the NEXT pair's tile A, layer 1
//   block 2   ReLUNorm of tile A's layer 2                    |  layer 2 of tile B  + ...
//   block 3   ReLUNorm of tile B's layer 2                    |  layer 3 of tile A  + ...
//   block 4   ReLUNorm of tile A's layer 3                    |  layer 3 of tile B  + ...
//   block 5   ReLUNorm of tile B's layer 3                    |  layer 4 of tile A  + ...
//   block 6   layer 4 of tile B, argmax / logits / class-id store of both tiles
// Layer 1 depends on nothing but the image tile: it is the filler that keeps the matrix core busy where the dependent chain
// ReLUNorm -> MFMAs -> ReLUNorm has only VALU work.  Only tile A's layer-1 sums cross the back edge (48 registers for M1 = 3):
// with both tiles' sums carried, four accumulator sets are live through the middle of the body and hipcc spills.
// Two pair buffers (four 8 KiB tile slots) per wave; a slot is refilled - with the tile two pairs on - behind the block that
// read its last K-step, so a tile has more than an iteration to arrive.
// Pairs are assigned with a fixed stride (pair = wave + k * waves): with one wave per SIMD there is no arbitration between
// the waves of a SIMD (the reason the two-waves-per-SIMD kernels take their work from a counter), and a scalar
// atomic in flight would turn every LDS wait of the body (the fillers' operand reads) into a wait for its round trip.
template <int M1, int M2, int M3, int M4, int RL, bool DBL, int NC8>
__global__ __launch_bounds__(64 * RW_WPB, 1) void fused_fc_regw_kernel(const int8_t *__restrict__ images, uint64_t n,
                                                                         const i32x4 *__restrict__ frags, uint32_t n_classes,
                                                                         uint32_t *__restrict__ cls_out, int32_t *__restrict__ logits_out) {
    using std::integral_constant;
    constexpr int KT0 = 8;
    constexpr int NL = M4 > 0 ? 4 : 3;
    constexpr int MC = M4 > 0 ? M4 : M3;                                             // tiles of the classifier layer
    constexpr int F1 = M1 * KT0, F2 = M2 * M1, F3 = M3 * M2, F4 = M4 * M3;           // fragments per layer
    constexpr int REGF = (RL >= 1 ? F1 : 0) + (RL >= 2 ? F2 : 0) + (RL >= 3 ? F3 : 0) + (RL >= 4 ? F4 : 0);
    constexpr int LDSW = (F1 + F2 + F3 + F4 - REGF) * 1024;                          // bytes of weights in LDS
    constexpr uint32_t TILE = FUSED_TILE_BYTES;
    static_assert(RL >= 1 && RL <= NL, "layer 1 is always register-resident");
    // LDS: [weights of the layers behind RL][per wave: 4 tile slots of 8 KiB][per wave: 2 KiB logits staging]
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;

    if constexpr (LDSW > 0) {
        const i32x4 *src = frags + REGF * 64;
        for (uint32_t o = threadIdx.x; o < (uint32_t)(LDSW / 16); o += 64 * RW_WPB) ((i32x4 *)smem)[o] = src[o];
        __syncthreads();
    }
    typename FragsOf<1, RL, M1, KT0>::type A1;
    typename FragsOf<2, RL, M2, M1>::type A2;
    typename FragsOf<3, RL, M3, M2>::type A3;
    typename FragsOf<4, RL, (M4 > 0 ? M4 : 1), M3>::type A4;
    {
        const i32x4 *fp = frags;
        const char *lp = smem + 16 * lane;
        A1.load(fp, lane);  fp += F1 * 64;
        if constexpr (RL >= 2) { A2.load(fp, lane); fp += F2 * 64; } else { A2.p = lp; lp += F2 * 1024; }
        if constexpr (RL >= 3) { A3.load(fp, lane); fp += F3 * 64; } else { A3.p = lp; lp += F3 * 1024; }
        if constexpr (M4 > 0) {
            if constexpr (RL >= 4) A4.load(fp, lane); else A4.p = lp;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the fragments' asm loads (RegFrags::load)
        __builtin_amdgcn_sched_barrier(0);
    }
    char *const tiles = smem + LDSW + wave * (4 * TILE);
    int32_t *const stage = (int32_t *)(smem + LDSW + RW_WPB * 4 * TILE + wave * 2048);

    const uint64_t n_pairs = n >> 6;            // the launcher hands this kernel whole 64-image pairs only
    const uint64_t stride = (uint64_t)gridDim.x * RW_WPB;
    uint64_t cur = (uint64_t)blockIdx.x * RW_WPB + (uint64_t)wave;      // the pair being finished; tile A's layer-1 sums are in a1A

    uint32_t voff[4];
#pragma unroll
    for (int u = 0; u < 4; u++) voff[u] = (uint32_t)(lane >> 4) * 256u + 16u * ((uint32_t)(lane & 15) ^ (uint32_t)(lane >> 4) ^ (4u * u));
    const uint32_t lds_wave = (uint32_t)(size_t)(__attribute__((address_space(3))) char *)tiles;
    const uint32_t rd_base = (uint32_t)j * 256u + 16u * ((uint32_t)h ^ (uint32_t)(j & 15));

    auto dma_tile = [&](uint64_t t, uint32_t slot_off) {
        const int8_t *base = images + t * (uint64_t)TILE;
        lds_dma_tile8_linear(lds_wave + slot_off, base, base + 4096, voff[0], voff[1], voff[2], voff[3]);
    };
    // a pair index past the end re-reads the wave's current pair: constant wait counts, branch-free body, never out of bounds
    auto clamp_pair = [&](uint64_t p) -> uint64_t { return p < n_pairs ? p : cur; };
    // B operand of K-step s of the tile in slot `slot_off` (row j, 16-byte slot 2s + h, XOR-swizzled: bnm_fused_fc.hip)
    auto b_operand = [&](uint32_t slot_off, int s) -> i32x4 { return *(const i32x4 *)(tiles + ((rd_base + slot_off) ^ (32u * (uint32_t)s))); };
    constexpr int NWAIT = 24;     // loads younger than the slot waited for: the three other slots' 8 pieces each

    i32x16 a1A[M1];
#pragma unroll
    for (int m = 0; m < M1; m++) a1A[m] = zero16();

    // ---- prologue: both buffers requested, layer 1 of the first pair's tile A computed, its slot refilled --------------
    const bool any = cur < n_pairs;
    if (any) {
        dma_tile(2ull * cur, 0);
        dma_tile(2ull * cur + 1ull, TILE);
        const uint64_t p1 = clamp_pair(cur + stride);
        dma_tile(2ull * p1, 2 * TILE);
        dma_tile(2ull * p1 + 1ull, 3 * TILE);
        bnm_wait_vmcnt<NWAIT>();
        i32x4 b0[KT0];
#pragma unroll
        for (int sk = 0; sk < KT0; sk++) b0[sk] = b_operand(0u, sk);
        static_for<0, KT0 * M1>([&](auto K_) {
            constexpr int sk = decltype(K_)::value / M1, m = decltype(K_)::value % M1;
            mfma_frag<sk == 0>(a1A[m], A1, m, sk, b0[sk]);
        });
        dma_tile(2ull * clamp_pair(cur + 2ull * stride), 0);
#pragma unroll
        for (int m = 0; m < M1; m++) asm volatile("s_nop 7\n\ts_nop 7" : "+v"(a1A[m]));      // asm MFMA result -> first reader
    }
    uint32_t par = 0;      // byte offset of the pair buffer that holds the current pair (0 or 2 tiles)
    RW_TIMING(uint64_t t_wait[2] = {0, 0}, t_iters = 0; const uint64_t t_start = __builtin_readcyclecounter();)

    while (cur < n_pairs) {
        const uint64_t fill_b = clamp_pair(cur + 2ull * stride), fill_a = clamp_pair(cur + 3ull * stride);
        const uint32_t slot_b = par + TILE, slot_a = par ^ (2u * TILE);      // tile B of this pair, tile A of the next one
        // One block: NCRIT = KC * MTC MFMAs of the dependent chain (layer weights AC, B operands bc, sums accc) and the K-steps
        // [g0, g1) of a layer-1 filler (tile slot fslot, sums accf), one MFMA per slot of the ReLUNorm of accv -> pv.  Critical
        // MFMAs first, K-step outermost.  The filler's B operands are read at the top of the block, a layer of MFMAs ahead of use.
        auto block = [&](auto G0_, auto G1_, uint32_t fslot, i32x16 (&accf)[M1], auto KC_, auto MTC_, const auto &AC, const auto &bc, auto &accc,
                         auto MTV_, const auto &accv, auto &pv) {
            constexpr int g0 = decltype(G0_)::value, g1 = decltype(G1_)::value;
            constexpr int KC = decltype(KC_)::value, MTC = decltype(MTC_)::value, MTV = decltype(MTV_)::value;
            constexpr int NG = g1 - g0, NCRIT = KC * MTC, NM = NCRIT + NG * M1, NS = MTV > 0 ? relunorm_slots<MTV>() : 1;
            __builtin_amdgcn_sched_barrier(0);
            i32x4 fb[NG > 0 ? NG : 1];
            if constexpr (g0 == 0 && NG > 0) {                                // the filler's tile has landed
                RW_TIMING(const uint64_t t0 = __builtin_readcyclecounter();)
                bnm_wait_vmcnt<NWAIT>();
                RW_TIMING(t_wait[KC == 0 ? 0 : 1] += __builtin_readcyclecounter() - t0;)
            }
#pragma unroll
            for (int g = g0; g < g1; g++) fb[g - g0] = b_operand(fslot, g);
            __builtin_amdgcn_sched_barrier(0);
            auto issue = [&](auto K_) {      // MFMA k of the block
                constexpr int k = decltype(K_)::value;
                if constexpr (k < NCRIT) {
                    constexpr int sk = k / MTC, m = k % MTC;
                    mfma_frag<sk == 0>(accc[m], AC, m, sk, bc[sk]);
                } else if constexpr (k < NM) {
                    constexpr int sk = g0 + (k - NCRIT) / M1, m = (k - NCRIT) % M1;
                    mfma_frag<sk == 0>(accf[m], A1, m, sk, fb[sk - g0]);
                }
            };
            // NM MFMAs over NS slots: slot i issues MFMAs [i * NM / NS, (i + 1) * NM / NS)
            auto slot = [&](auto I_) {
                constexpr int i = decltype(I_)::value;
                static_for<(i * NM) / NS, ((i + 1) * NM) / NS>(issue);
                __builtin_amdgcn_sched_barrier(0);
            };
            if constexpr (MTV > 0) relunorm_pack_sliced<MTV, DBL>(accv, pv, h, slot);
            else static_for<0, NM>(issue);
            __builtin_amdgcn_sched_barrier(0);
        };
        typedef integral_constant<int, 0> I0;
        typedef integral_constant<int, KT0> I8;
        typedef FillerPlan<NL> FP;
#define FILL(B) integral_constant<int, FP::first[B]>{}, integral_constant<int, FP::first[B + 1]>{}, slot_a, a1A
#define REFILL_A(B) if constexpr (FP::first[B + 1] == KT0 && FP::first[B] < KT0) dma_tile(2ull * fill_a, slot_a)

        i32x4 p1A[M1], p1B[M1], p2A[M2], p2B[M2];
        i32x16 a1B[M1], a2A[M2], a2B[M2], a3A[M3], a3B[M3];
        // block 0: layer 1 of tile B (no dependent chain: its 8 K-steps ARE the block's MFMAs) beside tile A's first ReLUNorm
        block(I0{}, I8{}, slot_b, a1B, I0{}, I0{}, A1, p1A, a1B, integral_constant<int, M1>{}, a1A, p1A);
        dma_tile(2ull * fill_b + 1ull, slot_b);
        block(FILL(1), integral_constant<int, M1>{}, integral_constant<int, M2>{}, A2, p1A, a2A, integral_constant<int, M1>{}, a1B, p1B);
        REFILL_A(1);
        block(FILL(2), integral_constant<int, M1>{}, integral_constant<int, M2>{}, A2, p1B, a2B, integral_constant<int, M2>{}, a2A, p2A);
        REFILL_A(2);
        block(FILL(3), integral_constant<int, M2>{}, integral_constant<int, M3>{}, A3, p2A, a3A, integral_constant<int, M2>{}, a2B, p2B);
        REFILL_A(3);

        const uint64_t imgA = (cur << 6) + (uint64_t)j, imgB = imgA + 32ull;
        int32_t *const tile_a = logits_out + (cur << 6) * n_classes, *const tile_b = tile_a + 32u * n_classes;
        uint32_t clsA, clsB;
        auto finish = [&](const i32x16 (&accA)[MC], const i32x16 (&accB)[MC]) {
            clsA = argmax_rows<MC, NC8>(accA, h);
            clsB = argmax_rows<MC, NC8>(accB, h);
#ifndef BNM_REGW_TIMING
            if (logits_out) {
                if (n_classes <= 16u) {
                    store_logits_tile<MC, NC8, 0>(accA, stage, tile_a, j, h, lane, n_classes);
                    store_logits_tile<MC, NC8, 0>(accB, stage, tile_b, j, h, lane, n_classes);
                } else {
                    store_logits<MC>(accA, tile_a + (uint32_t)j * n_classes, h, n_classes);
                    store_logits<MC>(accB, tile_b + (uint32_t)j * n_classes, h, n_classes);
                }
            }
#endif
        };
        // the classifier's sums are read a few instructions behind their last MFMA: 16 wait states between an asm MFMA (whose
        // latency hipcc does not know) and its first reader
        auto settle = [&](i32x16 (&acc)[MC]) {
#pragma unroll
            for (int m = 0; m < MC; m++) asm volatile("s_nop 7\n\ts_nop 7" : "+v"(acc[m]));
        };
        if constexpr (M4 > 0) {
            i32x4 p3A[M3], p3B[M3];
            i32x16 a4A[MC], a4B[MC];
            block(FILL(4), integral_constant<int, M2>{}, integral_constant<int, M3>{}, A3, p2B, a3B, integral_constant<int, M3>{}, a3A, p3A);
            REFILL_A(4);
            block(FILL(5), integral_constant<int, M3>{}, integral_constant<int, MC>{}, A4, p3A, a4A, integral_constant<int, M3>{}, a3B, p3B);
            REFILL_A(5);
            block(I0{}, I0{}, 0u, a1A, integral_constant<int, M3>{}, integral_constant<int, MC>{}, A4, p3B, a4B, I0{}, a3B, p3B);      // block 6
            settle(a4B);
            finish(a4A, a4B);
        } else {
            block(FILL(4), integral_constant<int, M2>{}, integral_constant<int, M3>{}, A3, p2B, a3B, I0{}, a2B, p2B);
            REFILL_A(4);
            settle(a3B);
            finish(a3A, a3B);
        }
#undef FILL
#undef REFILL_A
        // lanes 0..31 store tile A's class ids, lanes 32..63 tile B's: one 256-byte nontemporal store per pair.  It is the
        // youngest vector-memory operation by far when the next waits run (24 younger LOADS are allowed to be outstanding,
        // this store is behind 32 of them), so no wait of the next iteration sits on its acknowledgement.
        __builtin_nontemporal_store(h ? clsB : clsA, cls_out + (h ? imgB : imgA));
        cur += stride;
        par ^= 2u * TILE;
        RW_TIMING(t_iters++;)
    }
    bnm_wait_vmcnt<0>();        // no LDS-DMA may outlive the workgroup's LDS allocation
#ifdef BNM_REGW_TIMING
    // the logits buffer is reused as the record array: 4 x uint64 per wave {loop cycles, wait B (block 0), wait A (block 1), iterations}
    if (logits_out && lane == 0) {
        uint64_t *rec = (uint64_t *)logits_out + 4ull * ((uint64_t)blockIdx.x * RW_WPB + (uint64_t)wave);
        rec[0] = __builtin_readcyclecounter() - t_start;
        rec[1] = t_wait[0];
        rec[2] = t_wait[1];
        rec[3] = t_iters;
    }
#endif
}

// ---- dispatch ------------------------------------------------------------------------------------------------------------
namespace {
typedef void (*regw_fn)(const int8_t *, uint64_t, const i32x4 *, uint32_t, uint32_t *, int32_t *);
struct RegwEntry {
    int M[4];
    bool dbl;
    int nc8;          // 0: any class count
    int rl;
    regw_fn fn;
};
#define REGW(M1, M2, M3, M4, RL, DBL, NC8) { {M1, M2, M3, M4}, DBL, NC8, RL, fused_fc_regw_kernel<M1, M2, M3, M4, RL, DBL, NC8> }
#define REGW_ANY_AND_10(M1, M2, M3, M4, RL, DBL) REGW(M1, M2, M3, M4, RL, DBL, 2), REGW(M1, M2, M3, M4, RL, DBL, 0)
const RegwEntry kRegw[] = {
    // 256-96-96-96-N: the ternary model of BASELINE configs[2] on the MFMA path; 45 fragments, all in registers
    REGW_ANY_AND_10(3, 3, 3, 1, 4, true),
    // the documented 12 KB shapes (docs/documentation.md:169-183): 112-96-96 2-bit (56 fragments, all in registers), 128-128-112
    // ternary (68: layers 1-3 = 64 fragments = all 256 AccVGPRs, the classifier's 4 from LDS).  (160-160-160 binary: 95 fragments,
    // 55 KiB of them in LDS beside four tile slots per wave = 191 KiB: does not fit; it compiles - zero scratch - with two slots.)
    REGW(4, 3, 3, 1, 4, true, 2),
    REGW(4, 4, 4, 1, 3, true, 2),
};
uint32_t regw_lds_bytes(const RegwEntry &e);
const RegwEntry *find_regw(const BnmFusedShape &sh) {
    if (sh.KT0 != 8 || sh.split) return nullptr;
    for (int pass = 0; pass < 2; pass++)
        for (const RegwEntry &e : kRegw)
            if (e.M[0] == sh.M[0] && e.M[1] == sh.M[1] && e.M[2] == sh.M[2] && e.M[3] == sh.M[3] && e.dbl == sh.dbl &&
                e.nc8 == (pass ? 0 : sh.nc8) && regw_lds_bytes(e) <= 160u * 1024u)
                return &e;
    return nullptr;
}
uint32_t regw_lds_bytes(const RegwEntry &e) {
    const int f[4] = {e.M[0] * 8, e.M[1] * e.M[0], e.M[2] * e.M[1], e.M[3] * e.M[2]};
    uint32_t w = 0;
    for (int l = e.rl; l < 4; l++) w += (uint32_t)f[l] * 1024u;
    return w + RW_WPB * (4u * FUSED_TILE_BYTES + 2048u);
}
hipError_t allow_big_lds(const void *fn) {
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
}  // namespace

bool bnmk_regw_supported(const BnmFusedShape &sh) { return find_regw(sh) != nullptr; }

// whole 64-image pairs only (a.n % 64 == 0): the caller gives the remainder to the generic kernel
hipError_t bnmk_fused_regw(const BnmFusedShape &sh, int grid_blocks, const BnmFusedArgs &a, hipStream_t s) {
    const RegwEntry *e = find_regw(sh);
    if (!e || (a.n & 63ull)) return hipErrorInvalidValue;
    if (a.n == 0) return hipSuccess;
    const uint64_t n_pairs = a.n >> 6;
    const uint64_t want = (n_pairs + RW_WPB - 1) / RW_WPB;
    const uint64_t cap = grid_blocks > 0 ? (uint64_t)grid_blocks : (uint64_t)bnm_num_cus();     // one workgroup per CU
    const uint64_t blocks = want < cap ? want : cap;
    if (hipError_t err = allow_big_lds((const void *)e->fn); err != hipSuccess) return err;
    e->fn<<<dim3((unsigned)blocks), dim3(64 * RW_WPB), regw_lds_bytes(*e), s>>>(a.images, a.n, (const i32x4 *)a.frags, a.n_classes, a.cls,
                                                                              a.logits);
    return hipGetLastError();
}
