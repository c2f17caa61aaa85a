// Fused whole-model FC kernel for the 3..5-tile shapes (96-, 112-, 128-, 160-wide hidden layers: the reference's documented
// 12 KB family, docs/documentation.md:169-183, and the ternary 96-96-96 of BASELINE configs[2] on its MFMA path):
// weight fragments RESIDENT IN THE REGISTER FILE at ONE wave per SIMD.  gfx950 (CDNA4 / MI355X) only.
// Reference semantics: BitNetMCU_inference.c:23-72 (ReLUNorm), :88-208 (processfclayer); schedule BitNetMCU_MNIST_dll.c:95-121.
//
// Why: the generic kernel (bnm_fused_generic_kernel.hpp) reads every A fragment from LDS - one 1 KiB ds_read_b128 per MFMA.
// Four matrix cores per CU at one MFMA per 32 clocks each ask the LDS for exactly its 128 B per clock, so from three tiles per
// layer on that kernel is LDS-bound (profiles/r03/r03z_pmc_doc12k_binary.md: 1.08 LDS instructions per MFMA, matrix cores 52 %
// busy), and the two-waves-per-SIMD register budget (256) cannot hold 45 fragments.  A wave that has its SIMD to itself owns
// all 512 registers of the unified file: 256 architectural VGPRs + 256 AccVGPRs, and an MFMA reads its A / B operands from
// either half (MI355X_MICROARCH.md, register files).  This translation unit is compiled with -amdgpu-mfma-vgpr-form, so the
// accumulators (which ReLUNorm's VALU code reads) stay in architectural VGPRs and hipcc's allocator places the loop-invariant
// weight fragments in AccVGPRs, where the matrix core reads them in place: no LDS traffic for weights at all when all layers
// fit (RL = number of layers, e.g. 96-96-96-10: 45 fragments = 180 registers), layer 1 in registers and the later layers in
// LDS when they do not (160-160-160-10: 40 of 95 fragments in registers, the LDS reads of the rest drop from 95 to 55 per tile).
//
// Loop structure = fused_fc_dual_kernel's (bnm_fused_fc.hip): a wave carries TWO independent 32-image tiles per iteration in
// one basic block, so one tile's ReLUNorm VALU work sits between the other tile's MFMAs - with a single wave per SIMD that
// in-wave overlap is the only overlap there is - and it is arranged by hand, across iterations (see the kernel).
#include <mutex>
#include <set>
#include <utility>
#include "bnm_fused_tile.hpp"
#include "bnm_fused_math.hpp"

namespace {

// A fragments of one layer: MT tiles x KS K-steps, (m, s) at fragment index m * KS + s of the layer's image
template <int MT, int KS>
struct RegFrags {       // held in registers for the whole persistent loop
    i32x4 a[MT][KS];
    BNM_DEVICE void load(const i32x4 *base, int lane) {
#pragma unroll
        for (int m = 0; m < MT; m++)
#pragma unroll
            for (int s = 0; s < KS; s++) a[m][s] = base[(m * KS + s) * 64 + lane];
    }
    BNM_DEVICE i32x4 get(int m, int s) const { return a[m][s]; }
};
template <int MT, int KS>
struct LdsFrags {       // read from the workgroup's LDS copy, lane-linear ds_read_b128 (conflict-free)
    const char *p;      // LDS address of the layer's image + 16 * lane
    BNM_DEVICE i32x4 get(int m, int s) const { return *(const i32x4 *)(p + (m * KS + s) * 1024); }
};

template <int MT, int KT, class F>
BNM_DEVICE void mma(const F &A, const i32x4 (&b)[KT], i32x16 (&acc)[MT]) {
#pragma unroll
    for (int m = 0; m < MT; m++) acc[m] = zero16();
    // K-step outermost: consecutive MFMAs go to different accumulators
#pragma unroll
    for (int s = 0; s < KT; s++)
#pragma unroll
        for (int m = 0; m < MT; m++) acc[m] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A.get(m, s), b[s], acc[m], 0, 0, 0);
}

// layer L (1-based) of a model with RL register-resident layers: its fragments' home
template <int L, int RL, int MT, int KS>
struct FragsOf {
    typedef typename std::conditional<(L <= RL), RegFrags<MT, KS>, LdsFrags<MT, KS>>::type type;
};

constexpr int RW_WPB = 4;     // one wave per SIMD, one workgroup per CU

}  // namespace

// Filler plan: how many K-step groups of the NEXT pair's layer 1 (16 groups: 8 K-steps x 2 tiles, M1 MFMAs each) go into each
// block of the loop body (see the kernel).  Block 0 holds none (its VALU work reads the accumulators the first groups overwrite).
template <int NL> struct FillerPlan;
template <> struct FillerPlan<4> { static constexpr int first[8] = {0, 0, 3, 6, 9, 12, 16, 16}; };   // blocks 1..5: 3 3 3 3 4
template <> struct FillerPlan<3> { static constexpr int first[6] = {0, 0, 5, 10, 15, 16}; };         // blocks 1..4: 5 5 5 1

// RL: layers 1..RL keep their fragments in registers, the others in LDS.
//
// The loop is software-pipelined BY HAND across iterations, because a wave that owns its SIMD has nobody to hide its stalls:
//   * layer 1 of the NEXT pair runs one iteration ahead.  Its MFMAs depend on nothing but the image tile, so they are the
//     filler that keeps the matrix core busy wherever the current pair's dependent chain (ReLUNorm -> MFMAs -> ReLUNorm ...)
//     has only VALU work: 16 K-step groups spread over the body's blocks (FillerPlan), each reading its B operand from the
//     tile slot just in time.  The layer-1 sums a1[2][M1] are the only state carried across the back edge, and a group
//     accumulates straight into the registers the current pair's first ReLUNorm has just vacated;
//   * the body alternates the two tiles of the current pair at block granularity: the MFMAs of one tile's next layer are
//     issued in front of the other tile's ReLUNorm, so every VALU block has the critical MFMAs of the other chain plus filler
//     beside it.  Per pair of a 96-96-96-10 model: 90 MFMAs = 2880 matrix-core clocks under ~930 VALU = 3700 clocks;
//   * two pair buffers (four 8 KiB tile slots) per wave: the pair whose layer 1 is computed in iteration i was requested in
//     iteration i-2; a slot is refilled (pair i+3) as soon as its last K-step has been read;
//   * pairs are assigned with a fixed stride (pair = wave + k * waves): with one wave per SIMD there is no arbitration between
//     waves of a SIMD (the reason the two-waves-per-SIMD kernels take their work from a counter), and a scalar
//     atomic in flight would turn every LDS wait of the body (the fillers' operand reads) into a wait for its round trip.
template <int M1, int M2, int M3, int M4, int RL, bool DBL, int NC8>
__global__ __launch_bounds__(64 * RW_WPB, 1) void fused_fc_regw_kernel(const int8_t *__restrict__ images, uint64_t n,
                                                                         const i32x4 *__restrict__ frags, uint32_t n_classes,
                                                                         uint32_t *__restrict__ cls_out, int32_t *__restrict__ logits_out) {
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
    }
    char *const tiles = smem + LDSW + wave * (4 * TILE);
    int32_t *const stage = (int32_t *)(smem + LDSW + RW_WPB * 4 * TILE + wave * 2048);

    const uint64_t n_pairs = n >> 6;            // the launcher hands this kernel whole 64-image pairs only
    const uint64_t stride = (uint64_t)gridDim.x * RW_WPB;
    uint64_t cur = (uint64_t)blockIdx.x * RW_WPB + (uint64_t)wave;      // the pair being finished; its layer-1 sums are in a1

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
    constexpr int NWAIT = 24;     // loads younger than the slot waited for: the three other slots' 8 pieces each

    i32x16 a1[2][M1];
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
        for (int m = 0; m < M1; m++) a1[t][m] = zero16();
    // K-step s of tile t from the pair buffer at par_off: a1[t][m] (+)= A1(m, s) x B
    auto l1_step = [&](uint32_t par_off, auto T_, auto S_) {
        constexpr int t = decltype(T_)::value, s = decltype(S_)::value;
        const i32x4 b = *(const i32x4 *)(tiles + ((rd_base + par_off + (uint32_t)t * TILE) ^ (32u * (uint32_t)s)));
#pragma unroll
        for (int m = 0; m < M1; m++) {
            const i32x16 c = s == 0 ? zero16() : a1[t][m];
            a1[t][m] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A1.get(m, s), b, c, 0, 0, 0);
        }
    };

    // ---- prologue: pairs cur and cur + stride requested, layer 1 of cur computed, cur + 2 stride requested -------------
    const bool any = cur < n_pairs;
    if (any) {
        dma_tile(2ull * cur, 0);
        dma_tile(2ull * cur + 1ull, TILE);
        const uint64_t p1 = clamp_pair(cur + stride);
        dma_tile(2ull * p1, 2 * TILE);
        dma_tile(2ull * p1 + 1ull, 3 * TILE);
        bnm_wait_vmcnt<16>();
        static_for<0, 8>([&](auto S_) { l1_step(0u, std::integral_constant<int, 0>{}, S_); });
        static_for<0, 8>([&](auto S_) { l1_step(0u, std::integral_constant<int, 1>{}, S_); });
        const uint64_t p2 = clamp_pair(cur + 2ull * stride);
        dma_tile(2ull * p2, 0);
        dma_tile(2ull * p2 + 1ull, TILE);
    }
    uint32_t par_off = 2u * TILE;      // the pair buffer whose tiles go through layer 1 in this iteration (cur + stride)

    while (cur < n_pairs) {
        const uint64_t fill = clamp_pair(cur + 3ull * stride);
        // One block of the body: the MFMAs of one tile's layer (they wait for nothing but the packed activations the previous
        // block produced) and G filler groups of the next pair's layer 1 (B operands read from the tile slot at the top of the
        // block, a whole layer of MFMAs ahead of their use), issued ONE PER SLOT of the other tile's sliced ReLUNorm
        // (relunorm_pack_sliced: 8-9 VALU per slot = the 32 clocks an MFMA occupies the matrix core), critical MFMAs first, K-step
        // outermost.  The order is pinned with sched_barrier - a wave issues in order, so two MFMAs back to back stall its VALU
        // work and a long VALU run leaves the matrix core idle; left to its own devices (and to sched_group_barrier) hipcc
        // produced both.  Filler group g: tile g >> 3, K-step g & 7; a slot is refilled behind the block that read its last K-step.
        auto block = [&](auto B_, auto KC_, auto MTC_, const auto &AC, const auto &bc, auto &accc, auto MTV_, const auto &accv, auto &pv) {
            constexpr int blk = decltype(B_)::value, g0 = FillerPlan<NL>::first[blk], g1 = FillerPlan<NL>::first[blk + 1];
            constexpr int KC = decltype(KC_)::value, MTC = decltype(MTC_)::value, MTV = decltype(MTV_)::value;
            constexpr int NG = g1 - g0, NCRIT = KC * MTC, NM = NCRIT + NG * M1, NS = MTV > 0 ? relunorm_slots<MTV>() : 0;
            __builtin_amdgcn_sched_barrier(0);
            i32x4 fb[NG > 0 ? NG : 1];
            static_for<g0, g1>([&](auto G_) {
                constexpr int g = decltype(G_)::value, t = g >> 3, sk = g & 7;
                if constexpr (g == 8) bnm_wait_vmcnt<NWAIT>();      // slot B (slot A: the top of the body)
                fb[g - g0] = *(const i32x4 *)(tiles + ((rd_base + par_off + (uint32_t)t * TILE) ^ (32u * (uint32_t)sk)));
            });
            __builtin_amdgcn_sched_barrier(0);
            auto issue = [&](auto K_) {      // MFMA k of the block
                constexpr int k = decltype(K_)::value;
                if constexpr (k < NCRIT) {
                    constexpr int sk = k / MTC, m = k % MTC;
                    const i32x16 c = sk == 0 ? zero16() : accc[m];
                    accc[m] = __builtin_amdgcn_mfma_i32_32x32x32_i8(AC.get(m, sk), bc[sk], c, 0, 0, 0);
                } else if constexpr (k < NM) {
                    constexpr int g = g0 + (k - NCRIT) / M1, m = (k - NCRIT) % M1, t = g >> 3, sk = g & 7;
                    const i32x16 c = sk == 0 ? zero16() : a1[t][m];
                    a1[t][m] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A1.get(m, sk), fb[g - g0], c, 0, 0, 0);
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
            static_for<g0, g1>([&](auto G_) {
                constexpr int g = decltype(G_)::value, t = g >> 3, sk = g & 7;
                if constexpr (sk == 7) dma_tile(2ull * fill + (uint64_t)t, par_off + (uint32_t)t * TILE);
            });
        };
        using std::integral_constant;
        bnm_wait_vmcnt<NWAIT>();        // slot A of the next pair has landed

        i32x4 p1A[M1], p1B[M1], p2A[M2], p2B[M2];
        i32x16 a2A[M2], a2B[M2], a3A[M3], a3B[M3];
        typedef integral_constant<int, 0> I0;
        relunorm_pack<M1, DBL>(a1[0], p1A, h);                                                                         // block 0
        block(integral_constant<int, 1>{}, integral_constant<int, M1>{}, integral_constant<int, M2>{}, A2, p1A, a2A,
              integral_constant<int, M1>{}, a1[1], p1B);
        block(integral_constant<int, 2>{}, integral_constant<int, M1>{}, integral_constant<int, M2>{}, A2, p1B, a2B,
              integral_constant<int, M2>{}, a2A, p2A);
        block(integral_constant<int, 3>{}, integral_constant<int, M2>{}, integral_constant<int, M3>{}, A3, p2A, a3A,
              integral_constant<int, M2>{}, a2B, p2B);

        const uint64_t imgA = (cur << 6) + (uint64_t)j, imgB = imgA + 32ull;
        int32_t *const tile_a = logits_out + (cur << 6) * n_classes, *const tile_b = tile_a + 32u * n_classes;
        uint32_t clsA, clsB;
        auto finish = [&](const i32x16 (&accA)[MC], const i32x16 (&accB)[MC]) {
            clsA = argmax_rows<MC, NC8>(accA, h);
            clsB = argmax_rows<MC, NC8>(accB, h);
            if (logits_out) {
                if (n_classes <= 16u) {
                    store_logits_tile<MC, NC8, 0>(accA, stage, tile_a, j, h, lane, n_classes);
                    store_logits_tile<MC, NC8, 0>(accB, stage, tile_b, j, h, lane, n_classes);
                } else {
                    store_logits<MC>(accA, tile_a + (uint32_t)j * n_classes, h, n_classes);
                    store_logits<MC>(accB, tile_b + (uint32_t)j * n_classes, h, n_classes);
                }
            }
        };
        if constexpr (M4 > 0) {
            i32x4 p3A[M3], p3B[M3];
            i32x16 a4A[MC], a4B[MC];
            block(integral_constant<int, 4>{}, integral_constant<int, M2>{}, integral_constant<int, M3>{}, A3, p2B, a3B,
                  integral_constant<int, M3>{}, a3A, p3A);
            block(integral_constant<int, 5>{}, integral_constant<int, M3>{}, integral_constant<int, MC>{}, A4, p3A, a4A,
                  integral_constant<int, M3>{}, a3B, p3B);
            mma<MC, M3>(A4, p3B, a4B);                                                                                 // block 6
            finish(a4A, a4B);
        } else {
            block(integral_constant<int, 4>{}, integral_constant<int, M2>{}, integral_constant<int, M3>{}, A3, p2B, a3B, I0{}, a3A, p2A);
            finish(a3A, a3B);
        }
        // lanes 0..31 store tile A's class ids, lanes 32..63 tile B's: one 256-byte nontemporal store per pair.  It is the
        // youngest vector-memory operation by far when the next waits run (24 younger LOADS are allowed to be outstanding,
        // this store is behind 32 of them), so no wait of the next iteration sits on its acknowledgement.
        __builtin_nontemporal_store(h ? clsB : clsA, cls_out + (h ? imgB : imgA));
        cur += stride;
        par_off ^= 2u * TILE;
    }
    bnm_wait_vmcnt<0>();        // no LDS-DMA may outlive the workgroup's LDS allocation
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
};
const RegwEntry *find_regw(const BnmFusedShape &sh) {
    if (sh.KT0 != 8 || sh.split) return nullptr;
    for (int pass = 0; pass < 2; pass++)
        for (const RegwEntry &e : kRegw)
            if (e.M[0] == sh.M[0] && e.M[1] == sh.M[1] && e.M[2] == sh.M[2] && e.M[3] == sh.M[3] && e.dbl == sh.dbl &&
                e.nc8 == (pass ? 0 : sh.nc8))
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
