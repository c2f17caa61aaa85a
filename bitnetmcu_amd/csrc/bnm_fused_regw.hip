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
// in-wave overlap is the only overlap there is.  The image stream is D pairs deep (D = 2: four 8 KiB tile slots per wave,
// the refill issued behind a tile's layer-1 MFMAs is the pair after next), since no second wave hides a late tile.
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

__device__ __forceinline__ void store_ids_masked(uint32_t *addr, uint32_t value, uint64_t mask) {
    uint64_t saved;
    asm volatile("s_and_saveexec_b64 %0, %1\n\tglobal_store_dword %2, %3, off nt\n\ts_mov_b64 exec, %0"
                 : "=&s"(saved) : "s"(mask), "v"(addr), "v"(value) : "memory", "scc");
}

constexpr int RW_WPB = 4;     // one wave per SIMD, one workgroup per CU

}  // namespace

// RL: layers 1..RL keep their fragments in registers, the others in LDS.  D: pairs in flight per wave (1 or 2).
template <int M1, int M2, int M3, int M4, int RL, bool DBL, int NC8, int D>
__global__ __launch_bounds__(64 * RW_WPB, 1) void fused_fc_regw_kernel(const int8_t *__restrict__ images, uint64_t n,
                                                                         const i32x4 *__restrict__ frags, uint32_t n_classes,
                                                                         uint32_t *__restrict__ cls_out, int32_t *__restrict__ logits_out,
                                                                         uint32_t *__restrict__ work, uint32_t *__restrict__ idle,
                                                                         uint32_t batch_arg) {
    constexpr int KT0 = 8;
    constexpr int F1 = M1 * KT0, F2 = M2 * M1, F3 = M3 * M2, F4 = M4 * M3;           // fragments per layer
    constexpr int REGF = (RL >= 1 ? F1 : 0) + (RL >= 2 ? F2 : 0) + (RL >= 3 ? F3 : 0) + (RL >= 4 ? F4 : 0);
    constexpr int LDSW = (F1 + F2 + F3 + F4 - REGF) * 1024;                          // bytes of weights in LDS
    constexpr int SLOTS = 2 * D;
    static_assert(D == 1 || D == 2, "one or two pairs in flight");
    static_assert(RL >= 1 && RL <= (M4 > 0 ? 4 : 3), "layer 1 is always register-resident");
    // LDS: [weights of the layers behind RL][per wave: SLOTS tile slots of 8 KiB][per wave: 2 KiB logits staging]
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;
    const uint32_t batch = batch_arg & 0xFFFFu;

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
    char *const tiles = smem + LDSW + wave * (SLOTS * FUSED_TILE_BYTES);
    int32_t *const stage = (int32_t *)(smem + LDSW + RW_WPB * SLOTS * FUSED_TILE_BYTES + wave * 2048);

    const uint64_t n_pairs = n >> 6;            // the launcher hands this kernel whole 64-image pairs only
    // batches of `batch` (>= 2) consecutive pairs: a wave's first batch is static, later ones come from the device-wide counter on
    // the scalar unit (bnm_device.hpp, work_take_*); same protocol as fused_fc_dual_kernel's variant 6, with the difference that
    // the pair being SCHEDULED (DMA issued) runs D pairs ahead of the pair being computed.
    const uint32_t wave_id = blockIdx.x * RW_WPB + (uint32_t)wave, total_waves = gridDim.x * RW_WPB;
    const uint32_t words = batch_arg >> 16, wshift = (uint32_t)__builtin_ctz(words | 0x100u);
    const uint32_t my_word = wave_id & (words - 1u), first_dyn = total_waves >> wshift;
    uint32_t taken = 0;

    uint32_t voff[4];
#pragma unroll
    for (int u = 0; u < 4; u++) voff[u] = (uint32_t)(lane >> 4) * 256u + 16u * ((uint32_t)(lane & 15) ^ (uint32_t)(lane >> 4) ^ (4u * u));
    const uint32_t lds_wave = (uint32_t)(size_t)(__attribute__((address_space(3))) char *)tiles;
    const uint32_t rd_base = (uint32_t)j * 256u + 16u * ((uint32_t)h ^ (uint32_t)(j & 15));

    auto dma_tile = [&](uint64_t t, uint32_t slot_off) {
        const int8_t *base = images + t * (uint64_t)FUSED_TILE_BYTES;
        lds_dma_tile8_linear(lds_wave + slot_off, base, base + 4096, voff[0], voff[1], voff[2], voff[3]);
    };
    auto read_tile = [&](uint32_t slot_off, i32x4(&b)[KT0]) {
        const uint32_t rd = rd_base + slot_off;      // slot offsets are multiples of 8 KiB: the XOR below only touches bits 5..7
#pragma unroll
        for (int s = 0; s < KT0; s++) b[s] = *(const i32x4 *)(tiles + (rd ^ (32u * s)));
    };

    // ---- prologue: schedule the wave's first D pairs -------------------------------------------------------------------
    uint32_t cur = wave_id * batch;               // pair being computed (32-bit: the launcher refuses 2^31 pairs)
    uint32_t nxt = cur + 1u;                      // D == 2: the pair in the other buffer
    uint32_t last = D == 2 ? nxt : cur;           // newest scheduled pair, `left` = pairs of its batch behind it
    uint32_t left = batch - (uint32_t)D;
    const bool any = cur < n_pairs;
    if (any) {
        dma_tile(2ull * cur, 0);
        dma_tile(2ull * cur + 1ull, FUSED_TILE_BYTES);
        if constexpr (D == 2) {
            const uint32_t p1 = nxt < n_pairs ? nxt : cur;
            dma_tile(2ull * p1, 2 * FUSED_TILE_BYTES);
            dma_tile(2ull * p1 + 1ull, 3 * FUSED_TILE_BYTES);
        }
    }
    // the take that the loop's first advance beyond the static batch will consume
    if (left == 0u) work_take_issue(taken, work + 16u * my_word, 1u);
    else work_take_issue(taken, idle + 16u * wave_id, 0u);

    uint64_t img_prev = ((uint64_t)cur << 6) + (uint64_t)lane;
    uint32_t cls_prev = 0;
    uint64_t store_mask = 0;      // the deferred class-id store: empty exec mask in a wave's first iteration
    uint32_t par_off = 0;         // D == 2: byte offset of the current pair's two slots (0 or 16 KiB)
    constexpr int NWAIT = 8 + 16 * (D - 1);       // loads younger than the slot waited for

    while (cur < n_pairs) {
        // ---- advance the schedule by one pair -------------------------------------------------------------------------
        work_take_wait(taken);
        const uint32_t cand = left != 0u ? last + 1u : (((first_dyn + taken) << wshift) + my_word) * batch;
        const uint32_t left_new = left != 0u ? left - 1u : batch - 1u;
        // past the end the refill re-reads the pair being computed (constant wait counts, branch-free body)
        const uint64_t fill = cand < n_pairs ? cand : cur;

        bnm_wait_vmcnt<NWAIT>();
        i32x4 bA[KT0], bB[KT0];
        i32x16 a1A[M1], a1B[M1];
        read_tile(par_off, bA);
        mma<M1, KT0>(A1, bA, a1A);
        dma_tile(2ull * fill, par_off);
        bnm_wait_vmcnt<NWAIT>();
        store_ids_masked(cls_out + img_prev, cls_prev, store_mask);
        read_tile(par_off + FUSED_TILE_BYTES, bB);
        mma<M1, KT0>(A1, bB, a1B);
        dma_tile(2ull * fill + 1ull, par_off + FUSED_TILE_BYTES);
        {
            // the take for the advance after next (real when the pair just scheduled is the last but one of its batch), behind the
            // iteration's last tile read so that no LDS wait of the iteration covers its round trip
            uint32_t *const addr = left_new == 0u ? work + 16u * my_word : idle + 16u * wave_id;
            uint32_t one = 1u;
            asm volatile("" : "+s"(one));
            const uint32_t amount = left_new == 0u ? one : 0u;
            asm volatile("s_mov_b32 %0, %2\n\ts_atomic_add %0, %1, 0x0 glc"
                         : "=&{s95}"(taken) : "s"(addr), "s"(amount), "v"(a1B[0][15]), "v"(a1B[M1 - 1][15]) : "memory");
        }

        i32x4 p1A[M1], p1B[M1];
        relunorm_pack<M1, DBL>(a1A, p1A, h);
        i32x16 a2A[M2], a2B[M2];
        mma<M2, M1>(A2, p1A, a2A);
        relunorm_pack<M1, DBL>(a1B, p1B, h);
        mma<M2, M1>(A2, p1B, a2B);

        i32x4 p2A[M2], p2B[M2];
        relunorm_pack<M2, DBL>(a2A, p2A, h);
        i32x16 a3A[M3], a3B[M3];
        mma<M3, M2>(A3, p2A, a3A);
        relunorm_pack<M2, DBL>(a2B, p2B, h);
        mma<M3, M2>(A3, p2B, a3B);

        const uint64_t imgA = ((uint64_t)cur << 6) + (uint64_t)j, imgB = imgA + 32ull;
        uint32_t clsA, clsB;
        int32_t *const tile_a = logits_out + ((uint64_t)cur << 6) * n_classes, *const tile_b = tile_a + 32u * n_classes;
        if constexpr (M4 > 0) {
            i32x4 p3A[M3], p3B[M3];
            relunorm_pack<M3, DBL>(a3A, p3A, h);
            i32x16 a4A[M4], a4B[M4];
            mma<M4, M3>(A4, p3A, a4A);
            relunorm_pack<M3, DBL>(a3B, p3B, h);
            mma<M4, M3>(A4, p3B, a4B);
            clsA = argmax_rows<M4, NC8>(a4A, h);
            clsB = argmax_rows<M4, NC8>(a4B, h);
            if (logits_out) {
                if (n_classes <= 16u) {
                    store_logits_tile<M4, NC8, 0>(a4A, stage, tile_a, j, h, lane, n_classes);
                    store_logits_tile<M4, NC8, 0>(a4B, stage, tile_b, j, h, lane, n_classes);
                } else {
                    store_logits<M4>(a4A, tile_a + (uint32_t)j * n_classes, h, n_classes);
                    store_logits<M4>(a4B, tile_b + (uint32_t)j * n_classes, h, n_classes);
                }
            }
        } else {
            clsA = argmax_rows<M3, NC8>(a3A, h);
            clsB = argmax_rows<M3, NC8>(a3B, h);
            if (logits_out) {
                if (n_classes <= 16u) {
                    store_logits_tile<M3, NC8, 0>(a3A, stage, tile_a, j, h, lane, n_classes);
                    store_logits_tile<M3, NC8, 0>(a3B, stage, tile_b, j, h, lane, n_classes);
                } else {
                    store_logits<M3>(a3A, tile_a + (uint32_t)j * n_classes, h, n_classes);
                    store_logits<M3>(a3B, tile_b + (uint32_t)j * n_classes, h, n_classes);
                }
            }
        }
        // lanes 0..31 keep tile A's classes, lanes 32..63 tile B's: one 256-byte store per pair, issued in the next iteration
        img_prev = h ? imgB : imgA;
        cls_prev = h ? clsB : clsA;
        store_mask = ~0ull;
        if constexpr (D == 2) {
            cur = nxt;
            nxt = cand;
            par_off ^= 2u * FUSED_TILE_BYTES;
        } else {
            cur = cand;
        }
        last = cand;
        left = left_new;
    }
    work_take_wait(taken);      // the loop's final take must have returned before its register can be reused
    if (any) __builtin_nontemporal_store(cls_prev, cls_out + img_prev);
    bnm_wait_vmcnt<0>();        // no LDS-DMA may outlive the workgroup's LDS allocation
    work_block_leave_s(work, total_waves);
}

// ---- dispatch ------------------------------------------------------------------------------------------------------------
namespace {
typedef void (*regw_fn)(const int8_t *, uint64_t, const i32x4 *, uint32_t, uint32_t *, int32_t *, uint32_t *, uint32_t *, uint32_t);
struct RegwEntry {
    int M[4];
    bool dbl;
    int nc8;          // 0: any class count
    int rl, depth;
    regw_fn fn;
};
#define REGW(M1, M2, M3, M4, RL, DBL, NC8, D) { {M1, M2, M3, M4}, DBL, NC8, RL, D, fused_fc_regw_kernel<M1, M2, M3, M4, RL, DBL, NC8, D> }
#define REGW_ANY_AND_10(M1, M2, M3, M4, RL, DBL, D) REGW(M1, M2, M3, M4, RL, DBL, 2, D), REGW(M1, M2, M3, M4, RL, DBL, 0, D)
const RegwEntry kRegw[] = {
    // 256-96-96-96-N: the ternary model of BASELINE configs[2] on the MFMA path; 45 fragments, all in registers
    REGW_ANY_AND_10(3, 3, 3, 1, 4, true, 2),
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
    return w + RW_WPB * (2u * (uint32_t)e.depth * FUSED_TILE_BYTES + 2048u);
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
    if (!e || (a.n & 63ull) || !a.work || !a.idle || (a.n >> 6) >= (1ull << 31)) return hipErrorInvalidValue;
    if (a.n == 0) return hipSuccess;
    uint32_t batch = a.batch >= 2 ? a.batch : BNM_DUAL_DEFAULT_BATCH;       // (the D-deep schedule needs batches of >= 2 pairs)
    if (batch > 0xFFFFu) batch = 0xFFFFu;
    const uint64_t n_pairs = a.n >> 6;
    uint64_t want = (n_pairs + (uint64_t)RW_WPB * batch - 1) / ((uint64_t)RW_WPB * batch);
    uint64_t cap = grid_blocks > 0 ? (uint64_t)grid_blocks : (uint64_t)bnm_num_cus();     // one workgroup per CU
    if (cap * RW_WPB > BNM_WORK_DUMMY_WAVES) cap = BNM_WORK_DUMMY_WAVES / RW_WPB;
    const uint64_t blocks = want < cap ? want : cap;
    const uint32_t words = ((blocks * RW_WPB) & 7ull) == 0ull ? 8u : 1u;
    if (hipError_t err = allow_big_lds((const void *)e->fn); err != hipSuccess) return err;
    e->fn<<<dim3((unsigned)blocks), dim3(64 * RW_WPB), regw_lds_bytes(*e), s>>>(a.images, a.n, (const i32x4 *)a.frags, a.n_classes, a.cls,
                                                                              a.logits, a.work, a.idle, batch | (words << 16));
    return hipGetLastError();
}
