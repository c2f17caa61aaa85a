// Probe (round 4): how much VALU work can ONE wave overlap with its own MFMAs on gfx950?
// A wave that owns its SIMD (one wave per SIMD, bnm_fused_regw.hip) has no second wave to fill the gaps, so the cost model of
// "MFMA, then NV independent VALU instructions, repeat" decides the design.  Per group of {1 v_mfma_i32_32x32x32_i8 + NV VALU}
// the probe reports shader clocks, for
//   mode 0  accumulators in architectural VGPRs (VGPR form, what a kernel whose VALU code reads the sums needs), A/B in VGPRs
//   mode 1  accumulators in AccVGPRs (AGPR form), A/B in VGPRs
//   mode 2  VGPR form, A operand in AccVGPRs
//   mode 3  VALU only            mode 4  MFMA only (VGPR form)
//   mode 5  VGPR form, the VALU instructions READ another MFMA accumulator tuple (as ReLUNorm does)
//   mode 6  as 5, and the VALU instructions are SDWA byte writes (dst_unused:UNUSED_PRESERVE)
//   mode 7  VALU only, 4-byte VOP2 encodings (v_max_i32_e32) instead of 8-byte VOP3 ones
// and every VALU-only / mixed row again with TWO waves per SIMD (512-thread workgroups): does a second wave double the issue rate?
// Three accumulators are cycled so that no MFMA waits for its own predecessor.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O2 -o mfma_valu_overlap profiles/probes/mfma_valu_overlap.hip && ./mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

template <int MODE, int NV>
__global__ __launch_bounds__(512, 1) void probe(uint64_t *out, int iters, int seed) {
    i32x16 acc[3], other;
    i32x4 a, b;
    int t[16], x[16];
    for (int i = 0; i < 16; i++) {
        acc[0][i] = seed + i; acc[1][i] = seed - i; acc[2][i] = seed ^ i; other[i] = seed * (i + 1);
        t[i] = seed + 3 * i; x[i] = seed - 7 * i;
    }
    for (int i = 0; i < 4; i++) { a[i] = seed + i; b[i] = seed - i; }
    i32x4 aa = a;
    if (MODE == 2) asm volatile("; pin" : "+a"(aa));
    i32x16 cacc[3] = {acc[0], acc[1], acc[2]};
    if (MODE == 1) asm volatile("; pin" : "+a"(cacc[0]), "+a"(cacc[1]), "+a"(cacc[2]));
    __builtin_amdgcn_s_barrier();
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int g = 0; g < 3; g++) {
            if (MODE == 0 || MODE == 4 || MODE == 5 || MODE == 6)
                asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(acc[g]) : "v"(a), "v"(b));
            else if (MODE == 1)
                asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+a"(cacc[g]) : "v"(a), "v"(b));
            else if (MODE == 2)
                asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(acc[g]) : "a"(aa), "v"(b));
            if (MODE != 4) {
#pragma unroll
                for (int k = 0; k < NV; k++) {
                    if (MODE == 5) asm volatile("v_med3_i32 %0, %1, 0, %2" : "=v"(t[k % 16]) : "v"(other[k % 16]), "v"(x[k % 16]));
                    else if (MODE == 6) asm volatile("v_lshrrev_b32_sdwa %0, %1, %2 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD"
                                                     : "+v"(t[k % 16]) : "v"(x[0]), "v"(other[k % 16]));
                    else if (MODE == 7) asm volatile("v_max_i32_e32 %0, %1, %2" : "=v"(t[k % 16]) : "v"(x[(k + 1) % 16]), "v"(x[k % 16]));
                    else asm volatile("v_med3_i32 %0, %1, 0, %2" : "=v"(t[k % 16]) : "v"(x[(k + 1) % 16]), "v"(x[k % 16]));
                }
            }
        }
    }
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
    const uint64_t t1 = __builtin_readcyclecounter();
    int sink = 0;
    if (MODE == 1) asm volatile("; unpin" : "+a"(cacc[0]), "+a"(cacc[1]), "+a"(cacc[2]));
    for (int i = 0; i < 16; i++) sink += t[i] + acc[0][i] + acc[1][i] + acc[2][i] + cacc[0][i] + cacc[1][i] + cacc[2][i];
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = (uint64_t)sink; }
}

template <int MODE, int NV>
void run(uint64_t *d_out, const char *what) {
    const int iters = 20000;
    uint64_t h[2];
    double c[2];
    for (int w = 1; w <= 2; w++) {      // waves per SIMD
        probe<MODE, NV><<<256, 256 * w>>>(d_out, iters, 3);
        hipDeviceSynchronize();
        probe<MODE, NV><<<256, 256 * w>>>(d_out, iters, 5);
        hipMemcpy(h, d_out, sizeof h, hipMemcpyDeviceToHost);
        c[w - 1] = (double)h[0] / (3.0 * iters);
    }
    printf("mode %d  NV %2d  %-44s %7.2f clocks per group (one wave per SIMD)   %7.2f per wave, two waves per SIMD\n", MODE, NV, what, c[0], c[1]);
}

#define SWEEP(MODE, WHAT) run<MODE, 0>(d, WHAT); run<MODE, 4>(d, WHAT); run<MODE, 8>(d, WHAT); run<MODE, 12>(d, WHAT); run<MODE, 16>(d, WHAT);

int main() {
    uint64_t *d;
    hipMalloc(&d, 64);
    run<3, 4>(d, "VALU only");
    run<3, 8>(d, "VALU only");
    run<3, 16>(d, "VALU only");
    run<7, 4>(d, "VALU only, VOP2");
    run<7, 16>(d, "VALU only, VOP2");
    run<4, 0>(d, "MFMA only, VGPR form");
    SWEEP(0, "VGPR form, A/B in VGPRs")
    SWEEP(1, "AGPR form (sums in AccVGPRs)")
    SWEEP(2, "VGPR form, A in AccVGPRs")
    SWEEP(5, "VGPR form, VALU reads another sums tuple")
    SWEEP(6, "VGPR form, SDWA byte writes")
    hipFree(d);
    return 0;
}
