// Probe (round 2): how many s_atomic_add takes per second does gfx950 serve on ONE word, on 8 words, on one word per wave?
//   hipcc --offload-arch=gfx950 -O2 s_atomic_rate.hip -o s_atomic_rate && ./s_atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ unsigned s_take(unsigned *p) {
    unsigned r, one = 1;
    asm volatile("s_mov_b32 %0, %2\n\ts_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=&s"(r) : "s"(p), "s"(one) : "memory");
    return r;
}
__global__ void k(unsigned *ctr, unsigned n_words, int iters, unsigned *sink) {
    const unsigned wave = blockIdx.x * (blockDim.x >> 6) + (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    unsigned *p = ctr + 16u * (wave % n_words);
    unsigned acc = 0;
    for (int i = 0; i < iters; i++) acc += s_take(p);
    if ((threadIdx.x & 63) == 0) sink[wave] = acc;
}
int main() {
    const int blocks = 512, threads = 256, iters = 2000, waves = blocks * threads / 64;
    unsigned *ctr, *sink;
    (void)hipMalloc(&ctr, 64 * (size_t)waves);
    (void)hipMalloc(&sink, 4 * (size_t)waves);
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    for (unsigned n_words : {1u, 8u, 64u, (unsigned)waves}) {
        (void)hipMemset(ctr, 0, 64 * (size_t)waves);
        k<<<blocks, threads>>>(ctr, n_words, 10, sink);
        (void)hipEventRecord(a);
        k<<<blocks, threads>>>(ctr, n_words, iters, sink);
        (void)hipEventRecord(b);
        if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
        float ms = 0;
        (void)hipEventElapsedTime(&ms, a, b);
        printf("%6u word(s), %d waves x %d takes, each waited for: %.3f ms -> %.1f M takes/s (%.0f ns per take and wave)\n", n_words, waves, iters, ms,
               (double)waves * iters / ms / 1e3, ms * 1e6 / iters);
    }
    return 0;
}
