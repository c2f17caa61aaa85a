// Probe (round 2): does the scalar-memory atomic s_atomic_add work on gfx950, and is it coherent (a) among the waves of one
// workgroup on a per-workgroup counter, (b) across the whole device on a single counter?
//   hipcc --offload-arch=gfx950 -O2 s_atomic_probe.hip -o s_atomic_probe && ./s_atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__device__ __forceinline__ unsigned s_take(unsigned *p) {
    unsigned r, one = 1;
    asm volatile("s_mov_b32 %0, %2\n\ts_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=&s"(r) : "s"(p), "s"(one) : "memory");
    return r;
}
__global__ void k(unsigned *per_block, unsigned *global_ctr, unsigned *out_block, unsigned *out_global, int iters) {
    const unsigned wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int i = 0; i < iters; i++) {
        unsigned a = s_take(per_block + 16 * blockIdx.x);
        unsigned b = s_take(global_ctr);
        if ((threadIdx.x & 63) == 0) {
            out_block[((size_t)blockIdx.x * nw + wave) * iters + i] = a;
            out_global[((size_t)blockIdx.x * nw + wave) * iters + i] = b;
        }
    }
}
int main() {
    const int blocks = 512, threads = 512, iters = 64, nw = threads / 64;
    unsigned *pb, *g, *ob, *og;
    hipMalloc(&pb, blocks * 64); hipMalloc(&g, 64);
    hipMalloc(&ob, (size_t)blocks * nw * iters * 4); hipMalloc(&og, (size_t)blocks * nw * iters * 4);
    hipMemset(pb, 0, blocks * 64); hipMemset(g, 0, 64);
    k<<<blocks, threads>>>(pb, g, ob, og, iters);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
    std::vector<unsigned> hb((size_t)blocks * nw * iters), hg(hb.size());
    hipMemcpy(hb.data(), ob, hb.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(hg.data(), og, hg.size() * 4, hipMemcpyDeviceToHost);
    int bad_blocks = 0;
    for (int b = 0; b < blocks; b++) {
        std::vector<unsigned> v(hb.begin() + (size_t)b * nw * iters, hb.begin() + (size_t)(b + 1) * nw * iters);
        std::sort(v.begin(), v.end());
        bool ok = true;
        for (size_t i = 0; i < v.size(); i++) ok = ok && v[i] == i;
        bad_blocks += !ok;
    }
    std::sort(hg.begin(), hg.end());
    size_t dup = 0, mx = hg.back();
    for (size_t i = 1; i < hg.size(); i++) dup += hg[i] == hg[i - 1];
    unsigned gfinal; hipMemcpy(&gfinal, g, 4, hipMemcpyDeviceToHost);
    printf("per-workgroup counters: %d of %d workgroups saw a permutation of 0..%d -> %s\n", blocks - bad_blocks, blocks, nw * iters - 1, bad_blocks ? "BROKEN" : "ok");
    printf("device-wide counter: %zu takes, %zu duplicates, max %zu, final value %u -> %s\n", hg.size(), dup, mx, gfinal, (dup == 0 && gfinal == hg.size()) ? "coherent" : "NOT coherent across the device");
    return 0;
}
