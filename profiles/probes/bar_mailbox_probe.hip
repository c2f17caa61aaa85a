// Probe (round 6): what does a host <-> resident-wave message cost on this box, by where the mailbox lives?
//   A  request + response in page-locked HOST memory (the GPU polls over PCIe: a non-posted read per poll)
//   B  request in fine-grained DEVICE memory written by the host through the PCIe BAR (if the platform lets the host map it),
//      response in host memory (both directions posted writes)
// One wave echoes sequence numbers: host writes seq, the wave answers seq.  Prints the round trip per message.
//   hipcc --offload-arch=gfx950 -O2 bar_mailbox_probe.hip -o bar_mailbox_probe && ./bar_mailbox_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>

__global__ void echo(volatile uint32_t *req, uint32_t *resp, uint32_t n) {
    uint32_t last = 0;
    while (last < n) {
        uint32_t v = __hip_atomic_load((uint32_t *)req, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (v == last + 1) {
            __hip_atomic_store(resp, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            last = v;
        }
    }
}

static double run(volatile uint32_t *req_host_view, uint32_t *req_dev, volatile uint32_t *resp_host, uint32_t *resp_dev, uint32_t n) {
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    *req_host_view = 0;
    *resp_host = 0;
    echo<<<1, 64, 0, s>>>((volatile uint32_t *)req_dev, resp_dev, n);
    // warm
    uint32_t seq = 0;
    auto t0 = std::chrono::steady_clock::now();
    for (; seq < n;) {
        if (seq == 1000) t0 = std::chrono::steady_clock::now();
        seq++;
        *req_host_view = seq;
        std::atomic_thread_fence(std::memory_order_seq_cst);
        while (*resp_host != seq) {}
    }
    auto t1 = std::chrono::steady_clock::now();
    hipStreamSynchronize(s);
    hipStreamDestroy(s);
    return std::chrono::duration<double, std::micro>(t1 - t0).count() / (n - 1000);
}

int main() {
    int large_bar = -1;
    hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, 0);
    printf("hipDeviceAttributeIsLargeBar = %d\n", large_bar);
    uint32_t *h = nullptr, *hd = nullptr;
    hipHostMalloc(&h, 4096, hipHostMallocMapped);
    hipHostGetDevicePointer((void **)&hd, h, 0);
    const uint32_t n = 20000;
    printf("A: request + response in page-locked host memory: %.2f us per round trip\n", run(h, hd, h + 64, hd + 64, n));
    // B: fine-grained device memory, host view = the same pointer (large BAR + fine-grained: the runtime maps it for the CPU)
    uint32_t *d = nullptr;
    hipError_t e = hipExtMallocWithFlags((void **)&d, 4096, hipDeviceMallocFinegrained);
    if (e != hipSuccess) { printf("B: hipExtMallocWithFlags(finegrained) failed: %s\n", hipGetErrorString(e)); return 0; }
    hipMemset(d, 0, 4096);
    hipDeviceSynchronize();
    hipPointerAttribute_t a;
    memset(&a, 0, sizeof a);
    e = hipPointerGetAttributes(&a, d);
    printf("B: fine-grained device allocation: hostPointer %p devicePointer %p type %d (%s)\n", a.hostPointer, a.devicePointer, (int)a.type, hipGetErrorString(e));
    if (getenv("BNM_PROBE_TOUCH")) {      // dereferencing device memory from the host faults where the BAR does not cover it: opt-in
        printf("B: request in device memory through the BAR: %.2f us per round trip\n", run(d, d, h + 64, hd + 64, n));
    }
    return 0;
}
