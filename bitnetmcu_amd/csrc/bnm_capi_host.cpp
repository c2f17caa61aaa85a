// C ABI, host-pointer inference (bnm_infer_host, bnm_infer_host_activations): the zero-copy latency path behind Inference(),
// the pipelined page-locked staging path, the synchronous tap path.
#include "bnm_capi_internal.hpp"

using namespace bnm_internal;

namespace {
constexpr uint64_t kChunk = 1ull << 20;   // images per internal chunk of the staged path
}

extern "C" {

// ---- host-pointer inference ---------------------------------------------------------------------------------------
// (1) n <= kLatencyMax: zero-copy.  The images are copied into a persistent page-locked buffer the GPU addresses
//     directly, the kernel reads it over PCIe and writes class ids (and logits) into another such buffer: one launch and one
//     stream wait per call, no hipMemcpy.  This is what the drop-in Inference() symbol runs (one image per call).
// (2) larger batches: two slots of page-locked staging + device buffers, each with its own stream.  Host threads copy chunk
//     k+1 into its slot while the DMA engines move chunk k and return chunk k-1's results; compute of consecutive chunks is
//     chained by an event (it shares per-context scratch on the CNN / layer-wise paths), which costs nothing: the kernels
//     take microseconds per chunk, the PCIe transfer a millisecond.
// (3) the activation tap (parity/debug): the plain synchronous path.
constexpr uint64_t kLatencyMax = 64;
constexpr uint64_t kHostChunk = 1ull << 18;      // images per pipelined chunk: 64 MiB of image bytes

static int infer_host_small(bnm_ctx *c, const int8_t *images, uint64_t n, uint32_t *cls, int32_t *logits) {
    const uint32_t ncls = c->model.num_classes();
    if (!c->lat_stream) {
        if (int e = c->lat_in.ensure(kLatencyMax * 256)) return e;
        if (int e = c->lat_cls.ensure(kLatencyMax * 4)) return e;
        if (int e = c->lat_logits.ensure(kLatencyMax * (size_t)ncls * 4)) return e;
        HIP_TRY(hipStreamCreateWithFlags(&c->lat_stream, hipStreamNonBlocking));
    }
    std::memcpy(c->lat_in.host, images, (size_t)n * 256);
    // class ids are <= 255: pre-set every slot to a sentinel and watch the page-locked words change — the kernel's stores to
    // fine-grained host memory are visible as soon as they are written, several microseconds before the stream's completion
    // signal has been processed.  (Each word is written exactly once, by the last instruction that touches the image, so a
    // slot that changed also means its image has been read - which is why the dual kernel's deferred store is masked off in a
    // wave's first iteration instead of writing a placeholder; logits have no spare value and take the stream wait.)
    volatile uint32_t *out = (volatile uint32_t *)c->lat_cls.host;
    const bool spin = !logits && c->lat_spin;
    if (spin) for (uint64_t i = 0; i < n; i++) out[i] = 0xFFFFFFFFu;
    if (int e = infer_device_locked(c, (const int8_t *)c->lat_in.dev, n, (uint32_t *)c->lat_cls.dev,
                                    logits ? (int32_t *)c->lat_logits.dev : nullptr, nullptr, 0, c->lat_stream))
        return e;
    bool done = false;
    if (spin) {
        const auto t0 = std::chrono::steady_clock::now();
        uint64_t i = 0;
        for (unsigned polls = 0; i < n;) {
            if (out[i] != 0xFFFFFFFFu) { i++; continue; }
            if ((++polls & 1023u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(500)) break;
        }
        done = i == n;      // else: slow start (first launch, clock ramp) — fall back to the stream wait
    }
    if (!done) HIP_TRY(hipStreamSynchronize(c->lat_stream));
    if (cls) std::memcpy(cls, c->lat_cls.host, (size_t)n * 4);
    if (logits) std::memcpy(logits, c->lat_logits.host, (size_t)n * ncls * 4);
    return BNM_OK;
}

static int infer_host_pipelined(bnm_ctx *c, const int8_t *images, uint64_t n, uint32_t *cls, int32_t *logits) {
    const uint32_t ncls = c->model.num_classes();
    if (!c->copier) {
        unsigned hw = std::thread::hardware_concurrency();
        c->copier = new ParallelCopier(c->host_threads ? c->host_threads : hw >= 16 ? 8u : hw >= 4 ? hw / 2u : 0u);
    }
    // whatever exit the function takes, nothing may stay in flight on the slot streams: the next call reuses the page-locked
    // buffers at once (an error return used to leave DMA running into / out of them)
    struct Quiesce {
        bnm_ctx *c;
        ~Quiesce() {
            for (auto &sl : c->slot) {
                if (sl.stream) (void)hipStreamSynchronize(sl.stream);
                sl.count = 0;
            }
        }
    } quiesce{c};
    for (auto &sl : c->slot) {
        if (!sl.stream) HIP_TRY(hipStreamCreateWithFlags(&sl.stream, hipStreamNonBlocking));
        if (!sl.computed) HIP_TRY(hipEventCreateWithFlags(&sl.computed, hipEventDisableTiming));
        if (int e = sl.in.ensure(kHostChunk * 256)) return e;
        if (int e = sl.d_in.ensure(kHostChunk * 256)) return e;
        if (int e = sl.cls.ensure(kHostChunk * 4)) return e;
        if (int e = sl.d_cls.ensure(kHostChunk * 4)) return e;
        if (logits) {
            if (int e = sl.logits.ensure(kHostChunk * (size_t)ncls * 4)) return e;
            if (int e = sl.d_logits.ensure(kHostChunk * (size_t)ncls * 4)) return e;
        }
        sl.count = 0;
    }
    auto drain = [&](bnm_ctx::HostSlot &sl) -> int {      // results of the chunk in flight on this slot -> caller's arrays
        if (!sl.count) return BNM_OK;
        HIP_TRY(hipStreamSynchronize(sl.stream));
        if (cls) std::memcpy(cls + sl.off, sl.cls.host, (size_t)sl.count * 4);
        if (logits) std::memcpy(logits + sl.off * ncls, sl.logits.host, (size_t)sl.count * ncls * 4);
        sl.count = 0;
        return BNM_OK;
    };
    int k = 0;
    hipEvent_t prev_computed = nullptr;
    for (uint64_t off = 0; off < n; off += kHostChunk, k ^= 1) {
        bnm_ctx::HostSlot &sl = c->slot[k];
        const uint64_t cn = n - off < kHostChunk ? n - off : kHostChunk;
        if (int e = drain(sl)) return e;
        c->copier->run(sl.in.host, images + off * 256, (size_t)cn * 256);
        HIP_TRY(hipMemcpyAsync(sl.d_in.p, sl.in.host, (size_t)cn * 256, hipMemcpyHostToDevice, sl.stream));
        if (prev_computed) HIP_TRY(hipStreamWaitEvent(sl.stream, prev_computed, 0));
        if (int e = infer_device_locked(c, (const int8_t *)sl.d_in.p, cn, (uint32_t *)sl.d_cls.p,
                                        logits ? (int32_t *)sl.d_logits.p : nullptr, nullptr, 0, sl.stream))
            return e;
        HIP_TRY(hipEventRecord(sl.computed, sl.stream));
        prev_computed = sl.computed;
        HIP_TRY(hipMemcpyAsync(sl.cls.host, sl.d_cls.p, (size_t)cn * 4, hipMemcpyDeviceToHost, sl.stream));
        if (logits) HIP_TRY(hipMemcpyAsync(sl.logits.host, sl.d_logits.p, (size_t)cn * ncls * 4, hipMemcpyDeviceToHost, sl.stream));
        sl.off = off;
        sl.count = cn;
    }
    if (int e = drain(c->slot[k])) return e;       // older chunk first
    return drain(c->slot[k ^ 1]);
}

static int infer_host_impl(bnm_ctx *c, const int8_t *images, uint64_t n, uint32_t *cls, int32_t *logits, int8_t *acts,
                           uint32_t acts_stride) {
    if (!c) return fail(BNM_EINVAL, "null ctx");
    if (n && (!images || (!cls && !acts))) return fail(BNM_EINVAL, "null host pointer");
    std::lock_guard<std::mutex> g(c->mu);
    DeviceGuard dg(c->device);
    HIP_TRY(dg.err);
    if (!n) return BNM_OK;
    if (!acts) {
        if (n <= kLatencyMax) return infer_host_small(c, images, n, cls, logits);
        if (c->host_mode == 0) return infer_host_pipelined(c, images, n, cls, logits);
    }
    const uint32_t ncls = c->model.num_classes();
    for (uint64_t off = 0; off < n; off += kChunk) {
        uint64_t cn = n - off < kChunk ? n - off : kChunk;
        if (int e = c->stage_img.ensure((size_t)cn * 256)) return e;
        if (int e = c->stage_cls.ensure((size_t)cn * 4)) return e;
        if (logits) if (int e = c->stage_logits.ensure((size_t)cn * ncls * 4)) return e;
        ScopedDev tap;
        if (acts) if (int e = tap.ensure((size_t)cn * acts_stride)) return e;
        HIP_TRY(hipMemcpy(c->stage_img.p, images + off * 256, (size_t)cn * 256, hipMemcpyHostToDevice));
        int e = infer_device_locked(c, (const int8_t *)c->stage_img.p, cn, (uint32_t *)c->stage_cls.p,
                                    logits ? (int32_t *)c->stage_logits.p : nullptr, acts ? (int8_t *)tap.p : nullptr, acts_stride, nullptr);
        if (e == BNM_OK) {
            hipError_t he = hipDeviceSynchronize();
            if (he != hipSuccess) e = fail(BNM_EHIP, std::string("kernel execution: ") + hipGetErrorString(he));
        }
        if (e == BNM_OK && cls) HIP_TRY(hipMemcpy(cls + off, c->stage_cls.p, (size_t)cn * 4, hipMemcpyDeviceToHost));
        if (e == BNM_OK && logits)
            HIP_TRY(hipMemcpy(logits + off * ncls, c->stage_logits.p, (size_t)cn * ncls * 4, hipMemcpyDeviceToHost));
        if (e == BNM_OK && acts) HIP_TRY(hipMemcpy(acts + off * acts_stride, tap.p, (size_t)cn * acts_stride, hipMemcpyDeviceToHost));
        if (e != BNM_OK) return e;
    }
    return BNM_OK;
}

int bnm_infer_host(bnm_ctx *c, const int8_t *images, uint64_t n, uint32_t *cls, int32_t *logits) {
    return infer_host_impl(c, images, n, cls, logits, nullptr, 0);
}

int bnm_infer_host_activations(bnm_ctx *c, const int8_t *images, uint64_t n, int8_t *acts, uint32_t acts_stride) {
    if (!c || !acts) return fail(BNM_EINVAL, "null argument");
    uint32_t need = c->model.kind == BNM_KIND_CNN ? c->channels * 4u : 0u;
    for (auto &l : c->fc) need += l.info.n_output;
    if (acts_stride < need) return fail(BNM_EINVAL, "acts_stride too small");
    std::vector<uint32_t> cls(n);
    return infer_host_impl(c, images, n, cls.data(), nullptr, acts, acts_stride);
}

}  // extern "C"
