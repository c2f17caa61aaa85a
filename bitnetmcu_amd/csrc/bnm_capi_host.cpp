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

// ---- one image through the resident kernel (opt-in; bnm_persist_kernel.hpp has the protocol) ---------------------------------------
// The mailbox words are plain page-locked memory: the tags are written with release stores after their line's payload (x86 keeps
// stores in program order; the fence keeps the compiler from reordering them), the response word is read with acquire loads.
constexpr uint32_t kBoxLines = 5, kBoxQuit = 80, kBoxResponse = 96, kBoxDwords = 128;      // = BNM_BOX_* of bnm_persist_kernel.hpp

static std::mutex g_boxes_mu;
static std::vector<volatile uint32_t *> g_boxes;      // live mailboxes: at process exit every resident kernel is told to leave

static void persist_quit_all() {
    std::lock_guard<std::mutex> g(g_boxes_mu);
    for (volatile uint32_t *b : g_boxes) b[kBoxQuit] = 1u;
}

static bool persist_wanted(bnm_ctx *c) {
    if (c->persist_mode < 0) {
        const char *e = std::getenv("BNM_PERSISTENT");
        c->persist_mode = (e && *e && std::strcmp(e, "0") != 0) ? 1 : 0;
        if (const char *t = std::getenv("BNM_PERSISTENT_IDLE_US")) {
            const long v = std::atol(t);
            if (v >= 100 && v <= 10000000) c->persist_idle_us = (uint32_t)v;
        }
    }
    return c->persist_mode == 1 && c->path == BNM_PATH_FUSED_MFMA && c->model.kind == BNM_KIND_FC && c->generic_ok &&
           bnmk_persistent_supported(c->gdesc, c->shape.dbl);
}

static int persist_start(bnm_ctx *c) {
    if (!c->persist_stream) {
        if (int e = c->persist_box.ensure(kBoxDwords * 4)) return e;
        std::memset(c->persist_box.host, 0, kBoxDwords * 4);
        HIP_TRY(hipStreamCreateWithFlags(&c->persist_stream, hipStreamNonBlocking));
        std::lock_guard<std::mutex> g(g_boxes_mu);
        if (g_boxes.empty()) std::atexit(persist_quit_all);
        g_boxes.push_back((volatile uint32_t *)c->persist_box.host);
    }
    // the kernel answers every call after `persist_seq - 1`... the call in hand is already in the mailbox when a kernel is (re)started
    const uint32_t answered = ((volatile uint32_t *)c->persist_box.host)[kBoxResponse] >> 8;
    HIP_TRY(bnmk_persistent_launch(c->gdesc, c->shape.dbl, c->gfrags, (uint32_t *)c->persist_box.dev, answered, (uint64_t)c->persist_idle_us * 100ull,
                                   c->persist_stream));
    c->persist_running = true;
    return BNM_OK;
}

// tells a running kernel to leave and waits for it (context teardown, bnm_ctx_set_persistent(0))
extern "C++" void bnm_internal::persist_stop(bnm_ctx *c) {
    if (!c->persist_stream) return;
    volatile uint32_t *box = (volatile uint32_t *)c->persist_box.host;
    box[kBoxQuit] = 1u;
    (void)hipStreamSynchronize(c->persist_stream);
    box[kBoxQuit] = 0u;
    c->persist_running = false;
}

extern "C++" void bnm_internal::persist_release(bnm_ctx *c) {
    if (!c->persist_stream) return;
    persist_stop(c);
    {
        std::lock_guard<std::mutex> g(g_boxes_mu);
        for (size_t i = 0; i < g_boxes.size(); i++)
            if (g_boxes[i] == (volatile uint32_t *)c->persist_box.host) { g_boxes.erase(g_boxes.begin() + (long)i); break; }
    }
    (void)hipStreamDestroy(c->persist_stream);
    c->persist_stream = nullptr;
    c->persist_box.release();
}

static int infer_host_persistent(bnm_ctx *c, const int8_t *image, uint32_t *cls) {
    if (!c->persist_stream || !c->persist_running) {
        if (int e = persist_start(c)) return e;
    }
    volatile uint32_t *box = (volatile uint32_t *)c->persist_box.host;
    const uint32_t seq = c->persist_seq == 0xFFFFFFu ? 1u : c->persist_seq + 1u;
    c->persist_seq = seq;
    uint32_t words[64];
    std::memcpy(words, image, 256);
    for (uint32_t l = 0; l < kBoxLines; l++) {
        const uint32_t cnt = l < 4u ? 15u : 4u;
        for (uint32_t k = 0; k < cnt; k++) box[16u * l + k] = words[15u * l + k];
        std::atomic_thread_fence(std::memory_order_release);
        box[16u * l + 15u] = seq;
    }
    const auto t0 = std::chrono::steady_clock::now();
    auto checked = t0;
    for (unsigned polls = 0;;) {
        const uint32_t r = box[kBoxResponse];
        if ((r >> 8) == seq) {
            std::atomic_thread_fence(std::memory_order_acquire);
            *cls = r & 0xFFu;
            return BNM_OK;
        }
        if ((++polls & 255u) != 0) continue;
        const auto now = std::chrono::steady_clock::now();
        if (now - checked < std::chrono::microseconds(50)) continue;
        checked = now;
        // slow: did the kernel leave (idle limit) before it saw this call?  Then another one takes it from the mailbox.
        const hipError_t q = hipStreamQuery(c->persist_stream);
        if (q == hipSuccess) {
            if ((box[kBoxResponse] >> 8) == seq) continue;      // (it answered and left)
            if (int e = persist_start(c)) return e;
        } else if (q != hipErrorNotReady) {
            c->persist_running = false;
            return fail(BNM_EHIP, std::string("the resident inference kernel failed: ") + hipGetErrorString(q));
        } else if (now - t0 > std::chrono::seconds(10)) {
            return fail(BNM_EHIP, "the resident inference kernel did not answer within 10 s");
        }
    }
}

extern "C" int bnm_ctx_persistent_last_call(bnm_ctx *c, uint32_t *wall_10ns, uint32_t *shader_clocks) {
    if (!c || !wall_10ns || !shader_clocks) return fail(BNM_EINVAL, "null argument");
    std::lock_guard<std::mutex> g(c->mu);
    if (!c->persist_box.host) return fail(BNM_EINVAL, "no resident kernel has run on this context");
    volatile uint32_t *box = (volatile uint32_t *)c->persist_box.host;
    *wall_10ns = box[112];
    *shader_clocks = box[113];
    return BNM_OK;
}

static int infer_host_small(bnm_ctx *c, const int8_t *images, uint64_t n, uint32_t *cls, int32_t *logits) {
    const uint32_t ncls = c->model.num_classes();
    if (n == 1 && cls && !logits && persist_wanted(c)) {
        c->last_kernel = "persistent_inference_kernel";
        return infer_host_persistent(c, images, cls);
    }
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
