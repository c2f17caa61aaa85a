// Internal declarations shared by the translation units of the C ABI (bnm_capi*.cpp): error state, device / page-locked
// buffers, the context structure, and the handful of functions one unit needs from another.  Not installed; the public interface
// is include/bitnetmcu_hip.h.  Host side only - there is NO CPU compute path behind the ABI: if HIP is unusable, the reference-ABI
// functions abort() and the bnm_* functions return BNM_EHIP.
#pragma once
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <map>
#include <mutex>
#include <thread>
#include <string>
#include <vector>
#include "bnm_device.hpp"
#include "bnm_kernels.h"
#include "bnm_model.hpp"
#include "bnm_capi_error.hpp"
#ifdef BNM_DIAG
#include "bnm_diag.h"
#endif

namespace bnm_internal {

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess)                                                                           \
            return fail(BNM_EHIP, std::string(#expr) + ": " + hipGetErrorString(_e));                   \
    } while (0)

inline uint32_t round_up(uint32_t v, uint32_t m) { return (v + m - 1u) / m * m; }
void persist_stop(::bnm_ctx *c);         // bnm_capi_host.cpp: the resident one-image kernel leaves (and has left on return)
void persist_release(::bnm_ctx *c);      // ... and its stream and mailbox are given back

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    // set when a launch that reads / writes the buffer was CAPTURED into a HIP graph: the graph holds the address for as long as
    // it may be replayed, so the buffer is neither grown (that frees it) nor released before its context goes
    bool frozen = false;
    int ensure(size_t need) {
        if (need <= bytes) return BNM_OK;
        if (frozen)
            return fail(BNM_EUNSUPPORTED, "this stream's scratch buffer is referenced by a captured graph and cannot grow: run calls "
                                          "larger than the captured ones on another stream (or capture the largest call first)");
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
        HIP_TRY(hipMalloc(&p, need));
        bytes = need;
        return BNM_OK;
    }
    void release(bool even_if_frozen = false) {
        if (frozen && !even_if_frozen) return;
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
        frozen = false;
    }
};

// scope-owned device buffer (temporary allocations inside one API call)
struct ScopedDev : DevBuf {
    ~ScopedDev() { release(); }
};

// page-locked host memory that the GPU can address directly (zero-copy): the latency path's buffers and the staging
// buffers of the pipelined host path
struct PinBuf {
    void *host = nullptr, *dev = nullptr;
    size_t bytes = 0;
    int ensure(size_t need) {
        if (need <= bytes) return BNM_OK;
        release();
        HIP_TRY(hipHostMalloc(&host, need, hipHostMallocMapped));
        HIP_TRY(hipHostGetDevicePointer(&dev, host, 0));
        bytes = need;
        return BNM_OK;
    }
    void release() {
        if (host) (void)hipHostFree(host);
        host = dev = nullptr;
        bytes = 0;
    }
};

// memcpy on several host threads (a pageable -> pinned staging copy runs at one core's ~10 GB/s otherwise, a fifth of what
// PCIe Gen5 x16 moves).  Persistent workers; run() returns when every slice has been copied.
class ParallelCopier {
public:
    explicit ParallelCopier(unsigned workers) {
        for (unsigned i = 0; i < workers; i++) th_.emplace_back([this, i, workers] { loop(i, workers); });
    }
    ~ParallelCopier() {
        {
            std::lock_guard<std::mutex> g(mu_);
            stop_ = true;
            gen_++;
        }
        cv_.notify_all();
        for (auto &t : th_) t.join();
    }
    void run(void *dst, const void *src, size_t bytes) {
        if (th_.empty() || bytes < (1u << 20)) { std::memcpy(dst, src, bytes); return; }
        {
            std::lock_guard<std::mutex> g(mu_);
            dst_ = (char *)dst; src_ = (const char *)src; bytes_ = bytes;
            pending_ = (unsigned)th_.size();
            gen_++;
        }
        cv_.notify_all();
        std::unique_lock<std::mutex> lk(mu_);
        done_.wait(lk, [this] { return pending_ == 0; });
    }

private:
    void loop(unsigned i, unsigned n) {
        uint64_t seen = 0;
        for (;;) {
            char *d; const char *s; size_t b;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return gen_ != seen; });
                seen = gen_;
                if (stop_) return;
                d = dst_; s = src_; b = bytes_;
            }
            const size_t per = ((b + n - 1) / n + 4095) & ~size_t(4095);
            const size_t lo = (size_t)i * per, hi = lo + per < b ? lo + per : b;
            if (lo < b) std::memcpy(d + lo, s + lo, hi - lo);
            {
                std::lock_guard<std::mutex> g(mu_);
                if (--pending_ == 0) done_.notify_all();
            }
        }
    }
    std::vector<std::thread> th_;
    std::mutex mu_;
    std::condition_variable cv_, done_;
    char *dst_ = nullptr;
    const char *src_ = nullptr;
    size_t bytes_ = 0;
    unsigned pending_ = 0;
    uint64_t gen_ = 0;
    bool stop_ = false;
};

struct FcDev {
    bnm_layer_info info{};
    uint32_t n_real = 0;      // activations actually consumed
    uint32_t act_stride = 0;  // bytes between consecutive input vectors of this layer
    void *packed = nullptr;
    int8_t *rows_lo = nullptr, *rows_hi = nullptr;
    uint32_t row_stride = 0;
    bool has_hi = false;      // the layer holds an FP1.3.0 +128 (second weight plane in use)
};

}  // namespace bnm_internal

using namespace bnm_internal;      // (an internal header: every unit that includes it speaks this vocabulary)

struct bnm_ctx {
    int device = 0;
    bnm_model model;
    std::vector<FcDev> fc;
    // CNN front end
    uint32_t channels = 0;
    int8_t *w_conv[3] = {nullptr, nullptr, nullptr};
    int *cnn_wtab = nullptr;       // per-channel weight table of the conv1-on-MFMA front end
    void *cnn_li_frags = nullptr;  // lane = image front end (cnn_variant 3): per-channel Toeplitz fragments ...
    int *cnn_li_bias = nullptr;    // ... and plane-offset constants; nullptr when the kernel does not serve the channel count
    bool cnn_auto = true;          // nobody has called bnm_ctx_set_cnn_variant: small calls of a variant-3 model go to the channel kernel
    int cnn_variant = 1;           // 3: lane = image kernel (the default wherever it runs: up to 170 channels), 1: conv1 on the matrix cores / a lane per channel, 0: round 1's all-VALU kernel
    uint32_t cnn_grab = 8;         // images a wave of the MFMA front end takes from the work counter at a time (0: fixed shares)
    uint32_t cnn_li_grab = 1;      // 32-image tiles a wave of the lane = image front end takes at a time
    bool cnn_li_plane2 = true;     // the lane = image kernels carry conv3's third operand plane (the weights can reach pooled conv2 values >= 2^16)
    bool cnn_li_plane2_model = true;   // ... what the model's weights say (bnm_cnn_li_tables); cnn_li_plane2 differs only under bnm_ctx_set_cnn_variant(5)
    bool cnn_li_sums16 = false;    // every conv1 sum the weights allow fits 16 bits (bnm_cnn_li_tables): the pipelined one-kernel form serves the model
    bool cnn_li_pipe = false;      // ... and runs (the default where it serves; bnm_ctx_set_cnn_variant(6): the four-waves-per-SIMD form, A/B)
    bool cnn_fused_ok = false;     // the lane = image kernel with the FC tail in the same wave serves this model (bnm_cnn_li_fused.hip)
    bool cnn_fuse_tail = true;     // ... and runs whenever the lane = image front end would (bnm_ctx_set_cnn_variant 3 / 300+g; 4 = unfused)
    // Work counters of the persistent kernels that hand their work out dynamically (dual-tile kernel, generic fused kernel, CNN
    // front end, streamed ternary kernel): one counter BLOCK (BNM_WORK_BLOCK_WORDS words, bnm_kernels.h) per STREAM the context
    // is used on.  Launches on one stream are ordered, and every kernel leaves its block all-zero (the last wave to leave puts it
    // back), so one block serves all of a stream's launches without a memset in between; launches on different streams never
    // share one.  A launch that is being CAPTURED into a HIP graph gets a block of its own that no eager launch will ever use
    // (the graph may be replayed on any stream, next to eager launches on the capturing one).
    std::vector<uint32_t *> work_free;               // blocks not handed out yet (zeroed)
    std::map<hipStream_t, uint32_t *> work_of;       // stream -> its block
    unsigned long long *nonfinite = nullptr;   // float calls: images that held a NaN or an infinity so far (bnm_ctx_float_nonfinite)
    uint32_t *idle_words = nullptr;   // fused variant 6: one word per resident wave for the loop's zero-adds (never changes value)
    bool tern_dynamic = true;
    uint32_t work_batch = 0;      // tiles / pairs a wave of the fused kernels takes from the work counter at a time (0 = kernel default)
    // fused MFMA path: shape-specialised kernels (register-resident weights, bnm_fused_fc.hip) and / or the generic
    // kernel (run-time widths, weights in LDS, bnm_fused_generic.hip; variant id BNM_FUSED_GENERIC)
    bool fused_ok = false;      // at least one of the two can run this model
    bool table_ok = false, generic_ok = false, regw_ok = false;
    BnmFusedShape shape{};
    BnmGenericDesc gdesc{};
    void *frags = nullptr, *gfrags = nullptr;
    uint32_t in_width = 256;    // bytes of one input row of the FC stack (256, or 4*C behind the CNN front end)
    int variant = -1, grid_blocks = 0;
    // float inputs (bnm_infer_float_device): the fused float-input kernel serves this model (FC, 256-value rows, 2- / 4-tile class)
    bool f32_ok = false;
    int float_mode = 0;         // 0: the fused kernel where it exists, else quantise + infer; 1: fused or BNM_EUNSUPPORTED; 2: always two kernels
    int f32_groups = 0;         // 8-image groups in flight per wave of the fused float kernel (0 = default)
    // ternary ALU path
    bool tern_ok = false;
    int *tern_stream = nullptr;   // the trits in the streamed kernel's consumption order (bnmk_ternary_stream_build)
    int tern_variant = 2;         // 2: streamed weights, two images per lane (default where it exists); 1: one image per lane; 0: round 1's kernel
    bool tern_two = false;        // the two-images-per-lane kernel exists for this model's widths
    int requested_path = BNM_PATH_AUTO, path = BNM_PATH_LAYERWISE_ALU;
    bool warned_layerwise = false;
    bool all_known = false;       // every FC layer's codec is one the C engine decodes (=> int8 rows, the MFMA layer-wise path)
    std::string fused_reason = "unknown";   // why fused_ok is false
#ifdef BNM_DIAG
    uint64_t diag_src_wrap = 0;
#endif
    // scratch
    // scratch of the CNN and layer-wise paths, one set per stream the context has been used on (launches on different
    // streams must not share feature rows / activation buffers)
    struct StreamScratch {
        DevBuf act_a, act_b, out32, cnn_feat;
        DevBuf q8;      // bnm_infer_float_device: the quantised images of one chunk
    };
    std::map<hipStream_t, StreamScratch> scratch;
    DevBuf argmax, stage_img, stage_cls, stage_logits;
    // host-pointer paths: zero-copy buffers of the latency path (n <= kLatencyMax) and the two slots of the pipelined path
    PinBuf lat_in, lat_cls, lat_logits;
    hipStream_t lat_stream = nullptr;
    bool lat_spin = true;            // poll the page-locked result words instead of waiting for the stream (bnm_ctx_set_host_tuning)
    // the resident single-wave kernel of the one-image path (opt-in: BNM_PERSISTENT=1 or bnm_ctx_set_persistent; bnm_persist_kernel.hpp)
    int persist_mode = -1;           // -1: not decided yet (the environment is read at the first one-image call), 0 off, 1 on
    uint32_t persist_idle_us = 5000; // the kernel leaves by itself after this long without a call
    PinBuf persist_box;              // its mailbox
    hipStream_t persist_stream = nullptr;
    bool persist_running = false;    // a kernel has been started and was not seen finished yet
    uint32_t persist_seq = 0;        // sequence number of the last call
    unsigned host_threads = 0;       // staging-copy threads of the pipelined path (0 = default)
    int host_mode = 0;               // 0 pipelined page-locked staging, 1 the HIP runtime's own pageable copies (synchronous)
    struct HostSlot {
        PinBuf in, cls, logits;
        DevBuf d_in, d_cls, d_logits;
        hipStream_t stream = nullptr;
        hipEvent_t computed = nullptr;
        uint64_t off = 0, count = 0;      // the chunk in flight on this slot (count == 0: idle)
    } slot[2];
    ParallelCopier *copier = nullptr;
    std::vector<void *> owned;
    // what the context's LAST inference call launched (bnm_ctx_last_kernel): kernel names joined by '+', in launch order
    std::string last_kernel;
    std::mutex mu;
};

namespace bnm_internal {

// The entry points work on the context's device and leave the calling thread's current device as they found it (a host that
// drives several GPUs from one thread - or PyTorch with another current device - must not find it changed behind its back).
struct DeviceGuard {
    int prev = -1;
    bool changed = false;
    hipError_t err = hipSuccess;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) { (void)hipGetLastError(); prev = -1; }
        if (prev != dev) {
            err = hipSetDevice(dev);
            changed = err == hipSuccess;
        }
    }
    ~DeviceGuard() {
        if (changed && prev >= 0) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};

// ---- bnm_capi_ctx.cpp ------------------------------------------------------------------------------------------------
int resolve_path(bnm_ctx *c);
// the counter block of a launch on stream s / the scratch the context keeps for stream s (see bnm_ctx)
int work_block(bnm_ctx *c, hipStream_t s_real, uint32_t **out);
bnm_ctx::StreamScratch &stream_scratch(bnm_ctx *c, hipStream_t s_real);
hipStream_t stream_key(hipStream_t s);
// a call of n images on this CNN context runs the one-kernel form (lane = image front end + FC tail in the same wave): the model
// fits it, nobody chose another front end, and - for a context left to itself - the call is not a small one (fewer than 2 C^2
// images go to the channel kernel)
inline bool cnn_one_kernel_call(const bnm_ctx *c, uint64_t n) {
    const bool small_call = c->cnn_auto && n < 2ull * c->channels * c->channels;
    return c->model.kind == BNM_KIND_CNN && c->path == BNM_PATH_FUSED_MFMA && c->cnn_fused_ok && c->cnn_fuse_tail && c->cnn_variant == 3 &&
           !small_call && n < (1ull << 31);
}
// ---- bnm_capi_infer.cpp ----------------------------------------------------------------------------------------------
// whole-model launch on device data (c->mu held, the context's device current); d_acts_tap: the parity tap (layer-wise path)
int infer_device_locked(bnm_ctx *c, const int8_t *d_images, uint64_t n, uint32_t *d_cls, int32_t *d_logits,
                        int8_t *d_acts_tap, uint32_t tap_stride, hipStream_t s);

}  // namespace bnm_internal
