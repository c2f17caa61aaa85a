// C ABI of libbitnetmcu_hip.so / Bitnet_inf.dll (declared in include/bitnetmcu_hip.h).
// Host side only: model handling, device residency, kernel selection, the reference's own symbols on top
// of the device kernels.  There is NO CPU compute path in this file: if HIP is unusable, the reference-ABI
// functions abort() and the bnm_* functions return BNM_EHIP.
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
#include <dlfcn.h>
#include <rccl/rccl.h>
#include "bnm_device.hpp"
#include "bnm_kernels.h"
#include "bnm_model.hpp"
#ifdef BNM_DIAG
#include "bnm_diag.h"
#endif

namespace {

thread_local std::string g_err;

int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess)                                                                           \
            return fail(BNM_EHIP, std::string(#expr) + ": " + hipGetErrorString(_e));                   \
    } while (0)

uint32_t round_up(uint32_t v, uint32_t m) { return (v + m - 1u) / m * m; }

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

}  // namespace

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
    // Work counters of the persistent kernels that hand their work out dynamically (dual-tile kernel, generic fused kernel, CNN
    // front end, streamed ternary kernel): one counter BLOCK (BNM_WORK_BLOCK_WORDS words, bnm_kernels.h) per STREAM the context
    // is used on.  Launches on one stream are ordered, and every kernel leaves its block all-zero (the last wave to leave puts it
    // back), so one block serves all of a stream's launches without a memset in between; launches on different streams never
    // share one.  A launch that is being CAPTURED into a HIP graph gets a block of its own that no eager launch will ever use
    // (the graph may be replayed on any stream, next to eager launches on the capturing one).
    std::vector<uint32_t *> work_free;               // blocks not handed out yet (zeroed)
    std::map<hipStream_t, uint32_t *> work_of;       // stream -> its block
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
    std::mutex mu;
};

namespace {

constexpr uint64_t kChunk = 1ull << 20;   // images per internal chunk of the staged / layer-wise paths
constexpr uint64_t kCnnChunk = 1ull << 22;   // images per launch of the CNN front end when the fused FC tail follows

int resolve_path(bnm_ctx *c) {
    int want = c->requested_path;
    bool all_tern = !c->fc.empty();
    for (auto &l : c->fc) all_tern = all_tern && l.info.bits_per_weight == 64;
    if (want == BNM_PATH_AUTO) {
        // the fastest bit-exact kernel: the fused MFMA kernels for every model they can run - all-ternary ones included (the
        // generic kernel does 1.6e10 inf/s on 256-96-96-96, the ALU kernel 3.0e9; BASELINE configs[2] asks for the ALU kernel
        // by name, and bench.py selects it explicitly with BNM_PATH_TERNARY_ALU)
        if (c->fused_ok) {
            want = BNM_PATH_FUSED_MFMA;
            // fragments that nearly fill the LDS leave room for very few waves beside them: the kernel still runs, far below its
            // usual rate (a lone wave per SIMD issues VALU at half rate, fewer leave SIMDs idle) - say so once
            const uint32_t waves = (!c->table_ok && c->generic_ok) ? bnmk_generic_resident_waves(c->gdesc, c->shape.dbl) : 8u;
            if (waves < 4u && !c->warned_layerwise && !std::getenv("BNM_QUIET")) {
                std::fprintf(stderr, "bitnetmcu_hip: this model's weight fragments (%u KiB) leave LDS for %u wave%s per compute unit of the fused "
                                     "kernel; it runs, well below the kernel's usual rate\n", c->gdesc.w_bytes >> 10, waves, waves == 1 ? "" : "s");
                c->warned_layerwise = true;
            }
        } else if (c->model.kind == BNM_KIND_FC && all_tern && c->tern_ok) want = BNM_PATH_TERNARY_ALU;
        else {
            // no silent cliffs: one kernel per layer with int32 sums through HBM - on the matrix cores when every codec decodes
            // to int8 rows (an order of magnitude below the fused kernels), else the bit-serial kernel (~500x below)
            want = c->all_known ? BNM_PATH_LAYERWISE_MFMA : BNM_PATH_LAYERWISE_ALU;
            if (!c->warned_layerwise && !std::getenv("BNM_QUIET")) {
                std::fprintf(stderr, "bitnetmcu_hip: model is outside the fused MFMA kernels (%s); using the layer-wise %s\n",
                             c->fused_reason.c_str(),
                             c->all_known ? "MFMA path (one GEMM kernel per layer, sums through HBM: about 10-30x slower than a fused kernel)"
                                          : "ALU path, which is about 500x slower");
                c->warned_layerwise = true;
            }
        }
    }
    if (want == BNM_PATH_LAYERWISE_MFMA && !c->all_known)
        return fail(BNM_EUNSUPPORTED, "the layer-wise MFMA path needs codecs the C engine decodes (int8 rows) in every layer");
    if (want == BNM_PATH_FUSED_MFMA && !c->fused_ok)
        return fail(BNM_EUNSUPPORTED, "model shape/codec is outside the fused MFMA kernel table");
    if (want == BNM_PATH_TERNARY_ALU && !(c->tern_ok && c->model.kind == BNM_KIND_FC))
        return fail(BNM_EUNSUPPORTED, "the ternary ALU kernels serve ternary FC models 256-H1-H2-H3-N with H1, H2 in {32, 64, 96, 128}, H3 a "
                                      "multiple of 16 up to 128 and N <= 64");
    c->path = want;
    return BNM_OK;
}

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

int dev_alloc(bnm_ctx *c, void **p, size_t bytes) {
    HIP_TRY(hipMalloc(p, bytes ? bytes : 16));
    c->owned.push_back(*p);
    return BNM_OK;
}

constexpr size_t kWorkBlocksPerChunk = 256;      // 256 KiB of counter blocks per allocation
constexpr size_t kMaxStreams = 32;               // streams a context keeps scratch / counter blocks for before it evicts

int work_blocks_grow(bnm_ctx *c) {
    void *q = nullptr;
    if (int e = dev_alloc(c, &q, kWorkBlocksPerChunk * BNM_WORK_BLOCK_WORDS * 4)) return e;
    HIP_TRY(hipMemset(q, 0, kWorkBlocksPerChunk * BNM_WORK_BLOCK_WORDS * 4));
    for (size_t i = kWorkBlocksPerChunk; i-- > 0;) c->work_free.push_back((uint32_t *)q + i * BNM_WORK_BLOCK_WORDS);
    return BNM_OK;
}

// Key of a stream in the per-stream tables.  Launches that share a key share a counter block and scratch buffers and must be
// ordered among themselves - true for a real stream handle, NOT for hipStreamPerThread: that is one constant handle value which
// names a different stream in every host thread, so its key is the address of a thread-local object (one entry per calling thread).
// tokens made by stream_key() are addresses of thread-local bytes, real handles come from the runtime: the context remembers
// which keys are tokens
std::mutex g_token_mu;
std::vector<const void *> g_tokens;
bool c_is_stream_handle(hipStream_t key) {
    std::lock_guard<std::mutex> g(g_token_mu);
    for (const void *t : g_tokens)
        if (t == (const void *)key) return false;
    return true;
}
hipStream_t stream_key(hipStream_t s) {
    static thread_local char per_thread_key;
    static thread_local bool registered = false;
    if (s != hipStreamPerThread) return s;
    if (!registered) {
        std::lock_guard<std::mutex> g(g_token_mu);
        g_tokens.push_back(&per_thread_key);
        registered = true;
    }
    return (hipStream_t)(void *)&per_thread_key;
}

bool stream_is_capturing(hipStream_t s) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess) {
        (void)hipGetLastError();     // e.g. the legacy stream queried while another stream captures: not capturing itself
        return false;
    }
    return st == hipStreamCaptureStatusActive;
}

// Everything the context keeps for streams other than `keep` goes: their scratch buffers are freed, their counter blocks
// return to the free list.  Device-synchronising; called when the per-stream tables have grown to kMaxStreams entries (a host
// that cycles through short-lived streams would otherwise grow them without bound) - never while `keep` is capturing.
void evict_other_streams(bnm_ctx *c, hipStream_t keep) {
    // a device-wide synchronisation would invalidate a stream capture in progress: not while any stream the context knows captures
    // (keys are stream handles or, for hipStreamPerThread, per-thread tokens - only the former can be asked)
    auto capturing = [](hipStream_t key) { return c_is_stream_handle(key) && stream_is_capturing(key); };
    for (auto &kv : c->scratch)
        if (capturing(kv.first)) return;
    for (auto &kv : c->work_of)
        if (capturing(kv.first)) return;
    if (hipDeviceSynchronize() != hipSuccess) { (void)hipGetLastError(); return; }
    for (auto it = c->scratch.begin(); it != c->scratch.end();) {
        bool frozen = false;
        for (DevBuf *b : {&it->second.act_a, &it->second.act_b, &it->second.out32, &it->second.cnn_feat, &it->second.q8}) frozen = frozen || b->frozen;
        if (it->first == keep || frozen) { ++it; continue; }      // (a captured graph may still replay on a frozen entry's buffers)
        for (DevBuf *b : {&it->second.act_a, &it->second.act_b, &it->second.out32, &it->second.cnn_feat, &it->second.q8}) b->release();
        it = c->scratch.erase(it);
    }
    for (auto it = c->work_of.begin(); it != c->work_of.end();) {
        if (it->first == keep) { ++it; continue; }
        // a block is all zero when its last launch left normally; one that did not (a launch that failed half-way) must not
        // poison the block's next owner: the device is idle here, so zero it on the way back to the free list
        (void)hipMemset(it->second, 0, sizeof(uint32_t) * BNM_WORK_BLOCK_WORDS);
        c->work_free.push_back(it->second);
        it = c->work_of.erase(it);
    }
}

// the counter block of a launch on stream s (see bnm_ctx)
int work_block(bnm_ctx *c, hipStream_t s_real, uint32_t **out) {
    const bool capturing = stream_is_capturing(s_real);
    const hipStream_t s = stream_key(s_real);
    if (!capturing) {
        auto it = c->work_of.find(s);
        if (it != c->work_of.end()) { *out = it->second; return BNM_OK; }
        if (c->work_of.size() >= kMaxStreams) evict_other_streams(c, s);
    }
    if (c->work_free.empty()) {
        if (capturing)
            return fail(BNM_EUNSUPPORTED, "no counter block left for a captured launch (256 per context): hipMalloc is not "
                                          "allowed during stream capture - run one eager call first or capture fewer launches");
        if (int e = work_blocks_grow(c)) return e;
    }
    uint32_t *b = c->work_free.back();
    c->work_free.pop_back();
    if (!capturing) c->work_of[s] = b;      // a captured launch's block belongs to the graph for the context's lifetime
    *out = b;
    return BNM_OK;
}

bnm_ctx::StreamScratch &stream_scratch(bnm_ctx *c, hipStream_t s_real) {
    const hipStream_t s = stream_key(s_real);
    const bool capturing = stream_is_capturing(s_real);
    auto it = c->scratch.find(s);
    if (it == c->scratch.end()) {
        if (c->scratch.size() >= kMaxStreams && !capturing) evict_other_streams(c, s);
        it = c->scratch.emplace(s, bnm_ctx::StreamScratch{}).first;
    }
    if (capturing)      // what a captured launch touches stays where it is (DevBuf::frozen)
        for (DevBuf *b : {&it->second.act_a, &it->second.act_b, &it->second.out32, &it->second.cnn_feat, &it->second.q8}) b->frozen = true;
    return it->second;
}

int ctx_build(bnm_ctx *c) {
    const bnm_model &m = c->model;
    hipStream_t s = nullptr;
    uint32_t width = 256;
    size_t li = 0;
    if (m.kind == BNM_KIND_CNN) {
        c->channels = m.layers[0].info.out_channels;
        const int conv_idx[3] = {0, 1, 3};
        for (int k = 0; k < 3; k++) {
            const BnmLayer &L = m.layers[conv_idx[k]];
            void *p = nullptr;
            if (int e = dev_alloc(c, &p, L.weights.size())) return e;
            HIP_TRY(hipMemcpy(p, L.weights.data(), L.weights.size(), hipMemcpyHostToDevice));
            c->w_conv[k] = (int8_t *)p;
        }
        {
            const uint32_t C = c->channels, C_pad = (C + 63u) / 64u * 64u;
            std::vector<int> tab((size_t)2 * C_pad * BNM_CNN_WTAB_DWORDS);
            bnm_cnn_weight_table((const int8_t *)m.layers[0].weights.data(), (const int8_t *)m.layers[1].weights.data(),
                                 (const int8_t *)m.layers[3].weights.data(), C, tab.data());
            void *p = nullptr;
            if (int e = dev_alloc(c, &p, tab.size() * sizeof(int))) return e;
            HIP_TRY(hipMemcpy(p, tab.data(), tab.size() * sizeof(int), hipMemcpyHostToDevice));
            c->cnn_wtab = (int *)p;
        }
        if (bnmk_cnn_li_waves(c->channels)) {
            const uint32_t C = c->channels;
            std::vector<int8_t> fr((size_t)C * 6 * 1024);
            std::vector<int> bi((size_t)C * 2);
            bnm_cnn_li_tables((const int8_t *)m.layers[0].weights.data(), (const int8_t *)m.layers[1].weights.data(),
                              (const int8_t *)m.layers[3].weights.data(), C, fr.data(), bi.data());
            void *p = nullptr, *q = nullptr;
            if (int e = dev_alloc(c, &p, fr.size())) return e;
            if (int e = dev_alloc(c, &q, bi.size() * sizeof(int))) return e;
            HIP_TRY(hipMemcpy(p, fr.data(), fr.size(), hipMemcpyHostToDevice));
            HIP_TRY(hipMemcpy(q, bi.data(), bi.size() * sizeof(int), hipMemcpyHostToDevice));
            c->cnn_li_frags = p;
            c->cnn_li_bias = (int *)q;
            // the default front end: the lane = image kernel wherever it runs (six or more waves per CU beside its records: <= 170
            // channels; 1.10 to 2.0 x the channel kernel at 8 .. 140 channels, profiles/r04/cnn_channels_r05a.log)
            c->cnn_variant = 3;
        }
        width = c->channels * 4u;
        li = 5;
    }
    {
        void *q = nullptr;
        if (int e = dev_alloc(c, &q, (size_t)16 * 4 * BNM_WORK_DUMMY_WAVES)) return e;
        HIP_TRY(hipMemset(q, 0, (size_t)16 * 4 * BNM_WORK_DUMMY_WAVES));
        c->idle_words = (uint32_t *)q;
        if (int e = work_blocks_grow(c)) return e;
    }
    const uint32_t in_width = width;
    bool all_known = true, any_fp130 = false, all_tern = true;
    for (; li < m.layers.size(); li++) {
        const BnmLayer &L = m.layers[li];
        FcDev d;
        d.info = L.info;
        d.n_real = bnm_fc_real_inputs(L.info, width);
        d.act_stride = width;
        if (int e = dev_alloc(c, &d.packed, L.weights.size())) return e;
        HIP_TRY(hipMemcpy(d.packed, L.weights.data(), L.weights.size(), hipMemcpyHostToDevice));
        d.row_stride = round_up(d.n_real, 32);
        const size_t rb = (size_t)round_up(L.info.n_output, 32) * d.row_stride;
        void *lo = nullptr, *hi = nullptr;
        if (int e = dev_alloc(c, &lo, rb)) return e;
        if (int e = dev_alloc(c, &hi, rb)) return e;
        HIP_TRY(hipMemset(lo, 0, rb));
        HIP_TRY(hipMemset(hi, 0, rb));
        d.rows_lo = (int8_t *)lo;
        d.rows_hi = (int8_t *)hi;
        // GPU unpack: packed words -> int8 rows
        HIP_TRY(bnmk_unpack_rows(d.packed, L.info.bits_per_weight, L.info.n_input, d.n_real, L.info.n_output, d.rows_lo,
                                 d.rows_hi, d.row_stride, s));
        all_known = all_known && bnm_codec_known(L.info.bits_per_weight);
        if (L.info.bits_per_weight == 20) {
            // FP1.3.0: only the code "sign 0, exponent 7" (+128) does not fit int8 and needs the second weight plane;
            // -128 fits.  Trained models rarely contain it (mcu/BitNetMCU_model_12k_FP130.h has none), so the
            // two-pass kernel is selected only when the packed words actually hold such a nibble.
            const uint32_t *w = (const uint32_t *)L.weights.data();
            for (size_t k = 0; k < L.weights.size() / 4 && !d.has_hi; k++)
                for (int nib = 0; nib < 8; nib++)
                    if (((w[k] >> (4 * nib)) & 15u) == 7u) { d.has_hi = true; break; }
            any_fp130 = any_fp130 || d.has_hi;
        }
        all_tern = all_tern && L.info.bits_per_weight == 64;
        width = L.info.n_output;
        c->fc.push_back(d);
    }

    c->all_known = all_known;
    // ---- fused MFMA path: shape + fragment buffers ------------------------------------------------------
    const size_t nfc = c->fc.size();
    c->in_width = in_width;
    uint32_t max_width = 0;
    for (auto &l : c->fc) max_width = l.info.n_output > max_width ? l.info.n_output : max_width;
    if (!all_known) c->fused_reason = "a layer uses a codec the C engine does not decode";
    else if (nfc != 3 && nfc != 4) c->fused_reason = "the reference wrapper's FC stack has 3 or 4 layers";
    else if (max_width > 256) c->fused_reason = "a layer is wider than 256 outputs";
    else if (in_width > 512) c->fused_reason = "input rows longer than 512 bytes";
    else {
        BnmFusedShape sh{};
        for (size_t i = 0; i < 4; i++) sh.M[i] = i < nfc ? (int)((c->fc[i].info.n_output + 31u) / 32u) : 0;
        sh.split = any_fp130;
        sh.nc8 = (int)((c->fc[nfc - 1].info.n_output + 7u) / 8u);
        // doubling needs |2w| <= 127 in every hidden layer: all codecs but 8-bit two's complement and FP1.3.0
        sh.dbl = true;
        for (size_t i = 0; i + 1 < nfc; i++)
            if (c->fc[i].info.bits_per_weight == 16 || c->fc[i].info.bits_per_weight == 20) sh.dbl = false;
        const int sp = sh.split ? 2 : 1;
        // fragment image for input rows of kt0 K-steps: per layer, per 32-row tile m: [KT lo fragments][KT hi fragments]
        // fragment image: per layer, per 32-row tile m: [ktp lo fragments][ktp hi fragments]; mt[i] tiles (>= the real
        // count: surplus tiles and K-steps hold zero weights), ktp[i] K-steps; layer i starts at layer_off[i]
        // kmajor: the generic kernel's layout - per layer [plane][K-step][tile] (fragment (p, s, m) at ((p * kt + s) * mt + m) KiB)
        auto build_frags = [&](const uint32_t *mt, const uint32_t *ktp, const uint32_t *layer_off, uint32_t total, bool kmajor, void **out) -> int {
            if (int e = dev_alloc(c, out, total)) return e;
            HIP_TRY(hipMemsetAsync(*out, 0, total, s));
            for (size_t i = 0; i < nfc; i++) {
                const FcDev &d = c->fc[i];
                char *dst = (char *)*out + layer_off[i];
                const uint32_t kt = ktp[i];
                const uint32_t real_tiles = (d.info.n_output + 31u) / 32u;
                for (uint32_t m = 0; m < mt[i]; m++) {
                    for (int part = 0; part < sp; part++) {
                        const bool past = m >= real_tiles;      // surplus tile: no rows to read
                        const int8_t *rows = (part == 0 ? d.rows_lo : d.rows_hi) + (past ? 0 : (size_t)m * 32u * d.row_stride);
                        const uint32_t rows_left = past ? 0u : d.info.n_output - m * 32u;
                        const int scale = (sh.dbl && i + 1 < nfc) ? 2 : 1;   // hidden layers only
                        // classifier layer: padding rows weigh -128 so they can never win the argmax (first plane only)
                        const int pad = (i + 1 == nfc && part == 0) ? -128 : 0;
                        if (kmajor)
                            HIP_TRY(bnmk_build_fragments(rows, d.row_stride, rows_left, d.n_real, 1, kt, i == 0 ? 0 : 1, scale, pad,
                                                         dst + ((size_t)part * kt * mt[i] + m) * 1024, mt[i] * 1024u, s));
                        else
                            HIP_TRY(bnmk_build_fragments(rows, d.row_stride, rows_left, d.n_real, 1, kt, i == 0 ? 0 : 1, scale, pad,
                                                         dst + ((size_t)m * kt * sp + (size_t)part * kt) * 1024, 1024u, s));
                    }
                }
            }
            return BNM_OK;
        };
        // (1) shape-specialised kernels: the reference zoo's shapes
        if (in_width % 32u == 0) {
            sh.KT0 = (int)(in_width / 32u);
            int var = bnmk_fused_default_variant(sh);
            if (bnmk_fused_supported(sh, var)) {
                uint32_t mt[4], ktp[4], off[4], bytes = 0, kt = (uint32_t)sh.KT0;
                for (size_t i = 0; i < nfc; i++) {
                    mt[i] = (uint32_t)sh.M[i]; ktp[i] = kt; off[i] = bytes;
                    bytes += mt[i] * kt * (uint32_t)sp * 1024u;
                    kt = mt[i];
                }
                if (int e = build_frags(mt, ktp, off, bytes, false, &c->frags)) return e;
                c->table_ok = true;
                c->variant = var;
            } else if (bnmk_regw_supported(sh)) {
                // shapes of the register-resident-weight kernel (variant 9, selected with bnm_ctx_set_tuning only - DESIGN.md 4.1c
                // says why it is not the default): the same fragment layout
                uint32_t mt[4], ktp[4], off[4], bytes = 0, kt = (uint32_t)sh.KT0;
                for (size_t i = 0; i < nfc; i++) {
                    mt[i] = (uint32_t)sh.M[i]; ktp[i] = kt; off[i] = bytes;
                    bytes += mt[i] * kt * 1024u;
                    kt = mt[i];
                }
                if (int e = build_frags(mt, ktp, off, bytes, false, &c->frags)) return e;
                c->regw_ok = true;
            }
        }
        c->shape = sh;
        // (2) generic kernel: any widths; input rows padded to 64 / 128 / 256 / 512 bytes (the CNN front end writes
        // its act rows with that stride; the fragment builder gives the padding columns weight 0)
        BnmGenericDesc gd{};
        uint32_t row = 64;
        while (row < in_width) row *= 2;
        gd.KT0 = row / 32u;
        gd.sp = (uint32_t)sp;
        gd.n_classes = c->fc[nfc - 1].info.n_output;
        uint32_t m_real[4];
        for (size_t i = 0; i < 4; i++) m_real[i] = (uint32_t)sh.M[i];
        if (bnmk_generic_plan(gd, m_real) && bnmk_generic_supported(gd, sh.dbl)) {
            if (int e = build_frags(gd.M, gd.KTP, gd.frag_off, gd.w_bytes, true, &c->gfrags)) return e;
            c->gdesc = gd;
            c->generic_ok = true;
            if (!c->table_ok) c->variant = BNM_FUSED_GENERIC;
        } else if (!c->table_ok) {
            c->fused_reason = "the weight fragments do not fit beside the image tiles in 160 KiB of LDS";
        }
        // the register-resident-weight kernel takes whole 64-image pairs; the generic kernel finishes its calls
        c->regw_ok = c->regw_ok && c->generic_ok;
        c->fused_ok = c->table_ok || c->generic_ok;
    }
    // ---- ternary ALU path ------------------------------------------------------------------------------
    if (m.kind == BNM_KIND_FC && all_tern && nfc == 4) {
        BnmTernArgs a{};
        for (int i = 0; i < 4; i++) {
            a.rows[i] = c->fc[i].rows_lo;
            a.stride[i] = c->fc[i].row_stride;
            a.n_in[i] = c->fc[i].n_real;
            a.n_out[i] = c->fc[i].info.n_output;
        }
        if (bnmk_ternary_alu_supported(a.n_in, a.n_out) && a.n_out[3] <= 64) {
            void *p = nullptr;
            if (int e = dev_alloc(c, &p, (size_t)bnmk_ternary_stream_dwords(a.n_out) * 4u)) return e;
            c->tern_stream = (int *)p;
            HIP_TRY(bnmk_ternary_stream_build(a, c->tern_stream, s));
            // two images per lane where that kernel exists (96-96-96), one per lane for the other shapes of the table
            c->tern_two = bnmk_ternary_stream_supported(a.n_out, 2);
            c->tern_variant = c->tern_two ? 2 : 1;
            c->tern_ok = true;
        }
    }
    HIP_TRY(hipDeviceSynchronize());
    return resolve_path(c);
}

// ---- whole-model launches on device data -----------------------------------------------------------
bool is_generic(int variant) { return variant == BNM_FUSED_GENERIC || variant == BNM_FUSED_GENERIC_T1 || variant == BNM_FUSED_GENERIC_T2; }

int run_fused(bnm_ctx *c, const int8_t *d_in, uint64_t n, uint32_t *d_cls, int32_t *d_logits, hipStream_t s) {
    uint32_t *block = nullptr;
    if (int e = work_block(c, s, &block)) return e;
    if (is_generic(c->variant)) {
        const int tiles = c->variant == BNM_FUSED_GENERIC_T1 ? 1 : c->variant == BNM_FUSED_GENERIC_T2 ? 2 : 0;
        HIP_TRY(bnmk_fused_generic(c->gdesc, c->shape.dbl, tiles, c->grid_blocks, d_in, n, c->gfrags, d_cls, d_logits, block,
                                   c->work_batch, s));
        return BNM_OK;
    }
    if (c->variant == BNM_FUSED_REGW) {
        // whole 64-image pairs to the register-resident-weight kernel, the last < 64 images - or all of a call too small to give
        // every resident wave a pair - to the generic kernel (same stream, same counter block: the launches are ordered and each
        // leaves the block all-zero)
        const uint64_t resident_waves = (c->grid_blocks > 0 ? (uint64_t)c->grid_blocks : (uint64_t)bnm_num_cus()) * 4ull;
        const uint64_t n_main = (n >> 6) < resident_waves ? 0ull : n & ~63ull;
        if (n_main) {
            BnmFusedArgs a{};
            a.images = d_in;
            a.n = n_main;
            a.frags = c->frags;
            a.n_classes = c->model.num_classes();
            a.cls = d_cls;
            a.logits = d_logits;
            a.work = block;
            a.idle = c->idle_words;
            a.batch = c->work_batch;
            HIP_TRY(bnmk_fused_fc(c->shape, c->variant, c->grid_blocks, a, s));
        }
        if (n > n_main)
            HIP_TRY(bnmk_fused_generic(c->gdesc, c->shape.dbl, 0, c->grid_blocks, d_in + n_main * (uint64_t)c->in_width, n - n_main, c->gfrags,
                                       d_cls + n_main, d_logits ? d_logits + n_main * c->model.num_classes() : nullptr, block, 0, s));
        return BNM_OK;
    }
    BnmFusedArgs a{};
    a.images = d_in;
    a.n = n;
    a.frags = c->frags;
    a.n_classes = c->model.num_classes();
    a.cls = d_cls;
    a.logits = d_logits;
#ifdef BNM_DIAG
    a.src_wrap = c->diag_src_wrap;   // diagnostic library only (bnm_diag_set_src_wrap)
#endif
    a.work = block;
    a.idle = c->idle_words;
    a.batch = c->work_batch;
    HIP_TRY(bnmk_fused_fc(c->shape, c->variant, c->grid_blocks, a, s));
    return BNM_OK;
}

// FC chain layer by layer on [n][in_stride] int8 inputs; n <= kChunk.  mfma: the layers as int8 GEMMs on the matrix cores
// (bnmk_fc_layer_mfma; activation rows padded to 32-byte K-steps) instead of the bit-serial kernel.
// in_stride: bytes between consecutive input rows (256 for FC models; the CNN front end's act-row stride)
int run_layerwise(bnm_ctx *c, const int8_t *d_in, uint32_t in_stride, uint64_t n, uint32_t *d_cls, int32_t *d_logits, int8_t *d_acts_tap,
                  uint32_t tap_stride, uint32_t tap_off, bool mfma, hipStream_t s) {
    uint32_t maxw = 0;
    for (auto &l : c->fc) maxw = l.info.n_output > maxw ? l.info.n_output : maxw;
    const uint32_t maxs = mfma ? round_up(maxw, 32) : maxw;      // stride of the scratch activation rows
    bnm_ctx::StreamScratch &sc = stream_scratch(c, s);
    if (int e = sc.act_a.ensure((size_t)n * maxs + 64)) return e;
    if (int e = sc.act_b.ensure((size_t)n * maxs + 64)) return e;
    if (int e = sc.out32.ensure((size_t)n * maxw * 4)) return e;
    const int8_t *act = d_in;
    uint32_t act_stride = in_stride;
    int8_t *bufs[2] = {(int8_t *)sc.act_a.p, (int8_t *)sc.act_b.p};
    for (size_t i = 0; i < c->fc.size(); i++) {
        const FcDev &d = c->fc[i];
        const bool last = i + 1 == c->fc.size();
        int32_t *out = (last && d_logits) ? d_logits : (int32_t *)sc.out32.p;
        if (mfma)
            HIP_TRY(bnmk_fc_layer_mfma(act, act_stride, d.rows_lo, d.has_hi ? d.rows_hi : nullptr, d.row_stride, d.info.n_output, out, n, s));
        else
            HIP_TRY(bnmk_fc_layer(act, act_stride, d.packed, d.info.bits_per_weight, d.info.n_input, d.info.n_output, out, n, s));
        int8_t *nxt = bufs[i & 1];
        const uint32_t nxt_stride = mfma ? round_up(d.info.n_output, 32) : d.info.n_output;
        HIP_TRY(bnmk_relunorm(out, d.info.n_output, nxt, nxt_stride, last ? d_cls : nullptr, n, s));
        if (d_acts_tap) {
            HIP_TRY(hipMemcpy2DAsync(d_acts_tap + tap_off, tap_stride, nxt, nxt_stride, d.info.n_output, n,
                                     hipMemcpyDeviceToDevice, s));
            tap_off += d.info.n_output;
        }
        act = nxt;
        act_stride = nxt_stride;
    }
    return BNM_OK;
}

int run_ternary(bnm_ctx *c, const int8_t *d_in, uint64_t n, uint32_t *d_cls, int32_t *d_logits, hipStream_t s) {
    BnmTernArgs a{};
    a.images = d_in;
    a.n = n;
    a.n_layers = 4;
    for (int i = 0; i < 4; i++) {
        a.rows[i] = c->fc[i].rows_lo;
        a.stride[i] = c->fc[i].row_stride;
        a.n_in[i] = c->fc[i].n_real;
        a.n_out[i] = c->fc[i].info.n_output;
    }
    a.cls = d_cls;
    a.logits = d_logits;
    a.wstream = c->tern_stream;
    a.variant = c->tern_variant;
    a.counter = nullptr;
    if (c->tern_dynamic)
        if (int e = work_block(c, s, &a.counter)) return e;
    HIP_TRY(bnmk_ternary_alu(a, c->grid_blocks, s));
    return BNM_OK;
}

int infer_device_locked(bnm_ctx *c, const int8_t *d_images, uint64_t n, uint32_t *d_cls, int32_t *d_logits,
                        int8_t *d_acts_tap, uint32_t tap_stride, hipStream_t s) {
    if (!n) return BNM_OK;
    if (!d_images || !d_cls) return fail(BNM_EINVAL, "null device pointer");
    if (((uintptr_t)d_images & 15u) != 0) return fail(BNM_EINVAL, "d_images must be 16-byte aligned");
    const uint32_t ncls = c->model.num_classes();
    const int path = d_acts_tap ? BNM_PATH_LAYERWISE_ALU : c->path;
    if (c->model.kind == BNM_KIND_FC) {
        if (path == BNM_PATH_FUSED_MFMA) return run_fused(c, d_images, n, d_cls, d_logits, s);
        if (path == BNM_PATH_TERNARY_ALU) return run_ternary(c, d_images, n, d_cls, d_logits, s);
        for (uint64_t off = 0; off < n; off += kChunk) {
            uint64_t cn = n - off < kChunk ? n - off : kChunk;
            if (int e = run_layerwise(c, d_images + off * 256, 256u, cn, d_cls + off, d_logits ? d_logits + off * ncls : nullptr,
                                      d_acts_tap ? d_acts_tap + off * tap_stride : nullptr, tap_stride, 0, path == BNM_PATH_LAYERWISE_MFMA, s))
                return e;
        }
        return BNM_OK;
    }
    // CNN: front end (conv/pool/ReLUNorm fused) -> int8 [n][4C] -> FC tail
    const uint32_t W = c->channels * 4u;
    // act rows: 4*C bytes, padded to the generic kernel's row length when that kernel runs the FC tail
    // (the layer-wise MFMA tail reads them in 32-byte K-steps with 16-byte loads: rows padded to a multiple of 32 - any channel
    // count then works, also one that is not a multiple of 4; the bytes between 4C and the stride meet zero weights)
    const uint32_t AS = (path == BNM_PATH_FUSED_MFMA && is_generic(c->variant)) ? c->gdesc.KT0 * 32u
                        : path == BNM_PATH_LAYERWISE_MFMA ? round_up(W, 32) : W;
    // chunks: 2^22 images when the fused tail consumes the act rows directly (1 GiB of act rows; every launch has a ramp and a
    // tail, so fewer, larger launches: +2 % over 2^20), 2^20 when the int32 features are needed as well (> 64 channels, taps)
    // or the layer-wise tail runs (its scratch is sized for kChunk)
    // (more than 64 channels on the MFMA front end: one fused launch, the feature buffer is two images of scratch)
    const bool feat_all = d_acts_tap != nullptr || (c->channels > 64 && !c->cnn_variant);
    const bool need_feat = c->channels > 64 || feat_all;
    const uint64_t chunk = (!feat_all && path == BNM_PATH_FUSED_MFMA) ? kCnnChunk : kChunk;
    for (uint64_t off = 0; off < n; off += chunk) {
        uint64_t cn = n - off < chunk ? n - off : chunk;
        // the FC tail reads act rows with 16-byte vector loads: keep the buffer padded
        const size_t feat_bytes = feat_all ? (size_t)cn * W * 4 : need_feat ? (size_t)2 * W * 4 : 0;
        DevBuf &cnn_feat = stream_scratch(c, s).cnn_feat;
        uint32_t *block = nullptr;
        if (int e = work_block(c, s, &block)) return e;
        if (int e = cnn_feat.ensure(feat_bytes + (size_t)cn * AS + 64)) return e;
        int32_t *feat = need_feat ? (int32_t *)cnn_feat.p : nullptr;
        int8_t *acts = (int8_t *)cnn_feat.p + feat_bytes;
        // A wave of the lane = image kernel walks ALL channels of its 32 images: a call's time has a floor of one such walk (2 us per
        // channel: 125 us at 64 channels, 36 us at 16), while the channel kernel spreads an image's channels over a wave (16 us for one
        // image).  Left to itself the context gives calls of fewer than 2 C^2 images - Inference(): one - to the channel kernel
        // (profiles/r04/cnn_small_n_r05c.log: the two cross at 500 / 3,000 / 17,000 images for 16 / 48 / 64 channels).
        const bool small_call = c->cnn_auto && n < 2ull * c->channels * c->channels;
        if (c->cnn_variant == 3 && c->cnn_li_frags && !small_call)
            HIP_TRY(bnmk_cnn_front_li(d_images + off * 256, cn, c->cnn_li_frags, c->cnn_li_bias, c->channels, acts, AS, block, c->cnn_li_grab, s));
        else
            HIP_TRY(bnmk_cnn_front(d_images + off * 256, cn, c->w_conv[0], c->w_conv[1], c->w_conv[2], c->cnn_variant ? c->cnn_wtab : nullptr,
                                   c->channels, 4, acts, AS, feat, d_acts_tap != nullptr, block, c->cnn_grab, s));
        uint32_t *cls = d_cls + off;
        int32_t *lg = d_logits ? d_logits + off * ncls : nullptr;
        if (d_acts_tap)
            HIP_TRY(hipMemcpy2DAsync(d_acts_tap + off * tap_stride, tap_stride, acts, AS, W, cn, hipMemcpyDeviceToDevice, s));
        if (path == BNM_PATH_FUSED_MFMA) {
            if (int e = run_fused(c, acts, cn, cls, lg, s)) return e;
        } else {
            if (int e = run_layerwise(c, acts, AS, cn, cls, lg, d_acts_tap ? d_acts_tap + off * tap_stride : nullptr, tap_stride,
                                      d_acts_tap ? W : 0, path == BNM_PATH_LAYERWISE_MFMA, s))
                return e;
        }
    }
    return BNM_OK;
}

}  // namespace

// =================================================================================================
// (B) additive ABI
// =================================================================================================
extern "C" {

const char *bnm_last_error(void) { return g_err.c_str(); }
const char *bnm_version(void) { return "bitnetmcu_hip 0.1 (gfx950)"; }

int bnm_model_from_header_text(const char *text, size_t len, bnm_model **out) {
    if (!text || !out) return fail(BNM_EINVAL, "null argument");
    bnm_model *m = new bnm_model();
    std::string err;
    if (!bnm_parse_header_text(text, len, *m, err)) {
        delete m;
        return fail(err.find("support") != std::string::npos ? BNM_EUNSUPPORTED : BNM_EPARSE, err);
    }
    *out = m;
    return BNM_OK;
}

int bnm_model_from_blob(const void *blob, size_t len, bnm_model **out) {
    if (!blob || !out) return fail(BNM_EINVAL, "null argument");
    bnm_model *m = new bnm_model();
    std::string err;
    if (!bnm_deserialize(blob, len, *m, err)) {
        delete m;
        return fail(BNM_EPARSE, err);
    }
    *out = m;
    return BNM_OK;
}

size_t bnm_model_blob_size(const bnm_model *m) { return m ? bnm_serialize(*m).size() : 0; }

int bnm_model_to_blob(const bnm_model *m, void *dst, size_t cap) {
    if (!m || !dst) return fail(BNM_EINVAL, "null argument");
    std::vector<uint8_t> b = bnm_serialize(*m);
    if (cap < b.size()) return fail(BNM_EINVAL, "destination too small");
    std::memcpy(dst, b.data(), b.size());
    return BNM_OK;
}

void bnm_model_free(bnm_model *m) { delete m; }
uint32_t bnm_model_kind(const bnm_model *m) { return m ? m->kind : 0; }
uint32_t bnm_model_num_layers(const bnm_model *m) { return m ? (uint32_t)m->layers.size() : 0; }
uint32_t bnm_model_num_classes(const bnm_model *m) { return m ? m->num_classes() : 0; }
uint32_t bnm_model_input_bytes(const bnm_model *) { return 256; }

int bnm_model_layer(const bnm_model *m, uint32_t i, bnm_layer_info *info) {
    if (!m || !info || i >= m->layers.size()) return fail(BNM_EINVAL, "layer index out of range");
    *info = m->layers[i].info;
    return BNM_OK;
}

const void *bnm_model_layer_weights(const bnm_model *m, uint32_t i) {
    if (!m || i >= m->layers.size()) return nullptr;
    return m->layers[i].weights.data();
}

int bnm_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int bnm_ctx_create(const bnm_model *m, int device, bnm_ctx **out) {
    if (!m || !out) return fail(BNM_EINVAL, "null argument");
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (ndev <= 0) return fail(BNM_EHIP, "no HIP device visible");
    if (device < 0) HIP_TRY(hipGetDevice(&device));
    if (device >= ndev) return fail(BNM_EINVAL, "device index out of range");
    DeviceGuard dg(device);
    HIP_TRY(dg.err);
    bnm_ctx *c = new bnm_ctx();
    c->device = device;
    c->model = *m;
    int e = ctx_build(c);
    if (e != BNM_OK) {
        std::string keep = g_err;
        bnm_ctx_destroy(c);
        g_err = keep;
        return e;
    }
    *out = c;
    return BNM_OK;
}

void bnm_ctx_destroy(bnm_ctx *c) {
    if (!c) return;
    DeviceGuard dg(c->device);
    for (void *p : c->owned) (void)hipFree(p);
    for (auto &kv : c->scratch)
        for (DevBuf *b : {&kv.second.act_a, &kv.second.act_b, &kv.second.out32, &kv.second.cnn_feat, &kv.second.q8}) b->release(true);
    for (DevBuf *b : {&c->argmax, &c->stage_img, &c->stage_cls, &c->stage_logits})
        b->release();
    for (PinBuf *b : {&c->lat_in, &c->lat_cls, &c->lat_logits}) b->release();
    if (c->lat_stream) (void)hipStreamDestroy(c->lat_stream);
    for (auto &sl : c->slot) {
        for (PinBuf *b : {&sl.in, &sl.cls, &sl.logits}) b->release();
        for (DevBuf *b : {&sl.d_in, &sl.d_cls, &sl.d_logits}) b->release();
        if (sl.stream) (void)hipStreamDestroy(sl.stream);
        if (sl.computed) (void)hipEventDestroy(sl.computed);
    }
    delete c->copier;
    delete c;
}

int bnm_ctx_device(const bnm_ctx *c) { return c ? c->device : -1; }

int bnm_ctx_set_path(bnm_ctx *c, int path) {
    if (!c || path < BNM_PATH_AUTO || path > BNM_PATH_LAYERWISE_MFMA) return fail(BNM_EINVAL, "bad path");
    std::lock_guard<std::mutex> g(c->mu);
    int old = c->requested_path;
    c->requested_path = path;
    int e = resolve_path(c);
    if (e != BNM_OK) { c->requested_path = old; (void)resolve_path(c); }
    return e;
}

int bnm_ctx_get_path(const bnm_ctx *c) { return c ? c->path : BNM_EINVAL; }
int bnm_ctx_get_variant(const bnm_ctx *c) {
    if (!c) return BNM_EINVAL;
    return c->path == BNM_PATH_FUSED_MFMA && c->fused_ok ? c->variant : -1;
}

int bnm_ctx_set_tuning(bnm_ctx *c, int variant, int grid_blocks) {
    if (!c) return fail(BNM_EINVAL, "null ctx");
    std::lock_guard<std::mutex> g(c->mu);
    if (variant >= 0) {
        const bool ok = variant == BNM_FUSED_GENERIC ? c->generic_ok
                        : variant == BNM_FUSED_GENERIC_T1 ? (c->generic_ok && bnmk_generic_tiles(c->gdesc, c->shape.dbl, 1, false) == 1)
                        : variant == BNM_FUSED_GENERIC_T2 ? (c->generic_ok && bnmk_generic_tiles(c->gdesc, c->shape.dbl, 2, false) == 2)
                        : variant == BNM_FUSED_REGW ? c->regw_ok
                        : (c->table_ok && bnmk_fused_supported(c->shape, variant));
        if (!ok) return fail(BNM_EUNSUPPORTED, "fused kernel variant not available for this model shape");
        c->variant = variant;
    }
    c->grid_blocks = grid_blocks > 0 ? grid_blocks : 0;
    return BNM_OK;
}

int bnm_ctx_get_cnn_variant(const bnm_ctx *c) { return c ? c->cnn_variant : BNM_EINVAL; }

int bnm_ctx_set_cnn_variant(bnm_ctx *c, int variant) {
    if (!c || variant < 0 || (variant > 3 && variant < 101) || (variant > 164 && variant < 301) || variant > 316) return fail(BNM_EINVAL, "bad argument");
    std::lock_guard<std::mutex> g(c->mu);
    c->cnn_auto = false;      // (an explicit choice holds for every call size)
    if (variant == 3 || variant > 300) {      // the lane = image kernel (301..316: tiles per take)
        if (!c->cnn_li_frags) return fail(BNM_EUNSUPPORTED, "the lane = image front end serves CNN models of up to 170 channels");
        c->cnn_variant = 3;
        c->cnn_li_grab = variant > 300 ? (uint32_t)(variant - 300) : 1u;
        return BNM_OK;
    }
    c->cnn_variant = variant == 0 ? 0 : 1;
    c->cnn_grab = variant == 2 ? 0u : variant > 100 ? (uint32_t)(variant - 100) : 8u;
    return BNM_OK;
}

int bnm_ctx_set_work_batch(bnm_ctx *c, int tiles) {
    if (!c || tiles < 0 || tiles > 4096) return fail(BNM_EINVAL, "bad argument");
    std::lock_guard<std::mutex> g(c->mu);
    c->work_batch = (uint32_t)tiles;
    return BNM_OK;
}

int bnm_ctx_set_ternary_variant(bnm_ctx *c, int variant) {
    if (!c || variant < 0 || (variant > 2 && variant != 11 && variant != 12)) return fail(BNM_EINVAL, "bad argument");
    std::lock_guard<std::mutex> g(c->mu);
    if (variant % 10 == 2 && c->tern_ok && !c->tern_two)
        return fail(BNM_EUNSUPPORTED, "the two-images-per-lane ternary kernel exists for 96-96-96 only; this model runs variant 1 (one image per lane)");
    if (variant % 10 == 0 && c->tern_ok) {
        uint32_t n_out[4] = {0, 0, 0, 0};
        for (size_t i = 0; i < c->fc.size() && i < 4; i++) n_out[i] = c->fc[i].info.n_output;
        if (!bnmk_ternary_stream_supported(n_out, 0))
            return fail(BNM_EUNSUPPORTED, "round 1's plain ternary kernel (variant 0) exists for 96-96-96, 128-128-112, 64-64-64 and 128-128-128 only");
    }
    c->tern_variant = variant % 10;
    c->tern_dynamic = variant < 10;
    return BNM_OK;
}

int bnm_ctx_set_host_tuning(bnm_ctx *c, int mode, int copy_threads, int spin) {
    if (!c || mode < 0 || mode > 1 || copy_threads < 0 || copy_threads > 256) return fail(BNM_EINVAL, "bad argument");
    std::lock_guard<std::mutex> g(c->mu);
    c->host_mode = mode;
    if ((unsigned)copy_threads != c->host_threads) { delete c->copier; c->copier = nullptr; }
    c->host_threads = (unsigned)copy_threads;
    c->lat_spin = spin != 0;
    return BNM_OK;
}

int bnm_infer_device(bnm_ctx *c, const int8_t *d_images, uint64_t n, uint32_t *d_cls, int32_t *d_logits, void *stream) {
    if (!c) return fail(BNM_EINVAL, "null ctx");
    std::lock_guard<std::mutex> g(c->mu);
    DeviceGuard dg(c->device);
    HIP_TRY(dg.err);
    return infer_device_locked(c, d_images, n, d_cls, d_logits, nullptr, 0, (hipStream_t)stream);
}

int bnm_ctx_release_stream(bnm_ctx *c, void *stream) {
    if (!c) return fail(BNM_EINVAL, "null ctx");
    std::lock_guard<std::mutex> g(c->mu);
    DeviceGuard dg(c->device);
    HIP_TRY(dg.err);
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    hipStream_t s = stream_key((hipStream_t)stream);
    auto it = c->scratch.find(s);
    if (it != c->scratch.end()) {
        bool frozen = false;
        for (DevBuf *b : {&it->second.act_a, &it->second.act_b, &it->second.out32, &it->second.cnn_feat, &it->second.q8}) frozen = frozen || b->frozen;
        if (frozen)
            return fail(BNM_EUNSUPPORTED, "launches captured on this stream reference its scratch buffers: they stay until the context is destroyed");
        for (DevBuf *b : {&it->second.act_a, &it->second.act_b, &it->second.out32, &it->second.cnn_feat, &it->second.q8}) b->release();
        c->scratch.erase(it);
    }
    auto wt = c->work_of.find(s);
    if (wt != c->work_of.end()) {
        c->work_free.push_back(wt->second);      // all zero again: the stream has drained
        c->work_of.erase(wt);
    }
    return BNM_OK;
}

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

int bnm_fc_layer_device(const int8_t *d_act, uint32_t act_stride, const void *d_weights, int32_t bpw, uint32_t n_input,
                        uint32_t n_output, int32_t *d_out, uint64_t batch, void *stream) {
    HIP_TRY(bnmk_fc_layer(d_act, act_stride, d_weights, bpw, n_input, n_output, d_out, batch, (hipStream_t)stream));
    return BNM_OK;
}

int bnm_relunorm_device(const int32_t *d_in, uint32_t n, int8_t *d_out, uint32_t out_stride, uint32_t *d_argmax,
                        uint64_t batch, void *stream) {
    HIP_TRY(bnmk_relunorm(d_in, n, d_out, out_stride, d_argmax, batch, (hipStream_t)stream));
    return BNM_OK;
}

int bnm_unpack_layer_host(const void *weights, int32_t bpw, uint32_t n_input, uint32_t n_output, int8_t *lo, int8_t *hi,
                          uint32_t row_stride) {
    if (!weights || !lo || row_stride % 4u) return fail(BNM_EINVAL, "bad argument");
    uint64_t cnt = bnm_fc_weight_count(bpw, n_input, n_output);
    size_t wbytes = (size_t)cnt * (bpw == 64 ? 2 : 4);
    size_t rbytes = (size_t)n_output * row_stride;
    ScopedDev dw, dlo, dhi;
    if (int e = dw.ensure(wbytes ? wbytes : 16)) return e;
    if (int e = dlo.ensure(rbytes)) return e;
    if (int e = dhi.ensure(rbytes)) return e;
    if (wbytes) HIP_TRY(hipMemcpy(dw.p, weights, wbytes, hipMemcpyHostToDevice));
    uint32_t n_real = n_input < row_stride ? n_input : row_stride;
    HIP_TRY(bnmk_unpack_rows(dw.p, bpw, n_input, n_real, n_output, (int8_t *)dlo.p, (int8_t *)dhi.p, row_stride, nullptr));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(lo, dlo.p, rbytes, hipMemcpyDeviceToHost));
    if (hi) HIP_TRY(hipMemcpy(hi, dhi.p, rbytes, hipMemcpyDeviceToHost));
    return BNM_OK;
}

int bnm_quantize_input_device(const float *d_x, uint64_t n, int8_t *d_out, void *stream) {
    if (n && (!d_x || !d_out)) return fail(BNM_EINVAL, "null pointer");
    if (((uintptr_t)d_x & 15u) || ((uintptr_t)d_out & 3u)) return fail(BNM_EINVAL, "d_x must be 16-byte aligned");
    HIP_TRY(bnmk_quantize_input(d_x, n, d_out, (hipStream_t)stream));
    return BNM_OK;
}

int bnm_infer_float_device(bnm_ctx *c, const float *d_x, uint64_t n, uint32_t *d_cls, int32_t *d_logits, void *stream) {
    if (!c) return fail(BNM_EINVAL, "null ctx");
    if (!n) return BNM_OK;
    if (!d_x || !d_cls) return fail(BNM_EINVAL, "null device pointer");
    if ((uintptr_t)d_x & 15u) return fail(BNM_EINVAL, "d_x must be 16-byte aligned");
    std::lock_guard<std::mutex> g(c->mu);
    DeviceGuard dg(c->device);
    HIP_TRY(dg.err);
    hipStream_t s = (hipStream_t)stream;
    const uint32_t ncls = c->model.num_classes();
    // chunks of 2^22 images (1 GiB of int8 scratch per stream): quantise, then the model's kernels, in stream order
    const uint64_t chunk = 1ull << 22;
    DevBuf &q8 = stream_scratch(c, s).q8;
    if (int e = q8.ensure((size_t)(n < chunk ? n : chunk) * 256 + 64)) return e;
    for (uint64_t off = 0; off < n; off += chunk) {
        const uint64_t cn = n - off < chunk ? n - off : chunk;
        HIP_TRY(bnmk_quantize_input(d_x + off * 256, cn, (int8_t *)q8.p, s));
        if (int e = infer_device_locked(c, (const int8_t *)q8.p, cn, d_cls + off, d_logits ? d_logits + off * ncls : nullptr, nullptr, 0, s))
            return e;
    }
    return BNM_OK;
}

uint64_t bnm_qat_workspace_bytes(uint32_t d, uint32_t k) { return bnmk_qat_workspace_bytes(d, k); }

int bnm_qat_bitlinear_forward_device(const float *d_x, uint64_t n, uint32_t d, const float *d_w, uint32_t k, const float *d_s,
                                     uint32_t s_count, int quant_type, int norm_type, float *d_y, void *d_workspace,
                                     uint64_t workspace_bytes, float *d_x_int_out, float *d_x_scale_out, float *d_w_deq_out,
                                     void *stream) {
    if (!d_w || !d_s || !d_workspace || (n && (!d_x || !d_y))) return fail(BNM_EINVAL, "null pointer");
    if (d == 0 || k == 0 || d > 1024u) return fail(BNM_EINVAL, "need 1 <= d <= 1024 and k >= 1");
    if (s_count != 1u && s_count != k) return fail(BNM_EINVAL, "s_count must be 1 (PerTensor) or k (PerOutput)");
    if (quant_type < BNM_QAT_NONE || quant_type > BNM_QAT_8BIT) return fail(BNM_EINVAL, "unknown quant_type");
    if (norm_type < BNM_QAT_NORM_RMS || norm_type > BNM_QAT_NORM_NONE) return fail(BNM_EINVAL, "unknown norm_type");
    if (workspace_bytes < bnmk_qat_workspace_bytes(d, k)) return fail(BNM_EINVAL, "workspace too small (bnm_qat_workspace_bytes)");
    if ((uintptr_t)d_workspace & 3u) return fail(BNM_EINVAL, "workspace must be 4-byte aligned");
    HIP_TRY(bnmk_qat_bitlinear_forward(d_x, n, d, d_w, k, d_s, s_count, quant_type, norm_type, d_y, (float *)d_workspace,
                                       d_x_int_out, d_x_scale_out, d_w_deq_out, (hipStream_t)stream));
    return BNM_OK;
}

int bnm_qat_bitconv2d_forward_device(const float *d_x, uint64_t n, uint32_t cin, uint32_t h, uint32_t w, const float *d_w,
                                     uint32_t cout, uint32_t kh, uint32_t kw, uint32_t pad, uint32_t stride, uint32_t groups,
                                     const float *d_s, int quant_type, int norm_type, float *d_y, void *d_workspace,
                                     uint64_t workspace_bytes, void *stream) {
    if (!d_w || !d_s || !d_workspace || (n && (!d_x || !d_y))) return fail(BNM_EINVAL, "null pointer");
    if (!cin || !cout || !kh || !kw || !h || !w || !stride || !groups) return fail(BNM_EINVAL, "zero dimension");
    if (cin % groups || cout % groups) return fail(BNM_EINVAL, "in_channels and out_channels must be multiples of groups");
    if (h + 2u * pad < kh || w + 2u * pad < kw) return fail(BNM_EINVAL, "kernel larger than the padded plane");
    if (quant_type < BNM_QAT_NONE || quant_type > BNM_QAT_8BIT) return fail(BNM_EINVAL, "unknown quant_type");
    if (norm_type != BNM_QAT_NORM_RMS && norm_type != BNM_QAT_NORM_NONE) return fail(BNM_EINVAL, "norm_type must be RMS or NONE");
    if (n * groups > 0x7fffffffull) return fail(BNM_EINVAL, "n * groups too large for one launch");
    if (bnmk_qat_bitconv2d_lds_bytes(cin, h, w, cout, kh, kw, pad, groups) > 160u * 1024u)
        return fail(BNM_EUNSUPPORTED, "a group's input planes + taps exceed 160 KiB of LDS");
    if (workspace_bytes < bnmk_qat_workspace_bytes((cin / groups) * kh * kw, cout))
        return fail(BNM_EINVAL, "workspace too small (bnm_qat_workspace_bytes((cin / groups) * kh * kw, cout))");
    HIP_TRY(bnmk_qat_bitconv2d_forward(d_x, n, cin, h, w, d_w, cout, kh, kw, pad, stride, groups, d_s, quant_type, norm_type, d_y,
                                       (float *)d_workspace, (hipStream_t)stream));
    return BNM_OK;
}

int bnm_synth_fill_device(int8_t *d_images, uint64_t first, uint64_t count, uint64_t seed, int dist, void *stream) {
    if (count && !d_images) return fail(BNM_EINVAL, "null pointer");
    if (dist != BNM_DIST_U && dist != BNM_DIST_M) return fail(BNM_EINVAL, "dist must be 0 or 1");
    HIP_TRY(bnmk_synth_fill(d_images, first, count, seed, dist, (hipStream_t)stream));
    return BNM_OK;
}

int bnm_class_digest_device(const uint32_t *d_cls, uint64_t first, uint64_t n, uint64_t *d_out, uint32_t n_bins, void *stream) {
    if (n && (!d_cls || !d_out)) return fail(BNM_EINVAL, "null pointer");
    if (n_bins > 64) return fail(BNM_EINVAL, "n_bins <= 64");
    HIP_TRY(bnmk_class_digest(d_cls, first, n, d_out, n_bins, (hipStream_t)stream));
    return BNM_OK;
}

int bnm_stream_read_device(const void *d_src, uint64_t bytes, uint32_t *d_sink, void *stream) {
    if (bytes && (!d_src || !d_sink)) return fail(BNM_EINVAL, "null pointer");
    if ((uintptr_t)d_src & 15u) return fail(BNM_EINVAL, "d_src must be 16-byte aligned");
    HIP_TRY(bnmk_stream_read(d_src, bytes, d_sink, (hipStream_t)stream));
    return BNM_OK;
}

int bnm_stream_rw_device(const void *d_src, uint64_t n_rows, void *d_dst, uint32_t out_bytes_per_row, uint32_t mode, void *stream) {
    if (n_rows && (!d_src || !d_dst)) return fail(BNM_EINVAL, "null pointer");
    if (((uintptr_t)d_src | (uintptr_t)d_dst) & 15u) return fail(BNM_EINVAL, "buffers must be 16-byte aligned");
    if (!out_bytes_per_row || (out_bytes_per_row & 3u) || out_bytes_per_row > 4096u) return fail(BNM_EINVAL, "out_bytes_per_row: a multiple of 4 up to 4096");
    HIP_TRY(bnmk_stream_rw(d_src, n_rows, d_dst, out_bytes_per_row, mode, (hipStream_t)stream));
    return BNM_OK;
}

#ifdef BNM_DIAG
// ---- diagnostic library only (bitnetmcu_amd/build.py --diag; declared in csrc/bnm_diag.h, not in the public header) ----
int bnm_diag_stream_device(const int8_t *d_images, uint64_t n, int mode, int grid_blocks, uint32_t *d_out, void *stream) {
    if (n && (!d_images || !d_out)) return fail(BNM_EINVAL, "null pointer");
    if (mode < 0 || mode > 7) return fail(BNM_EINVAL, "mode must be 0..7");
    if (mode >= 5) {   // pipe-overlap probe: n = tiles per wave, no image traffic
        HIP_TRY(bnmk_diag_pipes(mode, n, d_out, (hipStream_t)stream));
        return BNM_OK;
    }
    HIP_TRY(bnmk_diag_stream(d_images, n, mode, grid_blocks, d_out, (hipStream_t)stream));
    return BNM_OK;
}
// the fused kernels read tile (t mod wrap) instead of tile t: WRONG class ids by design, timing only
#ifdef BNM_DIAG_TIMING
int bnm_diag_cnn_set_record(uint64_t *d_rec) {
    HIP_TRY(bnmk_diag_cnn_set_record(d_rec));
    return BNM_OK;
}
#endif
int bnm_diag_set_src_wrap(bnm_ctx *c, uint64_t wrap) {
    if (!c) return fail(BNM_EINVAL, "null ctx");
    c->diag_src_wrap = wrap;
    return BNM_OK;
}
#endif

// ---- RCCL, bound at run time ---------------------------------------------------------------------------------------
// Only bnm_run_synth_multi_gpu needs it, so the library does not link librccl (a Bitnet_inf.dll must load wherever the HIP
// runtime does): dlopen at first use.  Types come from <rccl/rccl.h>, the entry points through these pointers.
struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok() const { return CommInitAll && CommDestroy && Broadcast && AllReduce && GetErrorString; }
};
static const Rccl &rccl() {
    static Rccl r = [] {
        Rccl x;
        if (std::getenv("BNM_NO_RCCL")) return x;      // (tests: exercise the host-transport fallback)
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            x.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (x.lib) break;
        }
        if (!x.lib) return x;
        x.CommInitAll = (decltype(x.CommInitAll))dlsym(x.lib, "ncclCommInitAll");
        x.CommDestroy = (decltype(x.CommDestroy))dlsym(x.lib, "ncclCommDestroy");
        x.Broadcast = (decltype(x.Broadcast))dlsym(x.lib, "ncclBroadcast");
        x.AllReduce = (decltype(x.AllReduce))dlsym(x.lib, "ncclAllReduce");
        x.GetErrorString = (decltype(x.GetErrorString))dlsym(x.lib, "ncclGetErrorString");
        return x;
    }();
    return r;
}
static thread_local const char *g_multi_gpu_transport = "none";
const char *bnm_multi_gpu_transport(void) { return g_multi_gpu_transport; }

// SURVEY.md 8(e) / north_star: "batch-shard over xGMI with RCCL model bcast".  One HOST THREAD per device (device setup, the
// launch and the wait of one GPU never sit behind another GPU's), one single-process RCCL communicator per device
// (ncclCommInitAll):
//   * the model leaves rank 0 as its BNMBLOB (bnm_model_to_blob, ~13 KB) through ONE ncclBroadcast over xGMI; ranks > 0 rebuild
//     their model from the bytes they received (bnm_model_from_blob) - the host uploads it to device 0 only;
//   * every rank generates its own contiguous shard of the synthetic stream on its own GPU (no image byte crosses a link) and
//     runs the whole-model path on it: an untimed pass, a host barrier, the timed pass;
//   * the order-independent digest + class histogram of the shards meet in ONE ncclAllReduce (uint64 sum, <= 65 words).
// Without a loadable librccl (or BNM_NO_RCCL set) the same threads run with the host as transport: it uploads the model to
// every device and adds the digests itself; bnm_multi_gpu_transport() says which one the last call used ("rccl" / "host").
int bnm_run_synth_multi_gpu(const bnm_model *m, uint64_t n_total, int n_gpus, int dist, uint64_t seed, uint64_t *digest_hist,
                            uint32_t n_bins, double *seconds) {
    if (!m || !digest_hist || n_bins > 64) return fail(BNM_EINVAL, "bad argument");
    if (dist != BNM_DIST_U && dist != BNM_DIST_M) return fail(BNM_EINVAL, "dist must be 0 or 1");
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (ndev <= 0) return fail(BNM_EHIP, "no HIP device visible");
    int caller_dev = 0;
    if (hipGetDevice(&caller_dev) != hipSuccess) caller_dev = 0;
    struct Restore {       // RCCL's init and the setup below touch every device: the caller's current device comes back at every exit
        int dev;
        ~Restore() { (void)hipSetDevice(dev); }
    } restore{caller_dev};
    const int G = (n_gpus <= 0 || n_gpus > ndev) ? ndev : n_gpus;
    const Rccl &nc = rccl();
    std::vector<ncclComm_t> comms(G, nullptr);
    bool use_rccl = nc.ok();
    if (use_rccl) {
        std::vector<int> devs(G);
        for (int g = 0; g < G; g++) devs[g] = g;
        ncclResult_t r = nc.CommInitAll(comms.data(), G, devs.data());
        if (r != ncclSuccess) return fail(BNM_EHIP, std::string("ncclCommInitAll: ") + nc.GetErrorString(r));
    }
    g_multi_gpu_transport = use_rccl ? "rccl" : "host";
    const size_t blob_bytes = bnm_model_blob_size(m);
    std::vector<uint8_t> blob0(blob_bytes);
    if (bnm_model_to_blob(m, blob0.data(), blob_bytes) != BNM_OK) return fail(BNM_EINVAL, "model does not serialise");

    // a reusable host barrier for the rank threads; `failed` is examined behind it, so that either every rank enters the next
    // collective or none does
    struct Barrier {
        std::mutex mu;
        std::condition_variable cv;
        int n, waiting = 0;
        uint64_t gen = 0;
        explicit Barrier(int n_) : n(n_) {}
        void wait() {
            std::unique_lock<std::mutex> l(mu);
            const uint64_t my = gen;
            if (++waiting == n) { waiting = 0; gen++; cv.notify_all(); }
            else cv.wait(l, [&] { return gen != my; });
        }
    } barrier(G);
    std::atomic<bool> failed{false};
    std::vector<std::string> errors(G);
    std::vector<double> elapsed(G, 0.0);
    std::vector<std::vector<uint64_t>> host_digest(G, std::vector<uint64_t>(65, 0));
    const uint64_t base = n_total / G, rem = n_total % G;      // contiguous shards differing by at most one image (dist.shard_range)

    auto rank_main = [&](int g) {
        auto bad = [&](const std::string &what) { errors[g] = "GPU " + std::to_string(g) + ": " + what; failed = true; };
        auto hip_ok = [&](hipError_t e, const char *what) { if (e != hipSuccess) bad(std::string(what) + ": " + hipGetErrorString(e)); return e == hipSuccess; };
        auto nccl_ok = [&](ncclResult_t r, const char *what) { if (r != ncclSuccess) bad(std::string(what) + ": " + nc.GetErrorString(r)); return r == ncclSuccess; };
        const uint64_t first = (uint64_t)g * base + ((uint64_t)g < rem ? (uint64_t)g : rem), count = base + ((uint64_t)g < rem ? 1 : 0);
        hipStream_t st = nullptr;
        bnm_model *mine = nullptr;
        bnm_ctx *ctx = nullptr;
        ScopedDev d_blob, img, cls, dig;
        bool up = hip_ok(hipSetDevice(g), "hipSetDevice") && hip_ok(hipStreamCreateWithFlags(&st, hipStreamNonBlocking), "hipStreamCreate") &&
                  d_blob.ensure(blob_bytes) == BNM_OK;
        if (!up && !failed) bad("device setup failed");
        // ---- the model: rank 0's blob to everybody ----------------------------------------------------------------
        if (up && (g == 0 || !use_rccl)) hip_ok(hipMemcpyAsync(d_blob.p, blob0.data(), blob_bytes, hipMemcpyHostToDevice, st), "blob upload");
        barrier.wait();
        if (!failed && use_rccl)
            nccl_ok(nc.Broadcast(d_blob.p, d_blob.p, blob_bytes, ncclUint8, 0, comms[g], st), "ncclBroadcast(model blob)");
        std::vector<uint8_t> got(blob_bytes);
        if (!failed && hip_ok(hipMemcpyAsync(got.data(), d_blob.p, blob_bytes, hipMemcpyDeviceToHost, st), "blob download") &&
            hip_ok(hipStreamSynchronize(st), "model broadcast")) {
            // every rank - the root too - builds its model from the bytes that came out of the collective
            if (bnm_model_from_blob(got.data(), blob_bytes, &mine) != BNM_OK) bad(std::string("received blob does not parse: ") + bnm_last_error());
            else if (bnm_ctx_create(mine, g, &ctx) != BNM_OK) bad(std::string("bnm_ctx_create: ") + bnm_last_error());
        }
        // ---- the shard: generated where it is consumed -----------------------------------------------------------
        if (!failed && (img.ensure((size_t)(count ? count : 1) * 256) != BNM_OK || cls.ensure((size_t)(count ? count : 1) * 4) != BNM_OK ||
                        dig.ensure(65 * 8) != BNM_OK)) bad("shard buffers");
        if (!failed) {
            hip_ok(hipMemsetAsync(dig.p, 0, 65 * 8, st), "hipMemsetAsync");
            hip_ok(bnmk_synth_fill((int8_t *)img.p, first, count, seed, dist, st), "bnmk_synth_fill");
            // one untimed pass first (clock ramp, code upload, first touch of the counters): `seconds` then is a warm launch
            if (bnm_infer_device(ctx, (const int8_t *)img.p, count, (uint32_t *)cls.p, nullptr, st) != BNM_OK) bad(bnm_last_error());
            hip_ok(hipStreamSynchronize(st), "warm-up pass");
        }
        barrier.wait();
        // ---- the timed pass: all ranks start together, each stops its own clock -------------------------------------
        if (!failed) {
            const auto t0 = std::chrono::steady_clock::now();
            if (bnm_infer_device(ctx, (const int8_t *)img.p, count, (uint32_t *)cls.p, nullptr, st) != BNM_OK) bad(bnm_last_error());
            hip_ok(hipStreamSynchronize(st), "kernel execution");
            elapsed[g] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            hip_ok(bnmk_class_digest((const uint32_t *)cls.p, first, count, (uint64_t *)dig.p, n_bins, st), "bnmk_class_digest");
            hip_ok(hipStreamSynchronize(st), "digest");
        }
        barrier.wait();
        // ---- digest + histogram: one all-reduce ---------------------------------------------------------------------
        if (!failed && use_rccl)
            nccl_ok(nc.AllReduce(dig.p, dig.p, 1 + n_bins, ncclUint64, ncclSum, comms[g], st), "ncclAllReduce(digest)");
        if (!failed && (g == 0 || !use_rccl)) {
            hip_ok(hipMemcpyAsync(host_digest[g].data(), dig.p, sizeof(uint64_t) * (1 + n_bins), hipMemcpyDeviceToHost, st), "digest download");
        }
        if (st) (void)hipStreamSynchronize(st);
        if (ctx) bnm_ctx_destroy(ctx);
        if (mine) bnm_model_free(mine);
        d_blob.release(); img.release(); cls.release(); dig.release();
        if (st) (void)hipStreamDestroy(st);
    };
    std::vector<std::thread> threads;
    for (int g = 1; g < G; g++) threads.emplace_back(rank_main, g);
    rank_main(0);
    for (auto &t : threads) t.join();
    if (use_rccl)
        for (int g = 0; g < G; g++) if (comms[g]) (void)nc.CommDestroy(comms[g]);
    if (failed) {
        std::string all;
        for (auto &e : errors) if (!e.empty()) all += (all.empty() ? "" : "; ") + e;
        return fail(BNM_EHIP, all.empty() ? "multi-GPU run failed" : all);
    }
    std::memset(digest_hist, 0, sizeof(uint64_t) * (1 + n_bins));
    for (int g = 0; g < (use_rccl ? 1 : G); g++)      // host transport: the sum of the shards' digests, here
        for (uint32_t k = 0; k <= n_bins; k++) digest_hist[k] += host_digest[g][k];
    if (seconds) {
        *seconds = 0.0;
        for (double e : elapsed) *seconds = e > *seconds ? e : *seconds;      // the slowest rank
    }
    return G;
}

int bnm_device_malloc(void **p, size_t bytes) { HIP_TRY(hipMalloc(p, bytes)); return BNM_OK; }
int bnm_device_free(void *p) { HIP_TRY(hipFree(p)); return BNM_OK; }
int bnm_memcpy_h2d(void *d, const void *h, size_t bytes) { HIP_TRY(hipMemcpy(d, h, bytes, hipMemcpyHostToDevice)); return BNM_OK; }
int bnm_memcpy_d2h(void *h, const void *d, size_t bytes) { HIP_TRY(hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost)); return BNM_OK; }
int bnm_device_synchronize(void) { HIP_TRY(hipDeviceSynchronize()); return BNM_OK; }

}  // extern "C"

// =================================================================================================
// (A) the reference's own symbols
// =================================================================================================
// A model-bound build (Bitnet_inf.dll) links dll_stub.c, which embeds the exporter's header text between
// these two symbols.  In the plain library they are absent (weak, null).
extern "C" __attribute__((weak)) const char bnm_embedded_header_begin[];
extern "C" __attribute__((weak)) const char bnm_embedded_header_end[];

namespace {

// The reference's entry points are stateless and re-entrant (BitNetMCU_inference.c has no globals); these are re-entrant too:
// g_mu guards only the small tables below (pools, weight cache) and is never held across a launch.  Every call LEASES what it
// needs - Inference() a GPU context of the bound model, a kernel symbol a set of staging buffers + a stream - from a pool that
// grows to the host's concurrency, so eight host threads run eight calls at a time (round 3: one global mutex around the call).
std::mutex g_mu;
bnm_model *g_default_model = nullptr;      // the bound model (embedded header or bnm_bind_default_model): contexts are made from it
std::vector<bnm_ctx *> g_ctx_free;         // idle contexts of the CURRENT model
unsigned g_ctx_generation = 0;             // bumped by bnm_bind_default_model: leased contexts of an older model die on release
// Scratch of the per-function host ABI.  A call is ONE kernel launch and ONE stream synchronisation: the caller's arrays go
// through page-locked buffers that the GPU addresses directly (a memcpy on the host, no hipMemcpy), the kernel reads its inputs
// and writes its results over PCIe, and weight arrays stay on the device between calls (g_weights): a layer-by-layer host calls
// processfclayer with the same array for every image.  (Round 3's form - three or four synchronous hipMemcpy per call, the weight
// array among them - cost 45-60 us per call; DESIGN.md 6 has the measured flow of examples/mnist_test.c.)
struct SymSlot {
    PinBuf in, out, arg;
    hipStream_t stream = nullptr;
    bool ready() { return stream || hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) == hipSuccess; }
};
std::vector<SymSlot *> g_sym_free;
unsigned g_sym_active = 0;                 // leases outstanding: the weight cache is only emptied when the caller is alone
class SymLease {
public:
    SymLease() {
        std::lock_guard<std::mutex> g(g_mu);
        if (g_sym_free.empty()) s_ = new SymSlot();
        else { s_ = g_sym_free.back(); g_sym_free.pop_back(); }
        g_sym_active++;
    }
    ~SymLease() {
        std::lock_guard<std::mutex> g(g_mu);
        g_sym_free.push_back(s_);
        g_sym_active--;
    }
    SymLease(const SymLease &) = delete;
    SymLease &operator=(const SymLease &) = delete;
    SymSlot *operator->() const { return s_; }
private:
    SymSlot *s_;
};
struct WeightKey {
    const void *ptr;
    size_t bytes;
    uint64_t hash;
    bool operator<(const WeightKey &o) const { return ptr != o.ptr ? ptr < o.ptr : bytes != o.bytes ? bytes < o.bytes : hash < o.hash; }
};
struct WeightEntry {
    void *dev = nullptr;
    uint32_t n_act = 0;      // ternary layers: highest activation index a trit can touch + 1
};
std::map<WeightKey, WeightEntry> g_weights;
constexpr size_t kMaxCachedWeightArrays = 1024;      // (a 64-channel CNN host presents 3 x 64 nine-byte kernels + 3 FC arrays)

uint64_t content_hash(const void *p, size_t bytes) {      // FNV-1a over 8-byte words (+ tail bytes): ~2 us for a 12 KB array
    const uint8_t *b = (const uint8_t *)p;
    uint64_t h = 0xcbf29ce484222325ull;
    size_t i = 0;
    for (; i + 8 <= bytes; i += 8) {
        uint64_t w;
        std::memcpy(&w, b + i, 8);
        h = (h ^ w) * 0x100000001b3ull;
    }
    for (; i < bytes; i++) h = (h ^ b[i]) * 0x100000001b3ull;
    return h;
}

// The kernel symbols keep their scratch buffers on ONE device - the calling thread's current device at their first use - and run
// there whatever the current device is later (a host that switches devices between calls must not mix buffers and launches).
std::atomic<int> g_sym_dev{-1};
int sym_device() {
    int d = g_sym_dev.load(std::memory_order_acquire);
    if (d >= 0) return d;
    int mine = 0;
    if (hipGetDevice(&mine) != hipSuccess) mine = 0;
    return g_sym_dev.compare_exchange_strong(d, mine) ? mine : d;      // (the first caller's device wins)
}

[[noreturn]] void die(const char *what) {
    std::fprintf(stderr, "bitnetmcu_hip: %s: %s\n(there is no CPU fallback; a HIP device is required)\n", what, g_err.c_str());
    std::abort();
}

// (g_mu held) the bound model; the embedded BitNetMCU_model.h is parsed on first use
const bnm_model *default_model_locked() {
    if (g_default_model) return g_default_model;
    if (!bnm_embedded_header_begin || !bnm_embedded_header_end || +bnm_embedded_header_end <= +bnm_embedded_header_begin) {
        g_err = "no model bound: build Bitnet_inf.dll with bitnetmcu_amd/build.py --dll <BitNetMCU_model.h> or call "
                "bnm_bind_default_model()";
        die("Inference");
    }
    bnm_model *m = nullptr;
    if (bnm_model_from_header_text(bnm_embedded_header_begin, (size_t)(bnm_embedded_header_end - bnm_embedded_header_begin), &m) != BNM_OK)
        die("embedded BitNetMCU_model.h");
    return g_default_model = m;
}

// A context of the bound model for the duration of one call.  The first call of a thread that finds no idle context creates one
// (under g_mu: a few milliseconds, once per level of concurrency).
class CtxLease {
public:
    CtxLease() {
        std::lock_guard<std::mutex> g(g_mu);
        gen_ = g_ctx_generation;
        if (!g_ctx_free.empty()) { c_ = g_ctx_free.back(); g_ctx_free.pop_back(); return; }
        if (bnm_ctx_create(default_model_locked(), -1, &c_) != BNM_OK) die("GPU context");
    }
    ~CtxLease() {
        {
            std::lock_guard<std::mutex> g(g_mu);
            if (gen_ == g_ctx_generation) { g_ctx_free.push_back(c_); return; }
        }
        bnm_ctx_destroy(c_);      // the model was replaced while this call ran
    }
    CtxLease(const CtxLease &) = delete;
    CtxLease &operator=(const CtxLease &) = delete;
    bnm_ctx *get() const { return c_; }
private:
    bnm_ctx *c_ = nullptr;
    unsigned gen_ = 0;
};

// highest activation index a ternary layer can touch + 1 (pad trits are zero: exportquant.py:132-137)
uint32_t ternary_used_inputs(const uint16_t *w, uint32_t n_input, uint32_t n_output) {
    uint32_t per_row = n_input / 10u, used = 0;
    for (uint32_t r = 0; r < n_output; r++)
        for (uint32_t e = 0; e < per_row; e++) {
            uint32_t chunk = w[r * per_row + e];
            for (uint32_t t = 0; t < 10; t++) {
                chunk *= 3u;
                if ((chunk >> 16) != 2u && e * 10u + t + 1u > used) used = e * 10u + t + 1u;
                chunk &= 0xFFFFu;
            }
        }
    return used;
}

// The device-resident copy of a host weight array: keyed by address, length AND content (a host may reuse a buffer for other
// weights), uploaded once.  compute_n_act: called on a miss only (the ternary scan is O(weights)).  Hash, scan and upload run
// outside g_mu; entries are only freed when the table is full AND the caller holds the only lease (nobody can be launching with
// one of them: every call synchronises its stream before it gives its lease back) - otherwise the table grows past its cap
// until that is the case.
template <class F>
bool cached_weights(const void *host, size_t bytes, F compute_n_act, WeightEntry *out) {
    const WeightKey key{host, bytes, content_hash(host, bytes)};
    {
        std::lock_guard<std::mutex> g(g_mu);
        auto it = g_weights.find(key);
        if (it != g_weights.end()) { *out = it->second; return true; }
    }
    WeightEntry e;
    if (hipMalloc(&e.dev, bytes + 16) != hipSuccess || hipMemcpy(e.dev, host, bytes, hipMemcpyHostToDevice) != hipSuccess) return false;
    e.n_act = compute_n_act();
    std::vector<void *> dead;
    {
        std::lock_guard<std::mutex> g(g_mu);
        auto it = g_weights.find(key);
        if (it != g_weights.end()) {      // another thread brought the same array meanwhile
            dead.push_back(e.dev);
            e = it->second;
        } else {
            if (g_weights.size() >= kMaxCachedWeightArrays && g_sym_active == 1) {
                for (auto &kv : g_weights) dead.push_back(kv.second.dev);
                g_weights.clear();
            }
            g_weights[key] = e;
        }
    }
    for (void *d : dead) (void)hipFree(d);
    *out = e;
    return true;
}

}  // namespace

extern "C" {

int bnm_bind_default_model(const bnm_model *m) {
    if (!m) return fail(BNM_EINVAL, "null model");
    bnm_ctx *c = nullptr;      // (made first: an unsupported model must leave the bound one in place)
    int e = bnm_ctx_create(m, -1, &c);
    if (e != BNM_OK) return e;
    bnm_model *copy = new bnm_model(*m);
    std::vector<bnm_ctx *> old;
    bnm_model *old_model = nullptr;
    {
        std::lock_guard<std::mutex> g(g_mu);
        old.swap(g_ctx_free);
        old_model = g_default_model;
        g_default_model = copy;
        g_ctx_generation++;
        g_ctx_free.push_back(c);
    }
    for (bnm_ctx *o : old) bnm_ctx_destroy(o);
    delete old_model;
    return BNM_OK;
}

uint32_t BitMnistInference(int8_t *input) {
    CtxLease c;
    uint32_t cls = 0;
    if (bnm_infer_host(c.get(), input, 1, &cls, nullptr) != BNM_OK) die("BitMnistInference");
    return cls;
}

uint32_t Inference(int8_t *input) { return BitMnistInference(input); }

void processfclayer(int8_t *activations, const uint32_t *weights, int32_t bpw, uint32_t n_input, uint32_t n_output,
                    int32_t *output) {
    DeviceGuard dg(sym_device());
    if (!n_output) return;
    uint64_t cnt = bnm_fc_weight_count(bpw, n_input, n_output);
    if (!bnm_codec_known(bpw)) {
        // BitNetMCU_inference.c:202: no branch taken -> sum stays 0
        std::memset(output, 0, sizeof(int32_t) * n_output);
        return;
    }
    const size_t wbytes = (size_t)cnt * (bpw == 64 ? 2 : 4);
    SymLease sl;
    if (!sl->ready()) die("processfclayer");
    WeightEntry w;
    if (!cached_weights(weights, wbytes, [&] {
            return bpw == 64 ? ternary_used_inputs((const uint16_t *)weights, n_input, n_output) : n_input;
        }, &w)) { g_err = hipGetErrorString(hipGetLastError()); die("processfclayer"); }
    const uint32_t n_act = w.n_act, stride = n_act ? n_act : 1;
    if (sl->in.ensure(stride + 16) || sl->out.ensure((size_t)n_output * 4)) die("processfclayer");
    if (n_act) std::memcpy(sl->in.host, activations, n_act);
    bool ok = bnmk_fc_layer((const int8_t *)sl->in.dev, stride, w.dev, bpw, n_input, n_output, (int32_t *)sl->out.dev, 1, sl->stream) == hipSuccess;
    ok = ok && hipStreamSynchronize(sl->stream) == hipSuccess;
    if (!ok) { g_err = hipGetErrorString(hipGetLastError()); die("processfclayer"); }
    std::memcpy(output, sl->out.host, (size_t)n_output * 4);
}

uint32_t ReLUNorm(int32_t *input, int8_t *output, uint32_t n_input) {
    DeviceGuard dg(sym_device());
    if (!n_input) return 255;
    SymLease sl;
    if (!sl->ready() || sl->in.ensure((size_t)n_input * 4) || sl->out.ensure(n_input) || sl->arg.ensure(4)) die("ReLUNorm");
    std::memcpy(sl->in.host, input, (size_t)n_input * 4);
    bool ok = bnmk_relunorm((const int32_t *)sl->in.dev, n_input, (int8_t *)sl->out.dev, n_input, (uint32_t *)sl->arg.dev, 1, sl->stream) == hipSuccess;
    ok = ok && hipStreamSynchronize(sl->stream) == hipSuccess;
    if (!ok) { g_err = hipGetErrorString(hipGetLastError()); die("ReLUNorm"); }
    // (the results leave the staging buffers after the kernel: output may alias input, BitNetMCU_MNIST_dll.c:80)
    std::memcpy(output, sl->out.host, n_input);
    return *(const uint32_t *)sl->arg.host;
}

int32_t *processconv33ReLU(int32_t *activations, const int8_t *weights, uint32_t xy, uint32_t n_shift, int32_t *output) {
    DeviceGuard dg(sym_device());
    if (xy < 3) return output;      // no output position exists (the reference's loops do not run either)
    uint32_t o = xy - 2;
    SymLease sl;
    if (!sl->ready() || sl->in.ensure((size_t)xy * xy * 4) || sl->out.ensure((size_t)o * o * 4)) die("processconv33ReLU");
    WeightEntry w;
    if (!cached_weights(weights, 9, [] { return 0u; }, &w)) { g_err = hipGetErrorString(hipGetLastError()); die("processconv33ReLU"); }
    std::memcpy(sl->in.host, activations, (size_t)xy * xy * 4);
    bool ok = bnmk_conv33((const int32_t *)sl->in.dev, (const int8_t *)w.dev, xy, n_shift, (int32_t *)sl->out.dev, sl->stream) == hipSuccess;
    ok = ok && hipStreamSynchronize(sl->stream) == hipSuccess;
    if (!ok) { g_err = hipGetErrorString(hipGetLastError()); die("processconv33ReLU"); }
    std::memcpy(output, sl->out.host, (size_t)o * o * 4);      // (output may alias activations: copied out after the kernel)
    return output + (size_t)o * o;
}

int32_t *processmaxpool22(int32_t *activations, uint32_t xy, int32_t *output) {
    DeviceGuard dg(sym_device());
    if (xy < 2) return output;      // no output position exists
    uint32_t o = xy / 2;
    SymLease sl;
    if (!sl->ready() || sl->in.ensure((size_t)xy * xy * 4) || sl->out.ensure((size_t)o * o * 4)) die("processmaxpool22");
    std::memcpy(sl->in.host, activations, (size_t)xy * xy * 4);
    bool ok = bnmk_maxpool22((const int32_t *)sl->in.dev, xy, (int32_t *)sl->out.dev, sl->stream) == hipSuccess;
    ok = ok && hipStreamSynchronize(sl->stream) == hipSuccess;
    if (!ok) { g_err = hipGetErrorString(hipGetLastError()); die("processmaxpool22"); }
    std::memcpy(output, sl->out.host, (size_t)o * o * 4);
    return output + (size_t)o * o;
}

}  // extern "C"
