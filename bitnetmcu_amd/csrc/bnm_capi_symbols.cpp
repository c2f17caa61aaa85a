// C ABI, group (A): the reference's own symbols (Inference, BitMnistInference, processfclayer, ReLUNorm, processconv33ReLU,
// processmaxpool22 - BitNetMCU_inference.h:15,31,45,60, BitNetMCU_MNIST_dll.c:24-26) on top of the device kernels, and the
// model a Bitnet_inf.dll is bound to.
#include "bnm_capi_internal.hpp"

using namespace bnm_internal;

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
// (the layer's geometry is part of the key: what is cached beside the device copy - n_act - depends on it, and two layers can
// share an array address, a byte count and a content: 64 x 32 and 32 x 64 four-bit weights are both 1024 bytes)
struct WeightKey {
    const void *ptr;
    size_t bytes;
    uint64_t hash;
    int32_t bpw;
    uint32_t n_input, n_output;
    bool operator<(const WeightKey &o) const {
        if (ptr != o.ptr) return ptr < o.ptr;
        if (bytes != o.bytes) return bytes < o.bytes;
        if (hash != o.hash) return hash < o.hash;
        if (bpw != o.bpw) return bpw < o.bpw;
        if (n_input != o.n_input) return n_input < o.n_input;
        return n_output < o.n_output;
    }
};
struct WeightEntry {
    void *dev = nullptr;
    uint32_t n_act = 0;      // ternary layers: highest activation index a trit can touch + 1
};
std::map<WeightKey, WeightEntry> g_weights;
constexpr size_t kMaxCachedWeightArrays = 1024;      // (a 64-channel CNN host presents 3 x 64 nine-byte kernels + 3 FC arrays)
// device copies taken out of the table while other calls were in flight: freed by the next call that finds itself alone
std::vector<void *> g_weights_retired;

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
bool cached_weights(const void *host, size_t bytes, int32_t bpw, uint32_t n_input, uint32_t n_output, F compute_n_act, WeightEntry *out) {
    const WeightKey key{host, bytes, content_hash(host, bytes), bpw, n_input, n_output};
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
            // a full table is emptied whoever else is running (a host that rewrites its weights in place must not grow it without
            // bound); the device copies that another call may be launching with right now are only RETIRED - every call
            // synchronises its stream before it returns its lease, so the first call that holds the only lease frees them
            if (g_weights.size() >= kMaxCachedWeightArrays) {
                for (auto &kv : g_weights) g_weights_retired.push_back(kv.second.dev);
                g_weights.clear();
            }
            if (g_sym_active == 1) dead.swap(g_weights_retired);
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
    if (!cached_weights(weights, wbytes, bpw, n_input, n_output, [&] {
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
    if (!cached_weights(weights, 9, 0, 9, 1, [] { return 0u; }, &w)) { g_err = hipGetErrorString(hipGetLastError()); die("processconv33ReLU"); }
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
