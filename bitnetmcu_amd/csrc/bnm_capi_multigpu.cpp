// C ABI, multi-GPU driver for C hosts: one host thread and one single-process RCCL communicator per GPU.
#include "bnm_capi_internal.hpp"
#include <dlfcn.h>

// The few RCCL declarations this unit needs, stated here (values and signatures of rccl.h / nccl.h, stable across releases): the
// library is bound with dlopen at run time, so neither its import library nor its headers are a build requirement.
extern "C" {
typedef struct ncclComm *ncclComm_t;
typedef int ncclResult_t;      // (an int, not an enum of one enumerator: every failure code must be a value the type can hold)
enum { ncclSuccess = 0 };      // (any other value is a failure: ncclGetErrorString says which)
typedef enum { ncclUint8 = 1, ncclUint64 = 5 } ncclDataType_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
}

using namespace bnm_internal;

extern "C" {

// ---- RCCL, bound at run time ---------------------------------------------------------------------------------------
// Only bnm_run_synth_multi_gpu needs it, so the library does not link librccl (a Bitnet_inf.dll must load wherever the HIP
// runtime does): dlopen at first use.  Types come from <rccl/rccl.h>, the entry points through these pointers.
struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    // (ncclCommAbort is part of the contract: the failure path must be able to leave a half-entered collective - without it the host transport runs)
    bool ok() const { return CommInitAll && CommDestroy && CommAbort && Broadcast && AllReduce && GetErrorString; }
};
static const Rccl &rccl() {
    static Rccl r = [] {
        Rccl x;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            x.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (x.lib) break;
        }
        if (!x.lib) return x;
        x.CommInitAll = (decltype(x.CommInitAll))dlsym(x.lib, "ncclCommInitAll");
        x.CommDestroy = (decltype(x.CommDestroy))dlsym(x.lib, "ncclCommDestroy");
        x.CommAbort = (decltype(x.CommAbort))dlsym(x.lib, "ncclCommAbort");
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
    // (everything that can fail without RCCL first: no exit below leaves communicators behind)
    const size_t blob_bytes = bnm_model_blob_size(m);
    std::vector<uint8_t> blob0(blob_bytes);
    if (bnm_model_to_blob(m, blob0.data(), blob_bytes) != BNM_OK) return fail(BNM_EINVAL, "model does not serialise");
    const Rccl &nc = rccl();
    std::vector<ncclComm_t> comms(G, nullptr);
    // BNM_NO_RCCL (read at every call; tests: the host-transport fallback)
    const bool use_rccl = nc.ok() && !std::getenv("BNM_NO_RCCL");
    if (use_rccl) {
        std::vector<int> devs(G);
        for (int g = 0; g < G; g++) devs[g] = g;
        ncclResult_t r = nc.CommInitAll(comms.data(), G, devs.data());
        if (r != ncclSuccess) {
            for (ncclComm_t cm : comms) if (cm) (void)nc.CommDestroy(cm);      // (whatever a partial init left)
            return fail(BNM_EHIP, std::string("ncclCommInitAll: ") + nc.GetErrorString(r));
        }
    }
    g_multi_gpu_transport = use_rccl ? "rccl" : "host";

    // a reusable host barrier for the rank threads; `failed` is examined behind it, so that either every rank enters the next
    // collective or none does
    // wait() returns ONE value of `failed` to all ranks: the last rank to arrive samples it while everybody else is parked, so a
    // rank that fails right behind the barrier cannot make some ranks skip a collective that the others have entered
    std::atomic<bool> failed{false};
    struct Barrier {
        std::mutex mu;
        std::condition_variable cv;
        int n, waiting = 0;
        uint64_t gen = 0;
        bool verdict = false;
        std::atomic<bool> &flag;
        Barrier(int n_, std::atomic<bool> &f) : n(n_), flag(f) {}
        bool wait() {
            std::unique_lock<std::mutex> l(mu);
            const uint64_t my = gen;
            if (++waiting == n) { waiting = 0; verdict = flag.load(); gen++; cv.notify_all(); }
            else cv.wait(l, [&] { return gen != my; });
            return verdict;
        }
    } barrier(G, failed);
    std::vector<std::string> errors(G);
    std::vector<double> elapsed(G, 0.0);
    std::vector<std::vector<uint64_t>> host_digest(G, std::vector<uint64_t>(65, 0));
    const uint64_t base = n_total / G, rem = n_total % G;      // contiguous shards differing by at most one image (dist.shard_range)

    auto rank_main = [&](int g) {
        auto bad = [&](const std::string &what) { errors[g] = "GPU " + std::to_string(g) + ": " + what; failed = true; };
        auto hip_ok = [&](hipError_t e, const char *what) { if (e != hipSuccess) bad(std::string(what) + ": " + hipGetErrorString(e)); return e == hipSuccess; };
        auto nccl_ok = [&](ncclResult_t r, const char *what) { if (r != ncclSuccess) bad(std::string(what) + ": " + nc.GetErrorString(r)); return r == ncclSuccess; };
        // behind a collective's enqueue: did EVERY rank enqueue it?  If one did not, the others' streams would wait for it for ever:
        // each rank aborts its communicator (which releases the stream) instead of synchronising
        auto entered_by_all = [&]() {
            const bool broken = barrier.wait();
            if (broken && use_rccl && comms[g] && nc.CommAbort) { (void)nc.CommAbort(comms[g]); comms[g] = nullptr; }
            return !broken;
        };
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
        bool stop = barrier.wait();      // (all ranks agree: the collective is entered by every rank or by none)
        if (!stop && use_rccl)
            nccl_ok(nc.Broadcast(d_blob.p, d_blob.p, blob_bytes, ncclUint8, 0, comms[g], st), "ncclBroadcast(model blob)");
        std::vector<uint8_t> got(blob_bytes);
        if (entered_by_all() && hip_ok(hipMemcpyAsync(got.data(), d_blob.p, blob_bytes, hipMemcpyDeviceToHost, st), "blob download") &&
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
        stop = barrier.wait();
        // ---- the timed pass: all ranks start together, each stops its own clock -------------------------------------
        if (!stop) {
            const auto t0 = std::chrono::steady_clock::now();
            if (bnm_infer_device(ctx, (const int8_t *)img.p, count, (uint32_t *)cls.p, nullptr, st) != BNM_OK) bad(bnm_last_error());
            hip_ok(hipStreamSynchronize(st), "kernel execution");
            elapsed[g] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            hip_ok(bnmk_class_digest((const uint32_t *)cls.p, first, count, (uint64_t *)dig.p, n_bins, st), "bnmk_class_digest");
            hip_ok(hipStreamSynchronize(st), "digest");
        }
        stop = barrier.wait();
        // ---- digest + histogram: one all-reduce ---------------------------------------------------------------------
        if (!stop && use_rccl)
            nccl_ok(nc.AllReduce(dig.p, dig.p, 1 + n_bins, ncclUint64, ncclSum, comms[g], st), "ncclAllReduce(digest)");
        if (entered_by_all() && (g == 0 || !use_rccl)) {
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
    if (use_rccl)      // a failed run may have left a collective half-entered: abort the communicators rather than drain them
        for (int g = 0; g < G; g++)
            if (comms[g]) (void)((failed && nc.CommAbort) ? nc.CommAbort(comms[g]) : nc.CommDestroy(comms[g]));
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

}  // extern "C"
