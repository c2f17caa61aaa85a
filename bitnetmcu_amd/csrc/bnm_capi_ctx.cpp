// C ABI, context part: building a model's device residency (unpacked rows, MFMA fragments, CNN tables), kernel selection,
// per-stream scratch and work-counter blocks, the bnm_ctx_* entry points (declared in include/bitnetmcu_hip.h).
#include "bnm_capi_internal.hpp"

using namespace bnm_internal;

namespace bnm_internal {

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

}  // namespace bnm_internal

namespace {

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

}  // namespace

namespace bnm_internal {

// Key of a stream in the per-stream tables.  Launches that share a key share a counter block and scratch buffers and must be
// ordered among themselves - true for a real stream handle, NOT for hipStreamPerThread: that is one constant handle value which
// names a different stream in every host thread.  Its key is a per-thread TOKEN: an odd number >= 3 from a process-wide counter,
// never reused.  A stream handle is a pointer to an object, hence even, and the runtime's other constants are 0, 1 and 2: a token
// can never equal a handle - unlike the address of a thread-local byte (round 4's token), which a later heap object may take
// over once its thread has gone.  What a context keeps for the token of a thread that has exited goes with the next eviction
// (evict_other_streams: at 32 entries).
std::atomic<uintptr_t> g_next_token{3};
bool c_is_stream_handle(hipStream_t key) { return ((uintptr_t)key & 1u) == 0; }
hipStream_t stream_key(hipStream_t s) {
    static thread_local uintptr_t token = 0;
    if (s != hipStreamPerThread) return s;
    if (!token) token = g_next_token.fetch_add(2);
    return (hipStream_t)token;
}
}  // namespace bnm_internal

namespace {

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

}  // namespace

namespace bnm_internal {

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

}  // namespace bnm_internal

namespace {

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
            c->cnn_li_plane2 = c->cnn_li_plane2_model = bnm_cnn_li_tables((const int8_t *)m.layers[0].weights.data(), (const int8_t *)m.layers[1].weights.data(),
                              (const int8_t *)m.layers[3].weights.data(), C, fr.data(), bi.data(), &c->cnn_li_sums16);
            c->cnn_li_pipe = c->cnn_li_sums16;
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
        if (int e = dev_alloc(c, &q, 8)) return e;
        HIP_TRY(hipMemset(q, 0, 8));
        c->nonfinite = (unsigned long long *)q;
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
            // (+ 16 KiB: kernels that read the fragments from global memory run their reads a few KiB ahead of the last fragment)
            if (int e = dev_alloc(c, out, (size_t)total + 16384u)) return e;
            HIP_TRY(hipMemsetAsync(*out, 0, (size_t)total + 16384u, s));
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
        c->f32_ok = m.kind == BNM_KIND_FC && c->generic_ok && bnmk_fused_f32_supported(c->gdesc, sh.dbl, 0);
        c->cnn_fused_ok = m.kind == BNM_KIND_CNN && c->cnn_li_frags && c->generic_ok && bnmk_cnn_li_fused_supported(c->channels, c->gdesc);
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

}  // namespace

extern "C" {

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
    persist_release(c);      // (a resident one-image kernel reads c->gfrags: it leaves before anything is freed)
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

const char *bnm_ctx_last_kernel(bnm_ctx *c) {
    static thread_local std::string copy;      // (the context may run another call on another thread meanwhile)
    if (!c) return "";
    std::lock_guard<std::mutex> g(c->mu);
    copy = c->last_kernel;
    return copy.c_str();
}

int bnm_ctx_set_cnn_variant(bnm_ctx *c, int variant) {
    if (!c || variant < 0 || (variant > 6 && variant < 101) || (variant > 164 && variant < 301) || (variant > 316 && variant < 401) || variant > 416)
        return fail(BNM_EINVAL, "bad argument");
    std::lock_guard<std::mutex> g(c->mu);
    c->cnn_auto = false;      // (an explicit choice holds for every call size)
    if (variant == 3 || variant == 4 || variant == 5 || variant == 6 || variant > 300) {      // the lane = image kernel (301..316: tiles per take; 4 / 401..416: the FC tail as its own launch)
        if (!c->cnn_li_frags) return fail(BNM_EUNSUPPORTED, "the lane = image front end serves CNN models of up to 170 channels");
        c->cnn_variant = 3;
        c->cnn_fuse_tail = !(variant == 4 || variant > 400);
        c->cnn_li_grab = variant > 400 ? (uint32_t)(variant - 400) : variant > 300 ? (uint32_t)(variant - 300) : 1u;
        c->cnn_li_plane2 = variant == 5 ? true : c->cnn_li_plane2_model;      // 5: as 3 with conv3's third plane kept whatever the weights say (A/B)
        c->cnn_li_pipe = variant == 6 ? false : c->cnn_li_sums16;              // 6: as 3 in the four-waves-per-SIMD form whatever the weights say (A/B)
        return BNM_OK;
    }
    c->cnn_variant = variant == 0 ? 0 : 1;
    c->cnn_grab = variant == 2 ? 0u : variant > 100 ? (uint32_t)(variant - 100) : 8u;
    return BNM_OK;
}

int bnm_ctx_cnn_planes(const bnm_ctx *c) {
    if (!c || c->model.kind != BNM_KIND_CNN || !c->cnn_li_frags) return 0;
    return c->cnn_li_plane2 ? 3 : 2;
}

int bnm_ctx_cnn_tail_fused(const bnm_ctx *c) {
    if (!c) return BNM_EINVAL;
    return (c->cnn_fused_ok && c->cnn_fuse_tail && c->cnn_variant == 3 && c->path == BNM_PATH_FUSED_MFMA) ? 1 : 0;
}

int bnm_ctx_cnn_pipelined(const bnm_ctx *c) {
    if (!c) return BNM_EINVAL;
    return (bnm_ctx_cnn_tail_fused(c) == 1 && c->cnn_li_pipe) ? 1 : 0;
}

int bnm_ctx_set_float_mode(bnm_ctx *c, int mode, int groups) {
    if (!c || mode < 0 || mode > 2 || (groups != 0 && groups != 1 && groups != 2 && groups != 4)) return fail(BNM_EINVAL, "bad argument");
    std::lock_guard<std::mutex> g(c->mu);
    if (mode == 1 && !c->f32_ok && !c->cnn_fused_ok)
        return fail(BNM_EUNSUPPORTED, "the fused float-input kernels serve FC models whose layers are at most 192 wide and CNN models "
                                      "of up to 64 channels whose FC layers are at most 96 wide");
    if (groups && c->f32_ok && !bnmk_fused_f32_supported(c->gdesc, c->shape.dbl, groups))
        return fail(BNM_EUNSUPPORTED, "this many groups in flight are not instantiated for the model's tile class");
    c->float_mode = mode;
    c->f32_groups = groups;
    return BNM_OK;
}

int bnm_ctx_float_fused(const bnm_ctx *c) {
    if (!c) return BNM_EINVAL;
    if (c->float_mode == 2 || c->path != BNM_PATH_FUSED_MFMA) return 0;
    if (c->model.kind == BNM_KIND_CNN) return (c->cnn_fused_ok && c->cnn_fuse_tail && c->cnn_variant == 3) ? 1 : 0;
    return c->f32_ok ? 1 : 0;
}

int bnm_ctx_float_nonfinite(bnm_ctx *c, uint64_t *count) {
    if (!c || !count) return fail(BNM_EINVAL, "null argument");
    std::lock_guard<std::mutex> g(c->mu);
    DeviceGuard dg(c->device);
    HIP_TRY(dg.err);
    HIP_TRY(hipDeviceSynchronize());      // (every stream the context was used on: the counter is fed by kernels)
    unsigned long long v = 0;
    HIP_TRY(hipMemcpy(&v, c->nonfinite, sizeof v, hipMemcpyDeviceToHost));
    *count = (uint64_t)v;
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

int bnm_ctx_set_persistent(bnm_ctx *c, int mode, uint32_t idle_us) {
    if (!c || mode < 0 || mode > 1 || (idle_us && (idle_us < 100u || idle_us > 10000000u))) return fail(BNM_EINVAL, "bad argument");
    std::lock_guard<std::mutex> g(c->mu);
    DeviceGuard dg(c->device);
    HIP_TRY(dg.err);
    if (mode == 1 && !(c->model.kind == BNM_KIND_FC && c->generic_ok && bnmk_persistent_supported(c->gdesc, c->shape.dbl)))
        return fail(BNM_EUNSUPPORTED, "the resident one-image kernel serves FC models with 256-byte inputs whose layers are at most 192 wide");
    if (idle_us && idle_us != c->persist_idle_us) {
        persist_stop(c);      // (a running kernel carries the old limit)
        c->persist_idle_us = idle_us;
    }
    if (mode == 0) persist_stop(c);
    c->persist_mode = mode;
    return BNM_OK;
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

}  // extern "C"
