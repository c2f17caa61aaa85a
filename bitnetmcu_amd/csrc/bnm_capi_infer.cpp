// C ABI, device-pointer inference: the whole-model launches of every path (fused MFMA, ternary ALU, layer-wise, CNN front end +
// FC tail) and bnm_infer_device.
#include "bnm_capi_internal.hpp"

using namespace bnm_internal;

namespace {

constexpr uint64_t kChunk = 1ull << 20;   // images per internal chunk of the staged / layer-wise paths
constexpr uint64_t kCnnChunk = 1ull << 22;   // images per launch of the CNN front end when the fused FC tail follows

// ---- whole-model launches on device data -----------------------------------------------------------
bool is_generic(int variant) { return variant == BNM_FUSED_GENERIC || variant == BNM_FUSED_GENERIC_T1 || variant == BNM_FUSED_GENERIC_T2; }

// ---- what a call launches, by name (bnm_ctx_last_kernel): the same decisions as the launch code below and the kernels' own
// dispatchers (bnmk_fused_fc: whole 64-image pairs to the dual-tile kernel, a remainder to the one-tile kernel) ---------------
std::string fused_names(const bnm_ctx *c, uint64_t n) {
    if (is_generic(c->variant)) return "fused_fc_generic_kernel";
    if (c->variant == BNM_FUSED_REGW) {
        const uint64_t resident_waves = (c->grid_blocks > 0 ? (uint64_t)c->grid_blocks : (uint64_t)bnm_num_cus()) * 4ull;
        const uint64_t n_main = (n >> 6) < resident_waves ? 0ull : n & ~63ull;
        return std::string(n_main ? "fused_fc_regw_kernel" : "") + (n_main && n > n_main ? "+" : "") + (n > n_main ? "fused_fc_generic_kernel" : "");
    }
    if (c->variant == 3 || c->variant == 5 || c->variant == 6) {
        const uint64_t n_main = n & ~63ull;
        return std::string(n_main ? "fused_fc_dual_kernel" : "") + (n_main && n > n_main ? "+" : "") + (n > n_main ? "fused_fc_kernel" : "");
    }
    return "fused_fc_kernel";
}
std::string tail_names(const bnm_ctx *c, int path, uint64_t n) {
    if (path == BNM_PATH_FUSED_MFMA) return fused_names(c, n);
    if (path == BNM_PATH_TERNARY_ALU) return c->tern_variant == 0 ? "ternary_alu_kernel" : "ternary_stream_kernel";
    return path == BNM_PATH_LAYERWISE_MFMA ? "fc_layer_mfma_kernel+relunorm_kernel" : "fc_layer_bitserial_kernel+relunorm_kernel";
}

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

}  // namespace

namespace bnm_internal {

int infer_device_locked(bnm_ctx *c, const int8_t *d_images, uint64_t n, uint32_t *d_cls, int32_t *d_logits,
                        int8_t *d_acts_tap, uint32_t tap_stride, hipStream_t s) {
    if (!n) return BNM_OK;
    if (!d_images || !d_cls) return fail(BNM_EINVAL, "null device pointer");
    if (((uintptr_t)d_images & 15u) != 0) return fail(BNM_EINVAL, "d_images must be 16-byte aligned");
    const uint32_t ncls = c->model.num_classes();
    const int path = d_acts_tap ? BNM_PATH_LAYERWISE_ALU : c->path;
    if (c->model.kind == BNM_KIND_FC) {
        c->last_kernel = tail_names(c, path, n);
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
    // CNN, one kernel: the lane = image front end with the FC tail in the same wave (no act rows in HBM, no scratch, one launch
    // however many images) wherever that kernel serves the model and the lane = image front end would run this call
    if (!d_acts_tap && cnn_one_kernel_call(c, n)) {
        uint32_t *block = nullptr;
        if (int e = work_block(c, s, &block)) return e;
        HIP_TRY(bnmk_cnn_li_fused(d_images, false, n, c->cnn_li_frags, c->cnn_li_bias, c->channels, c->cnn_li_plane2, c->cnn_li_pipe, c->gfrags, c->gdesc, c->shape.dbl, d_cls, d_logits,
                                  block, c->cnn_li_grab, nullptr, s));
        c->last_kernel = c->cnn_li_pipe ? "cnn_li_fused_pipe_kernel" : "cnn_li_fused_kernel";
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
        const bool lane_image = c->cnn_variant == 3 && c->cnn_li_frags && !small_call;
        if (off == 0)
            c->last_kernel = std::string(lane_image ? "cnn_li_kernel" : c->cnn_variant ? "cnn_front_mfma_kernel" : "cnn_front_kernel") + "+" +
                             tail_names(c, path, cn);
        if (lane_image)
            HIP_TRY(bnmk_cnn_front_li(d_images + off * 256, cn, c->cnn_li_frags, c->cnn_li_bias, c->channels, c->cnn_li_plane2, acts, AS, block, c->cnn_li_grab, s));
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

}  // namespace bnm_internal

extern "C" {

int bnm_infer_device(bnm_ctx *c, const int8_t *d_images, uint64_t n, uint32_t *d_cls, int32_t *d_logits, void *stream) {
    if (!c) return fail(BNM_EINVAL, "null ctx");
    std::lock_guard<std::mutex> g(c->mu);
    DeviceGuard dg(c->device);
    HIP_TRY(dg.err);
    return infer_device_locked(c, d_images, n, d_cls, d_logits, nullptr, 0, (hipStream_t)stream);
}

}  // extern "C"
