// C ABI, float inputs: the reference's Python-side input quantisation on the GPU (test_inference.py:140-141) and float images ->
// class ids in one call.
#include "bnm_capi_internal.hpp"

using namespace bnm_internal;

extern "C" {

int bnm_quantize_input_counted_device(const float *d_x, uint64_t n, int8_t *d_out, uint64_t *d_nonfinite, void *stream) {
    if (n && (!d_x || !d_out)) return fail(BNM_EINVAL, "null pointer");
    if (((uintptr_t)d_x & 15u) || ((uintptr_t)d_out & 3u)) return fail(BNM_EINVAL, "d_x must be 16-byte aligned");
    if ((uintptr_t)d_nonfinite & 7u) return fail(BNM_EINVAL, "d_nonfinite must be 8-byte aligned");
    HIP_TRY(bnmk_quantize_input(d_x, n, d_out, (unsigned long long *)d_nonfinite, (hipStream_t)stream));
    return BNM_OK;
}

int bnm_quantize_input_device(const float *d_x, uint64_t n, int8_t *d_out, void *stream) {
    return bnm_quantize_input_counted_device(d_x, n, d_out, nullptr, stream);
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
    // ONE kernel where it exists: floats in, class ids out (1,028 bytes per image instead of 1,540; bnm_fused_f32_kernel.hpp)
    if (c->model.kind == BNM_KIND_FC && bnm_ctx_float_fused(c)) {
        uint32_t *block = nullptr;
        if (int e = work_block(c, s, &block)) return e;
        HIP_TRY(bnmk_fused_f32(c->gdesc, c->shape.dbl, c->f32_groups, c->grid_blocks, d_x, n, c->gfrags, d_cls, d_logits, block,
                               c->work_batch, c->nonfinite, s));
        c->last_kernel = "fused_fc_f32_kernel";
        return BNM_OK;
    }
    // CNN models: the one-kernel form quantises in front of its convolution operands (cnn_li_fused_kernel<.., true>)
    if (c->float_mode != 2 && cnn_one_kernel_call(c, n)) {
        uint32_t *block = nullptr;
        if (int e = work_block(c, s, &block)) return e;
        HIP_TRY(bnmk_cnn_li_fused(d_x, true, n, c->cnn_li_frags, c->cnn_li_bias, c->channels, c->cnn_li_plane2, c->cnn_li_pipe, c->gfrags, c->gdesc, c->shape.dbl, d_cls, d_logits,
                                  block, c->cnn_li_grab, c->nonfinite, s));
        c->last_kernel = c->cnn_li_pipe ? "cnn_li_fused_pipe_kernel<float>" : "cnn_li_fused_kernel<float>";
        return BNM_OK;
    }
    if (c->float_mode == 1) return fail(BNM_EUNSUPPORTED, "no fused float-input kernel serves this call on the context's current path");
    // chunks of 2^22 images (1 GiB of int8 scratch per stream): quantise, then the model's kernels, in stream order
    const uint64_t chunk = 1ull << 22;
    DevBuf &q8 = stream_scratch(c, s).q8;
    if (int e = q8.ensure((size_t)(n < chunk ? n : chunk) * 256 + 64)) return e;
    for (uint64_t off = 0; off < n; off += chunk) {
        const uint64_t cn = n - off < chunk ? n - off : chunk;
        HIP_TRY(bnmk_quantize_input(d_x + off * 256, cn, (int8_t *)q8.p, c->nonfinite, s));
        if (int e = infer_device_locked(c, (const int8_t *)q8.p, cn, d_cls + off, d_logits ? d_logits + off * ncls : nullptr, nullptr, 0, s))
            return e;
    }
    c->last_kernel = "quantize_input_kernel+" + c->last_kernel;
    return BNM_OK;
}

}  // extern "C"
