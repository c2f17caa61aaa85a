// C ABI of libbitnetmcu_hip.so / Bitnet_inf.dll (declared in include/bitnetmcu_hip.h): the small device utilities.  The rest of
// the ABI lives beside this file: bnm_capi_model.cpp (error state, version, the model objects: no HIP), bnm_capi_ctx.cpp (contexts, kernel
// selection, per-stream scratch), bnm_capi_infer.cpp (device-pointer inference), bnm_capi_host.cpp (host-pointer inference),
// bnm_capi_float.cpp (float inputs), bnm_capi_qat.cpp, bnm_capi_multigpu.cpp, bnm_capi_symbols.cpp (the reference's own symbols).
#include "bnm_capi_internal.hpp"

using namespace bnm_internal;

extern "C" {

int bnm_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
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

int bnm_device_malloc(void **p, size_t bytes) { HIP_TRY(hipMalloc(p, bytes)); return BNM_OK; }
int bnm_device_free(void *p) { HIP_TRY(hipFree(p)); return BNM_OK; }
int bnm_memcpy_h2d(void *d, const void *h, size_t bytes) { HIP_TRY(hipMemcpy(d, h, bytes, hipMemcpyHostToDevice)); return BNM_OK; }
int bnm_memcpy_d2h(void *h, const void *d, size_t bytes) { HIP_TRY(hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost)); return BNM_OK; }
int bnm_device_synchronize(void) { HIP_TRY(hipDeviceSynchronize()); return BNM_OK; }

}  // extern "C"
