// C ABI, QAT forward ops: argument checks in front of bnm_qat.hip.
#include "bnm_capi_internal.hpp"

using namespace bnm_internal;

extern "C" {

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

uint64_t bnm_qat_model_workspace_bytes(uint32_t n_layers, const uint32_t *widths) {
    if (!widths || n_layers < 2u || n_layers > (uint32_t)BNM_QAT_MODEL_MAX_LAYERS) return 0;
    return bnmk_qat_model_workspace_bytes(n_layers, widths);
}

int bnm_qat_model_supported(uint32_t n_layers, const uint32_t *widths, const int *quant_types, int norm_type) {
    if (!widths || !quant_types || n_layers < 2u || n_layers > (uint32_t)BNM_QAT_MODEL_MAX_LAYERS) return 0;
    return bnmk_qat_model_supported(n_layers, widths, quant_types, norm_type) ? 1 : 0;
}

int bnm_qat_model_forward_device(const float *d_x, uint64_t n, uint32_t n_layers, const uint32_t *widths, const float *const *d_w,
                                 const float *const *d_s, const uint32_t *s_count, const int *quant_types, int norm_type,
                                 float *d_logits, float *d_hidden, float *const *d_w_deq, void *d_workspace, uint64_t workspace_bytes,
                                 void *stream) {
    if (!widths || !d_w || !d_s || !s_count || !quant_types || !d_workspace || (n && (!d_x || !d_logits))) return fail(BNM_EINVAL, "null pointer");
    if (n_layers < 2u || n_layers > (uint32_t)BNM_QAT_MODEL_MAX_LAYERS) return fail(BNM_EINVAL, "need 2 <= n_layers <= BNM_QAT_MODEL_MAX_LAYERS");
    for (uint32_t l = 0; l < n_layers; l++) {
        if (!d_w[l] || !d_s[l]) return fail(BNM_EINVAL, "null weight or clipping-scalar pointer");
        if (quant_types[l] < BNM_QAT_NONE || quant_types[l] > BNM_QAT_8BIT) return fail(BNM_EINVAL, "unknown quant_type");
        if (widths[l + 1] == 0u) return fail(BNM_EINVAL, "zero width");
        if (s_count[l] != 1u && s_count[l] != widths[l + 1]) return fail(BNM_EINVAL, "s_count must be 1 (PerTensor) or the layer's outputs (PerOutput)");
    }
    if (norm_type < BNM_QAT_NORM_RMS || norm_type > BNM_QAT_NORM_NONE) return fail(BNM_EINVAL, "unknown norm_type");
    if (!bnmk_qat_model_supported(n_layers, widths, quant_types, norm_type))
        return fail(BNM_EUNSUPPORTED, "the fused model forward serves <= 256 inputs, hidden widths <= 192, <= 64 classes, int8-level QuantTypes "
                                      "and NormType RMS / Lin / LayerNorm (bnm_qat_model_supported); run the layers with bnm_qat_bitlinear_forward_device");
    if (workspace_bytes < bnmk_qat_model_workspace_bytes(n_layers, widths)) return fail(BNM_EINVAL, "workspace too small (bnm_qat_model_workspace_bytes)");
    if (((uintptr_t)d_workspace & 15u) || ((uintptr_t)d_logits & 15u) || ((uintptr_t)d_x & 15u))
        return fail(BNM_EINVAL, "d_x, d_logits and the workspace must be 16-byte aligned");
    if (n >= (1ull << 36)) return fail(BNM_EINVAL, "n too large for one call");
    HIP_TRY(bnmk_qat_model_forward(d_x, n, n_layers, widths, d_w, d_s, s_count, quant_types, norm_type, d_logits, d_hidden, d_w_deq,
                                   d_workspace, (hipStream_t)stream));
    return BNM_OK;
}

int bnm_qat_cnn_front_supported(uint32_t channels, const uint32_t *s_count, const int *quant_types) {
    if (!s_count || !quant_types) return 0;
    return bnmk_qat_cnn_front_supported(channels, s_count, quant_types) ? 1 : 0;
}

uint64_t bnm_qat_cnn_front_workspace_bytes(uint32_t channels) { return bnmk_qat_cnn_front_workspace_bytes(channels); }

int bnm_qat_cnn_front_forward_device(const float *d_x, uint64_t n, uint32_t channels, const float *const *d_w, const float *const *d_s,
                                     const uint32_t *s_count, const int *quant_types, float *d_features, void *d_workspace,
                                     uint64_t workspace_bytes, void *stream) {
    return bnm_qat_cnn_front_forward_train_device(d_x, n, channels, d_w, d_s, s_count, quant_types, d_features, nullptr, nullptr, nullptr,
                                                  d_workspace, workspace_bytes, stream);
}

int bnm_qat_cnn_front_forward_train_device(const float *d_x, uint64_t n, uint32_t channels, const float *const *d_w, const float *const *d_s,
                                           const uint32_t *s_count, const int *quant_types, float *d_features, float *d_y1, float *d_y2,
                                           float *d_y3, void *d_workspace, uint64_t workspace_bytes, void *stream) {
    if (!d_w || !d_s || !s_count || !quant_types || !d_workspace || (n && (!d_x || !d_features))) return fail(BNM_EINVAL, "null pointer");
    if ((d_y1 != nullptr) != (d_y2 != nullptr) || (d_y1 != nullptr) != (d_y3 != nullptr)) return fail(BNM_EINVAL, "d_y1, d_y2 and d_y3: all three or none");
    if (((uintptr_t)d_y1 | (uintptr_t)d_y2 | (uintptr_t)d_y3) & 15u) return fail(BNM_EINVAL, "d_y1, d_y2 and d_y3 must be 16-byte aligned");
    for (int l = 0; l < 3; l++) {
        if (!d_w[l] || !d_s[l]) return fail(BNM_EINVAL, "null weight or clipping-scalar pointer");
        if (quant_types[l] < BNM_QAT_NONE || quant_types[l] > BNM_QAT_8BIT) return fail(BNM_EINVAL, "unknown quant_type");
    }
    if (!bnmk_qat_cnn_front_supported(channels, s_count, quant_types))
        return fail(BNM_EUNSUPPORTED, "the fused convolution front serves an even number of 16 .. 128 channels with per-tensor clipping scalars and a QuantType other than 'None' "
                                      "(bnm_qat_cnn_front_supported); run the layers with bnm_qat_bitconv2d_forward_device");
    if (workspace_bytes < bnmk_qat_cnn_front_workspace_bytes(channels)) return fail(BNM_EINVAL, "workspace too small (bnm_qat_cnn_front_workspace_bytes)");
    if (((uintptr_t)d_workspace & 15u) || ((uintptr_t)d_features & 15u) || ((uintptr_t)d_x & 15u))
        return fail(BNM_EINVAL, "d_x, d_features and the workspace must be 16-byte aligned");
    if (n >= (1ull << 36)) return fail(BNM_EINVAL, "n too large for one call");
    HIP_TRY(bnmk_qat_cnn_front_forward(d_x, n, channels, d_w, d_s, quant_types, d_features, d_y1, d_y2, d_y3, d_workspace, (hipStream_t)stream));
    return BNM_OK;
}

}  // extern "C"
