/* The reference's CNNMNIST forward (models.py:93-139, the model trainingparameters.yaml names) from a gcc-only C host: two kernels of
 * libbitnetmcu_hip.so - the convolution front (bnm_qat_cnn_front_forward_device) and the FC stack behind Flatten
 * (bnm_qat_model_forward_device) - on float images (include/bitnetmcu_hip.h).
 *
 *   gcc -std=c99 -Iinclude examples/qat_cnn_forward.c -Lbitnetmcu_amd -lbitnetmcu_hip -Wl,-rpath,$PWD/bitnetmcu_amd -o qat_cnn_forward
 *   ./qat_cnn_forward model.f32 images.f32 > logits.txt
 *
 * model.f32 (little-endian float32 / int32 words): channels (64: the FC stack then has 256 inputs), n_fc (2 .. 4 FC layers), the FC
 * widths[1 .. n_fc] (hidden widths, classes), 3 + n_fc quant types (BNM_QAT_*: conv1, conv2, conv3, then the FC layers), norm_type of
 * the FC layers (BNM_QAT_NORM_*); then per layer in that order its clipping scalar `s` and its weights (convolutions [C][9], FC layers
 * [out][in]) - what the module holds in `layer.s` / `layer.weight`.  images.f32: n x 256 float32 values (16x16).  Prints the float32
 * logits, one image per line.  Needs an MI355X.
 */
#include <stdio.h>
#include <stdlib.h>
#include "bitnetmcu_hip.h"

static void *slurp(const char *path, size_t *len) {
    FILE *f = fopen(path, "rb");
    if (!f) return NULL;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    void *buf = malloc((size_t)n + 1);
    if (buf && fread(buf, 1, (size_t)n, f) != (size_t)n) { free(buf); buf = NULL; }
    fclose(f);
    *len = (size_t)n;
    return buf;
}

#define CHECK(call)                                                          \
    do {                                                                     \
        if ((call) != BNM_OK) {                                              \
            fprintf(stderr, "%s: %s\n", #call, bnm_last_error());            \
            return 1;                                                        \
        }                                                                    \
    } while (0)

/* a host array -> a device buffer of its own */
static int upload(const float *h, size_t floats, const float **d) {
    void *p = NULL;
    if (bnm_device_malloc(&p, floats * 4 < 16 ? 16 : floats * 4) != BNM_OK || bnm_memcpy_h2d(p, h, floats * 4) != BNM_OK) return 1;
    *d = (const float *)p;
    return 0;
}

int main(int argc, char **argv) {
    if (argc != 3) {
        fprintf(stderr, "usage: %s <model.f32> <images.f32>\n", argv[0]);
        return 2;
    }
    size_t mlen = 0, xlen = 0;
    int32_t *mi = (int32_t *)slurp(argv[1], &mlen);
    float *x = (float *)slurp(argv[2], &xlen);
    if (!mi || !x || xlen % 1024 != 0 || mlen < 32) {
        fprintf(stderr, "cannot read inputs (images.f32 must hold n x 256 float32 values)\n");
        return 2;
    }
    const uint32_t channels = (uint32_t)mi[0], n_fc = (uint32_t)mi[1];
    if (4u * channels != 256u || n_fc < 2 || n_fc > BNM_QAT_MODEL_MAX_LAYERS) {
        fprintf(stderr, "this example: 64 channels (256 features), 2 .. %d FC layers\n", BNM_QAT_MODEL_MAX_LAYERS);
        return 2;
    }
    uint32_t widths[BNM_QAT_MODEL_MAX_LAYERS + 1], ones[BNM_QAT_MODEL_MAX_LAYERS] = {1, 1, 1, 1};
    int conv_q[3], fc_q[BNM_QAT_MODEL_MAX_LAYERS];
    size_t pos = 2;
    widths[0] = 4u * channels;
    for (uint32_t l = 1; l <= n_fc; l++) widths[l] = (uint32_t)mi[pos++];
    for (int l = 0; l < 3; l++) conv_q[l] = mi[pos++];
    for (uint32_t l = 0; l < n_fc; l++) fc_q[l] = mi[pos++];
    const int norm = mi[pos++];
    const float *mf = (const float *)mi;
    const float *conv_w[3], *conv_s[3], *fc_w[BNM_QAT_MODEL_MAX_LAYERS], *fc_s[BNM_QAT_MODEL_MAX_LAYERS];
    for (int l = 0; l < 3; l++) {
        if (upload(mf + pos, 1, &conv_s[l]) || upload(mf + pos + 1, (size_t)channels * 9, &conv_w[l])) return 1;
        pos += 1 + (size_t)channels * 9;
    }
    for (uint32_t l = 0; l < n_fc; l++) {
        const size_t count = (size_t)widths[l + 1] * widths[l];
        if (upload(mf + pos, 1, &fc_s[l]) || upload(mf + pos + 1, count, &fc_w[l])) return 1;
        pos += 1 + count;
    }
    if (pos * 4 != mlen) {
        fprintf(stderr, "model.f32 holds %zu bytes, its own header describes %zu\n", mlen, pos * 4);
        return 2;
    }
    if (!bnm_qat_cnn_front_supported(channels, ones, conv_q) || !bnm_qat_model_supported(n_fc, widths, fc_q, norm)) {
        fprintf(stderr, "this configuration is not served by the fused kernels (run the layers one by one)\n");
        return 3;
    }
    const uint64_t n = xlen / 1024;
    const float *d_x = NULL;
    void *d_features = NULL, *d_logits = NULL, *d_ws1 = NULL, *d_ws2 = NULL;
    if (upload(x, xlen / 4, &d_x)) return 1;
    const size_t lbytes = (size_t)n * widths[n_fc] * 4;
    const uint64_t wsb1 = bnm_qat_cnn_front_workspace_bytes(channels), wsb2 = bnm_qat_model_workspace_bytes(n_fc, widths);
    CHECK(bnm_device_malloc(&d_features, xlen ? xlen : 16));      /* 4 C = 256 floats per image */
    CHECK(bnm_device_malloc(&d_logits, lbytes ? lbytes : 16));
    CHECK(bnm_device_malloc(&d_ws1, (size_t)wsb1));
    CHECK(bnm_device_malloc(&d_ws2, (size_t)wsb2));
    /* both calls on the default stream: the FC stack's kernel finds the features the front's left in HBM */
    CHECK(bnm_qat_cnn_front_forward_device(d_x, n, channels, conv_w, conv_s, ones, conv_q, (float *)d_features, d_ws1, wsb1, NULL));
    CHECK(bnm_qat_model_forward_device((const float *)d_features, n, n_fc, widths, fc_w, fc_s, ones, fc_q, norm, (float *)d_logits, NULL, NULL,
                                       d_ws2, wsb2, NULL));
    CHECK(bnm_device_synchronize());
    float *logits = (float *)malloc(lbytes ? lbytes : 16);
    CHECK(bnm_memcpy_d2h(logits, d_logits, lbytes));
    for (uint64_t i = 0; i < n; i++) {
        for (uint32_t c = 0; c < widths[n_fc]; c++) printf(c ? " %.9g" : "%.9g", logits[i * widths[n_fc] + c]);
        printf("\n");
    }
    free(logits);
    free(x);
    free(mi);
    return 0;      /* (device buffers go with the process) */
}
