/* Minimal C host for the whole-model QAT forward of libbitnetmcu_hip.so (include/bitnetmcu_hip.h).
 *
 *   gcc -std=c99 -Iinclude examples/qat_forward.c -Lbitnetmcu_amd -lbitnetmcu_hip -Wl,-rpath,$PWD/bitnetmcu_amd -o qat_forward
 *   ./qat_forward model.f32 rows.f32 > logits.txt
 *
 * model.f32 (little-endian float32 / int32 words): n_layers, widths[n_layers + 1], quant_type per layer (BNM_QAT_*), norm_type
 * (BNM_QAT_NORM_*), then per layer its clipping scalar `s` and its weights [widths[l+1]][widths[l]] - what the reference's FCMNIST
 * (models.py:56-90) holds in `layer.s` and `layer.weight`.  rows.f32: n x 256 float32 values.  Prints the float32 logits of
 * FCMNIST.forward (BitLinear.forward per layer, BitNetMCU.py:214-235, ReLU between the layers), one row per line, computed by ONE
 * kernel behind a weight-preparation launch.  Needs an MI355X.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
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

int main(int argc, char **argv) {
    if (argc != 3) {
        fprintf(stderr, "usage: %s <model.f32> <rows.f32>\n", argv[0]);
        return 2;
    }
    size_t mlen = 0, xlen = 0;
    int32_t *mi = (int32_t *)slurp(argv[1], &mlen);
    float *x = (float *)slurp(argv[2], &xlen);
    if (!mi || !x || xlen % 1024 != 0 || mlen < 16) {
        fprintf(stderr, "cannot read inputs (rows.f32 must hold n x 256 float32 values)\n");
        return 2;
    }
    const uint32_t nl = (uint32_t)mi[0];
    if (nl < 2 || nl > BNM_QAT_MODEL_MAX_LAYERS) {
        fprintf(stderr, "2 .. %d layers\n", BNM_QAT_MODEL_MAX_LAYERS);
        return 2;
    }
    uint32_t widths[BNM_QAT_MODEL_MAX_LAYERS + 1], s_count[BNM_QAT_MODEL_MAX_LAYERS];
    int quant[BNM_QAT_MODEL_MAX_LAYERS];
    const float *host_w[BNM_QAT_MODEL_MAX_LAYERS], *host_s[BNM_QAT_MODEL_MAX_LAYERS];
    size_t pos = 1;
    for (uint32_t l = 0; l <= nl; l++) widths[l] = (uint32_t)mi[pos++];
    for (uint32_t l = 0; l < nl; l++) quant[l] = mi[pos++];
    const int norm = mi[pos++];
    const float *mf = (const float *)mi;
    for (uint32_t l = 0; l < nl; l++) {
        host_s[l] = mf + pos;
        pos += 1;
        s_count[l] = 1;                                   /* PerTensor clipping scalars in this example */
        host_w[l] = mf + pos;
        pos += (size_t)widths[l + 1] * widths[l];
    }
    if (pos * 4 != mlen) {
        fprintf(stderr, "model.f32 holds %zu bytes, its own header describes %zu\n", mlen, pos * 4);
        return 2;
    }
    if (!bnm_qat_model_supported(nl, widths, quant, norm)) {
        fprintf(stderr, "this stack is not served by the fused kernel (run the layers with bnm_qat_bitlinear_forward_device)\n");
        return 3;
    }
    const uint64_t n = xlen / 1024;
    /* device buffers: the rows, every layer's weights and scalar, the logits, the workspace */
    void *d_x = NULL, *d_logits = NULL, *d_ws = NULL;
    const float *d_w[BNM_QAT_MODEL_MAX_LAYERS], *d_s[BNM_QAT_MODEL_MAX_LAYERS];
    CHECK(bnm_device_malloc(&d_x, xlen ? xlen : 16));
    CHECK(bnm_memcpy_h2d(d_x, x, xlen));
    for (uint32_t l = 0; l < nl; l++) {
        void *p = NULL, *q = NULL;
        const size_t wbytes = (size_t)widths[l + 1] * widths[l] * 4;
        CHECK(bnm_device_malloc(&p, wbytes));
        CHECK(bnm_memcpy_h2d(p, host_w[l], wbytes));
        CHECK(bnm_device_malloc(&q, 16));
        CHECK(bnm_memcpy_h2d(q, host_s[l], 4));
        d_w[l] = (const float *)p;
        d_s[l] = (const float *)q;
    }
    const size_t lbytes = (size_t)n * widths[nl] * 4;
    CHECK(bnm_device_malloc(&d_logits, lbytes ? lbytes : 16));
    const uint64_t wsb = bnm_qat_model_workspace_bytes(nl, widths);
    CHECK(bnm_device_malloc(&d_ws, (size_t)wsb));
    CHECK(bnm_qat_model_forward_device((const float *)d_x, n, nl, widths, d_w, d_s, s_count, quant, norm, (float *)d_logits, NULL, NULL, d_ws,
                                       wsb, NULL));
    CHECK(bnm_device_synchronize());
    float *logits = (float *)malloc(lbytes ? lbytes : 16);
    CHECK(bnm_memcpy_d2h(logits, d_logits, lbytes));
    for (uint64_t i = 0; i < n; i++) {
        for (uint32_t c = 0; c < widths[nl]; c++) printf(c ? " %.9g" : "%.9g", logits[i * widths[nl] + c]);
        printf("\n");
    }
    for (uint32_t l = 0; l < nl; l++) {
        bnm_device_free((void *)d_w[l]);
        bnm_device_free((void *)d_s[l]);
    }
    bnm_device_free(d_x);
    bnm_device_free(d_logits);
    bnm_device_free(d_ws);
    free(logits);
    free(x);
    free(mi);
    return 0;
}
