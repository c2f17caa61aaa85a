/* Minimal C host for the batched API of libbitnetmcu_hip.so.
 *
 *   gcc -std=c99 -Iinclude examples/batch_infer.c -Lbitnetmcu_amd -lbitnetmcu_hip -Wl,-rpath,$PWD/bitnetmcu_amd -o batch_infer
 *   ./batch_infer BitNetMCU_model.h images.i8        # images.i8: n x 256 int8 bytes (16x16 images, row-major)
 *
 * Prints one class id per image — the values the reference's Inference() returns for the same header and images
 * (BitNetMCU_MNIST_dll.c:123-130).  Needs an MI355X; without a HIP device bnm_ctx_create reports BNM_EHIP.
 */
#include <stdio.h>
#include <stdlib.h>
#include "bitnetmcu_hip.h"

static char *slurp(const char *path, size_t *len) {
    FILE *f = fopen(path, "rb");
    if (!f) return NULL;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    char *buf = (char *)malloc((size_t)n + 1);
    if (buf && fread(buf, 1, (size_t)n, f) != (size_t)n) { free(buf); buf = NULL; }
    fclose(f);
    if (buf) buf[n] = 0;
    *len = (size_t)n;
    return buf;
}

int main(int argc, char **argv) {
    if (argc != 3) {
        fprintf(stderr, "usage: %s <BitNetMCU_model.h> <images.i8>\n", argv[0]);
        return 2;
    }
    size_t hlen = 0, ilen = 0;
    char *header = slurp(argv[1], &hlen);
    char *images = slurp(argv[2], &ilen);
    if (!header || !images || ilen % 256 != 0) {
        fprintf(stderr, "cannot read inputs (the image file must hold n x 256 bytes)\n");
        return 2;
    }
    bnm_model *model = NULL;
    bnm_ctx *ctx = NULL;
    if (bnm_model_from_header_text(header, hlen, &model) != BNM_OK || bnm_ctx_create(model, -1, &ctx) != BNM_OK) {
        fprintf(stderr, "%s\n", bnm_last_error());
        return 1;
    }
    uint64_t n = ilen / 256;
    uint32_t *cls = (uint32_t *)malloc(n * sizeof(uint32_t));
    if (bnm_infer_host(ctx, (const int8_t *)images, n, cls, NULL) != BNM_OK) {
        fprintf(stderr, "%s\n", bnm_last_error());
        return 1;
    }
    for (uint64_t i = 0; i < n; i++) printf("%u\n", cls[i]);
    bnm_ctx_destroy(ctx);
    bnm_model_free(model);
    free(cls);
    free(images);
    free(header);
    return 0;
}
