/* C host that drives the reference's kernel symbols ONE BY ONE on the GPU library - the way a program written against
 * BitNetMCU_inference.h does (the reference's own such program is BitNetMCU_MNIST_test.c; it #includes BitNetMCU_inference.c
 * textually, this one links the same symbols from libbitnetmcu_hip.so instead).
 *
 *   gcc -std=c99 -I<dir with BitNetMCU_model.h and BitNetMCU_MNIST_test_data.h> -Iinclude examples/mnist_test.c \
 *       -Lbitnetmcu_amd -lbitnetmcu_hip -Wl,-rpath,$PWD/bitnetmcu_amd -o mnist_test && ./mnist_test
 *
 * BitNetMCU_model.h: the exporter's header (its L<k>_* macros and weight arrays, BitNetMCU_model_fc.h / _cnn.h dialect);
 * BitNetMCU_MNIST_test_data.h: `int8_t input_data_<k>[256]` / `uint8_t label_<k>` for k = 0..9 (the reference's layout).
 * Output: one line per image, "label: <l> predicted: <p>" - what BitNetMCU_MNIST_test.c:17-40 prints.
 * Every call is a round trip to the GPU: this is symbol-level compatibility, not the fast path (examples/batch_infer.c).
 * BNM_TIME_REPEATS=<n> in the environment: afterwards the ten images are classified n more times and the mean time per image
 * goes to stderr (profiles/r04_symbol_flow.py; DESIGN.md 6 has the numbers).
 */
#define _POSIX_C_SOURCE 199309L
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "BitNetMCU_model.h"
#include "BitNetMCU_MNIST_test_data.h"
#include "bitnetmcu_hip.h"

typedef struct {
    const void *weights;
    int32_t codec;
    uint32_t n_in, n_out;
} fc_layer;

#define FC(k) {L##k##_weights, L##k##_bitperweight, L##k##_incoming_weights, L##k##_outgoing_weights}

#if defined(MODEL_CNNMNIST)
/* per channel: conv (1 -> C), depthwise conv + pool, depthwise conv + pool; then three fully connected layers */
typedef struct {
    const int8_t *kernels;   /* [channels][9] */
    uint32_t side;           /* side of the square input plane */
    uint32_t pool_side;      /* side of the plane the 2x2 pool behind this conv reads; 0: no pool */
} conv_stage;
static const conv_stage front[] = {{L2_weights, L2_incoming_x, 0}, {L4_weights, L4_incoming_x, L6_incoming_x}, {L7_weights, L7_incoming_x, L9_incoming_x}};
enum { N_STAGES = sizeof front / sizeof front[0], CHANNELS = L7_out_channels, CONV_SHIFT = 4 };
static const fc_layer fc_chain[] = {FC(11), FC(13), FC(15)};
#elif defined(MODEL_FCMNIST)
#ifdef L4_active
static const fc_layer fc_chain[] = {FC(1), FC(2), FC(3), FC(4)};
#else
static const fc_layer fc_chain[] = {FC(1), FC(2), FC(3)};
#endif
#else
#error "BitNetMCU_model.h defines neither MODEL_FCMNIST nor MODEL_CNNMNIST"
#endif

static uint32_t classify(const int8_t *image) {
    static int32_t sums[4 * MAX_N_ACTIVATIONS];
    static int8_t act[4 * MAX_N_ACTIVATIONS];
    uint32_t winner = 0;
#if defined(MODEL_CNNMNIST)
    static int32_t plane[16 * 16];
    int32_t *features = sums, *next = sums;
    for (uint32_t c = 0; c < CHANNELS; c++) {
        for (int i = 0; i < 256; i++) plane[i] = image[i];
        /* every stage works in place on the channel's plane; the last pool appends its 2x2 block to the feature row */
        for (int k = 0; k < N_STAGES; k++) {
            processconv33ReLU(plane, front[k].kernels + 9 * c, front[k].side, CONV_SHIFT, plane);
            if (front[k].pool_side && k + 1 < N_STAGES) processmaxpool22(plane, front[k].pool_side, plane);
            else if (front[k].pool_side) next = processmaxpool22(plane, front[k].pool_side, next);
        }
    }
    ReLUNorm(features, act, (uint32_t)(next - features));
#else
    memcpy(act, image, 256);
#endif
    for (size_t k = 0; k < sizeof fc_chain / sizeof fc_chain[0]; k++) {
        const fc_layer *l = &fc_chain[k];
        processfclayer(act, (const uint32_t *)l->weights, l->codec, l->n_in, l->n_out, sums);
        winner = ReLUNorm(sums, act, l->n_out);
    }
    return winner;
}

int main(void) {
#define IMAGE(k) {input_data_##k, &label_##k}
    static const struct {
        const int8_t *pixels;
        const uint8_t *label;
    } tests[] = {IMAGE(0), IMAGE(1), IMAGE(2), IMAGE(3), IMAGE(4), IMAGE(5), IMAGE(6), IMAGE(7), IMAGE(8), IMAGE(9)};
    for (size_t t = 0; t < sizeof tests / sizeof tests[0]; t++)
        printf("label: %d predicted: %d\n", (int)*tests[t].label, (int)classify(tests[t].pixels));
    const char *rep = getenv("BNM_TIME_REPEATS");
    if (rep && atoi(rep) > 0) {
        const int n = atoi(rep), images = (int)(sizeof tests / sizeof tests[0]);
        struct timespec t0, t1;
        uint32_t sink = 0;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        for (int r = 0; r < n; r++)
            for (int t = 0; t < images; t++) sink += classify(tests[t].pixels);
        clock_gettime(CLOCK_MONOTONIC, &t1);
        const double us = ((double)(t1.tv_sec - t0.tv_sec) * 1e9 + (double)(t1.tv_nsec - t0.tv_nsec)) / 1e3 / ((double)n * images);
        fprintf(stderr, "symbol flow: %.1f us per image over %d images (checksum %u)\n", us, n * images, (unsigned)sink);
    }
    return 0;
}
