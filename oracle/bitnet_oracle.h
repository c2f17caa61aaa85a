/*
 * oracle/bitnet_oracle.h — TEST INFRASTRUCTURE (see oracle/README.md).  Not part of the product.
 *
 * CPU restatement ("port") of the BitNetMCU ANSI-C inference path, written from the algorithm.
 * Parity status: PINNED — against the reference's own embedded images/labels, against
 * known-answer vectors generated from the compiled reference (tests/golden/), and live
 * against oracle/_ref/<model>/Bitnet_inf.dll whenever present (tests/test_oracle_*.py).
 */
#ifndef BNM_ORACLE_H
#define BNM_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Per-function restatements.  Signatures equal the reference's so tests can call either side
 * through the same ctypes prototypes.
 *   orc_processfclayer    <- BitNetMCU_inference.c:88-208   (all codecs, incl. "unknown -> 0")
 *   orc_ReLUNorm          <- BitNetMCU_inference.c:23-72
 *   orc_processconv33ReLU <- BitNetMCU_inference.c:238-277
 *   orc_processmaxpool22  <- BitNetMCU_inference.c:300-322                                    */
void     orc_processfclayer(const int8_t *activations, const uint32_t *weights,
                            int32_t bits_per_weight, uint32_t n_input, uint32_t n_output,
                            int32_t *output);
uint32_t orc_ReLUNorm(int32_t *input, int8_t *output, uint32_t n_input);
int32_t *orc_processconv33ReLU(int32_t *activations, const int8_t *weights, uint32_t xy_input,
                               uint32_t n_shift, int32_t *output);
int32_t *orc_processmaxpool22(int32_t *activations, uint32_t xy_input, int32_t *output);

/* One decoded weight w(row,k) of a packed layer (0 for an unknown codec).  Used by tests to
 * pin the GPU-side unpack kernels. */
int32_t  orc_weight_at(const void *weights, int32_t bits_per_weight, uint32_t n_input,
                       uint32_t row, uint32_t k);

/* Model-level restatements of the BitMnistInference schedules.
 *   FC  variant <- BitNetMCU_MNIST_dll.c:95-121  (any number of FC layers >= 1)
 *   CNN variant <- BitNetMCU_MNIST_dll.c:48-91   (3 depthwise conv + 2 pools per channel,
 *                                                 ReLUNorm over channels*4, then FC layers) */
typedef struct {
    int32_t bits_per_weight;
    uint32_t n_input;   /* as in the header: padded count for ternary */
    uint32_t n_output;
    const void *weights;
} orc_fc_layer;

typedef struct {
    uint32_t channels;          /* L7_out_channels */
    const int8_t *w_conv1;      /* [channels*9]  L2_weights */
    const int8_t *w_conv2;      /* [channels*9]  L4_weights */
    const int8_t *w_conv3;      /* [channels*9]  L7_weights */
    uint32_t n_shift;           /* 4 at every reference call site */
} orc_cnn_front;

/* image: 256 int8.  logits: n_output of the last layer (may be NULL).
 * acts: optional int8 activations after every ReLUNorm, concatenated (may be NULL). */
uint32_t orc_fc_model(const int8_t *image, const orc_fc_layer *layers, uint32_t n_layers,
                      int32_t *logits, int8_t *acts);
uint32_t orc_cnn_model(const int8_t *image, const orc_cnn_front *front,
                       const orc_fc_layer *layers, uint32_t n_layers, int32_t *logits,
                       int8_t *acts);
/* Batch drivers: images [n][256]; cls [n]; logits [n][n_output_last] or NULL.
 * front == NULL selects the FC schedule. */
void orc_model_batch(const int8_t *images, uint64_t n, const orc_cnn_front *front,
                     const orc_fc_layer *layers, uint32_t n_layers, uint32_t *cls,
                     int32_t *logits);

/* Synthetic images (oracle/synth.h). */
void orc_synth(uint64_t seed, int dist, uint64_t first, uint64_t count, int8_t *out);

/* Order-independent digest of (global index, class id) pairs + class histogram, the CPU side
 * of the full-size "checksum of checksums" property.
 * digest = sum over i of splitmix64((first+i) * 64 + cls[i])   (mod 2^64)                    */
uint64_t orc_class_digest(const uint32_t *cls, uint64_t first, uint64_t n, uint64_t *hist,
                          uint32_t n_bins);

#ifdef __cplusplus
}
#endif
#endif
