/*
 * oracle/bitnet_oracle.c — TEST INFRASTRUCTURE (see oracle/README.md).  Not part of the product.
 *
 * CPU restatement of the BitNetMCU inference path.  Structure is "decode one weight, then MAC",
 * which is not how the reference is written (it shifts a weight word through a per-codec inner
 * loop) but is provably the same integer function: every partial product is an exact int32 and
 * |sum| <= 256*128*128 < 2^31, so the order of accumulation is immaterial.
 * Parity status: PINNED (tests/test_oracle_vs_ref.py, tests/test_oracle_golden.py).
 */
#include "bitnet_oracle.h"
#include "synth.h"
#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------------------------------------
 * Weight codecs.  Codec ids are the header's Lk_bitperweight values (exportquant.py:104-177):
 *   1 Binary, 2 2bitsym, 4 4bitsym, 12 4-bit two's complement, 16 8-bit two's complement,
 *   20 FP1.3.0, 64 Ternary (10 trits / uint16).  Anything else (e.g. 36 = NF4) has no branch in
 *   the reference (BitNetMCU_inference.c:96-202) and therefore yields output 0.
 * 32-bit codecs: fields are packed MSB-first, rows are contiguous, n_input*bits is a multiple
 * of 32 (exportquant.py:97-98, 182-187).
 * ------------------------------------------------------------------------------------------- */
static int codec_field_bits(int32_t bpw) {
    switch (bpw) {
        case 1: return 1;
        case 2: return 2;
        case 4: case 12: case 20: return 4;
        case 16: return 8;
        default: return 0;
    }
}

/* Ternary: BitNetMCU_inference.c:116-136.  A uint16 holds ceil(v*65536/59049) where v is the
 * base-3, most-significant-trit-first value of 10 trits (exportquant.py:139-157).  Multiplying
 * the 16-bit fraction by 3 pops the next trit into bits 17:16: 0 -> +1, 1 -> -1, 2 -> 0. */
static int32_t ternary_weight(const uint16_t *h, uint32_t n_input, uint32_t row, uint32_t k) {
    uint32_t chunk = h[row * (n_input / 10u) + k / 10u];
    uint32_t digit = 0;
    for (uint32_t t = 0; t <= k % 10u; t++) {
        chunk *= 3u;
        digit = chunk >> 16;
        chunk &= 0xFFFFu;
    }
    return digit == 0 ? 1 : (digit == 1 ? -1 : 0);
}

int32_t orc_weight_at(const void *weights, int32_t bpw, uint32_t n_input, uint32_t row,
                      uint32_t k) {
    if (bpw == 64) return ternary_weight((const uint16_t *)weights, n_input, row, k);
    int fb = codec_field_bits(bpw);
    if (!fb) return 0;
    uint32_t per_word = 32u / (uint32_t)fb;
    uint32_t words_per_row = (n_input + per_word - 1u) / per_word;
    uint32_t word = ((const uint32_t *)weights)[row * words_per_row + k / per_word];
    uint32_t f = (word >> (32u - (uint32_t)fb * (k % per_word + 1u))) & ((1u << fb) - 1u);
    switch (bpw) {
        case 1:  /* :96-104  bit set => +1, clear => -1 */
            return f ? 1 : -1;
        case 2:  /* :105-115 sign bit, then one magnitude bit: +-1, +-3 */
            return ((f & 2u) ? -1 : 1) * (int32_t)(1u + 2u * (f & 1u));
        case 4:  /* :156-168 sign bit + 3 magnitude bits m: +-(2m+1) */
            return ((f & 8u) ? -1 : 1) * (int32_t)(2u * (f & 7u) + 1u);
        case 12: /* :169-178 4-bit two's complement */
            return (int32_t)(f ^ 8u) - 8;
        case 16: /* :179-188 8-bit two's complement */
            return (int32_t)(int8_t)f;
        case 20: /* :190-201 sign bit + 3-bit exponent: +-2^e */
            return ((f & 8u) ? -1 : 1) * (int32_t)(1u << (f & 7u));
    }
    return 0;
}

/* Decode a whole layer once: W[row][k], zero for an unknown codec.  (int16: FP1.3.0 reaches +128.) */
static void decode_layer(const void *weights, int32_t bpw, uint32_t n_input, uint32_t n_output, int16_t *W) {
    for (uint32_t row = 0; row < n_output; row++)
        for (uint32_t k = 0; k < n_input; k++)
            W[(size_t)row * n_input + k] = (int16_t)orc_weight_at(weights, bpw, n_input, row, k);
}

/* sum_k act[k] * W[row][k]; weights of value 0 never touch their activation (ternary pads, :128-131) */
static void fc_dense(const int8_t *act, const int16_t *W, uint32_t n_input, uint32_t n_act, uint32_t n_output,
                     int32_t *output) {
    for (uint32_t row = 0; row < n_output; row++) {
        const int16_t *w = W + (size_t)row * n_input;
        int32_t sum = 0;
        for (uint32_t k = 0; k < n_act; k++) sum += (int32_t)w[k] * (int32_t)act[k];
        output[row] = sum;
    }
}

/* highest k + 1 with a non-zero weight: the activations a layer can touch */
static uint32_t used_inputs(const int16_t *W, uint32_t n_input, uint32_t n_output) {
    uint32_t used = 0;
    for (uint32_t row = 0; row < n_output; row++)
        for (uint32_t k = n_input; k > used; k--)
            if (W[(size_t)row * n_input + k - 1]) { used = k; break; }
    return used;
}

/* BitNetMCU_inference.c:88-208.  Activations past the last non-zero ternary trit are never
 * dereferenced in the reference (:128-131); the same holds here. */
void orc_processfclayer(const int8_t *activations, const uint32_t *weights, int32_t bpw,
                        uint32_t n_input, uint32_t n_output, int32_t *output) {
    int16_t *W = (int16_t *)malloc((size_t)n_input * n_output * sizeof(int16_t) + 2);
    decode_layer(weights, bpw, n_input, n_output, W);
    fc_dense(activations, W, n_input, used_inputs(W, n_input, n_output), n_output, output);
    free(W);
}

/* BitNetMCU_inference.c:23-72.  First strict maximum (:32-37); shift = number of significant
 * bits of max above bit 6 (:41-47); round half up (:51,:59); negative -> 0 (:56); clip to 127
 * (:62-66).  For max <= 0 every output is 0 whatever the (then ill-defined, :51) shift is.
 * Reads input[i] completely before writing output[i], ascending, so the int32 -> int8 in-place
 * use of BitNetMCU_MNIST_dll.c:80 is preserved. */
uint32_t orc_ReLUNorm(int32_t *input, int8_t *output, uint32_t n) {
    int32_t top = -INT32_MAX;
    uint32_t pos = 255;
    for (uint32_t i = 0; i < n; i++)
        if (input[i] > top) { top = input[i]; pos = i; }
    uint32_t shift = 0;
    if (top > 0)
        for (uint32_t s = (uint32_t)top >> 7; s; s >>= 1) shift++;
    int32_t half = shift ? (int32_t)(1u << (shift - 1u)) : 0;
    for (uint32_t i = 0; i < n; i++) {
        int32_t v = input[i];
        int32_t q = 0;
        if (v >= 0 && top > 0) {
            q = (v + half) >> shift;
            if (q > 127) q = 127;
        }
        output[i] = (int8_t)q;
    }
    return pos;
}

/* BitNetMCU_inference.c:238-277: valid 3x3, single channel, ReLU then arithmetic >> n_shift,
 * no rounding, no clip; output may alias input (write index never passes read index). */
int32_t *orc_processconv33ReLU(int32_t *act, const int8_t *w_in, uint32_t xy, uint32_t n_shift,
                               int32_t *out) {
    int32_t w[9];
    for (int t = 0; t < 9; t++) w[t] = w_in[t];
    uint32_t o = xy - 2u;
    for (uint32_t y = 0; y < o; y++)
        for (uint32_t x = 0; x < o; x++) {
            int32_t s = 0;
            for (uint32_t dy = 0; dy < 3; dy++)
                for (uint32_t dx = 0; dx < 3; dx++)
                    s += w[3 * dy + dx] * act[(y + dy) * xy + x + dx];
            *out++ = s < 0 ? 0 : (s >> n_shift);
        }
    return out;
}

/* BitNetMCU_inference.c:300-322 */
int32_t *orc_processmaxpool22(int32_t *act, uint32_t xy, int32_t *out) {
    uint32_t o = xy / 2u;
    for (uint32_t y = 0; y < o; y++)
        for (uint32_t x = 0; x < o; x++) {
            const int32_t *p = act + 2u * y * xy + 2u * x;
            int32_t m = p[0];
            if (p[xy] > m) m = p[xy];
            if (p[1] > m) m = p[1];
            if (p[xy + 1] > m) m = p[xy + 1];
            *out++ = m;
        }
    return out;
}

#define ORC_MAX_ACT 1024

/* FC tail shared by both schedules: fc -> ReLUNorm, repeated; the class id is the argmax
 * returned by the last ReLUNorm (BitNetMCU_MNIST_dll.c:99-120 / :83-90).  `dec` (optional): the layers'
 * weights already decoded by decode_model(), so a batch pays for the unpacking once. */
typedef struct { int16_t *W; uint32_t used; } dec_layer;

static uint32_t run_fc_chain(int8_t *act, const orc_fc_layer *L, uint32_t n_layers, const dec_layer *dec,
                             int32_t *logits, int8_t *acts_out) {
    int32_t acc[ORC_MAX_ACT];
    uint32_t cls = 255;
    for (uint32_t l = 0; l < n_layers; l++) {
        if (dec) fc_dense(act, dec[l].W, L[l].n_input, dec[l].used, L[l].n_output, acc);
        else orc_processfclayer(act, (const uint32_t *)L[l].weights, L[l].bits_per_weight, L[l].n_input, L[l].n_output, acc);
        if (l + 1 == n_layers && logits)
            memcpy(logits, acc, sizeof(int32_t) * L[l].n_output);
        cls = orc_ReLUNorm(acc, act, L[l].n_output);
        if (acts_out) { memcpy(acts_out, act, L[l].n_output); acts_out += L[l].n_output; }
    }
    return cls;
}

static dec_layer *decode_model(const orc_fc_layer *L, uint32_t n_layers) {
    dec_layer *d = (dec_layer *)calloc(n_layers, sizeof(dec_layer));
    for (uint32_t l = 0; l < n_layers; l++) {
        d[l].W = (int16_t *)malloc((size_t)L[l].n_input * L[l].n_output * sizeof(int16_t) + 2);
        decode_layer(L[l].weights, L[l].bits_per_weight, L[l].n_input, L[l].n_output, d[l].W);
        d[l].used = used_inputs(d[l].W, L[l].n_input, L[l].n_output);
    }
    return d;
}

static void free_model(dec_layer *d, uint32_t n_layers) {
    for (uint32_t l = 0; l < n_layers; l++) free(d[l].W);
    free(d);
}

static uint32_t fc_model_dec(const int8_t *image, const orc_fc_layer *L, uint32_t n_layers, const dec_layer *dec,
                             int32_t *logits, int8_t *acts) {
    int8_t act[ORC_MAX_ACT];
    memset(act, 0, sizeof act);
    memcpy(act, image, 256);
    return run_fc_chain(act, L, n_layers, dec, logits, acts);
}

uint32_t orc_fc_model(const int8_t *image, const orc_fc_layer *L, uint32_t n_layers,
                      int32_t *logits, int8_t *acts) {
    return fc_model_dec(image, L, n_layers, 0, logits, acts);
}

/* BitNetMCU_MNIST_dll.c:48-91.  Per channel: widen the image to int32 (:68-70), conv 16->14,
 * conv 14->12, pool ->6, conv 6->4, pool ->2 appended channel-major (:71-76); one ReLUNorm over
 * all channels*4 values (:80); FC layers. */
static uint32_t cnn_model_dec(const int8_t *image, const orc_cnn_front *F, const orc_fc_layer *L,
                              uint32_t n_layers, const dec_layer *dec, int32_t *logits, int8_t *acts) {
    int32_t plane[256];
    int32_t feat[ORC_MAX_ACT];
    int8_t act[ORC_MAX_ACT];
    int32_t *dst = feat;
    for (uint32_t c = 0; c < F->channels; c++) {
        for (int i = 0; i < 256; i++) plane[i] = image[i];
        orc_processconv33ReLU(plane, F->w_conv1 + 9u * c, 16, F->n_shift, plane);
        orc_processconv33ReLU(plane, F->w_conv2 + 9u * c, 14, F->n_shift, plane);
        orc_processmaxpool22(plane, 12, plane);
        orc_processconv33ReLU(plane, F->w_conv3 + 9u * c, 6, F->n_shift, plane);
        dst = orc_processmaxpool22(plane, 4, dst);
    }
    memset(act, 0, sizeof act);
    orc_ReLUNorm(feat, act, F->channels * 4u);
    if (acts) { memcpy(acts, act, F->channels * 4u); acts += F->channels * 4u; }
    return run_fc_chain(act, L, n_layers, dec, logits, acts);
}

uint32_t orc_cnn_model(const int8_t *image, const orc_cnn_front *F, const orc_fc_layer *L,
                       uint32_t n_layers, int32_t *logits, int8_t *acts) {
    return cnn_model_dec(image, F, L, n_layers, 0, logits, acts);
}

void orc_model_batch(const int8_t *images, uint64_t n, const orc_cnn_front *F,
                     const orc_fc_layer *L, uint32_t n_layers, uint32_t *cls, int32_t *logits) {
    uint32_t n_out = L[n_layers - 1].n_output;
    dec_layer *dec = decode_model(L, n_layers);
    for (uint64_t i = 0; i < n; i++) {
        int32_t *lg = logits ? logits + i * n_out : 0;
        cls[i] = F ? cnn_model_dec(images + 256 * i, F, L, n_layers, dec, lg, 0)
                   : fc_model_dec(images + 256 * i, L, n_layers, dec, lg, 0);
    }
    free_model(dec, n_layers);
}

void orc_synth(uint64_t seed, int dist, uint64_t first, uint64_t count, int8_t *out) {
    orc_synth_images(seed, dist, first, count, out);
}

uint64_t orc_class_digest(const uint32_t *cls, uint64_t first, uint64_t n, uint64_t *hist,
                          uint32_t n_bins) {
    uint64_t d = 0;
    for (uint64_t i = 0; i < n; i++) {
        d += orc_splitmix64((first + i) * 64ull + cls[i]);
        if (hist && cls[i] < n_bins) hist[cls[i]]++;
    }
    return d;
}
