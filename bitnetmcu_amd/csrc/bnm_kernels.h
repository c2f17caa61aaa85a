// Launch wrappers for the gfx950 kernels in bnm_*.hip.  Everything here takes DEVICE
// pointers and a hipStream_t and is asynchronous.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

// ---- synthetic workload / digest -------------------------------------------------------------
hipError_t bnmk_synth_fill(int8_t *d_images, uint64_t first, uint64_t count, uint64_t seed, int dist,
                           hipStream_t s);
hipError_t bnmk_class_digest(const uint32_t *d_cls, uint64_t first, uint64_t n, uint64_t *d_out,
                             uint32_t n_bins, hipStream_t s);

// plain 16 B/lane nontemporal read of `bytes` bytes (a multiple of 16 is read); d_sink: one device dword that is practically never written
hipError_t bnmk_stream_read(const void *d_src, uint64_t bytes, uint32_t *d_sink, hipStream_t s);
// mixed stream: 32-row tiles of 256-byte rows read, out_bytes_per_row (a multiple of 4... x 32 rows = whole 16-byte units) written per row
hipError_t bnmk_stream_rw(const void *d_src, uint64_t n_rows, void *d_dst, uint32_t out_bytes_per_row, uint32_t mode, hipStream_t s);

// ---- GPU unpack: packed words -> int8 rows -> MFMA A-operand fragments ------------------------
// lo/hi: [n_output][row_stride] int8, w = lo + hi (hi != 0 only for FP1.3.0's +-128).
hipError_t bnmk_unpack_rows(const void *d_packed, int32_t bpw, uint32_t n_input, uint32_t n_real,
                            uint32_t n_output, int8_t *d_lo, int8_t *d_hi, uint32_t row_stride,
                            hipStream_t s);
// kmap 0: natural K order (layer fed by the raw image); 1: K order of the previous layer's packed
// ReLUNorm output (see DESIGN.md §fragment layout).  dst: MT*KT fragments of 64 lanes x 16 B.
// pad_row_weight: weight given to rows >= n_output on the real input columns — 0 for hidden layers (ReLUNorm must
// see 0 there), -128 for the classifier layer (argmax_rows in bnm_fused_fc.hip relies on it).
// frag_stride: bytes between the fragments of consecutive K-steps of one tile (1024 = contiguous; the generic kernel's K-step
// major layout passes M KiB with MT == 1)
hipError_t bnmk_build_fragments(const int8_t *d_rows, uint32_t row_stride, uint32_t n_output,
                                uint32_t n_real, uint32_t MT, uint32_t KT, int kmap, int scale, int pad_row_weight,
                                void *d_dst, uint32_t frag_stride, hipStream_t s);

// ---- fused whole-model FC kernel (int8 MFMA) -------------------------------------------------
struct BnmFusedShape {
    int KT0;        // input row bytes / 32
    int M[4];       // 32-row tiles of each FC layer's outputs; M[3] == 0 for 3-layer models
    bool split;     // FP1.3.0: two A passes per K-step
    bool dbl;       // hidden-layer weight fragments doubled (every codec except 8-bit and FP1.3.0)
    int nc8;        // ceil(n_classes / 8); kernels specialised on it skip accumulator registers without class rows
    bool operator==(const BnmFusedShape &o) const {
        return KT0 == o.KT0 && M[0] == o.M[0] && M[1] == o.M[1] && M[2] == o.M[2] && M[3] == o.M[3] &&
               split == o.split && dbl == o.dbl && nc8 == o.nc8;
    }
};
struct BnmFusedArgs {
    const int8_t *images;   // [n][32*KT0]
    uint64_t n;
    const void *frags;      // fragment buffer built by bnmk_build_fragments, layers concatenated
    uint32_t n_classes;
    uint32_t *cls;          // [n]
    int32_t *logits;        // [n][n_classes] or nullptr
    uint64_t src_wrap = 0;  // diagnostic library only (BNM_DIAG): read tile (t mod src_wrap); ignored by the product build
    uint32_t *work = nullptr;   // variant 6: the launch's counter block (BNM_WORK_BLOCK_WORDS words, all zero between launches)
    uint32_t *idle = nullptr;   // variant 6: device words [16 * BNM_WORK_DUMMY_WAVES], one per wave, for the zero-adds (shared by all launches)
    uint32_t batch = 0;         // variant 6: pairs per take (0 = default)
};
constexpr uint32_t BNM_WORK_DUMMY_WAVES = 4096;
// A counter block: counter words at [16 k], k < 8 (64 bytes apart), the leave count at [BNM_WORK_EXIT_WORD].  Owned by one stream
// (bnm_capi_ctx.cpp); all zero between launches - the last wave of a launch to leave puts it back (bnm_device.hpp, work_block_leave_*),
// so no launch is preceded by a memset.
constexpr uint32_t BNM_WORK_BLOCK_WORDS = 256;
constexpr uint32_t BNM_WORK_EXIT_WORD = 128;
constexpr uint32_t BNM_DUAL_DEFAULT_BATCH = 2;     // pairs per take of the dual-tile kernel (profiles/headline_ab.py sweeps)
// variant: 0 = direct global->VGPR image loads, 1 = LDS-DMA staged (256-byte rows only), 2 = LDS-DMA with two
// tiles in flight per wave, 3 = two tiles computed per wave per iteration (default where instantiated)
bool bnmk_fused_supported(const BnmFusedShape &sh, int variant);
hipError_t bnmk_fused_fc(const BnmFusedShape &sh, int variant, int grid_blocks, const BnmFusedArgs &a,
                         hipStream_t s);
int bnmk_fused_default_variant(const BnmFusedShape &sh);
// variant 9 (bnm_fused_regw.hip): weights resident in the register file at one wave per SIMD, for the 3..5-tile shapes.  Takes
// whole 64-image pairs only (n % 64 == 0): the caller (bnm_capi_infer.cpp, run_fused) gives the remainder to the generic kernel.
constexpr int BNM_FUSED_REGW = 9;
bool bnmk_regw_supported(const BnmFusedShape &sh);
hipError_t bnmk_fused_regw(const BnmFusedShape &sh, int grid_blocks, const BnmFusedArgs &a, hipStream_t s);

// ---- generic fused whole-model FC kernel (bnm_fused_generic.hip): run-time layer widths, weights in LDS ----------
struct BnmGenericDesc {      // passed to the kernel by value
    uint32_t KT0;            // input row bytes / 32: 2, 4, 8 or 16 (rows of 64 / 128 / 256 / 512 bytes)
    uint32_t mmax;           // tile class: 2, 4, 6 or 8 = upper bound of 32-row tiles per layer the kernel is compiled for
    uint32_t M[4];           // 32-row output tiles per FC layer (exact); M[3] == 0 for 3-layer models
    uint32_t KTP[4];         // K-steps per layer (exact): KT0 for layer 1, the previous layer's tile count for the others
    uint32_t frag_off[4];    // byte offset of each layer's fragments inside the fragment image
    uint32_t w_bytes;        // size of the fragment image (a multiple of 1 KiB): sp * sum M[i] * KTP[i] KiB, nothing padded
    uint32_t sp;             // 1, or 2 when a second weight plane follows each layer's first (FP1.3.0's +128)
    uint32_t n_classes;      // <= 256
    uint32_t stage;          // set by the launcher: a 2 KiB logits staging area per wave follows the tile buffers in LDS
};
// Fragment (plane p, K-step s, tile m) of layer i sits at frag_off[i] + ((p * KTP[i] + s) * M[i] + m) KiB of the image.
// waves per SIMD an instantiation of the generic kernel is compiled for (its launch bound is 256 * this many threads);
// tiles: image tiles a wave carries per iteration (1 or 2)
constexpr int bnmk_generic_wps(int mmax, int kt0, int sp, int tiles) {
    return mmax == 2 ? (tiles == 2 || kt0 == 16 ? 3 : 4) : mmax == 4 ? ((tiles == 2 || (sp == 2 && kt0 == 16)) ? 2 : 3) : mmax == 6 ? 2 : 1;
}
// variant ids of the generic kernel in bnm_ctx_set_tuning / bnm_ctx_get_variant: 4 = tiles per wave chosen by the library,
// 7 / 8 = one / two tiles per wave forced (A/B measurements)
enum { BNM_FUSED_GENERIC = 4, BNM_FUSED_GENERIC_T1 = 7, BNM_FUSED_GENERIC_T2 = 8 };
bool bnmk_generic_plan(BnmGenericDesc &d, const uint32_t m_real[4]);   // fills mmax, M, KTP, frag_off, w_bytes from KT0, sp
// tiles: 0 = the library's choice, 1 / 2 forced; logits: the call writes logits (a staging area is reserved when it fits).
// Returns the tiles per wave a launch would use (1 or 2), or 0 when the model does not fit the kernel in that form.
int bnmk_generic_tiles(const BnmGenericDesc &d, bool dbl, int tiles, bool logits);
bool bnmk_generic_supported(const BnmGenericDesc &d, bool dbl);
uint32_t bnmk_generic_resident_waves(const BnmGenericDesc &d, bool dbl);   // waves per CU of the default launch (0: does not fit)
// d_counter: a counter block of the caller (BNM_WORK_BLOCK_WORDS words, all zero; the kernel leaves it all zero); batch: units per take (0 = default)
hipError_t bnmk_fused_generic(const BnmGenericDesc &d, bool dbl, int tiles, int grid_blocks, const int8_t *d_images, uint64_t n,
                              const void *d_frags, uint32_t *d_cls, int32_t *d_logits, uint32_t *d_counter, uint32_t batch,
                              hipStream_t s);

// ---- fused float-input whole-model FC kernel (bnm_fused_f32.hip): float32 [n][256] -> input quantisation -> the FC stack, one
// kernel, on the generic kernel's descriptor and fragment image.  Serves 256-value rows (KT0 == 8) of the 2- and 4-tile classes
// whose weights leave LDS for at least one wave per SIMD.  groups: 8-image groups in flight per wave (0 = default; 2 or 4).
bool bnmk_fused_f32_supported(const BnmGenericDesc &d, bool dbl, int groups);
hipError_t bnmk_fused_f32(const BnmGenericDesc &d, bool dbl, int groups, int grid_blocks, const float *d_x, uint64_t n, const void *d_frags,
                          uint32_t *d_cls, int32_t *d_logits, uint32_t *d_counter, uint32_t batch, unsigned long long *d_nonfinite, hipStream_t s);

// the resident single-wave kernel behind Inference() (bnm_persist_kernel.hpp; opt-in): box = the device address of a page-locked
// mailbox of 128 dwords, seq0 = the last sequence number already answered, idle_ticks of the 100 MHz wall clock without a call end it
bool bnmk_persistent_supported(const BnmGenericDesc &d, bool dbl);
hipError_t bnmk_persistent_launch(const BnmGenericDesc &d, bool dbl, const void *d_frags, uint32_t *box, uint32_t seq0, uint64_t idle_ticks,
                                  hipStream_t s);

// ---- layer-wise ALU kernels (bit-serial unpack + wave-shuffle reduction) ----------------------
hipError_t bnmk_fc_layer(const int8_t *d_act, uint32_t act_stride, const void *d_packed, int32_t bpw,
                         uint32_t n_input, uint32_t n_output, int32_t *d_out, uint64_t batch,
                         hipStream_t s);
// the same layer on the matrix cores from the unpacked int8 rows (any width; see bnm_layerwise.hip)
hipError_t bnmk_fc_layer_mfma(const int8_t *d_act, uint32_t act_stride, const int8_t *d_rows_lo, const int8_t *d_rows_hi,
                              uint32_t row_stride, uint32_t n_output, int32_t *d_out, uint64_t batch, hipStream_t s);
hipError_t bnmk_relunorm(const int32_t *d_in, uint32_t n, int8_t *d_out, uint32_t out_stride,
                         uint32_t *d_argmax, uint64_t batch, hipStream_t s);
hipError_t bnmk_conv33(const int32_t *d_in, const int8_t *d_w, uint32_t xy, uint32_t n_shift,
                       int32_t *d_out, hipStream_t s);
hipError_t bnmk_maxpool22(const int32_t *d_in, uint32_t xy, int32_t *d_out, hipStream_t s);

// ---- CNN front end: 3 depthwise 3x3 convs + 2 pools per channel, batched ----------------------
// images [n][256] int8 -> acts [n][4*C] int8 after the fused ReLUNorm (channel-major,
// BitNetMCU_MNIST_dll.c:65,76,80); feat (optional) = the int32 values before ReLUNorm
// acts_stride: bytes between consecutive images' act rows (>= 4*C; bytes past 4*C are left untouched)
// d_wtab: the per-channel weight table of the conv1-on-MFMA kernel (bnm_cnn_weight_table, uploaded by the caller) — or
// nullptr to run the all-VALU kernel of round 1 (kept for A/B measurements, bnm_ctx_set_cnn_variant)
// d_counter / grab: the MFMA kernel's waves take batches of `grab` images from word 0 of this counter block (all zero on entry,
// left all zero); nullptr / 0: a fixed share per wave
// Channel segments (MFMA kernel): a model's last <= 16 channels beyond a multiple of 32 (C <= 16; 33..48; 65..80; ...) run TWO
// images per item.  The whole front end is one fused launch over image pairs; beyond 64 channels the last one or two images of
// a call take the segment path (channel segments that each write d_feat, ReLUNorm as its own kernel afterwards).  d_feat:
// REQUIRED for C > 64 - [n][4C] ints when feat_is_output is set (it then holds every image's int32 features), else scratch of
// [2][4C] ints; optional output for C <= 64.  (The all-VALU kernel, d_wtab == nullptr, needs [n][4C] for C > 64.)
hipError_t bnmk_cnn_front(const int8_t *d_images, uint64_t n, const int8_t *d_w1, const int8_t *d_w2,
                          const int8_t *d_w3, const int *d_wtab, uint32_t C, uint32_t n_shift, int8_t *d_acts,
                          uint32_t acts_stride, int32_t *d_feat, bool feat_is_output, uint32_t *d_counter, uint32_t grab,
                          hipStream_t s);
// lane = image formulation (bnm_cnn_li.hip): all three convolutions as Toeplitz products on the matrix cores, ReLUNorm fused, any
// channel count whose records fit the LDS beside six waves (C <= 170), n_shift 4.  frags / bias: bnm_cnn_li_tables' output on the
// device (6 KiB + 8 bytes per channel); counter: the launch's counter block.
// returns whether conv3 needs its third operand plane (a pooled conv2 output of 2^16 or more is possible with these weights): the
// launchers take the answer as `plane2`; *sums16 (optional): no conv1 sum of any channel can exceed 65535 with these weights - what
// the pipelined one-kernel form (cnn_li_fused_pipe_kernel) requires
bool bnm_cnn_li_tables(const int8_t *w1, const int8_t *w2, const int8_t *w3, uint32_t C, int8_t *frag_out, int *bias_out, bool *sums16 = nullptr);
uint32_t bnmk_cnn_li_waves(uint32_t C);      // waves per workgroup; 0: the kernel does not serve this channel count
hipError_t bnmk_cnn_front_li(const int8_t *d_images, uint64_t n, const void *d_frags, const int *d_bias, uint32_t C, bool plane2, int8_t *d_acts,
                             uint32_t acts_stride, uint32_t *d_counter, uint32_t grab, hipStream_t s);
// the lane = image front end with the FC tail fused into the same wave (bnm_cnn_li_fused.hip): images -> class ids (+ logits), one
// launch, no act rows in HBM.  tail_frags / d: the generic kernel's fragment image and descriptor of the model's FC tail (the
// image must be followed by 16 KiB of readable padding: fragment reads run a few KiB ahead of the last fragment).
bool bnmk_cnn_li_fused_supported(uint32_t C, const BnmGenericDesc &d);
// float_images: d_images is float32 [n][256], quantised in front of the operands (test_inference.py:140-141) instead of int8 [n][256]
// pipe: the pipelined three-waves-per-SIMD form (only for models whose bnm_cnn_li_tables said sums16)
hipError_t bnmk_cnn_li_fused(const void *d_images, bool float_images, uint64_t n, const void *d_frags, const int *d_bias, uint32_t C, bool plane2, bool pipe,
                             const void *d_tail_frags,
                             const BnmGenericDesc &d, bool dbl, uint32_t *d_cls, int32_t *d_logits, uint32_t *d_counter, uint32_t grab,
                             unsigned long long *d_nonfinite, hipStream_t s);
constexpr int BNM_CNN_WTAB_DWORDS = 20;      // per (band, channel); 2 bands x C rounded up to 64 channels
void bnm_cnn_weight_table(const int8_t *w1, const int8_t *w2, const int8_t *w3, uint32_t C, int *out);

// ---- ternary ALU whole-model kernel (sign-accumulate, no MFMA) -------------------------------
struct BnmTernArgs {
    const int8_t *images;  // [n][256]
    uint64_t n;
    const int8_t *rows[4];  // unpacked trits per layer [n_out][stride]
    uint32_t stride[4];
    uint32_t n_in[4];       // real inputs (256, 96, 96, 96)
    uint32_t n_out[4];
    uint32_t n_layers;
    uint32_t *cls;
    int32_t *logits;
    const int *wstream;     // bnmk_ternary_stream_build's output (variants 1, 2)
    int variant;            // 0: round 1's kernel; 1: streamed weights, one image per lane; 2: two images per lane
    uint32_t *counter;      // counter block (word 0 hands the image groups out; all zero on entry, left all zero); nullptr: fixed stride
};
// shapes of the ALU path: 256 real inputs, hidden widths out of the instantiation table (bnm_ternary.hip); the streamed kernel
// exists with one image per lane (variant 1) for every shape of the table, with two per lane (variant 2) for 96-96-96
bool bnmk_ternary_alu_supported(const uint32_t n_in[4], const uint32_t n_out[4]);
bool bnmk_ternary_stream_supported(const uint32_t n_out[4], int images_per_lane);
uint32_t bnmk_ternary_stream_dwords(const uint32_t n_out[4]);
hipError_t bnmk_ternary_stream_build(const BnmTernArgs &a, int *d_stream, hipStream_t s);
hipError_t bnmk_ternary_alu(const BnmTernArgs &a, int grid_blocks, hipStream_t s);

// ---- diagnostics (bnm_diag.hip, diagnostic library only): cost of the image stream alone, pipe overlap ----
#ifdef BNM_DIAG_TIMING
hipError_t bnmk_diag_cnn_set_record(uint64_t *d_rec);   // 5 x uint64 per wave of cnn_front_mfma_kernel
#endif
#ifdef BNM_DIAG
hipError_t bnmk_diag_stream(const int8_t *d_images, uint64_t n, int mode, int grid_blocks, uint32_t *d_out, hipStream_t s);
hipError_t bnmk_diag_pipes(int mode, uint64_t tiles_per_wave, uint32_t *d_out, hipStream_t s);   // modes 5/6/7: pipe overlap probe
#endif

// ---- input quantisation: float32 [n][256] -> int8 [n][256] (test_inference.py:140-141) --------------
// nonfinite (optional): += the images that held a NaN or an infinity (they quantise to all zeros)
hipError_t bnmk_quantize_input(const float *d_x, uint64_t n, int8_t *d_out, unsigned long long *d_nonfinite, hipStream_t s);

// ---- QAT forward op (bnm_qat.hip) ---------------------------------------
size_t bnmk_qat_workspace_bytes(uint32_t d, uint32_t k);
hipError_t bnmk_qat_bitlinear_forward(const float *d_x, uint64_t n, uint32_t d, const float *d_w, uint32_t k,
                                      const float *d_s, uint32_t s_count, int quant_type, int norm_type, float *d_y,
                                      float *d_workspace, float *d_x_int_out, float *d_x_scale_out, float *d_w_deq_out,
                                      hipStream_t s);
// BitConv2d forward: any groups / stride, zero padding `pad`, PerTensor clipping scalar.
// whole-model QAT forward (bnm_qat_model.hip): widths[0 .. n_layers]; workspace bnmk_qat_model_workspace_bytes (0: shape not served)
size_t bnmk_qat_model_workspace_bytes(uint32_t n_layers, const uint32_t *widths);
bool bnmk_qat_model_supported(uint32_t n_layers, const uint32_t *widths, const int *quant_types, int norm_type);
hipError_t bnmk_qat_model_forward(const float *d_x, uint64_t n, uint32_t n_layers, const uint32_t *widths, const float *const *d_w,
                                  const float *const *d_s, const uint32_t *s_count, const int *quant_types, int norm_type,
                                  float *d_logits, float *d_hidden, float *const *d_w_deq, void *d_workspace, hipStream_t stream);
// CNNMNIST's convolution front in one kernel (bnm_qat_cnn.hip): w / s / quant_types are three-element HOST arrays (conv1, conv2, conv3)
bool bnmk_qat_cnn_front_supported(uint32_t channels, const uint32_t *s_count, const int *quant_types);
size_t bnmk_qat_cnn_front_workspace_bytes(uint32_t channels);
// d_y1 / d_y2 / d_y3: all null, or the training form's planes, CHANNELS-LAST: [n][14][14][C] / [n][12][12][C] / [n][4][4][C]
hipError_t bnmk_qat_cnn_front_forward(const float *d_x, uint64_t n, uint32_t channels, const float *const *d_w, const float *const *d_s,
                                      const int *quant_types, float *d_features, float *d_y1, float *d_y2, float *d_y3, void *d_workspace,
                                      hipStream_t stream);
// workspace: bnmk_qat_workspace_bytes((cin / groups) * kh * kw, cout) bytes; dynamic LDS: bnmk_qat_bitconv2d_lds_bytes (<= 160 KiB).
size_t bnmk_qat_bitconv2d_lds_bytes(uint32_t cin, uint32_t h, uint32_t w, uint32_t cout, uint32_t kh, uint32_t kw, uint32_t pad,
                                    uint32_t groups);
hipError_t bnmk_qat_bitconv2d_forward(const float *d_x, uint64_t n, uint32_t cin, uint32_t h, uint32_t w, const float *d_w,
                                      uint32_t cout, uint32_t kh, uint32_t kw, uint32_t pad, uint32_t stride, uint32_t groups,
                                      const float *d_s, int quant_type, int norm_type, float *d_y, float *d_workspace, hipStream_t s);
