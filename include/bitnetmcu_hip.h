/*
 * bitnetmcu_hip.h — C ABI of the MI355X-native BitNetMCU inference path.
 *
 * Two groups of entry points:
 *
 *  (A) The reference's own symbols, kept bit-for-bit compatible so that a library built from
 *      this repository is a drop-in `Bitnet_inf.dll` for test_inference.py:134-150 and for any
 *      code that links the reference's C functions directly.  All pointers are HOST pointers,
 *      buffers are caller-owned, nothing is retained, there is no init/teardown call and no
 *      error channel (the reference has none; an unsupported codec yields zeros,
 *      BitNetMCU_inference.c:202).  GPU context and model upload happen lazily on first use.
 *      If no HIP device is usable these functions abort() with a message on stderr — there is
 *      deliberately NO CPU fallback.
 *
 *  (B) Additive entry points (prefix bnm_): run-time model loading from header text, batched
 *      host- and device-pointer inference, synthetic workload generation, digests.  They return
 *      0 on success or a negative BNM_E* code; bnm_last_error() gives the message.
 */
#ifndef BITNETMCU_HIP_H
#define BITNETMCU_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(_WIN32)
#define BNM_API __declspec(dllexport)
#else
#define BNM_API __attribute__((visibility("default")))
#endif

/* =============================== (A) reference ABI ========================================= */

/* Replaces BitNetMCU_MNIST_dll.c:24-26 (`EXPORT uint32_t Inference(int8_t *input)`), the symbol
 * test_inference.py:146-150 binds with argtypes=[POINTER(c_int8)], restype=c_uint32.
 * input: 256 int8 (16x16 row-major, already quantised by the caller).  Returns the class id.
 * Only present in a model-bound build (`Bitnet_inf.dll`) or after
 * bnm_bind_default_model(). */
BNM_API uint32_t Inference(int8_t *input);

/* Replaces BitNetMCU_MNIST_dll.c:48 (CNN) / :95 (FC) `uint32_t BitMnistInference(int8_t*)`. */
BNM_API uint32_t BitMnistInference(int8_t *input);

/* Replaces BitNetMCU_inference.c:88 / BitNetMCU_inference.h:31.  Same argument meaning:
 * activations int8[n_input], weights packed as in BitNetMCU_model.h (uint32 words, or uint16
 * words reinterpreted for bits_per_weight == 64), bits_per_weight = the header's codec id
 * (1,2,4,12,16,20,64; anything else => output zeros), output int32[n_output]. */
BNM_API void processfclayer(int8_t *activations, const uint32_t *weights, int32_t bits_per_weight,
                            uint32_t n_input, uint32_t n_output, int32_t *output);

/* Replaces BitNetMCU_inference.c:23 / BitNetMCU_inference.h:15.  input int32[n_input] ->
 * output int8[n_input] in [0,127]; returns the index of the first maximum.  output may alias
 * input (the int32 -> int8 in-place use of BitNetMCU_MNIST_dll.c:80). */
BNM_API uint32_t ReLUNorm(int32_t *input, int8_t *output, uint32_t n_input);

/* Replaces BitNetMCU_inference.c:238 / BitNetMCU_inference.h:45.  Single channel valid 3x3
 * convolution + ReLU + >> n_shift; output may alias activations; returns output + (xy-2)^2. */
BNM_API int32_t *processconv33ReLU(int32_t *activations, const int8_t *weights, uint32_t xy_input,
                                   uint32_t n_shift, int32_t *output);

/* Replaces BitNetMCU_inference.c:300 / BitNetMCU_inference.h:60.  2x2/stride-2 max pool;
 * output may alias activations; returns output + (xy/2)^2. */
BNM_API int32_t *processmaxpool22(int32_t *activations, uint32_t xy_input, int32_t *output);

/* =============================== (B) additive ABI ========================================== */

#define BNM_OK 0
#define BNM_EINVAL (-1)      /* bad argument */
#define BNM_EPARSE (-2)      /* model header text / blob not understood */
#define BNM_EUNSUPPORTED (-3) /* model topology outside what the reference wrapper can run */
#define BNM_EHIP (-4)        /* HIP runtime error (no device, launch failure, ...) */
#define BNM_ENOMODEL (-5)    /* Inference() called with no bound model */

typedef struct bnm_model bnm_model;   /* host-side parsed model (no GPU needed) */
typedef struct bnm_ctx bnm_ctx;       /* a model resident on one GPU */

BNM_API const char *bnm_last_error(void);     /* thread-local message of the last failure */
BNM_API const char *bnm_version(void);

/* ---- model: the BitNetMCU_model.h interchange format (exportquant.py:49-263) -------------- */

/* Parse exporter-written header TEXT at run time.  Layers are discovered in order of
 * appearance, not by fixed names; all three textual variants in the reference
 * tree are accepted. */
BNM_API int bnm_model_from_header_text(const char *text, size_t len, bnm_model **out);
/* Compact binary form of the same information ("BNMBLOB1"): what rank 0 broadcasts. */
BNM_API int bnm_model_from_blob(const void *blob, size_t len, bnm_model **out);
BNM_API size_t bnm_model_blob_size(const bnm_model *m);
BNM_API int bnm_model_to_blob(const bnm_model *m, void *dst, size_t cap);
BNM_API void bnm_model_free(bnm_model *m);

#define BNM_KIND_FC 0
#define BNM_KIND_CNN 1
#define BNM_LAYER_FC 1
#define BNM_LAYER_CONV 2
#define BNM_LAYER_POOL 3

typedef struct {
    uint32_t type;            /* BNM_LAYER_* */
    uint32_t order;           /* k of the header's Lk_ prefix */
    int32_t bits_per_weight;  /* Lk_bitperweight */
    uint32_t n_input;         /* FC: Lk_incoming_weights (padded count for ternary) */
    uint32_t n_output;        /* FC: Lk_outgoing_weights */
    uint32_t in_channels, out_channels, groups, kernel_size;   /* conv */
    uint32_t incoming_x, outgoing_x;                           /* conv / pool */
    uint32_t pool_size;                                        /* pool */
    uint32_t weight_elem_bytes;  /* 4 (uint32), 2 (uint16 ternary), 1 (int8 conv), 0 (pool) */
    uint32_t weight_count;       /* number of array elements */
} bnm_layer_info;

BNM_API uint32_t bnm_model_kind(const bnm_model *m);        /* BNM_KIND_* */
BNM_API uint32_t bnm_model_num_layers(const bnm_model *m);
BNM_API uint32_t bnm_model_num_classes(const bnm_model *m); /* n_output of the last FC layer */
BNM_API uint32_t bnm_model_input_bytes(const bnm_model *m); /* 256 */
BNM_API int bnm_model_layer(const bnm_model *m, uint32_t i, bnm_layer_info *info);
BNM_API const void *bnm_model_layer_weights(const bnm_model *m, uint32_t i); /* as in the header */

/* ---- context: model uploaded + unpacked on one GPU ---------------------------------------- */

/* device < 0: current HIP device.  Uploads the packed weights and runs the GPU unpack kernels
 * (packed words -> int8 rows -> MFMA operand fragments).  Every entry point that takes a context works on the context's
 * device and leaves the calling thread's current HIP device as it found it; the group (A) symbols run on the device that was
 * current at their first use.  A context may be shared by host threads (calls on ONE context are serialised by a mutex inside;
 * the reference's entry points - Inference(), the four kernel symbols - lease contexts / staging slots from pools and run
 * concurrently from concurrent host threads, as the reference's stateless functions do). */
BNM_API int bnm_ctx_create(const bnm_model *m, int device, bnm_ctx **out);
BNM_API void bnm_ctx_destroy(bnm_ctx *c);
BNM_API int bnm_ctx_device(const bnm_ctx *c);

/* Kernel selection for the whole-model path. */
#define BNM_PATH_AUTO 0      /* the fastest bit-exact kernel: fused MFMA wherever the model fits it (ternary included), else layer-wise */
#define BNM_PATH_FUSED_MFMA 1
#define BNM_PATH_LAYERWISE_ALU 2   /* one bit-serial kernel per layer (the reference's structure) */
#define BNM_PATH_TERNARY_ALU 3     /* fused sign-accumulate kernel, no MFMA (ternary models) */
#define BNM_PATH_LAYERWISE_MFMA 4  /* one int8 GEMM kernel per layer on the matrix cores + ReLUNorm kernel: ANY widths (what AUTO
                                    * falls back to for models outside the fused kernels) */
BNM_API int bnm_ctx_set_path(bnm_ctx *c, int path);
BNM_API int bnm_ctx_get_path(const bnm_ctx *c);   /* the path AUTO resolved to */
/* Tuning knobs of the fused kernel: variant id (0 direct loads, 1 LDS-DMA, 2 LDS-DMA with two tiles in flight,
 * 3 two tiles per wavefront per iteration with a fixed stride, 4 the generic kernel (any widths), 5 as 3 with the CU's waves
 * sharing an LDS work counter, 6 as 3 with batches from the device-wide work counter (default where instantiated), 7 / 8 the
 * generic kernel with one / two image tiles per wave forced, 9 the register-resident-weight kernel of the 96..128-wide shapes
 * (one wave per SIMD; opt-in); -1 keeps the current one) and grid size
 * (workgroups; 0 = default).  BNM_EUNSUPPORTED if the model's shape has no such instantiation. */
BNM_API int bnm_ctx_set_tuning(bnm_ctx *c, int variant, int grid_blocks);
/* the fused-kernel variant in use, or -1 when the resolved path is not the fused kernel */
BNM_API int bnm_ctx_get_variant(const bnm_ctx *c);

/* Whole-model batched inference, DEVICE pointers, asynchronous on `stream` (a hipStream_t;
 * NULL = default stream).  images: int8 [n][256]; cls: uint32 [n]; logits: int32
 * [n][num_classes] or NULL.  No host synchronisation is performed on the fused FC paths.
 * ANY number of launches of one context may be in flight on ANY number of streams: the work counters of the persistent kernels
 * and the scratch of the CNN and layer-wise paths (feature rows, activation buffers) are kept per stream, and a kernel leaves
 * its counters zeroed for the stream's next launch - no memset precedes a launch, so a fused-FC call is exactly one dispatch.
 * The first call on a stream, and a call with a larger batch than any before on it, allocates or grows that scratch
 * (hipMalloc / hipFree: device-synchronising); a context that has seen 32 streams drops what it keeps for the others
 * (device-synchronising; bnm_ctx_release_stream does it for one stream explicitly).  Apart from that a call enqueues stream
 * work only, so after one eager call on the capturing stream it may be captured into a HIP graph and replayed on any stream,
 * next to eager launches: a captured launch gets counters of its own (up to 256 captured launches per context).  On the CNN and
 * layer-wise paths a captured launch uses the CAPTURING stream's scratch (nothing can be allocated under capture): do not run
 * eager launches of the context on the capturing stream while its graph replays on another stream.  Scratch that a captured
 * launch touched stays at its address until the context is destroyed: it is not evicted, bnm_ctx_release_stream refuses the
 * stream, and an eager call that would need it larger fails with BNM_EUNSUPPORTED (capture the largest call, or use another stream).
 * Every class-id word is written exactly once per call (never a placeholder first): a host that maps d_cls in
 * page-locked memory may poll a pre-set sentinel, as bnm_infer_host does for n <= 64. */
BNM_API int bnm_infer_device(bnm_ctx *c, const int8_t *d_images, uint64_t n, uint32_t *d_cls,
                             int32_t *d_logits, void *stream);
/* Drop what the context keeps for `stream` (scratch buffers, counter block) - call it before destroying a stream the context
 * was used on.  Synchronises the stream. */
BNM_API int bnm_ctx_release_stream(bnm_ctx *c, void *stream);
/* One-image calls (bnm_infer_host with n = 1 and no logits: what the drop-in Inference() symbol runs) through a RESIDENT single-wave
 * kernel instead of a launch per call: the image and the class id travel through a page-locked mailbox, a call costs a PCIe round trip
 * and the model's arithmetic, not a kernel launch.  mode 1 on / 0 off (a running kernel leaves before the call returns); default: off
 * unless the environment says BNM_PERSISTENT=1 when the context meets its first one-image call.  The kernel leaves by itself after
 * idle_us microseconds without a call (0 = keep: default 5000, or BNM_PERSISTENT_IDLE_US) - a hipDeviceSynchronize() anywhere in the
 * process waits at most that long - and the next call starts another one.  Same results as every other path.  BNM_EUNSUPPORTED for
 * models it does not serve (CNNs, inputs other than 256 bytes, layers wider than 192): their one-image calls stay launches. */
BNM_API int bnm_ctx_set_persistent(bnm_ctx *c, int mode, uint32_t idle_us);
/* What the resident kernel's last call took INSIDE the wave, request seen -> answer stored: ticks of the 100 MHz wall clock and shader
 * clocks (their ratio is the shader clock the wave ran at; measured: 1.44 us at 2.4 GHz for the 64-64-64 model). */
BNM_API int bnm_ctx_persistent_last_call(bnm_ctx *c, uint32_t *wall_10ns, uint32_t *shader_clocks);
/* Same with HOST pointers; synchronous.  Up to 64 images: zero-copy (page-locked buffers the GPU addresses directly, one launch,
 * results polled in place) — the path behind Inference().  Larger batches: two page-locked staging slots on two streams,
 * host copy threads, H2D / compute / D2H of consecutive chunks overlapped. */
BNM_API int bnm_infer_host(bnm_ctx *c, const int8_t *images, uint64_t n, uint32_t *cls,
                           int32_t *logits);
/* CNN front end.  3 (the default up to 170 channels): the lane = image kernel - a wave owns 32 images and walks the channels, all
 * three convolutions are Toeplitz products on the matrix cores, the ReLUNorm is fused; serves up to 170 channels
 * (BNM_EUNSUPPORTED beyond); 300 + g (g = 1..16): g 32-image tiles per take from the work counter.  Where the model's FC tail fits
 * (act row of at most 256 bytes = 64 channels, FC layers at most 96 wide, no FP1.3.0 +128: every CNN of the reference's zoo) the
 * SAME wave also runs the tail - one kernel from the image bytes to the class id, nothing but 256 + 4 bytes per image through HBM
 * (bnm_ctx_cnn_tail_fused says so); 4 / 400 + g: the lane = image kernel with the tail as its own launch over act rows in per-stream
 * scratch (what every other model gets; kept selectable for A/B measurements); 5: as 3 with conv3's third operand plane kept although
 * the model's weights rule it out (bnm_ctx_cnn_planes; A/B measurements); 6: as 3 in the four-waves-per-SIMD form
 * (cnn_li_fused_kernel) although the model's weights allow the pipelined form (cnn_li_fused_pipe_kernel: three waves per SIMD that
 * never wait for a matrix-core result, conv1's ReLU and packing in one v_cvt_pk_i16_i32 per pair - the form kernel 3 takes for every
 * model whose conv1 sums stay below 2^16, i.e. every CNN of the reference's zoo; bnm_ctx_last_kernel names the form).  1 (the default beyond 170
 * channels): a lane = a channel, conv1 on the matrix cores, waves take batches of 8 images from a device-wide work counter; models
 * whose channel count leaves 1..16 channels beyond a multiple of 32 (16, 48, 80 ... channels) run those channels two images per
 * work item.  2 the same kernel with a fixed share of the images per wave; 100 + g (g = 1..64): batches of g images; 0 the
 * all-VALU kernel of round 1 (the non-default values are kept for A/B measurements).
 * AUTO (a context nobody has called this function on): calls of fewer than 2 C^2 images (C = channels; one-image Inference() calls
 * among them) go to kernel 1, larger ones to kernel 3 - a wave of the lane = image kernel walks all channels of its 32 images,
 * 2 us per channel, however few images the call has.  bnm_ctx_get_cnn_variant returns the SETTING (3 for such a context);
 * bnm_ctx_last_kernel names what the last call really ran.  An explicit choice holds for every call size. */
BNM_API int bnm_ctx_set_cnn_variant(bnm_ctx *c, int variant);
BNM_API int bnm_ctx_get_cnn_variant(const bnm_ctx *c);      /* the setting: 3, 1 or 0 */
/* Operand planes of conv3 in the lane = image kernels: 2 when the model's conv1 / conv2 weights bound every pooled conv2 output below
 * 2^16 (the third plane - bits 16..23 - would be all zero: it is then not in the kernel; every CNN of the reference's zoo), else 3;
 * 0 for FC models and CNNs the lane = image kernels do not serve. */
BNM_API int bnm_ctx_cnn_planes(const bnm_ctx *c);
BNM_API int bnm_ctx_cnn_tail_fused(const bnm_ctx *c);       /* 1: calls that take the lane = image front end run the one-kernel form */
/* 1: ... and that form is the pipelined one (cnn_li_fused_pipe_kernel: the model's conv1 sums stay below 2^16 and nobody chose variant 6) */
BNM_API int bnm_ctx_cnn_pipelined(const bnm_ctx *c);
/* The kernels the context's LAST inference call launched (bnm_infer_device / _host / _float_device; the first chunk's of a call
 * that runs in chunks), by name and in launch order, joined by '+': e.g. "fused_fc_dual_kernel", "fused_fc_dual_kernel+fused_fc_kernel"
 * (a remainder of fewer than 64 images), "cnn_li_kernel+fused_fc_kernel", "cnn_front_mfma_kernel+fused_fc_kernel" (an AUTO context's
 * small call), "fused_fc_f32_kernel", "quantize_input_kernel+fused_fc_dual_kernel".  Empty before the first call.  The pointer
 * stays valid until the calling thread's next call of this function. */
BNM_API const char *bnm_ctx_last_kernel(bnm_ctx *c);
/* Fused kernels that hand their work out from the device-wide counter: units of one or two 32-image tiles (generic kernel,
 * variants 4 / 7 / 8; default 4 / 4 / 2) or 64-image pairs (dual-tile kernel, variant 6; default 2) a wave takes at a time;
 * 0 = default. */
BNM_API int bnm_ctx_set_work_batch(bnm_ctx *c, int tiles);
/* Ternary ALU kernels (BNM_PATH_TERNARY_ALU; ternary FC models 256-H1-H2-H3-N with H1, H2 in {32, 64, 96, 128}, H3 a multiple of
 * 16 up to 128, N <= 64).  2: weights streamed through double-buffered scalar registers, two images per lane, image groups handed
 * out from a device-wide work counter (96-96-96 only, the default there; BNM_EUNSUPPORTED for the other shapes); 1: the same with
 * one image per lane (every shape of the family; the default for the shapes other than 96-96-96); 12 / 11: as 2 / 1 with a fixed
 * stride per wave; 0 round 1's plain ALU kernel (96-96-96, 128-128-112, 64-64-64, 128-128-128 only; the non-default values are
 * kept for A/B measurements). */
BNM_API int bnm_ctx_set_ternary_variant(bnm_ctx *c, int variant);
/* Tuning of the host-pointer path.  mode 0 (default): pipelined page-locked staging; 1: the HIP runtime's own pageable copies,
 * chunk by chunk.  copy_threads: host threads of the staging copy (0 = default).  spin: poll the page-locked result words of the
 * <= 64-image path (default 1) instead of waiting for the stream. */
BNM_API int bnm_ctx_set_host_tuning(bnm_ctx *c, int mode, int copy_threads, int spin);
/* Debug/parity tap: int8 activations after every ReLUNorm of the FC chain, concatenated per
 * image (host pointers, layer-wise path). */
BNM_API int bnm_infer_host_activations(bnm_ctx *c, const int8_t *images, uint64_t n,
                                       int8_t *acts, uint32_t acts_stride);

/* Batched single-layer entry points, DEVICE pointers (the building blocks behind group A). */
BNM_API int bnm_fc_layer_device(const int8_t *d_act, uint32_t act_stride, const void *d_weights,
                                int32_t bits_per_weight, uint32_t n_input, uint32_t n_output,
                                int32_t *d_out, uint64_t batch, void *stream);
BNM_API int bnm_relunorm_device(const int32_t *d_in, uint32_t n, int8_t *d_out,
                                uint32_t out_stride, uint32_t *d_argmax, uint64_t batch,
                                void *stream);
/* GPU unpack of one packed layer to int8 rows [n_output][row_stride] (+128 for FP130 is split
 * into lo/hi parts: w = lo + hi).  Used by tests to pin the unpack kernels. */
BNM_API int bnm_unpack_layer_host(const void *weights, int32_t bits_per_weight, uint32_t n_input,
                                  uint32_t n_output, int8_t *lo, int8_t *hi, uint32_t row_stride);

/* Input quantisation on the GPU — the step the reference does in Python before every Inference() call
 * (test_inference.py:140-141; BitNetMCU.py:435-436): scale = 127/max(max|x|,1e-5), round half to even, clip.
 * d_x: float32 [n][256], d_out: int8 [n][256]; bit-identical to the numpy float32 formula for finite inputs.  Non-finite values
 * are outside the reference's contract (numpy's result for them is platform-defined): an image that holds a NaN or an infinity
 * quantises to ALL ZEROS - what the reference's expression yields on x86 - in this kernel and in the fused float-input kernels alike,
 * and is COUNTED: the _counted form adds the number of such images to *d_nonfinite (a device uint64 the caller zeroed; NULL = do not
 * count), contexts keep the count themselves (bnm_ctx_float_nonfinite). */
BNM_API int bnm_quantize_input_device(const float *d_x, uint64_t n, int8_t *d_out, void *stream);
BNM_API int bnm_quantize_input_counted_device(const float *d_x, uint64_t n, int8_t *d_out, uint64_t *d_nonfinite, void *stream);
/* The two steps of the reference's per-image Python flow (test_inference.py:140-150: quantise, then Inference()) for a batch of
 * float images resident on the GPU: d_x float32 [n][256] -> class ids (and the int32 logits if d_logits != NULL), asynchronous on
 * `stream`; the quantised images live in per-stream scratch of the context (allocated on first use: not under stream capture). */
BNM_API int bnm_infer_float_device(bnm_ctx *c, const float *d_x, uint64_t n, uint32_t *d_cls, int32_t *d_logits, void *stream);
/* How bnm_infer_float_device runs.  ONE kernel that reads the floats, quantises them in registers and feeds the matrix-core operands
 * (1,028 bytes of HBM traffic per image; exactly one dispatch, nothing allocated: capturable) for
 *   - FC models on the fused MFMA path whose layers are at most 192 wide (tile classes 2, 4 and 6 of the generic kernel: every FC
 *     model of the reference's zoo, the documented 160-160-160 binary one included): fused_fc_f32_kernel;
 *   - CNN models the one-kernel CNN form serves (up to 64 channels, FC layers at most 96 wide: every CNN of the zoo), for the calls
 *     that take it (bnm_ctx_set_cnn_variant: not the small calls of a context left to itself): cnn_li_fused_pipe_kernel / cnn_li_fused_kernel in their float form;
 * everything else runs bnm_quantize_input_device into per-stream scratch followed by the model's kernels (1,540 bytes per image).
 * mode 0 (default): one kernel where it exists; 1: one kernel or BNM_EUNSUPPORTED; 2: always two kernels (A/B measurements).
 * groups (FC kernel only): 8-image groups of floats in flight per wave: 0 = default (4 in the 2-tile class, 2 in the 4-tile class, 1 in
 * the 6-tile class), 1, 2 or 4 where instantiated.  Results are identical in every mode. */
BNM_API int bnm_ctx_set_float_mode(bnm_ctx *c, int mode, int groups);
/* Float calls say when their input was out of contract: the number of images, over all bnm_infer_float_device calls of this context so
 * far, that held a NaN or an infinity (each was classified as the all-zero image: see bnm_quantize_input_device).  Synchronises the
 * device.  0 on a context that only ever saw finite images. */
BNM_API int bnm_ctx_float_nonfinite(bnm_ctx *c, uint64_t *count);
BNM_API int bnm_ctx_float_fused(const bnm_ctx *c);      /* 1: float calls of this context run one kernel now (CNN: those that take the one-kernel form); 0: two kernels */

/* ---- QAT forward op ------------------------------------------------------
 * The forward pass of the reference's training layer BitLinear (BitNetMCU.py:198-235):
 *   y = F.linear(act_quant(Normalize(x)), weight_quant(w))        x [n][d], w [k][d], y [n][k], float32
 * quant_type / norm_type name the reference's QuantType / NormType strings.  s: the layer's clipping scalar
 * `self.s` (BitNetMCU.py:51; s_count 1 = PerTensor, k = PerOutput).  Floating point: parity with PyTorch is
 * within tolerance (integer sums are exact on the fp32 matrix cores; see csrc/bnm_qat.hip), not bit-exact.
 * x_int_out [n][d] / x_scale_out [n] (optional) return activation_quant's integers and scales, w_deq_out [k][d]
 * (optional) the fake-quantised weights w_int / w_scale — what a straight-through backward pass multiplies by.
 * All pointers are DEVICE pointers; workspace must hold bnm_qat_workspace_bytes(d, k) bytes; d <= 1024. */
#define BNM_QAT_NONE 0        /* 'None': no fake quantisation, plain linear on the normalised input */
#define BNM_QAT_BINARY 1      /* 'Binary' */
#define BNM_QAT_BINARYSYM 2   /* 'BinarySym' */
#define BNM_QAT_TERNARY 3     /* 'Ternary' */
#define BNM_QAT_2BITSYM 4     /* '2bitsym' */
#define BNM_QAT_4BIT 5        /* '4bit' */
#define BNM_QAT_4BITSYM 6     /* '4bitsym' */
#define BNM_QAT_FP130 7       /* 'FP130' */
#define BNM_QAT_NF4 8         /* 'NF4' */
#define BNM_QAT_5BITSYM 9     /* '5bitsym' */
#define BNM_QAT_8BIT 10       /* '8bit' */
#define BNM_QAT_NORM_RMS 0        /* 'RMS' */
#define BNM_QAT_NORM_LIN 1        /* 'Lin' */
#define BNM_QAT_NORM_BATCHNORM 2  /* 'BatchNorm' */
#define BNM_QAT_NORM_LAYERNORM 3  /* 'LayerNorm' */
#define BNM_QAT_NORM_NONE 4       /* no normalisation (BitConv2d's 'None') */
BNM_API uint64_t bnm_qat_workspace_bytes(uint32_t d, uint32_t k);
BNM_API int bnm_qat_bitlinear_forward_device(const float *d_x, uint64_t n, uint32_t d, const float *d_w, uint32_t k,
                                             const float *d_s, uint32_t s_count, int quant_type, int norm_type,
                                             float *d_y, void *d_workspace, uint64_t workspace_bytes,
                                             float *d_x_int_out, float *d_x_scale_out, float *d_w_deq_out,
                                             void *stream);

/* BitConv2d forward (BitNetMCU.py:264-322): any group structure (groups = 1, depthwise as in models.py:111-116, anything
 * between), any stride, kernel kh x kw, symmetric zero padding `pad`, PerTensor clipping scalar.
 * d_x [n][cin][h][w], d_w [cout][cin/groups][kh][kw], d_y [n][cout][(h+2p-kh)/stride+1][(w+2p-kw)/stride+1], float32.
 * norm_type: BNM_QAT_NORM_RMS (over each plane) or BNM_QAT_NORM_NONE.  Activations are quantised per image row of each plane
 * (the reference's max over the last dimension).  workspace: bnm_qat_workspace_bytes((cin/groups)*kh*kw, cout) bytes.
 * BNM_EUNSUPPORTED when one group's input planes + weights exceed 160 KiB of LDS. */
BNM_API int bnm_qat_bitconv2d_forward_device(const float *d_x, uint64_t n, uint32_t cin, uint32_t h, uint32_t w,
                                             const float *d_w, uint32_t cout, uint32_t kh, uint32_t kw, uint32_t pad,
                                             uint32_t stride, uint32_t groups, const float *d_s, int quant_type, int norm_type,
                                             float *d_y, void *d_workspace, uint64_t workspace_bytes, void *stream);

/* Whole-model QAT forward (models.py:56-90 FCMNIST; the FC stack of CNNMNIST, :120-135): n_layers BitLinear layers
 * (BitNetMCU.py:214-235, bias-free) with ReLU between them, ONE kernel per call behind a per-call weight preparation launch:
 *   widths[0] <= 256 inputs (the 16x16 image, or CNNMNIST's channels x 4; fewer than 256: d_x still holds rows of 256 floats, the
 *   caller pads them with zeros, and NormType LayerNorm is not served), widths[1 .. n_layers-1] the hidden widths
 *   (each <= 192), widths[n_layers] the classes (<= 64); 2 <= n_layers <= BNM_QAT_MODEL_MAX_LAYERS.
 *   d_w[l] [widths[l+1]][widths[l]] float32, d_s[l] the layer's clipping scalar(s), s_count[l] 1 (PerTensor) or widths[l+1]
 *   (PerOutput), quant_types[l] the layer's QuantType - d_w / d_s / s_count / quant_types / widths are HOST arrays (of device
 *   pointers where they hold pointers).
 *   quant_types: those whose levels (x 2 for the half-integer types) are int8 - Binary, BinarySym, Ternary, 2bitsym, 4bitsym,
 *   5bitsym, 8bit; norm_type: BNM_QAT_NORM_RMS, BNM_QAT_NORM_LIN or BNM_QAT_NORM_LAYERNORM (BatchNorm needs the whole batch per layer).  Anything else:
 *   BNM_EUNSUPPORTED (bnm_qat_model_supported tells beforehand) - run the layers one by one with bnm_qat_bitlinear_forward_device.
 *   d_x [n][256] float32 in, d_logits [n][classes] float32 out (16-byte aligned).  Optional: d_hidden [n][sum of the hidden widths]
 *   - every hidden layer's output after ReLU, i.e. the next layer's input, layer after layer within a row - and d_w_deq[l]
 *   [widths[l+1]][widths[l]] = w_int / w_scale: together with d_x what a straight-through backward pass needs.
 *   workspace: bnm_qat_model_workspace_bytes(n_layers, widths) bytes, 16-byte aligned, owned by the call's stream while it runs.
 * Floating point: within the tolerances of tests/test_gpu_qat_model.py of the reference module, not bit-exact.  A row whose input
 * to some layer is all zero gets NaN logits, as in the reference (0 / 0 in Normalize; not under LayerNorm, whose epsilon keeps it finite). */
#define BNM_QAT_MODEL_MAX_LAYERS 4
BNM_API uint64_t bnm_qat_model_workspace_bytes(uint32_t n_layers, const uint32_t *widths);
BNM_API int bnm_qat_model_supported(uint32_t n_layers, const uint32_t *widths, const int *quant_types, int norm_type);
BNM_API int bnm_qat_model_forward_device(const float *d_x, uint64_t n, uint32_t n_layers, const uint32_t *widths,
                                         const float *const *d_w, const float *const *d_s, const uint32_t *s_count,
                                         const int *quant_types, int norm_type, float *d_logits, float *d_hidden,
                                         float *const *d_w_deq, void *d_workspace, uint64_t workspace_bytes, void *stream);

/* The convolution front of the reference's CNNMNIST (models.py:109-119) in ONE kernel behind a tap-preparation launch:
 *   BitConv2d(1 -> C, 3x3) -> ReLU -> BitConv2d(C -> C, 3x3, groups = C) -> ReLU -> MaxPool2d(2) -> BitConv2d(C -> C, 3x3, groups = C)
 *   -> ReLU -> MaxPool2d(2) -> Flatten, every BitConv2d with NormType 'None' and stride 1 / no padding (BitNetMCU.py:285-305).
 *   d_x [n][256] float32 (16x16 images) in, d_features [n][4 C] float32 out (Flatten's order: channel-major, then 2x2) - the input of
 *   the model's FC stack (bnm_qat_model_forward_device where 4 C = 256).  d_w[l] [C][9] float32 (the layers' weight tensors
 *   [C][1][3][3] as they lie), d_s[l] the layer's clipping scalar, s_count[l] = 1, quant_types[l] any BNM_QAT_* but BNM_QAT_NONE (which skips
 *   activation_quant as well) - d_w / d_s / s_count / quant_types are three-element HOST arrays.  Serves an even C of 16 .. 128
 *   with per-tensor clipping scalars (bnm_qat_cnn_front_supported); anything else: BNM_EUNSUPPORTED - run the layers with bnm_qat_bitconv2d_forward_device.
 *   workspace: bnm_qat_cnn_front_workspace_bytes(C) bytes, 16-byte aligned like d_x and d_features.
 * Floating point: within the tolerances of tests/test_gpu_qat_cnn.py of the reference module, not bit-exact. */
BNM_API int bnm_qat_cnn_front_supported(uint32_t channels, const uint32_t *s_count, const int *quant_types);
BNM_API uint64_t bnm_qat_cnn_front_workspace_bytes(uint32_t channels);
BNM_API int bnm_qat_cnn_front_forward_device(const float *d_x, uint64_t n, uint32_t channels, const float *const *d_w,
                                             const float *const *d_s, const uint32_t *s_count, const int *quant_types,
                                             float *d_features, void *d_workspace, uint64_t workspace_bytes, void *stream);
/* The training form: the same call, which also writes the three convolutions' outputs BEFORE their ReLU, CHANNELS-LAST -
 * d_y1 [n][14][14][C], d_y2 [n][12][12][C], d_y3 [n][4][4][C] (float32, 16-byte aligned: the NHWC memory order of an [n, C, k, k]
 * tensor; all three or all NULL = the call above).  With d_x and the weights they are all a straight-through backward pass
 * needs: every ReLU mask, pooling argument and layer input is a function of them.  356 C floats per image: bound by those writes. */
BNM_API int bnm_qat_cnn_front_forward_train_device(const float *d_x, uint64_t n, uint32_t channels, const float *const *d_w,
                                                   const float *const *d_s, const uint32_t *s_count, const int *quant_types,
                                                   float *d_features, float *d_y1, float *d_y2, float *d_y3, void *d_workspace,
                                                   uint64_t workspace_bytes, void *stream);

/* ---- synthetic workload + digests ---------------------------------------- */
#define BNM_DIST_U 0
#define BNM_DIST_M 1
#define BNM_SEED_DIST_U 0xB17E7001ull
#define BNM_SEED_DIST_M 0xB17E7002ull
/* Fill count images (256 B each) for global image indices [first, first+count). */
BNM_API int bnm_synth_fill_device(int8_t *d_images, uint64_t first, uint64_t count, uint64_t seed,
                                  int dist, void *stream);
/* d_out[0] += sum_i splitmix64((first+i)*64 + cls[i]);  d_out[1+c] += #(cls == c), c < n_bins
 * (uint64 accumulators, caller zeroes them). */
BNM_API int bnm_class_digest_device(const uint32_t *d_cls, uint64_t first, uint64_t n,
                                    uint64_t *d_out, uint32_t n_bins, void *stream);

/* The box's plain read rate: nontemporal 16 B/lane loads of `bytes` bytes at d_src (16-byte aligned), nothing written (d_sink:
 * one device dword the kernel practically never touches).  bench.py times it over the resident image set next to the inference
 * kernels: what an HBM-bound kernel can at best approach on the box it runs on. */
BNM_API int bnm_stream_read_device(const void *d_src, uint64_t bytes, uint32_t *d_sink, void *stream);
/* The box's MIXED stream rate, no arithmetic: n_rows / 32 tiles of 32 x 256 bytes read at d_src, every tile's 32 x
 * out_bytes_per_row bytes (a multiple of 4; 44 = a class id + ten int32 logits) written contiguously at d_dst (>= n_rows x
 * out_bytes_per_row bytes) with nontemporal 16 B/lane stores.  mode 0: what the dual-tile kernel does (a wave reads two tiles,
 * then writes their results; two waves per SIMD); otherwise tiles per batch (1, 2, 4, 8) + 16 for plain stores + 32 x (waves per
 * SIMD - 2).  bench.py times it next to the ids + logits row. */
BNM_API int bnm_stream_rw_device(const void *d_src, uint64_t n_rows, void *d_dst, uint32_t out_bytes_per_row, uint32_t mode, void *stream);

/* ---- multi-GPU, single process (C hosts; PyTorch hosts use one process per GPU, see bench.py) ------------------
 * Shards the global synthetic image stream [0, n_total) contiguously over the first n_gpus visible devices
 * (n_gpus <= 0: all), generates every shard on its own GPU, runs the whole-model path on all of them concurrently and
 * returns the combined order-independent digest + class histogram (digest_hist[0], digest_hist[1..n_bins]) and the
 * wall-clock seconds of the inference phase (one warm pass over all shards, after an untimed one; the slowest GPU's time).
 * One host thread and one single-process RCCL communicator per GPU: the model (~13 KB BNMBLOB) is uploaded to GPU 0 only and
 * reaches the others through ncclBroadcast over xGMI, the digests meet in one ncclAllReduce; no image byte crosses a link.
 * librccl is bound with dlopen at the first call (the library does not link it); where it cannot be loaded the host uploads the
 * model to every GPU and sums the digests itself - bnm_multi_gpu_transport() names the transport of the calling thread's last
 * call ("rccl" or "host").  Returns the number of GPUs used (> 0) or a negative BNM_E* code. */
BNM_API int bnm_run_synth_multi_gpu(const bnm_model *m, uint64_t n_total, int n_gpus, int dist, uint64_t seed,
                                    uint64_t *digest_hist, uint32_t n_bins, double *seconds);
BNM_API const char *bnm_multi_gpu_transport(void);

/* ---- model binding for group A ------------------------------------------------------------ */
/* Bind the model that Inference()/BitMnistInference() run.  A `Bitnet_inf.dll` built by
 * bitnetmcu_amd/build.py --dll <header> does this itself from the embedded header text. */
BNM_API int bnm_bind_default_model(const bnm_model *m);

/* ---- device helpers (so a C host needs no other HIP code) -------------------------------- */
BNM_API int bnm_device_count(void);
BNM_API int bnm_device_malloc(void **p, size_t bytes);
BNM_API int bnm_device_free(void *p);
BNM_API int bnm_memcpy_h2d(void *d, const void *h, size_t bytes);
BNM_API int bnm_memcpy_d2h(void *h, const void *d, size_t bytes);
BNM_API int bnm_device_synchronize(void);

#ifdef __cplusplus
}
#endif
#endif
