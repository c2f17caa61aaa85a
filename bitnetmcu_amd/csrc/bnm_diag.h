/* Diagnostic entry points of libbitnetmcu_hip_diag.so (bitnetmcu_amd/build.py --diag).  NOT part of the product ABI
 * (include/bitnetmcu_hip.h): the product library does not export them and its kernels ignore src_wrap. */
#ifndef BNM_DIAG_H
#define BNM_DIAG_H
#include "../../include/bitnetmcu_hip.h"
#ifdef __cplusplus
extern "C" {
#endif
/* Read the image stream without the model math.  mode 0: plain 16 B/lane loads; mode 1/2: the fused kernel's own
 * LDS-DMA tile loop (one tile ahead / two tiles in flight); mode 3/4: the stream under a synthetic compute load, fed
 * by the LDS-DMA loop / by plain loads into VGPRs; mode 5/6/7: no memory traffic, n = tiles per wave of 26 MFMAs /
 * ~400 VALU / both.  d_out: uint32 [n]. */
BNM_API int bnm_diag_stream_device(const int8_t *d_images, uint64_t n, int mode, int grid_blocks, uint32_t *d_out,
                                   void *stream);
/* Make the fused kernels of this context read tile (t mod wrap) — a cache-resident source, for compute-side timing.
 * Class ids are WRONG by design while wrap != 0. */
BNM_API int bnm_diag_set_src_wrap(bnm_ctx *c, uint64_t wrap);
/* --diag-timing build only: where cnn_front_mfma_kernel's waves wait.  d_rec: uint64 [waves][8] = {loop cycles, cycles waiting
 * for an item's head loads, for the patch tiles, for the partner exchange, items, start stamp, XCC_ID, HW_ID}; NULL switches the records off. */
BNM_API int bnm_diag_cnn_set_record(uint64_t *d_rec);
#ifdef __cplusplus
}
#endif
#endif
