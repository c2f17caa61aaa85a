/*
 * Translation unit that turns libbitnetmcu_hip into a model-bound, drop-in `Bitnet_inf.dll`.
 *
 * The reference builds its DLL from BitNetMCU_MNIST_dll.c, which #includes "BitNetMCU_model.h"
 * (BitNetMCU_MNIST_dll.c:4, Makefile:5-6).  This stub keeps that build flow — the exporter's header is
 * consumed unchanged at compile time — but instead of compiling a fixed layer schedule against fixed
 * L1..L4 names (which today's exporter no longer emits for FC models) it
 *   1. #includes the header, so the Lk_weights[] data symbols exist exactly as in the reference DLL, and
 *   2. embeds the header TEXT, which the run-time loader (bnm_model.cpp) parses on the first Inference().
 * BNM_MODEL_HEADER_PATH is passed by bitnetmcu_amd/build.py --dll.
 */
#include <stdint.h>
#include BNM_MODEL_HEADER_PATH

__asm__(
    ".section .rodata\n"
    ".global bnm_embedded_header_begin\n"
    ".type bnm_embedded_header_begin, @object\n"
    "bnm_embedded_header_begin:\n"
    ".incbin \"" BNM_MODEL_HEADER_PATH "\"\n"
    ".global bnm_embedded_header_end\n"
    ".type bnm_embedded_header_end, @object\n"
    "bnm_embedded_header_end:\n"
    ".byte 0\n"
    ".previous\n");
