#!/bin/bash
# counter passes for the round-2 ternary (streamed) and CNN kernels
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r02p
PMC_STALL=1 profiles/pmc_kernel.sh tern_stream --model tern_96 --path 3 --images 20000000 > gpurun_out/r02p/pmc_tern_stream.md 2>&1
PMC_STALL=1 profiles/pmc_kernel.sh cnn_mfma2 --model cnn_64 --images 1048576 > gpurun_out/r02p/pmc_cnn_mfma2.md 2>&1
cp gpurun_out/pmc_tern_stream/table.json gpurun_out/r02p/table_tern_stream.json
cp gpurun_out/pmc_cnn_mfma2/table.json gpurun_out/r02p/table_cnn_mfma2.json
head -50 gpurun_out/r02p/pmc_tern_stream.md
