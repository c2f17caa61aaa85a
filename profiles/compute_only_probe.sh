#!/bin/bash
# Runs ON THE GPU BOX: the fused kernel with a cache-resident source (BNM_DIAG_SRC_WRAP) under the SQ counters, to
# compare its CYCLE count with the HBM-streaming run (profiles/run_profiles.sh pass sq1): same cycles at a higher
# clock => the kernel is issue-bound and HBM only costs clock (power); fewer cycles => memory stalls matter.
set -u
TAG=${1:-probe}
REPO=$(pwd)
OUT=$REPO/gpurun_out/conly_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
CTR="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
for wrap in 256 0; do
  BNM_DIAG_SRC_WRAP=$wrap rocprofv3 --kernel-trace --pmc $CTR -f csv -d "$OUT/wrap$wrap" -o pmc -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu --no-verify > "$OUT/wrap$wrap.log" 2>&1
done
cd "$REPO"
for wrap in 256 0; do echo "== BNM_DIAG_SRC_WRAP=$wrap"; python profiles/summarize.py "$OUT/wrap$wrap" 2>/dev/null | grep -E "^\| (GRBM|SQ_)|/ SQ_WAVE"; grep -o '"ms_per_step": [0-9.]*' "$OUT/wrap$wrap.log"; done
