#!/bin/bash
# Runs ON THE GPU BOX (via gpurun) from the repo root.  Writes raw rocprofv3 output under gpurun_out/prof_<tag>/;
# profiles/summarize.py turns it into the committed summaries under profiles/<round>/.
#   usage: profiles/run_profiles.sh <tag> [bench.py args...]
set -u
TAG=${1:-r01}; shift || true
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python $REPO/bench.py --steps 3 --warmup 1 --no-cpu --no-verify $*"
cd /tmp
# 1. kernel trace + stats (timing; never combined with counters): bench.py's DEFAULT step counts, so the kernel's
#    average duration here is comparable with the live HIP-event number bench.py prints
rocprofv3 --kernel-trace --stats -f csv -d "$OUT/stats" -o stats -- python $REPO/bench.py --no-cpu $* > "$OUT/stats.log" 2>&1
# 2. counters, one block per pass (TCC: FETCH_SIZE needs 3 of 4 slots, WRITE_SIZE 2 — separate passes)
rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d "$OUT/pmc_fetch" -o pmc -- $BENCH > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d "$OUT/pmc_write" -o pmc -- $BENCH > "$OUT/pmc_write.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -f csv -d "$OUT/pmc_sq1" -o pmc -- $BENCH > "$OUT/pmc_sq1.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM -f csv -d "$OUT/pmc_sq2" -o pmc -- $BENCH > "$OUT/pmc_sq2.log" 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -f csv -d "$OUT/pmc_tcc" -o pmc -- $BENCH > "$OUT/pmc_tcc.log" 2>&1
cd "$REPO"
find "$OUT" -name "*.csv" | head -40
python profiles/summarize.py "$OUT" > "$OUT/summary.md" 2>&1
cat "$OUT/summary.md"
