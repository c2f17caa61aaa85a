#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): memory-side PMC counters of the fused kernel next to the plain streaming kernels
# of profiles/stream_ceiling.py — average TCC->EA (HBM) read latency and occupancy, DRAM credit stalls, TCP->TCC latency.
# Counters only (--kernel-trace + --pmc), one block per pass.   usage: profiles/mem_latency_probe.sh <tag>
set -u
TAG=${1:-probe}
REPO=$(pwd)
OUT=$REPO/gpurun_out/memlat_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
A="TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum"
B="TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum"
C="TCC_BUSY_sum TCC_CYCLE_sum GRBM_GUI_ACTIVE"
for pass in A B C; do
  eval CTR=\$$pass
  rocprofv3 --kernel-trace --pmc $CTR -f csv -d "$OUT/fused_$pass" -o pmc -- python $REPO/bench.py --steps 3 --warmup 1 --no-cpu --no-verify > "$OUT/fused_$pass.log" 2>&1
  rocprofv3 --kernel-trace --pmc $CTR -f csv -d "$OUT/stream_$pass" -o pmc -- python $REPO/profiles/stream_ceiling.py > "$OUT/stream_$pass.log" 2>&1
done
cd "$REPO"
python profiles/mem_latency_summary.py "$OUT" | tee "$OUT/summary.md"
