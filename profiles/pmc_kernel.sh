#!/bin/bash
# Runs ON THE GPU BOX from the repo root: rocprofv3 stats + SQ counter passes for ONE bench.py configuration.
#   usage: profiles/pmc_kernel.sh <tag> <bench.py args...>          e.g.  pmc_kernel.sh tern_generic --model tern_96 --path 1 --images 20000000
#          PMC_CMD='python profiles/qat_model_bench.py --rows 10000000 --steps 3 --warmup 1' profiles/pmc_kernel.sh qat
# Each counter block is collected in its own pass (never combined with tracing domains other than --kernel-trace).
# Output: gpurun_out/pmc_<tag>/{stats,sq1,sq2[,fetch,write]}/...csv and a per-kernel table on stdout (profiles/pmc_table.py).
set -u
TAG=$1; shift
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp; cd /tmp
B="python $REPO/bench.py --steps 3 --warmup 1 --no-cpu --no-verify --no-extra $*"
BS="python $REPO/bench.py --steps 10 --warmup 3 --no-cpu --no-verify --no-extra $*"
# PMC_CMD: another command than bench.py (e.g. profiles/qat_model_bench.py: kernels that are not bench.py's main workload); PMC_CMD_STATS: its timing form
if [ -n "${PMC_CMD:-}" ]; then B="$PMC_CMD"; BS="${PMC_CMD_STATS:-$PMC_CMD}"; fi
rocprofv3 --kernel-trace --stats -f csv -d "$OUT/stats" -o stats -- $BS > "$OUT/stats.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -f csv -d "$OUT/sq1" -o pmc -- $B > "$OUT/sq1.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM -f csv -d "$OUT/sq2" -o pmc -- $B > "$OUT/sq2.log" 2>&1
if [ "${PMC_STALL:-0}" = "1" ]; then
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_COEXEC_CYCLES SQ_INST_CYCLES_VMEM -f csv -d "$OUT/sq3" -o pmc -- $B > "$OUT/sq3.log" 2>&1
fi
if [ "${PMC_TRAFFIC:-0}" = "1" ]; then
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d "$OUT/fetch" -o pmc -- $B > "$OUT/fetch.log" 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d "$OUT/write" -o pmc -- $B > "$OUT/write.log" 2>&1
fi
cd "$REPO"
python profiles/pmc_table.py "$OUT" --json "$OUT/table.json"
