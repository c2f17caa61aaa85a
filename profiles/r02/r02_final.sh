#!/bin/bash
# Runs ON THE GPU BOX from the repo root: the round-2 evidence set.
#  1. rocprofv3 --kernel-trace --stats of the DEFAULT bench command (the line the driver records), summary -> gpurun_out/r02_final/
#  2. counter passes (each in its own run) for the headline kernel (+ HBM traffic), the ternary ALU kernel, the CNN front end,
#     the generic kernel on the headline model -> table.json per kernel
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r02_final; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -f csv -d "$OUT/stats" -o stats -- python $REPO/bench.py --no-cpu > "$OUT/bench_under_rocprof.json" 2> "$OUT/bench_under_rocprof.err"
cd "$REPO"
python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
PMC_TRAFFIC=1 PMC_STALL=1 profiles/pmc_kernel.sh dual > "$OUT/pmc_dual.md" 2>&1
profiles/pmc_kernel.sh tern_alu --model tern_96 --path 3 --images 20000000 > "$OUT/pmc_tern_alu.md" 2>&1
profiles/pmc_kernel.sh cnn_mfma --model cnn_64 --images 1048576 > "$OUT/pmc_cnn_mfma.md" 2>&1
profiles/pmc_kernel.sh fc_generic --variant 4 > "$OUT/pmc_fc_generic.md" 2>&1
for t in dual tern_alu cnn_mfma fc_generic; do cp gpurun_out/pmc_$t/table.json "$OUT/table_$t.json"; done
python - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
f = glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True)
print("| kernel | calls | total ms | avg ms | min ms | max ms | % |\n|---|---|---|---|---|---|---|")
for r in csv.DictReader(open(f[0])):
    g = lambda k: float(r.get(k, 0) or 0)
    print(f"| `{r['Name'][:100]}` | {r['Calls']} | {g('TotalDurationNs')/1e6:.3f} | {g('AverageNs')/1e6:.4f} | {g('MinNs')/1e6:.4f} | {g('MaxNs')/1e6:.4f} | {r['Percentage']} |")
PY
