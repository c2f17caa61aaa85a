#!/bin/bash
# Runs ON THE GPU BOX from the repo root: the round-2 evidence set with the round's final kernels.
#  1. rocprofv3 --kernel-trace --stats of the DEFAULT bench command without the extra rows, and of the full default line
#  2. python bench.py (the line the driver records)
#  3. counter passes (each in its own run): headline kernel (+ HBM traffic), streamed ternary kernel, CNN front end, generic
#     kernel on the headline model and on the ternary model -> table.json per kernel
set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r02_final4; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -f csv -d "$OUT/stats_headline" -o stats -- python $REPO/bench.py --no-cpu --no-extra > "$OUT/bench_headline_under_rocprof.json" 2> "$OUT/bench_headline_under_rocprof.err"
rocprofv3 --kernel-trace --stats -f csv -d "$OUT/stats" -o stats -- python $REPO/bench.py --no-cpu > "$OUT/bench_under_rocprof.json" 2> "$OUT/bench_under_rocprof.err"
cd "$REPO"
python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
PMC_TRAFFIC=1 PMC_STALL=1 profiles/pmc_kernel.sh dual > "$OUT/pmc_dual.md" 2>&1
PMC_STALL=1 profiles/pmc_kernel.sh tern_stream --model tern_96 --path 3 --images 20000000 > "$OUT/pmc_tern_stream.md" 2>&1
PMC_STALL=1 profiles/pmc_kernel.sh cnn_mfma --model cnn_64 --images 1048576 > "$OUT/pmc_cnn_mfma.md" 2>&1
profiles/pmc_kernel.sh fc_generic --variant 4 > "$OUT/pmc_fc_generic.md" 2>&1
profiles/pmc_kernel.sh tern_generic --model tern_96 --path 1 --images 20000000 > "$OUT/pmc_tern_generic.md" 2>&1
for t in dual tern_stream cnn_mfma fc_generic tern_generic; do cp gpurun_out/pmc_$t/table.json "$OUT/table_$t.json"; done
for d in stats_headline stats; do
python - "$OUT/$d" <<'PY' > "$OUT/kernel_$d.md"
import csv, glob, sys
out = sys.argv[1]
f = glob.glob(out + "/**/*kernel_stats.csv", recursive=True)
print("| kernel | calls | total ms | avg ms | min ms | max ms | % |\n|---|---|---|---|---|---|---|")
for r in csv.DictReader(open(f[0])):
    g = lambda k: float(r.get(k, 0) or 0)
    print(f"| `{r['Name'][:110]}` | {r['Calls']} | {g('TotalDurationNs')/1e6:.3f} | {g('AverageNs')/1e6:.4f} | {g('MinNs')/1e6:.4f} | {g('MaxNs')/1e6:.4f} | {r['Percentage']} |")
PY
done
cat "$OUT/kernel_stats_headline.md"
