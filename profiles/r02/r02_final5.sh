#!/bin/bash
# Runs ON THE GPU BOX from the repo root: evidence for the headline kernel with nontemporal class-id stores (round 2, last change).
#  1. rocprofv3 --kernel-trace --stats of the default bench command without the extra rows  2. python bench.py (the driver's line)
#  3. counter passes of the headline kernel (+ FETCH_SIZE / WRITE_SIZE)  4. store flavours A/B (diagnostic library)  5. pytest -m gpu, smoke
set -u
REPO=$(pwd); TAG=${1:-r02_final5}; OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -f csv -d "$OUT/stats_headline" -o stats -- python $REPO/bench.py --no-cpu --no-extra > "$OUT/bench_headline_under_rocprof.json" 2> "$OUT/bench_headline_under_rocprof.err"
cd "$REPO"
python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
PMC_TRAFFIC=1 PMC_STALL=1 profiles/pmc_kernel.sh dual > "$OUT/pmc_dual.md" 2>&1
cp gpurun_out/pmc_dual/table.json "$OUT/table_dual.json"
BNM_LIBRARY=bitnetmcu_amd/libbitnetmcu_hip_diag.so timeout 300 python profiles/store_modes_ab.py > "$OUT/store_modes_ab.log" 2>&1
BNM_LIBRARY=bitnetmcu_amd/libbitnetmcu_hip_diag.so timeout 300 python profiles/logits_store_ab.py > "$OUT/logits_store_ab.log" 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1; tail -2 "$OUT/pytest_gpu.log"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; tail -3 "$OUT/smoke.log"
python - "$OUT/stats_headline" <<'PY' > "$OUT/kernel_stats_headline.md"
import csv, glob, sys
out = sys.argv[1]
f = glob.glob(out + "/**/*kernel_stats.csv", recursive=True)
print("| kernel | calls | total ms | avg ms | min ms | max ms | % |\n|---|---|---|---|---|---|---|")
for r in csv.DictReader(open(f[0])):
    g = lambda k: float(r.get(k, 0) or 0)
    print(f"| `{r['Name'][:110]}` | {r['Calls']} | {g('TotalDurationNs')/1e6:.3f} | {g('AverageNs')/1e6:.4f} | {g('MinNs')/1e6:.4f} | {g('MaxNs')/1e6:.4f} | {r['Percentage']} |")
PY
cat "$OUT/kernel_stats_headline.md" | head -5; grep -v '^{' "$OUT/store_modes_ab.log" | tail -11; grep -v '^{' "$OUT/logits_store_ab.log" | tail -4
python - "$OUT/bench.json" <<'PY'
import json, sys
t = open(sys.argv[1]).read(); d = json.loads(t[t.index('{"metric'):])
print("bench.py:", d["ms_per_step"], d["value"], d["roofline"]["frac"], d.get("digest"))
for k, v in d.get("extra_configs", {}).items(): print("  ", k, v.get("value"), v.get("verified_vs_oracle"))
PY
