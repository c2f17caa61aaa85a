#!/bin/bash
# Round-3 evidence set, ON THE GPU BOX from the repo root, one box visit:
#   1. the default bench line (20 timed launches after 3 warm-ups; every extra row);
#   2. the headline command under rocprofv3 --kernel-trace --stats: per-launch durations of the dominant kernel with the
#      warm-ups excluded, next to the HIP-event numbers of THAT run (profiles/kernel_trace_summary.py);
#   3. counter passes (each block in its own run, never combined with other trace domains): headline kernel incl. FETCH_SIZE /
#      WRITE_SIZE, generic kernel on the headline model, on the ternary and the binary 12 KB models, CNN front end, ternary ALU.
# Output under gpurun_out/r03e/; copy what is to be judged into profiles/r03/.
set -u
TAG=${1:-r03e}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
timeout 400 python bench.py --steps 20 --warmup 3 > "$OUT/bench.json" 2> "$OUT/bench.err"
export TMPDIR=/tmp
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d "$OUT/trace_headline" -o t -- python "$REPO/bench.py" --steps 20 --warmup 3 --no-extra --no-cpu > "$OUT/bench_under_rocprof.log" 2>&1 )
python profiles/kernel_trace_summary.py "$OUT/trace_headline" "$OUT/bench_under_rocprof.log" "void fused_fc_dual_kernel" > "$OUT/rocprof_kernel_trace_headline.md" 2>&1
python profiles/summarize.py "$OUT/trace_headline" > "$OUT/rocprof_kernel_stats_headline.md" 2>/dev/null
PMC_TRAFFIC=1 timeout 600 bash profiles/pmc_kernel.sh ${TAG}_dual > "$OUT/pmc_dual.md" 2>&1
timeout 400 bash profiles/pmc_kernel.sh ${TAG}_fc_generic --variant 4 > "$OUT/pmc_fc_generic.md" 2>&1
timeout 400 bash profiles/pmc_kernel.sh ${TAG}_tern_generic --model tern_96 --images 20000000 > "$OUT/pmc_tern_generic.md" 2>&1
timeout 400 bash profiles/pmc_kernel.sh ${TAG}_binary160 --model doc12k_binary --images 20000000 > "$OUT/pmc_doc12k_binary.md" 2>&1
timeout 400 bash profiles/pmc_kernel.sh ${TAG}_cnn --model cnn_64 --images 1048576 > "$OUT/pmc_cnn.md" 2>&1
timeout 400 bash profiles/pmc_kernel.sh ${TAG}_tern_alu --model tern_96 --path 3 --images 20000000 > "$OUT/pmc_tern_alu.md" 2>&1
timeout 400 bash profiles/pmc_kernel.sh ${TAG}_cnn16 --model mcu_cnn_16 --images 1048576 > "$OUT/pmc_cnn16.md" 2>&1
timeout 400 bash profiles/pmc_kernel.sh ${TAG}_cnn48 --model mcu_cnn_48 --images 1048576 > "$OUT/pmc_cnn48.md" 2>&1
timeout 400 bash profiles/pmc_kernel.sh ${TAG}_tern128_alu --model doc12k_ternary --path 3 --images 20000000 > "$OUT/pmc_tern128_alu.md" 2>&1
for t in dual fc_generic tern_generic binary160 cnn tern_alu cnn16 cnn48 tern128_alu; do cp "gpurun_out/pmc_${TAG}_$t/table.json" "$OUT/table_$t.json" 2>/dev/null; done
tail -c 400 "$OUT/bench.json"; cat "$OUT/rocprof_kernel_trace_headline.md"
