#!/bin/bash
# Round-4 evidence set, ON THE GPU BOX from the repo root, one box visit:
#   1. the default bench line (20 timed launches after 3 warm-ups; every extra row);
#   2. the headline command under rocprofv3 --kernel-trace --stats: per-launch durations of the dominant kernel with the
#      warm-ups excluded, next to the HIP-event numbers of THAT run (profiles/kernel_trace_summary.py);
#   3. counter passes (each block in its own run, never combined with other trace domains): headline kernel incl. FETCH_SIZE /
#      WRITE_SIZE; the CNN's lane = image kernel on the 64-, 16- and 48-channel models and the channel kernel on the 64-channel
#      one; generic and register-resident-weight kernels on the ternary 96-96-96 model; generic on the binary 12 KB model;
#      the ternary ALU kernels.
# Output under gpurun_out/<tag>/; profiles/make_counters_json.py turns the tables into profiles/pmc_*.json (stamped with the
# kernels' code hashes); copy what is to be judged into profiles/r04/.
set -u
TAG=${1:-r04z}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
timeout 500 python bench.py --steps 20 --warmup 3 > "$OUT/bench.json" 2> "$OUT/bench.err"
export TMPDIR=/tmp
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d "$OUT/trace_headline" -o t -- python "$REPO/bench.py" --steps 20 --warmup 3 --no-extra --no-cpu > "$OUT/bench_under_rocprof.log" 2>&1 )
python profiles/kernel_trace_summary.py "$OUT/trace_headline" "$OUT/bench_under_rocprof.log" "void fused_fc_dual_kernel" > "$OUT/rocprof_kernel_trace_headline.md" 2>&1
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d "$OUT/trace_cnn" -o t -- python "$REPO/bench.py" --model cnn_64 --images 10000000 --steps 10 --warmup 2 --no-extra --no-cpu > "$OUT/bench_cnn_under_rocprof.log" 2>&1 )
python profiles/kernel_trace_summary.py "$OUT/trace_cnn" "$OUT/bench_cnn_under_rocprof.log" "cnn_li_kernel" --timed 30 > "$OUT/rocprof_kernel_trace_cnn.md" 2>&1
PMC_TRAFFIC=1 timeout 600 bash profiles/pmc_kernel.sh ${TAG}_dual > "$OUT/pmc_dual.md" 2>&1
timeout 400 bash profiles/pmc_kernel.sh ${TAG}_fc_generic --variant 4 > "$OUT/pmc_fc_generic.md" 2>&1
timeout 400 bash profiles/pmc_kernel.sh ${TAG}_cnn_li --model cnn_64 --images 4194304 > "$OUT/pmc_cnn_li.md" 2>&1
timeout 400 bash profiles/pmc_kernel.sh ${TAG}_cnn_li16 --model mcu_cnn_16 --images 4194304 > "$OUT/pmc_cnn_li16.md" 2>&1
timeout 400 bash profiles/pmc_kernel.sh ${TAG}_cnn_li48 --model mcu_cnn_48 --images 4194304 > "$OUT/pmc_cnn_li48.md" 2>&1
timeout 400 bash profiles/pmc_kernel.sh ${TAG}_cnn_channel --model cnn_64 --images 1048576 --cnn-variant 1 > "$OUT/pmc_cnn_channel.md" 2>&1
timeout 400 bash profiles/pmc_kernel.sh ${TAG}_tern_generic --model tern_96 --images 20000000 > "$OUT/pmc_tern_generic.md" 2>&1
timeout 400 bash profiles/pmc_kernel.sh ${TAG}_tern_regw --model tern_96 --images 20000000 --variant 9 > "$OUT/pmc_tern_regw.md" 2>&1
timeout 400 bash profiles/pmc_kernel.sh ${TAG}_binary160 --model doc12k_binary --images 20000000 > "$OUT/pmc_doc12k_binary.md" 2>&1
timeout 400 bash profiles/pmc_kernel.sh ${TAG}_tern_alu --model tern_96 --path 3 --images 20000000 > "$OUT/pmc_tern_alu.md" 2>&1
timeout 400 bash profiles/pmc_kernel.sh ${TAG}_tern128_alu --model doc12k_ternary --path 3 --images 20000000 > "$OUT/pmc_tern128_alu.md" 2>&1
for t in dual fc_generic cnn_li cnn_li16 cnn_li48 cnn_channel tern_generic tern_regw binary160 tern_alu tern128_alu; do cp "gpurun_out/pmc_${TAG}_$t/table.json" "$OUT/table_$t.json" 2>/dev/null; done
tail -c 300 "$OUT/bench.json"; cat "$OUT/rocprof_kernel_trace_headline.md" "$OUT/rocprof_kernel_trace_cnn.md"
