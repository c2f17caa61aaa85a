#!/bin/bash
# Round-5 evidence set, ON THE GPU BOX from the repo root, one box visit:
#   1. the default bench line (20 timed launches after 3 warm-ups; every extra row, the float-input and one-kernel CNN rows among them);
#   2. rocprofv3 --kernel-trace --stats of the headline command, of the float-input workload (fused_fc_f32_kernel) and of the CNN
#      (cnn_li_fused_pipe_kernel: one launch per step): per-launch durations with the warm-ups excluded next to the HIP-event numbers
#      of THAT run (profiles/kernel_trace_summary.py);
#   3. counter passes, each block in its own run (never combined with other trace domains), FETCH_SIZE / WRITE_SIZE in passes
#      of their own: the headline kernel, the float-input kernel (2- and 4-tile class), the one-kernel CNN on the 64-, 16- and
#      48-channel models and the two-launch form on the 64-channel one.
# Output under gpurun_out/<tag>/; profiles/make_counters_json.py turns the tables into profiles/pmc_*.json (stamped with the
# kernels' code hashes); copy what is to be judged into profiles/r05/.
set -u
TAG=${1:-r05e}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
timeout 600 python bench.py --steps 20 --warmup 3 > "$OUT/bench.json" 2> "$OUT/bench.err"
export TMPDIR=/tmp
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d "$OUT/trace_headline" -o t -- python "$REPO/bench.py" --steps 20 --warmup 3 --no-extra --no-cpu > "$OUT/bench_under_rocprof.log" 2>&1 )
python profiles/kernel_trace_summary.py "$OUT/trace_headline" "$OUT/bench_under_rocprof.log" "void fused_fc_dual_kernel" > "$OUT/rocprof_kernel_trace_headline.md" 2>&1
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d "$OUT/trace_float" -o t -- python "$REPO/bench.py" --input float --steps 10 --warmup 3 --no-extra --no-cpu > "$OUT/bench_float_under_rocprof.log" 2>&1 )
python profiles/kernel_trace_summary.py "$OUT/trace_float" "$OUT/bench_float_under_rocprof.log" "void fused_fc_f32_kernel" > "$OUT/rocprof_kernel_trace_float.md" 2>&1
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d "$OUT/trace_cnn" -o t -- python "$REPO/bench.py" --model cnn_64 --images 10000000 --steps 10 --warmup 2 --no-extra --no-cpu > "$OUT/bench_cnn_under_rocprof.log" 2>&1 )
python profiles/kernel_trace_summary.py "$OUT/trace_cnn" "$OUT/bench_cnn_under_rocprof.log" "void cnn_li_fused_pipe_kernel" > "$OUT/rocprof_kernel_trace_cnn.md" 2>&1
PMC_TRAFFIC=1 timeout 600 bash profiles/pmc_kernel.sh ${TAG}_dual > "$OUT/pmc_dual.md" 2>&1
PMC_TRAFFIC=1 timeout 600 bash profiles/pmc_kernel.sh ${TAG}_f32 --input float --images 50000000 > "$OUT/pmc_f32.md" 2>&1
timeout 400 bash profiles/pmc_kernel.sh ${TAG}_f32_tern --input float --model tern_96 --images 50000000 > "$OUT/pmc_f32_tern.md" 2>&1
PMC_TRAFFIC=1 timeout 600 bash profiles/pmc_kernel.sh ${TAG}_cnn_fused --model cnn_64 --images 4194304 > "$OUT/pmc_cnn_fused.md" 2>&1
timeout 400 bash profiles/pmc_kernel.sh ${TAG}_cnn_fused16 --model mcu_cnn_16 --images 4194304 > "$OUT/pmc_cnn_fused16.md" 2>&1
timeout 400 bash profiles/pmc_kernel.sh ${TAG}_cnn_fused48 --model mcu_cnn_48 --images 4194304 > "$OUT/pmc_cnn_fused48.md" 2>&1
timeout 400 bash profiles/pmc_kernel.sh ${TAG}_cnn_f32 --model cnn_64 --images 4194304 --input float --cnn-variant 3 > "$OUT/pmc_cnn_f32.md" 2>&1
PMC_TRAFFIC=1 timeout 600 bash profiles/pmc_kernel.sh ${TAG}_cnn_two --model cnn_64 --images 4194304 --cnn-variant 4 > "$OUT/pmc_cnn_two.md" 2>&1
for t in dual f32 f32_tern cnn_fused cnn_fused16 cnn_fused48 cnn_f32 cnn_two; do cp "gpurun_out/pmc_${TAG}_$t/table.json" "$OUT/table_$t.json" 2>/dev/null; done
tail -c 300 "$OUT/bench.json"; cat "$OUT/rocprof_kernel_trace_headline.md" "$OUT/rocprof_kernel_trace_float.md" "$OUT/rocprof_kernel_trace_cnn.md"
