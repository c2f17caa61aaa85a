#!/bin/bash
# Round-6 evidence set, ON THE GPU BOX from the repo root, one box visit:
#   1. the default bench line (the driver's command: 20 timed launches after 5 warm-ups): the compact last line + bench_full.json;
#   2. rocprofv3 --kernel-trace --stats of the headline command, of the float-input workload, of the CNN and of the whole-model QAT
#      forward (profiles/qat_model_bench.py, 1e6 and 1e7 rows per call): per-launch durations next to the HIP-event numbers of THAT run;
#   3. counter passes, each block in its own run (never combined with other trace domains), FETCH_SIZE / WRITE_SIZE in passes of
#      their own: the kernels whose binaries changed this round (float-input FC kernel 2- and 4-tile class, the one-kernel CNN in its
#      float form) and the new one (qat_fc_model_fwd_kernel); the headline kernel once more (same binary as round 5: the replayed
#      entry stays valid, this pass re-measures its traffic on this round's box);
#   4. the per-image Inference() loop three ways (profiles/inference_latency.py) and the host-buffer rates (profiles/host_paths.py).
# Output under gpurun_out/<tag>/; profiles/make_counters_json.py turns the tables into profiles/pmc_*.json (stamped with the kernels'
# code hashes); copy what is to be judged into profiles/r06/.
set -u
TAG=${1:-r06z}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
timeout 600 python bench.py --steps 20 --warmup 5 --full-json "$OUT/bench_full.json" > "$OUT/bench_stdout.txt" 2> "$OUT/bench.err"
tail -n 1 "$OUT/bench_stdout.txt" > "$OUT/bench_line.json"
export TMPDIR=/tmp
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d "$OUT/trace_headline" -o t -- python "$REPO/bench.py" --steps 20 --warmup 5 --no-extra --no-cpu --full-json /tmp/full_h.json > "$OUT/bench_under_rocprof.log" 2>&1 )
python profiles/kernel_trace_summary.py "$OUT/trace_headline" "$OUT/bench_under_rocprof.log" "void fused_fc_dual_kernel" > "$OUT/rocprof_kernel_trace_headline.md" 2>&1
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d "$OUT/trace_float" -o t -- python "$REPO/bench.py" --input float --steps 10 --warmup 3 --no-extra --no-cpu --full-json /tmp/full_f.json > "$OUT/bench_float_under_rocprof.log" 2>&1 )
python profiles/kernel_trace_summary.py "$OUT/trace_float" "$OUT/bench_float_under_rocprof.log" "void fused_fc_f32_kernel" > "$OUT/rocprof_kernel_trace_float.md" 2>&1
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d "$OUT/trace_cnn" -o t -- python "$REPO/bench.py" --model cnn_64 --images 10000000 --steps 10 --warmup 2 --no-extra --no-cpu --full-json /tmp/full_c.json > "$OUT/bench_cnn_under_rocprof.log" 2>&1 )
python profiles/kernel_trace_summary.py "$OUT/trace_cnn" "$OUT/bench_cnn_under_rocprof.log" "void cnn_li_fused_pipe_kernel" > "$OUT/rocprof_kernel_trace_cnn.md" 2>&1
for ROWS in 1000000 10000000; do
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d "$OUT/trace_qat_$ROWS" -o t -- python "$REPO/profiles/qat_model_bench.py" --rows $ROWS --steps 20 --warmup 3 > "$OUT/qat_bench_$ROWS.log" 2>&1 )
  python profiles/qat_trace_summary.py "$OUT/trace_qat_$ROWS" "$OUT/qat_bench_$ROWS.log" 20 > "$OUT/rocprof_kernel_trace_qat_$ROWS.md" 2>&1
done
PMC_TRAFFIC=1 timeout 600 bash profiles/pmc_kernel.sh ${TAG}_dual > "$OUT/pmc_dual.md" 2>&1
PMC_TRAFFIC=1 timeout 600 bash profiles/pmc_kernel.sh ${TAG}_f32 --input float --images 50000000 > "$OUT/pmc_f32.md" 2>&1
timeout 400 bash profiles/pmc_kernel.sh ${TAG}_f32_tern --input float --model tern_96 --images 50000000 > "$OUT/pmc_f32_tern.md" 2>&1
timeout 400 bash profiles/pmc_kernel.sh ${TAG}_cnn_f32 --model cnn_64 --images 4194304 --input float --cnn-variant 3 > "$OUT/pmc_cnn_f32.md" 2>&1
PMC_TRAFFIC=1 PMC_CMD="python $REPO/profiles/qat_model_bench.py --rows 10000000 --steps 3 --warmup 1" PMC_CMD_STATS="python $REPO/profiles/qat_model_bench.py --rows 10000000 --steps 10 --warmup 3" \
  timeout 600 bash profiles/pmc_kernel.sh ${TAG}_qat > "$OUT/pmc_qat.md" 2>&1
for t in dual f32 f32_tern cnn_f32 qat; do cp "gpurun_out/pmc_${TAG}_$t/table.json" "$OUT/table_$t.json" 2>/dev/null; done
timeout 300 python profiles/inference_latency.py > "$OUT/inference_latency.json" 2> "$OUT/inference_latency.err"
N=4000000 timeout 300 python profiles/host_paths.py > "$OUT/host_paths.json" 2> /dev/null
head -c 600 "$OUT/bench_line.json"; echo; cat "$OUT/rocprof_kernel_trace_headline.md" "$OUT/rocprof_kernel_trace_float.md" "$OUT/rocprof_kernel_trace_cnn.md" "$OUT"/rocprof_kernel_trace_qat_*.md; cat "$OUT/inference_latency.json"
