#!/bin/bash
# The CNN part of profiles/r05_evidence.sh for the pipelined one-kernel form (cnn_li_fused_pipe_kernel), ON THE GPU BOX from the repo
# root: the default bench line, rocprofv3 --kernel-trace --stats of the CNN command, counter passes (each block in its own run,
# FETCH_SIZE / WRITE_SIZE in passes of their own) on the 64- / 16- / 48-channel models, the float form, and the four-wave form
# (variant 6) for comparison.  Output under gpurun_out/<tag>/.
set -u
TAG=${1:-r05w}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
timeout 600 python bench.py --steps 20 --warmup 3 > "$OUT/bench.json" 2> "$OUT/bench.err"
export TMPDIR=/tmp
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d "$OUT/trace_cnn" -o t -- python "$REPO/bench.py" --model cnn_64 --images 10000000 --steps 10 --warmup 2 --no-extra --no-cpu > "$OUT/bench_cnn_under_rocprof.log" 2>&1 )
python profiles/kernel_trace_summary.py "$OUT/trace_cnn" "$OUT/bench_cnn_under_rocprof.log" "void cnn_li_fused_pipe_kernel" > "$OUT/rocprof_kernel_trace_cnn.md" 2>&1
PMC_TRAFFIC=1 timeout 600 bash profiles/pmc_kernel.sh ${TAG}_cnn_pipe --model cnn_64 --images 4194304 > "$OUT/pmc_cnn_pipe.md" 2>&1
timeout 400 bash profiles/pmc_kernel.sh ${TAG}_cnn_pipe16 --model mcu_cnn_16 --images 4194304 > "$OUT/pmc_cnn_pipe16.md" 2>&1
timeout 400 bash profiles/pmc_kernel.sh ${TAG}_cnn_pipe48 --model mcu_cnn_48 --images 4194304 > "$OUT/pmc_cnn_pipe48.md" 2>&1
timeout 400 bash profiles/pmc_kernel.sh ${TAG}_cnn_pipe_f32 --model cnn_64 --images 4194304 --input float --cnn-variant 3 > "$OUT/pmc_cnn_pipe_f32.md" 2>&1
for t in cnn_pipe cnn_pipe16 cnn_pipe48 cnn_pipe_f32; do cp "gpurun_out/pmc_${TAG}_$t/table.json" "$OUT/table_$t.json" 2>/dev/null; done
tail -c 300 "$OUT/bench.json"; cat "$OUT/rocprof_kernel_trace_cnn.md"
