#!/bin/bash
# Round 6, the one-kernel convolution front of the QAT forward (ON THE GPU BOX from the repo root): CNNMNIST's forward three ways
# (whole module, front fused / layer by layer, FC stack), the kernel trace of the fused front at 1e6 images, its counter passes
# (each block in its own run; FETCH_SIZE / WRITE_SIZE in passes of their own) -> profiles/pmc_counters.json gains the entry
# bench.py's qat_cnn_forward row replays; then the default bench line.
set -u
TAG=${1:-r06u}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
python profiles/cnnmnist_forward_bench.py --rows 65536 > "$OUT/cnnmnist_forward_65536.json" 2> "$OUT/cnnmnist.err"
python profiles/cnnmnist_forward_bench.py --rows 1000000 --steps 5 > "$OUT/cnnmnist_forward_1000000.json" 2>> "$OUT/cnnmnist.err"
PMC_TRAFFIC=1 PMC_CMD="python $REPO/profiles/cnnmnist_forward_bench.py --rows 1000000 --steps 3 --front-only" \
  PMC_CMD_STATS="python $REPO/profiles/cnnmnist_forward_bench.py --rows 1000000 --steps 10 --front-only" \
  timeout 600 bash profiles/pmc_kernel.sh ${TAG}_qat_cnn > "$OUT/pmc_qat_cnn.md" 2>&1
cp "gpurun_out/pmc_${TAG}_qat_cnn/table.json" "$OUT/table_qat_cnn.json" 2>/dev/null
cp gpurun_out/pmc_${TAG}_qat_cnn/stats/*kernel_stats.csv "$OUT/rocprof_stats_qat_cnn.csv" 2>/dev/null
python profiles/make_counters_json.py $TAG qat_cnn_front_kernel=$OUT/table_qat_cnn.json:1000000 > "$OUT/make_counters.log" 2>&1
cp profiles/pmc_counters.json "$OUT/pmc_counters.json"
timeout 900 python bench.py --steps 20 --warmup 5 --full-json "$OUT/bench_full.json" > "$OUT/bench_stdout.txt" 2> "$OUT/bench.err"
tail -n 1 "$OUT/bench_stdout.txt" > "$OUT/bench_line.json"
head -c 1500 "$OUT/bench_line.json"; echo; cat "$OUT/cnnmnist_forward_1000000.json"; tail -3 "$OUT/make_counters.log"
