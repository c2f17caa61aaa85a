#!/bin/bash
# Round 6, second counter visit (ON THE GPU BOX from the repo root): the one-kernel CNN's int8 binaries changed this round (a
# parameter for the float form's non-finite count), so their replayed entries no longer match the library - re-collected here for
# the 64-, 16- and 48-channel models in both forms; and the round-4 passes of the generic / ternary kernels (same binaries since,
# still valid by code hash) are refreshed on this round's box.  Then the default bench line once more: its
# rows replay the new entries.
set -u
TAG=${1:-r06y}
REPO=$(pwd); OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
PMC_TRAFFIC=1 timeout 600 bash profiles/pmc_kernel.sh ${TAG}_cnn_pipe --model cnn_64 --images 4194304 > "$OUT/pmc_cnn_pipe.md" 2>&1
timeout 400 bash profiles/pmc_kernel.sh ${TAG}_cnn_pipe16 --model mcu_cnn_16 --images 4194304 > "$OUT/pmc_cnn_pipe16.md" 2>&1
timeout 400 bash profiles/pmc_kernel.sh ${TAG}_cnn_pipe48 --model mcu_cnn_48 --images 4194304 > "$OUT/pmc_cnn_pipe48.md" 2>&1
timeout 400 bash profiles/pmc_kernel.sh ${TAG}_cnn_w4 --model cnn_64 --images 4194304 --cnn-variant 6 > "$OUT/pmc_cnn_w4.md" 2>&1
timeout 400 bash profiles/pmc_kernel.sh ${TAG}_cnn_w4_16 --model mcu_cnn_16 --images 4194304 --cnn-variant 6 > "$OUT/pmc_cnn_w4_16.md" 2>&1
timeout 400 bash profiles/pmc_kernel.sh ${TAG}_cnn_w4_48 --model mcu_cnn_48 --images 4194304 --cnn-variant 6 > "$OUT/pmc_cnn_w4_48.md" 2>&1
timeout 400 bash profiles/pmc_kernel.sh ${TAG}_fc_generic --variant 4 > "$OUT/pmc_fc_generic.md" 2>&1
timeout 400 bash profiles/pmc_kernel.sh ${TAG}_tern_generic --model tern_96 --images 20000000 > "$OUT/pmc_tern_generic.md" 2>&1
timeout 400 bash profiles/pmc_kernel.sh ${TAG}_binary160 --model doc12k_binary --images 20000000 > "$OUT/pmc_doc12k_binary.md" 2>&1
timeout 400 bash profiles/pmc_kernel.sh ${TAG}_tern_alu --model tern_96 --path 3 --images 20000000 > "$OUT/pmc_tern_alu.md" 2>&1
timeout 400 bash profiles/pmc_kernel.sh ${TAG}_tern128_alu --model doc12k_ternary --path 3 --images 20000000 > "$OUT/pmc_tern128_alu.md" 2>&1
PMC_TRAFFIC=1 PMC_CMD="python $REPO/profiles/qat_model_bench.py --rows 10000000 --steps 3 --warmup 1" PMC_CMD_STATS="python $REPO/profiles/qat_model_bench.py --rows 10000000 --steps 10 --warmup 3" \
  timeout 600 bash profiles/pmc_kernel.sh ${TAG}_qat > "$OUT/pmc_qat.md" 2>&1
for t in qat cnn_pipe cnn_pipe16 cnn_pipe48 cnn_w4 cnn_w4_16 cnn_w4_48 fc_generic tern_generic binary160 tern_alu tern128_alu; do cp "gpurun_out/pmc_${TAG}_$t/table.json" "$OUT/table_$t.json" 2>/dev/null; done
# the in-tree library is the one the counters were collected from: stamp and write profiles/pmc_counters.json HERE, then the bench line replays it
python profiles/make_counters_json.py $TAG \
  cnn_li_fused_pipe_kernel=$OUT/table_cnn_pipe.json:4194304 cnn_li_fused_pipe_kernel@cnn_64=$OUT/table_cnn_pipe.json:4194304 \
  cnn_li_fused_pipe_kernel@mcu_cnn_16=$OUT/table_cnn_pipe16.json:4194304 cnn_li_fused_pipe_kernel@mcu_cnn_48=$OUT/table_cnn_pipe48.json:4194304 \
  cnn_li_fused_kernel=$OUT/table_cnn_w4.json:4194304 cnn_li_fused_kernel@cnn_64=$OUT/table_cnn_w4.json:4194304 \
  cnn_li_fused_kernel@mcu_cnn_16=$OUT/table_cnn_w4_16.json:4194304 cnn_li_fused_kernel@mcu_cnn_48=$OUT/table_cnn_w4_48.json:4194304 \
  fused_fc_generic_kernel=$OUT/table_fc_generic.json:100000000 fused_fc_generic_kernel@tern_96=$OUT/table_tern_generic.json:20000000 \
  fused_fc_generic_kernel@doc12k_binary=$OUT/table_binary160.json:20000000 \
  ternary_stream_kernel=$OUT/table_tern_alu.json:20000000 ternary_stream_kernel@doc12k_ternary=$OUT/table_tern128_alu.json:20000000 \
  qat_fc_model_fwd_kernel=$OUT/table_qat.json:10000000 > "$OUT/make_counters.log" 2>&1
cp profiles/pmc_counters.json "$OUT/pmc_counters.json"
timeout 600 python bench.py --steps 20 --warmup 5 --full-json "$OUT/bench_full.json" > "$OUT/bench_stdout.txt" 2> "$OUT/bench.err"
tail -n 1 "$OUT/bench_stdout.txt" > "$OUT/bench_line.json"
head -c 1200 "$OUT/bench_line.json"; echo; tail -5 "$OUT/make_counters.log"
