#!/bin/bash
# CNN front end A/B (round 2): parity, then 1e7-image timings on the same box: conv1-on-MFMA front end with dynamic image
# batches (1 = batches of 8, 100 + g = batches of g), with fixed shares per wave (2), and round 1's VALU kernel (0).
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dropin.py -x -q -m gpu -k "cnn or golden_all_paths or ragged or dropin" 2>&1 | tail -15 > gpurun_out/cnn_ab_pytest.log
cat gpurun_out/cnn_ab_pytest.log
for v in ${CNN_VARIANTS:-1 2 0 101 102 104 116 132 1}; do
  timeout 300 python bench.py --model cnn_64 --images 10000000 --cnn-variant $v --no-extra --no-cpu --steps 3 --warmup 1 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cnn variant $v', d['value'], d['ms_per_step'], d['verified_vs_oracle'], d['roofline']['kernel'])" | tee -a gpurun_out/cnn_ab.log
done
BNM_LIBRARY=bitnetmcu_amd/libbitnetmcu_hip_timing.so python profiles/cnn_wait_timing.py > gpurun_out/cnn_wait_timing_dynamic.json 2>&1
python - <<'PY'
import json
t = open("gpurun_out/cnn_wait_timing_dynamic.json").read()
d = json.loads(t[t.index("{"):])
print({k: d[k] for k in ("ms_front_plus_tail", "items_per_wave_median", "loop_cycles_percentiles_1_10_25_50_75_90_99", "head_share", "tile_share")})
PY
