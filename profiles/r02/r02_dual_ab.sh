#!/bin/bash
# Headline kernel A/B (round 2): dual-tile loop with a fixed stride per wave (variant 3) vs one 8-wave workgroup per CU taking
# pairs from a counter in LDS (variant 5); parity first, then alternating 1e8-image timings on the same box, then the per-wave
# loop-time distribution of both (diagnostic build).
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden_all_paths or random_models or ragged or synthetic" 2>&1 | tail -8 > gpurun_out/dual_ab_pytest.log
cat gpurun_out/dual_ab_pytest.log
for v in ${DUAL_VARIANTS:-3 5 3 5 3 5}; do
  timeout 300 python bench.py --variant $v --no-extra --no-cpu --steps 20 --warmup 3 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fused variant $v', d['value'], d['ms_per_step'], d['verified_vs_oracle'], d['digest'], d['roofline']['frac'])" | tee -a gpurun_out/dual_ab.log
done
for v in 3 5; do
  VARIANT=$v BNM_LIBRARY=bitnetmcu_amd/libbitnetmcu_hip_timing.so python profiles/wait_timing.py > gpurun_out/wait_timing_variant$v.json 2>&1
  python - $v <<'PY'
import json, sys
t = open(f"gpurun_out/wait_timing_variant{sys.argv[1]}.json").read()
d = json.loads(t[t.index("{"):])
print("variant", sys.argv[1], {k: d[k] for k in ("ms", "loop_cycles_percentiles_1_10_25_50_75_90_99_100", "iterations_percentiles_1_50_99")})
PY
done
