#!/bin/bash
# Generic fused kernel with the device-wide scalar work counter (round 2): parity, then 1e8-image timings for batch sizes.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/generic_ab_pytest.log
cat gpurun_out/generic_ab_pytest.log
for wb in ${BATCHES:-8 1 2 4 16 32 8}; do
  timeout 300 python bench.py --variant 4 --work-batch $wb --no-extra --no-cpu --steps 10 --warmup 3 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fc64 generic batch $wb', d['value'], d['ms_per_step'], d['verified_vs_oracle'], d['digest'])" | tee -a gpurun_out/generic_ab.log
  timeout 300 python bench.py --model tern_96 --path 1 --work-batch $wb --no-extra --no-cpu --steps 10 --warmup 3 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('tern96 generic batch $wb', d['value'], d['ms_per_step'], d['verified_vs_oracle'])" | tee -a gpurun_out/generic_ab.log
done
