#!/bin/bash
# Ternary ALU kernels A/B (round 2): parity of all variants, then 1e8-image timings on the same box.
# variants: 2 / 1 streamed weights, two / one image per lane, groups from the work counter; 12 / 11 the same with a fixed
# stride per wave; 0 round 1's kernel.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tern or random_models or ragged or golden_all_paths" 2>&1 | tail -15 > gpurun_out/tern_ab_pytest.log
cat gpurun_out/tern_ab_pytest.log
for v in ${TERN_VARIANTS:-2 12 1 11 0 2 12}; do
  timeout 300 python bench.py --model tern_96 --path 3 --ternary-variant $v --no-extra --no-cpu --steps 5 --warmup 2 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('tern variant $v', d['value'], d['ms_per_step'], d['verified_vs_oracle'], d['roofline']['kernel'])" | tee -a gpurun_out/tern_ab.log
done
