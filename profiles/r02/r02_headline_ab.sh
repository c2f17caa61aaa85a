#!/bin/bash
# Headline workload (fc_4bitsym_64, 1e8 Dist-U) on one box, alternating: the dual-tile kernel with a fixed stride (3), with the
# CU-shared LDS counter (5), with the device-wide scalar counter (6, batches of pairs: --work-batch), the generic kernel (4).
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
if [ "${PARITY:-1}" = "1" ]; then
  timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "golden_all_paths or random_models or ragged or full_size_properties" 2>&1 | tail -5 | tee gpurun_out/headline_ab_pytest.log
fi
for spec in ${SPECS:-3:0 5:0 6:16 6:8 6:32 4:0 3:0 5:0 6:16 6:64}; do
  v=${spec%%:*}; wb=${spec##*:}
  timeout 300 python bench.py --variant $v --work-batch $wb --no-extra --no-cpu --steps 20 --warmup 3 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('headline variant $v batch $wb', d['value'], d['ms_per_step'], d['verified_vs_oracle'], d['digest'], round(d['roofline']['frac'],4))" | tee -a gpurun_out/headline_ab.log
done
