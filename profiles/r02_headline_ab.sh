#!/bin/bash
# Headline workload (fc_4bitsym_64, 1e8 Dist-U): the specialised dual-tile kernel (3), its CU-shared-counter form (5) and the
# generic kernel with the device-wide work counter (4), alternating on the same box.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for v in ${VARIANTS:-3 4 5 3 4 5 3 4}; do
  timeout 300 python bench.py --variant $v --no-extra --no-cpu --steps 20 --warmup 3 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('headline variant $v', d['value'], d['ms_per_step'], d['verified_vs_oracle'], d['digest'], round(d['roofline']['frac'],4), d['roofline']['kernel'])" | tee -a gpurun_out/headline_ab.log
done
