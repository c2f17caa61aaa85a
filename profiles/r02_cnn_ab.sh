#!/bin/bash
# CNN front end A/B (round 2): parity, then 1e7-image timings of the MFMA front end vs round 1's VALU kernel on the same box.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dropin.py -x -q -m gpu -k "cnn or golden_all_paths or ragged or dropin" 2>&1 | tail -15 > gpurun_out/cnn_ab_pytest.log
cat gpurun_out/cnn_ab_pytest.log
for v in 1 0 1; do
  timeout 300 python bench.py --model cnn_64 --images 10000000 --cnn-variant $v --no-extra --no-cpu --steps 3 --warmup 1 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cnn variant $v', d['value'], d['ms_per_step'], d['verified_vs_oracle'], d['roofline']['kernel'])" | tee -a gpurun_out/cnn_ab.log
done
