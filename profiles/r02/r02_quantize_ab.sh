#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for r in 1 2; do
  for lib in libbitnetmcu_hip_old.so libbitnetmcu_hip.so; do
    BNM_LIBRARY=bitnetmcu_amd/$lib timeout 200 python profiles/quantize_bench.py 2>&1 | grep -v amdgpu.ids
  done
done | tee gpurun_out/quantize_bench.log
timeout 300 python -m pytest tests -m gpu -x -q -k "quantiz" 2>&1 | tail -2
