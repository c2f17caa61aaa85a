#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_final8.log 2>&1; tail -2 gpurun_out/pytest_gpu_final8.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/smoke_final8.log
