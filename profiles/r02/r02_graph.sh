#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 300 python profiles/graph_replay.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/graph_replay.log | head -8
timeout 300 python profiles/host_paths.py > gpurun_out/host_paths.json 2>&1; tail -c 1200 gpurun_out/host_paths.json
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_small_batch.log 2>&1; tail -3 gpurun_out/pytest_gpu_small_batch.log
