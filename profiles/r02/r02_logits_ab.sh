#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
BNM_LIBRARY=bitnetmcu_amd/libbitnetmcu_hip_diag.so timeout 600 python profiles/logits_store_ab.py > gpurun_out/logits_store_ab.log 2>&1
grep -v '^{' gpurun_out/logits_store_ab.log | tail -8
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/logits_pytest.log 2>&1; tail -3 gpurun_out/logits_pytest.log
