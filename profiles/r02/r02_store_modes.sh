#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
BNM_LIBRARY=bitnetmcu_amd/libbitnetmcu_hip_diag.so timeout 600 python profiles/store_modes_ab.py > gpurun_out/store_modes_ab.log 2>&1
grep -v '^{' gpurun_out/store_modes_ab.log | tail -14
