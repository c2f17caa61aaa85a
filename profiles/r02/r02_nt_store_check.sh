#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
echo "== placeholder + final, both nontemporal (the build that failed the single-pair test)"
BNM_LIBRARY=bitnetmcu_amd/libbitnetmcu_hip_ntplaceholder.so timeout 300 python profiles/nt_store_order_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/nt_store_order_placeholder.log
echo "== product build (first store masked off)"
timeout 300 python profiles/nt_store_order_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/nt_store_order_product.log
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/nt_pytest.log 2>&1; tail -3 gpurun_out/nt_pytest.log
bash profiles/r02_store_modes.sh
timeout 300 python bench.py --no-extra --no-cpu > gpurun_out/nt_bench.json 2>&1; tail -c 1500 gpurun_out/nt_bench.json | cut -c1-300
