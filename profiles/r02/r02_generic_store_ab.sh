#!/bin/bash
# generic kernel (variant 4): nontemporal class-id store (product) vs a build with the plain store, alternating processes on one box.
# The second library is built beforehand (in the build container) from the same sources with -DBNM_EXPERIMENT_PLAIN_CLASS_STORE:
#   python - <<'PY'
#   import sys, os; sys.path.insert(0, "bitnetmcu_amd"); import build as B
#   objs = B.objects(extra_flags=["-DBNM_EXPERIMENT_PLAIN_CLASS_STORE"], obj_dir=os.path.join(B.HERE, "_build_exp"))
#   B.run([B.HIPCC, f"--offload-arch={B.ARCH}", "-shared", "-o", os.path.join(B.HERE, "libbitnetmcu_hip_exp.so")] + objs)
#   PY
cd "$GRAFT_REPO_ROOT" || exit 1
for r in 1 2 3; do
  for lib in libbitnetmcu_hip.so libbitnetmcu_hip_exp.so; do
    BNM_LIBRARY=bitnetmcu_amd/$lib timeout 200 python bench.py --variant 4 --no-extra --no-cpu --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', round(d['ms_per_step'],3), d['roofline']['kernel'], d.get('verified_vs_oracle'))"
  done
done
