#!/bin/bash
# another box: the default kernel against the box's plain stream (ceiling_ab.py), then the default bench line
cd "$GRAFT_REPO_ROOT" || exit 1
BNM_LIBRARY=bitnetmcu_amd/libbitnetmcu_hip_diag.so timeout 300 python profiles/ceiling_ab.py > gpurun_out/ceiling_ab_final5.json 2>&1
grep -v amdgpu.ids gpurun_out/ceiling_ab_final5.json
timeout 400 python bench.py > gpurun_out/bench_final5b.json 2> gpurun_out/bench_final5b.err
python - <<'PY'
import json
t = open("gpurun_out/bench_final5b.json").read(); d = json.loads(t[t.index('{"metric'):])
print("bench.py:", d["ms_per_step"], d["value"], d["roofline"]["frac"], d.get("digest"))
for k, v in d["extra_configs"].items(): print("  ", k, v.get("value"), v.get("ms_per_step"))
PY
