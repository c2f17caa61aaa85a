#!/bin/bash
# One box, one call: the slow-state probe (profiles/slow_state_probe.py), then the default bench line on the same box.
cd "$GRAFT_REPO_ROOT" || exit 1
BNM_LIBRARY=bitnetmcu_amd/libbitnetmcu_hip_diag.so timeout 600 python profiles/slow_state_probe.py > gpurun_out/slow_state_probe.log 2>&1
grep -v '^{' gpurun_out/slow_state_probe.log | cut -c1-600
timeout 300 python bench.py --no-extra --no-cpu > gpurun_out/slow_state_bench.json 2>&1
python - <<'PY'
import json
t=open("gpurun_out/slow_state_bench.json").read(); d=json.loads(t[t.rindex("\n{")+1:] if "\n{" in t else t[t.index("{"):])
print("bench --no-extra --no-cpu: ms_per_step", d["ms_per_step"], "frac", d["roofline"]["frac"])
PY
