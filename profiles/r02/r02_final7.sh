#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/smoke_final7.log
timeout 400 python bench.py > gpurun_out/bench_final7.json 2> gpurun_out/bench_final7.err
python - <<'PY'
import json
t = open("gpurun_out/bench_final7.json").read(); d = json.loads(t[t.index('{"metric'):])
print("bench.py:", d["ms_per_step"], d["value"], d["roofline"]["frac"], d.get("digest"), d.get("verified_vs_oracle"))
for k, v in d["extra_configs"].items(): print("  ", k, v.get("value"), v.get("verified_vs_oracle"))
PY
