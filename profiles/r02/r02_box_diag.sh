#!/bin/bash
# One box, one call: where does the headline kernel stand against the box's own streaming rates?  (diagnostic library)
cd "$GRAFT_REPO_ROOT" || exit 1
export BNM_LIBRARY=bitnetmcu_amd/libbitnetmcu_hip_diag.so
python profiles/ceiling_ab.py > gpurun_out/boxdiag_ceiling.json 2>&1
python profiles/stream_ceiling.py > gpurun_out/boxdiag_stream.json 2>&1
python - <<'PY'
import json
t=open("gpurun_out/boxdiag_ceiling.json").read(); d=json.loads(t[t.index("{"):])
print("ceiling_ab: stream", round(d["plain_stream"]["median_ms"],3), "kernel", round(d["kernel"]["median_ms"],3))
t=open("gpurun_out/boxdiag_stream.json").read(); d=json.loads(t[t.index("{"):])
for k,v in d.items(): print("stream_ceiling", k, round(v["ms"],3), round(v["GB/s"]))
PY
python - <<'PY'
# compute-only time of the default kernel (cache-resident source) and zero-image time, same process
import ctypes as C, sys, os, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import bitnetmcu_amd as b
from bitnetmcu_amd import synth, _lib as L
import util
lib = b.load(); n = 100_000_000
model = util.load_golden_model("fc_4bitsym_64")
imgs = torch.empty((n, 256), dtype=torch.int8, device="cuda"); synth.fill_device(imgs, first=0, dist=b.DIST_U)
cls = torch.empty(n, dtype=torch.int32, device="cuda")
def t(ctx, reps=20):
    for _ in range(3): ctx.infer_device(imgs, cls)
    ms=[]
    for _ in range(reps):
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record(); ctx.infer_device(imgs, cls); e1.record(); e1.synchronize(); ms.append(e0.elapsed_time(e1))
    return round(float(np.median(ms)),3)
for v in (6, 3, 2):
    ctx = b.Context(model); ctx.set_tuning(variant=v)
    hbm = t(ctx)
    L.check(lib, lib.bnm_diag_set_src_wrap(ctx._h, 256)); wrap = t(ctx); L.check(lib, lib.bnm_diag_set_src_wrap(ctx._h, 0))
    print("variant", v, "HBM source", hbm, "cache-resident source", wrap)
    ctx.close()
imgs.zero_(); ctx = b.Context(model); print("variant 6 zero images", t(ctx))
PY
