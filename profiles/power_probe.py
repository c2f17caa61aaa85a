#!/usr/bin/env python3
"""Is the headline kernel's last 15 % a power/clock effect?  Same kernel, same bytes, different switching activity:
Dist-U images vs all-zero images vs constant images.  (MI355X_MICROARCH.md 'DVFS give-back': zero-filled inputs ran
+19 % on an MFMA kernel at identical cycle counts.)  Run on the GPU box; prints ms per 1e8 images."""
import json
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import bitnetmcu_amd as b  # noqa: E402
import util  # noqa: E402

n = 100_000_000
ctx = b.Context(util.load_golden_model("fc_4bitsym_64"))
imgs = torch.empty((n, 256), dtype=torch.int8, device="cuda")
cls = torch.empty(n, dtype=torch.int32, device="cuda")
res = {}
for name in ("dist_u", "zeros", "const_m20", "dist_u_again"):
    if name.startswith("dist_u"):
        b.synth.fill_device(imgs)
    elif name == "zeros":
        imgs.zero_()
    else:
        imgs.fill_(-20)
    for _ in range(3):
        ctx.infer_device(imgs, cls)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ctx.infer_device(imgs, cls)
    e1.record()
    torch.cuda.synchronize()
    res[name] = e0.elapsed_time(e1) / 10
print(json.dumps(res))
