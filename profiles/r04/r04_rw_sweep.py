#!/usr/bin/env python3
"""Round 4: what mixing 13 % writes into the image stream costs with no arithmetic (bnm_stream_rw_device: 32-image tiles read, 44
bytes per image written) over the probe's shapes - tiles per batch, waves per SIMD, nontemporal / plain stores - next to the plain
read of the same images.  The yardstick of bench.py's ids + logits row.  usage (GPU box): python profiles/r04_rw_sweep.py"""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
import bitnetmcu_amd as b
n = 100_000_000
x = torch.empty((n, 256), dtype=torch.int8, device="cuda")
b.synth.fill_device(x, first=0, dist=0)
out = torch.empty((n * 44 + 64) // 4, dtype=torch.int32, device="cuda")
def t(f, reps=5):
    for _ in range(2): f()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for k in range(reps):
        f(); ev[k + 1].record()
    torch.cuda.synchronize()
    return float(np.median([ev[k].elapsed_time(ev[k + 1]) for k in range(reps)]))
sink = torch.zeros(1, dtype=torch.int32, device="cuda")
rd = t(lambda: b.synth.stream_read_device(x, sink))
print("plain read", round(rd, 3), "ms", round(n * 256 / rd / 1e6), "GB/s", flush=True)
for ob in (44,):
    for wps in (0, 1, 2, 4, 6):
        for batch in (1, 2, 4, 8):
            for plain in (0, 1):
                mode = batch + 16 * plain + 32 * wps
                ms = t(lambda: b.synth.stream_rw_device(x, out, ob, mode))
                print(f"out {ob} B  waves/SIMD {wps + 2}  batch {batch}  {'plain' if plain else 'nt'}: {ms:.3f} ms  {n * (256 + ob) / ms / 1e6:.0f} GB/s  x{ms / (rd * (256 + ob) / 256):.3f} byte-proportional", flush=True)
