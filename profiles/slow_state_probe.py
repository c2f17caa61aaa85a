#!/usr/bin/env python3
"""What makes the headline kernel run at 4.5 ms in one process and at 4.0 ms in another ON THE SAME BOX (r02/box_diag_r02.log)?

One process, a sequence of blocks, each = the same launch back to back (or interleaved with another) with the device's hwmon
sampled beside it.  Candidates separated here:
  * sustained back-to-back launches vs launches interleaved with a light kernel (power / thermal state),
  * the class-id stores (diagnostic flag: bit 63 of bnm_diag_set_src_wrap drops them),
  * where the buffers live (the image / class buffers are re-allocated between blocks, in different orders),
  * plain drift with time (block A is repeated at the end).
Needs the diagnostic library:  python bitnetmcu_amd/build.py --diag;
    BNM_LIBRARY=bitnetmcu_amd/libbitnetmcu_hip_diag.so python profiles/slow_state_probe.py
Prints one line per block as it finishes and a JSON object at the end."""
import json
import os
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "profiles"))
import bitnetmcu_amd as b                   # noqa: E402
from bitnetmcu_amd import _lib as L, synth  # noqa: E402
import util                                 # noqa: E402
from power_telemetry import Sampler         # noqa: E402

NO_STORE = 1 << 63


def telemetry(smp):
    if smp.mode != "sysfs" or not smp.files:
        return None
    out = {}
    for k in smp.files:
        v = np.array([s[1][smp.ours][k] for s in smp.samples if s[1][smp.ours].get(k) is not None], dtype=np.float64)
        if len(v):
            scale = 1e-6 if k.endswith("_uW") else 1e-9 if k.endswith("_Hz") else 1e-3
            out[k.rsplit("_", 1)[0]] = round(float(v.mean() * scale), 3)
    return out


def timed(launches):
    """launches: list of callables run in this order, each timed by its own pair of events"""
    ev = []
    for f in launches:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        f()
        e1.record()
        ev.append((e0, e1))
    torch.cuda.synchronize()
    return [a.elapsed_time(c) for a, c in ev]


def block(name, arms, pattern, res):
    """arms: {label: callable}; pattern: list of labels (one launch each)"""
    for f in arms.values():
        f()
    torch.cuda.synchronize()
    smp = Sampler()
    th = threading.Thread(target=smp.run)
    th.start()
    ms = timed([arms[k] for k in pattern])
    smp.stop = True
    th.join()
    o = {"telemetry": telemetry(smp)}
    for k in arms:
        v = np.array([m for m, p in zip(ms, pattern) if p == k])
        o[k] = {"median_ms": round(float(np.median(v)), 3), "min_ms": round(float(v.min()), 3), "first3": [round(float(x), 3) for x in v[:3]],
                "last3": [round(float(x), 3) for x in v[-3:]], "launches": int(len(v))}
    res[name] = o
    print(name, json.dumps(o), flush=True)


def main():
    lib = b.load()
    if not hasattr(lib, "bnm_diag_stream_device"):
        sys.exit("needs the diagnostic library (see the docstring)")
    n = int(os.environ.get("N", 100_000_000))
    reps = int(os.environ.get("REPS", 24))
    model = util.load_golden_model("fc_4bitsym_64")
    ctx = b.Context(model)
    s = torch.cuda.current_stream().cuda_stream
    res = {"N": n, "variant": ctx.variant}

    def alloc(order):
        bufs = {}
        for what in order:
            if what == "imgs":
                bufs["imgs"] = torch.empty((n, 256), dtype=torch.int8, device="cuda")
            elif what == "cls":
                bufs["cls"] = torch.empty(n, dtype=torch.int32, device="cuda")
            elif what == "out":
                bufs["out"] = torch.zeros(n, dtype=torch.int32, device="cuda")
            elif what == "pad":
                bufs["pad"] = torch.empty(3 * 1024 * 1024 + 4096, dtype=torch.int8, device="cuda")
        synth.fill_device(bufs["imgs"], first=0, dist=b.DIST_U)
        torch.cuda.synchronize()
        return bufs

    def arms_for(bufs):
        imgs, cls = bufs["imgs"], bufs["cls"]
        out = bufs.get("out", cls)

        def kernel():
            ctx.infer_device(imgs, cls)

        def kernel_no_store():
            L.check(lib, lib.bnm_diag_set_src_wrap(ctx._h, NO_STORE))
            ctx.infer_device(imgs, cls)
            L.check(lib, lib.bnm_diag_set_src_wrap(ctx._h, 0))

        def stream():
            L.check(lib, lib.bnm_diag_stream_device(imgs.data_ptr(), n, 0, 0, out.data_ptr(), s))

        def tile_loop():
            L.check(lib, lib.bnm_diag_stream_device(imgs.data_ptr(), n, 2, 0, out.data_ptr(), s))

        return {"kernel": kernel, "kernel_no_store": kernel_no_store, "stream": stream, "tile_loop": tile_loop}

    def pick(a, *names):
        return {k: a[k] for k in names}

    bufs = alloc(["imgs", "cls"])
    a = arms_for(bufs)
    block("A1_kernel_back_to_back", pick(a, "kernel"), ["kernel"] * reps, res)
    block("B_kernel_stream_5_5", pick(a, "kernel", "stream"), (["kernel"] * 5 + ["stream"] * 5) * 4, res)
    block("C_kernel_stream_1_1", pick(a, "kernel", "stream"), ["kernel", "stream"] * 12, res)
    block("A2_kernel_back_to_back", pick(a, "kernel"), ["kernel"] * reps, res)
    block("D_no_store_back_to_back", pick(a, "kernel_no_store"), ["kernel_no_store"] * reps, res)
    block("E_kernel_no_store_1_1", pick(a, "kernel", "kernel_no_store"), ["kernel", "kernel_no_store"] * 12, res)
    block("F_stream_back_to_back", pick(a, "stream"), ["stream"] * reps, res)
    block("G_tile_loop_back_to_back", pick(a, "tile_loop"), ["tile_loop"] * reps, res)
    for v in (3,):
        ctx.set_tuning(variant=v)
        block(f"H_variant{v}_store_vs_none", pick(a, "kernel", "kernel_no_store"), ["kernel", "kernel_no_store"] * 12, res)
    ctx.set_tuning(variant=res["variant"])
    # the same buffers again, allocated in another order / with other neighbours
    del a, bufs
    torch.cuda.empty_cache()
    bufs = alloc(["pad", "cls", "out", "imgs"])
    a = arms_for(bufs)
    block("I_realloc_cls_first", pick(a, "kernel"), ["kernel"] * reps, res)
    block("J_realloc_kernel_stream_5_5", pick(a, "kernel", "stream"), (["kernel"] * 5 + ["stream"] * 5) * 4, res)
    del a, bufs
    torch.cuda.empty_cache()
    bufs = alloc(["imgs", "cls", "out"])
    a = arms_for(bufs)
    block("K_realloc_as_ceiling_ab", pick(a, "kernel"), ["kernel"] * reps, res)
    time.sleep(1.0)
    block("A3_kernel_back_to_back_after_1s_idle", pick(a, "kernel"), ["kernel"] * reps, res)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
