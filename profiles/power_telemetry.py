#!/usr/bin/env python3
"""Power / clock telemetry while the headline kernel runs (settle "power-throttled issue" with evidence).

Run on the GPU box.  With the diagnostic library it also measures the cache-resident-source run and the plain stream:
    python bitnetmcu_amd/build.py --diag        (before gpurun; the .so travels)
    BNM_LIBRARY=bitnetmcu_amd/libbitnetmcu_hip_diag.so python profiles/power_telemetry.py [seconds per arm]

For each arm the kernel is launched back to back for `seconds` while a host thread samples, every few milliseconds, the
amdgpu hwmon / sysfs files of the device (average socket power, power cap, shader clock, memory clock, temperature) —
or, if sysfs is not visible in the container, `rocm-smi` / `amd-smi` at whatever rate they answer.  Prints one JSON object:
per arm the launch-time statistics (first launches vs steady state) and the telemetry statistics."""
import glob
import json
import os
import subprocess
import sys
import threading
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import bitnetmcu_amd as b  # noqa: E402
from bitnetmcu_amd import _lib as L  # noqa: E402
import util  # noqa: E402


def read_int(path):
    try:
        with open(path) as f:
            return int(f.read().split()[0])
    except Exception:
        return None


def our_pci_bus_id():
    """PCI address of HIP device 0 (the sysfs tree shows every GPU of the host, the container sees one of them)"""
    try:
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        buf = ctypes.create_string_buffer(64)
        if hip.hipDeviceGetPCIBusId(buf, 64, 0) == 0:
            return buf.value.decode().lower()
    except Exception:
        pass
    return None


class Sampler:
    """samples EVERY amdgpu card's hwmon files; `ours` marks the card whose PCI address is HIP device 0's"""

    def __init__(self):
        self.cards = {}
        want = our_pci_bus_id()
        self.ours = None
        for card in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
            hw = glob.glob(os.path.join(card, "hwmon", "hwmon*"))
            if not hw:
                continue
            h = hw[0]
            files = {}
            cand = {"power_uW": ["power1_average", "power1_input"], "power_cap_uW": ["power1_cap"], "sclk_Hz": ["freq1_input"],
                    "mclk_Hz": ["freq2_input"], "temp_mC": ["temp1_input"], "temp_hbm_mC": ["temp3_input"]}
            for k, names in cand.items():
                for nm in names:
                    p = os.path.join(h, nm)
                    if read_int(p) is not None:
                        files[k] = p
                        break
            if files:
                addr = os.path.basename(os.path.realpath(card)).lower()
                name = os.path.basename(os.path.dirname(card)) + "@" + addr
                self.cards[name] = files
                if want and addr == want:
                    self.ours = name
        self.files = self.cards.get(self.ours) if self.ours else None
        self.mode = "sysfs" if self.cards else None
        if not self.mode:
            for tool, args in (("rocm-smi", ["--showpower", "--showclocks", "--json"]), ("amd-smi", ["metric", "-p", "-c", "--json"])):
                try:
                    r = subprocess.run([tool] + args, capture_output=True, text=True, timeout=20)
                    if r.returncode == 0 and r.stdout.strip():
                        self.mode, self.cmd = tool, [tool] + args
                        break
                except Exception:
                    pass
        self.samples = []
        self.stop = False

    def run(self):
        t0 = time.perf_counter()
        while not self.stop:
            t = time.perf_counter() - t0
            if self.mode == "sysfs":
                self.samples.append((t, {c: {k: read_int(p) for k, p in f.items()} for c, f in self.cards.items()}))
                time.sleep(0.002)
            elif self.mode:
                try:
                    r = subprocess.run(self.cmd, capture_output=True, text=True, timeout=20)
                    self.samples.append((t, {"raw": r.stdout.strip()[:2000]}))
                except Exception:
                    pass
            else:
                time.sleep(0.05)

    def summary(self):
        if self.mode != "sysfs":
            return {"mode": self.mode, "n_samples": len(self.samples), "last": self.samples[-1][1] if self.samples else None,
                    "first": self.samples[0][1] if self.samples else None}
        out = {"mode": "sysfs", "n_samples": len(self.samples), "our_card": self.ours, "hip_device_0_pci": our_pci_bus_id(),
               "sample_period_ms": 1e3 * (self.samples[-1][0] - self.samples[0][0]) / max(1, len(self.samples) - 1) if len(self.samples) > 1 else None,
               "cards": {}}
        for card, files in self.cards.items():
            o = {}
            for k in files:
                v = np.array([s[1][card][k] for s in self.samples if s[1][card].get(k) is not None], dtype=np.float64)
                if len(v):
                    scale = 1e-6 if k.endswith("_uW") else 1e-9 if k.endswith("_Hz") else 1e-3
                    unit = "W" if k.endswith("_uW") else "GHz" if k.endswith("_Hz") else "C"
                    half = v[len(v) // 2:]
                    o[k.rsplit("_", 1)[0] + "_" + unit] = {"mean": round(float(v.mean() * scale), 4), "max": round(float(v.max() * scale), 4),
                                                           "min": round(float(v.min() * scale), 4),
                                                           "second_half_mean": round(float(half.mean() * scale), 4)}
            out["cards"][card + (" (ours)" if card == self.ours else "")] = o
        return out


def arm(name, launch, seconds):
    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    time.sleep(1.0)     # let the chip idle back down: every arm starts from the same state
    smp = Sampler()
    th = threading.Thread(target=smp.run)
    th.start()
    evs = []
    t_end = time.perf_counter() + seconds
    e_prev = torch.cuda.Event(enable_timing=True)
    e_prev.record()
    while time.perf_counter() < t_end:
        for _ in range(8):
            launch()
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            evs.append((e_prev, e))
            e_prev = e
        torch.cuda.synchronize()
    smp.stop = True
    th.join()
    ms = np.array([a.elapsed_time(c) for a, c in evs])
    k = max(1, len(ms) // 10)
    return {"launches": int(len(ms)), "ms_first_tenth": float(ms[:k].mean()), "ms_last_tenth": float(ms[-k:].mean()), "ms_mean": float(ms.mean()),
            "ms_min": float(ms.min()), "ms_max": float(ms.max()), "telemetry": smp.summary()}


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
    n = int(os.environ.get("N", 100_000_000))
    lib = b.load()
    model = util.load_golden_model("fc_4bitsym_64")
    imgs = torch.empty((n, 256), dtype=torch.int8, device="cuda")
    b.synth.fill_device(imgs)
    cls = torch.empty(n, dtype=torch.int32, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    res = {"N": n, "seconds_per_arm": seconds, "library": L.LIB_PATH}
    idle = Sampler()
    th = threading.Thread(target=idle.run)
    th.start()
    time.sleep(1.0)
    idle.stop = True
    th.join()
    res["idle"] = idle.summary()
    ctx = b.Context(model)
    default_variant = ctx.variant
    res["default_variant"] = default_variant
    res["fused_dual_hbm"] = arm("fused", lambda: ctx.infer_device(imgs, cls), seconds)       # the default kernel (= bench.py)
    for v in (3, 2, 4):          # dual-tile loop with a fixed stride, one tile per iteration, the generic kernel
        ctx.set_tuning(variant=v)
        res[f"fused_variant{v}_hbm"] = arm(f"fused{v}", lambda: ctx.infer_device(imgs, cls), seconds)
    ctx.set_tuning(variant=default_variant)
    if hasattr(lib, "bnm_diag_set_src_wrap"):
        L.check(lib, lib.bnm_diag_set_src_wrap(ctx._h, 256))
        res["fused_dual_cache_resident"] = arm("fused_wrap", lambda: ctx.infer_device(imgs, cls), seconds)
        L.check(lib, lib.bnm_diag_set_src_wrap(ctx._h, 0))
        out = cls
        res["plain_stream_mode0"] = arm("stream", lambda: L.check(lib, lib.bnm_diag_stream_device(imgs.data_ptr(), n, 0, 0, out.data_ptr(), s)), seconds)
    # the same kernel on all-zero images (data-dependent power)
    imgs.zero_()
    res["fused_dual_hbm_zero_images"] = arm("fused_zero", lambda: ctx.infer_device(imgs, cls), seconds)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
