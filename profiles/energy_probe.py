#!/usr/bin/env python3
"""What do the kernels' instructions cost in ENERGY?  (GPU box, diagnostic library: python bitnetmcu_amd/build.py --diag;
BNM_LIBRARY=bitnetmcu_amd/libbitnetmcu_hip_diag.so python profiles/energy_probe.py)

Every FC kernel of the bench line runs the socket at its 1400 W cap (profiles/r03/power_telemetry_r03.json), so their time is
energy per image / 1400 W.  This probe measures the pieces: socket power (amdgpu hwmon, sampler of power_telemetry.py) while the
diagnostic library's pipe probes run back to back with no memory traffic - 26 MFMAs per tile (mode 5), ~400 VALU per tile
(mode 6), both (mode 7) - and while the plain read of 25.6 GB runs, each for `seconds`.  (power - idle) / rate = joules per
v_mfma_i32_32x32x32_i8, per VALU wave instruction, per byte read.  Prints one JSON object."""
import json
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bitnetmcu_amd as b  # noqa: E402
from bitnetmcu_amd import _lib as L  # noqa: E402
from power_telemetry import Sampler, arm  # noqa: E402


def ours(t):
    c = [v for k, v in t.get("cards", {}).items() if k.endswith("(ours)")]
    return c[0] if c else {}


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 2.5
    lib = b.load()
    if not hasattr(lib, "bnm_diag_stream_device"):
        sys.exit("needs the diagnostic library (build.py --diag, BNM_LIBRARY=...)")
    out = torch.zeros(1024, dtype=torch.int32, device="cuda")
    dummy = torch.zeros(256, dtype=torch.int8, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    idle = Sampler()
    th = threading.Thread(target=idle.run)
    th.start()
    time.sleep(1.5)
    idle.stop = True
    th.join()
    p_idle = ours(idle.summary()).get("power_W", {}).get("second_half_mean")
    res = {"idle_W": p_idle, "seconds_per_arm": seconds}
    tiles = 1526 * 4          # tiles per wave and launch (2048 waves): ~5 ms per launch
    waves = 2048
    for mode, name, per_tile in ((5, "mfma_only", {"mfma": 26}), (6, "valu_only", {"valu": 400}), (7, "both", {"mfma": 26, "valu": 400})):
        r = arm(name, lambda: L.check(lib, lib.bnm_diag_stream_device(dummy.data_ptr(), tiles, mode, 0, out.data_ptr(), s)), seconds)
        o = ours(r["telemetry"])
        p, clk = o.get("power_W", {}).get("second_half_mean"), o.get("sclk_GHz", {}).get("second_half_mean")
        e = {"ms_per_launch": r["ms_last_tenth"], "power_W": p, "sclk_GHz": clk}
        tiles_per_s = tiles * waves / (r["ms_last_tenth"] * 1e-3)
        for k, cnt in per_tile.items():
            e[k + "_per_s"] = cnt * tiles_per_s
        if p and p_idle and len(per_tile) == 1:
            (k, cnt), = per_tile.items()
            e["nJ_per_" + k + "_wave_instruction"] = (p - p_idle) / (cnt * tiles_per_s) * 1e9
        res[name] = e
    n = 100_000_000
    imgs = torch.empty((n, 256), dtype=torch.int8, device="cuda")
    b.synth.fill_device(imgs)
    sink = torch.zeros(1, dtype=torch.int32, device="cuda")
    r = arm("read", lambda: b.synth.stream_read_device(imgs, sink), seconds)
    o = ours(r["telemetry"])
    p = o.get("power_W", {}).get("second_half_mean")
    res["plain_read"] = {"ms_per_launch": r["ms_last_tenth"], "power_W": p, "sclk_GHz": o.get("sclk_GHz", {}).get("second_half_mean"),
                         "pJ_per_byte": (p - p_idle) / (n * 256 / (r["ms_last_tenth"] * 1e-3)) * 1e12 if p and p_idle else None}
    # energy budget of the bench line's FC models at the cap: bytes + MFMAs + VALU per image from the counters
    if "nJ_per_mfma_wave_instruction" in res["mfma_only"] and "nJ_per_valu_wave_instruction" in res["valu_only"] and res["plain_read"]["pJ_per_byte"]:
        jm, jv, jb = res["mfma_only"]["nJ_per_mfma_wave_instruction"], res["valu_only"]["nJ_per_valu_wave_instruction"], res["plain_read"]["pJ_per_byte"] * 1e-3
        cap = 1400.0
        budget = {}
        for name, (mfma_img, valu_img) in {"headline fc64 (dual kernel)": (26 / 32, 10.41 - 26 / 32), "generic fc64": (26 / 32, 11.0 - 26 / 32),
                                           "ternary 96-96-96 (generic)": (45 / 32, 15.7 - 45 / 32), "binary 160-160-160 (generic)": (95 / 32, 25.65 - 95 / 32)}.items():
            nj = p_idle / cap * 0 + 260 * jb + mfma_img * jm + valu_img * jv
            budget[name] = {"nJ_per_image_above_idle": nj, "ms_per_1e8_at_the_cap": nj * 1e-9 * 1e8 / (cap - p_idle) * 1e3}
        res["energy_budget_at_1400W"] = budget
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
