#!/usr/bin/env python3
"""Round 3: socket power and shader clock under every kernel of the bench line (GPU box; amdgpu hwmon sampled every ~3 ms on the
card that is HIP device 0 - the sampler of profiles/power_telemetry.py).  Each arm launches its kernel back to back for
`seconds` on the same resident 1e8 images: the specialised headline kernel, the box's plain read of the images
(bnm_stream_read_device), the generic kernel on the headline / ternary 96 / 12 KB binary models, the ternary ALU kernel, the CNN.
Question it answers: at which clock does each kernel run, i.e. which of them are bound by the 1400 W cap rather than by a pipe.
  usage: python profiles/power_telemetry_r03.py [seconds per arm]"""
import json
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bitnetmcu_amd as b  # noqa: E402
from power_telemetry import Sampler, arm  # noqa: E402


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 2.5
    n = int(os.environ.get("N", 100_000_000))
    imgs = torch.empty((n, 256), dtype=torch.int8, device="cuda")
    b.synth.fill_device(imgs)
    cls = torch.empty(n, dtype=torch.int32, device="cuda")
    sink = torch.zeros(1, dtype=torch.int32, device="cuda")
    res = {"N": n, "seconds_per_arm": seconds}
    idle = Sampler()
    th = threading.Thread(target=idle.run)
    th.start()
    time.sleep(1.0)
    idle.stop = True
    th.join()
    res["idle"] = idle.summary()

    def brief(r):
        t = r["telemetry"]
        ours = [v for k, v in t.get("cards", {}).items() if k.endswith("(ours)")]
        o = ours[0] if ours else {}
        return {"ms_mean": round(r["ms_mean"], 4), "ms_last_tenth": round(r["ms_last_tenth"], 4), "launches": r["launches"],
                "power_W": o.get("power_W", {}).get("second_half_mean"), "power_cap_W": o.get("power_cap_W", {}).get("mean"),
                "sclk_GHz": o.get("sclk_GHz", {}).get("second_half_mean"), "hbm_temp_C": o.get("temp_hbm_C", {}).get("max")}

    res["plain_read_of_the_images"] = brief(arm("stream", lambda: b.synth.stream_read_device(imgs, sink), seconds))
    for label, name, variant, path, count in (("headline_dual_kernel", "fc_4bitsym_64", -1, 0, n),
                                              ("generic_kernel_fc64", "fc_4bitsym_64", 4, 0, n),
                                              ("generic_kernel_8bit_40_32_32", "doc12k_8bit", -1, 0, n),
                                              ("generic_kernel_ternary_96", "tern_96", -1, 0, n),
                                              ("generic_kernel_binary_160", "doc12k_binary", -1, 0, n),
                                              ("ternary_alu_kernel_96", "tern_96", -1, b.PATH_TERNARY_ALU, n),
                                              ("cnn_64", "cnn_64", -1, 0, min(n, 10_000_000))):
        ctx = b.Context(b.Model.from_zoo(name))
        if path:
            ctx.set_path(path)
        if variant >= 0:
            ctx.set_tuning(variant=variant)
        x, c = imgs[:count], cls[:count]
        res[label] = brief(arm(label, lambda: ctx.infer_device(x, c), seconds))
        res[label]["images"] = count
        ctx.close()
        print(label, json.dumps(res[label]), flush=True)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
