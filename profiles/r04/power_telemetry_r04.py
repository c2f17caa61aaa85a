#!/usr/bin/env python3
"""Round 4: socket power and shader clock under the kernels this round added, next to the ones they are compared with (GPU box; the
sampler of profiles/power_telemetry.py: amdgpu hwmon of the card that is HIP device 0, every ~3 ms, each arm launched back to back
for `seconds` on the same resident images): the CNN's lane = image kernel against the channel kernel (64 / 16 / 48 channels), the
register-resident-weight kernel against the generic kernel on the ternary 96-96-96 model, the headline kernel and the plain read.
  usage: python profiles/power_telemetry_r04.py [seconds per arm]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bitnetmcu_amd as b  # noqa: E402
from power_telemetry import arm  # noqa: E402


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 2.5
    n = int(os.environ.get("N", 100_000_000))
    imgs = torch.empty((n, 256), dtype=torch.int8, device="cuda")
    b.synth.fill_device(imgs)
    cls = torch.empty(n, dtype=torch.int32, device="cuda")
    sink = torch.zeros(1, dtype=torch.int32, device="cuda")
    res = {"N": n, "seconds_per_arm": seconds}

    def brief(r):
        t = r["telemetry"]
        ours = [v for k, v in t.get("cards", {}).items() if k.endswith("(ours)")]
        o = ours[0] if ours else {}
        return {"ms_mean": round(r["ms_mean"], 4), "launches": r["launches"], "power_W": o.get("power_W", {}).get("second_half_mean"),
                "power_cap_W": o.get("power_cap_W", {}).get("mean"), "sclk_GHz": o.get("sclk_GHz", {}).get("second_half_mean")}

    res["plain_read_of_the_images"] = brief(arm("stream", lambda: b.synth.stream_read_device(imgs, sink), seconds))
    for label, name, variant, cnn_variant, count in (("headline_dual_kernel", "fc_4bitsym_64", -1, -1, n),
                                                     ("tern96_generic_kernel", "tern_96", 4, -1, n),
                                                     ("tern96_regw_kernel", "tern_96", 9, -1, n),
                                                     ("cnn_64_lane_image_kernel", "cnn_64", -1, 3, 10_000_000),
                                                     ("cnn_64_channel_kernel", "cnn_64", -1, 1, 10_000_000),
                                                     ("cnn_16_lane_image_kernel", "mcu_cnn_16", -1, 3, 10_000_000),
                                                     ("cnn_16_channel_kernel", "mcu_cnn_16", -1, 1, 10_000_000),
                                                     ("cnn_48_lane_image_kernel", "mcu_cnn_48", -1, 3, 10_000_000)):
        ctx = b.Context(b.Model.from_zoo(name))
        if variant >= 0:
            ctx.set_tuning(variant=variant)
        if cnn_variant >= 0:
            ctx.set_cnn_variant(cnn_variant)
        x, c = imgs[:count], cls[:count]
        res[label] = brief(arm(label, lambda: ctx.infer_device(x, c), seconds))
        res[label]["inferences_per_s"] = count / (res[label]["ms_mean"] * 1e-3)
        ctx.close()
    print(json.dumps(res))


if __name__ == "__main__":
    main()
