#!/usr/bin/env python3
"""Per-channel bounds of the zoo CNNs' convolution outputs, from the weights alone (no GPU): can conv2's SECOND operand plane be
dropped for some channel?  conv1's sums reach at most 127 x (sum of positive weights) + 128 x |sum of negative weights| (inputs
-128..127), its outputs m1 = that >> 4 (BitNetMCU_inference.c:261-271: ReLU, then the shift); conv2's operand is that output as two
int8 planes (16 A + B) - ONE plane would do for a channel whose m1 fits a byte (<= 255: an unsigned byte less 128).
Prints a markdown table: per model the number of channels, min / median / max of m1, and how many channels fit one plane."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bitnetmcu_amd as b  # noqa: E402

MODELS = ("cnn_64", "mcu_cnn_16", "mcu_cnn_16small", "mcu_cnn_32", "mcu_cnn_48", "mcu_cnn_64", "mcu_cnn_letters")


def main():
    print("| model | channels | conv1 sum bound: min / median / max | m1 = bound >> 4: min / median / max | channels with m1 <= 255 (one conv2 plane) | pooled conv2 bound m2: max |")
    print("|---|---|---|---|---|---|")
    for name in MODELS:
        m = b.Model.from_zoo(name)
        convs = [i for i, li in enumerate(m.layers()) if li.type == 2]
        w1 = np.frombuffer(m.layer_weights(convs[0]), dtype=np.int8).reshape(-1, 9).astype(np.int64)
        w2 = np.frombuffer(m.layer_weights(convs[1]), dtype=np.int8).reshape(-1, 9).astype(np.int64)
        s1 = 127 * np.clip(w1, 0, None).sum(1) + 128 * np.clip(-w1, 0, None).sum(1)
        m1 = s1 >> 4
        m2 = (m1 * np.clip(w2, 0, None).sum(1)) >> 4
        print(f"| {name} | {len(w1)} | {s1.min()} / {int(np.median(s1))} / {s1.max()} | {m1.min()} / {int(np.median(m1))} / {m1.max()} | "
              f"{int((m1 <= 255).sum())} | {m2.max()} |")


if __name__ == "__main__":
    main()
