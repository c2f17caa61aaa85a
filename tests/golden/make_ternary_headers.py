#!/usr/bin/env python3
"""Generate the ternary FC 96-96-96 model headers (BASELINE.json config 3) with the REFERENCE's
own quantiser and header writer, run in this container from /root/reference.

Why a script: exportquant.py cannot export a ternary FC model through its CLI — the 32-bit
alignment guard (exportquant.py:97-98) runs before the Ternary branch with bpw = 1.6 and rejects
256 and 96 inputs.  Following §0.6 we build the intermediate list-of-dicts with
the reference's QuantizedModel.quantize (BitNetMCU.py:351-418), set bpw = 0 and layer_order =
1..4 (the names the reference FC wrapper hard-codes, BitNetMCU_MNIST_dll.c:95-120), and call the
UNMODIFIED export_to_hfile (exportquant.py:49-263).

Outputs (committed fixtures, consumed on the GPU box where /root/reference does not exist):
  tests/golden/headers/tern_96.h         FCMNIST(96,96,96,'Ternary'), torch.manual_seed(0) init
                                         (trits ~37.5/25/37.5 % for -1/0/+1)
  tests/golden/headers/tern_96_sparse.h  same net, weights rescaled so ~49 % of the trits are 0
                                         (the trained-model statistic of docs/documentation.md:877)
"""
import os
import sys
import types

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def import_reference_exporter():
    """exportquant.py imports torchvision / seaborn / matplotlib at module top
    (exportquant.py:2,9,12); none is installed and none is used by export_to_hfile."""
    sys.path.insert(0, REF)
    tv = _stub("torchvision")
    tv.datasets = _stub("torchvision.datasets")
    tv.transforms = _stub("torchvision.transforms")
    mpl = _stub("matplotlib")
    mpl.pyplot = _stub("matplotlib.pyplot")
    _stub("seaborn")
    import exportquant  # noqa: E402
    import BitNetMCU  # noqa: E402
    import models  # noqa: E402
    return exportquant, BitNetMCU, models


def make(path, zero_fraction=None):
    import torch
    exportquant, BitNetMCU, models = import_reference_exporter()
    torch.manual_seed(0)
    net = models.FCMNIST(network_width1=96, network_width2=96, network_width3=96,
                         QuantType="Ternary", NormType="RMS", WScale="PerTensor")
    if zero_fraction is not None:
        # weight_quant rounds w / mean|w| (BitNetMCU.py:145-152): |w| < 0.5*mean|w| -> trit 0.
        # Push a fraction of the smallest weights towards 0 so that ~zero_fraction quantise to 0.
        with torch.no_grad():
            for m in net.modules():
                if isinstance(m, BitNetMCU.BitLinear):
                    w = m.weight
                    thr = w.abs().flatten().kthvalue(int(zero_fraction * w.numel())).values
                    w[w.abs() <= thr] *= 1e-3
    q = BitNetMCU.QuantizedModel(net)
    order = 1
    for layer in q.quantized_model:
        layer["layer_order"] = order
        layer["bpw"] = 0  # bypass the alignment guard; the Ternary branch ignores bpw
        order += 1
    os.makedirs(os.path.dirname(path), exist_ok=True)
    exportquant.export_to_hfile(q, path, "synthetic_seed0_FCMNIST_Ternary_width96_96_96", "FCMNIST")
    # the writer stamps datetime.now() on line 2 (exportquant.py:71): make the fixture reproducible
    lines = open(path).read().split("\n")
    lines[1] = "// Date: (fixed by tests/golden/make_ternary_headers.py)"
    open(path, "w").write("\n".join(lines))
    import numpy as np
    trits = np.concatenate([np.array(l["quantized_weights"]).ravel() for l in q.quantized_model])
    print(path, {v: float((trits == v).mean()) for v in (-1, 0, 1)})


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("needs /root/reference")
    make(os.path.join(HERE, "headers", "tern_96.h"))
    make(os.path.join(HERE, "headers", "tern_96_sparse.h"), zero_fraction=0.49)
