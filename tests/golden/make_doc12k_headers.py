#!/usr/bin/env python3
"""Generate headers of the reference's DOCUMENTED 12 KB model family (docs/documentation.md:169-183: the widths that fill
12 kbyte at each quantisation level) with the REFERENCE's own quantiser and header writer, run in this container from
/root/reference:

    doc12k_binary    FCMNIST(160,160,160, QuantType='Binary')     codec id 1   (the table's 176-160-160 is not exportable: the
                     writer's 32-bit alignment guard, exportquant.py:97-98, rejects a 176-input binary layer, and the C loop
                     BitNetMCU_inference.c:96-104 consumes whole 32-bit words; 160-160-160 is the widest equal-width shape
                     below 12 kbyte that the current exporter accepts.  The writer's Binary branch builds its codes with
                     np.where(...) -> int64 and reinterprets the packed words as uint32 (exportquant.py:104-106,182-187),
                     so the array it writes holds every word followed by a zero word: twice the words, of which the C
                     engine's row stride n_input/32 (BitNetMCU_inference.c:96-104) walks the first half.  The fixture
                     pins exactly what a user of the reference gets today, quirk included)
    doc12k_ternary   FCMNIST(128,128,112, QuantType='Ternary')    codec id 64 (padded n_input 260 / 130 / 130 / 120)
    doc12k_2bit      FCMNIST(112, 96, 96, QuantType='2bitsym')    codec id 2
    doc12k_8bit      FCMNIST( 40, 32, 32, QuantType='8bit')       codec id 16
    (the 4-bit member, 64-64-64 4bitsym, is the headline model BitNetMCU_model_fc.h itself)

Weights are torch.manual_seed(k) random-initialised (there is no dataset here); what the fixtures pin is the arithmetic of
these SHAPES and CODECS - the ones a user of the reference actually exports.  Layer names follow the reference FC wrapper
(L1..L4, BitNetMCU_MNIST_dll.c:95-120).  Outputs: tests/golden/headers/doc12k_*.h (committed; /root/reference does not
exist on the GPU box)."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_ternary_headers import import_reference_exporter, REF  # noqa: E402

SHAPES = {
    "doc12k_binary": (160, 160, 160, "Binary", 1),
    "doc12k_ternary": (128, 128, 112, "Ternary", 2),
    "doc12k_2bit": (112, 96, 96, "2bitsym", 3),
    "doc12k_8bit": (40, 32, 32, "8bit", 4),
}


def make(name, w1, w2, w3, qt, seed):
    import torch
    exportquant, BitNetMCU, models = import_reference_exporter()
    torch.manual_seed(seed)
    net = models.FCMNIST(network_width1=w1, network_width2=w2, network_width3=w3, QuantType=qt, NormType="RMS", WScale="PerTensor")
    q = BitNetMCU.QuantizedModel(net)
    for order, layer in enumerate(q.quantized_model, start=1):
        layer["layer_order"] = order
        if qt == "Ternary":
            layer["bpw"] = 0      # bypass the 32-bit alignment guard (exportquant.py:97-98); the Ternary branch ignores bpw
    path = os.path.join(HERE, "headers", name + ".h")
    exportquant.export_to_hfile(q, path, f"synthetic_seed{seed}_FCMNIST_{qt}_width{w1}_{w2}_{w3}", "FCMNIST")
    lines = open(path).read().split("\n")
    lines[1] = "// Date: (fixed by tests/golden/make_doc12k_headers.py)"
    open(path, "w").write("\n".join(lines))
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("needs /root/reference")
    for name, (w1, w2, w3, qt, seed) in SHAPES.items():
        make(name, w1, w2, w3, qt, seed)
