#!/usr/bin/env python3
"""Fixtures for SURVEY.md §8f row 3 (GPU-backed QuantizedModel.inference_quantized), generated in the build container with
the REFERENCE's own code imported from /root/reference:

  * models built by the reference's models.py (random init, fixed seed), quantised by the reference's
    QuantizedModel.quantize (BitNetMCU.py:341-418) -> the list of dicts `quantized_model`;
  * that list written by the reference's UNMODIFIED export_to_hfile (exportquant.py:49-263) -> header text;
  * the reference's UNMODIFIED C engine compiled against that header (the two gcc lines of oracle/build_oracle.py, outputs in
    a temp dir) -> class ids, logits and activations for seeded float inputs quantised as test_inference.py:140-141 does.

Output: tests/golden/f3_<name>.npz = {layers_json, arrays..., header_text, x_float, x_int8, cls, logits}.
FC exports name their layers by module index, which the reference's FC wrapper cannot compile; as for the
ternary fixtures the dicts' layer_order is set to 1..4 before the export — the renumbered list is what is stored.

Run:  python tests/golden/make_f3_golden.py      (needs /root/reference; consumed by tests/test_evaluate_f3.py everywhere)
"""
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import util  # noqa: E402
from make_ternary_headers import import_reference_exporter  # noqa: E402
from bitnetmcu_amd import Model, harness  # noqa: E402

REF = "/root/reference"

SPECS = {
    # name: (model class, kwargs, renumber FC layers to 1..n?)
    "fc64_4bitsym": ("FCMNIST", dict(network_width1=64, network_width2=64, network_width3=64, QuantType="4bitsym"), True),
    "fc128_47cls": ("FCMNIST", dict(network_width1=128, network_width2=64, network_width3=64, QuantType="4bitsym", num_classes=47), True),
    "fc96_64_3layer_2bitsym": ("FCMNIST", dict(network_width1=96, network_width2=64, network_width3=0, QuantType="2bitsym"), True),
    "cnn16_4bitsym": ("CNNMNIST", dict(network_width1=96, network_width2=64, network_width3=0, cnn_width=16, QuantType="4bitsym"), False),
}


def build_dll(header_path, outdir):
    with open(os.path.join(outdir, "BitNetMCU_model.h"), "w") as f:
        f.write(f'#include "{header_path}"\n')
    dll = os.path.join(outdir, "Bitnet_inf.dll")
    subprocess.check_call(["gcc", "-fno-strict-aliasing", "-w", "-fPIC", "-shared", "-D_DLL", "-O2", "-I", outdir, "-I", REF,
                           os.path.join(REF, "BitNetMCU_MNIST_dll.c"), "-o", dll])
    return C.CDLL(dll)


def main():
    import torch
    exportquant, BitNetMCU, models = import_reference_exporter()
    for name, (cls_name, kw, renumber) in SPECS.items():
        torch.manual_seed(sum(map(ord, name)))
        net = getattr(models, cls_name)(NormType="RMS", WScale="PerTensor", **kw)
        qm = BitNetMCU.QuantizedModel(net)
        layers = qm.quantized_model
        rng = np.random.default_rng(sum(map(ord, name)))
        x_float = rng.normal(size=(40, 256)).astype(np.float32)
        x_float[0] = 0.0                          # all-zero image: scale clamps at 1e-5
        x_float[1] *= 1e-3
        if cls_name == "CNNMNIST":
            qm.inference_quantized(x_float[:2].copy())         # the reference fills the conv/pool geometry while it runs
        if renumber:
            for k, l in enumerate(layers, start=1):
                l["layer_order"] = k
        with tempfile.TemporaryDirectory() as tmp:
            hdr = os.path.join(tmp, "model.h")
            exportquant.export_to_hfile(qm, hdr, "fixture", cls_name)
            text = open(hdr).read()
            dll = build_dll(hdr, tmp)
            model = Model.from_header_text(text)
            x_int8 = harness.quantize_input(x_float)
            f = util.Funcs(dll)
            cls, logits = [], []
            for img in x_int8:
                c, lg, _ = util.run_schedule(f, model, img)
                cls.append(c)
                logits.append(lg)
            if not (cls_name == "CNNMNIST" and kw["cnn_width"] * 4 < 256):      # the x86 CNN wrapper smashes its stack there
                dll.Inference.restype = C.c_uint32
                dll.Inference.argtypes = [C.POINTER(C.c_int8)]
                for img, c in zip(x_int8, cls):
                    buf = np.ascontiguousarray(img)
                    assert dll.Inference(buf.ctypes.data_as(C.POINTER(C.c_int8))) == c
        meta, arrays = [], {}
        for i, l in enumerate(layers):
            m = {k: v for k, v in l.items() if k != "quantized_weights"}
            meta.append(m)
            if "quantized_weights" in l:
                arrays[f"w{i}"] = np.asarray(l["quantized_weights"], dtype=np.float64)
        np.savez_compressed(os.path.join(HERE, f"f3_{name}.npz"), layers_json=json.dumps(meta), header_text=text, x_float=x_float,
                            x_int8=x_int8, cls=np.array(cls, np.uint32), logits=np.array(logits, np.int32), **arrays)
        print(name, "classes", model.num_classes, "ids", cls[:10])


if __name__ == "__main__":
    main()
