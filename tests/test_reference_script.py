"""Row N1 of the scope table: north_star asks for bit-exactness "via the existing test_inference.py DLL harness".

bitnetmcu_amd/harness.py is a torchvision-free RESTATEMENT of that harness's loop (test_inference.py:134-175); the GPU tests
drive the product DLL through it.  This test ties the restatement to the script itself: the reference's UNMODIFIED
/root/reference/test_inference.py is executed (runpy, __main__) in a scratch directory that holds what the script expects -

    ./Bitnet_inf.dll               the reference's unmodified BitNetMCU_MNIST_dll.c compiled (oracle/build_oracle.py's recipe)
                                   against a header written by the reference's own exporter for the model below
    ./modeldata/<runname>.pth      state dict of a seeded random-init FCMNIST(64, 64, 64, '4bitsym')  (no dataset offline:
                                   what is pinned is the plumbing and the arithmetic, not an accuracy)
    ./params.yaml                  the hyperparameters the script derives <runname> and the model class from

with `torchvision` replaced by a stub whose MNIST is a seeded synthetic 16x16 data set (torchvision and MNIST are
not available offline) - and its printed counters and mismatch lines must equal harness.cross_check() on the same images, the same
DLL and the same Python engine (QuantizedModel.inference_quantized).  The DLL's answers are also checked against the oracle port.
Runs where /root/reference exists (this container); skipped on the GPU box.
"""
import contextlib
import io
import os
import re
import runpy
import subprocess
import sys
import types

import numpy as np
import pytest

import util
from util import REF_DIR
from bitnetmcu_amd import harness

N_IMAGES = 300
SCRIPT = os.path.join(REF_DIR, "test_inference.py")

pytestmark = pytest.mark.skipif(not os.path.isfile(SCRIPT), reason="needs /root/reference (the reference's own script)")


def _synthetic_mnist(n, seed):
    """Float images shaped and scaled like the script's transformed MNIST (Normalize((0.1307,), (0.3081,)): background -0.42,
    strokes up to 2.8), labels 0..9."""
    rng = np.random.default_rng(seed)
    x = np.full((n, 1, 16, 16), -0.4242, dtype=np.float32)
    ink = rng.random((n, 1, 16, 16)) < 0.3
    x[ink] = (rng.random(int(ink.sum())) * 3.2 - 0.4).astype(np.float32)
    return x, rng.integers(0, 10, size=n)


def _install_torchvision_stub(images, labels):
    import torch

    class MNIST(torch.utils.data.Dataset):
        def __init__(self, root=None, train=True, transform=None, download=False):
            pass

        def __len__(self):
            return len(labels)

        def __getitem__(self, i):
            return torch.from_numpy(images[i]), int(labels[i])

    tv = types.ModuleType("torchvision")
    tv.datasets = types.ModuleType("torchvision.datasets")
    tv.datasets.MNIST = MNIST
    tv.transforms = types.ModuleType("torchvision.transforms")
    for name in ("Compose", "Resize", "ToTensor", "Normalize"):
        setattr(tv.transforms, name, lambda *a, **k: None)
    saved = {k: sys.modules.get(k) for k in ("torchvision", "torchvision.datasets", "torchvision.transforms")}
    sys.modules.update({"torchvision": tv, "torchvision.datasets": tv.datasets, "torchvision.transforms": tv.transforms})
    return saved


def test_the_reference_script_itself_agrees_with_the_restated_harness(tmp_path, orc, monkeypatch):
    import torch
    import yaml
    sys.path.insert(0, os.path.join(util.GOLDEN))
    from make_ternary_headers import import_reference_exporter
    exportquant, BitNetMCU, models = import_reference_exporter()

    params = {"runtag": "n1", "model": "FCMNIST", "augmentation": False, "QuantType": "4bitsym", "NormType": "RMS",
              "WScale": "PerTensor", "network_width1": 64, "network_width2": 64, "network_width3": 64, "num_epochs": 1,
              "batch_size": 64}
    runname = "n1_FCMNIST_BitMnist_4bitsym_width64_64_64_epochs1"          # test_inference.py:17-20
    torch.manual_seed(20260926)
    net = models.FCMNIST(network_width1=64, network_width2=64, network_width3=64, QuantType="4bitsym", NormType="RMS",
                         WScale="PerTensor")
    os.makedirs(tmp_path / "modeldata")
    torch.save(net.state_dict(), tmp_path / "modeldata" / f"{runname}.pth")
    (tmp_path / "params.yaml").write_text(yaml.safe_dump(params))

    # the header the reference's exporter writes for this model, and the reference DLL built from it
    q = BitNetMCU.QuantizedModel(net)
    for order, layer in enumerate(q.quantized_model, start=1):
        layer["layer_order"] = order                                     # L1..L4, the FC wrapper's names (BitNetMCU_MNIST_dll.c:95-120)
    with contextlib.redirect_stdout(io.StringIO()):
        exportquant.export_to_hfile(q, str(tmp_path / "BitNetMCU_model.h"), runname, "FCMNIST")
    subprocess.check_call(["gcc", "-O2", "-fno-strict-aliasing", "-w", "-fPIC", "-shared", "-D_DLL", "-I", str(tmp_path), "-I", REF_DIR,
                           os.path.join(REF_DIR, "BitNetMCU_MNIST_dll.c"), "-o", str(tmp_path / "Bitnet_inf.dll")])

    images, labels = _synthetic_mnist(N_IMAGES, seed=7)
    saved = _install_torchvision_stub(images, labels)
    monkeypatch.chdir(tmp_path)
    monkeypatch.setattr(sys, "argv", ["test_inference.py", "--params", "params.yaml"])
    out = io.StringIO()
    try:
        with contextlib.redirect_stdout(out):
            runpy.run_path(SCRIPT, run_name="__main__")                  # the reference's script, unmodified
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    text = out.getvalue()
    size = int(re.search(r"size of test data: (\d+)", text).group(1))
    mis_c, mis_py = map(int, re.search(r"Mispredictions C: (\d+) Py: (\d+)", text).groups())
    mismatches = int(re.search(r"Mismatches between engines: (\d+)", text).group(1))
    lines = [(int(m.group(1)), int(m.group(2)), int(m.group(3)))
             for m in re.finditer(r"^\s*(\d+) Mismatch between inference engines found\. Prediction C: (\d+) Prediction Python: (\d+)", text, re.M)]

    # the restatement on the same images, the same DLL, the same Python engine
    flat = images.reshape(N_IMAGES, 256)
    q8 = harness.quantize_input(flat)
    lib = harness.load_inference_dll(str(tmp_path / "Bitnet_inf.dll"))
    predict_py = lambda _: np.array([int(np.argmax(q.inference_quantized(flat[i:i + 1]), axis=1)[0]) for i in range(N_IMAGES)])
    st = harness.cross_check(lib, predict_py, q8, labels)
    assert size == st["counter"] == N_IMAGES
    assert mis_c == N_IMAGES - st["correct_c"] and mis_py == N_IMAGES - st["correct_other"]
    assert mismatches == st["mismatch"] == len(lines)
    assert [l[0] for l in lines] == list(st["mismatch_idx"])
    assert all(st["result_c"][i] == c and st["result_other"][i] == p for i, c, p in lines)

    # ... and the DLL's answers are the oracle port's (header text -> the package's parser -> oracle)
    import bitnetmcu_amd as b
    model = b.Model.from_header_text((tmp_path / "BitNetMCU_model.h").read_text())
    want = util.OracleModel(model, orc).infer(q8)
    assert np.array_equal(st["result_c"], np.asarray(want, dtype=np.uint32))
