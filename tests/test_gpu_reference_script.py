"""Row N1 of the scope table, on the GPU: north_star asks for bit-exactness "via the existing test_inference.py DLL harness".

The reference's UNMODIFIED test_inference.py (a byte-identical copy staged by oracle/refscript.py under oracle/_ref/harness/, which
gpurun ships like the compiled reference DLLs; sha256 in its MANIFEST.json) is executed as __main__ - CDLL('./Bitnet_inf.dll'),
per-image quantisation in Python, lib.Inference per image, the reference's Python engine beside it, 10,000 images
(test_inference.py:134-175) - in a directory whose ./Bitnet_inf.dll is the PRODUCT's drop-in DLL bound to the exporter-written
header of the staged checkpoint.  Everything it prints must equal, line for line, what it prints in the same process against the
reference's own compiled DLL (ref/Bitnet_inf.dll, gcc), and its C-side lines must equal the output recorded in the build
container.  Only torchvision / MNIST (absent offline) are stubbed, from outside the script.
"""
import json
import os
import re
import sys

import pytest

import util

sys.path.insert(0, os.path.join(util.REPO, "oracle"))
import refscript  # noqa: E402

pytestmark = pytest.mark.gpu


def _c_side(text):
    """The lines of the script's output that depend on the DLL under test only (not on the host's float arithmetic)."""
    size = re.search(r"size of test data: (\d+)", text).group(1)
    mis_c = re.search(r"Mispredictions C: (\d+)", text).group(1)
    acc_c = re.search(r"Overall accuracy C: ([\d.]+)", text).group(1)
    lines = [(m.group(1), m.group(2)) for m in re.finditer(r"^\s*(\d+) Mismatch between inference engines found\. Prediction C: (\d+)", text, re.M)]
    return size, mis_c, acc_c, lines


def test_the_unmodified_reference_script_drives_the_product_dll(gpu_ok):
    stage = refscript.STAGE
    need = [os.path.join(stage, f) for f in refscript.SCRIPT_FILES + ("MANIFEST.json", "expected_stdout.txt", "params.yaml")] + \
           [os.path.join(stage, "product", "Bitnet_inf.dll"), os.path.join(stage, "ref", "Bitnet_inf.dll")]
    missing = [p for p in need if not os.path.isfile(p)]
    assert not missing, f"the staged reference harness is incomplete ({missing}): run python -c 'import __graft_entry__ as g; g.build()' where /root/reference exists"
    manifest = json.load(open(os.path.join(stage, "MANIFEST.json")))
    for f in refscript.SCRIPT_FILES:       # the files that run are the files that were copied from the reference
        assert refscript.sha256(os.path.join(stage, f)) == manifest[f], f
    if os.path.isfile(os.path.join(refscript.REF, "test_inference.py")):      # (build container: the copy IS the reference's file)
        for f in refscript.SCRIPT_FILES:
            assert refscript.sha256(os.path.join(refscript.REF, f)) == manifest[f], f
    images, labels = refscript.synthetic_mnist()
    script = os.path.join(stage, "test_inference.py")
    out_product = refscript.run_script(script, os.path.join(stage, "product"), stage, images, labels)
    out_ref = refscript.run_script(script, os.path.join(stage, "ref"), stage, images, labels)
    assert f"size of test data: {refscript.N_IMAGES}" in out_product
    assert out_product == out_ref, "the script prints something else with the product DLL than with the reference's own DLL"
    expected = open(os.path.join(stage, "expected_stdout.txt")).read()
    assert _c_side(out_product) == _c_side(expected), "C-side lines differ from the run recorded in the build container"
    assert len(_c_side(out_product)[3]) > 0       # (the two engines do disagree on some images: those lines carry the DLL's answers)
    print(out_product[-400:])


SUBPROCESS_DRIVER = r"""
import json, os, sys, time
sys.path.insert(0, {repo!r}); sys.path.insert(0, os.path.join({repo!r}, "oracle"))
import refscript
images, labels = refscript.synthetic_mnist()
stage = refscript.STAGE
t0 = time.perf_counter()
out = refscript.run_script(os.path.join(stage, "test_inference.py"), os.path.join(stage, {kind!r}), stage, images, labels)
print(json.dumps({{"seconds": time.perf_counter() - t0, "out": out}}))
"""


def test_the_unmodified_reference_script_with_the_resident_kernel(gpu_ok):
    """the same unmodified script, in a process of its own whose environment says BNM_PERSISTENT=1 - every
    lib.Inference call of its 10,000-image loop is a mailbox message to the resident kernel instead of a launch.  What it prints
    must equal what it prints without the flag and against the reference's own DLL, and the process must exit cleanly."""
    import subprocess
    stage = refscript.STAGE
    runs = {}
    subprocess.run([sys.executable, "-c", "import torch, numpy"], timeout=600)      # (a fresh box pages the interpreter's libraries in once)
    for tag, kind, flag in (("launch_per_call", "product", "0"), ("resident_kernel", "product", "1"), ("reference_dll", "ref", "0")):
        r = subprocess.run([sys.executable, "-c", SUBPROCESS_DRIVER.format(repo=util.REPO, kind=kind)], capture_output=True, text=True,
                           timeout=900, env=dict(os.environ, BNM_PERSISTENT=flag))
        assert r.returncode == 0, (tag, r.stderr[-2000:])
        runs[tag] = json.loads(r.stdout.strip().splitlines()[-1])
    assert runs["resident_kernel"]["out"] == runs["launch_per_call"]["out"] == runs["reference_dll"]["out"]
    expected = open(os.path.join(stage, "expected_stdout.txt")).read()
    assert _c_side(runs["resident_kernel"]["out"]) == _c_side(expected)
    print({k: round(v["seconds"], 2) for k, v in runs.items()}, "seconds for the whole script (10,000 images: the reference's Python engine runs beside every call)")
