"""TEST INFRASTRUCTURE (see oracle/README.md): running the reference's own harness script, test_inference.py, UNMODIFIED.

north_star asks for bit-exactness "via the existing test_inference.py DLL harness".  The script (test_inference.py:1-175) loads
`./Bitnet_inf.dll` from the current directory, quantises one MNIST test image at a time in Python, calls `lib.Inference`, runs
the reference's Python engine on the same image and prints counters.  What it needs that does not exist offline - torchvision and
the MNIST files - is supplied from outside the script: a stub `torchvision` whose MNIST is a seeded synthetic
16x16 data set.  The script file itself is executed byte for byte (runpy, run_name "__main__").

  stage()      (this container, /root/reference present; called by __graft_entry__.build()): writes oracle/_ref/harness/ -
               git-ignored, shipped to the GPU box by gpurun like the compiled reference DLLs beside it:
                 test_inference.py BitNetMCU.py models.py      byte-identical copies of the reference's files (sha256 in MANIFEST.json)
                 modeldata/<runname>.pth  params.yaml          a seeded random-init FCMNIST(64,64,64,'4bitsym') checkpoint + its hyperparameters
                 BitNetMCU_model.h                             written by the reference's own exporter (exportquant.export_to_hfile) for it
                 ref/Bitnet_inf.dll                            the reference's BitNetMCU_MNIST_dll.c compiled against that header (gcc)
                 product/Bitnet_inf.dll                        the PRODUCT's drop-in DLL bound to the same header (bitnetmcu_amd/build.py --dll)
                 expected_stdout.txt                           what the script printed here, driving ref/Bitnet_inf.dll
  run_script() executes the staged script in a directory holding one of the two DLLs and returns its stdout.
The GPU test (tests/test_gpu_reference_script.py) runs the script against product/Bitnet_inf.dll on the MI355X box and compares its
output with expected_stdout.txt line for line.
"""
import contextlib
import hashlib
import io
import json
import os
import runpy
import shutil
import subprocess
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF = "/root/reference"
STAGE = os.path.join(HERE, "_ref", "harness")
SCRIPT_FILES = ("test_inference.py", "BitNetMCU.py", "models.py")
N_IMAGES = 10000            # the size of the MNIST test set the script walks (BASELINE configs[0])
DATA_SEED = 7
MODEL_SEED = 20260926
PARAMS = {"runtag": "n1", "model": "FCMNIST", "augmentation": False, "QuantType": "4bitsym", "NormType": "RMS",
          "WScale": "PerTensor", "network_width1": 64, "network_width2": 64, "network_width3": 64, "num_epochs": 1,
          "batch_size": 64}
RUNNAME = "n1_FCMNIST_BitMnist_4bitsym_width64_64_64_epochs1"          # test_inference.py:17-20


def synthetic_mnist(n=N_IMAGES, seed=DATA_SEED):
    """Float images shaped and scaled like the script's transformed MNIST (Normalize((0.1307,), (0.3081,)): background -0.42,
    strokes up to 2.8), labels 0..9."""
    rng = np.random.default_rng(seed)
    x = np.full((n, 1, 16, 16), -0.4242, dtype=np.float32)
    ink = rng.random((n, 1, 16, 16)) < 0.3
    x[ink] = (rng.random(int(ink.sum())) * 3.2 - 0.4).astype(np.float32)
    return x, rng.integers(0, 10, size=n)


@contextlib.contextmanager
def script_environment(images, labels, module_dir):
    """What the script imports but the box does not have (torchvision), its own modules' directory on sys.path, and torch told
    that there is no CUDA device - the script's two PyTorch engines then compute on the host as they did where
    expected_stdout.txt was written; the DLL under test is not affected (it talks to HIP itself)."""
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
    names = ("torchvision", "torchvision.datasets", "torchvision.transforms", "BitNetMCU", "models")
    saved = {k: sys.modules.pop(k, None) for k in names}
    sys.modules.update({"torchvision": tv, "torchvision.datasets": tv.datasets, "torchvision.transforms": tv.transforms})
    sys.path.insert(0, module_dir)
    cuda_available = torch.cuda.is_available
    torch.cuda.is_available = lambda: False
    try:
        yield
    finally:
        torch.cuda.is_available = cuda_available
        sys.path.remove(module_dir)
        for k in names:
            sys.modules.pop(k, None)
            if saved[k] is not None:
                sys.modules[k] = saved[k]


def run_script(script, cwd, module_dir, images, labels):
    """Execute the reference's script file as __main__ in `cwd` (which holds ./Bitnet_inf.dll, params.yaml, modeldata/);
    returns everything it printed."""
    out = io.StringIO()
    old_cwd, old_argv = os.getcwd(), sys.argv
    os.chdir(cwd)
    sys.argv = ["test_inference.py", "--params", "params.yaml"]
    try:
        with script_environment(images, labels, module_dir), contextlib.redirect_stdout(out):
            runpy.run_path(script, run_name="__main__")
    finally:
        os.chdir(old_cwd)
        sys.argv = old_argv
    return out.getvalue()


def sha256(path):
    return hashlib.sha256(open(path, "rb").read()).hexdigest()


def make_run_dir(kind):
    """A directory the script can run in: ./Bitnet_inf.dll (kind: 'ref' or 'product'), params.yaml, modeldata/."""
    d = os.path.join(STAGE, kind)
    shutil.copyfile(os.path.join(STAGE, "params.yaml"), os.path.join(d, "params.yaml"))
    shutil.copytree(os.path.join(STAGE, "modeldata"), os.path.join(d, "modeldata"), dirs_exist_ok=True)
    return d


def stage(build_product=True):
    """See the module docstring.  Needs /root/reference; returns the staging directory."""
    import torch
    import yaml
    if not os.path.isfile(os.path.join(REF, "test_inference.py")):
        raise RuntimeError(f"{REF} is not present: the staged harness can only be (re)built where the reference is")
    sys.path.insert(0, os.path.join(REPO, "tests", "golden"))
    from make_ternary_headers import import_reference_exporter
    exportquant, BitNetMCU, models = import_reference_exporter()
    os.makedirs(os.path.join(STAGE, "modeldata"), exist_ok=True)
    manifest = {}
    for f in SCRIPT_FILES:
        shutil.copyfile(os.path.join(REF, f), os.path.join(STAGE, f))
        manifest[f] = sha256(os.path.join(STAGE, f))
    torch.manual_seed(MODEL_SEED)
    net = models.FCMNIST(network_width1=64, network_width2=64, network_width3=64, QuantType="4bitsym", NormType="RMS",
                         WScale="PerTensor")
    torch.save(net.state_dict(), os.path.join(STAGE, "modeldata", f"{RUNNAME}.pth"))
    with open(os.path.join(STAGE, "params.yaml"), "w") as f:
        f.write(yaml.safe_dump(PARAMS))
    q = BitNetMCU.QuantizedModel(net)
    for order, layer in enumerate(q.quantized_model, start=1):
        layer["layer_order"] = order                                     # L1..L4, the FC wrapper's names (BitNetMCU_MNIST_dll.c:95-120)
    header = os.path.join(STAGE, "BitNetMCU_model.h")
    with contextlib.redirect_stdout(io.StringIO()):
        exportquant.export_to_hfile(q, header, RUNNAME, "FCMNIST")
    manifest["BitNetMCU_model.h"] = sha256(header)
    for m in ("exportquant", "BitNetMCU", "models"):
        sys.modules.pop(m, None)
    os.makedirs(os.path.join(STAGE, "ref"), exist_ok=True)
    subprocess.check_call(["gcc", "-O2", "-fno-strict-aliasing", "-w", "-fPIC", "-shared", "-D_DLL", "-I", STAGE, "-I", REF,
                           os.path.join(REF, "BitNetMCU_MNIST_dll.c"), "-o", os.path.join(STAGE, "ref", "Bitnet_inf.dll")])
    if build_product:
        os.makedirs(os.path.join(STAGE, "product"), exist_ok=True)
        subprocess.check_call([sys.executable, os.path.join(REPO, "bitnetmcu_amd", "build.py"), "--dll", header, "-o",
                               os.path.join(STAGE, "product")])
    images, labels = synthetic_mnist()
    text = run_script(os.path.join(STAGE, "test_inference.py"), make_run_dir("ref"), STAGE, images, labels)
    with open(os.path.join(STAGE, "expected_stdout.txt"), "w") as f:
        f.write(text)
    make_run_dir("product")
    manifest["expected_stdout.txt"] = sha256(os.path.join(STAGE, "expected_stdout.txt"))
    manifest["n_images"] = N_IMAGES
    with open(os.path.join(STAGE, "MANIFEST.json"), "w") as f:
        json.dump(manifest, f, indent=1)
    return STAGE


if __name__ == "__main__":
    print(stage())
