"""Drop-in boundary on the GPU box: a model-bound Bitnet_inf.dll built by this repository, loaded and driven
exactly as test_inference.py:134-150 does (CDLL, argtypes=[POINTER(c_int8)], restype=c_uint32, one call per image),
must return the class ids of the reference DLL."""
import os

import numpy as np
import pytest

import util
from util import GOLDEN, REPO
from bitnetmcu_amd import harness, synth, DIST_U, DIST_M

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["fc_4bitsym_64", "cnn_64", "tern_96", "mcu_1k", "mcu_12k_fp130", "doc12k_binary", "doc12k_ternary"])
def test_our_dll_in_the_reference_harness(name, gpu_ok, orc):
    dll = os.path.join(REPO, "bitnetmcu_amd", "dlls", name, "Bitnet_inf.dll")
    # (a missing DLL FAILS: on a GPU box the drop-in boundary must not go green by skipping; `python -c "import __graft_entry__ as
    # g; g.build()"` in the build container makes them - they travel with the tree)
    assert os.path.isfile(dll), f"{dll} not built"
    ours = harness.load_inference_dll(dll)
    model = util.load_golden_model(name)
    om = util.OracleModel(model, orc)
    r = np.load(os.path.join(GOLDEN, "real_images.npz"))
    k = np.load(os.path.join(GOLDEN, f"kat_{name}.npz"))
    # 13 real images: labels for the trained 10-class models, reference class ids for all
    st = harness.cross_check(ours, om.infer, r["images"], r["labels"])
    assert st["mismatch"] == 0
    if not name.startswith(("tern", "doc12k")):       # trained models: every one of the 13 real images is classified correctly
        assert st["correct_c"] == 13
    assert np.array_equal(harness.run_inference_loop(ours, k["images"]), k["cls"])
    # config 1's contract (test_inference.py:136-168 loops all 10,000 test images through lib.Inference; SURVEY.md 8d: 13 real +
    # 10,000 synthetic): 10,000 one-image calls with varying inputs through the latency path's polling protocol
    x = np.concatenate([synth.images(0, 5000, DIST_U), synth.images(0, 5000, DIST_M)])
    st = harness.cross_check(ours, om.infer, x)
    assert st["counter"] == 10000 and st["mismatch"] == 0
    if util.have_ref_dll(name) and not (model.kind == 1 and model.layer(0).out_channels * 4 < 256):
        ref = harness.load_inference_dll(util.ref_dll_path(name))       # the compiled reference, same loop
        assert np.array_equal(harness.run_inference_loop(ref, x), st["result_c"])


@pytest.mark.parametrize("name", ["fc_4bitsym_64", "mcu_1k", "mcu_12k_fp130", "tern_96", "cnn_64", "mcu_cnn_16"])
def test_c_host_calling_the_kernel_symbols_one_by_one(name, gpu_ok, orc, tmp_path):
    """A C program written against BitNetMCU_inference.h - processfclayer / ReLUNorm / processconv33ReLU / processmaxpool22 called
    layer by layer on the arrays of an exporter-dialect BitNetMCU_model.h, the flow of BitNetMCU_MNIST_test.c:43-139 - compiled
    with gcc and linked against the GPU library instead of BitNetMCU_inference.c: prints the reference's lines with the
    oracle's class ids (3- and 4-layer FC, FP130, ternary uint16 rows, 64- and 16-channel CNN with in-place planes)."""
    import subprocess
    model = util.load_golden_model(name)
    r = np.load(os.path.join(GOLDEN, "real_images.npz"))
    images, labels = r["images"][3:13], r["labels"][3:13]
    (tmp_path / "BitNetMCU_model.h").write_text(model.to_header_text())
    (tmp_path / "BitNetMCU_MNIST_test_data.h").write_text(util.test_data_header(images, labels))
    exe = util.compile_c_host("mnist_test.c", tmp_path)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    want = util.OracleModel(model, orc).infer(images)
    assert out.stdout.splitlines() == [f"label: {int(l)} predicted: {int(c)}" for l, c in zip(labels, want)]


_COLD_START_SCRIPT = r"""
import sys, threading
import numpy as np
from ctypes import c_int8
sys.path.insert(0, sys.argv[1])
from bitnetmcu_amd import harness
lib = harness.load_inference_dll(sys.argv[2])            # nothing called yet: the GPU context does not exist
x = np.load(sys.argv[3])
out = np.full(len(x), 0xFFFFFFFF, np.uint32)
T = 8
def work(t):
    for i in range(t, len(x), T):
        out[i] = lib.Inference((c_int8 * 256)(*x[i].tolist()))
threads = [threading.Thread(target=work, args=(t,)) for t in range(T)]
for th in threads: th.start()                            # eight cold first calls race for the lazy initialisation
for th in threads: th.join()
np.save(sys.argv[4], out)
"""


@pytest.mark.parametrize("name,persistent", [("fc_4bitsym_64", "0"), ("cnn_64", "0"), ("fc_4bitsym_64", "1"), ("cnn_64", "1")])
def test_dll_cold_start_from_eight_threads(name, persistent, gpu_ok, orc, tmp_path):
    """SURVEY 8(b) threading: the reference DLL is stateless and re-entrant; ours creates its GPU context on the first call.  Eight
    host threads make their FIRST Inference() call at the same time in a fresh process (ctypes releases the GIL) and go on
    calling concurrently: every class id equals the oracle's - with a launch per call and with the resident kernels (round 6)."""
    import subprocess
    import sys
    dll = os.path.join(REPO, "bitnetmcu_amd", "dlls", name, "Bitnet_inf.dll")
    if not os.path.isfile(dll):
        pytest.fail(f"{dll} not built: run __graft_entry__.build() where the model headers are")
    model = util.load_golden_model(name)
    x = np.concatenate([synth.images(77, 1200, DIST_U), synth.images(77, 1200, DIST_M)])
    np.save(tmp_path / "x.npy", x)
    script = tmp_path / "cold.py"
    script.write_text(_COLD_START_SCRIPT)
    # persistent "1": BNM_PERSISTENT=1 in the process's environment - every thread's leased context starts its own resident
    # one-image kernel (the FC model; the CNN model is not served by it and keeps its launches), the process exits with them resident
    subprocess.check_call([sys.executable, str(script), REPO, dll, str(tmp_path / "x.npy"), str(tmp_path / "out.npy")], timeout=300,
                          env=dict(os.environ, BNM_PERSISTENT=persistent))
    assert np.array_equal(np.load(tmp_path / "out.npy"), util.OracleModel(model, orc).infer(x))


def test_one_context_and_the_kernel_symbols_from_many_host_threads(gpu_ok, orc, bnm):
    """Six host threads share ONE context (batched host-pointer calls of different sizes, the <= 64-image latency path among
    them) while six more call the reference's kernel symbols: every result equals the oracle's."""
    import threading
    model = util.load_golden_model("fc_4bitsym_64")
    om = util.OracleModel(model, orc)
    import bitnetmcu_amd as b
    ctx = b.Context(model)
    sizes = [1, 17, 64, 65, 5000, 300_000]
    xs = [synth.images(1000 * k, n, DIST_U) for k, n in enumerate(sizes)]
    want = [om.infer(x[:20000], logits=True) for x in xs]
    f_ours, f_ref = util.Funcs(bnm), util.Funcs(orc, "orc_")
    rng = np.random.default_rng(5)
    sym_in = [(rng.integers(-128, 128, size=256).astype(np.int8), rng.integers(0, 2**32, size=64 * 32, dtype=np.uint32)) for _ in range(6)]
    errors = []

    def batch(k):
        try:
            for rep in range(6):
                cls, lg = ctx.infer(xs[k], logits=True)
                m = min(len(cls), 20000)
                if not (np.array_equal(cls[:m], want[k][0][:m]) and np.array_equal(lg[:m], want[k][1][:m])):
                    errors.append(("batch", k, rep))
        except Exception as e:       # noqa: BLE001
            errors.append(("batch", k, repr(e)))

    def symbols(k):
        try:
            act, w = sym_in[k]
            ref_sum = f_ref.processfclayer(act, w, 4, 256, 64)
            ref_out = f_ref.relunorm(ref_sum)
            for rep in range(150):
                s = f_ours.processfclayer(act, w, 4, 256, 64)
                o = f_ours.relunorm(s)
                if not (np.array_equal(s, ref_sum) and np.array_equal(o[0], ref_out[0]) and o[1] == ref_out[1]):
                    errors.append(("symbols", k, rep))
        except Exception as e:       # noqa: BLE001
            errors.append(("symbols", k, repr(e)))

    threads = [threading.Thread(target=batch, args=(k,)) for k in range(6)] + [threading.Thread(target=symbols, args=(k,)) for k in range(6)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    ctx.close()
    assert not errors, errors[:5]
