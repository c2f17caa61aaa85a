"""the reference's in-memory `QuantizedModel.quantized_model` (list of dicts) -> model -> GPU, C-exact.
Fixtures: tests/golden/f3_*.npz, written by tests/golden/make_f3_golden.py from the reference's own quantiser, exporter and
compiled C engine."""
import glob
import json
import os

import numpy as np
import pytest

import util
from bitnetmcu_amd import Model, evaluate

FIXTURES = sorted(glob.glob(os.path.join(util.GOLDEN, "f3_*.npz")))


def load(path):
    d = np.load(path)
    layers = json.loads(str(d["layers_json"]))
    for i, l in enumerate(layers):
        if f"w{i}" in d:
            l["quantized_weights"] = d[f"w{i}"].tolist()          # as QuantizedModel.quantize stores them (BitNetMCU.py:372)
    return d, layers


class FakeQuantizedModel:
    """what evaluate.* needs of a reference QuantizedModel: the list and a method to patch"""

    def __init__(self, layers):
        self.quantized_model = layers

    def inference_quantized(self, x):
        raise AssertionError("the reference's numpy loop must not run once the GPU evaluator is attached")


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[3:-4] for p in FIXTURES])
def test_in_memory_packing_equals_the_exporters_header(path):
    """pack_bitlinear / header_text restate exportquant.py:104-259: parsed, they must equal the unmodified exporter's output
    word for word — also when the conv geometry has not been filled in yet (exportquant.py calls inference_quantized first)."""
    d, layers = load(path)
    want = Model.from_header_text(str(d["header_text"])).to_blob()
    assert Model.from_header_text(evaluate.header_text(layers)).to_blob() == want
    for l in layers:
        for k in ("incoming_x", "incoming_y", "outgoing_x", "outgoing_y"):
            if k in l:
                l[k] = 0
    assert Model.from_header_text(evaluate.header_text(layers)).to_blob() == want


def test_record_geometry_fills_the_dicts_as_the_reference_method_does():
    """export_to_hfile writes Lk_incoming_x ... from the dicts, which the reference's inference_quantized fills in as a
    side effect (BitNetMCU.py:479-483, 509-513).  evaluate.record_geometry (run by attach / from_quantized_model) must leave the
    values the reference run left in the fixture."""
    seen = 0
    for path in FIXTURES:
        d, layers = load(path)
        want = [{k: l.get(k) for k in ("incoming_x", "incoming_y", "outgoing_x", "outgoing_y")} for l in layers]
        if not any(w["incoming_x"] for w in want):
            continue
        for l in layers:
            for k in ("incoming_x", "incoming_y", "outgoing_x", "outgoing_y"):
                if k in l:
                    l[k] = 0
        evaluate.record_geometry(layers)
        got = [{k: l.get(k) for k in ("incoming_x", "incoming_y", "outgoing_x", "outgoing_y")} for l in layers]
        assert got == want, path
        seen += 1
    assert seen >= 1, "no CNN fixture with geometry"


def test_per_output_scales_raise_a_warning():
    import warnings

    class Stop(Exception):
        pass

    _, layers = load(FIXTURES[0])
    layers[-1]["WScale"] = "PerOutput"
    orig = evaluate.header_text
    evaluate.header_text = lambda *a, **k: (_ for _ in ()).throw(Stop())      # stop before anything needs a GPU
    try:
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            with pytest.raises(Stop):
                evaluate.QuantizedEvaluator(layers)
        assert any("PerOutput" in str(x.message) for x in w)
    finally:
        evaluate.header_text = orig


def test_fixtures_exist():
    assert len(FIXTURES) >= 4


def test_unexportable_types_are_refused():
    _, layers = load(FIXTURES[0])
    fc = [l for l in layers if l["layer_type"] == "BitLinear"][0]
    fc["quantization_type"] = "5bitsym"
    with pytest.raises(ValueError):
        evaluate.header_text(layers)


@pytest.mark.gpu
@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[3:-4] for p in FIXTURES])
def test_inference_quantized_on_gpu_equals_the_compiled_reference(path, gpu_ok):
    d, layers = load(path)
    qm = FakeQuantizedModel(layers)
    ev = evaluate.attach(qm)
    x = d["x_float"]
    lg = qm.inference_quantized(x)                       # the patched method: float in, C-exact int32 logits out
    assert lg.dtype == np.int32 and np.array_equal(lg, d["logits"])
    assert np.array_equal(ev.predict(x), d["cls"])
    import torch
    assert np.array_equal(qm.inference_quantized(torch.from_numpy(x).view(-1, 1, 16, 16)), d["logits"])
    cls, lg2 = ev.infer_int8(d["x_int8"])
    assert np.array_equal(cls, d["cls"]) and np.array_equal(lg2, d["logits"])
    ev.close()
