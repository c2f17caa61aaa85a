"""Run-time loader of the BitNetMCU_model.h interchange format (bitnetmcu_amd/csrc/bnm_model.cpp) — host logic,
no GPU.  Where /root/reference exists, every shipped header is parsed and compared with the committed blobs."""
import os

import numpy as np
import pytest

import util
from util import GOLDEN, MODEL_NAMES, REPO
from bitnetmcu_amd.headerwriter import write_header
from bitnetmcu_amd import Model, BnmError, KIND_CNN, KIND_FC


def same_model(a, b, orders=None):
    assert a.kind == b.kind and a.num_layers == b.num_layers and a.num_classes == b.num_classes
    for i in range(a.num_layers):
        la, lb = a.layer(i), b.layer(i)
        for f, _ in la._fields_:
            if f == "order" and orders is not None:
                assert lb.order == orders[i]
            else:
                assert getattr(la, f) == getattr(lb, f), (i, f)
        assert np.array_equal(a.layer_weights(i), b.layer_weights(i))


@pytest.mark.parametrize("name", MODEL_NAMES)
def test_blob_roundtrip_and_dialects(name):
    m = util.load_golden_model(name)
    same_model(m, Model.from_blob(m.to_blob()))
    same_model(m, Model.from_header_text(write_header(m, "exporter")))
    same_model(m, Model.from_header_text(write_header(m, "oneline")))


def test_layer_names_are_discovered_not_assumed():
    """Today's exporter names FC layers by module index: L3,L5,L7,L9."""
    m = util.load_golden_model("fc_4bitsym_64")
    same_model(m, Model.from_header_text(write_header(m, "exporter", renumber=[3, 5, 7, 9])), orders=[3, 5, 7, 9])


def test_shipped_headers_match_committed_blobs():
    if not util.have_reference():
        pytest.skip("/root/reference not present")
    import sys
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    from build_oracle import REF_MODELS
    for name, hdr in REF_MODELS.items():
        same_model(util.load_golden_model(name), Model.from_header(hdr))


def test_committed_ternary_header_parses():
    m = Model.from_header(os.path.join(GOLDEN, "headers", "tern_96.h"))
    assert m.kind == KIND_FC and [l.n_input for l in m.layers()] == [260, 100, 100, 100]
    assert m.layer(0).weight_elem_bytes == 2 and m.layer(0).weight_count == 96 * 26
    same_model(m, util.load_golden_model("tern_96"))


def test_model_facts():
    m = util.load_golden_model("fc_4bitsym_64")
    assert sum(l.weight_count * 4 for l in m.layers()) == 12608          # docs/documentation.md:488 (100,864 bits)
    c = util.load_golden_model("cnn_64")
    assert c.kind == KIND_CNN and c.layer(0).out_channels == 64 and c.num_classes == 10
    assert util.load_golden_model("mcu_cnn_letters").num_classes == 37


@pytest.mark.parametrize("bad,why", [
    ("", "no Lk_active"),
    ("#define L1_active\n#define L1_bitperweight 4\n", "incomplete"),
    ("#define L1_active\n#define L1_bitperweight 4\n#define L1_incoming_weights 256\n#define L1_outgoing_weights 4\n"
     "const uint32_t L1_weights[] = {0x1,0x2};\n", "too short"),
    ("#define L1_active\n#define L1_bitperweight 4\n#define L1_incoming_weights 100\n#define L1_outgoing_weights 4\n"
     "const uint32_t L1_weights[] = {0x1};\n", "does not match"),
    ("#define L1_active\n#define L1_type Dense\n", "unknown layer type"),
])
def test_malformed_headers_are_rejected(bad, why):
    with pytest.raises(BnmError) as e:
        Model.from_header_text(bad)
    assert why in str(e.value)


def test_bad_blob_rejected():
    with pytest.raises(BnmError):
        Model.from_blob(b"not a blob at all........")
    good = util.load_golden_model("mcu_1k").to_blob()
    with pytest.raises(BnmError):
        Model.from_blob(good[:100])


def test_loader_survives_mutated_headers():
    """Robustness of the run-time header parser: random truncations / byte flips / line drops of a valid header must
    end in a model or in a BnmError — never in a crash (the parser runs inside the caller's process)."""
    rng = np.random.default_rng(123)
    base = write_header(util.load_golden_model("mcu_1k"), "exporter").encode()
    cnn = write_header(util.load_golden_model("mcu_cnn_16small"), "exporter").encode()
    ok = bad = 0
    for it in range(600):
        src = bytearray(base if it % 3 else cnn)
        kind = it % 4
        if kind == 0:
            src = src[:int(rng.integers(0, len(src)))]
        elif kind == 1:
            for _ in range(int(rng.integers(1, 8))):
                src[int(rng.integers(0, len(src)))] = int(rng.integers(0, 256))
        elif kind == 2:
            lines = bytes(src).split(b"\n")
            del lines[int(rng.integers(0, len(lines)))]
            src = bytearray(b"\n".join(lines))
        else:
            a = int(rng.integers(0, len(src)))
            src = src[:a] + src[a:a + int(rng.integers(1, 200))] + src[a:]
        try:
            m = Model.from_header_text(bytes(src))
            assert m.num_layers >= 1 and m.num_classes >= 1
            Model.from_blob(m.to_blob())
            ok += 1
        except BnmError:
            bad += 1
    assert ok + bad == 600 and bad > 100


def test_blob_survives_corruption():
    rng = np.random.default_rng(5)
    good = bytearray(util.load_golden_model("mcu_1k").to_blob())
    for it in range(300):
        b2 = bytearray(good)
        for _ in range(int(rng.integers(1, 6))):
            b2[int(rng.integers(0, 200))] = int(rng.integers(0, 256))     # header + layer table region
        try:
            Model.from_blob(bytes(b2[:int(rng.integers(1, len(b2) + 1))] if it % 2 else b2))
        except BnmError:
            pass


def test_wrong_element_types_are_rejected():
    """a weight array of the right element COUNT but the wrong C type would make the kernels read past
    the uploaded bytes.  Text and blob paths both refuse it."""
    import re
    import struct
    fc = write_header(util.load_golden_model("mcu_1k"), "exporter")
    for bad_type in ("int8_t", "uint16_t", "uint8_t"):
        with pytest.raises(BnmError):
            Model.from_header_text(re.sub(r"const uint32_t (L\d+_weights)", rf"const {bad_type} \1", fc, count=1))
    tern = write_header(util.load_golden_model("tern_96"), "exporter")
    with pytest.raises(BnmError):
        Model.from_header_text(re.sub(r"const uint16_t (L\d+_weights)", r"const uint32_t \1", tern, count=1))
    cnn = write_header(util.load_golden_model("mcu_cnn_16small"), "exporter")
    with pytest.raises(BnmError):
        Model.from_header_text(re.sub(r"const int8_t (L\d+_weights)", r"const uint32_t \1", cnn, count=1))
    # blob: patch weight_elem_bytes / byte counts of the first layer record (header 24 B, then {info 56 B, offset, bytes})
    good = bytearray(util.load_golden_model("mcu_cnn_16small").to_blob())
    info_off = 24
    eb_off = info_off + 12 * 4           # weight_elem_bytes is the 13th uint32 of bnm_layer_info
    for eb, cnt, nbytes in ((0, 9 * 16, 0), (4, 9 * 16, 9 * 16 * 4), (1, 9 * 16, 9 * 16 - 1)):
        b2 = bytearray(good)
        struct.pack_into("<II", b2, eb_off, eb, cnt)
        struct.pack_into("<I", b2, info_off + 56 + 4, nbytes)
        with pytest.raises(BnmError):
            Model.from_blob(bytes(b2))
    Model.from_blob(bytes(good))


def test_staged_reference_headers_are_the_reference_files_and_parse_to_the_zoo_blobs():
    """tests/golden/_ref_headers/ (staged by oracle/build_oracle.py where /root/reference exists, shipped to the GPU box): every
    file has the sha256 its MANIFEST records, is byte-identical to the reference's file where that is present, and parses - through
    the run-time text parser - to the committed weight blob.  bench.py / smoke() load the headline and CNN models from these bytes."""
    import hashlib
    import json
    from bitnetmcu_amd import Model
    d = os.path.join(util.GOLDEN, "_ref_headers")
    if not os.path.isdir(d):
        pytest.skip("no staged reference headers in this tree (a fresh clone that never saw /root/reference)")
    man = json.load(open(os.path.join(d, "MANIFEST.json")))
    assert {"fc_4bitsym_64", "cnn_64"} <= set(man)
    for name, e in man.items():
        raw = open(os.path.join(d, name + ".h"), "rb").read()
        assert hashlib.sha256(raw).hexdigest() == e["sha256"], name
        if os.path.isfile(e["origin"]):
            assert raw == open(e["origin"], "rb").read(), name
        assert Model.from_header(os.path.join(d, name + ".h")).to_blob() == Model.from_zoo(name).to_blob(), name
