#!/usr/bin/env python3
"""Round 6, one-off seeded runs beyond the suite on the final build (GPU box), the suite's own test bodies called directly:
  1. FC / CNN / float-input model fuzz on new seeds (profiles/fuzz_more.py's loop) - the float kernels changed this round;
  2. the whole-model QAT forward on random shapes (tests/test_gpu_qat_model.py::test_fuzz_random_model_shapes);
  3. the resident one-image kernel on random FC models (the FC fuzz's generator): every model it serves, 200 one-image calls on
     synthetic + extreme images against the oracle, interleaved with a batched call.
  4. the one-kernel convolution front of the QAT forward on random channel counts / QuantTypes / magnitudes
     (tests/test_gpu_qat_cnn.py::test_fuzz_random_fronts).
usage: python profiles/r06_fuzz.py [first_seed] [fc] [cnn] [float] [qat] [resident] [front]"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "tests"))
sys.path.insert(0, os.path.join(HERE, ".."))

import util                      # noqa: E402
import bitnetmcu_amd as b        # noqa: E402
import test_gpu_parity as t      # noqa: E402
import test_gpu_qat_model as q   # noqa: E402
import test_gpu_qat_cnn as qc    # noqa: E402


def resident(seed, orc):
    rng = np.random.default_rng(7000 + seed)
    n_layers = int(rng.choice([3, 4]))
    codecs = tuple(int(c) for c in rng.choice([1, 2, 4, 12, 16, 20, 64], size=n_layers))
    need = {1: 32, 2: 16, 4: 8, 12: 8, 20: 8, 16: 4, 64: 8}
    widths = []
    for k in range(1, n_layers):
        g = need[codecs[k]]
        widths.append(int(rng.integers(1, int(rng.choice([64, 128, 192])) // g + 1)) * g)
    n_classes = int(rng.integers(2, 65))
    os.environ["BNM_QUIET"] = "1"
    try:
        model = b.Model.from_header_text(t._random_model_text(rng, codecs, tuple(widths), n_classes))
        ctx = b.Context(model)
    finally:
        os.environ.pop("BNM_QUIET", None)
    try:
        ctx.set_persistent(True)
    except b.BnmError:
        ctx.close()
        return False          # (a model the resident kernel does not serve)
    x = np.concatenate([b.synth.images(seed, 90, b.DIST_U), b.synth.images(seed, 100, b.DIST_M), np.zeros((2, 256), np.int8),
                        np.full((4, 256), -128, np.int8), np.full((4, 256), 127, np.int8)])
    want = util.OracleModel(model, orc).infer(x)
    got = np.array([int(ctx.infer(x[i:i + 1])[0]) for i in range(len(x))], dtype=np.uint32)
    assert ctx.last_kernel == "persistent_inference_kernel"
    assert np.array_equal(got, want), (codecs, widths, n_classes, int((got != want).sum()))
    assert np.array_equal(ctx.infer(x), want)
    assert int(ctx.infer(x[5:6])[0]) == int(want[5])
    ctx.close()
    return True


def main():
    a = [int(v) for v in sys.argv[1:]] + [None] * 7
    first = a[0] if a[0] is not None else 3000
    counts = [a[1] if a[1] is not None else 100, a[2] if a[2] is not None else 60, a[3] if a[3] is not None else 100,
              a[4] if a[4] is not None else 300, a[5] if a[5] is not None else 150, a[6] if a[6] is not None else 200]
    orc = util.load_oracle()
    served = [0]

    def res(seed, _gpu, o):
        served[0] += 1 if resident(seed, o) else 0
    for name, fn, count in (("FC", t.test_fuzz_random_models_every_available_path, counts[0]), ("CNN", t.test_fuzz_random_cnn_models, counts[1]),
                            ("FC float input", t.test_fuzz_fused_float_input_kernel_on_random_models, counts[2]),
                            ("QAT whole-model forward", lambda s, g, o: q.test_fuzz_random_model_shapes(s, g), counts[3]),
                            ("resident one-image kernel", res, counts[4]),
                            ("QAT convolution front", lambda s, g, o: qc.test_fuzz_random_fronts(s, g), counts[5])):
        t0, bad = time.time(), []
        for seed in range(first, first + count):
            try:
                fn(seed, True, orc)
            except Exception as e:      # noqa: BLE001 - report every failing seed, keep going
                bad.append((seed, repr(e)[:300]))
        extra = f" ({served[0]} models served by the resident kernel)" if name.startswith("resident") else ""
        print(f"{name}: seeds {first}..{first + count - 1}: {count - len(bad)} passed, {len(bad)} failed in {time.time() - t0:.1f} s{extra}", flush=True)
        for x in bad:
            print("  FAILED", x, flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
