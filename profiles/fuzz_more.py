"""One-off extension of the seeded fuzz tests (tests/test_gpu_parity.py) to seeds the suite does not run: the same test bodies,
called directly.  python profiles/fuzz_more.py [first_seed] [fc_count] [cnn_count] [float_count] -> one summary line per family on stdout."""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "tests"))
sys.path.insert(0, os.path.join(HERE, ".."))

import util                      # noqa: E402
import test_gpu_parity as t      # noqa: E402


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    n_fc = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    n_cnn = int(sys.argv[3]) if len(sys.argv) > 3 else 150
    orc = util.load_oracle()
    n_f32 = int(sys.argv[4]) if len(sys.argv) > 4 else 0      # (round 5) the float-input fuzz body as well
    for name, fn, count in (("FC", t.test_fuzz_random_models_every_available_path, n_fc), ("CNN", t.test_fuzz_random_cnn_models, n_cnn),
                            ("FC float input", t.test_fuzz_fused_float_input_kernel_on_random_models, n_f32)):
        t0, bad = time.time(), []
        for seed in range(first, first + count):
            try:
                fn(seed, True, orc)
            except Exception as e:      # noqa: BLE001 - report every failing seed, keep going
                bad.append((seed, repr(e)[:300]))
        print(f"{name}: seeds {first}..{first + count - 1}: {count - len(bad)} passed, {len(bad)} failed in {time.time() - t0:.1f} s", flush=True)
        for b in bad:
            print("  FAILED", b, flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
