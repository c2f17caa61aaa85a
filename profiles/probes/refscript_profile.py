#!/usr/bin/env python3
"""Where does the reference's unmodified test_inference.py spend its time with the product DLL in the process (20 s) vs the
reference's own DLL (8 s)?  cProfile around oracle/refscript.run_script.  usage: refscript_profile.py product|ref"""
import cProfile
import os
import pstats
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "oracle"))
import refscript  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "product"
images, labels = refscript.synthetic_mnist()
stage = refscript.STAGE
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
out = refscript.run_script(os.path.join(stage, "test_inference.py"), os.path.join(stage, kind), stage, images, labels)
pr.disable()
print(kind, "seconds", round(time.perf_counter() - t0, 2), "threads", len(os.listdir("/proc/self/task")))
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
