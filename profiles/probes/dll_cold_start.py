import time, sys, os
t00 = time.perf_counter()
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import bitnetmcu_amd as b
t0 = time.perf_counter()
lib = b.harness.load_inference_dll(sys.argv[1])
x = b.synth.images(0, 2000, b.DIST_M)
t1 = time.perf_counter()
out = b.harness.run_inference_loop(lib, x[:1])
t2 = time.perf_counter()
out = b.harness.run_inference_loop(lib, x)
t3 = time.perf_counter()
# numpy-heavy work beside the DLL (what the script's Python engine does): is it slower with the HIP runtime in the process?
a = np.random.default_rng(0).standard_normal((64, 256)).astype(np.float32)
t4 = time.perf_counter()
for _ in range(20000):
    (a @ a.T).sum()
t5 = time.perf_counter()
print(f"{sys.argv[1][-40:]}: import {t0-t00:.2f}s dlopen {t1-t0:.3f}s first call {t2-t1:.3f}s 2000 calls {t3-t2:.3f}s numpy loop {t5-t4:.3f}s threads {len(os.listdir('/proc/self/task'))}")
