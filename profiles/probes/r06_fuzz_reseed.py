import sys
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
import test_gpu_qat_model as q, test_gpu_qat_cnn as qc
for s in (21093, 21788, 22205, 40205):
    q.test_fuzz_random_model_shapes(s, True); print("qat", s, "ok")
for s in (20187, 20822, 24079):
    qc.test_fuzz_random_fronts(s, True); print("front", s, "ok")
