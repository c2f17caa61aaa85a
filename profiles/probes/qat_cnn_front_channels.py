import sys, json, numpy as np, torch
sys.path.insert(0, "/root/repo")
from bitnetmcu_amd import qat
for C in (16, 32, 48, 64, 128):
    torch.manual_seed(C)
    m = qat.CNNMNIST(64, 64, 0, cnn_width=C, QuantType="4bitsym").cuda()
    cs = [c for c in m.model if isinstance(c, qat.BitConv2d)]
    n = 500000
    x = torch.randn(n, 1, 16, 16, device="cuda")
    f = lambda: qat.cnn_front_forward(x, [c.weight for c in cs], [c.s for c in cs])
    for _ in range(2): f()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
    ev[0].record()
    for k in range(5):
        f(); ev[k + 1].record()
    torch.cuda.synchronize()
    ms = float(np.median([ev[k].elapsed_time(ev[k + 1]) for k in range(5)]))
    print(C, round(ms, 3), "ms", f"{n / ms * 1e3:.3e} images/s", f"{n * C / ms * 1e3:.3e} channel-images/s")
